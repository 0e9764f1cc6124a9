"""Reads the in-kernel clock stamps of an attention STAMP build (experiment library, TVC_LIB_PATH): per wave
[start, fill issued+written, barrier passed, then per query block: start, QK+max done, exp done, PV done] and the wave's
HW_ID / XCC_ID, written over output rows 100+4w of every (sequence, head) item."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import tvc_amd as pkg
eng = pkg.TVCEngine(); dev = "cuda:0"; torch.manual_seed(0)
n_seq, T, H = 512, 257, 16
qkv = torch.randn((n_seq * T, 3 * H * 64), device=dev).to(torch.bfloat16)
for _ in range(3): out = eng.attention(qkv, n_seq, T, H, False)
torch.cuda.synchronize()
o = out.view(torch.int16).cpu().numpy().reshape(n_seq, T, H, 64)
recs = []
for s in range(n_seq):
    for h in range(H):
        for w in range(4):
            raw = np.concatenate([o[s, 100 + 4 * w, h], o[s, 100 + 4 * w + 1, h]]).view(np.uint64)
            nst = int(raw[31]); hw = int(raw[30])
            recs.append((s * H + h, w, hw & 0xffffffff, hw >> 32, raw[:nst].astype(np.int64)))
t0 = min(r[4][0] for r in recs)
tend = max(r[4][-1] for r in recs)
print("kernel span (clock ticks):", tend - t0)
# group by CU: (xcc, se, sh?, cu)
def cu_key(hwid, xcc): return (xcc & 0xf, (hwid >> 13) & 7, (hwid >> 12) & 1, (hwid >> 8) & 0xf)
by_cu = {}
for item, w, hwid, xcc, st in recs:
    by_cu.setdefault(cu_key(hwid, xcc), []).append((st[0] - t0, item, w, (hwid >> 4) & 3, st - t0))
print("distinct CUs:", len(by_cu))
key = sorted(by_cu)[len(by_cu) // 2]
rows = sorted(by_cu[key], key=lambda r: r[0])
print("CU", key, "waves:", len(rows))
for start, item, w, simd, st in rows[:16]:
    print(f"item {item:5d} w{w} simd{simd}  " + " ".join(f"{int(x):7d}" for x in st))
# phase statistics over all waves with 4 blocks (w != 0)
d = {"fill": [], "barrier": [], "qk": [], "exp": [], "pv": [], "tail": [], "block": []}
for item, w, hwid, xcc, st in recs:
    d["fill"].append(st[1] - st[0]); d["barrier"].append(st[2] - st[1])
    nb = (len(st) - 4) // 4
    for b in range(nb):
        a = st[3 + 4 * b: 3 + 4 * b + 5]
        d["qk"].append(a[1] - a[0]); d["exp"].append(a[2] - a[1]); d["pv"].append(a[3] - a[2]); d["tail"].append(a[4] - a[3] if 3 + 4 * b + 4 < len(st) else 0)
        d["block"].append(a[4] - a[0])
for k, v in d.items():
    v = np.array(v); print(f"{k:8s} mean {v.mean():9.1f}  p10 {np.percentile(v,10):9.1f}  p50 {np.percentile(v,50):9.1f}  p90 {np.percentile(v,90):9.1f}")
