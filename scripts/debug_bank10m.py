import sys
sys.path.insert(0, ".")
import torch
import tvc_amd as pkg
R, D, M, k = 10_000_000, 768, 5120, 5
eng = pkg.TVCEngine()
bank = pkg.synth.make_bank(R, D, seed=7, device="cuda:0", dtype=torch.bfloat16)
g = torch.Generator(device="cuda:0").manual_seed(5)
q = torch.randn((M, D), device="cuda:0", generator=g); q = q / q.norm(dim=-1, keepdim=True)
ns, stride = 65536, R // 65536
sample = bank[::stride][:ns].float()
qs = q[:64]
S0 = qs @ sample.t()
gm = S0.view(64, ns // 256, 256).max(1).values          # group t = i % 256
tau = gm.sort(1, descending=True).values[:, 15]
cnt = torch.zeros(64, device="cuda:0")
cnt_chunk_max = torch.zeros(64, device="cuda:0")
chunk_rows = (39063 + 63) // 64 * 256
for lo in range(0, R, chunk_rows):
    S = qs @ bank[lo:lo + chunk_rows].float().t()
    c = (S > (tau - 0.0012)[:, None]).sum(1).float()
    cnt += c; cnt_chunk_max = torch.maximum(cnt_chunk_max, c)
print("tau", tau[:8].tolist())
print("survivors per query: mean %.0f max %.0f ; per-chunk max %.0f" % (cnt.mean().item(), cnt.max().item(), cnt_chunk_max.max().item()))
eng.set_bank(bank)
for m in (64, 1024, 5120):
    i, s, _ = eng.bank_search(q[:m], k, want_moments=False)
    try:
        eng.bank_status(); print(m, "ok")
    except Exception as e:
        print(m, "overflow", str(e)[:80])
