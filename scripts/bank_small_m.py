"""Bank search in the HBM-bound regime of SURVEY.md 8(d) (small query batches: the reference searches ONE query at a time,
src/retrieval.py:636-680; configs[0] has M = 48): ms per search and GB/s = R * D * 2 B / t at M in {1, 10, 48, 256}, R = 1 M
and 10 M bf16 rows (fast form, no moments -- the detection path).  Also prints the in-process category split."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import tvc_amd as pkg
D, k = 768, 10
Rs = [int(r) for r in os.environ.get("BANK_ROWS", "1000000,10000000").split(",")]
for R in Rs:
    eng = pkg.TVCEngine()
    bank = pkg.synth.make_bank(R, D, seed=7, device="cuda:0", dtype=torch.bfloat16)
    eng.set_bank(bank)
    for M in [int(m) for m in os.environ.get("BANK_MS", "1,10,48,256,1024").split(",")]:
        g = torch.Generator(device="cuda:0").manual_seed(M)
        q = torch.randn((M, D), generator=g, device="cuda:0"); q = q / q.norm(dim=-1, keepdim=True)
        for _ in range(3): eng.bank_search(q, k, 0.1, want_moments=False)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        a.record()
        for _ in range(n): eng.bank_search(q, k, 0.1, want_moments=False)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / n
        try: eng.bank_status(); st = "ok"
        except Exception as e: st = str(e)[:50]
        # exactness spot check against a dense fp32 matmul of the same bf16 rows
        idx, sim, _ = eng.bank_search(q, k, 0.1, want_moments=False)
        ref = (q[: min(M, 4)] @ bank.float().T)
        rs, ri = ref.topk(k, dim=1)
        ok = bool((idx[: min(M, 4)].long() == ri).all()) and float((sim[: min(M, 4)] - rs).abs().max()) < 1e-5
        print(f"R={R:9d} M={M:5d}: {ms:8.3f} ms  bank stream {R * D * 2 / ms / 1e6:8.1f} GB/s ({R * D * 2 / ms / 1e6 / 6300:.3f} of 6.3 TB/s)  "
              f"status {st}  top-k {'exact' if ok else 'MISMATCH'}", flush=True)
        del ref
    eng.close(); del bank
