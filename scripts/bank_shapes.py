"""Bank search (fast form, no moments) at the single-GPU shape and at the per-shard shapes of the 8-GPU row-sharded layout
(every rank searches ALL ranks' query rows on 1/8 of the rows): same products, different aspect -- does the stage keep its rate?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import tvc_amd as pkg
D, k = 768, 10
for (R, M) in ((1_000_000, 5120), (125_000, 40960), (1_250_000, 40960), (10_000_000, 5120)):
    eng = pkg.TVCEngine()
    bank = pkg.synth.make_bank(R, D, seed=7, device="cuda:0", dtype=torch.bfloat16)
    g = torch.Generator(device="cuda:0").manual_seed(0)
    q = torch.randn((M, D), generator=g, device="cuda:0"); q = q / q.norm(dim=-1, keepdim=True)
    eng.set_bank(bank)
    for _ in range(2): eng.bank_search(q, k, 0.1, want_moments=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): eng.bank_search(q, k, 0.1, want_moments=False)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    try: eng.bank_status(); st = "ok"
    except Exception as e: st = str(e)[:50]
    print(f"R={R:9d} M={M:6d}: {ms:8.2f} ms  {2.0 * R * M * D / ms / 1e9:7.1f} TFLOP/s-equivalent (one product)  bank bytes {R * D * 2 / 1e9:.2f} GB  status {st}", flush=True)
    eng.close(); del bank, q
