"""GPU micro-benchmark of tvc_attention at the ViT-L/14 shape (not part of the product)."""
import sys
sys.path.insert(0, ".")
import torch
import tvc_amd as pkg
eng = pkg.TVCEngine(); dev = "cuda:0"; torch.manual_seed(0)
def bench(n_seq, T, heads, causal, iters=5):
    qkv = torch.randn((n_seq * T, 3 * heads * 64), device=dev).to(torch.bfloat16)
    for _ in range(2): eng.attention(qkv, n_seq, T, heads, causal)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters): eng.attention(qkv, n_seq, T, heads, causal)
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / iters
    fl = 4.0 * n_seq * heads * T * T * 64 * (0.5 if causal else 1.0)
    print(f"n_seq={n_seq} T={T} heads={heads} causal={causal}: {ms*1e3:.0f} us  {fl/ms/1e9:.1f} TFLOP/s", flush=True)
bench(512, 257, 16, False)
bench(4608, 77, 12, True)
bench(4608, 17, 12, True)
