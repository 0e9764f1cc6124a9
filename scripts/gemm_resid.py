"""Is the ring GEMM sensitive to where the token operand comes from?  Same per-tile work at J = 131072
(operand streamed from HBM) and J = 16384 (operand + output stay in the 256 MiB Infinity Cache)."""
import sys
sys.path.insert(0, ".")
import torch
import tvc_amd as pkg
eng = pkg.TVCEngine(); dev = "cuda:0"; torch.manual_seed(0)
def bench(I, J, K, epi, iters=8):
    a = (torch.randn(I, K, device=dev) * K ** -0.5).to(torch.bfloat16); b = torch.randn(J, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(I, device=dev) * 0.1
    out = torch.zeros((J, I), device=dev, dtype=torch.bfloat16)
    for _ in range(3): eng.gemm(a, b, bias, epi, out=out)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters): eng.gemm(a, b, bias, epi, out=out)
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / iters
    return ms, 2.0 * I * J * K / ms / 1e9
for name, I, K, epi in (("fc1", 4096, 1024, 2), ("qkv", 3072, 1024, 1), ("fc2", 1024, 4096, 1), ("proj", 1024, 1024, 1)):
    for J in (131072, 32768, 16384):
        ms, tf = bench(I, J, K, epi)
        print(f"{name} J={J:6d}: {ms:.3f} ms  {tf:7.1f} TF/s")
