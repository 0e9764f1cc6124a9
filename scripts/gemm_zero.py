"""Same GEMM on random and on all-zero operands: the gap is what the chip gives back as clock when
the data toggles less (DVFS), i.e. how much of the remaining headroom is power, not schedule."""
import sys
sys.path.insert(0, ".")
import torch
import tvc_amd as pkg
eng = pkg.TVCEngine(); dev = "cuda:0"; torch.manual_seed(0)
def bench(I, J, K, epi, zero, iters=8):
    if zero:
        a = torch.zeros(I, K, device=dev, dtype=torch.bfloat16); b = torch.zeros(J, K, device=dev, dtype=torch.bfloat16)
    else:
        a = (torch.randn(I, K, device=dev) * K ** -0.5).to(torch.bfloat16); b = torch.randn(J, K, device=dev).to(torch.bfloat16)
    bias = torch.zeros(I, device=dev)
    out = torch.zeros((J, I), device=dev, dtype=torch.bfloat16)
    for _ in range(3): eng.gemm(a, b, bias, epi, out=out)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters): eng.gemm(a, b, bias, epi, out=out)
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / iters
    return ms, 2.0 * I * J * K / ms / 1e9
for name, I, K, epi in (("fc1", 4096, 1024, 2), ("qkv", 3072, 1024, 1), ("fc2", 1024, 4096, 1)):
    r = bench(I, 131072, K, epi, False); z = bench(I, 131072, K, epi, True)
    print(f"{name}: random {r[0]:.3f} ms {r[1]:.0f} TF/s | zeros {z[0]:.3f} ms {z[1]:.0f} TF/s | x{r[0]/z[0]:.2f}")
