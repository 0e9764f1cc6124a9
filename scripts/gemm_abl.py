"""Ablation timing of the ring GEMM (experiment builds; outputs are wrong by construction)."""
import os, sys
sys.path.insert(0, ".")
import torch
import tvc_amd as pkg
eng = pkg.TVCEngine(); dev = "cuda:0"; torch.manual_seed(0)
def bench(I, J, K, epi, iters=6):
    a = (torch.randn(I, K, device=dev) * K ** -0.5).to(torch.bfloat16); b = torch.randn(J, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(I, device=dev) * 0.1
    out = torch.zeros((J, I), device=dev, dtype=torch.float32 if epi in (0, 3) else torch.bfloat16)
    for _ in range(2): eng.gemm(a, b, bias, epi, out=out)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters): eng.gemm(a, b, bias, epi, out=out)
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters
print(os.environ.get("TVC_LIB_PATH", "full"), " fc2 %.3f ms  fc1 %.3f ms  qkv %.3f ms  proj %.3f ms" % (
    bench(1024, 131072, 4096, 1), bench(4096, 131072, 1024, 2), bench(3072, 131072, 1024, 1), bench(1024, 131072, 1024, 1)))
