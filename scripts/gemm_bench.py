"""GPU micro-benchmark of tvc_gemm_bf16 on the ViT-L/14 tower shapes (not part of the product).
TVC_GEMM_VARIANT=0|1 selects the main loop (one tile per workgroup | persistent ring)."""
import os, sys
sys.path.insert(0, ".")
import torch
import tvc_amd as pkg

eng = pkg.TVCEngine()
torch.manual_seed(0)
dev = "cuda:0"

def check(I, J, K, epi):
    a = (torch.randn(I, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(J, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(I, device=dev) * 0.1
    ref = b.float() @ a.float().t() + bias
    if epi == 3:
        base = torch.randn(J, I, device=dev)
        out = base.clone()
        eng.gemm(a, b, bias, 3, out=out)
        ref = base + ref
    else:
        out = eng.gemm(a, b, bias, epi)
    if epi == 2:
        ref = ref * torch.sigmoid(1.702 * ref)
    err = (out.float() - ref).abs().max().item()
    return err

def bench(I, J, K, epi, iters=8):
    a = (torch.randn(I, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(J, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(I, device=dev) * 0.1
    out = torch.zeros((J, I), device=dev, dtype=torch.float32 if epi in (0, 3) else torch.bfloat16)
    for _ in range(2):
        eng.gemm(a, b, bias, epi, out=out)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        eng.gemm(a, b, bias, epi, out=out)
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / iters
    return ms, 2.0 * I * J * K / ms / 1e9

print("variant", os.environ.get("TVC_GEMM_VARIANT", "auto"))
for (I, J, K) in [(512, 4096 + 77, 192), (768, 2048 * 3 + 5, 1024)]:
    for epi in range(4):
        print(f"  check I={I} J={J} K={K} epi={epi}: max err {check(I, J, K, epi):.3e}")
rows_v, rows_t = 131584, 76000
for name, (I, J, K, epi) in {
    "v.qkv": (3072, rows_v, 1024, 1), "v.out": (1024, rows_v, 1024, 3), "v.fc1": (4096, rows_v, 1024, 2),
    "v.fc2": (1024, rows_v, 4096, 3), "t.qkv": (2304, rows_t, 768, 1), "t.fc1": (3072, rows_t, 768, 2),
    "t.fc2": (768, rows_t, 3072, 3)}.items():
    ms, tf = bench(I, J, K, epi)
    print(f"  {name:6s} I={I:5d} J={J:6d} K={K:5d} epi={epi}: {ms:7.3f} ms  {tf:7.1f} TFLOP/s", flush=True)
