"""Square bf16 GEMMs (the shapes the HIP guide quotes its 256^2 templates on) and the tower shapes, same kernel."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
pkg = importlib.import_module("multimodal-detection-consistency_amd")
eng = pkg.TVCEngine(); dev = "cuda:0"; torch.manual_seed(0)
def bench(I, J, K, epi, rnd=True, iters=8):
    mk = (lambda *s: torch.rand(*s, device=dev) * 2 - 1) if rnd else (lambda *s: torch.zeros(*s, device=dev))
    a = mk(I, K).to(torch.bfloat16); b = mk(J, K).to(torch.bfloat16)
    out = torch.zeros((J, I), device=dev, dtype=torch.bfloat16)
    for _ in range(3): eng.gemm(a, b, None, epi, out=out)
    torch.cuda.synchronize()
    ts = []
    for r in range(3):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(iters): eng.gemm(a, b, None, epi, out=out)
        t1.record(); torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1) / iters)
    ms = sorted(ts)[1]
    print(f"I={I} J={J} K={K} epi={epi} {'random' if rnd else 'zeros'}: {ms:.3f} ms  {2.0*I*J*K/ms/1e9:6.0f} TFLOP/s", flush=True)
for n in (4096, 8192):
    bench(n, n, n, 1, True); bench(n, n, n, 1, False)
bench(1024, 131584, 4096, 1); bench(1024, 131072, 4096, 1); bench(4096, 32768, 4096, 1); bench(3072, 131072, 1024, 1)
