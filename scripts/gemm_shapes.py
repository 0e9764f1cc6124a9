"""Times the four ViT-L/14 tower GEMM shapes with the library selected by TVC_LIB_PATH (experiment builds)."""
import importlib, os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
pkg = importlib.import_module("multimodal-detection-consistency_amd")
eng = pkg.TVCEngine(); dev = "cuda:0"; torch.manual_seed(0)
J = int(os.environ.get("ROWS", "131584"))
def bench(I, K, epi, iters=6):
    a = (torch.randn(I, K, device=dev) * K ** -0.5).to(torch.bfloat16); b = torch.randn(J, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(I, device=dev) * 0.1
    out = torch.zeros((J, I), device=dev, dtype=torch.bfloat16)
    for _ in range(2): eng.gemm(a, b, bias, epi, out=out)
    torch.cuda.synchronize()
    ts = []
    for r in range(3):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(iters): eng.gemm(a, b, bias, epi, out=out)
        t1.record(); torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1) / iters)
    ms = sorted(ts)[1]
    return ms, 2.0 * I * J * K / ms / 1e9
line = os.path.basename(os.environ.get("TVC_LIB_PATH", "product")) + ": "
for name, (I, K, epi) in {"qkv": (3072, 1024, 1), "out": (1024, 1024, 1), "fc1": (4096, 1024, 2), "fc2": (1024, 4096, 1)}.items():
    ms, tf = bench(I, K, epi)
    line += f" {name} {ms:.3f} ms {tf:5.0f} TF |"
print(line, flush=True)
