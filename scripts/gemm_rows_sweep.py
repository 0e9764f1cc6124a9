"""Times out[J, I] GEMMs (K = 1024, bf16 store) over the out-feature count I: how the tile walk copes with tile-row counts
that do not divide a round (library selected by TVC_LIB_PATH; ROWS = token rows)."""
import importlib, os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
pkg = importlib.import_module("multimodal-detection-consistency_amd")
eng = pkg.TVCEngine(); dev = "cuda:0"; torch.manual_seed(0)
J = int(os.environ.get("ROWS", "131072")); K = 1024
line = os.path.basename(os.environ.get("TVC_LIB_PATH", "product")) + f" J={J}:"
for I in (1024, 2048, 2560, 3072, 3584, 4096):
    a = (torch.randn(I, K, device=dev) * K ** -0.5).to(torch.bfloat16); b = torch.randn(J, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(I, device=dev) * 0.1
    out = torch.zeros((J, I), device=dev, dtype=torch.bfloat16)
    for _ in range(2): eng.gemm(a, b, bias, 1, out=out)
    torch.cuda.synchronize()
    ts = []
    for r in range(3):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(6): eng.gemm(a, b, bias, 1, out=out)
        t1.record(); torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1) / 6)
    ms = sorted(ts)[1]
    line += f" I={I} {2.0 * I * J * K / ms / 1e9:5.0f} TF |"
print(line, flush=True)
