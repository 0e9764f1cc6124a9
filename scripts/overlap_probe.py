"""Probe: can an HBM-bound row kernel (LayerNorm) run UNDER a persistent ring GEMM on a second stream?  (The GEMM holds
132 KiB of LDS and 416 of 512 VGPRs per SIMD on every CU; the LayerNorm needs no LDS and few registers.)  Prints the times
of each alone, back to back on one stream, and concurrently on two streams."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
pkg = importlib.import_module("multimodal-detection-consistency_amd")
eng = pkg.TVCEngine(); dev = "cuda:0"; torch.manual_seed(0)
J, I, K = 65536, 4096, 1024
a = (torch.randn(I, K, device=dev) * K ** -0.5).to(torch.bfloat16); b = torch.randn(J, K, device=dev).to(torch.bfloat16)
bias = torch.randn(I, device=dev) * 0.1; out = torch.zeros((J, I), device=dev, dtype=torch.bfloat16)
x = torch.randn(J, 1024, device=dev); g = torch.ones(1024, device=dev); bb = torch.zeros(1024, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def gemm(n=4):
    for _ in range(n): eng.gemm(a, b, bias, 2, out=out)
def ln(n=12):
    for _ in range(n): eng.layernorm(x, g, bb)
def timed(f):
    for _ in range(2): f()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); f(); t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1)
def both():
    e0 = torch.cuda.Event(); e0.record()
    with torch.cuda.stream(s1):
        s1.wait_event(e0); gemm()
    with torch.cuda.stream(s2):
        s2.wait_event(e0); ln()
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
tg, tl = timed(gemm), timed(ln)
ts = timed(lambda: (gemm(), ln()))
tb = timed(both)
print(f"gemm alone {tg:.3f} ms, layernorm alone {tl:.3f} ms, one stream {ts:.3f} ms, two streams {tb:.3f} ms (max {max(tg, tl):.3f}, sum {tg + tl:.3f})")
