"""sd_flash_attention_kernel at the UNet's shapes (24 samples = 12 images x CFG): time, TFLOP/s on the real head dim."""
import importlib, sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
pkg = importlib.import_module("multimodal-detection-consistency_amd")
eng = pkg.TVCEngine()
k = pkg.SDKernels.__new__(pkg.SDKernels); k.engine = eng; k.arch = pkg.SDArch()
n, heads = 24, 8
for (dh, Tq, Tk) in ((32, 4096, 4096), (40, 4096, 4096), (80, 1024, 1024), (160, 256, 256), (160, 64, 64), (40, 4096, 77), (80, 1024, 77)):
    C = heads * dh
    q = torch.randn((n * Tq, C), device="cuda").to(torch.bfloat16)
    kk = torch.randn((n * Tk, C), device="cuda").to(torch.bfloat16)
    v = torch.randn((n * Tk, C), device="cuda").to(torch.bfloat16)
    for _ in range(3):
        k.attention(q, kk, v, n, heads)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        k.attention(q, kk, v, n, heads)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    fl = 4.0 * n * heads * Tq * Tk * dh
    print(f"dh={dh:3d} Tq={Tq:4d} Tk={Tk:4d}: {ms*1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
