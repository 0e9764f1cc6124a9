"""Does the row stride of the GEMM operands matter (L2 / HBM channel distribution of the 64-B row segments
the LDS-DMA stream reads)?  The tower shapes at dense strides (ld = K: every row starts at a multiple of
2 / 8 KiB) against padded strides (ld = K + pad).  Interleaved rounds, one process."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
pkg = importlib.import_module("multimodal-detection-consistency_amd")
eng = pkg.TVCEngine(); dev = "cuda:0"; torch.manual_seed(0)
J = 131584
shapes = {"qkv": (3072, 1024, 1), "out": (1024, 1024, 1), "fc1": (4096, 1024, 2), "fc2": (1024, 4096, 1)}
pads = [0, 64, 128, 192, 320]
def mk(rows, K, pad, scale):
    t = (torch.randn(rows, K + pad, device=dev) * scale).to(torch.bfloat16)
    return t
res = {}
for name, (I, K, epi) in shapes.items():
    bufs = {}
    for pa in pads:
        a = mk(I, K, pa, K ** -0.5); b = mk(J, K, pa, 1.0)
        out = torch.empty((J, I), dtype=torch.bfloat16, device=dev)
        bias = torch.randn(I, device=dev) * 0.1
        bufs[pa] = (a, b, bias, out)
    for pa in pads:
        a, b, bias, out = bufs[pa]
        for _ in range(2): eng.gemm(a, b, bias, epi, out=out, k=K)
    torch.cuda.synchronize()
    times = {pa: [] for pa in pads}
    for rnd in range(5):
        for pa in pads:
            a, b, bias, out = bufs[pa]
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(3): eng.gemm(a, b, bias, epi, out=out, k=K)
            t1.record(); torch.cuda.synchronize()
            times[pa].append(t0.elapsed_time(t1) / 3)
    line = f"{name:4s} I={I} K={K}: "
    for pa in pads:
        ms = sorted(times[pa])[len(times[pa]) // 2]
        line += f" pad{pa}: {ms:.3f} ms {2.0 * I * J * K / ms / 1e9:6.0f} TF |"
    print(line, flush=True)
    del bufs
# only ONE operand padded (which one matters?)
for name, (I, K, epi) in shapes.items():
    out = torch.empty((J, I), dtype=torch.bfloat16, device=dev); bias = torch.randn(I, device=dev) * 0.1
    combos = {"a0b0": (0, 0), "a64b0": (64, 0), "a0b64": (0, 64)}
    # different row strides for a and b need the explicit entry: emulate by padding both and passing k (strides differ per tensor)
    line = f"{name:4s} single-operand: "
    for cn, (pa, pb) in combos.items():
        a = mk(I, K, pa, K ** -0.5); b = mk(J, K, pb, 1.0)
        import ctypes as C
        def run():
            eng._check(eng.lib.tvc_gemm_bf16(eng.handle, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(bias.data_ptr()),
                                             C.c_void_p(out.data_ptr()), I, J, K, K + pa, K + pb, I, epi, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        for _ in range(2): run()
        torch.cuda.synchronize()
        ts = []
        for rnd in range(4):
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(3): run()
            t1.record(); torch.cuda.synchronize(); ts.append(t0.elapsed_time(t1) / 3)
        ms = sorted(ts)[len(ts) // 2]
        line += f" {cn}: {ms:.3f} ms {2.0 * I * J * K / ms / 1e9:6.0f} TF |"
    print(line, flush=True)
