"""Diagnostic: where a tile's time goes in GEMM form 4 (build: scripts/build_variant.sh STAMPS -DTVC_RING_STAMPS; run with
TVC_LIB_PATH=gpurun_abl/libtvc_STAMPS.so).  Per wave group and tile, shader clocks of: the first K-tile (which follows the
previous tile's epilogue), the mean other K-tile, the epilogue (issue of its instructions) and the tile-end barrier."""
import ctypes as C, importlib, os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
pkg = importlib.import_module("multimodal-detection-consistency_amd")
eng = pkg.TVCEngine(); dev = "cuda:0"; torch.manual_seed(0)
J = int(os.environ.get("ROWS", "131072"))
def run(I, K, epi, name):
    a = (torch.randn(I, K, device=dev) * K ** -0.5).to(torch.bfloat16); b = torch.randn(J, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(I, device=dev) * 0.1
    out = torch.zeros((J, I), device=dev, dtype=torch.bfloat16)
    for _ in range(4): eng.gemm(a, b, bias, epi, out=out)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); eng.gemm(a, b, bias, epi, out=out); t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1)
    buf = np.zeros(256 * 8 * 4, dtype=np.uint64)
    assert eng.lib.tvc_debug_ring4_tile_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
    st = buf.reshape(256, 8, 4).astype(np.float64)
    tiles = (I // 256) * (J // 256) / 256; nkt = K // 64
    print(f"{name}: {ms:.3f} ms, {2*I*J*K/ms/1e9:.0f} TFLOP/s, {tiles:.2f} tiles/WG of {nkt} K-tiles; wall {ms*1e3/tiles:.1f} us/tile")
    for g, sl in (("group0", slice(0, 4)), ("group1", slice(4, 8))):
        m = st[:, sl, :].mean(axis=(0, 1)) / tiles
        print(f"  {g}: first K-tile {m[0]:.0f}  other K-tiles {m[1] / max(nkt - 1, 1):.0f} each  epilogue {m[2]:.0f}  tile-end barrier {m[3]:.0f}  | tile {m.sum():.0f} clk")
run(3072, 1024, 1, "qkv")
run(1024, 1024, 1, "out")
run(4096, 1024, 2, "fc1")
run(1024, 4096, 1, "fc2")
