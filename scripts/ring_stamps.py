"""Diagnostic: per-phase shader-clock totals of the ring GEMM loop (build with -DTVC_RING_STAMPS,
TVC_LIB_PATH=gpurun_abl/libtvc_STAMPS.so).  Phases: 0 loop top (SALU), 1 issue LDS-DMA, 2 ds_read + wait,
3 MFMA issue, 4 finish (epilogue at tile ends), 5 vmcnt wait + barrier."""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np, torch
import tvc_amd as pkg
eng = pkg.TVCEngine(); dev = "cuda:0"; torch.manual_seed(0)
def run(I, J, K, epi, name):
    a = (torch.randn(I, K, device=dev) * K ** -0.5).to(torch.bfloat16); b = torch.randn(J, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(I, device=dev) * 0.1
    out = torch.zeros((J, I), device=dev, dtype=torch.float32 if epi in (0, 3) else torch.bfloat16)
    for _ in range(3): eng.gemm(a, b, bias, epi, out=out)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); eng.gemm(a, b, bias, epi, out=out); t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1)
    buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
    rc = eng.lib.tvc_debug_ring_stamps(buf.ctypes.data_as(C.c_void_p)); assert rc == 0
    st = buf.reshape(256, 8, 8).astype(np.float64)
    ntiles = (I // 256) * (J // 256); stages = ntiles / 256 * (K // 32)
    print(f"{name}: {ms:.3f} ms, {stages:.0f} stages/WG, wall {ms*1e6/stages:.0f} ns/stage")
    for g, sl in (("group0 (load->mfma)", slice(0, 4)), ("group1 (mfma->load)", slice(4, 8))):
        m = st[:, sl, :].mean(axis=(0, 1)) / stages
        print(f"  {g}: top {m[0]:.0f}  issue {m[1]:.0f}  dsread+wait {m[2]:.0f}  mfma {m[3]:.0f}  finish {m[4]:.0f}  vmwait+barrier {m[5]:.0f}  | sum {m[:6].sum():.0f} clk/stage")
run(4096, 131072, 1024, 2, "fc1")
run(3072, 131072, 1024, 1, "qkv")
run(1024, 131072, 4096, 1, "fc2")
buf = np.zeros(4 * 512, dtype=np.uint64)
assert eng.lib.tvc_debug_ring_trace(buf.ctypes.data_as(C.c_void_p)) == 0
tr = buf.reshape(4, 512).astype(np.int64)
for r, name in enumerate(("blk8 g0", "blk8 g1", "blk100 g0", "blk100 g1")):
    d = np.diff(tr[r][:300])
    print(name, "per-stage cycles (last gemm = fc2, 128 stages/tile):")
    print("  ", " ".join(str(int(x)) for x in d[:280]))
