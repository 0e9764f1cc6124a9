"""Calibration of FETCH_SIZE / WRITE_SIZE for the ring GEMM's access pattern (LDS-DMA reads of 64-B row
segments) on a KNOWN byte count, as /opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes before trusting an
absolute.  One weight tile (I = 256: every token tile is read exactly once), K = 1024, J = 2^20 token rows:
  read  = J * K * 2 B = 2.147 GB (+ 0.5 MB of weights, L2-resident)      write = J * 256 * 2 B = 0.537 GB
Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` / `--pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum`;
the per-dispatch counter of gemm_ring_kernel<1> is then compared with these numbers."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
pkg = importlib.import_module("multimodal-detection-consistency_amd")
eng = pkg.TVCEngine()
J, K, I = 1 << 20, 1024, 256
g = torch.Generator(device="cuda").manual_seed(0)
a = (torch.randn((I, K), device="cuda", generator=g) * 0.05).to(torch.bfloat16)
b = (torch.randn((J, K), device="cuda", generator=g)).to(torch.bfloat16)
out = torch.empty((J, I), dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    eng.gemm(a, b, None, 1, out=out)
torch.cuda.synchronize()
print("calibration: read bytes", J * K * 2 + I * K * 2, "write bytes", J * I * 2)
