"""Correctness + bit-identity check of a ring GEMM form (TVC_GEMM_RING_FORM, read once per process): prints per shape the
max error vs torch and a checksum of the raw output bits (equal checksums across forms = bit-identical results)."""
import importlib, os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
pkg = importlib.import_module("multimodal-detection-consistency_amd")
eng = pkg.TVCEngine()
g = torch.Generator(device="cuda:0").manual_seed(11)
print("form", os.environ.get("TVC_GEMM_RING_FORM", "default"), flush=True)
for I, J, K, epi in ((1024, 131072, 1024, 1), (3072, 45056, 1024, 1), (4096, 33024, 1024, 2), (1024, 133120, 4096, 1),
                     (1024, 131072, 64, 1), (512, 262144, 128, 0), (768, 174080, 768, 2), (2304, 58368, 768, 1)):
    a = (torch.randn(I, K, device="cuda:0", generator=g) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(J, K, device="cuda:0", generator=g).to(torch.bfloat16)
    bias = torch.randn(I, device="cuda:0", generator=g) * 3.0
    out = eng.gemm(a, b, bias, epi)
    torch.cuda.synchronize()
    o2 = eng.gemm(a, b, bias, epi)
    assert torch.equal(out, o2), "not deterministic"
    # reference on a slice of rows (the full fp32 product of the big shapes is large)
    rows = torch.cat([torch.arange(0, 512), torch.arange(J // 2, J // 2 + 512), torch.arange(J - 512, J)]).cuda()
    ref = b[rows].float() @ a.float().t() + bias
    if epi == 2:
        ref = ref * torch.sigmoid(1.702 * ref)
    err = (out[rows].float() - ref).abs().max().item()
    bits = out.view(torch.int16 if out.dtype == torch.bfloat16 else torch.int32).to(torch.int64)
    chk = int((bits * (torch.arange(bits.numel(), device="cuda:0").view(bits.shape) % 1000003 + 1)).sum().item())
    ok = err < 1e-2 * (1 + ref.abs().max().item())
    print(f"I={I} J={J} K={K} epi={epi}: max err {err:.3e} {'OK' if ok else 'BAD'}  checksum {chk}", flush=True)
    assert ok
print("FORM_OK")
