import sys
sys.path.insert(0, ".")
import torch
import tvc_amd as pkg
eng = pkg.TVCEngine(); dev = "cuda:0"
def run(I, J, K, epi):
    g = torch.Generator().manual_seed(1)
    a = (torch.randn(I, K, generator=g) * K ** -0.5).to(torch.bfloat16); b = torch.randn(J, K, generator=g).to(torch.bfloat16)
    bias = torch.randn(I, generator=g) * 0.1
    ref = b.float() @ a.float().t() + bias
    out = eng.gemm(a.to(dev), b.to(dev), bias.to(dev), epi).float().cpu()
    err = (out - ref).abs()
    bad = err > 1e-2 * (1 + ref.abs().max())
    print(f"I={I} J={J} K={K} epi={epi}: max err {err.max():.4f}, bad {int(bad.sum())} of {bad.numel()}")
    if bad.any():
        jj, ii = bad.nonzero(as_tuple=True)
        print("  bad token rows:", sorted(set((jj // 16 * 16).tolist()))[:20], " feature cols:", sorted(set((ii // 16 * 16).tolist()))[:20])
for shape in ((768, 1000, 640, 1), (768, 1024, 640, 2), (1024, 2048, 1024, 1), (256, 2048, 256, 1), (1024, 131072, 1024, 1)):
    run(*shape)
