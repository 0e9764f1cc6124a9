import os, sys
sys.path.insert(0, ".")
import torch
import tvc_amd as pkg
eng = pkg.TVCEngine(); dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
out = {}
for (I, J, K, epi) in [(768, 12800, 768, 1), (3072, 12800, 768, 2), (768, 12800, 3072, 1), (768, 2000, 768, 0)]:
    a = (torch.randn(I, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(J, K, device=dev, generator=g).to(torch.bfloat16)
    bias = torch.randn(I, device=dev, generator=g) * 0.1
    out[(I, J, K, epi)] = eng.gemm(a, b, bias, epi).float().cpu()
torch.save(out, sys.argv[1])
