"""Per-category time of one configs[2]-shaped step in a tower precision mode (TVC_PREC = fp32 | split | bf16): what the
parity mode costs and where (GEMM / attention / row kernels), via the in-process HIP-event profile."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import tvc_amd as pkg
prec = os.environ.get("TVC_PREC", "fp32")
B, N = int(os.environ.get("PB", "512")), 8
arch = pkg.get_arch("ViT-L/14")
w = pkg.synth.make_clip_weights(arch, seed=0)
eng = pkg.TVCEngine(arch, w[0], w[1], precision=prec)
images = pkg.synth.make_images(B, arch.image_size, seed=1).cuda()
tokens = pkg.synth.make_tokens(B, N, arch.ctx, seed=2).cuda().view(-1, arch.ctx)
def step():
    ft = eng.encode_text(tokens, group=N + 1)
    fi = eng.encode_image(images)
    return fi, ft
step(); torch.cuda.synchronize()
t0 = time.perf_counter(); step(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
eng.profile_begin(); step(); p = eng.profile_end()
print(f"precision={prec} B={B} N={N}: towers {dt * 1e3:.1f} ms per step;", {k: (round(v['ms'], 1), v['launches']) for k, v in p.items()},
      "GEMM TFLOP/s", round(p['gemm']['work'] / p['gemm']['ms'] / 1e9, 1), flush=True)
eng.close()
