"""Time the two towers separately (serial) at the bench configuration, with the per-category split."""
import sys, time
sys.path.insert(0, ".")
import torch
import tvc_amd as pkg
arch = pkg.arch.get_arch("ViT-L/14"); dev = torch.device("cuda:0")
w = pkg.synth.make_clip_weights(arch, seed=0)
eng = pkg.TVCEngine(arch, w[0], w[1], device="cuda:0")
B, N = 512, 8
img = pkg.synth.make_images(B, arch.image_size, seed=1).to(dev)
tok = pkg.synth.make_tokens(B, N, arch.ctx, seed=2).to(dev).view(B * (N + 1), arch.ctx)
for name, fn in (("image", lambda: eng.encode_image(img, True)), ("text", lambda: eng.encode_text(tok, True))):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
    eng.profile_begin(); fn(); p = eng.profile_end()
    print(name, f"{ms:.1f} ms", {k: (round(v["ms"], 2), v["launches"], round(v["work"] / max(v["ms"], 1e-9) / 1e9, 0)) for k, v in p.items()})
