"""GEMM rate of the latent-diffusion model's 3 x 3 convolutions at their real shapes (24 UNet samples = 12 images x CFG, 12 VAE
images), from the engine's in-process GEMM category: useful TFLOP/s (2 * 9 * Cin * Cout * tokens) per shape."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("multimodal-detection-consistency_amd")
arch = pkg.SDArch()
uw, vw = pkg.make_sd_weights(arch, seed=0)
eng = pkg.TVCEngine()
k = pkg.SDKernels(eng, arch, uw, vw)
cases = [("down_blocks.0.resnets.0.conv1.", 320, 320, 64, 24, False), ("up_blocks.3.resnets.1.conv1.", 640, 320, 64, 24, False),
         ("down_blocks.1.resnets.1.conv1.", 640, 640, 32, 24, False), ("up_blocks.2.resnets.0.conv1.", 1280, 640, 32, 24, False),
         ("down_blocks.2.resnets.1.conv1.", 1280, 1280, 16, 24, False), ("up_blocks.1.resnets.0.conv1.", 2560, 1280, 16, 24, False),
         ("down_blocks.3.resnets.0.conv1.", 1280, 1280, 8, 24, False),
         ("decoder.mid_block.resnets.0.conv1.", 512, 512, 64, 12, True), ("decoder.up_blocks.1.resnets.1.conv1.", 512, 512, 128, 12, True),
         ("decoder.up_blocks.2.resnets.1.conv1.", 256, 256, 256, 12, True), ("decoder.up_blocks.3.resnets.1.conv1.", 128, 128, 512, 12, True)]
for prefix, cin, cout, hw, n, vae in cases:
    x = torch.randn((n, cin, hw, hw), device="cuda")
    for _ in range(2): k.block(3, prefix, x, cout, vae=vae)
    torch.cuda.synchronize()
    eng.profile_begin(); k.block(3, prefix, x, cout, vae=vae); torch.cuda.synchronize(); prof = eng.profile_end()
    fl = 2.0 * 9 * cin * cout * n * hw * hw
    ms = prof["gemm"]["ms"]
    print(f"{prefix:42s} {cin:4d}->{cout:4d} {hw:3d}x{hw:<3d} n={n:2d}: GEMM {ms:7.3f} ms  {fl / ms / 1e9:7.1f} useful TFLOP/s  ({prof['gemm']['work'] / ms / 1e9:7.1f} on the padded rows)", flush=True)
    del x
