"""One PGD inner-loop step (forward with kept activations + input gradient + projected sign step) of the ViT-L/14 image
tower at batch PB (default 256), for `rocprofv3 --kernel-trace --stats`; prints the in-process category times too."""
import importlib, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("multimodal-detection-consistency_amd")
pb = int(os.environ.get("PB", "256"))
arch = pkg.get_arch("ViT-L/14")
model = pkg.CLIPModel(pkg.CLIPConfig(model_name="ViT-L/14", device="cuda:0"), weights=pkg.synth.make_clip_weights(arch, seed=0))
eng = model.engine
torch.manual_seed(0)
clean = pkg.synth.make_images(pb, arch.image_size, seed=1).to("cuda:0")
adv, mom = clean.clone(), torch.zeros_like(clean)
g_out = (torch.randn((pb, arch.embed_dim), device="cuda") / pb).contiguous()
def it():
    eng.encode_image_grad(adv, True)
    g = eng.encode_image_backward(g_out)
    eng.pgd_step(adv, clean, g, mom, 8 / 255, 2 / 255, 0.9, 0.0, 1.0, False)
it(); torch.cuda.synchronize()
eng.profile_begin(); it(); torch.cuda.synchronize(); prof = eng.profile_end()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(3): it()
b.record(); torch.cuda.synchronize()
print(json.dumps({"batch": pb, "ms_per_step": a.elapsed_time(b) / 3, "ms": {k: round(v["ms"], 2) for k, v in prof.items()}}))
