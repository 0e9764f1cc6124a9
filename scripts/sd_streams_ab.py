"""A/B of TVC_OPT_SD_STREAMS (the two classifier-free-guidance halves of a UNet evaluation on one / two HIP streams):
images/s of one batched generation (20 steps, 64 x 64 latents + VAE decode), unprofiled, and the md5 of the images
(equal = bit-identical).  Usage: python scripts/sd_streams_ab.py [steps] [n ...]"""
import hashlib, importlib, json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("multimodal-detection-consistency_amd")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ns = [int(x) for x in sys.argv[2:]] or [12, 40]
sd = pkg.StableDiffusionModel(pkg.SDModelConfig(random_init=True))
eng = sd.text_engine
for n in ns:
    prompts = [f"a photo of object number {i}" for i in range(n)]
    for streams in (1, 2, 1, 2):
        eng.set_option(pkg._lib.TVC_OPT_SD_STREAMS, streams)
        sd.generate_batch(prompts, list(range(n)), 2, 7.5, 512, 512)          # warm-up (workspaces)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        imgs = sd.generate_batch(prompts, list(range(n)), steps, 7.5, 512, 512)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        md5 = hashlib.md5(torch.as_tensor(imgs).float().cpu().numpy().tobytes()).hexdigest()
        print(json.dumps({"images": n, "steps": steps, "streams": streams, "seconds": round(dt, 4),
                          "images_per_s": round(n / dt, 2), "images_md5": md5}), flush=True)
