"""One batched SD generation (12 images, N steps at 64 x 64 latents + VAE decode) for rocprofv3 --kernel-trace --stats,
and the in-process category times (HIP events)."""
import importlib, json, sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
pkg = importlib.import_module("multimodal-detection-consistency_amd")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
streams = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # category times / kernel traces: the one-stream form (unoverlapped launches)
sd = pkg.StableDiffusionModel(pkg.SDModelConfig(random_init=True))
eng = sd.text_engine
eng.set_option(pkg._lib.TVC_OPT_SD_STREAMS, streams)
prompts = [f"a photo of object number {i}" for i in range(n)]
sd.generate_batch(prompts, list(range(n)), steps, 7.5, 512, 512)
torch.cuda.synchronize()
eng.profile_begin()
t0 = time.perf_counter()
imgs = sd.generate_batch(prompts, list(range(n)), steps, 7.5, 512, 512)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
prof = eng.profile_end()
import hashlib
md5 = hashlib.md5(torch.as_tensor(imgs).float().cpu().numpy().tobytes()).hexdigest()
print(json.dumps({"steps": steps, "images": n, "streams": streams, "seconds": round(dt, 4), "images_md5": md5,
                  "ms": {c: round(v["ms"], 2) for c, v in prof.items()},
                  "gemm_tflops": round(prof["gemm"]["work"] / (prof["gemm"]["ms"] * 1e-3) / 1e12, 1),
                  "attn_tflops": round(prof["attention"]["work"] / (prof["attention"]["ms"] * 1e-3) / 1e12, 1)}), flush=True)
