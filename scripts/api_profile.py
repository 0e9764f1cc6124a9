"""Host-side profile (cProfile) of the string-in / objects-out API calls the bench's `through_api` extra times:
MultiModalDefenseDetector.batch_detect and MultiModalDetectionPipeline.detect at B = 512, N = 8, 1 M-row bank."""
import cProfile, importlib, io, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module("multimodal-detection-consistency_amd")
B, N, R = 512, 8, 1_000_000
arch = pkg.get_arch("ViT-L/14")
clip = pkg.CLIPModel(pkg.CLIPConfig(model_name="ViT-L/14", device="cuda:0"), weights=pkg.synth.make_clip_weights(arch, seed=0))
images = pkg.synth.make_images(B, arch.image_size, seed=1).to("cuda:0")
bank = pkg.synth.make_bank(R, arch.embed_dim, seed=7, device="cuda:0", dtype=torch.bfloat16)
texts, vocab = bench.make_captions(B)
gen = bench.WordReplaceVariants(N, vocab)
defense = pkg.MultiModalDefenseDetector(clip, config=pkg.DetectionConfig(text_variant_count=N), text_generator=gen)
defense.set_reference_bank(bank)
pipe = pkg.MultiModalDetectionPipeline(pkg.PipelineConfig(enable_sd_reference=False, detector_config=pkg.DetectorConfig(clip_model="ViT-L/14", num_text_variants=N)),
                                       clip_model=clip, text_augmenter=gen)
pipe.retriever.set_image_features(bank)
for name, fn in (("defense.batch_detect", lambda: defense.batch_detect(images, texts)),
                 ("pipeline.detect", lambda: pipe.detect(images=images, texts=texts, return_details=True))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    pr = cProfile.Profile(); pr.enable(); fn(); torch.cuda.synchronize(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
    print(f"==== {name}: {dt * 1e3:.1f} ms per call ({B / dt:.0f} q/s)"); print("\n".join(s.getvalue().splitlines()[4:26]))
