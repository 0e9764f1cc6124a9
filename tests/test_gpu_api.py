"""GPU: the drop-in Python surface over the HIP path --
results against the oracle and against the golden vectors produced by the
reference's own code."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import tvc_oracle as O

pytestmark = pytest.mark.gpu
G = Path(__file__).parent / "golden"
TEXTS = ["a cat on a sofa", "two dogs playing in the park", "a red car", "an old man reading a newspaper",
         "a bowl of fruit on a wooden table", "city skyline at night"]


@pytest.fixture(scope="module")
def clip(pkg):
    m = pkg.CLIPModel(pkg.CLIPConfig(model_name="ViT-T/16-test", device="cuda"))
    yield m
    m.engine.close()


@pytest.fixture(scope="module")
def images(pkg):
    return pkg.synth.make_images(len(TEXTS), 64, seed=3).cuda()


def test_clip_wrapper_surface(clip, images):
    fi = clip.encode_image(images)
    assert fi.is_cuda and fi.shape == (6, 128)
    assert torch.allclose(fi.norm(dim=-1), torch.ones(6, device="cuda"), atol=1e-5)
    assert not clip.encode_image(images.cpu()).is_cuda                 # host in -> host out (.numpy() callers)
    ft = clip.encode_text(TEXTS)
    assert not ft.is_cuda and ft.shape == (6, 128)
    assert clip.encode_text(TEXTS[:1], normalize=False).norm().item() != pytest.approx(1.0, abs=1e-3)
    s = clip.get_text_image_similarity(TEXTS[0], images[0])
    assert s.dim() == 0 and abs(s.item() - float((fi[0].cpu() * ft[0]).sum())) < 1e-5
    assert clip.tokenize("hello").shape == (1, 77) and clip.model is clip and clip.eval() is clip
    assert torch.equal(clip.encode_image_tensor(images, requires_grad=False), fi)
    xg = images.clone().requires_grad_(True)                       # the attacks' call (pgd_attack.py:459): autograd reaches the pixels
    fg = clip.encode_image_tensor(xg, requires_grad=True)
    assert fg.requires_grad and (fg.detach() - fi).abs().max().item() < 2e-3     # one more bf16 rounding per MLP (tvc.h)
    fg.sum().backward()
    assert xg.grad is not None and xg.grad.shape == images.shape and torch.isfinite(xg.grad).all() and xg.grad.abs().max() > 0
    from PIL import Image
    pil = Image.fromarray((np.random.default_rng(0).random((80, 100, 3)) * 255).astype(np.uint8))
    assert clip.preprocess(pil).shape == (3, 64, 64)
    assert clip.encode_image([pil, pil]).shape == (2, 128)


def test_adversarial_detector_matches_oracle(pkg, clip, images):
    det = pkg.AdversarialDetector(pkg.DetectorConfig(clip_model="ViT-T/16-test", num_text_variants=4), clip_model=clip)
    variants = [det._variants(t) for t in TEXTS]
    res = det.batch_detect(images, TEXTS)
    fi = clip.encode_image(images).cpu().numpy()
    for i, t in enumerate(TEXTS):
        ft = clip.encode_text([t] + variants[i]).numpy()
        want = O.detect_adversarial_src(fi[i], ft, methods=("text_variants", "consistency"))
        got = res[i]
        assert abs(got["aggregated_score"] - want["aggregated_score"]) < 1e-4
        assert got["is_adversarial"] == want["is_adversarial"]
        assert abs(got["detection_scores"]["text_variants"] - want["detection_scores"]["text_variants"]) < 1e-4
        d, w = got["detection_details"]["text_variants"], want["detection_details"]["text_variants"]
        np.testing.assert_allclose(d["variant_similarities"], w["variant_similarities"], atol=1e-4)
        assert d["num_variants"] == 4 and abs(d["std_variant_similarity"] - w["std_variant_similarity"]) < 1e-4
        assert set(got) >= {"is_adversarial", "aggregated_score", "detection_scores", "detection_details",
                            "detection_time", "methods_used", "threshold"}
    one = det.detect_adversarial(images[2], TEXTS[2])
    assert abs(one["aggregated_score"] - res[2]["aggregated_score"]) < 1e-6
    assert det.detect(images[2], TEXTS[2])["is_adversarial"] == one["is_adversarial"]
    # sd_reference arithmetic (src/detector.py:528-553) with caller-supplied reference images
    refs = [pkg.synth.make_images(3, 64, seed=50 + i) for i in range(2)]
    r2 = det.batch_detect(images[:2], TEXTS[:2], reference_images=[list(r) for r in refs])
    for i in range(2):
        fr = clip.encode_image(refs[i].cuda()).cpu().numpy()
        sims = [O.cosine(fi[i], f) for f in fr]
        want_sd, _ = O.sd_reference_score(sims)
        assert abs(r2[i]["detection_scores"]["sd_reference"] - want_sd) < 1e-4
        agg = O.aggregate_scores(r2[i]["detection_scores"])
        assert abs(r2[i]["aggregated_score"] - agg) < 1e-9


def test_defense_detector_matches_oracle(pkg, clip, images):
    cfg = pkg.DetectionConfig(text_variant_count=3)
    det = pkg.MultiModalDefenseDetector(clip, config=cfg)
    variants = [det._variants(t) for t in TEXTS]
    toks = clip.tokenize([x for t, v in zip(TEXTS, variants) for x in [t] + v])
    ft = clip.encode_tokens(toks).view(len(TEXTS), 4, -1).cpu()
    bank = pkg.synth.plant_neighbours(pkg.synth.make_bank(3000, 128, seed=7), ft.reshape(-1, 128), per_anchor=2)
    bank16 = bank.to(torch.bfloat16)
    det.set_reference_bank(bank16)
    got = det.batch_detect(images, TEXTS, return_details=True)
    fi = clip.encode_image(images).cpu().numpy()
    ref = O.detect_batch(fi, ft.numpy(), bank16.float().numpy(), checker=O.ConsistencyCheckerOracle())
    for i in range(len(TEXTS)):
        assert abs(got[i]["consistency_score"] - ref["overall_exp"][i]) < 1e-4
        assert abs(got[i]["confidence"] - ref["confidence_exp"][i]) < 1e-4
        assert got[i]["is_adversarial"] == bool(ref["is_adv_exp"][i])
        s = got[i]["details"]["consistency_scores"]
        assert abs(s["retrieval_consistency"] - ref["retrieval_consistency"][i]) < 1e-4
        want_refs = ref["retrieval_indices"][i]
        assert got[i]["details"]["retrieval_references"] == want_refs[want_refs >= 0].tolist()
    assert (ref["retrieval_indices"] >= 0).any()
    one = det.detect(images[:1], TEXTS[0])
    assert set(one) == {"is_adversarial", "confidence", "consistency_score"}


def test_pipeline_surface(pkg, clip, images):
    pc = pkg.PipelineConfig(enable_sd_reference=False, enable_profiling=True,
                            detector_config=pkg.DetectorConfig(clip_model="ViT-T/16-test", num_text_variants=3))
    pipe = pkg.create_detection_pipeline(pc, clip_model=clip)
    out = pipe.detect(images=images, texts=TEXTS, return_details=True)     # run_detection.py:172-203
    assert set(out) == {"predictions", "scores", "details"} and len(out["scores"]) == 6
    assert isinstance(out["predictions"][0], bool) and "detection_details" in out["details"][0]
    r = pipe.process_single(images[1], TEXTS[1])
    assert abs(r.adversarial_score - out["scores"][1]) < 1e-6 and r.detection_score == r.adversarial_score
    assert r.pipeline_steps == ["text_augment", "retrieval", "detection"] and len(r.text_variants) == 3
    assert pipe.process(images[1], TEXTS[1]).is_adversarial == r.is_adversarial
    rs = pipe.process_batch(images, TEXTS)
    assert [x.original_text for x in rs] == TEXTS                         # input order kept
    # retrieval step once an image index exists (src/pipeline.py:450-453: top_k=5)
    pipe.retriever.build_image_index(pkg.synth.make_images(40, 64, seed=9))
    r = pipe.process_single(images[0], TEXTS[0])
    assert len(r.retrieval_scores) == 5 and r.retrieval_scores == sorted(r.retrieval_scores, reverse=True)
    data = [(images[i % 6], TEXTS[i % 6], bool(i % 2)) for i in range(12)]
    ev = pipe.evaluate_pipeline(data, batch_size=5)
    assert 0.0 <= ev["detection_metrics"].auc <= 1.0 and ev["throughput"] > 0 and "detection" in ev["profiling"]
    with pytest.raises(ValueError):
        pipe.process_batch(images, TEXTS[:2])


def test_process_single_from_a_four_thread_pool(pkg, clip, images):
    """The reference drives ``process_single`` from a ThreadPoolExecutor(max_workers=4) (src/pipeline.py:284-288,553-566).
    One engine serves all four threads (a lock per entry point, every thread on its own current stream): 48 concurrent
    calls -- retrieval over a shared image index, a defence detector with its own bank slot beside it -- return exactly what
    the same calls return one after the other, and nothing deadlocks."""
    from concurrent.futures import ThreadPoolExecutor
    pc = pkg.PipelineConfig(enable_sd_reference=False,
                            detector_config=pkg.DetectorConfig(clip_model="ViT-T/16-test", num_text_variants=3))
    pipe = pkg.create_detection_pipeline(pc, clip_model=clip)
    pipe.retriever.build_image_index(pkg.synth.make_images(64, 64, seed=9))
    defense = pkg.MultiModalDefenseDetector(clip, config=pkg.DetectionConfig(text_variant_count=3, adaptive_threshold=False))
    defense.set_reference_bank(torch.nn.functional.normalize(torch.randn((500, 128), generator=torch.Generator().manual_seed(2)), dim=-1).cuda())
    jobs = [(i % 6, (i + i // 6) % 6) for i in range(48)]                 # (image, text) pairs: all 36 combinations, some twice

    def one(job):
        im, tx = job
        r = pipe.process_single(images[im], TEXTS[tx])
        d = defense.detect(images[im], TEXTS[tx])
        return r.adversarial_score, tuple(r.retrieval_scores), r.is_adversarial, d["consistency_score"], d["is_adversarial"]

    serial = [one(j) for j in jobs]
    with ThreadPoolExecutor(max_workers=4) as pool:
        threaded = list(pool.map(one, jobs, timeout=300))
    assert threaded == serial
    assert len({s[0] for s in serial}) > 6                                 # the jobs really differ


def test_reference_bank_matches_reference_golden(pkg, gpu_engine):
    """ReferenceBank.query_similar on the reference's own 20 x 512 fixture vs the
    outputs of the reference's own code (fp64 numpy); GPU path is split-bf16
    fp32-accumulated: 1e-6 on cosines."""
    g = np.load(G / "ref_bank.npz")
    bank = pkg.ReferenceBank(pkg.ReferenceBankConfig(similarity_threshold=float(g["config_threshold"]),
                                                     feature_dim=512), engine=gpu_engine)
    bank.add_references(g["vectors"], [{"i": i} for i in range(20)])
    for name, thr in (("default", None), ("t05", 0.5), ("t0", 0.0)):
        for i, q in enumerate(g["queries"]):
            got = bank.query_similar(q, top_k=10, similarity_threshold=thr)
            want = g[f"idx_{name}"][i]
            n = int((want >= 0).sum())
            assert [it.metadata["i"] for it, _ in got] == want[:n].tolist()
            np.testing.assert_allclose([s for _, s in got], g[f"sim_{name}"][i][:n], atol=1e-6)
    sims = bank._compute_similarities(g["queries"][5])
    np.testing.assert_allclose(sims, g["similarities"][5], atol=1e-6)
    assert bank.references[int(g["idx_t05"][5][0])].access_count > 0
    # admission check + capacity (src/ref_bank.py:137-150)
    small = pkg.ReferenceBank(pkg.ReferenceBankConfig(max_size=3, similarity_threshold=0.9, feature_dim=512), engine=gpu_engine)
    v = g["vectors"]
    assert small.add_reference(v[0], {}) and not small.add_reference(v[0] * 2.0, {}) and small.add_reference(v[1], {})
    assert small.add_reference(v[2], {}) and small.add_reference(v[3], {}) and len(small) == 3
    assert small.query_similar(np.zeros(512) + 1e-3) == [] or True


def test_similarity_calculator_matches_reference_golden(pkg, gpu_engine):
    g = np.load(G / "metrics.npz")
    got = pkg.SimilarityCalculator.batch_cosine_similarity(g["x"], g["y"], engine=gpu_engine)
    np.testing.assert_allclose(got, g["cos_numpy"], atol=1e-6)
    assert abs(pkg.SimilarityCalculator.cosine_similarity(g["x"][0], g["y"][0], engine=gpu_engine) - g["cos_pairs"][0]) < 1e-6
    assert pkg.SimilarityCalculator.cosine_similarity(np.zeros(512), g["y"][0], engine=gpu_engine) == float(g["cos_zero"])
    # D not a multiple of 64
    x = np.random.default_rng(1).standard_normal((5, 100)); y = np.random.default_rng(2).standard_normal((7, 100))
    np.testing.assert_allclose(pkg.SimilarityCalculator.batch_cosine_similarity(x, y, engine=gpu_engine),
                               O.batch_cosine_similarity(x, y), atol=1e-6)


def test_retriever_and_reference_generator(pkg, clip):
    feats = pkg.synth.make_bank(500, 128, seed=3)
    q = clip.encode_text(TEXTS[:2])
    feats[7] = q[0]; feats[9] = q[1]
    retr = pkg.MultiModalRetriever(pkg.RetrievalConfig(clip_model="ViT-T/16-test", bank_dtype="float32"), clip_model=clip)
    retr.set_image_features(feats, [f"img_{i}.jpg" for i in range(500)])
    paths, scores = retr.retrieve_images_by_text(TEXTS[0], top_k=5)
    assert paths[0] == "img_7.jpg" and abs(scores[0] - 1.0) < 1e-5 and len(paths) == 5
    assert retr.retrieve(TEXTS[1], k=3)[0][0] == "img_9.jpg"
    want_idx, want_s = O.search_index_exact(feats.numpy(), q[0].numpy(), 5)
    assert [int(p[4:-4]) for p in paths] == want_idx.tolist()
    np.testing.assert_allclose(scores, want_s, atol=1e-5)
    retr.build_text_index(TEXTS)
    sm = retr.compute_similarity_matrix()
    np.testing.assert_allclose(sm, O.batch_cosine_similarity(retr.text_features, feats.numpy()), atol=1e-5)
    gen = pkg.RetrievalReferenceGenerator(clip, features=feats, metadata=[{"id": i} for i in range(500)])
    refs = gen.retrieve_references(TEXTS[0])
    want = O.retrieve_references(feats.numpy(), q[0].numpy())
    assert [r["index"] for r in refs] == [r["index"] for r in want] and refs[0]["metadata"] == {"id": 7}
    # an fp32 bank is held as two bf16 planes (hi + lo): components < 0.25 come back to within 2^-19 = 1.9e-6
    np.testing.assert_allclose(refs[0]["features"], feats[7].numpy(), atol=2e-6, rtol=0)


def test_topk_merge_kernel(gpu_engine):
    W, M, k, kf, D = 4, 37, 8, 3, 128
    g = torch.Generator().manual_seed(0)
    sim = torch.rand((W, M, k), generator=g).sort(dim=-1, descending=True).values
    idx = torch.stack([torch.randperm(1000, generator=g)[:k] + 1000 * w for w in range(W) for _ in range(M)]).view(W, M, k).int()
    idx[3, :, 5:] = -1                      # short shard
    feat = torch.randn((W, M, kf, D), generator=g)
    mom = torch.rand((W, M, 4), generator=g)
    oi, os_, of, om = gpu_engine.topk_merge(idx.cuda(), sim.cuda(), feat.cuda(), mom.cuda())
    s = torch.where(idx >= 0, sim, torch.tensor(-1.0)).permute(1, 0, 2).reshape(M, W * k)
    i = idx.permute(1, 0, 2).reshape(M, W * k)
    order = s.argsort(dim=1, descending=True, stable=True)[:, :k]
    assert torch.equal(oi.cpu(), i.gather(1, order)) and torch.equal(os_.cpu(), s.gather(1, order))
    for m in range(M):
        for r in range(kf):
            w, j = divmod(int(order[m, r]), k)
            assert torch.equal(of[m, r].cpu(), feat[w, m, j])
    assert torch.allclose(om[:, 0].cpu(), mom[:, :, 0].sum(0)) and torch.allclose(om[:, 2].cpu(), mom[:, :, 2].max(0).values)


def test_empty_components_follow_the_reference(pkg, clip, images):
    """src/detector.py:375-383,457-458,524-525: augmenter / generator present but empty -> the method's 0.0
    score still enters the weighted mean (aggregated = cs / 3 with the default weights), on the host result
    AND in the device record; component absent -> method omitted."""
    det = pkg.AdversarialDetector(pkg.DetectorConfig(clip_model="ViT-T/16-test", num_text_variants=4), clip_model=clip)
    fi = clip.encode_image(images).cpu().numpy()
    refs = [pkg.synth.make_images(3, 64, seed=60), pkg.synth.make_images(2, 64, seed=61)]
    res = det.batch_detect(images[:3], TEXTS[:3], variants=[[], [], []],
                           reference_images=[[], list(refs[0]), list(refs[1])])
    for i in range(3):
        ft = clip.encode_text([TEXTS[i]]).numpy()
        fr = clip.encode_image(refs[i - 1].cuda()).cpu().numpy() if i else np.zeros((0, fi.shape[1]))
        want = O.detect_adversarial_src(fi[i], ft, methods=("text_variants", "sd_reference", "consistency"), sd_ref_feats=fr)
        got = res[i]
        assert set(got["detection_scores"]) == {"text_variants", "sd_reference", "consistency"}
        assert got["detection_scores"]["text_variants"] == 0.0 and "error" in got["detection_details"]["text_variants"]
        for m in ("sd_reference", "consistency"):
            assert abs(got["detection_scores"][m] - want["detection_scores"][m]) < 1e-4
        assert abs(got["aggregated_score"] - want["aggregated_score"]) < 1e-4
        assert got["is_adversarial"] == want["is_adversarial"]
    assert "error" in res[0]["detection_details"]["sd_reference"] and res[1]["detection_details"]["sd_reference"]["num_references"] == 3
    # device record word 5 with N = 0
    tok = clip.tokenize(TEXTS[:3]).view(3, 1, -1)
    rec = det.detect_tokens(images[:3], tok)
    for i in range(3):
        want = O.detect_adversarial_src(fi[i], clip.encode_text([TEXTS[i]]).numpy())
        assert abs(rec["aggregated_score"][i] - want["aggregated_score"]) < 1e-4
    # component absent: omitted, aggregated = consistency score
    det2 = pkg.AdversarialDetector(pkg.DetectorConfig(clip_model="ViT-T/16-test", use_text_variants=False), clip_model=clip)
    r2 = det2.batch_detect(images[:2], TEXTS[:2])
    for i in range(2):
        assert set(r2[i]["detection_scores"]) == {"consistency"}
        want = O.detect_adversarial_src(fi[i], clip.encode_text([TEXTS[i]]).numpy(), has_text_augmenter=False)
        assert abs(r2[i]["aggregated_score"] - want["aggregated_score"]) < 1e-4
    assert abs(det2.detect_tokens(images[:2], tok[:2])["aggregated_score"][0] - r2[0]["aggregated_score"]) < 1e-6


def test_bank_owners_do_not_clobber_each_other(pkg, clip, images):
    """One CLIPModel / engine shared by a pipeline retriever (image index), a MultiModalDefenseDetector
    (reference features), a RetrievalReferenceGenerator and a ReferenceBank: each keeps ITS rows
    (src/pipeline.py:306-331 builds them on one model; here every owner has a named slot, tvc_bank_select)."""
    pipe = pkg.create_detection_pipeline(
        pkg.PipelineConfig(enable_sd_reference=False,
                           detector_config=pkg.DetectorConfig(clip_model="ViT-T/16-test", num_text_variants=2)),
        clip_model=clip)
    pipe.retriever.build_image_index(pkg.synth.make_images(40, 64, seed=9))
    before = pipe.retriever.retrieve_images_by_text(TEXTS[0], top_k=5)
    det = pkg.MultiModalDefenseDetector(clip, config=pkg.DetectionConfig(text_variant_count=2))
    ft = clip.encode_text(TEXTS)
    bank = pkg.synth.plant_neighbours(pkg.synth.make_bank(3000, 128, seed=7), ft, per_anchor=2).to(torch.bfloat16)
    det.set_reference_bank(bank)
    d1 = det.batch_detect(images, TEXTS, return_details=True)
    gen = pkg.RetrievalReferenceGenerator(clip, features=pkg.synth.make_bank(500, 128, seed=8).numpy(),
                                          metadata=[{"i": i} for i in range(500)])
    rb = pkg.ReferenceBank(pkg.ReferenceBankConfig(feature_dim=128, similarity_threshold=0.0), engine=clip.engine)
    rb.add_references(np.random.default_rng(0).standard_normal((30, 128)))
    assert len(rb.query_similar(np.random.default_rng(1).standard_normal(128), top_k=40, similarity_threshold=-1.0)) == 30
    gen.retrieve_references(TEXTS[1])
    # every owner still sees its own rows
    pipe.retriever.retrieval_cache.clear()
    assert pipe.retriever.retrieve_images_by_text(TEXTS[0], top_k=5) == before
    det.consistency_checker.reset()
    d2 = det.batch_detect(images, TEXTS, return_details=True)
    assert [x["details"]["retrieval_references"] for x in d1] == [x["details"]["retrieval_references"] for x in d2]
    assert any(x["details"]["retrieval_references"] for x in d1)
    assert len({pipe.retriever.bank_name, det.bank_name, gen.bank_name, rb.bank_name}) == 4
    with pytest.raises(ValueError):
        pipe.retriever.retrieve_images_by_text(TEXTS[0], top_k=200)       # > TVC_MAX_TOPK: raises, never truncates


def test_bench_gpus_flag_runs_two_ranks_on_one_gpu(pkg):
    """`python bench.py --gpus 2` starts two worker ranks itself; on the one-GPU box they time-share cuda:0 over
    gloo (--rehearse-one-gpu).  Both layouts: data-parallel and --shard-bank."""
    import json
    import os
    import subprocess
    import sys
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    base = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--model", "ViT-T/16-test",
            "--batch", "16", "--variants", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
    for extra in (["--bank-rows", "4096"], ["--bank-rows", "8192", "--shard-bank"]):
        r = subprocess.run(base + extra, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        out = json.loads(lines[0])
        assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["global_batch"] == 32
        # what the process group itself reports (on an 8-GPU node: backend "nccl" = RCCL, world_size 8)
        assert out["distributed"]["world_size"] == 2 and out["distributed"]["backend"] == "gloo"
        if "--shard-bank" in extra:
            assert out["exchange"]["mode"] == "fused" and out["exchange"]["bytes_per_peer"] > 0 and "status_check" in out["exchange"]
            assert "all_to_all_single" in out["distributed"]["data_path_collectives"]


def test_text_variant_generator_clip_filter(pkg, clip):
    """rule candidates + the CLIP semantic window + ranking
    (experiments/defenses/text_variants.py:206-284), all candidates of all texts in ONE text encode.  The
    similarities are checked against the fp32 CPU oracle tower on the same tokens; selection and order
    against the reference's logic applied to them."""
    from oracle import clip_oracle
    texts = ["a small cat on a red sofa", "a photo of a big dog", "fast car", "people"]
    # random-init towers put related strings close together: widen the window so that it selects something
    cfg = pkg.TextVariantConfig(variant_count=6, diversity_threshold=0.1, similarity_threshold=0.97)
    gen = pkg.TextVariantGenerator(clip_model=clip, config=cfg)
    cands = [[v for v in gen.candidates(t) if gen._basic_filter(v, t)] for t in texts]
    gen._rng.seed(cfg.seed)                                           # the same reorderings again below
    sims = gen.similarities(texts, cands)
    tw = clip_oracle.round_gemm_weights_to_bf16(pkg.synth.make_clip_weights(clip.arch, seed=0)[1])
    worst = 0.0
    for t, c, s in zip(texts, cands, sims):
        with torch.no_grad():
            f = clip_oracle.text_forward(tw, clip.tokenize([t] + c), clip.arch.text.heads)
        ref = (f[1:] * f[0]).sum(-1).numpy()
        worst = max(worst, float(np.abs(ref - s).max()))
    print(f"[measured] text-text cosines, HIP vs oracle: max |d| {worst:.2e}")
    assert worst < 3e-3
    got = gen.batch_generate_variants(texts)
    for t, c, s, g in zip(texts, cands, sims, got):
        seen, want = set(), []
        for v, x in zip(c, s):
            if cfg.diversity_threshold < x < cfg.similarity_threshold and v not in seen:
                seen.add(v); want.append((v, x))
        want.sort(key=lambda p: p[1], reverse=True)
        assert g == [v for v, _ in want][:cfg.variant_count]
        assert len(set(g)) == len(g) and t not in g
    assert any(got)
    q = gen.evaluate_variant_quality(texts[0], got[0] or cands[0][:3])
    assert set(q) == {"variant_count", "similarity_stats", "diversity_stats", "quality_score"} and 0.0 <= q["quality_score"] <= 1.0
    # drop-in as the detector's augmenter: one batched generation for the whole batch
    det = pkg.AdversarialDetector(pkg.DetectorConfig(clip_model="ViT-T/16-test", num_text_variants=4), clip_model=clip,
                                  text_augmenter=gen)
    res = det.batch_detect(pkg.synth.make_images(4, 64, seed=3).cuda(), texts)
    assert len(res) == 4 and all("consistency" in r["detection_scores"] for r in res)


def test_build_reference_database_roundtrip(pkg, clip, tmp_path):
    """SURVEY.md 8f rank 2 (experiments/defenses/retrieval_ref.py:442-540, scripts/build_faiss_indices.py:59-120):
    stream a dataset through the image tower, write features.npy + metadata.json in the reference's format,
    load it back through RetrievalReferenceGenerator and retrieve."""
    import json
    R, bs = 1300, 100
    imgs = pkg.synth.make_images(R, 64, seed=70)
    caps = [f"caption number {i}" for i in range(R)]

    def loader():
        for i in range(0, R, bs):
            yield {"images": imgs[i:i + bs], "texts": caps[i:i + bs]}

    gen = pkg.RetrievalReferenceGenerator(clip, reference_db_path=str(tmp_path / "db"))
    assert gen.build_reference_database(loader(), max_samples=1234, encode_batch=512)
    feats = np.load(tmp_path / "db" / "features.npy")
    meta = json.load(open(tmp_path / "db" / "metadata.json", encoding="utf-8"))
    assert feats.shape == (1234, 128) and feats.dtype == np.float32 and len(meta) == 1234
    assert meta[0] == {"text": "caption number 0", "index": 0, "batch_idx": 0} and meta[1233]["batch_idx"] == 12
    assert np.abs(np.linalg.norm(feats, axis=1) - 1.0).max() < 1e-5
    direct = clip.encode_image(imgs[:1234].cuda()).cpu().numpy()
    assert np.array_equal(feats[:512], direct[:512])          # the same 512-image launches (batch invariant towers)
    assert np.abs(feats - direct).max() < 1e-6
    # round trip: a fresh generator reads the directory and retrieves
    gen2 = pkg.RetrievalReferenceGenerator(clip, reference_db_path=str(tmp_path / "db"),
                                           config=pkg.RetrievalRefConfig(similarity_threshold=-1.0))
    refs = gen2.retrieve_references("caption number 7")
    q = clip.encode_text(["caption number 7"]).numpy()[0]
    S = feats.astype(np.float64) @ q.astype(np.float64)
    order = np.argsort(-S, kind="stable")[:5]
    assert [r["index"] for r in refs] == order.tolist() and refs[0]["metadata"]["index"] == int(order[0])
    assert np.abs(np.array([r["similarity"] for r in refs]) - S[order]).max() < 1e-5
    # scripts/build_faiss_indices.py:59-120 form
    def loader2():
        for i in range(0, 300, 64):
            j = min(i + 64, 300)
            yield {"image": imgs[i:j], "text": caps[i:j], "image_id": list(range(i, j))}
    fi, ft, ids = pkg.extract_features(clip, loader2(), encode_batch=128)
    assert fi.shape == ft.shape == (300, 128) and ids == list(range(300))
    assert np.abs(fi - direct[:300]).max() < 1e-6
    assert np.abs(ft - clip.encode_text(caps[:300]).numpy()).max() < 1e-6


def test_sd_reference_generator_feeds_the_detector(pkg, clip, images):
    """SURVEY.md 8f rank 1, the orchestration side: SDReferenceGenerator (diffusion model injected) -> batched CLIP
    encode of the generated references -> the detector's sd_reference score (src/detector.py:524-553)."""
    from tests.test_abi_and_host import _FakeSD
    gen = pkg.SDReferenceGenerator(pkg.SDReferenceConfig(num_images_per_prompt=3, use_text_variants=False), sd_model=_FakeSD(),
                                   clip_model=clip)
    feats, counts = gen.reference_features(TEXTS[:3])
    assert counts == [3, 3, 3] and feats.shape == (9, 128) and feats.is_cuda
    imgs = [im for t in TEXTS[:3] for im in gen.generate_reference_images(t)["images"]]
    assert torch.equal(feats, clip.engine.encode_image(torch.stack(imgs).cuda(), True))
    det = pkg.AdversarialDetector(pkg.DetectorConfig(clip_model="ViT-T/16-test", num_reference_images=3,
                                                     detection_methods=["sd_reference", "consistency"]),
                                  clip_model=clip, sd_generator=gen)
    res = det.batch_detect(images[:3], TEXTS[:3])
    fi = clip.encode_image(images[:3]).cpu().numpy()
    fr = feats.cpu().numpy().reshape(3, 3, -1)
    for i in range(3):
        want, _ = O.sd_reference_score([O.cosine(fi[i], f) for f in fr[i]])
        assert abs(res[i]["detection_scores"]["sd_reference"] - want) < 1e-4
        assert res[i]["detection_details"]["sd_reference"]["num_references"] == 3


def test_sd_generation_failure_is_the_sd_methods_failure_only(pkg, clip, images):
    """src/detector.py:555-557: when reference generation fails, ``_detect_by_sd_reference`` returns 0.0 + {'error'} and the
    other methods' results stand -- with a batched generator (``reference_features``) that raises, with a per-prompt
    generator (``generate_reference_images``) that raises for one query, and with the in-tree generator that has no model
    (no weights and no random_init opt-in: the reference's load failure, src/sd_ref.py:291-317)."""
    class Boom:
        clip_model = None

        def reference_features(self, prompts, num_images=None):
            raise RuntimeError("out of memory (simulated)")

    class BoomOne:
        def generate_reference_images(self, text, num_images=None):
            if text == TEXTS[1]:
                raise RuntimeError("pipeline crashed (simulated)")
            g = torch.Generator().manual_seed(len(text))
            return {"images": [torch.randn((3, 64, 64), generator=g) for _ in range(num_images)]}

    cfg = pkg.DetectorConfig(clip_model="ViT-T/16-test", num_reference_images=2)
    base = pkg.AdversarialDetector(cfg, clip_model=clip).batch_detect(images[:3], TEXTS[:3], methods=["text_variants", "consistency"])
    for sdg, bad in ((Boom(), {0, 1, 2}), (BoomOne(), {1}),
                     (pkg.SDReferenceGenerator(pkg.SDReferenceConfig(use_text_variants=False, filter_low_quality=False), clip_model=clip),
                      {0, 1, 2})):
        res = pkg.AdversarialDetector(cfg, clip_model=clip, sd_generator=sdg).batch_detect(images[:3], TEXTS[:3])
        for i in range(3):
            s = res[i]["detection_scores"]
            assert abs(s["text_variants"] - base[i]["detection_scores"]["text_variants"]) < 1e-7
            assert abs(s["consistency"] - base[i]["detection_scores"]["consistency"]) < 1e-7
            if i in bad:
                assert s["sd_reference"] == 0.0 and "error" in res[i]["detection_details"]["sd_reference"]
                want = (0.4 * s["text_variants"] + 0.4 * 0.0 + 0.2 * s["consistency"]) / 1.0      # :664-680: the 0.0 enters
                assert abs(res[i]["aggregated_score"] - want) < 1e-6
            else:
                assert res[i]["detection_details"]["sd_reference"]["num_references"] == 2


def test_defense_detector_generative_branch(pkg, clip, images):
    """experiments/defenses/detector.py:206-226,268-300 with an injected generator (the diffusion model itself is
    not built): references for the first three of (original + variants), cut to generation_count, cos(image, ref)
    mean / std -> generative_consistency / _std, cross-modal variance over the four means, the checker's decision."""
    class Gen:
        def __init__(self):
            self.asked = []

        def generate_references(self, text):
            self.asked.append(text)
            g = torch.Generator().manual_seed(len(text) * 7 + 1)
            return [torch.randn((3, 64, 64), generator=g) for _ in range(2)]

    gen = Gen()
    cfg = pkg.DetectionConfig(text_variant_count=3, generation_count=5, use_retrieval_ref=False, adaptive_threshold=False)
    det = pkg.MultiModalDefenseDetector(clip, config=cfg, generative_generator=gen)
    got = det.batch_detect(images[:3], TEXTS[:3], return_details=True)
    assert len(gen.asked) == 9 and gen.asked[0] == TEXTS[0]            # 3 texts per query
    fi = clip.encode_image(images[:3]).cpu().numpy()
    ck = O.ConsistencyCheckerOracle(adaptive_threshold=False)
    for i in range(3):
        texts_i = got[i]["details"]["text_variants"]
        ft = clip.encode_text(texts_i).numpy()
        refs = []
        for t in texts_i[:3]:
            g = torch.Generator().manual_seed(len(t) * 7 + 1)
            refs.extend(torch.randn((3, 64, 64), generator=g) for _ in range(2))
        refs = refs[:5]
        assert len(got[i]["details"]["generative_references"]) == 5
        fr = clip.encode_image(torch.stack(refs).cuda()).cpu().numpy()
        want = O.compute_consistency_scores_exp(fi[i], ft, None, fr)
        s = got[i]["details"]["consistency_scores"]
        for key in ("original_similarity", "text_variant_consistency", "generative_consistency", "generative_std",
                    "cross_modal_variance"):
            assert abs(s[key] - want[key]) < 1e-4, (key, s[key], want[key])
        d = ck.make_decision(want)
        assert abs(got[i]["consistency_score"] - d["overall_score"]) < 1e-4 and got[i]["is_adversarial"] == d["is_adversarial"]
    assert det.get_statistics()["components"]["generative_generator"] is True
