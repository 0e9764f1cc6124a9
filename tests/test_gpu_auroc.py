"""GPU: detection AUROC on identical synthetic PGD-perturbed inputs -- HIP path vs
the CPU oracle (fp32 towers + reference arithmetic).  BASELINE.json bar: within
+-0.002.  Inputs: half clean, half PGD (oracle/synth_pgd.py = the reference's
recipe), labels 1 = adversarial, AUROC = sklearn roc_auc_score on the src-polarity
aggregated score (src/utils/metrics.py:300).

``test_auroc_three_method_defence_...`` is BASELINE configs[4] ("Full TVC defense: SD ref generation + PGD-perturbed
inputs, end-to-end queries/sec + AUROC") at a geometry the CPU oracle finishes in about a minute: all three methods of
``src/detector.py:375-399`` -- text_variants + sd_reference + consistency -- with the references GENERATED on both sides
(HIP: ``tvc_sd_generate``; oracle: ``sd_oracle.generate``, parity unpinned -- see its header) from the same prompts and
seeds, aggregated as ``src/detector.py:643-682`` and scored as ``src/utils/metrics.py:286-329``."""
import numpy as np
import pytest
import torch

from oracle import clip_oracle, synth_pgd, tvc_oracle

pytestmark = pytest.mark.gpu


def test_auroc_matches_cpu_oracle(pkg):
    arch = pkg.get_arch("ViT-T/16-test")
    vw, tw = pkg.synth.make_clip_weights(arch, seed=0)
    Q, N = 128, 4
    clean = pkg.synth.make_images(Q, arch.image_size, seed=1)
    tokens = pkg.synth.make_tokens(Q, N, arch.ctx, seed=2)
    half = Q // 2
    adv = synth_pgd.pgd_images(vw, tw, clean[half:], tokens[half:, 0].long(), arch.vision.heads, arch.text.heads,
                               arch.patch)
    images = torch.cat([clean[:half], adv])
    labels = np.r_[np.zeros(half), np.ones(half)]
    # CPU oracle scores (fp32 towers on the fp32 weights, reference arithmetic)
    with torch.no_grad():
        fi = clip_oracle.vision_forward(vw, images, arch.vision.heads, arch.patch).numpy()
        ft = clip_oracle.text_forward(tw, tokens.view(-1, arch.ctx).long(), arch.text.heads).view(Q, N + 1, -1).numpy()
    ref = np.array([tvc_oracle.detect_adversarial_src(fi[i], ft[i])["aggregated_score"] for i in range(Q)])
    # HIP path
    clip = pkg.CLIPModel(pkg.CLIPConfig(model_name=arch.name), weights=(vw, tw))
    det = pkg.AdversarialDetector(pkg.DetectorConfig(clip_model=arch.name), clip_model=clip)
    got = det.detect_tokens(images.cuda(), tokens.cuda())["aggregated_score"]
    auc_ref = tvc_oracle.detection_metrics(ref, labels)["auc"]
    auc_gpu = tvc_oracle.detection_metrics(got, labels)["auc"]
    print(f"AUROC oracle {auc_ref:.4f} gpu {auc_gpu:.4f}  max|dscore| {np.abs(got - ref).max():.2e}")
    assert abs(auc_gpu - auc_ref) <= 0.002
    # scores themselves: bf16 towers vs fp32 towers
    assert np.abs(got - ref).max() < 1.5e-3           # measured 5.8e-4
    clip.engine.close()


def test_auroc_q1000_vit_b32_committed_pgd_fixture(pkg):
    """Q = 1000 (500 clean + 500 PGD) at ViT-B/32, N = 4, 1k-row planted bank, on the
    COMMITTED fixture (tests/golden/pgd_b32_q1000.npz, generated once in the build container by
    oracle/make_pgd_fixture.py through oracle/synth_pgd.py = /root/reference/src/attacks/pgd_attack.py:406-523).
    The fixture also carries the CPU oracle's scores (fp32 towers + reference arithmetic), so both sides saw
    identical inputs.  Bar: |dAUROC| <= 0.002 in BOTH polarities (src: high = adversarial,
    src/detector.py:399; exp: low = adversarial, consistency_checker.py:93; AUROC =
    sklearn.roc_auc_score, src/utils/metrics.py:300)."""
    from oracle import make_pgd_fixture as F
    fx = F.load_fixture(pkg)
    arch, (vw, tw) = fx["arch"], fx["weights"]
    Q, N = F.Q, F.N
    eng = pkg.TVCEngine(arch, vw, tw)
    eng.set_bank(fx["bank"].cuda())
    cfg = pkg.ConsistencyConfig()
    recs, fis, fts = [], [], []
    for i in range(0, Q, 250):
        fi = eng.encode_image(fx["images"][i:i + 250].cuda())
        ft = eng.encode_text(fx["tokens"][i:i + 250].reshape(-1, arch.ctx).cuda(), group=N + 1).view(-1, N + 1, arch.embed_dim)
        recs.append(eng.detect_embeddings(fi, ft, cfg, robust=True).cpu().numpy())
        fis.append(fi.cpu()); fts.append(ft.cpu())
    rec = np.concatenate(recs)
    o, labels = fx["oracle"], fx["labels"]
    auc_src = tvc_oracle.detection_metrics(rec[:, 5], labels)["auc"]
    auc_exp = tvc_oracle.detection_metrics(-rec[:, 10].astype(np.float64), labels)["auc"]
    d_src = np.abs(rec[:, 5] - o["score_src"])
    d_exp = np.abs(rec[:, 10] - o["overall_exp"])
    d_s0 = np.abs(rec[:, 0] - o["original_similarity"])
    # embeddings on the 128-query sample the fixture keeps
    S = o["feat_sample"]
    fi, ft = torch.cat(fis)[S], torch.cat(fts)[S]
    cos_i = (fi * torch.from_numpy(o["image_feats"])).sum(-1)
    cos_t = (ft * torch.from_numpy(o["text_feats"])).sum(-1)
    print(f"[measured] ViT-B/32 Q=1000: AUROC src oracle {float(o['auc_src']):.4f} gpu {auc_src:.4f}; "
          f"exp oracle {float(o['auc_exp']):.4f} gpu {auc_exp:.4f}")
    print(f"[measured] ViT-B/32 Q=1000 |score_gpu - score_cpu_fp32|: score_src max {d_src.max():.2e} p99 {np.percentile(d_src, 99):.2e} "
          f"median {np.median(d_src):.2e}; overall_exp max {d_exp.max():.2e} p99 {np.percentile(d_exp, 99):.2e}; "
          f"original_similarity max {d_s0.max():.2e}; image min cos {cos_i.min().item():.6f} max|d| "
          f"{(fi - torch.from_numpy(o['image_feats'])).abs().max().item():.2e}; text min cos {cos_t.min().item():.6f} "
          f"max|d| {(ft - torch.from_numpy(o['text_feats'])).abs().max().item():.2e}")
    assert abs(auc_src - float(o["auc_src"])) <= 0.002
    assert abs(auc_exp - float(o["auc_exp"])) <= 0.002
    # the published end-to-end figure: bf16 towers vs fp32 CPU towers, ViT-B/32
    # measured on MI355X: score_src max 7.5e-4 (p99 6.0e-4), original_similarity max 1.4e-3; the exp-polarity
    # overall score is DISCONTINUOUS where a component cosine crosses 0 (consistency_checker.py:152 keeps scores > 0
    # only), so its worst case (1.3e-2, p99 3.6e-3) is set by sign flips of near-zero cosines, not by rounding
    assert d_src.max() < 1.5e-3 and d_s0.max() < 3e-3 and np.percentile(d_exp, 99) < 8e-3
    assert cos_i.min().item() > 0.99998 and cos_t.min().item() > 0.9999         # measured 0.999993 / 0.999964
    # decisions: identical except where the score sits within the measured deviation of the threshold
    flip = (rec[:, 5] > 0.5) != o["is_adv_src"]
    assert (np.abs(o["score_src"][flip] - 0.5) < 1.5e-3).all()
    eng.close()


def _captions(n: int, seed: int = 0):
    """n distinct synthetic captions (COCO-shaped: 6..12 words)."""
    import random
    rnd = random.Random(seed)
    nouns = ["dog", "cat", "man", "woman", "child", "car", "bus", "train", "horse", "bird", "table", "pizza", "kite", "boat", "bench"]
    verbs = ["sitting on", "standing near", "running past", "looking at", "holding", "riding", "jumping over", "next to"]
    adjs = ["red", "small", "large", "old", "wooden", "bright", "two", "several", "young", "white"]
    places = ["in a park", "on the beach", "in a kitchen", "on a city street", "at night", "in the snow", "near a lake", "indoors"]
    out, seen = [], set()
    while len(out) < n:
        c = f"a {rnd.choice(adjs)} {rnd.choice(nouns)} {rnd.choice(verbs)} a {rnd.choice(adjs)} {rnd.choice(nouns)} {rnd.choice(places)}"
        if c not in seen:
            seen.add(c); out.append(c)
    return out


def test_auroc_three_method_defence_with_generated_sd_references_on_pgd_inputs(pkg):
    """BASELINE configs[4] at the toy geometry: Q = 128 queries (64 clean, 64 perturbed by the in-tree PGDAttacker --
    the SAME pixels go to both sides), N = 4 template variants, 2 generated references per query (3 PLMS steps + CFG,
    16 x 16 latents -> 32 x 32 pixels -> CLIP preprocess -> image tower).  HIP: AdversarialDetector.batch_detect with the
    in-tree SDReferenceGenerator; oracle: clip_oracle towers, sd_oracle.generate from the same prompts / seeds / noise,
    tvc_oracle.detect_adversarial_src(sd_ref_feats=...).  Bar: |dAUROC| <= 0.002 (BASELINE.json)."""
    from oracle import sd_oracle
    F = torch.nn.functional
    carch = pkg.get_arch("ViT-T/16-test")
    cw = pkg.synth.make_clip_weights(carch, seed=0)
    sarch = pkg.SDArch(block_out_channels=(64, 128), down_block_attn=(True, False), layers_per_block=1, heads=8,
                       cross_attention_dim=128, vae_block_out_channels=(64, 128), vae_layers_per_block=1, sample_size=16)
    uw, vw = pkg.make_sd_weights(sarch, seed=3)
    Q, N, J, steps, guidance, px = 128, 4, 2, 3, 5.0, 32
    half = Q // 2
    texts = _captions(Q)
    variants = pkg.variants.batch_variants(None, N, texts)
    clip = pkg.CLIPModel(pkg.CLIPConfig(model_name=carch.name), weights=cw)
    sdm = pkg.StableDiffusionModel(pkg.SDModelConfig(), clip_model=clip, arch=sarch, weights=(uw, vw))
    gen = pkg.SDReferenceGenerator(pkg.SDReferenceConfig(num_images_per_prompt=J, num_inference_steps=steps, guidance_scale=guidance,
                                                         height=px, width=px, use_text_variants=False, filter_low_quality=False,
                                                         enable_cache=False), sd_model=sdm, clip_model=clip)
    clean = pkg.synth.make_images(Q, carch.image_size, seed=1)
    atk = pkg.PGDAttacker(clip, pkg.PGDAttackConfig(batch_size=half, random_seed=7))
    adv = atk.perturb(clean[half:].cuda(), texts[half:]).cpu()
    images = torch.cat([clean[:half], adv])
    labels = np.r_[np.zeros(half), np.ones(half)]
    methods = ["text_variants", "sd_reference", "consistency"]
    det = pkg.AdversarialDetector(pkg.DetectorConfig(clip_model=carch.name, num_text_variants=N, num_reference_images=J),
                                  clip_model=clip, sd_generator=gen)
    res = det.batch_detect(images.cuda(), texts, methods=methods, variants=variants)
    got = np.array([r["aggregated_score"] for r in res])
    got_sd = np.array([r["detection_scores"]["sd_reference"] for r in res])
    assert all(r["detection_details"]["sd_reference"]["num_references"] == J for r in res)
    # ---- CPU oracle: towers, generated references, reference arithmetic
    flat = [t for i in range(Q) for t in [texts[i]] + list(variants[i])]
    with torch.no_grad():
        fi = clip_oracle.vision_forward(cw[0], images, carch.vision.heads, carch.patch).numpy()
        ft = clip_oracle.text_forward(cw[1], clip.tokenize(flat), carch.text.heads).view(Q, N + 1, -1).numpy()
        cond = clip_oracle.text_hidden(cw[1], sdm.tokenize(texts).long(), carch.text.heads)
        unc = clip_oracle.text_hidden(cw[1], sdm.tokenize([""]).long(), carch.text.heads)
        seeds = gen._generate_seeds(J)
        up = 2 ** (len(sarch.vae_block_out_channels) - 1)
        refs = []
        for i0 in range(0, Q, 32):                                   # 32 prompts x J seeds per oracle pass
            c = cond[i0:i0 + 32].repeat_interleave(J, 0)
            lat0 = sdm.initial_latents(seeds * (c.shape[0] // J), sarch.in_channels, px // up, px // up)
            refs.append(sd_oracle.generate(uw, vw, sarch, c, unc.expand(c.shape[0], -1, -1), lat0, steps, guidance))
        refs = torch.cat(refs)                                       # [Q * J, 3, 32, 32] in [0, 1]
        S = carch.image_size                                         # CLIP preprocess (clip.preprocess_tensor): bicubic, crop, mean / std
        r = F.interpolate(refs, size=(S, S), mode="bicubic", antialias=True, align_corners=False)
        mean = torch.tensor((0.48145466, 0.4578275, 0.40821073)).view(1, 3, 1, 1)
        std = torch.tensor((0.26862954, 0.26130258, 0.27577711)).view(1, 3, 1, 1)
        fr = clip_oracle.vision_forward(cw[0], (r - mean) / std, carch.vision.heads, carch.patch).view(Q, J, -1).numpy()
    ref_res = [tvc_oracle.detect_adversarial_src(fi[i], ft[i], methods=methods, sd_ref_feats=fr[i]) for i in range(Q)]
    ref = np.array([r["aggregated_score"] for r in ref_res])
    ref_sd = np.array([r["detection_scores"]["sd_reference"] for r in ref_res])
    auc_ref = tvc_oracle.detection_metrics(ref, labels)["auc"]
    auc_gpu = tvc_oracle.detection_metrics(got, labels)["auc"]
    auc_sd_ref = tvc_oracle.detection_metrics(ref_sd, labels)["auc"]
    auc_sd_gpu = tvc_oracle.detection_metrics(got_sd, labels)["auc"]
    d_agg, d_sd = np.abs(got - ref), np.abs(got_sd - ref_sd)
    print(f"[measured] three-method defence, Q={Q}: AUROC oracle {auc_ref:.4f} gpu {auc_gpu:.4f} (sd_reference alone: "
          f"{auc_sd_ref:.4f} / {auc_sd_gpu:.4f}); |d aggregated_score| max {d_agg.max():.2e} median {np.median(d_agg):.2e}; "
          f"|d sd_reference| max {d_sd.max():.2e} median {np.median(d_sd):.2e}; score spread (std) {ref.std():.3f}")
    assert abs(auc_gpu - auc_ref) <= 0.002
    assert abs(auc_sd_gpu - auc_sd_ref) <= 0.002
    assert d_agg.max() < 1.5e-3 and d_sd.max() < 1e-3               # measured 5.5e-4 / 3.1e-4 with 3 references x 4 steps (bf16 towers + bf16 UNet / VAE)
    flip = np.array([r["is_adversarial"] for r in res]) != np.array([r["is_adversarial"] for r in ref_res])
    assert (np.abs(ref[flip] - 0.5) < 1.5e-3).all()
    clip.engine.close()
