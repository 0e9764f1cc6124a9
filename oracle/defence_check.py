"""The three-method defence of ``src/detector.py:375-399`` -- text_variants + sd_reference + consistency -- on PGD-perturbed
inputs with GENERATED references, HIP against the CPU oracle, at a geometry the oracle finishes in well under a minute
(BASELINE configs[4] in miniature: toy CLIP towers, a two-level latent-diffusion model, 16 x 16 latents -> 32 x 32 pixels).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``: used by ``tests/test_gpu_auroc.py`` and by the ``cpu_baseline`` leg of
``bench.py`` (``sd_reference.auroc_delta``).  The HIP side goes through the product's public API (``PGDAttacker``,
``SDReferenceGenerator``, ``AdversarialDetector.batch_detect``); the oracle side is ``clip_oracle`` towers, ``sd_oracle.generate``
from the same prompts / seeds / noise (parity unpinned: see its header) and ``tvc_oracle.detect_adversarial_src(sd_ref_feats=...)``
aggregated as ``src/detector.py:643-682`` and scored as ``src/utils/metrics.py:286-329``.
"""
from __future__ import annotations

import random
from typing import Dict

import numpy as np
import torch

from . import clip_oracle, sd_oracle, tvc_oracle


def captions(n: int, seed: int = 0):
    """n distinct synthetic captions (COCO-shaped: 10..12 words)."""
    rnd = random.Random(seed)
    nouns = ["dog", "cat", "man", "woman", "child", "car", "bus", "train", "horse", "bird", "table", "pizza", "kite", "boat", "bench"]
    verbs = ["sitting on", "standing near", "running past", "looking at", "holding", "riding", "jumping over", "next to"]
    adjs = ["red", "small", "large", "old", "wooden", "bright", "two", "several", "young", "white"]
    places = ["in a park", "on the beach", "in a kitchen", "on a city street", "at night", "in the snow", "near a lake", "indoors"]
    out, seen = [], set()
    while len(out) < n:
        c = f"a {rnd.choice(adjs)} {rnd.choice(nouns)} {rnd.choice(verbs)} a {rnd.choice(adjs)} {rnd.choice(nouns)} {rnd.choice(places)}"
        if c not in seen:
            seen.add(c)
            out.append(c)
    return out


def three_method_auroc(pkg, Q: int = 128, N: int = 4, J: int = 2, steps: int = 3, guidance: float = 5.0, px: int = 32) -> Dict:
    """Q queries (half clean, half perturbed by the in-tree ``PGDAttacker`` -- the SAME pixels go to both sides), N template
    variants, J generated references per query (``steps`` PLMS steps + classifier-free guidance).  Returns AUROCs and score
    deviations of the HIP path against the oracle."""
    F = torch.nn.functional
    carch = pkg.get_arch("ViT-T/16-test")
    cw = pkg.synth.make_clip_weights(carch, seed=0)
    sarch = pkg.SDArch(block_out_channels=(64, 128), down_block_attn=(True, False), layers_per_block=1, heads=8,
                       cross_attention_dim=128, vae_block_out_channels=(64, 128), vae_layers_per_block=1, sample_size=16)
    uw, vw = pkg.make_sd_weights(sarch, seed=3)
    half = Q // 2
    texts = captions(Q)
    variants = pkg.variants.batch_variants(None, N, texts)
    clip = pkg.CLIPModel(pkg.CLIPConfig(model_name=carch.name), weights=cw)
    try:
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
        n_refs = [r["detection_details"]["sd_reference"].get("num_references", 0) for r in res]
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
            refs = torch.cat(refs)                                       # [Q * J, 3, px, px] in [0, 1]
            S = carch.image_size                                         # CLIP preprocess (clip.preprocess_tensor): bicubic, crop, mean / std
            r = F.interpolate(refs, size=(S, S), mode="bicubic", antialias=True, align_corners=False)
            mean = torch.tensor((0.48145466, 0.4578275, 0.40821073)).view(1, 3, 1, 1)
            std = torch.tensor((0.26862954, 0.26130258, 0.27577711)).view(1, 3, 1, 1)
            fr = clip_oracle.vision_forward(cw[0], (r - mean) / std, carch.vision.heads, carch.patch).view(Q, J, -1).numpy()
        ref_res = [tvc_oracle.detect_adversarial_src(fi[i], ft[i], methods=methods, sd_ref_feats=fr[i]) for i in range(Q)]
    finally:
        clip.engine.close()
    ref = np.array([r["aggregated_score"] for r in ref_res])
    ref_sd = np.array([r["detection_scores"]["sd_reference"] for r in ref_res])
    auc = lambda s: float(tvc_oracle.detection_metrics(s, labels)["auc"])
    flip = np.array([r["is_adversarial"] for r in res]) != np.array([r["is_adversarial"] for r in ref_res])
    return {"Q": Q, "variants": N, "references_per_query": J, "steps": steps, "num_references": n_refs,
            "auroc_oracle": auc(ref), "auroc_gpu": auc(got), "auroc_sd_oracle": auc(ref_sd), "auroc_sd_gpu": auc(got_sd),
            "max_abs_aggregated_dev": float(np.abs(got - ref).max()), "median_abs_aggregated_dev": float(np.median(np.abs(got - ref))),
            "max_abs_sd_reference_dev": float(np.abs(got_sd - ref_sd).max()), "score_std": float(ref.std()),
            "max_flip_distance_to_threshold": float(np.abs(ref[flip] - 0.5).max()) if flip.any() else 0.0}
