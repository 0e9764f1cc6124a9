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


def test_auroc_three_method_defence_with_generated_sd_references_on_pgd_inputs(pkg):
    """BASELINE configs[4] at the toy geometry (oracle/defence_check.py): Q = 128 queries (64 clean, 64 perturbed by the
    in-tree PGDAttacker -- the SAME pixels go to both sides), N = 4 template variants, 2 generated references per query
    (3 PLMS steps + CFG, 16 x 16 latents -> 32 x 32 pixels -> CLIP preprocess -> image tower).  HIP:
    AdversarialDetector.batch_detect with the in-tree SDReferenceGenerator; oracle: clip_oracle towers, sd_oracle.generate
    from the same prompts / seeds / noise, tvc_oracle.detect_adversarial_src(sd_ref_feats=...).
    Bar: |dAUROC| <= 0.002 (BASELINE.json).  bench.py reports the same check as sd_reference.auroc_delta."""
    from oracle import defence_check
    r = defence_check.three_method_auroc(pkg, Q=128, N=4, J=2, steps=3)
    print(f"[measured] three-method defence, Q={r['Q']}: AUROC oracle {r['auroc_oracle']:.4f} gpu {r['auroc_gpu']:.4f} (sd_reference alone: "
          f"{r['auroc_sd_oracle']:.4f} / {r['auroc_sd_gpu']:.4f}); |d aggregated_score| max {r['max_abs_aggregated_dev']:.2e} median "
          f"{r['median_abs_aggregated_dev']:.2e}; |d sd_reference| max {r['max_abs_sd_reference_dev']:.2e}; score spread (std) {r['score_std']:.3f}")
    assert all(n == 2 for n in r["num_references"])
    assert abs(r["auroc_gpu"] - r["auroc_oracle"]) <= 0.002
    assert abs(r["auroc_sd_gpu"] - r["auroc_sd_oracle"]) <= 0.002
    # measured 5.5e-4 / 3.1e-4 with 3 references x 4 steps (bf16 towers + bf16 UNet / VAE)
    assert r["max_abs_aggregated_dev"] < 1.5e-3 and r["max_abs_sd_reference_dev"] < 1e-3
    assert r["max_flip_distance_to_threshold"] < 1.5e-3          # decisions differ only where the score sits on the threshold
