"""GPU: detection AUROC on identical synthetic PGD-perturbed inputs -- HIP path vs
the CPU oracle (fp32 towers + reference arithmetic).  BASELINE.json bar: within
+-0.002.  Inputs: half clean, half PGD (oracle/synth_pgd.py = the reference's
recipe), labels 1 = adversarial, AUROC = sklearn roc_auc_score on the src-polarity
aggregated score (src/utils/metrics.py:300)."""
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
    assert np.abs(got - ref).max() < 2e-2
    clip.engine.close()
