"""GPU: BASELINE.json configs[2] -- the configuration every headline number is measured on --
checked for CORRECTNESS: ViT-L/14 (T = 257 exact-length attention, width 1024, 24 layers, patch 14 ->
K = 588 padded to 640), batch 512, N = 8 variants, 1 M-row bf16 bank.

Replaces the reference call sites /root/reference/src/detector.py:461-471,626 and
/root/reference/experiments/defenses/detector.py:238-252.

(a) towers: a handful of images / two queries' 18 texts, HIP vs the fp32 CPU oracle, against the
    bf16-rounded weights (isolates the kernels' bf16 activations) and against the fp32 weights (the
    end-to-end deviation of the bf16 path);
(b) the full B = 512, N = 8, R = 1 M step exactly as bench.py runs it; 16 sampled queries' records vs
    ``tvc_oracle.detect_batch`` on the SAME embeddings (the BASELINE 1e-4 bar) and on the oracle's own
    fp32 embeddings (end to end).  Batch-split invariance (tested at configs[1]) makes the sample valid.

The MEASURED deviations are printed (pytest -s) and recorded in DESIGN.md section 2; the asserted bounds
are ~2x the measured values.
"""
import numpy as np
import pytest
import torch

from oracle import clip_oracle, tvc_oracle

pytestmark = pytest.mark.gpu

# asserted bounds = ~2x the deviations measured on MI355X (DESIGN.md section 2, table "measured deviation")
TOWER_BOUNDS = {          # (min cos, max |d embedding|) vs bf16-rounded weights / vs fp32 weights
    "image": {"bf16w": (0.99999, 7e-4), "fp32w": (0.99998, 1e-3)},      # measured 0.999997 / 3.2e-4, 0.999994 / 4.6e-4
    "text": {"bf16w": (0.99996, 1.5e-3), "fp32w": (0.99994, 2.2e-3)},     # measured 0.999985 / 6.9e-4, 0.999973 / 1.0e-3
}
SCORE_BOUND_SAME_EMB = 1e-4      # BASELINE.json bar, identical embeddings (measured 1.7e-8)
SCORE_BOUND_END_TO_END = 2e-3    # bf16 towers vs fp32 CPU towers: score_src measured 2.5e-4, original_similarity 6.9e-4


@pytest.fixture(scope="module")
def l14(pkg):
    arch = pkg.get_arch("ViT-L/14")
    w = pkg.synth.make_clip_weights(arch, seed=0)
    eng = pkg.TVCEngine(arch, w[0], w[1])
    yield arch, w, eng
    eng.close()


def _dev(got: torch.Tensor, ref: torch.Tensor):
    cos = (got * ref).sum(-1)
    return cos.min().item(), (got - ref).abs().max().item()


def test_config2_vit_l14_towers_vs_oracle(pkg, l14):
    arch, (vw, tw), eng = l14
    imgs = pkg.synth.make_images(4, arch.image_size, seed=1)
    toks = pkg.synth.make_tokens(2, 8, arch.ctx, seed=2).reshape(-1, arch.ctx)      # two queries' 9 texts each
    gi = eng.encode_image(imgs.cuda()).cpu()
    gt = eng.encode_text(toks.cuda(), group=9).cpu()
    assert torch.isfinite(gi).all() and torch.isfinite(gt).all()
    vr, tr = clip_oracle.round_gemm_weights_to_bf16(vw), clip_oracle.round_gemm_weights_to_bf16(tw)
    with torch.no_grad():
        refs = {"image": {"bf16w": clip_oracle.vision_forward(vr, imgs, arch.vision.heads, arch.patch),
                          "fp32w": clip_oracle.vision_forward(vw, imgs, arch.vision.heads, arch.patch)},
                "text": {"bf16w": clip_oracle.text_forward(tr, toks.long(), arch.text.heads),
                         "fp32w": clip_oracle.text_forward(tw, toks.long(), arch.text.heads)}}
    for tower, got in (("image", gi), ("text", gt)):
        for wk, ref in refs[tower].items():
            mc, md = _dev(got, ref)
            print(f"[measured] ViT-L/14 {tower} tower vs oracle ({wk}): min cos {mc:.6f}  max|d emb| {md:.2e}")
            lo, hi = TOWER_BOUNDS[tower][wk]
            assert mc > lo and md < hi, (tower, wk, mc, md)
    # the image tower does not depend on the batch it runs in (B = 4 here, 512 in the bench)
    assert torch.equal(eng.encode_image(imgs[1:3].cuda()).cpu(), gi[1:3])


def test_config2_full_step_records_vs_oracle(pkg, l14):
    arch, (vw, tw), eng = l14
    B, N, R, D = 512, 8, 1_000_000, arch.embed_dim
    images = pkg.synth.make_images(B, arch.image_size, seed=1).cuda()
    tokens = pkg.synth.make_tokens(B, N, arch.ctx, seed=2).cuda()
    cfg = pkg.ConsistencyConfig()
    k = max(cfg.search_k, cfg.reference_count)
    # ---- the step of bench.py (two streams for the towers, one search over the B*(N+2) rows)
    s_img, s_txt = torch.cuda.Stream(), torch.cuda.Stream()
    ft0 = eng.encode_text(tokens.view(B * (N + 1), arch.ctx), group=N + 1)
    bank = pkg.synth.make_bank(R, D, seed=7, device="cuda:0", dtype=torch.bfloat16)
    bank = pkg.synth.plant_neighbours(bank, ft0.cpu(), per_anchor=1, seed=11)
    eng.set_bank(bank)
    main = torch.cuda.current_stream()
    s_txt.wait_stream(main); s_img.wait_stream(main)
    with torch.cuda.stream(s_txt):
        ft = eng.encode_text(tokens.view(B * (N + 1), arch.ctx), group=N + 1)
    with torch.cuda.stream(s_img):
        fi = eng.encode_image(images)
    main.wait_stream(s_txt); main.wait_stream(s_img)
    assert torch.equal(ft, ft0)
    rows = torch.cat([fi, ft])
    idx, sim, _ = eng.bank_search(rows, k, cfg.similarity_threshold, want_moments=False)
    eng.bank_status()
    tidx, tsim = idx[B:], sim[B:]
    feat = eng.bank_gather(tidx[:, :cfg.reference_count].contiguous())
    rec = eng.consistency(fi, ft.view(B, N + 1, D), cfg, tidx.contiguous(), tsim.contiguous(), feat).cpu().numpy()
    assert np.isfinite(rec[:, :11]).all()
    assert rec[:, 8].mean() > 1.0, "the planted bank must give the consistency kernel references to keep"

    # ---- 16 sampled queries vs the oracle on the SAME embeddings (bar 1e-4, kept-reference indices exact)
    sample = np.linspace(0, B - 1, 16).astype(int)
    fi_c = fi.cpu().numpy()[sample]
    ft_c = ft.view(B, N + 1, D).cpu().numpy()[sample]
    bank_c = bank.float().cpu().numpy()
    ck = tvc_oracle.ConsistencyCheckerOracle(adaptive_threshold=False)
    same = tvc_oracle.detect_batch(fi_c, ft_c, bank_c, checker=ck)
    worst = 0.0
    for col, key in ((0, "original_similarity"), (1, "variant_mean"), (2, "variant_std"), (5, "score_src"),
                     (6, "retrieval_consistency"), (7, "retrieval_std"), (9, "cross_modal_variance"), (10, "overall_exp")):
        err = np.abs(rec[sample, col] - same[key]).max()
        worst = max(worst, err)
        assert err < SCORE_BOUND_SAME_EMB, (key, err)
    kept = np.ascontiguousarray(rec[sample, 12 + N:12 + N + 16]).view(np.int32)
    for j in range(len(sample)):
        want = same["retrieval_indices"][j]
        want = want[want >= 0]
        assert kept[j, :len(want)].tolist() == want.tolist()
    assert (same["retrieval_indices"] >= 0).any()
    print(f"[measured] configs[2] records vs oracle, same embeddings: max |d| = {worst:.2e} over 16 queries")

    # ---- end to end: the oracle's OWN fp32 towers (fp32 weights) on 8 of the sampled queries
    sub = sample[::2]
    with torch.no_grad():
        ri = clip_oracle.vision_forward(vw, images[sub].cpu(), arch.vision.heads, arch.patch)
        rt = clip_oracle.text_forward(tw, tokens[sub].reshape(-1, arch.ctx).cpu().long(), arch.text.heads).view(len(sub), N + 1, D)
    mc_i, md_i = _dev(fi.cpu()[sub], ri)
    mc_t, md_t = _dev(ft.view(B, N + 1, D).cpu()[sub].reshape(-1, D), rt.reshape(-1, D))
    ref = tvc_oracle.detect_batch(ri.numpy(), rt.numpy(), bank_c, checker=tvc_oracle.ConsistencyCheckerOracle(adaptive_threshold=False))
    e_src = np.abs(rec[sub, 5] - ref["score_src"]).max()
    e_s0 = np.abs(rec[sub, 0] - ref["original_similarity"]).max()
    e_exp = np.abs(rec[sub, 10] - ref["overall_exp"]).max()
    print(f"[measured] configs[2] end to end (bf16 HIP towers vs fp32 CPU towers, 8 queries in the B=512 batch): "
          f"image min cos {mc_i:.6f} max|d| {md_i:.2e}; text min cos {mc_t:.6f} max|d| {md_t:.2e}; "
          f"|d score_src| {e_src:.2e} |d s0| {e_s0:.2e} |d overall_exp| {e_exp:.2e}")
    assert mc_i > 0.99998 and mc_t > 0.99993 and md_i < 1e-3 and md_t < 2.5e-3      # measured 0.999993 / 4.8e-4, 0.999967 / 1.2e-3
    assert e_src < SCORE_BOUND_END_TO_END and e_s0 < SCORE_BOUND_END_TO_END
