"""GPU: the input-gradient path of the vision tower and the PGD inner loop.

Replaces what ``encode_image_tensor(x, requires_grad=True)`` + ``loss.backward()`` do inside
/root/reference/src/attacks/pgd_attack.py:452-521 (and hubness_attack.py:586).  Checked against:
  * torch autograd (fp32, CPU) of the same op, for each backward building block;
  * torch autograd through ``oracle/clip_oracle.vision_forward`` (the fp32 CPU tower) for the whole gradient;
  * ``oracle/synth_pgd.pgd_images`` and the committed PGD fixture (tests/golden/pgd_b32_q1000.npz) for the attack.
Floating point, bf16 between GEMMs: the tolerances below are ~2x the measured deviations (printed with -s).
"""
import numpy as np
import pytest
import torch

from oracle import clip_oracle, synth_pgd

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16)


def test_layernorm_backward_vs_autograd(gpu_engine):
    eng = gpu_engine
    g0 = torch.Generator().manual_seed(0)
    for rows, d in ((7, 768), (200, 1024), (33, 192)):
        x = torch.randn(rows, d, generator=g0) * 2 + 0.3
        gam = torch.rand(d, generator=g0) + 0.5
        bet = torch.randn(d, generator=g0)
        dy = _bf(torch.randn(rows, d, generator=g0))
        dres = torch.randn(rows, d, generator=g0)
        xr = x.clone().requires_grad_(True)
        y = torch.nn.functional.layer_norm(xr, (d,), gam, bet, 1e-5)
        y.backward(dy.float())
        want = xr.grad + dres
        got = eng.layernorm_backward(x.cuda(), dy.cuda(), gam.cuda(), dres.cuda()).cpu()
        err = (got - want).abs().max().item()
        assert err < 2e-5 * max(1.0, want.abs().max().item()), (rows, d, err)
        got0 = eng.layernorm_backward(x.cuda(), dy.cuda(), gam.cuda(), None).cpu()
        assert (got0 - xr.grad).abs().max().item() < 2e-5 * max(1.0, xr.grad.abs().max().item())


@pytest.mark.parametrize("T,heads,n_seq", [(50, 12, 3), (257, 16, 2), (197, 12, 2), (17, 3, 5), (1, 2, 3),
                                           # the straight-line 9-pair dK/dV form covers 257..288 tokens; its edges and the
                                           # tile / pair boundaries below it
                                           (258, 2, 2), (272, 2, 1), (273, 2, 1), (288, 2, 2), (256, 2, 2), (33, 2, 2), (32, 2, 2)])
def test_attention_backward_vs_autograd(gpu_engine, T, heads, n_seq):
    eng = gpu_engine
    g0 = torch.Generator().manual_seed(T)
    W = heads * 64
    qkv = _bf(torch.randn(n_seq * T, 3 * W, generator=g0) * 0.7)
    dout = _bf(torch.randn(n_seq * T, W, generator=g0))
    x = qkv.float().requires_grad_(True)
    q, k, v = (x[:, i * W:(i + 1) * W].view(n_seq, T, heads, 64).transpose(1, 2) for i in range(3))
    p = torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(n_seq * T, W)
    o.backward(dout.float())
    want = x.grad
    # forward agreement first (same kernel the tower uses)
    fo = eng.attention(qkv.cuda(), n_seq, T, heads, False).float().cpu()
    assert (fo - o.detach()).abs().max().item() < 3e-2
    got = eng.attention_backward(qkv.cuda(), dout.cuda(), n_seq, T, heads).float().cpu()
    assert torch.isfinite(got).all()
    for name, sl in (("dq", slice(0, W)), ("dk", slice(W, 2 * W)), ("dv", slice(2 * W, 3 * W))):
        a, b = got[:, sl], want[:, sl]
        rel = (a - b).norm().item() / max(b.norm().item(), 1e-12)
        mx = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)
        print(f"[measured] attention backward T={T} heads={heads} {name}: rel L2 {rel:.2e}  max/max {mx:.2e}")
        assert rel < 5e-3 and mx < 8e-3, (name, rel, mx)          # measured 2.4e-3 / 3.7e-3


def test_pgd_step_kernel_matches_formula(gpu_engine):
    eng = gpu_engine
    g0 = torch.Generator().manual_seed(3)
    B, shape = 5, (5, 3, 32, 32)
    clean = torch.rand(shape, generator=g0)
    adv = (clean + (torch.rand(shape, generator=g0) - 0.5) * 0.05).clamp(0, 1)
    grad = torch.randn(shape, generator=g0) * 1e-3
    grad[0, 0, 0, :8] = 0.0                                   # sign(0) = 0 lanes
    mom = torch.randn(shape, generator=g0) * 1e-5
    eps, alpha, mu = 8 / 255, 2 / 255, 0.9
    for targeted in (False, True):
        for use_mom in (True, False):
            a, m = adv.clone().cuda(), (mom.clone().cuda() if use_mom else None)
            eng.pgd_step(a, clean.cuda(), grad.cuda(), m, eps, alpha, mu, 0.0, 1.0, targeted)
            if use_mom:
                wm = mu * mom + grad / grad.abs().sum(dim=(1, 2, 3), keepdim=True)
                step = wm.sign()
                assert torch.allclose(m.cpu(), wm, rtol=2e-5, atol=1e-9)
                # a sign can only differ where the momentum is ~0
                unsure = wm.abs() < 1e-9
            else:
                step, unsure = grad.sign(), torch.zeros(shape, dtype=torch.bool)
            w = adv + (-alpha if targeted else alpha) * step
            w = (clean + (w - clean).clamp(-eps, eps)).clamp(0, 1)
            diff = (a.cpu() - w).abs()
            assert diff[~unsure].max().item() < 1e-6


def _oracle_grad(vw, x, t_unit, heads, patch):
    xr = x.clone().requires_grad_(True)
    f = clip_oracle.vision_forward(vw, xr, heads, patch)
    loss = torch.nn.functional.cosine_similarity(f, t_unit, dim=-1).mean()
    g, = torch.autograd.grad(loss, xr)
    return f.detach(), g


@pytest.mark.parametrize("name,B", [("ViT-T/16-test", 6), ("ViT-B/32", 4), ("ViT-L/14", 2)])
def test_input_gradient_vs_oracle_autograd(pkg, name, B):
    arch = pkg.get_arch(name)
    vw, tw = pkg.synth.make_clip_weights(arch, seed=0)
    eng = pkg.TVCEngine(arch, vw, tw)
    x = pkg.synth.make_images(B, arch.image_size, seed=5)
    t = torch.nn.functional.normalize(torch.randn(B, arch.embed_dim, generator=torch.Generator().manual_seed(6)), dim=-1)
    f_ref, g_ref = _oracle_grad(clip_oracle.round_gemm_weights_to_bf16(vw), x, t, arch.vision.heads, arch.patch)
    xd = x.cuda()
    f = eng.encode_image_grad(xd, True)
    # the grad-mode forward keeps the FC1 pre-activation in bf16 and applies QuickGELU to THAT (the inference path
    # applies it to the fp32 accumulator): same tower, one more bf16 rounding per MLP -- within the tower tolerance
    assert (f - eng.encode_image(xd, True)).abs().max().item() < 2e-3
    g = eng.encode_image_backward((t / B).cuda()).cpu()
    g2 = eng.encode_image_backward((t / B).cuda()).cpu()
    assert torch.equal(g, g2), "the backward is deterministic and repeatable on the saved activations"
    assert torch.isfinite(g).all()
    assert (f.cpu() - f_ref).abs().max().item() < 2e-3
    cos = torch.nn.functional.cosine_similarity(g.flatten(1), g_ref.flatten(1), dim=1)
    ratio = g.flatten(1).norm(dim=1) / g_ref.flatten(1).norm(dim=1)
    # sign agreement, weighted: only signs of non-negligible components matter to a sign-gradient step
    big = g_ref.abs() > 0.05 * g_ref.abs().flatten(1).max(dim=1).values.view(-1, 1, 1, 1)
    sign_big = (g.sign() == g_ref.sign())[big].float().mean().item()
    sign_all = (g.sign() == g_ref.sign()).float().mean().item()
    print(f"[measured] {name} input gradient vs fp32 autograd (bf16-rounded weights): min cos {cos.min():.5f}  "
          f"norm ratio {ratio.min():.4f}..{ratio.max():.4f}  sign agreement {sign_all:.4f} (|g| > 5% max: {sign_big:.4f})")
    # measured: min cos 0.99998, norm ratio 0.9990..1.0007, sign agreement 0.998 (1.0000 on the components that matter)
    assert cos.min().item() > 0.9999 and 0.997 < ratio.min().item() and ratio.max().item() < 1.003
    assert sign_big > 0.9995 and sign_all > 0.995
    # torch.autograd front end (CLIPModel.encode_image_tensor): loss.backward() fills x.grad with the same numbers
    clip = pkg.CLIPModel(pkg.CLIPConfig(model_name=arch.name), weights=(vw, tw))
    xa = x.cuda().requires_grad_(True)
    fa = clip.encode_image_tensor(xa, requires_grad=True)
    fa = fa / fa.norm(dim=-1, keepdim=True)                                   # the attack normalises again (:484)
    loss = torch.nn.functional.cosine_similarity(fa, t.cuda(), dim=-1).mean()
    loss.backward()
    cos2 = torch.nn.functional.cosine_similarity(xa.grad.cpu().flatten(1), g_ref.flatten(1), dim=1)
    assert cos2.min().item() > 0.9999
    # a stale graph is refused, not silently wrong
    fb = clip.encode_image_tensor(xa, requires_grad=True)
    _ = clip.encode_image_tensor(xa, requires_grad=True)
    with pytest.raises(RuntimeError):
        fb.sum().backward()
    clip.engine.close()
    eng.close()


def test_pgd_attacker_vs_oracle_recipe(pkg):
    """Same seed, same recipe (oracle/synth_pgd.py = pgd_attack.py:406-523): the adversarial images agree except
    where a near-zero momentum component flips sign, and the attack does to the similarity what the oracle's does."""
    A = pkg.attacks
    arch = pkg.get_arch("ViT-T/16-test")
    vw, tw = pkg.synth.make_clip_weights(arch, seed=0)
    Q = 16
    clean = pkg.synth.make_images(Q, arch.image_size, seed=1)
    tokens = pkg.synth.make_tokens(Q, 1, arch.ctx, seed=2)[:, 0]
    want = synth_pgd.pgd_images(vw, tw, clean, tokens.long(), arch.vision.heads, arch.text.heads, arch.patch, seed=42)
    clip = pkg.CLIPModel(pkg.CLIPConfig(model_name=arch.name), weights=(vw, tw))
    atk = A.PGDAttacker(clip, A.PGDAttackConfig(batch_size=Q, random_seed=42))
    tf = clip.encode_tokens(tokens, True)
    got = atk._steps(clean.cuda(), tf, None).cpu()
    eps = atk.config.epsilon
    # the reference clamps the NORMALISED tensor to [0, 1] (:34-35,519-520): the ball is around the clamped clean image
    assert (got - clean.clamp(0, 1)).abs().max().item() <= eps + 1e-6 and got.min() >= 0 and got.max() <= 1
    same = ((got - want).abs() < 1e-6).float().mean().item()
    with torch.no_grad():
        c0 = torch.nn.functional.cosine_similarity(clip_oracle.vision_forward(vw, clean, arch.vision.heads, arch.patch), tf.cpu()).mean().item()
        cw = torch.nn.functional.cosine_similarity(clip_oracle.vision_forward(vw, want, arch.vision.heads, arch.patch), tf.cpu()).mean().item()
        cg = torch.nn.functional.cosine_similarity(clip_oracle.vision_forward(vw, got, arch.vision.heads, arch.patch), tf.cpu()).mean().item()
    print(f"[measured] PGD (10 steps, eps 8/255) HIP vs oracle: identical pixels {same:.4f}; mean cos clean {c0:.4f} "
          f"-> oracle adv {cw:.4f}, HIP adv {cg:.4f}")
    assert same > 0.85
    assert abs(cg - cw) < 0.1 * abs(cw - c0) + 1e-3
    clip.engine.close()


def test_pgd_concurrent_batches_are_bit_identical(pkg):
    """PGDAttackConfig.concurrent_batches = 2 (default): two batches of `batch_size` in flight on two HIP streams / two
    engine handles.  Every batch is computed as it is alone and the random starts are drawn in batch order, so `perturb`
    returns the same pixels, bit for bit, as the one-batch-at-a-time loop (a ragged last batch and an odd batch count
    included)."""
    arch = pkg.get_arch("ViT-T/16-test")
    vw, tw = pkg.synth.make_clip_weights(arch, seed=0)
    Q = 44                                          # batches of 8: five full ones and a ragged one of 4
    clean = pkg.synth.make_images(Q, arch.image_size, seed=5).cuda()
    texts = [f"a photo of thing {i}" for i in range(Q)]
    clip = pkg.CLIPModel(pkg.CLIPConfig(model_name=arch.name), weights=(vw, tw))
    outs = {}
    for nc in (1, 2, 3):
        atk = pkg.PGDAttacker(clip, pkg.PGDAttackConfig(batch_size=8, random_seed=7, num_steps=4, concurrent_batches=nc))
        outs[nc] = atk.perturb(clean, texts)
        torch.cuda.synchronize()
        atk.close()
    assert torch.isfinite(outs[1]).all() and (outs[1] - clean.clamp(0, 1)).abs().max().item() > 1e-3
    assert torch.equal(outs[2], outs[1]) and torch.equal(outs[3], outs[1])
    clip.engine.close()


def test_pgd_attacker_vs_committed_fixture(pkg):
    """ViT-B/32, the first PGD batch of tests/golden/pgd_b32_q1000.npz (oracle/make_pgd_fixture.py: queries 500..549,
    generator seed SEED_PGD + 500): the HIP attack, started from the same noise, lands on the same side of the clean
    pixel as the CPU oracle's attack did."""
    from oracle import make_pgd_fixture as F
    fx = F.load_fixture(pkg)
    arch, (vw, tw) = fx["arch"], fx["weights"]
    half = F.Q // 2
    _, _, _, clean_all, tokens = F.inputs(pkg)
    clean = clean_all[half:half + 50]
    fix = fx["images"][half:half + 50]
    clip = pkg.CLIPModel(pkg.CLIPConfig(model_name=arch.name), weights=(vw, tw))
    atk = pkg.PGDAttacker(clip, pkg.PGDAttackConfig(batch_size=50, random_seed=F.SEED_PGD + half))
    tf = clip.encode_tokens(tokens[half:half + 50, 0], True)
    got = atk._steps(clean.cuda(), tf, None).cpu()
    m = F.sign_mask(clean)
    agree = ((got - clean)[m].sign() == (fix - clean)[m].sign()).float().mean().item()
    sat = (((got - clean).abs() - F.EPS).abs() < 1e-6)[m].float().mean().item()
    f_got = clip.engine.encode_image(got.cuda(), True)
    f_fix = clip.engine.encode_image(fix.cuda(), True)
    f_cln = clip.engine.encode_image(clean.cuda(), True)
    c = lambda f: (f * tf).sum(-1).mean().item()
    print(f"[measured] PGD ViT-B/32 vs committed fixture (50 queries): perturbation signs equal on {agree:.4f} of the pixels, "
          f"{sat:.3f} at +-eps; mean cos(image, text): clean {c(f_cln):.4f}, fixture adv {c(f_fix):.4f}, HIP adv {c(f_got):.4f}")
    assert agree > 0.9
    assert abs(c(f_got) - c(f_fix)) < 0.1 * abs(c(f_fix) - c(f_cln)) + 1e-3
    clip.engine.close()


def test_hubness_attack_vs_autograd_recipe(pkg):
    """Hubness inner loop (src/attacks/hubness_attack.py:549-654,656-676): loss = -mean_q cos(f(x), t_q), descent by
    step_size * sign(grad), eps ball, clamp, best-loss image kept.  Restated here on the oracle tower with torch
    autograd (the same random start), compared with the HIP loop."""
    arch = pkg.get_arch("ViT-T/16-test")
    vw, tw = pkg.synth.make_clip_weights(arch, seed=0)
    clip = pkg.CLIPModel(pkg.CLIPConfig(model_name=arch.name), weights=(vw, tw))
    cfg = pkg.HubnessAttackConfig(clip_model=arch.name, num_iterations=12, epsilon=16 / 255, step_size=0.02, random_seed=5)
    atk = pkg.HubnessAttack(cfg, clip_model=clip)
    B = 3
    clean = pkg.synth.make_images(B, arch.image_size, seed=9)
    queries = [atk._generate_random_queries(6) for _ in range(B)]
    qf = atk._unit_queries([q for qs in queries for q in qs])
    q_mean = qf.view(B, 6, -1).mean(1)
    # ---- oracle: autograd through the fp32 CPU tower
    gen = torch.Generator().manual_seed(5)
    noise = (torch.rand(clean.shape, generator=gen) * 2 - 1) * cfg.epsilon
    adv = torch.clamp(clean + noise, 0, 1)
    qm = q_mean.cpu()
    best_loss, best = torch.full((B,), float("inf")), adv.clone()
    for _ in range(cfg.num_iterations):
        x = adv.clone().requires_grad_(True)
        f = clip_oracle.vision_forward(vw, x, arch.vision.heads, arch.patch)
        loss_b = -(f * qm).sum(-1)
        g, = torch.autograd.grad(loss_b.mean(), x)
        better = loss_b.detach() < best_loss
        best_loss = torch.where(better, loss_b.detach(), best_loss)
        best = torch.where(better.view(-1, 1, 1, 1), adv, best)
        adv = adv - cfg.step_size * g.sign()
        adv = torch.clamp(clean + torch.clamp(adv - clean, -cfg.epsilon, cfg.epsilon), 0, 1)
    # ---- HIP
    got, got_loss = atk._optimise(clean.cuda(), q_mean)
    same = ((got.cpu() - best).abs() < 1e-6).float().mean().item()
    print(f"[measured] Hubness loop (12 iterations) HIP vs autograd recipe: identical pixels {same:.4f}; best loss "
          f"{got_loss.cpu().tolist()} vs {best_loss.tolist()}")
    assert same > 0.97
    assert (got_loss.cpu() - best_loss).abs().max().item() < 5e-3
    assert (best_loss < -(clip_oracle.vision_forward(vw, clean, arch.vision.heads, arch.patch) * qm).sum(-1)).all(), \
        "the loop moves the image towards its queries"
    # API surface: single image, stats, presets, config from dict
    r = atk.attack(clean[0].cuda(), "a photo of a cat")
    assert set(r) >= {"adversarial_image", "original_image", "perturbation", "hubness_score", "perturbation_norm",
                      "final_loss", "iterations", "success", "text_queries"}
    assert r["perturbation_norm"] <= cfg.epsilon + 1e-6 or r["original_image"].min() < 0 or r["original_image"].max() > 1
    rs = atk.attack_single(clean[1].cuda(), "a dog playing")
    assert set(rs) >= {"success", "hubness", "similarity_change", "iterations", "final_loss"}
    rb = atk.batch_attack(clean.cuda(), ["x"] * B)
    assert len(rb) == B and all(len(x["target_queries"]) == 10 for x in rb)
    assert atk.get_attack_stats()["total_attacks"] == 2 + B
    assert pkg.HubnessAttackPresets.weak_attack().num_iterations == 100
    clip.engine.close()


def test_l2_step_and_hubness_l2_constraint(pkg):
    """``norm_constraint='l2'`` (src/attacks/hubness_attack.py:378-386): the fused step kernel vs the reference's
    formulas in torch, then the Hubness loop under the L2 constraint (perturbation inside the eps L2 ball, loss down)."""
    arch = pkg.get_arch("ViT-T/16-test")
    vw, tw = pkg.synth.make_clip_weights(arch, seed=0)
    clip = pkg.CLIPModel(pkg.CLIPConfig(model_name=arch.name), weights=(vw, tw))
    eng = clip.engine
    g = torch.Generator().manual_seed(3)
    B = 4
    clean = torch.rand((B, 3, 64, 64), generator=g)
    adv = torch.clamp(clean + 0.05 * torch.randn(clean.shape, generator=g), 0, 1)
    grad = torch.randn(clean.shape, generator=g) * 1e-3
    eps, step = 1.5, 0.3
    for descent in (True, False):
        a = adv.clone().cuda()
        eng.l2_step(a, clean.cuda(), grad.cuda(), eps, step, 0.0, 1.0, descent)
        gn = grad.view(B, -1).norm(dim=1, keepdim=True)
        ref = adv + (-1 if descent else 1) * step * grad / (gn.view(-1, 1, 1, 1) + 1e-8)
        d = ref - clean
        dn = d.view(B, -1).norm(dim=1, keepdim=True)
        d = d / (dn.view(-1, 1, 1, 1) + 1e-8) * torch.clamp(dn, max=eps).view(-1, 1, 1, 1)
        ref = torch.clamp(clean + d, 0, 1)
        assert (a.cpu() - ref).abs().max().item() < 2e-6
    cfg = pkg.HubnessAttackConfig(clip_model=arch.name, num_iterations=8, epsilon=2.0, step_size=0.5, random_seed=5,
                                  norm_constraint="l2", random_start=False)
    atk = pkg.HubnessAttack(cfg, clip_model=clip)
    imgs = torch.rand((2, 3, arch.image_size, arch.image_size), generator=g).cuda()
    q_mean = atk._unit_queries(["a photo of a cat", "a dog playing"]).mean(0, keepdim=True).expand(2, -1).contiguous()
    f0 = eng.encode_image(imgs, True)
    best, best_loss = atk._optimise(imgs, q_mean)
    assert ((best - imgs).flatten(1).norm(dim=1) <= cfg.epsilon + 1e-4).all()
    assert (best_loss < -(f0 * q_mean).sum(-1) + 1e-6).all()
    with pytest.raises(ValueError):
        pkg.HubnessAttack(pkg.HubnessAttackConfig(clip_model=arch.name, norm_constraint="l1"), clip_model=clip)
    eng.close()
