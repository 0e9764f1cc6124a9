"""GPU: the latent-diffusion reference generator (BASELINE configs[4]) -- UNet blocks,
streaming attention, VAE decoder, PLMS sampling loop -- HIP (through the C-ABI) vs ``oracle/sd_oracle.py`` (PyTorch
fp32 on the CPU) at the REAL Stable-Diffusion-v1.5 geometry (block widths 320 / 640 / 1280 / 1280, 8 heads of
40 / 80 / 160, cross-attention onto 77 x 768 text states; VAE 128 / 256 / 512 / 512), seeded random-init weights.

**Parity unpinned against the reference** (oracle/sd_oracle.py header): diffusers is not importable, the wrapper and
the weights are absent, the reference holds no vectors at this boundary -- the oracle restates the published
algorithms of the classes named by the config.json files the reference holds.

The HIP path keeps activations in bf16 between GEMMs (fp32 accumulation and statistics), the oracle is fp32
throughout on the same fp32 weights: bounds are ~2x the measured deviations, printed with ``pytest -s``.
"""
import numpy as np
import pytest
import torch

from oracle import sd_oracle

pytestmark = pytest.mark.gpu


def rel(got: torch.Tensor, ref: torch.Tensor):
    """(relative L2 error, max |d| / std(ref))."""
    d = got.double().cpu() - ref.double()
    return (d.norm() / ref.double().norm()).item(), (d.abs().max() / ref.double().std()).item()


@pytest.fixture(scope="module")
def sd(pkg):
    arch = pkg.SDArch()
    uw, vw = pkg.make_sd_weights(arch, seed=0, device="cuda")          # 0.9 G parameters: drawn on the device, kept on the host
    eng = pkg.TVCEngine()
    k = pkg.SDKernels(eng, arch, uw, vw)
    yield arch, uw, vw, k
    eng.close()


@pytest.mark.parametrize("n,heads,dh,Tq,Tk", [(1, 8, 40, 4096, 4096), (2, 8, 80, 1024, 1024), (2, 8, 160, 256, 256),
                                             (3, 8, 160, 64, 64), (2, 8, 40, 1024, 77), (2, 8, 160, 64, 77),
                                             (1, 2, 64, 100, 50), (1, 3, 8, 70, 130),
                                             # enough (sample, head, 256-query block) items for the eight-wave form, which at
                                             # head dims <= 48 runs two workgroups per CU on ONE staging register set: ragged
                                             # query blocks, a ragged last key tile, odd / short key-tile counts
                                             (22, 8, 40, 700, 700), (22, 8, 40, 600, 77), (24, 8, 32, 520, 130), (22, 8, 80, 700, 200)])
def test_streaming_attention_vs_fp64(pkg, sd, n, heads, dh, Tq, Tk):
    k = sd[3]
    g = torch.Generator().manual_seed(Tq + dh)
    C = heads * dh
    q, kk, v = (torch.randn((n * T, C), generator=g).to(torch.bfloat16) for T in (Tq, Tk, Tk))
    out = k.attention(q, kk, v, n, heads).float().cpu()
    sp = lambda t, T: t.double().view(n, T, heads, dh).transpose(1, 2)
    s = sp(q, Tq) @ sp(kk, Tk).transpose(-1, -2) * dh ** -0.5
    ref = (s.softmax(-1) @ sp(v, Tk)).transpose(1, 2).reshape(n * Tq, C)
    r2, rm = rel(out, ref)
    print(f"[measured] streaming attention n={n} heads={heads} dh={dh} Tq={Tq} Tk={Tk}: rel L2 {r2:.2e}  max|d|/std {rm:.2e}")
    assert r2 < 5e-3 and rm < 5e-2          # measured 2.1e-3 .. 2.3e-3 / 1.4e-2 .. 3.4e-2 (bf16 probabilities and outputs; the max grows with the element count)


@pytest.mark.parametrize("kind,prefix,cin,cout,hw", [(3, "down_blocks.0.resnets.0.conv1.", 320, 320, 16),
                                                     (4, "down_blocks.1.downsamplers.0.conv.", 640, 640, 16),
                                                     (5, "up_blocks.1.upsamplers.0.conv.", 1280, 1280, 8)])
def test_conv_kinds_vs_oracle(pkg, sd, kind, prefix, cin, cout, hw, n=2):
    arch, uw, vw, k = sd
    x = torch.randn((n, cin, hw, hw), generator=torch.Generator().manual_seed(kind))
    got = k.block(kind, prefix, x, cout)
    xs = x.to(torch.bfloat16).float()          # the block converts its fp32 input to bf16 rows
    F = torch.nn.functional
    if kind == 5:
        xs = F.interpolate(xs, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xs, uw[prefix + "weight"], uw[prefix + "bias"], stride=2 if kind == 4 else 1, padding=1)
    r2, rm = rel(got, ref)
    print(f"[measured] conv kind {kind} {prefix}: rel L2 {r2:.2e} max|d|/std {rm:.2e}")
    assert got.shape == ref.shape and r2 < 5e-3 and rm < 4e-2      # measured 2.35e-3 / 1.4e-2 .. 1.7e-2


@pytest.mark.parametrize("prefix,cin,cout,hw", [("down_blocks.2.resnets.1.conv1.", 1280, 1280, 16),     # 6 144 tokens x 1280: 120 tiles
                                                ("down_blocks.1.resnets.1.conv1.", 640, 640, 32)])      # 24 576 tokens x 640: 288 tiles
def test_conv_at_generation_batch_takes_the_split_k_paths(pkg, sd, prefix, cin, cout, hw):
    """24 samples (12 images x classifier-free guidance), the batch a generation runs at: the 16 x 16 level's convolutions
    split K (a fixed factor chosen from the per-sample shape: tvc_sd.cpp, Run::fixed_split), the 32 x 32 level's do not.
    Same bound as the two-sample cases, and every sample equals its own two-sample result bit for bit."""
    test_conv_kinds_vs_oracle(pkg, sd, 3, prefix, cin, cout, hw, n=24)
    arch, uw, vw, k = sd
    x = torch.randn((24, cin, hw, hw), generator=torch.Generator().manual_seed(3))
    assert torch.equal(k.block(3, prefix, x, cout)[5:7], k.block(3, prefix, x[5:7], cout))


@pytest.mark.parametrize("prefix,cin,cout,hw,vae", [("down_blocks.0.resnets.0.", 320, 320, 32, False),
                                                    ("up_blocks.1.resnets.2.", 1920, 1280, 16, False),
                                                    ("up_blocks.3.resnets.0.", 960, 320, 32, False),
                                                    ("decoder.up_blocks.2.resnets.0.", 512, 256, 32, True)])
def test_resnet_block_vs_oracle(pkg, sd, prefix, cin, cout, hw, vae):
    arch, uw, vw, k = sd
    g = torch.Generator().manual_seed(cin)
    x = torch.randn((2, cin, hw, hw), generator=g)
    temb = None if vae else torch.randn((2, arch.time_dim), generator=g)
    got = k.block(0, prefix, x, cout, temb=temb, vae=vae)
    with torch.no_grad():
        ref = sd_oracle.resnet(vw if vae else uw, prefix, x, temb, arch.norm_groups, 1e-6 if vae else arch.norm_eps)
    r2, rm = rel(got, ref)
    print(f"[measured] resnet {prefix}: rel L2 {r2:.2e} max|d|/std {rm:.2e}")
    assert r2 < 8e-3 and rm < 6e-2           # measured 2.9e-3 .. 3.9e-3 / 2.4e-2 .. 2.8e-2


@pytest.mark.parametrize("prefix,c,hw", [("down_blocks.0.attentions.0.", 320, 32), ("down_blocks.1.attentions.1.", 640, 16),
                                         ("mid_block.attentions.0.", 1280, 8)])
def test_transformer_block_vs_oracle(pkg, sd, prefix, c, hw):
    arch, uw, vw, k = sd
    g = torch.Generator().manual_seed(c)
    x = torch.randn((2, c, hw, hw), generator=g)
    ctx = torch.randn((2, arch.ctx, arch.cross_attention_dim), generator=g)
    got = k.block(1, prefix, x, c, ctx=ctx)
    with torch.no_grad():
        ref = sd_oracle.transformer(uw, prefix, x, ctx, arch.heads, arch.norm_groups)
    r2, rm = rel(got, ref)
    print(f"[measured] transformer {prefix} (head_dim {c // arch.heads}): rel L2 {r2:.2e} max|d|/std {rm:.2e}")
    assert r2 < 7e-3 and rm < 6e-2           # measured 3.4e-3 / 2.0e-2 .. 2.6e-2


def test_vae_attention_block_vs_oracle(pkg, sd):
    arch, uw, vw, k = sd
    x = torch.randn((2, 512, 16, 16), generator=torch.Generator().manual_seed(9))
    got = k.block(2, "decoder.mid_block.attentions.0.", x, 512, vae=True)
    with torch.no_grad():
        ref = sd_oracle.vae_attention(vw, "decoder.mid_block.attentions.0.", x, arch.norm_groups)
    r2, rm = rel(got, ref)
    print(f"[measured] VAE attention block: rel L2 {r2:.2e} max|d|/std {rm:.2e}")
    assert r2 < 5e-3 and rm < 6e-2           # measured 2.4e-3 / 2.6e-2


def test_unet_forward_vs_oracle(pkg, sd):
    """One full UNet evaluation (all 4 down / mid / 4 up blocks, 16 transformers) on 2 samples of 16 x 16 latents."""
    arch, uw, vw, k = sd
    g = torch.Generator().manual_seed(4)
    lat = torch.randn((2, 4, 16, 16), generator=g)
    ctx = torch.randn((2, arch.ctx, arch.cross_attention_dim), generator=g)
    got = k.unet(lat, 951.0, ctx)
    with torch.no_grad():
        ref = sd_oracle.unet_forward(uw, arch, lat, 951, ctx)
    r2, rm = rel(got, ref)
    print(f"[measured] UNet forward (16 x 16 latents, t = 951): rel L2 {r2:.2e} max|d|/std {rm:.2e}")
    assert torch.isfinite(got).all() and r2 < 2.5e-2 and rm < 1e-1      # measured 1.2e-2 / 4.6e-2
    # another timestep goes through the time embedding differently
    got2 = k.unet(lat, 1.0, ctx)
    with torch.no_grad():
        ref2 = sd_oracle.unet_forward(uw, arch, lat, 1, ctx)
    assert rel(got2, ref2)[0] < 2.5e-2 and (got2 - got).abs().max().item() > 1e-3


def test_unet_forward_at_64x64_latents_vs_oracle(pkg, sd):
    """The ASSEMBLED UNet at the geometry BASELINE configs[4] generates at: 2 samples of 64 x 64 latents (T = 4096
    self-attention at head_dim 40, 120-tile 9-plane convolutions, all 25 blocks), against the fp32 CPU oracle evaluated
    here (about 25 s on the box's cores) -- same bound as the 16 x 16 evaluation."""
    arch, uw, vw, k = sd
    g = torch.Generator().manual_seed(14)
    lat = torch.randn((2, 4, 64, 64), generator=g)
    ctx = torch.randn((2, arch.ctx, arch.cross_attention_dim), generator=g)
    got = k.unet(lat, 601.0, ctx)
    with torch.no_grad():
        ref = sd_oracle.unet_forward(uw, arch, lat, 601, ctx)
    r2, rm = rel(got, ref)
    print(f"[measured] UNet forward (64 x 64 latents, 2 samples, t = 601): rel L2 {r2:.2e} max|d|/std {rm:.2e}")
    assert torch.isfinite(got).all() and r2 < 2.5e-2 and rm < 1e-1


def test_vae_decode_vs_oracle(pkg, sd):
    arch, uw, vw, k = sd
    lat = torch.randn((2, 4, 16, 16), generator=torch.Generator().manual_seed(5))
    got = k.vae_decode(lat)
    with torch.no_grad():
        ref = (sd_oracle.vae_decode(vw, arch, lat / arch.vae_scaling) / 2 + 0.5).clamp(0, 1)
    d = (got.cpu() - ref).abs()
    print(f"[measured] VAE decode (16 x 16 latents -> 128 x 128 pixels in [0, 1]): max |d| {d.max().item():.2e} mean |d| {d.mean().item():.2e}")
    assert got.shape == (2, 3, 128, 128) and d.max().item() < 4e-2 and d.mean().item() < 4e-3      # measured 1.9e-2 / 2.0e-3


def test_sampling_loop_vs_oracle(pkg, sd):
    """PNDM (PLMS) + classifier-free guidance: 5 scheduler steps (6 UNet evaluations on 2n samples) from the same noise."""
    arch, uw, vw, k = sd
    g = torch.Generator().manual_seed(6)
    n, steps, guidance = 2, 5, 7.5
    cond = torch.randn((n, arch.ctx, arch.cross_attention_dim), generator=g)
    uncond = torch.randn((n, arch.ctx, arch.cross_attention_dim), generator=g)
    lat0 = torch.randn((n, 4, 8, 8), generator=g)            # 8 x 8 latents: the smallest size the four-level UNet takes
    lat, img = k.generate(cond, uncond, lat0, steps, guidance, decode=True)
    with torch.no_grad():
        ref = sd_oracle.generate(uw, vw, arch, cond, uncond, lat0, steps, guidance, return_latents=True)
    r2, rm = rel(lat, ref)
    print(f"[measured] sampling loop, {steps} PLMS steps, guidance {guidance}: final latents rel L2 {r2:.2e} max|d|/std {rm:.2e}")
    assert torch.isfinite(lat).all() and r2 < 3e-2           # measured 1.35e-2 (16 x 16 latents, round 3)
    assert img.shape == (n, 3, 64, 64) and float(img.min()) >= 0.0 and float(img.max()) <= 1.0
    # the scheduler arithmetic alone (same eps on both sides would be exact): timesteps visited
    sch = sd_oracle.PNDMOracle(arch)
    assert sch.set_timesteps(steps) == [801, 601, 601, 401, 201, 1]


def test_sampling_loop_20_steps_vs_oracle(pkg, sd):
    """The reference's fast setting (experiments/defenses/generative_ref.py: 20 steps; src/sd_ref.py:226-230 defaults to
    50): 20 PLMS steps = 21 UNet evaluations with guidance 7.5 at 8 x 8 latents (the CPU oracle's 21 evaluations of the full
    SD-1.5 UNet take 70 s at 16 x 16), one image.  The deviation of the final latents is MEASURED (printed) and bounded at
    ~2x: bf16 activations inside a 21-evaluation feedback loop (8.9e-3 at 16 x 16 latents, gpurun_out/r04_t1.log)."""
    arch, uw, vw, k = sd
    g = torch.Generator().manual_seed(16)
    n, steps, guidance = 1, 20, 7.5
    cond = torch.randn((n, arch.ctx, arch.cross_attention_dim), generator=g)
    uncond = torch.randn((n, arch.ctx, arch.cross_attention_dim), generator=g)
    lat0 = torch.randn((n, 4, 8, 8), generator=g)
    lat, _ = k.generate(cond, uncond, lat0, steps, guidance, decode=False)
    with torch.no_grad():
        ref = sd_oracle.generate(uw, vw, arch, cond, uncond, lat0, steps, guidance, return_latents=True)
    r2, rm = rel(lat, ref)
    print(f"[measured] sampling loop, {steps} PLMS steps (21 UNet evaluations), guidance {guidance}: final latents rel L2 {r2:.2e} "
          f"max|d|/std {rm:.2e}")
    assert torch.isfinite(lat).all() and r2 < 2e-2           # measured 8.9e-3


def test_generate_validates_sizes_and_chunks_by_the_arena_budget(pkg, sd):
    """tvc_sd_generate rejects latent sizes the UNet cannot halve and double back (H = 12: 12 -> 6 -> 3 -> 2 down, 2 -> 4 ->
    8 -> 16 up) instead of reading mismatched skip tensors, and generates a batch that exceeds TVC_OPT_SD_ARENA_BYTES in
    chunks of whole sampling loops: the images are BIT-identical to the one-pass ones."""
    arch, uw, vw, k = sd
    g = torch.Generator().manual_seed(18)
    n = 5
    cond = torch.randn((n, arch.ctx, arch.cross_attention_dim), generator=g)
    uncond = torch.randn((n, arch.ctx, arch.cross_attention_dim), generator=g)
    with pytest.raises(pkg.TVCError) as e:
        k.generate(cond, uncond, torch.randn((n, 4, 12, 12), generator=g), 3, 7.5, decode=False)
    assert e.value.code == pkg._lib.TVC_E_INVALID
    with pytest.raises(pkg.TVCError):
        k.generate(cond, uncond, torch.randn((n, 4, 8, 12), generator=g), 3, 7.5, decode=True)      # VAE attention: H * W % 64
    lat0 = torch.randn((n, 4, 16, 16), generator=g)
    one, _ = k.generate(cond, uncond, lat0, 3, 7.5, decode=False)
    k.engine.set_option(pkg._lib.TVC_OPT_SD_ARENA_BYTES, 1 << 28)           # 256 MiB: two images' worth at 16 x 16 latents
    try:
        parts, _ = k.generate(cond, uncond, lat0, 3, 7.5, decode=False)
    finally:
        k.engine.set_option(pkg._lib.TVC_OPT_SD_ARENA_BYTES, 48 << 30)
    d = (parts - one).abs().max().item() / one.abs().max().item()
    print(f"[measured] chunked vs one-pass generation (5 images, 3 steps): max |d| / max |x| {d:.2e}")
    assert torch.equal(parts, one)            # an image does not depend on its batch mates (src/sd_ref.py:389-412: seed policy)


def test_two_stream_guidance_halves_are_bit_identical(pkg, sd):
    """TVC_OPT_SD_STREAMS = 2 (default): the unconditional and the conditional half of every UNet evaluation run on two HIP
    streams, each in its own half of the arena; = 1: one launch sequence over both halves.  Same images, bit for bit (a
    sample's arithmetic does not depend on its batch mates), with and without the arena chunking, latents and decoded."""
    arch, uw, vw, k = sd
    g = torch.Generator().manual_seed(19)
    n = 3
    cond = torch.randn((n, arch.ctx, arch.cross_attention_dim), generator=g)
    uncond = torch.randn((n, arch.ctx, arch.cross_attention_dim), generator=g)
    lat0 = torch.randn((n, 4, 16, 16), generator=g)
    two, img2 = k.generate(cond, uncond, lat0, 4, 7.5, decode=True)
    k.engine.set_option(pkg._lib.TVC_OPT_SD_STREAMS, 1)
    try:
        one, img1 = k.generate(cond, uncond, lat0, 4, 7.5, decode=True)
    finally:
        k.engine.set_option(pkg._lib.TVC_OPT_SD_STREAMS, 2)
    assert torch.isfinite(two).all() and torch.equal(two, one) and torch.equal(img2, img1)
    again, _ = k.generate(cond, uncond, lat0, 4, 7.5, decode=False)         # and the run repeats itself
    assert torch.equal(again, two)
    with pytest.raises(pkg.TVCError):
        k.engine.set_option(pkg._lib.TVC_OPT_SD_STREAMS, 3)


def test_preprocess_images_matches_torch_antialias(pkg, sd):
    """``tvc_preprocess_images`` (resize + centre crop + normalise on the device) vs torch's antialiased interpolate on
    the CPU (the PIL filter semantics on float pixels): bicubic short-side + crop (CLIP preprocess) and bilinear
    both-sides (torchvision Resize((224, 224)) of experiments/defenses/generative_ref.py:55-59)."""
    F = torch.nn.functional
    eng = sd[3].engine
    g = torch.Generator().manual_seed(8)
    for (H, W) in ((512, 512), (300, 420), (224, 224), (150, 100)):
        x = torch.rand((2, 3, H, W), generator=g)
        mean, std = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
        got = eng.preprocess_images(x.cuda(), 224, mean, std, bicubic=True, keep_aspect=True).cpu()
        s = 224 / min(H, W)
        Hr, Wr = (224, max(224, int(W * 224 / H + 0.5))) if H <= W else (max(224, int(H * 224 / W + 0.5)), 224)
        r = F.interpolate(x, size=(Hr, Wr), mode="bicubic", antialias=True, align_corners=False)
        oy, ox = (Hr - 224) // 2, (Wr - 224) // 2
        r = r[:, :, oy:oy + 224, ox:ox + 224]
        ref = (r - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
        assert (got - ref).abs().max().item() < 2e-4, (H, W, (got - ref).abs().max().item())
        got2 = eng.preprocess_images(x.cuda(), 224, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225), bicubic=False, keep_aspect=False).cpu()
        r2 = F.interpolate(x, size=(224, 224), mode="bilinear", antialias=True, align_corners=False)
        ref2 = (r2 - torch.tensor((0.485, 0.456, 0.406)).view(1, 3, 1, 1)) / torch.tensor((0.229, 0.224, 0.225)).view(1, 3, 1, 1)
        assert (got2 - ref2).abs().max().item() < 2e-4, (H, W)


def test_config4_smoke_sd_reference_generator_feeds_the_detector(pkg):
    """BASELINE configs[4] in miniature: 2 prompts x 3 images x 4 PLMS steps at the full 64 x 64 latent / 512 x 512 pixel
    geometry -> references -> CLIP embeddings -> the detector's sd_reference score (src/detector.py:503-557), checked
    against the oracle's arithmetic on the SAME embeddings (1e-4); plus the defence detector's generative branch."""
    from oracle import tvc_oracle
    arch = pkg.get_arch("ViT-L/14")
    clip = pkg.CLIPModel(pkg.CLIPConfig(model_name="ViT-L/14"), weights=pkg.synth.make_clip_weights(arch, seed=0))
    gen = pkg.SDReferenceGenerator(pkg.SDReferenceConfig(num_images_per_prompt=3, num_inference_steps=4, use_text_variants=False,
                                                         filter_low_quality=False, enable_cache=False, random_init=True),
                                   clip_model=clip)
    assert isinstance(gen.sd_model, pkg.StableDiffusionModel) and gen.sd_model.text_engine is clip.engine
    # without weights and without the opt-in there is NO model (the reference's load failure, src/sd_ref.py:291-317): every
    # call reports an error, the detector scores sd_reference 0.0 + 'error' and carries on
    none = pkg.SDReferenceGenerator(pkg.SDReferenceConfig(use_text_variants=False), clip_model=clip)
    assert none.sd_model is None and "error" in none.generate_reference_images("x")
    with pytest.raises(RuntimeError):
        pkg.StableDiffusionModel(pkg.SDModelConfig(), clip_model=clip)
    with pytest.raises(ValueError):
        pkg.StableDiffusionModel(pkg.SDModelConfig(unet_weights="/nonexistent/unet.safetensors"), clip_model=clip)
    texts = ["a dog running on the beach", "two people riding bicycles in a city street"]
    feats, counts = gen.reference_features(texts, 3)
    assert counts == [3, 3] and feats.shape == (6, arch.embed_dim) and torch.isfinite(feats).all()
    assert (feats.norm(dim=-1) - 1).abs().max().item() < 1e-4
    # different prompts / seeds give different references; the same call again gives the same ones (seed policy)
    assert (feats[0] - feats[1]).abs().max().item() > 1e-4 and (feats[0] - feats[3]).abs().max().item() > 1e-4
    feats2, _ = gen.reference_features(texts, 3)
    assert torch.equal(feats, feats2)
    # ---- the detector consumes them: sd_reference = 1 - mean cos(image, ref_j) on the same embeddings
    images = pkg.synth.make_images(2, arch.image_size, seed=3).cuda()
    det = pkg.AdversarialDetector(pkg.DetectorConfig(clip_model="ViT-L/14", num_reference_images=3, use_text_variants=False),
                                  clip_model=clip, sd_generator=gen)
    res = det.batch_detect(images, texts, methods=["sd_reference", "consistency"])
    fi = clip.engine.encode_image(images).cpu().numpy().astype(np.float64)
    fr = feats.cpu().numpy().astype(np.float64).reshape(2, 3, -1)
    for i, r in enumerate(res):
        want, _ = tvc_oracle.sd_reference_score(fr[i] @ fi[i])           # rows are unit vectors: cosines
        got = r["detection_scores"]["sd_reference"]
        assert abs(got - want) < 1e-4, (got, want)
        assert r["detection_details"]["sd_reference"]["num_references"] == 3
        agg = (0.4 * got + 0.2 * r["detection_scores"]["consistency"]) / 0.6
        assert abs(r["aggregated_score"] - agg) < 1e-6
    # ---- PIL-returning API of the reference (src/sd_ref.py:389-399) and the generative branch of the defence detector
    pil = gen.sd_model.generate_image(prompt=texts[0], num_images=1, seed=0, num_inference_steps=2, height=512, width=512)
    assert len(pil) == 1 and pil[0].size == (512, 512)
    dd = pkg.MultiModalDefenseDetector(clip, sd_model=gen.sd_model,
                                       config=pkg.DetectionConfig(use_text_variants=False, use_retrieval_ref=False,
                                                                  generation_count=2, adaptive_threshold=False))
    dd.generative_generator.config.num_inference_steps = 2
    dd.generative_generator.config.seed = 5
    out = dd.detect(images[:1], texts[0], return_details=True)
    assert "generative_consistency" in out["details"]["consistency_scores"]
    assert np.isfinite(out["consistency_score"])
    clip.engine.close()


def test_unet_and_vae_shapes_and_batch_split_invariance(pkg, sd):
    """Non-square latents, a single sample, and the VAE's image chunking (6 images of 512 x 512 are decoded in two chunks):
    every sample's result is independent of its batch mates (no cross-sample arithmetic anywhere: statistics are per image)."""
    arch, uw, vw, k = sd
    g = torch.Generator().manual_seed(12)
    lat = torch.randn((1, 4, 16, 24), generator=g)
    ctx = torch.randn((1, arch.ctx, arch.cross_attention_dim), generator=g)
    got = k.unet(lat, 500.0, ctx)
    with torch.no_grad():
        ref = sd_oracle.unet_forward(uw, arch, lat, 500, ctx)
    r2, _ = rel(got, ref)
    print(f"[measured] UNet forward, 1 sample of 16 x 24 latents: rel L2 {r2:.2e}")
    assert got.shape == (1, 4, 16, 24) and r2 < 2.5e-2
    # the same sample inside a batch of 3 (the other samples differ)
    lat3 = torch.cat([torch.randn((1, 4, 16, 24), generator=g), lat, torch.randn((1, 4, 16, 24), generator=g)])
    ctx3 = torch.cat([torch.randn((1, arch.ctx, arch.cross_attention_dim), generator=g), ctx, ctx])
    got3 = k.unet(lat3, 500.0, ctx3)
    # BIT-identical: every GEMM's K split is chosen from the per-sample shape, never from the launch size (round 4;
    # round 3 asserted 2e-2 here, and 4e-2 for the VAE below)
    assert torch.equal(got3[1:2], got)
    # VAE: 6 images of 64 x 64 latents -> chunks of 5 + 1; each equals its own single-image decode
    z = torch.randn((6, 4, 64, 64), generator=g)
    imgs = k.vae_decode(z)
    assert imgs.shape == (6, 3, 512, 512) and torch.isfinite(imgs).all()
    for i in (0, 4, 5):
        one = k.vae_decode(z[i:i + 1])
        assert torch.equal(imgs[i:i + 1], one), (i, (imgs[i:i + 1] - one).abs().max().item())
    with pytest.raises(pkg.TVCError):
        k.unet(torch.zeros((1, 4, 12, 12)), 1.0, ctx)          # H, W must be multiples of 8 (three stride-2 levels)


def test_stable_diffusion_model_full_pipeline_toy_geometry_vs_oracle(pkg, tmp_path):
    """The WHOLE reference-generation path at a toy geometry the CPU oracle finishes in seconds: prompts -> tokens (EOT-padded
    as the SD tokenizer pads) -> CLIP text states (tvc_encode_text_hidden) -> PNDM loop with classifier-free guidance ->
    VAE -> pixels, HIP vs clip_oracle.text_hidden + sd_oracle.generate on the same weights; the weights travel through
    diffusers-style safetensors files (``SDModelConfig(unet_weights=, vae_weights=)``)."""
    from safetensors.torch import save_file
    from oracle import clip_oracle
    arch = pkg.SDArch(block_out_channels=(64, 128), down_block_attn=(True, False), layers_per_block=1, heads=8,
                      cross_attention_dim=128, vae_block_out_channels=(64, 128), vae_layers_per_block=1, sample_size=16)
    uw, vw = pkg.make_sd_weights(arch, seed=3)
    save_file({k: v.contiguous() for k, v in uw.items()}, str(tmp_path / "unet.safetensors"))
    save_file({k: v.contiguous() for k, v in vw.items()}, str(tmp_path / "vae.safetensors"))
    carch = pkg.get_arch("ViT-T/16-test")                     # text width 128 = the toy UNet's cross_attention_dim
    cw = pkg.synth.make_clip_weights(carch, seed=0)
    clip = pkg.CLIPModel(pkg.CLIPConfig(model_name=carch.name), weights=cw)
    sd = pkg.StableDiffusionModel(pkg.SDModelConfig(unet_weights=str(tmp_path / "unet.safetensors"),
                                                    vae_weights=str(tmp_path / "vae.safetensors")), clip_model=clip, arch=arch)
    assert sd.text_engine is clip.engine
    prompts, seeds, steps, guidance = ["a red cube on a table", "two birds"], [11, 12], 6, 5.0
    imgs = sd.generate_batch(prompts, seeds, steps, guidance, 32, 32, negative_prompts=["blurry", "blurry"]).cpu()
    assert imgs.shape == (2, 3, 32, 32)
    # ---- oracle
    tok = sd.tokenize(prompts)
    assert (tok[:, -1] == pkg.synth.EOT).all() and (tok != 0).all()          # EOT padding, no zero pads
    with torch.no_grad():
        cond = clip_oracle.text_hidden(cw[1], tok.long(), carch.text.heads)
        unc = clip_oracle.text_hidden(cw[1], sd.tokenize(["blurry", "blurry"]).long(), carch.text.heads)
        lat0 = sd.initial_latents(seeds, 4, 16, 16)
        ref = sd_oracle.generate(uw, vw, arch, cond, unc, lat0, steps, guidance)
    d = (imgs - ref).abs()
    print(f"[measured] full SD pipeline, toy geometry, {steps} steps: pixels in [0, 1] max |d| {d.max().item():.2e} mean |d| {d.mean().item():.2e}")
    assert d.max().item() < 2.5e-2 and d.mean().item() < 4e-3          # measured 1.05e-2 / 1.8e-3
    # the PIL-returning forms of the reference's wrapper
    pil = sd.generate_image(prompt=prompts[0], num_images=2, seed=11, num_inference_steps=steps, guidance_scale=guidance, height=32,
                            width=32, negative_prompt="blurry")
    assert len(pil) == 2 and pil[0].size == (32, 32)
    a0 = torch.from_numpy(np.asarray(pil[0], dtype=np.float32) / 255.0).permute(2, 0, 1)
    assert (a0 - imgs[0]).abs().max().item() < 0.5 / 255 + 1e-6                # image 0 of the call = seed 11 = the batch's first image
    out = sd.generate(prompt=prompts[1], negative_prompt="blurry", height=32, width=32, guidance_scale=guidance,
                      num_inference_steps=steps, generator=torch.Generator().manual_seed(12))
    a1 = torch.from_numpy(np.asarray(out.images[0], dtype=np.float32) / 255.0).permute(2, 0, 1)
    assert (a1 - imgs[1]).abs().max().item() < 0.5 / 255 + 1e-6
    clip.engine.close()


# ---------------------------------------------------------------------------------------------------------------
# Stable Diffusion 2.x geometry (BASELINE configs[4] names "SD-2.1"; the reference lists it as supported,
# src/__init__.py:110-113, while every default it holds is SD-1.5): per-level head counts 5 / 10 / 20 / 20 at head dim 64,
# linear proj_in / proj_out, cross-attention onto 1024-wide text states, optional v-prediction, an erf-GELU text tower.
# Oracle: oracle/sd_oracle.py on the same weights -- PARITY UNPINNED, as for SD-1.5 (see its header).
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def sd2(pkg):
    arch = pkg.SDArch.sd21_base()
    uw, _ = pkg.make_sd_weights(arch, seed=1, which="unet", device="cuda")
    eng = pkg.TVCEngine()
    k = pkg.SDKernels(eng, arch, uw, None)
    yield arch, uw, k
    eng.close()


def test_sd21_geometry_and_parameter_count(pkg):
    from importlib import import_module
    import math
    sa = import_module(pkg.__name__ + ".sd_arch")
    a = pkg.SDArch.sd21_base()
    assert [a.heads_at(i) for i in range(4)] == [5, 10, 20, 20] and all(c // a.heads_at(i) == 64 for i, c in enumerate(a.block_out_channels))
    # the published size of stable-diffusion-2-1's UNet (865.9 M parameters) pins names and shapes, as 859.5 M does for SD-1.5
    assert sum(math.prod(s) for _, s in sa.unet_param_shapes(a)) == 865_910_724
    assert sum(math.prod(s) for _, s in sa.unet_param_shapes(pkg.SDArch.sd15())) == 859_520_964
    assert dict(sa.unet_param_shapes(a))["down_blocks.0.attentions.0.proj_in.weight"] == (320, 320)           # nn.Linear
    assert pkg.get_arch("SD2-text").text.act == "gelu" and pkg.get_arch("SD2-text").text.width == a.cross_attention_dim


@pytest.mark.parametrize("prefix,c,hw,level", [("down_blocks.0.attentions.0.", 320, 32, 0), ("down_blocks.1.attentions.1.", 640, 16, 1),
                                               ("up_blocks.1.attentions.2.", 1280, 8, 2), ("mid_block.attentions.0.", 1280, 8, 3)])
def test_sd21_transformer_block_vs_oracle(pkg, sd2, prefix, c, hw, level):
    arch, uw, k = sd2
    g = torch.Generator().manual_seed(c + level)
    x = torch.randn((2, c, hw, hw), generator=g)
    ctx = torch.randn((2, arch.ctx, arch.cross_attention_dim), generator=g)
    got = k.block(1, prefix, x, c, ctx=ctx)
    with torch.no_grad():
        ref = sd_oracle.transformer(uw, prefix, x, ctx, arch.heads_at(level), arch.norm_groups)
    r2, rm = rel(got, ref)
    print(f"[measured] SD-2.1 transformer {prefix} ({arch.heads_at(level)} heads of 64, linear projections, ctx 1024): rel L2 {r2:.2e} max|d|/std {rm:.2e}")
    assert r2 < 7e-3 and rm < 6e-2           # the SD-1.5 blocks' bound (measured there 3.4e-3 / 2.0e-2 .. 2.6e-2)


def test_sd21_unet_forward_vs_oracle(pkg, sd2):
    arch, uw, k = sd2
    g = torch.Generator().manual_seed(24)
    lat = torch.randn((2, 4, 16, 16), generator=g)
    ctx = torch.randn((2, arch.ctx, arch.cross_attention_dim), generator=g)
    got = k.unet(lat, 701.0, ctx)
    with torch.no_grad():
        ref = sd_oracle.unet_forward(uw, arch, lat, 701, ctx)
    r2, rm = rel(got, ref)
    print(f"[measured] SD-2.1 UNet forward (16 x 16 latents, t = 701): rel L2 {r2:.2e} max|d|/std {rm:.2e}")
    assert torch.isfinite(got).all() and r2 < 2.5e-2 and rm < 1e-1


def test_v_prediction_sampling_loop_vs_oracle(pkg):
    """prediction_type "v_prediction" (stable-diffusion-2-1 at 768 px): PNDMScheduler._get_prev_sample turns the combined
    model output into sqrt(a_t) v + sqrt(b_t) x before the update -- folded into the two coefficients of the update kernel
    (tvc_sd.cpp).  Toy geometry with per-level head counts, 6 steps, guidance 6, against the oracle's scheduler."""
    arch = pkg.SDArch(block_out_channels=(64, 128), down_block_attn=(True, True), layers_per_block=1, heads=8, heads_per_block=(1, 2),
                      linear_projection=True, prediction_type="v_prediction", cross_attention_dim=128,
                      vae_block_out_channels=(64, 128), vae_layers_per_block=1, sample_size=16)
    uw, vw = pkg.make_sd_weights(arch, seed=4)
    eng = pkg.TVCEngine()
    k = pkg.SDKernels(eng, arch, uw, vw)
    g = torch.Generator().manual_seed(26)
    n, steps, guidance = 2, 6, 6.0
    cond, uncond = (torch.randn((n, arch.ctx, arch.cross_attention_dim), generator=g) for _ in range(2))
    lat0 = torch.randn((n, 4, 16, 16), generator=g)
    lat, img = k.generate(cond, uncond, lat0, steps, guidance, decode=True)
    with torch.no_grad():
        ref = sd_oracle.generate(uw, vw, arch, cond, uncond, lat0, steps, guidance, return_latents=True)
        arch_eps = pkg.SDArch(**{**arch.__dict__, "prediction_type": "epsilon"})
        ref_eps = sd_oracle.generate(uw, vw, arch_eps, cond, uncond, lat0, steps, guidance, return_latents=True)
    r2, rm = rel(lat, ref)
    print(f"[measured] v-prediction sampling loop, {steps} PLMS steps: final latents rel L2 {r2:.2e} max|d|/std {rm:.2e} "
          f"(an epsilon-prediction loop on the same weights differs by {rel(ref_eps, ref)[0]:.2f})")
    assert torch.isfinite(lat).all() and r2 < 8e-2 and rel(ref_eps, ref)[0] > 0.5          # measured 4.1e-2 (the v update amplifies bf16 noise on random weights)
    assert img.shape == (n, 3, 32, 32)
    eng.close()


@pytest.mark.parametrize("precision", ["bf16", "split", "fp32"])
def test_erf_gelu_text_tower_vs_oracle(pkg, precision):
    """The SD-2.x text encoder is OpenCLIP ViT-H/14's text tower: exact erf GELU instead of QuickGELU (TVC_ACT_GELU: a
    store-only FC1 followed by a row kernel).  Toy geometry "ViT-T/16-gelu-test", hidden states (what conditions the UNet)
    and pooled embeddings, in all three tower precisions."""
    from oracle import clip_oracle
    arch = pkg.get_arch("ViT-T/16-gelu-test")
    vw, tw = pkg.synth.make_clip_weights(arch, seed=0)
    eng = pkg.TVCEngine(arch, vw, tw, precision=precision)
    toks = pkg.synth.make_tokens(3, 2, arch.ctx, seed=2).view(-1, arch.ctx)
    imgs = pkg.synth.make_images(3, arch.image_size, seed=1)
    hid = eng.encode_text_hidden(toks.cuda()).cpu()
    ft = eng.encode_text(toks.cuda(), group=3).cpu()
    fi = eng.encode_image(imgs.cuda()).cpu()
    with torch.no_grad():
        rh = clip_oracle.text_hidden(tw, toks.long(), arch.text.heads, act="gelu")
        rt = clip_oracle.text_forward(tw, toks.long(), arch.text.heads, act="gelu")
        ri = clip_oracle.vision_forward(vw, imgs, arch.vision.heads, arch.patch, act="gelu")
        rq = clip_oracle.text_forward(tw, toks.long(), arch.text.heads, act="quick_gelu")
    dh, dt, di = (hid - rh).abs().max().item(), (ft - rt).abs().max().item(), (fi - ri).abs().max().item()
    print(f"[measured] erf-GELU towers, {precision}: hidden max|d| {dh:.2e}  text max|d| {dt:.2e}  image max|d| {di:.2e}  "
          f"(QuickGELU on the same weights differs by {(rq - rt).abs().max().item():.2e})")
    bound = 3e-3 if precision == "bf16" else 5e-5
    assert dt < bound and di < bound and dh < (6e-2 if precision == "bf16" else 2e-4)
    if precision != "bf16":
        assert (rq - rt).abs().max().item() > 10 * bound        # the activation really matters at this tolerance (2.8e-3)
    if precision == "bf16":
        with pytest.raises(pkg.TVCError):                       # no backward pass for erf-GELU towers
            eng.encode_image_grad(imgs.cuda(), True)
    eng.close()
