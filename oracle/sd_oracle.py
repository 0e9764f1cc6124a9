"""CPU restatement (PyTorch fp32) of the latent-diffusion model behind the SD reference generator
(BASELINE configs[4]).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

**Parity unpinned.**  The reference reaches Stable Diffusion through ``StableDiffusionModel.generate_image``
(``/root/reference/src/sd_ref.py:389-399``) and ``sd_model.generate`` (``/root/reference/experiments/defenses/
generative_ref.py:139-147``): a wrapper (absent from the snapshot) around the third-party ``diffusers`` package
(``requirements.txt``: ``diffusers>=0.21.0``, not vendored, not importable here), with weights that are missing too
(``.MISSING_LARGE_BLOBS``).  The reference holds no vectors at this boundary.  What it does hold is the geometry --
``/root/reference/cache/sd/models--runwayml--stable-diffusion-v1-5/snapshots/*/{unet,vae,scheduler}/*.json`` -- and
this file restates the published algorithms of the classes those files name:

* ``UNet2DConditionModel`` (sinusoidal timestep embedding with flip_sin_to_cos / freq_shift 0, two-layer SiLU time
  MLP, ResnetBlock2D with GroupNorm(32) + SiLU + time projection, Transformer2DModel with conv 1x1 projections,
  self-attention, cross-attention onto the 77 x 768 text states, GEGLU feed-forward, stride-2 conv downsample,
  nearest-2x + conv upsample, skip concatenation);
* ``AutoencoderKL.decode`` (post_quant_conv, mid block with one single-head attention, four up blocks, eps 1e-6);
* ``PNDMScheduler`` with ``skip_prk_steps`` (the PLMS multistep rule) and the classifier-free-guidance loop of
  ``StableDiffusionPipeline.__call__``.

Weights: dict name -> fp32 tensor with the diffusers state-dict names (the product's ``sd_arch.make_sd_weights``
produces them; the oracle walks the architecture itself, so a naming / shape slip on either side fails loudly).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ building blocks
def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """``Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)``: [cos | sin] of t * 10000^(-i / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = t.float()[:, None] * freqs[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def resnet(w: Dict, p: str, x: torch.Tensor, temb: Optional[torch.Tensor], groups: int, eps: float) -> torch.Tensor:
    h = F.silu(F.group_norm(x, groups, w[p + "norm1.weight"], w[p + "norm1.bias"], eps))
    h = F.conv2d(h, w[p + "conv1.weight"], w[p + "conv1.bias"], padding=1)
    if temb is not None:
        h = h + F.linear(F.silu(temb), w[p + "time_emb_proj.weight"], w[p + "time_emb_proj.bias"])[:, :, None, None]
    h = F.silu(F.group_norm(h, groups, w[p + "norm2.weight"], w[p + "norm2.bias"], eps))
    h = F.conv2d(h, w[p + "conv2.weight"], w[p + "conv2.bias"], padding=1)
    if p + "conv_shortcut.weight" in w:
        x = F.conv2d(x, w[p + "conv_shortcut.weight"], w[p + "conv_shortcut.bias"])
    return x + h


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int) -> torch.Tensor:
    """q [B, Tq, C], k / v [B, Tk, C] -> [B, Tq, C]; scale head_dim ** -0.5."""
    B, Tq, C = q.shape
    dh = C // heads
    sp = lambda t: t.view(B, -1, heads, dh).transpose(1, 2)
    s = (sp(q) @ sp(k).transpose(-1, -2)) * dh ** -0.5
    return (s.softmax(-1) @ sp(v)).transpose(1, 2).reshape(B, Tq, C)


def transformer(w: Dict, p: str, x: torch.Tensor, ctx: torch.Tensor, heads: int, groups: int) -> torch.Tensor:
    """``Transformer2DModel`` with one ``BasicTransformerBlock``; ``proj_in`` / ``proj_out`` are 1 x 1 convolutions
    (SD 1.x) or, with ``use_linear_projection`` (SD 2.x: 2-D weights), ``nn.Linear`` applied after / before the reshape."""
    B, C, H, W = x.shape
    t = p + "transformer_blocks.0."
    linear = w[p + "proj_in.weight"].dim() == 2
    h = F.group_norm(x, groups, w[p + "norm.weight"], w[p + "norm.bias"], 1e-6)
    if linear:
        h = F.linear(h.permute(0, 2, 3, 1).reshape(B, H * W, C), w[p + "proj_in.weight"], w[p + "proj_in.bias"])
    else:
        h = F.conv2d(h, w[p + "proj_in.weight"], w[p + "proj_in.bias"])
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    n = F.layer_norm(h, (C,), w[t + "norm1.weight"], w[t + "norm1.bias"], 1e-5)
    a = attention(F.linear(n, w[t + "attn1.to_q.weight"]), F.linear(n, w[t + "attn1.to_k.weight"]),
                  F.linear(n, w[t + "attn1.to_v.weight"]), heads)
    h = h + F.linear(a, w[t + "attn1.to_out.0.weight"], w[t + "attn1.to_out.0.bias"])
    n = F.layer_norm(h, (C,), w[t + "norm2.weight"], w[t + "norm2.bias"], 1e-5)
    a = attention(F.linear(n, w[t + "attn2.to_q.weight"]), F.linear(ctx, w[t + "attn2.to_k.weight"]),
                  F.linear(ctx, w[t + "attn2.to_v.weight"]), heads)
    h = h + F.linear(a, w[t + "attn2.to_out.0.weight"], w[t + "attn2.to_out.0.bias"])
    n = F.layer_norm(h, (C,), w[t + "norm3.weight"], w[t + "norm3.bias"], 1e-5)
    g = F.linear(n, w[t + "ff.net.0.proj.weight"], w[t + "ff.net.0.proj.bias"])
    val, gate = g.chunk(2, dim=-1)
    h = h + F.linear(val * F.gelu(gate), w[t + "ff.net.2.weight"], w[t + "ff.net.2.bias"])
    if linear:
        h = F.linear(h, w[p + "proj_out.weight"], w[p + "proj_out.bias"])
        return x + h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return x + F.conv2d(h, w[p + "proj_out.weight"], w[p + "proj_out.bias"])


# ------------------------------------------------------------------ UNet2DConditionModel
def unet_forward(w: Dict, arch, sample: torch.Tensor, t, ctx: torch.Tensor) -> torch.Tensor:
    """sample [B, 4, H, W], t scalar or [B], ctx [B, 77, 768] -> predicted noise [B, 4, H, W]."""
    B = sample.shape[0]
    ch, G, eps = arch.block_out_channels, arch.norm_groups, arch.norm_eps
    heads_at = lambda lvl: arch.heads_per_block[lvl] if getattr(arch, "heads_per_block", None) is not None else arch.heads
    tt = torch.as_tensor(t, dtype=torch.float32).reshape(-1).expand(B) if not torch.is_tensor(t) or t.dim() == 0 else t
    temb = timestep_embedding(tt, ch[0])
    temb = F.linear(F.silu(F.linear(temb, w["time_embedding.linear_1.weight"], w["time_embedding.linear_1.bias"])),
                    w["time_embedding.linear_2.weight"], w["time_embedding.linear_2.bias"])
    x = F.conv2d(sample, w["conv_in.weight"], w["conv_in.bias"], padding=1)
    skips = [x]
    for i in range(len(ch)):
        for j in range(arch.layers_per_block):
            x = resnet(w, f"down_blocks.{i}.resnets.{j}.", x, temb, G, eps)
            if arch.down_block_attn[i]:
                x = transformer(w, f"down_blocks.{i}.attentions.{j}.", x, ctx, heads_at(i), G)
            skips.append(x)
        if i != len(ch) - 1:
            x = F.conv2d(x, w[f"down_blocks.{i}.downsamplers.0.conv.weight"], w[f"down_blocks.{i}.downsamplers.0.conv.bias"],
                         stride=2, padding=1)
            skips.append(x)
    x = resnet(w, "mid_block.resnets.0.", x, temb, G, eps)
    x = transformer(w, "mid_block.attentions.0.", x, ctx, heads_at(len(ch) - 1), G)
    x = resnet(w, "mid_block.resnets.1.", x, temb, G, eps)
    attn_rev = list(reversed(arch.down_block_attn))
    for i in range(len(ch)):
        for j in range(arch.layers_per_block + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet(w, f"up_blocks.{i}.resnets.{j}.", x, temb, G, eps)
            if attn_rev[i]:
                x = transformer(w, f"up_blocks.{i}.attentions.{j}.", x, ctx, heads_at(len(ch) - 1 - i), G)
        if i != len(ch) - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, w[f"up_blocks.{i}.upsamplers.0.conv.weight"], w[f"up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    x = F.silu(F.group_norm(x, G, w["conv_norm_out.weight"], w["conv_norm_out.bias"], eps))
    return F.conv2d(x, w["conv_out.weight"], w["conv_out.bias"], padding=1)


# ------------------------------------------------------------------ AutoencoderKL.decode
def vae_attention(w: Dict, p: str, x: torch.Tensor, groups: int) -> torch.Tensor:
    """``AttentionBlock`` (one head over all channels): q / k each scaled by C ** -0.25."""
    B, C, H, W = x.shape
    h = F.group_norm(x, groups, w[p + "group_norm.weight"], w[p + "group_norm.bias"], 1e-6)
    h = h.view(B, C, H * W).transpose(1, 2)
    q = F.linear(h, w[p + "query.weight"], w[p + "query.bias"])
    k = F.linear(h, w[p + "key.weight"], w[p + "key.bias"])
    v = F.linear(h, w[p + "value.weight"], w[p + "value.bias"])
    scale = 1.0 / math.sqrt(math.sqrt(C))
    s = ((q * scale) @ (k * scale).transpose(-1, -2)).float().softmax(-1)
    h = F.linear(s @ v, w[p + "proj_attn.weight"], w[p + "proj_attn.bias"])
    return x + h.transpose(1, 2).reshape(B, C, H, W)


def vae_decode(w: Dict, arch, z: torch.Tensor) -> torch.Tensor:
    """z [B, 4, h, w] (already divided by the scaling factor) -> image [B, 3, 8h, 8w] in about [-1, 1]."""
    G, eps, ch = arch.norm_groups, 1e-6, arch.vae_block_out_channels
    x = F.conv2d(z, w["post_quant_conv.weight"], w["post_quant_conv.bias"])
    x = F.conv2d(x, w["decoder.conv_in.weight"], w["decoder.conv_in.bias"], padding=1)
    x = resnet(w, "decoder.mid_block.resnets.0.", x, None, G, eps)
    x = vae_attention(w, "decoder.mid_block.attentions.0.", x, G)
    x = resnet(w, "decoder.mid_block.resnets.1.", x, None, G, eps)
    for i in range(len(ch)):
        for j in range(arch.vae_layers_per_block + 1):
            x = resnet(w, f"decoder.up_blocks.{i}.resnets.{j}.", x, None, G, eps)
        if i != len(ch) - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, w[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"], w[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    x = F.silu(F.group_norm(x, G, w["decoder.conv_norm_out.weight"], w["decoder.conv_norm_out.bias"], eps))
    return F.conv2d(x, w["decoder.conv_out.weight"], w["decoder.conv_out.bias"], padding=1)


# ------------------------------------------------------------------ PNDMScheduler (skip_prk_steps: PLMS)
class PNDMOracle:
    def __init__(self, arch):
        betas = torch.linspace(arch.beta_start ** 0.5, arch.beta_end ** 0.5, arch.num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]          # set_alpha_to_one = False
        self.T = arch.num_train_timesteps
        self.offset = arch.steps_offset
        self.v_prediction = getattr(arch, "prediction_type", "epsilon") == "v_prediction"

    def set_timesteps(self, n: int) -> List[int]:
        self.n = n
        ratio = self.T // n
        ts = [i * ratio + self.offset for i in range(n)]
        plms = ts[:-1] + ts[-2:-1] + ts[-1:]                        # the second-to-last value is visited twice
        self.timesteps = plms[::-1]
        self.ets: List[torch.Tensor] = []
        self.counter = 0
        self.cur_sample = None
        return self.timesteps

    def _prev_sample(self, sample, t: int, prev_t: int, eps):
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t, b_prev = 1 - a_t, 1 - a_prev
        if self.v_prediction:                                   # PNDMScheduler._get_prev_sample, prediction_type "v_prediction"
            eps = (a_t ** 0.5) * eps + (b_t ** 0.5) * sample
        sample_coeff = (a_prev / a_t) ** 0.5
        denom = a_t * b_prev ** 0.5 + (a_t * b_t * a_prev) ** 0.5
        return sample_coeff * sample - (a_prev - a_t) * eps / denom

    def step(self, eps: torch.Tensor, t: int, sample: torch.Tensor) -> torch.Tensor:
        prev_t = t - self.T // self.n
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(eps)
        else:
            prev_t, t = t, t + self.T // self.n
        if len(self.ets) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            eps = (eps + self.ets[-1]) / 2
            sample, self.cur_sample = self.cur_sample, None
        elif len(self.ets) == 2:
            eps = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            eps = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            eps = (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4]) / 24
        self.counter += 1
        return self._prev_sample(sample, t, prev_t, eps)


def generate(unet_w: Dict, vae_w: Dict, arch, cond: torch.Tensor, uncond: torch.Tensor, latents: torch.Tensor,
             steps: int, guidance: float, return_latents: bool = False):
    """``StableDiffusionPipeline.__call__`` after tokenisation / text encoding: cond / uncond [B, 77, 768] text states,
    latents [B, 4, h, w] initial noise (init_noise_sigma = 1 for PNDM) -> images [B, 3, 8h, 8w] in [0, 1]."""
    sch = PNDMOracle(arch)
    for t in sch.set_timesteps(steps):
        x2 = torch.cat([latents, latents])
        e = unet_forward(unet_w, arch, x2, t, torch.cat([uncond, cond]))
        eu, ec = e.chunk(2)
        latents = sch.step(eu + guidance * (ec - eu), t, latents)
    if return_latents:
        return latents
    img = vae_decode(vae_w, arch, latents / arch.vae_scaling)
    return (img / 2 + 0.5).clamp(0, 1)
