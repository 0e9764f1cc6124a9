"""Geometry and parameter inventory of the latent-diffusion model behind the SD reference generator
(BASELINE configs[4]).

The reference reaches Stable Diffusion through ``StableDiffusionModel.generate_image`` (``src/sd_ref.py:389-399``) /
``sd_model.generate`` (``experiments/defenses/generative_ref.py:139-147``) -- a wrapper around
``diffusers.StableDiffusionPipeline`` whose source and weights are absent from the snapshot.  What the snapshot DOES
hold is the geometry: ``cache/sd/models--runwayml--stable-diffusion-v1-5/snapshots/*/{unet,vae,scheduler}/*.json``
(``UNet2DConditionModel``: block_out_channels [320, 640, 1280, 1280], attention_head_dim 8 -- used by that diffusers
version as the NUMBER of heads, so head dims 40 / 80 / 160 --, cross_attention_dim 768, sample_size 64;
``AutoencoderKL``: [128, 256, 512, 512], latent_channels 4; ``PNDMScheduler``: scaled_linear 0.00085..0.012,
skip_prk_steps, steps_offset 1).  ``SDArch()`` is that geometry; parameter names follow the diffusers state-dict
layout of that version so that a real checkpoint's ``unet`` / ``vae`` safetensors load without renaming.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch


@dataclass(frozen=True)
class SDArch:
    # unet/config.json
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_block_attn: Tuple[bool, ...] = (True, True, True, False)     # CrossAttnDownBlock2D x3, DownBlock2D
    layers_per_block: int = 2
    heads: int = 8                       # "attention_head_dim": 8 (the number of heads in this diffusers version)
    # SD 2.x: "attention_head_dim": [5, 10, 20, 20] -- the NUMBER of heads per resolution level (head dim 64 everywhere);
    # None = `heads` at every level (SD 1.x)
    heads_per_block: Optional[Tuple[int, ...]] = None
    linear_projection: bool = False      # "use_linear_projection": Transformer2DModel.proj_in / proj_out are nn.Linear [C, C]
    prediction_type: str = "epsilon"     # scheduler_config.json: "epsilon" (SD 1.x, 2.1-base) or "v_prediction" (2.1 at 768 px)
    text_arch: str = "ViT-L/14"          # text_encoder/config.json: CLIP ViT-L/14 (768) or "SD2-text" (OpenCLIP ViT-H/14, 1024)
    name: str = "runwayml/stable-diffusion-v1-5"
    cross_attention_dim: int = 768
    norm_groups: int = 32
    norm_eps: float = 1e-5
    sample_size: int = 64
    # vae/config.json
    vae_block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    vae_layers_per_block: int = 2
    latent_channels: int = 4
    vae_scaling: float = 0.18215
    # scheduler/scheduler_config.json
    beta_start: float = 0.00085
    beta_end: float = 0.012
    num_train_timesteps: int = 1000
    steps_offset: int = 1
    ctx: int = 77

    @property
    def time_dim(self) -> int:
        return self.block_out_channels[0] * 4

    def heads_at(self, level: int) -> int:
        return self.heads_per_block[level] if self.heads_per_block is not None else self.heads

    def head_dim(self, channels: int, level: int = 0) -> int:
        return channels // self.heads_at(level)

    @staticmethod
    def sd15() -> "SDArch":
        """runwayml/stable-diffusion-v1-5: the geometry of the config.json files the reference holds (its default,
        src/sd_ref.py:221)."""
        return SDArch()

    @staticmethod
    def sd21_base() -> "SDArch":
        """stabilityai/stable-diffusion-2-1-base (listed as supported by the reference, src/__init__.py:110-113; the model
        BASELINE configs[4] names): per-level head counts 5 / 10 / 20 / 20 at head dim 64, linear proj_in / proj_out,
        cross-attention onto the 1024-wide OpenCLIP ViT-H/14 text states, epsilon prediction at 512 px.  (The 768-px
        "stable-diffusion-2-1" differs by prediction_type="v_prediction" and sample_size 96.)"""
        return SDArch(heads_per_block=(5, 10, 20, 20), linear_projection=True, cross_attention_dim=1024,
                      text_arch="SD2-text", name="stabilityai/stable-diffusion-2-1-base")


def _resnet_shapes(p: str, cin: int, cout: int, temb: int) -> List[Tuple[str, Tuple[int, ...]]]:
    s = [(p + "norm1.weight", (cin,)), (p + "norm1.bias", (cin,)),
         (p + "conv1.weight", (cout, cin, 3, 3)), (p + "conv1.bias", (cout,))]
    if temb:
        s += [(p + "time_emb_proj.weight", (cout, temb)), (p + "time_emb_proj.bias", (cout,))]
    s += [(p + "norm2.weight", (cout,)), (p + "norm2.bias", (cout,)),
          (p + "conv2.weight", (cout, cout, 3, 3)), (p + "conv2.bias", (cout,))]
    if cin != cout:
        s += [(p + "conv_shortcut.weight", (cout, cin, 1, 1)), (p + "conv_shortcut.bias", (cout,))]
    return s


def _transformer_shapes(p: str, c: int, ctx: int, linear: bool = False) -> List[Tuple[str, Tuple[int, ...]]]:
    t = p + "transformer_blocks.0."
    pj = (c, c) if linear else (c, c, 1, 1)           # use_linear_projection: nn.Linear instead of a 1 x 1 convolution
    s = [(p + "norm.weight", (c,)), (p + "norm.bias", (c,)),
         (p + "proj_in.weight", pj), (p + "proj_in.bias", (c,))]
    for i in (1, 2, 3):
        s += [(t + f"norm{i}.weight", (c,)), (t + f"norm{i}.bias", (c,))]
    for a, kv in (("attn1", c), ("attn2", ctx)):
        s += [(t + f"{a}.to_q.weight", (c, c)), (t + f"{a}.to_k.weight", (c, kv)), (t + f"{a}.to_v.weight", (c, kv)),
              (t + f"{a}.to_out.0.weight", (c, c)), (t + f"{a}.to_out.0.bias", (c,))]
    s += [(t + "ff.net.0.proj.weight", (8 * c, c)), (t + "ff.net.0.proj.bias", (8 * c,)),
          (t + "ff.net.2.weight", (c, 4 * c)), (t + "ff.net.2.bias", (c,)),
          (p + "proj_out.weight", pj), (p + "proj_out.bias", (c,))]
    return s


def unet_param_shapes(a: SDArch) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) of every UNet parameter, in module order (``UNet2DConditionModel.state_dict()`` names)."""
    ch, T = a.block_out_channels, a.time_dim
    s = [("time_embedding.linear_1.weight", (T, ch[0])), ("time_embedding.linear_1.bias", (T,)),
         ("time_embedding.linear_2.weight", (T, T)), ("time_embedding.linear_2.bias", (T,)),
         ("conv_in.weight", (ch[0], a.in_channels, 3, 3)), ("conv_in.bias", (ch[0],))]
    out = ch[0]
    for i, c in enumerate(ch):
        cin, out = out, c
        for j in range(a.layers_per_block):
            s += _resnet_shapes(f"down_blocks.{i}.resnets.{j}.", cin if j == 0 else c, c, T)
            if a.down_block_attn[i]:
                s += _transformer_shapes(f"down_blocks.{i}.attentions.{j}.", c, a.cross_attention_dim, a.linear_projection)
        if i != len(ch) - 1:
            s += [(f"down_blocks.{i}.downsamplers.0.conv.weight", (c, c, 3, 3)), (f"down_blocks.{i}.downsamplers.0.conv.bias", (c,))]
    m = ch[-1]
    s += _resnet_shapes("mid_block.resnets.0.", m, m, T) + _transformer_shapes("mid_block.attentions.0.", m, a.cross_attention_dim, a.linear_projection) + \
        _resnet_shapes("mid_block.resnets.1.", m, m, T)
    rev = list(reversed(ch))
    attn_rev = list(reversed(a.down_block_attn))
    prev = rev[0]
    for i, c in enumerate(rev):
        skip_in = rev[min(i + 1, len(ch) - 1)]
        for j in range(a.layers_per_block + 1):
            res_skip = skip_in if j == a.layers_per_block else c
            res_in = prev if j == 0 else c
            s += _resnet_shapes(f"up_blocks.{i}.resnets.{j}.", res_in + res_skip, c, T)
            if attn_rev[i]:
                s += _transformer_shapes(f"up_blocks.{i}.attentions.{j}.", c, a.cross_attention_dim, a.linear_projection)
        if i != len(ch) - 1:
            s += [(f"up_blocks.{i}.upsamplers.0.conv.weight", (c, c, 3, 3)), (f"up_blocks.{i}.upsamplers.0.conv.bias", (c,))]
        prev = c
    s += [("conv_norm_out.weight", (ch[0],)), ("conv_norm_out.bias", (ch[0],)),
          ("conv_out.weight", (a.out_channels, ch[0], 3, 3)), ("conv_out.bias", (a.out_channels,))]
    return s


def vae_decoder_param_shapes(a: SDArch) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) of what ``AutoencoderKL.decode`` reads: post_quant_conv + decoder (diffusers 0.6 names)."""
    ch = a.vae_block_out_channels
    L = a.latent_channels
    top = ch[-1]
    s = [("post_quant_conv.weight", (L, L, 1, 1)), ("post_quant_conv.bias", (L,)),
         ("decoder.conv_in.weight", (top, L, 3, 3)), ("decoder.conv_in.bias", (top,))]
    s += _resnet_shapes("decoder.mid_block.resnets.0.", top, top, 0)
    p = "decoder.mid_block.attentions.0."
    s += [(p + "group_norm.weight", (top,)), (p + "group_norm.bias", (top,))]
    for n in ("query", "key", "value", "proj_attn"):
        s += [(p + f"{n}.weight", (top, top)), (p + f"{n}.bias", (top,))]
    s += _resnet_shapes("decoder.mid_block.resnets.1.", top, top, 0)
    prev = top
    for i, c in enumerate(reversed(ch)):
        for j in range(a.vae_layers_per_block + 1):
            s += _resnet_shapes(f"decoder.up_blocks.{i}.resnets.{j}.", prev if j == 0 else c, c, 0)
        if i != len(ch) - 1:
            s += [(f"decoder.up_blocks.{i}.upsamplers.0.conv.weight", (c, c, 3, 3)), (f"decoder.up_blocks.{i}.upsamplers.0.conv.bias", (c,))]
        prev = c
    s += [("decoder.conv_norm_out.weight", (ch[0],)), ("decoder.conv_norm_out.bias", (ch[0],)),
          ("decoder.conv_out.weight", (3, ch[0], 3, 3)), ("decoder.conv_out.bias", (3,))]
    return s


def make_sd_weights(a: SDArch, seed: int = 0, which: str = "both", device: str = "cpu") -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """Seeded random-init (unet, vae-decoder) weights: fan-in-scaled matrices, norm gains 1 + 0.1 N, biases 0.02 N.
    There is no network for the real checkpoint; parity of the kernels does not depend on the values.  ``device``: where
    the generator runs (the tensors come back on the CPU either way); "cuda" draws the 0.9 G parameters of the full
    geometry in a second instead of half a minute -- other numbers than the CPU generator's, equally deterministic."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev).manual_seed(seed)

    def init(shapes):
        w = {}
        for name, shp in shapes:
            if name.endswith("bias"):
                t = torch.randn(shp, generator=gen, device=dev) * 0.02
            elif len(shp) == 1:
                t = 1.0 + 0.1 * torch.randn(shp, generator=gen, device=dev)
            else:
                fan_in = math.prod(shp[1:])
                t = torch.randn(shp, generator=gen, device=dev) * (0.7 / math.sqrt(fan_in))
            w[name] = t.cpu()
        return w

    unet = init(unet_param_shapes(a)) if which in ("both", "unet") else {}
    vae = init(vae_decoder_param_shapes(a)) if which in ("both", "vae") else {}
    return unet, vae
