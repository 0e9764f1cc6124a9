"""``StableDiffusionModel``: the latent-diffusion model behind the SD reference generator, on the HIP kernels of
``include/tvc.h`` ("latent-diffusion reference generator"; SURVEY.md section 8f rank 1, BASELINE configs[4]).

Mirror of the reference's (absent) ``src/models/sd_model.py`` wrapper, reconstructed from its call sites:
``generate_image(prompt=, num_images=, seed=, num_inference_steps=, guidance_scale=, height=, width=) -> list of PIL
images`` (``src/sd_ref.py:389-399``) and ``generate(prompt=, negative_prompt=, height=, width=, guidance_scale=,
num_inference_steps=, generator=) -> object with .images`` (``experiments/defenses/generative_ref.py:139-147``).
Everything between the token ids and the pixels runs on the GPU: CLIP ViT-L/14 text states
(``tvc_encode_text_hidden``), the UNet sampling loop under the PNDM (PLMS) scheduler with classifier-free guidance
and the VAE decoder (``tvc_sd_generate``).  ``generate_batch`` is the batched form the reference does not have: all
prompts x seeds of a batch share every UNet launch.

Weights: a pair of diffusers safetensors files (``SDModelConfig(unet_weights=, vae_weights=)``), tensors handed in
(``weights=``), or -- ONLY with the explicit opt-in ``SDModelConfig(random_init=True)`` -- the seeded random init of
``sd_arch.make_sd_weights`` (there is no network for a checkpoint; benchmarks and parity tests run on it).  Without
weights and without the opt-in the constructor raises, as the reference's pipeline load fails without a checkpoint
(``src/sd_ref.py:291-317`` then goes on without a model and reports an error per call).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence, Union

import logging

import numpy as np
import torch

from . import _lib, synth
from .arch import get_arch
from .engine import TVCEngine, _ptr, _stream
from .sd_arch import SDArch, make_sd_weights, unet_param_shapes, vae_decoder_param_shapes

logger = logging.getLogger(__name__)


def _conv3x3_rows(w: torch.Tensor, pad_to: int = 0) -> torch.Tensor:
    """[Co, Ci, 3, 3] -> [Co, 9 * Ci], column (ky * 3 + kx) * Ci + ci (the row layout of ``sd_im2col3x3``)."""
    co, ci = w.shape[:2]
    r = w.permute(0, 2, 3, 1).reshape(co, 9 * ci)
    if pad_to and r.shape[1] < pad_to:
        r = torch.cat([r, r.new_zeros((co, pad_to - r.shape[1]))], 1)
    return r


def prepare_sd_tensors(unet_w: Optional[Dict[str, torch.Tensor]], vae_w: Optional[Dict[str, torch.Tensor]],
                       device: torch.device) -> Dict[str, torch.Tensor]:
    """diffusers state dicts -> the device tensors ``tvc_sd_load`` documents (include/tvc.h): GEMM operands bf16 with
    3x3 kernels flattened tap-major, q/k/v fused, everything else fp32."""
    out: Dict[str, torch.Tensor] = {}

    def put(name: str, t: torch.Tensor, dtype) -> None:
        t = t.detach().to(dtype=dtype)
        if dtype == torch.bfloat16 and t.dim() == 2 and t.shape[0] % 256:
            # tvc_sd_load's contract: GEMM weights are readable up to the next multiple of 256 rows (zero rows), so the
            # fast GEMM form stages whole 256-row tiles even where the width (320, 640, 4 ...) is not a multiple of 256
            t = torch.cat([t, t.new_zeros((256 - t.shape[0] % 256, t.shape[1]))])
        out[name] = t.to(device=device).contiguous()

    for w in (unet_w or {}), (vae_w or {}):
        fused = set()
        for name, t in w.items():
            if name in fused:
                continue
            if name.endswith(".bias") or t.dim() == 1:
                if name.endswith((".query.bias", ".key.bias", ".value.bias")):
                    p = name.rsplit(".", 2)[0] + "."
                    put(p + "to_qkv.bias", torch.cat([w[p + "query.bias"], w[p + "key.bias"], w[p + "value.bias"]]), torch.float32)
                    fused.update({p + "query.bias", p + "key.bias", p + "value.bias"})
                else:
                    put(name, t, torch.float32)
            elif name == "post_quant_conv.weight":
                put(name, t.reshape(t.shape[0], t.shape[1]), torch.float32)
            elif t.dim() == 4 and t.shape[-1] == 3:
                put(name, _conv3x3_rows(t.float(), 64 if t.shape[1] * 9 < 64 else 0), torch.bfloat16)
            elif t.dim() == 4:
                put(name, t.reshape(t.shape[0], t.shape[1]), torch.bfloat16)
            elif name.endswith(("attn1.to_q.weight", "attn1.to_k.weight", "attn1.to_v.weight")):
                p = name.rsplit(".", 2)[0] + "."
                put(p + "to_qkv.weight", torch.cat([w[p + "to_q.weight"], w[p + "to_k.weight"], w[p + "to_v.weight"]]), torch.bfloat16)
                fused.update({p + "to_q.weight", p + "to_k.weight", p + "to_v.weight"})
            elif name.endswith(("attn2.to_k.weight", "attn2.to_v.weight")):
                p = name.rsplit(".", 2)[0] + "."
                put(p + "to_kv.weight", torch.cat([w[p + "to_k.weight"], w[p + "to_v.weight"]]), torch.bfloat16)
                fused.update({p + "to_k.weight", p + "to_v.weight"})
            elif name.endswith((".query.weight", ".key.weight", ".value.weight")):
                p = name.rsplit(".", 2)[0] + "."
                put(p + "to_qkv.weight", torch.cat([w[p + "query.weight"], w[p + "key.weight"], w[p + "value.weight"]]), torch.bfloat16)
                fused.update({p + "query.weight", p + "key.weight", p + "value.weight"})
            else:
                put(name, t, torch.bfloat16)
    return out


def sd_desc(a: SDArch) -> "_lib.SDDesc":
    d = _lib.SDDesc()
    d.in_channels, d.out_channels, d.n_blocks = a.in_channels, a.out_channels, len(a.block_out_channels)
    for i, c in enumerate(a.block_out_channels):
        d.block_out_channels[i] = c
        d.down_block_attn[i] = int(a.down_block_attn[i])
    d.layers_per_block, d.heads, d.cross_attention_dim = a.layers_per_block, a.heads, a.cross_attention_dim
    for i in range(len(a.block_out_channels)):
        d.heads_per_block[i] = a.heads_per_block[i] if a.heads_per_block is not None else 0
    if a.prediction_type not in ("epsilon", "v_prediction"):
        raise ValueError(f"unsupported prediction_type {a.prediction_type!r}")
    d.prediction_type = int(a.prediction_type == "v_prediction")
    d.norm_groups, d.norm_eps = a.norm_groups, a.norm_eps
    d.vae_n_blocks = len(a.vae_block_out_channels)
    for i, c in enumerate(a.vae_block_out_channels):
        d.vae_block_out_channels[i] = c
    d.vae_layers_per_block, d.latent_channels, d.vae_scaling, d.ctx = a.vae_layers_per_block, a.latent_channels, a.vae_scaling, a.ctx
    d.beta_start, d.beta_end = a.beta_start, a.beta_end
    d.num_train_timesteps, d.steps_offset = a.num_train_timesteps, a.steps_offset
    return d


class SDKernels:
    """The ``tvc_sd_*`` entry points on one engine (handle)."""

    def __init__(self, engine: TVCEngine, arch: SDArch, unet_w: Optional[Dict] = None, vae_w: Optional[Dict] = None):
        self.engine, self.arch = engine, arch
        self.tensors = prepare_sd_tensors(unet_w, vae_w, engine.device)
        names = sorted(self.tensors)
        arr = (_lib.NamedTensor * len(names))()
        self._names = [n.encode() for n in names]
        for i, n in enumerate(names):
            arr[i].name = self._names[i]
            arr[i].ptr = self.tensors[n].data_ptr()
        desc = sd_desc(arch)
        with engine._lock, torch.cuda.device(engine.device):
            engine._check(engine.lib.tvc_sd_load(engine.handle, C.byref(desc), arr, len(names), _stream()))
            torch.cuda.current_stream().synchronize()

    def unet(self, latents: torch.Tensor, timestep: float, ctx: torch.Tensor) -> torch.Tensor:
        """latents fp32 [n, 4, H, W], ctx fp32 [n, 77, 768] -> predicted noise fp32 [n, 4, H, W]."""
        e = self.engine
        latents = latents.to(e.device, torch.float32).contiguous()
        ctx = ctx.to(e.device, torch.float32).contiguous()
        n, _, H, W = latents.shape
        out = torch.empty_like(latents)
        with e._lock, torch.cuda.device(e.device):
            e._check(e.lib.tvc_sd_unet(e.handle, _ptr(latents), n, H, W, float(timestep), _ptr(ctx), _ptr(out), _stream()))
        return out

    def vae_decode(self, latents: torch.Tensor) -> torch.Tensor:
        """latents fp32 [n, 4, H, W] -> images fp32 [n, 3, 8H, 8W] in [0, 1]."""
        e = self.engine
        latents = latents.to(e.device, torch.float32).contiguous()
        n, _, H, W = latents.shape
        up = 2 ** (len(self.arch.vae_block_out_channels) - 1)
        out = torch.empty((n, 3, H * up, W * up), dtype=torch.float32, device=e.device)
        with e._lock, torch.cuda.device(e.device):
            e._check(e.lib.tvc_sd_vae_decode(e.handle, _ptr(latents), n, H, W, _ptr(out), _stream()))
        return out

    def generate(self, cond: torch.Tensor, uncond: torch.Tensor, latents: torch.Tensor, steps: int, guidance: float,
                 decode: bool = True):
        """cond / uncond fp32 [n, 77, 768], latents fp32 [n, 4, H, W] initial noise -> (final latents, images | None)."""
        e = self.engine
        cond = cond.to(e.device, torch.float32).contiguous()
        uncond = uncond.to(e.device, torch.float32).contiguous()
        lat = latents.to(e.device, torch.float32).contiguous().clone()
        n, _, H, W = lat.shape
        up = 2 ** (len(self.arch.vae_block_out_channels) - 1)
        img = torch.empty((n, 3, H * up, W * up), dtype=torch.float32, device=e.device) if decode else None
        with e._lock, torch.cuda.device(e.device):
            e._check(e.lib.tvc_sd_generate(e.handle, _ptr(cond), _ptr(uncond), _ptr(lat), n, H, W, int(steps), float(guidance),
                                           _ptr(img), _stream()))
        return lat, img

    def block(self, kind: int, prefix: str, x: torch.Tensor, cout: int, temb: Optional[torch.Tensor] = None,
              ctx: Optional[torch.Tensor] = None, vae: bool = False) -> torch.Tensor:
        """One block on fp32 NCHW tensors (parity tests); ``kind`` as in include/tvc.h (tvc_sd_block)."""
        e = self.engine
        x = x.to(e.device, torch.float32).contiguous()
        n, cin, H, W = x.shape
        Ho, Wo = (H // 2, W // 2) if kind == 4 else ((2 * H, 2 * W) if kind == 5 else (H, W))
        out = torch.empty((n, cout, Ho, Wo), dtype=torch.float32, device=e.device)
        temb = None if temb is None else temb.to(e.device, torch.float32).contiguous()
        ctx = None if ctx is None else ctx.to(e.device, torch.float32).contiguous()
        with e._lock, torch.cuda.device(e.device):
            e._check(e.lib.tvc_sd_block(e.handle, kind, prefix.encode(), _ptr(x), n, cin, H, W, _ptr(temb), _ptr(ctx), cout,
                                        int(vae), _ptr(out), _stream()))
        return out

    def attention(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, n: int, heads: int) -> torch.Tensor:
        """q [n * Tq, heads * dh], k / v [n * Tk, heads * dh] bf16 -> [n * Tq, heads * dh] bf16."""
        e = self.engine
        q, k, v = (t.to(e.device, torch.bfloat16).contiguous() for t in (q, k, v))
        dh = q.shape[1] // heads
        out = torch.empty_like(q)
        with e._lock, torch.cuda.device(e.device):
            e._check(e.lib.tvc_sd_attention(e.handle, _ptr(q), _ptr(k), _ptr(v), _ptr(out), n, heads, q.shape[0] // n,
                                            k.shape[0] // n, dh, _stream()))
        return out


@dataclass
class SDModelConfig:
    model_name: str = "runwayml/stable-diffusion-v1-5"      # src/sd_ref.py:220
    device: str = "cuda"
    seed: int = 0                                           # random-init seed (no checkpoint without a network)
    random_init: bool = False                               # explicit opt-in: seeded random UNet / VAE (/ text tower) weights
    unet_weights: Optional[str] = None                      # diffusers unet / vae safetensors files, if present
    vae_weights: Optional[str] = None
    text_model: Optional[str] = None                        # conditioning tower; None = the architecture's (SDArch.text_arch:
                                                            # CLIP ViT-L/14 for SD 1.x, "SD2-text" = OpenCLIP ViT-H/14 for SD 2.x)
    tokenizer_dir: Optional[str] = None


class StableDiffusionModel:
    """``generate_image`` / ``generate`` as the reference calls them, plus the batched ``generate_batch``."""

    PAD_ID = synth.EOT          # the SD 1.x tokenizer pads with <|endoftext|> (tokenizer/special_tokens_map.json); SD 2.x pads with "!" (id 0)

    def __init__(self, config: Optional[SDModelConfig] = None, clip_model=None, arch: Optional[SDArch] = None,
                 weights: Optional[tuple] = None):
        from .clip import BPETokenizer, HashTokenizer
        self.config = config or SDModelConfig()
        if arch is None:
            # src/__init__.py:110-113: "stable-diffusion-v1-5" (the reference's default, src/sd_ref.py:221) and "stable-diffusion-2-1"
            arch = SDArch.sd21_base() if "stable-diffusion-2" in self.config.model_name else SDArch.sd15()
        self.arch = arch
        dev = self.config.device
        self.device = torch.device("cuda:0" if dev in ("cuda", "auto") else dev)
        # conditioning tower: the caller's CLIP when its text tower has the UNet's cross-attention width
        if bool(self.config.unet_weights) != bool(self.config.vae_weights):
            raise ValueError("SDModelConfig: unet_weights and vae_weights must be given together (one alone would leave the "
                             "other half of the model at random weights)")
        have_files = bool(self.config.unet_weights and self.config.vae_weights)
        if weights is None and not have_files and not self.config.random_init:
            raise RuntimeError(f"no weights for '{self.config.model_name}': pass unet_weights= / vae_weights= (diffusers "
                               "safetensors) or weights=, or opt into seeded random weights with SDModelConfig(random_init=True)")
        if clip_model is not None and clip_model.arch.text.width == self.arch.cross_attention_dim:
            self.text_engine, self.text_arch, self.tokenizer = clip_model.engine, clip_model.arch, clip_model.tokenizer
        else:
            self.text_arch = get_arch(self.config.text_model if self.config.text_model else self.arch.text_arch)
            if self.text_arch.text.width != self.arch.cross_attention_dim:
                raise ValueError("the text tower's width must equal the UNet's cross_attention_dim")
            if not self.config.random_init:
                raise RuntimeError("the caller's CLIP text tower does not have the UNet's cross-attention width and no "
                                   "conditioning tower was given: only SDModelConfig(random_init=True) builds a random one")
            logger.warning("latent-diffusion conditioning tower: seeded RANDOM %s text weights + hash tokenizer (random_init)",
                           self.text_arch.name)
            _, tw = synth.make_clip_weights(self.text_arch, self.config.seed)
            self.text_engine = TVCEngine(self.text_arch, None, tw, device=str(self.device))
            self.tokenizer = (BPETokenizer(self.config.tokenizer_dir, self.text_arch.ctx) if self.config.tokenizer_dir
                              else HashTokenizer(self.text_arch.ctx))
        if weights is None:
            if have_files:
                from safetensors.torch import load_file
                weights = (load_file(self.config.unet_weights), load_file(self.config.vae_weights))
            else:
                logger.warning("latent-diffusion model '%s': seeded RANDOM UNet / VAE weights (random_init=True) -- its images "
                               "are noise; for benchmarks and parity tests only", self.config.model_name)
                weights = make_sd_weights(self.arch, self.config.seed, device=str(self.device))
        self.kernels = SDKernels(self.text_engine, self.arch, weights[0], weights[1])
        self.generation_count = 0

    # ---- conditioning --------------------------------------------------------------------------------------
    def tokenize(self, prompts: Sequence[str]) -> torch.Tensor:
        ids = self.tokenizer(list(prompts)).clone()
        eot = ids.argmax(dim=1)
        pos = torch.arange(ids.shape[1]).unsqueeze(0)
        pad = 0 if self.arch.text_arch == "SD2-text" else self.PAD_ID
        ids[pos > eot.unsqueeze(1)] = pad                       # CLIP pads with 0, the SD 1.x pipeline with the EOT id
        return ids

    def encode_prompts(self, prompts: Sequence[str]) -> torch.Tensor:
        """[n, 77, 768] text states (``CLIPTextModel(...).last_hidden_state``)."""
        return self.text_engine.encode_text_hidden(self.tokenize(prompts).to(self.device, torch.int32))

    @staticmethod
    def initial_latents(seeds: Sequence[int], channels: int, h: int, w: int) -> torch.Tensor:
        """One seeded CPU generator per image (platform-independent; the oracle draws the same numbers)."""
        return torch.stack([torch.randn((channels, h, w), generator=torch.Generator().manual_seed(int(s))) for s in seeds])

    # ---- generation ----------------------------------------------------------------------------------------
    def generate_batch(self, prompts: Sequence[str], seeds: Sequence[int], num_inference_steps: int = 50,
                       guidance_scale: float = 7.5, height: int = 512, width: int = 512,
                       negative_prompts: Optional[Sequence[str]] = None, return_latents: bool = False) -> torch.Tensor:
        """n prompts x their seeds -> images fp32 [n, 3, height, width] in [0, 1] on the device; every UNet evaluation
        of the sampling loop runs on all 2n (unconditional | conditional) samples at once."""
        n = len(prompts)
        if len(seeds) != n:
            raise ValueError("one seed per prompt")
        up = 2 ** (len(self.arch.vae_block_out_channels) - 1)
        if height % (up * 8) or width % (up * 8):
            raise ValueError(f"height / width must be multiples of {up * 8}")
        cond = self.encode_prompts(prompts)
        uncond = self.encode_prompts(list(negative_prompts) if negative_prompts is not None else [""] * n)
        lat0 = self.initial_latents(seeds, self.arch.in_channels, height // up, width // up)
        lat, img = self.kernels.generate(cond, uncond, lat0, num_inference_steps, guidance_scale, decode=not return_latents)
        self.generation_count += n
        return lat if return_latents else img

    @staticmethod
    def to_pil(images: torch.Tensor) -> List:
        from PIL import Image
        a = (images.clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy()
        return [Image.fromarray(x) for x in a]

    def generate_image(self, prompt: str, num_images: int = 1, seed: Optional[int] = None, num_inference_steps: int = 50,
                       guidance_scale: float = 7.5, height: int = 512, width: int = 512, negative_prompt: Optional[str] = None) -> List:
        """``src/sd_ref.py:389-399``: list of PIL images (image i of a call uses seed + i)."""
        s0 = int(seed) if seed is not None else int(np.random.randint(0, 2 ** 31 - 1))
        imgs = self.generate_batch([prompt] * num_images, [s0 + i for i in range(num_images)], num_inference_steps,
                                   guidance_scale, height, width,
                                   [negative_prompt] * num_images if negative_prompt is not None else None)
        return self.to_pil(imgs)

    def generate(self, prompt: str, negative_prompt: Optional[str] = None, height: int = 512, width: int = 512,
                 guidance_scale: float = 7.5, num_inference_steps: int = 20, generator: Optional[torch.Generator] = None):
        """``experiments/defenses/generative_ref.py:139-147``: object with ``.images`` (one PIL image)."""
        seed = generator.initial_seed() if generator is not None else None
        return SimpleNamespace(images=self.generate_image(prompt, 1, seed, num_inference_steps, guidance_scale, height, width,
                                                          negative_prompt))


def create_sd_model(config: Optional[SDModelConfig] = None, **kw) -> StableDiffusionModel:
    return StableDiffusionModel(config or SDModelConfig(), **kw)
