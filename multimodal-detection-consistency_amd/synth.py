"""Synthetic inputs of the reference's shapes: random-init
CLIP weights, ImageNet-normalised smooth images, CLIP-BPE-shaped token ids with
variants, L2-normalised Gaussian banks.  Pure data generation on seeded
generators -- there is no network for real weights or datasets.

The same tensors are handed to the HIP path and (in tests / the CPU baseline)
to the oracle, so parity never depends on the generator.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

from .arch import ClipArch, Tower

IMAGENET_MEAN = (0.485, 0.456, 0.406)    # src/utils/config.py:60
IMAGENET_STD = (0.229, 0.224, 0.225)     # src/utils/config.py:61
SOT, EOT = 49406, 49407


def _normal(gen: torch.Generator, shape, std: float) -> torch.Tensor:
    return torch.randn(shape, generator=gen, dtype=torch.float32) * std


def _tower_layers(gen: torch.Generator, t: Tower) -> list:
    d, L = t.width, t.layers
    attn_std = d ** -0.5 * (2 * L) ** -0.5
    out_std = d ** -0.5
    fc1_std = (2 * d) ** -0.5
    fc2_std = d ** -0.5 * (2 * L) ** -0.5
    layers = []
    for _ in range(L):
        layers.append({
            'ln1_g': 1.0 + _normal(gen, (d,), 0.1), 'ln1_b': _normal(gen, (d,), 0.02),
            'wqkv': _normal(gen, (3 * d, d), attn_std), 'bqkv': _normal(gen, (3 * d,), 0.02),
            'wo': _normal(gen, (d, d), out_std), 'bo': _normal(gen, (d,), 0.02),
            'ln2_g': 1.0 + _normal(gen, (d,), 0.1), 'ln2_b': _normal(gen, (d,), 0.02),
            'w1': _normal(gen, (t.mlp, d), fc1_std), 'b1': _normal(gen, (t.mlp,), 0.02),
            'w2': _normal(gen, (d, t.mlp), fc2_std), 'b2': _normal(gen, (d,), 0.02),
        })
    return layers


def make_clip_weights(arch: ClipArch, seed: int = 0) -> Tuple[Dict, Dict]:
    """Random-init CLIP of geometry ``arch`` (CLIP-style init scales; LayerNorm
    gains / biases perturbed so that parity tests exercise them).  fp32, CPU."""
    gen = torch.Generator().manual_seed(seed)
    v, t = arch.vision, arch.text
    vision = {
        'patch_w': _normal(gen, (v.width, arch.patch_k), 0.02),
        'cls': _normal(gen, (v.width,), v.width ** -0.5),
        'pos': _normal(gen, (arch.vision_tokens, v.width), 0.02),
        'ln_pre_g': 1.0 + _normal(gen, (v.width,), 0.1), 'ln_pre_b': _normal(gen, (v.width,), 0.02),
        'layers': _tower_layers(gen, v),
        'ln_post_g': 1.0 + _normal(gen, (v.width,), 0.1), 'ln_post_b': _normal(gen, (v.width,), 0.02),
        'proj': _normal(gen, (arch.embed_dim, v.width), v.width ** -0.5),
    }
    text = {
        'tok_emb': _normal(gen, (arch.vocab, t.width), 0.02),
        'pos': _normal(gen, (arch.ctx, t.width), 0.01),
        'layers': _tower_layers(gen, t),
        'ln_final_g': 1.0 + _normal(gen, (t.width,), 0.1), 'ln_final_b': _normal(gen, (t.width,), 0.02),
        'proj': _normal(gen, (arch.embed_dim, t.width), t.width ** -0.5),
    }
    return vision, text


def make_images(n: int, size: int = 224, seed: int = 1, block: int = 8) -> torch.Tensor:
    """[n, 3, size, size] fp32: U[0,1] on ``block`` x ``block`` cells (nearest
    upsample), then ImageNet-normalised as the reference's loader does
    (src/utils/data_loader.py:466-473)."""
    gen = torch.Generator().manual_seed(seed)
    cells = (size + block - 1) // block
    low = torch.rand((n, 3, cells, cells), generator=gen, dtype=torch.float32)
    img = low.repeat_interleave(block, 2).repeat_interleave(block, 3)[:, :, :size, :size]
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    return ((img - mean) / std).contiguous()


def make_tokens(n: int, n_variants: int, ctx: int = 77, seed: int = 2,
                min_len: int = 5, max_len: int = 20, replace_ratio: float = 0.3) -> torch.Tensor:
    """int32 [n, n_variants + 1, ctx]: SOT, ``len`` ids in [1, 49405], EOT, 0-pad.
    Variant v = original with ceil(0.3 * len) positions resampled (mirrors
    ``synonym_replacement_ratio`` 0.3, src/text_augment.py:57), seed + 1 + v."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, n_variants + 1, ctx), dtype=np.int32)
    lens = rng.integers(min_len, max_len + 1, size=n)
    for i in range(n):
        L = int(lens[i])
        out[i, :, 0] = SOT
        out[i, 0, 1:1 + L] = rng.integers(1, SOT, size=L)
        out[i, :, 1 + L] = EOT
    for v in range(n_variants):
        vr = np.random.default_rng(seed + 1 + v)
        for i in range(n):
            L = int(lens[i])
            ids = out[i, 0, 1:1 + L].copy()
            k = int(math.ceil(replace_ratio * L))
            pos = vr.choice(L, size=k, replace=False)
            ids[pos] = vr.integers(1, SOT, size=k)
            out[i, v + 1, 1:1 + L] = ids
    return torch.from_numpy(out)


def make_bank(R: int, D: int, seed: int = 7, device: str = "cpu", dtype=torch.float32,
              chunk: int = 1 << 18) -> torch.Tensor:
    """[R, D] Gaussian rows, L2-normalised in fp32 (the on-disk ``features.npy``
    format, scripts/build_faiss_indices.py:108-109), cast to ``dtype`` last."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev).manual_seed(seed)
    out = torch.empty((R, D), dtype=dtype, device=dev)
    for r0 in range(0, R, chunk):
        n = min(chunk, R - r0)
        x = torch.randn((n, D), generator=gen, dtype=torch.float32, device=dev)
        x = x / x.norm(dim=-1, keepdim=True)
        out[r0:r0 + n] = x.to(dtype)
    return out


def plant_neighbours(bank: torch.Tensor, anchors: torch.Tensor, per_anchor: int = 3,
                     noise_lo: float = 0.15, noise_hi: float = 2.0, seed: int = 11) -> torch.Tensor:
    """Overwrite ``len(anchors) * per_anchor`` random bank rows with noisy copies
    of ``anchors`` (unit rows), so that retrieval finds references above the 0.3
    similarity threshold (random unit vectors never reach it).  The noise norm
    is drawn per row from [noise_lo, noise_hi]: cosines from ~0.99 (exercises
    the 0.95 de-duplication) down to ~0.45."""
    gen = torch.Generator().manual_seed(seed)
    R, D = bank.shape
    n = anchors.shape[0] * per_anchor
    rows = torch.randperm(R, generator=gen)[:n]
    a = anchors.float().cpu().repeat_interleave(per_anchor, 0)
    a = a / a.norm(dim=-1, keepdim=True)
    level = noise_lo + (noise_hi - noise_lo) * torch.rand((a.shape[0], 1), generator=gen)
    x = a + level * torch.randn(a.shape, generator=gen) / math.sqrt(D)
    x = x / x.norm(dim=-1, keepdim=True)
    bank = bank.clone()
    bank[rows.to(bank.device)] = x.to(bank.dtype).to(bank.device)
    return bank
