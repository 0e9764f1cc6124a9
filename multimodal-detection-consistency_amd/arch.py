"""CLIP geometries the hot path supports (OpenAI-CLIP layouts).

The reference selects the model by name through ``CLIPConfig(model_name=...)``
(``src/retrieval.py:356-361``; ``"ViT-B/32"`` default, ``"ViT-L/14"`` supported,
``src/__init__.py:99-108``).  ``"ViT-T/16-test"`` is a tiny geometry used only
by the test-suite and ``smoke()``.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Tower:
    width: int
    layers: int
    heads: int
    mlp: int
    act: str = "quick_gelu"     # "quick_gelu" (OpenAI CLIP) or "gelu" (exact erf GELU: OpenCLIP ViT-H/14, the SD-2.x text encoder)


@dataclass(frozen=True)
class ClipArch:
    name: str
    image_size: int
    patch: int
    vision: Tower
    text: Tower
    embed_dim: int
    vocab: int = 49408
    ctx: int = 77

    @property
    def n_patches(self) -> int:
        return (self.image_size // self.patch) ** 2

    @property
    def vision_tokens(self) -> int:
        return self.n_patches + 1

    @property
    def patch_k(self) -> int:
        return 3 * self.patch * self.patch

    @property
    def patch_k_padded(self) -> int:
        return (self.patch_k + 63) // 64 * 64

    def flops_image(self) -> float:
        """Forward FLOPs (2 x MACs) of the vision tower for one image."""
        v, T = self.vision, self.vision_tokens
        per_layer = 2 * T * v.width * (3 * v.width + v.width + 2 * v.mlp) + 4 * T * T * v.width
        return 2 * self.n_patches * self.patch_k * v.width + v.layers * per_layer + 2 * v.width * self.embed_dim

    def flops_text(self) -> float:
        t, T = self.text, self.ctx
        per_layer = 2 * T * t.width * (3 * t.width + t.width + 2 * t.mlp) + 4 * T * T * t.width
        return t.layers * per_layer + 2 * t.width * self.embed_dim


ARCHS = {
    "ViT-B/32": ClipArch("ViT-B/32", 224, 32, Tower(768, 12, 12, 3072), Tower(512, 12, 8, 2048), 512),
    "ViT-L/14": ClipArch("ViT-L/14", 224, 14, Tower(1024, 24, 16, 4096), Tower(768, 12, 12, 3072), 768),
    # the text encoder of Stable Diffusion 2.x (text_encoder/config.json of stabilityai/stable-diffusion-2-1-base: OpenCLIP
    # ViT-H/14's text tower cut to 23 layers -- its penultimate layer --, width 1024, 16 heads, erf GELU, projection 1024).
    # TEXT ONLY: the vision entry is a placeholder geometry that is never instantiated (ViT-H's 1280-wide vision tower is
    # outside the kernels' width limit and the detector never uses it).
    "SD2-text": ClipArch("SD2-text", 224, 14, Tower(1024, 1, 16, 4096, "gelu"), Tower(1024, 23, 16, 4096, "gelu"), 1024),
    "ViT-T/16-gelu-test": ClipArch("ViT-T/16-gelu-test", 64, 16, Tower(256, 2, 4, 512, "gelu"), Tower(128, 2, 2, 256, "gelu"), 128,
                                   vocab=49408, ctx=77),
    "ViT-T/16-test": ClipArch("ViT-T/16-test", 64, 16, Tower(256, 2, 4, 512), Tower(128, 2, 2, 256), 128,
                              vocab=49408, ctx=77),
}


# the names the reference's configs use for the same geometries (src/attacks/hubness_attack.py:43,
# configs/default.yaml: "openai/clip-vit-base-patch32", "ViT-B/32", "ViT-B-32")
ALIASES = {"openai/clip-vit-base-patch32": "ViT-B/32", "ViT-B-32": "ViT-B/32", "clip-vit-base-patch32": "ViT-B/32",
           "openai/clip-vit-large-patch14": "ViT-L/14", "ViT-L-14": "ViT-L/14", "clip-vit-large-patch14": "ViT-L/14"}


def get_arch(name: str) -> ClipArch:
    name = ALIASES.get(name, name)
    try:
        return ARCHS[name]
    except KeyError:
        raise ValueError(f"unknown CLIP geometry {name!r}; supported: {sorted(ARCHS)}") from None
