"""SD reference generation -- the orchestration side (partial).

Mirror of ``src/sd_ref.py``: ``QualityMetrics`` (:24-46), ``GenerationResult`` (:49-84), ``QualityFilter``
(:87-163), ``SDReferenceConfig`` (:217-255), ``SDReferenceGenerator`` (:258-745), ``create_sd_reference_generator``
(:748).  What IS built here: the prompts x seeds loop, seed policy, the heuristic quality filter, caching, statistics,
and -- new -- ``reference_features``: the CLIP image embeddings of everything generated, encoded in ONE batched
``tvc_encode_image`` launch (what K6 of the detector consumes; ``src/detector.py:524-548``).

What is NOT built: the latent-diffusion model itself (UNet + VAE + scheduler).  The reference reaches it through
``StableDiffusionModel.generate_image(prompt=, num_images=, seed=, num_inference_steps=, guidance_scale=, height=,
width=)`` (:389-399), a wrapper that is absent from the reference snapshot too, like its weights; here it is an
injected object with that method (``sd_model=``), as the language model is for the text variants.  Its text
conditioning, the CLIP text tower's per-token output, IS available on the GPU: ``TVCEngine.encode_text_hidden``
(``tvc_encode_text_hidden``).
"""
from __future__ import annotations

import hashlib
import json
import logging
import random
import time
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

logger = logging.getLogger(__name__)


@dataclass
class QualityMetrics:
    """:24-46."""
    aesthetic_score: float = 0.0
    clip_score: float = 0.0
    safety_score: float = 1.0
    technical_score: float = 0.0
    overall_score: float = 0.0

    def to_dict(self) -> Dict[str, Any]:
        return dict(self.__dict__)

    @classmethod
    def from_dict(cls, data: Dict[str, Any]) -> "QualityMetrics":
        return cls(**data)


@dataclass
class GenerationResult:
    """:49-84."""
    images: List[Any]
    prompts: List[str]
    seeds: List[int]
    quality_metrics: List[QualityMetrics]
    generation_time: float = 0.0
    cache_hit: bool = False

    def is_high_quality(self, threshold: float = 0.5) -> List[bool]:
        return [m.overall_score >= threshold for m in self.quality_metrics]

    def filter_high_quality(self, threshold: float = 0.5) -> "GenerationResult":
        keep = self.is_high_quality(threshold)
        pick = lambda xs: [x for x, k in zip(xs, keep) if k]
        return GenerationResult(pick(self.images), pick(self.prompts), pick(self.seeds), pick(self.quality_metrics),
                                self.generation_time, self.cache_hit)

    def to_dict(self) -> Dict[str, Any]:
        return {"num_images": len(self.images), "prompts": self.prompts, "seeds": self.seeds,
                "quality_metrics": [m.to_dict() for m in self.quality_metrics],
                "generation_time": self.generation_time, "cache_hit": self.cache_hit}


def _pixels(image) -> np.ndarray:
    """PIL image / array / tensor -> array with the value range the reference's heuristics assume (0..255)."""
    if isinstance(image, torch.Tensor):
        a = image.detach().float().cpu().numpy()
        if a.ndim == 3 and a.shape[0] in (1, 3):
            a = np.transpose(a, (1, 2, 0))
        return a * 255.0 if a.max() <= 1.0 + 1e-6 else a
    return np.array(image)


class QualityFilter:
    """:87-163 (the reference's own 'simplified' scores, kept as they are: they are what its filter computes)."""

    def __init__(self, aesthetic_threshold: float = 0.5, clip_threshold: float = 0.5):
        self.aesthetic_threshold = aesthetic_threshold
        self.clip_threshold = clip_threshold

    def compute_aesthetic_score(self, image) -> float:
        a = _pixels(image)
        return float(min(1.0, (np.var(a) + np.std(a)) / 10000))

    def compute_clip_score(self, image, prompt: str) -> float:
        return float(min(1.0, len(prompt.split()) / 20))           # :104-109 -- a prompt-length proxy, not CLIP

    def compute_safety_score(self, image) -> float:
        return 1.0

    def compute_technical_score(self, image) -> float:
        return float(min(1.0, np.var(_pixels(image)) / 5000))

    def evaluate_quality(self, image, prompt: str = "") -> QualityMetrics:
        a, c = self.compute_aesthetic_score(image), (self.compute_clip_score(image, prompt) if prompt else 0.0)
        s, t = self.compute_safety_score(image), self.compute_technical_score(image)
        return QualityMetrics(a, c, s, t, (a + c + s + t) / 4)

    def filter_images(self, images: Sequence[Any], prompts: Optional[Sequence[str]] = None):
        prompts = prompts if prompts is not None else [""] * len(images)
        out_i, out_m = [], []
        for im, p in zip(images, prompts):
            m = self.evaluate_quality(im, p)
            if m.overall_score >= min(self.aesthetic_threshold, self.clip_threshold):
                out_i.append(im)
                out_m.append(m)
        return out_i, out_m

    def batch_evaluate_quality(self, images: Sequence[Any], prompts: Optional[Sequence[str]] = None) -> List[QualityMetrics]:
        prompts = prompts if prompts is not None else [""] * len(images)
        return [self.evaluate_quality(im, p) for im, p in zip(images, prompts)]


@dataclass
class SDReferenceConfig:
    """:217-255 (same names and defaults)."""
    sd_model: str = "runwayml/stable-diffusion-v1-5"
    device: str = "cuda"
    torch_dtype: str = "float16"
    num_images_per_prompt: int = 3
    num_inference_steps: int = 50
    guidance_scale: float = 7.5
    height: int = 512
    width: int = 512
    use_text_variants: bool = True
    num_text_variants: int = 3
    variant_methods: Optional[List[str]] = None
    use_fixed_seeds: bool = True
    seed_range: Tuple[int, int] = (0, 10000)
    enable_safety_checker: bool = False
    filter_low_quality: bool = True
    quality_threshold: float = 0.5
    enable_cache: bool = True
    cache_dir: Optional[str] = None
    batch_size: int = 4

    def __post_init__(self):
        if self.variant_methods is None:
            self.variant_methods = ["synonym", "paraphrase"]


class SDReferenceGenerator:
    """``SDReferenceGenerator(config, sd_model=..., text_augmenter=..., clip_model=...)``.  ``sd_model``: object with
    ``generate_image(prompt=, num_images=, seed=, num_inference_steps=, guidance_scale=, height=, width=) -> list of
    images`` (the reference builds its own ``StableDiffusionModel``, :291-317; absent there, injected here)."""

    def __init__(self, config: Optional[SDReferenceConfig] = None, sd_model=None, text_augmenter=None, clip_model=None):
        self.config = config or SDReferenceConfig()
        self.sd_model = sd_model
        self.text_augmenter = text_augmenter if self.config.use_text_variants else None
        self.clip_model = clip_model
        self.generation_cache: Dict[str, Dict[str, Any]] = {}
        self.generation_stats = {"total_generated": 0, "cache_hits": 0, "generation_time": 0.0}
        self._rng = random.Random(0)

    # ---- the prompts x seeds loop (:342-452) ---------------------------------------------------------------
    def generate_reference_images(self, prompt: str, num_images: Optional[int] = None, use_variants: Optional[bool] = None,
                                  seeds: Optional[List[int]] = None) -> Dict[str, Any]:
        c = self.config
        try:
            if self.sd_model is None:
                raise RuntimeError("no sd_model injected (the latent-diffusion model is not part of this build)")
            num_images = num_images or c.num_images_per_prompt
            use_variants = use_variants if use_variants is not None else c.use_text_variants
            key = self._get_cache_key(prompt, num_images, use_variants, seeds)
            if c.enable_cache and key in self.generation_cache:
                self.generation_stats["cache_hits"] += 1
                return self.generation_cache[key]
            t0 = time.time()
            prompts = [prompt]
            if use_variants and self.text_augmenter is not None:
                try:
                    variants = self.text_augmenter.generate_variants(prompt, methods=c.variant_methods)
                except TypeError:                                  # generators without a `methods` argument
                    variants = self.text_augmenter.generate_variants(prompt)
                prompts.extend(list(variants)[:c.num_text_variants])
            if seeds is None:
                seeds = self._generate_seeds(num_images * len(prompts))
            images, used_p, used_s, k = [], [], [], 0
            for p in prompts:
                for _ in range(num_images):
                    seed = seeds[k] if k < len(seeds) else self._rng.randint(*c.seed_range)
                    got = self.sd_model.generate_image(prompt=p, num_images=1, seed=seed,
                                                       num_inference_steps=c.num_inference_steps,
                                                       guidance_scale=c.guidance_scale, height=c.height, width=c.width)
                    if got:
                        images.extend(got)
                        used_p.append(p)
                        used_s.append(seed)
                    k += 1
            if c.filter_low_quality:
                images, used_p, used_s = self._filter_low_quality_images(images, used_p, used_s)
            result = {"images": images, "prompts": used_p, "seeds": used_s, "original_prompt": prompt,
                      "generation_time": time.time() - t0, "num_generated": len(images)}
            if c.enable_cache:
                self.generation_cache[key] = result
            self.generation_stats["total_generated"] += len(images)
            self.generation_stats["generation_time"] += result["generation_time"]
            return result
        except Exception as e:                                     # noqa: BLE001 -- :441-452 reports, never raises
            logger.error("reference image generation failed: %s", e)
            return {"images": [], "prompts": [], "seeds": [], "original_prompt": prompt, "generation_time": 0.0,
                    "num_generated": 0, "error": str(e)}

    def batch_generate_reference_images(self, prompts: List[str], num_images_per_prompt: Optional[int] = None) -> List[Dict[str, Any]]:
        return [self.generate_reference_images(p, num_images=num_images_per_prompt) for p in prompts]      # :588-609

    # ---- what the detector's K6 stage consumes -----------------------------------------------------------------
    def reference_features(self, prompts: Sequence[str], num_images: Optional[int] = None):
        """CLIP image embeddings of the references of every prompt: ONE batched image-tower launch over all generated
        images -> (features [n_total, D] on the device, counts per prompt).  Replaces the per-prompt
        ``encode_image(sd_references)`` of ``src/detector.py:527-533``."""
        if self.clip_model is None:
            raise ValueError("reference_features needs a clip_model")
        results = self.batch_generate_reference_images(list(prompts), num_images)
        imgs = [im for r in results for im in r["images"]]
        counts = [len(r["images"]) for r in results]
        if not imgs:
            return torch.zeros((0, self.clip_model.arch.embed_dim), device=self.clip_model.device), counts
        x, _ = self.clip_model._images_to_device(imgs)
        return self.clip_model.engine.encode_image(x, True), counts

    def generate_reference_vectors(self, prompt: str, num_images: Optional[int] = None) -> np.ndarray:
        """:611-648: the VAE latents of the generated images, flattened (``sd_model.encode_image``)."""
        try:
            images = self.generate_reference_images(prompt, num_images)["images"]
            if not images:
                return np.array([])
            return np.array([np.asarray(self.sd_model.encode_image(im)).flatten() for im in images])
        except Exception as e:                                     # noqa: BLE001
            logger.error("reference vector generation failed: %s", e)
            return np.array([])

    # ---- helpers (:454-586) ----------------------------------------------------------------------------------
    def _get_cache_key(self, prompt: str, num_images: int, use_variants: bool, seeds: Optional[List[int]]) -> str:
        c = self.config
        key = {"prompt": prompt, "num_images": num_images, "use_variants": use_variants, "seeds": seeds,
               "config": {"sd_model": c.sd_model, "num_inference_steps": c.num_inference_steps,
                          "guidance_scale": c.guidance_scale, "height": c.height, "width": c.width}}
        return hashlib.md5(json.dumps(key, sort_keys=True).encode()).hexdigest()

    def _generate_seeds(self, num_seeds: int) -> List[int]:
        c = self.config
        if not c.use_fixed_seeds:
            return [self._rng.randint(*c.seed_range) for _ in range(num_seeds)]
        seeds = list(range(c.seed_range[0], min(c.seed_range[0] + num_seeds, c.seed_range[1])))
        while len(seeds) < num_seeds:
            s = self._rng.randint(*c.seed_range)
            if s not in seeds:
                seeds.append(s)
        return seeds

    def _filter_low_quality_images(self, images, prompts, seeds):
        try:
            keep = [i for i, im in enumerate(images) if self._assess_image_quality(im) >= self.config.quality_threshold]
            return [images[i] for i in keep], [prompts[i] for i in keep], [seeds[i] for i in keep]
        except Exception as e:                                     # noqa: BLE001 -- :543-545
            logger.warning("quality filter failed: %s", e)
            return images, prompts, seeds

    def _assess_image_quality(self, image) -> float:
        """:547-586."""
        try:
            a = _pixels(image)
            if a.std() < 10:
                return 0.0
            contrast = a.std() / 255.0
            brightness_score = 1.0 - abs(a.mean() / 255.0 - 0.5) * 2
            color_score = min(np.var(a, axis=(0, 1)).mean() / 1000.0, 1.0) if a.ndim == 3 else 0.5
            return float(min(contrast * 0.4 + brightness_score * 0.3 + color_score * 0.3, 1.0))
        except Exception as e:                                     # noqa: BLE001
            logger.warning("image quality assessment failed: %s", e)
            return 0.5

    def clear_cache(self) -> None:
        self.generation_cache.clear()

    def get_stats(self) -> Dict[str, Any]:
        s = dict(self.generation_stats)
        s["cache_size"] = len(self.generation_cache)
        s["average_generation_time"] = s["generation_time"] / s["total_generated"] if s["total_generated"] else 0.0
        return s

    def update_config(self, **kwargs) -> None:
        for k, v in kwargs.items():
            if hasattr(self.config, k):
                setattr(self.config, k, v)


def create_sd_reference_generator(config: Optional[SDReferenceConfig] = None, **kw) -> SDReferenceGenerator:
    return SDReferenceGenerator(config or SDReferenceConfig(), **kw)
