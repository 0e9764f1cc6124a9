"""SD reference generation -- the orchestration side (partial).

Mirror of ``src/sd_ref.py``: ``QualityMetrics`` (:24-46), ``GenerationResult`` (:49-84), ``QualityFilter``
(:87-163), ``SDReferenceConfig`` (:217-255), ``SDReferenceGenerator`` (:258-745), ``create_sd_reference_generator``
(:748).  What IS built here: the prompts x seeds loop, seed policy, the heuristic quality filter, caching, statistics,
and -- new -- ``reference_features``: the CLIP image embeddings of everything generated, encoded in ONE batched
``tvc_encode_image`` launch (what K6 of the detector consumes; ``src/detector.py:524-548``).

The latent-diffusion model itself (UNet + VAE decoder + PNDM scheduler) is ``sd_model.StableDiffusionModel`` on the HIP
kernels of ``tvc_sd_*``: the reference reaches it through ``StableDiffusionModel.generate_image(prompt=, num_images=,
seed=, num_inference_steps=, guidance_scale=, height=, width=)`` (:389-399).  ``sd_model=None`` builds that model
(``sd_model="none"`` keeps the generator without one, as the reference does when the pipeline fails to load); any
object with ``generate_image`` can be injected instead.  ``reference_features`` is the batched form: every prompt x
seed of a batch is denoised in the SAME UNet launches (``generate_batch``), preprocessed on the device
(``tvc_preprocess_images``) and encoded in one image-tower launch -- no PIL round trip.

``GenerativeReferenceGenerator`` mirrors ``experiments/defenses/generative_ref.py`` (:31-330): prompt preprocessing,
``generate_references`` -> image tensors resized to 224 and ImageNet-normalised, consistency / quality metrics.
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
    # not in the reference: where the in-tree model's weights come from (diffusers safetensors), or the explicit opt-in to
    # seeded random weights (benchmarks / parity tests); with neither, building the model fails as the reference's
    # pipeline load does without a checkpoint, and every generation call reports an error (:291-317)
    unet_weights: Optional[str] = None
    vae_weights: Optional[str] = None
    random_init: bool = False

    def __post_init__(self):
        if self.variant_methods is None:
            self.variant_methods = ["synonym", "paraphrase"]


class SDReferenceGenerator:
    """``SDReferenceGenerator(config, sd_model=..., text_augmenter=..., clip_model=...)``.  ``sd_model``: object with
    ``generate_image(prompt=, num_images=, seed=, num_inference_steps=, guidance_scale=, height=, width=) -> list of
    images`` (the reference builds its own ``StableDiffusionModel``, :291-317; absent there, injected here)."""

    def __init__(self, config: Optional[SDReferenceConfig] = None, sd_model=None, text_augmenter=None, clip_model=None):
        self.config = config or SDReferenceConfig()
        if sd_model is None:
            # src/sd_ref.py:291-317 builds its StableDiffusionModel here and, when that fails, logs and goes on without one
            # (every generation call then reports an error instead of raising) -- mirrored
            try:
                from .sd_model import SDModelConfig, StableDiffusionModel
                sd_model = StableDiffusionModel(SDModelConfig(model_name=self.config.sd_model, device=self.config.device,
                                                              unet_weights=self.config.unet_weights,
                                                              vae_weights=self.config.vae_weights,
                                                              random_init=self.config.random_init), clip_model=clip_model)
            except Exception as e:                                 # noqa: BLE001
                logger.error("could not build the latent-diffusion model: %s", e)
                sd_model = None
        elif isinstance(sd_model, str) and sd_model == "none":
            sd_model = None
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
                raise RuntimeError("this generator has no latent-diffusion model (sd_model='none', or building it failed: no GPU?)")
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
        if hasattr(self.sd_model, "generate_batch") and not self.config.filter_low_quality and \
                not (self.config.use_text_variants and self.text_augmenter is not None):
            # the batched path: all prompts x seeds in the same UNet launches, pixels stay on the device
            c = self.config
            num = num_images or c.num_images_per_prompt
            flat_p = [p for p in prompts for _ in range(num)]
            flat_s = [s for _ in prompts for s in self._generate_seeds(num)]
            t0 = time.time()
            imgs = self.sd_model.generate_batch(flat_p, flat_s, c.num_inference_steps, c.guidance_scale, c.height, c.width)
            feats = self.clip_model.engine.encode_image(self.clip_model.preprocess_tensor(imgs), True)
            self.generation_stats["total_generated"] += len(flat_p)
            self.generation_stats["generation_time"] += time.time() - t0
            return feats, [num] * len(prompts)
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


# ---------------------------------------------------------------------------------------------------------------
@dataclass
class GenerativeConfig:
    """``experiments/defenses/generative_ref.py:17-28`` (same names and defaults)."""
    generation_count: int = 3
    image_size: int = 512
    guidance_scale: float = 7.5
    num_inference_steps: int = 20
    seed: Optional[int] = None
    negative_prompt: str = "blurry, low quality, distorted, deformed"
    use_safety_checker: bool = True
    batch_size: int = 1
    device: str = "cuda"


class GenerativeReferenceGenerator:
    """Mirror of ``experiments/defenses/generative_ref.py:31-330``: ``generate_references(text)`` -> list of image
    tensors [3, 224, 224] (resized to 224 x 224 bilinear, ImageNet mean / std, :55-59), ``batch_generate_references``,
    ``compute_generation_consistency``, ``evaluate_generation_quality``, statistics.  ``sd_model`` = anything with the
    reference's ``generate(prompt=, negative_prompt=, height=, width=, guidance_scale=, num_inference_steps=, generator=)``;
    a ``StableDiffusionModel`` is driven through its batched form instead (all ``generation_count`` images of a text
    share the UNet launches, resize + normalise on the device)."""

    IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    QUALITY_WORDS = ("high quality", "detailed", "realistic", "professional photography")

    def __init__(self, sd_model, clip_model, config: Optional[GenerativeConfig] = None):
        self.sd_model, self.clip_model = sd_model, clip_model
        self.config = config or GenerativeConfig()
        self.generation_stats = {"total_generations": 0, "successful_generations": 0, "failed_generations": 0,
                                 "average_generation_time": 0.0}

    def _preprocess_text(self, text: str) -> str:
        """:160-184."""
        t = text.strip()
        if not any(w in t.lower() for w in self.QUALITY_WORDS):
            t = f"{t}, high quality, detailed"
        if len(t) > 200:
            t = t[:200].rsplit(" ", 1)[0]
        return t

    def _to_clip_tensor(self, images01: torch.Tensor) -> torch.Tensor:
        return self.clip_model.engine.preprocess_images(images01, 224, self.IMAGENET_MEAN, self.IMAGENET_STD, bicubic=False,
                                                        keep_aspect=False)

    def generate_references(self, text: str) -> List[torch.Tensor]:
        c = self.config
        t0 = time.time()
        out: List[torch.Tensor] = []
        try:
            prompt = self._preprocess_text(text)
            if hasattr(self.sd_model, "generate_batch"):
                seeds = [(c.seed + i) if c.seed is not None else random.randint(0, 2 ** 31 - 1) for i in range(c.generation_count)]
                imgs = self.sd_model.generate_batch([prompt] * c.generation_count, seeds, c.num_inference_steps, c.guidance_scale,
                                                    c.image_size, c.image_size, [c.negative_prompt] * c.generation_count)
                out = list(self._to_clip_tensor(imgs))
                self.generation_stats["successful_generations"] += len(out)
            else:
                for i in range(c.generation_count):
                    try:
                        gen = torch.Generator().manual_seed(c.seed + i) if c.seed is not None else None
                        res = self.sd_model.generate(prompt=prompt, negative_prompt=c.negative_prompt, height=c.image_size,
                                                     width=c.image_size, guidance_scale=c.guidance_scale,
                                                     num_inference_steps=c.num_inference_steps, generator=gen)
                        img = res.images[0] if hasattr(res, "images") and res.images else (res[0] if isinstance(res, list) and res else res)
                        if img is None:
                            self.generation_stats["failed_generations"] += 1
                            continue
                        a = torch.from_numpy(np.asarray(img.convert("RGB"), dtype=np.float32) / 255.0).permute(2, 0, 1)[None]
                        out.append(self._to_clip_tensor(a.to(self.clip_model.device))[0])
                        self.generation_stats["successful_generations"] += 1
                    except Exception as e:                     # noqa: BLE001 -- :118-121
                        logger.warning("generation %d failed: %s", i + 1, e)
                        self.generation_stats["failed_generations"] += 1
            self.generation_stats["total_generations"] += c.generation_count
            n = self.generation_stats["total_generations"]
            avg = self.generation_stats["average_generation_time"]
            self.generation_stats["average_generation_time"] = (avg * (n - c.generation_count) + (time.time() - t0)) / n
        except Exception as e:                                 # noqa: BLE001 -- :132-134
            logger.error("reference generation failed: %s", e)
            out = []
        return out

    def batch_generate_references(self, texts: List[str]) -> List[List[torch.Tensor]]:
        return [self.generate_references(t) for t in texts]

    def compute_generation_consistency(self, text: str, generated_images: List[torch.Tensor]) -> Dict[str, float]:
        """:224-275 with ONE image-tower launch and one cosine matrix instead of a launch per image / pair."""
        if not generated_images:
            return {"text_image_consistency": 0.0, "inter_image_consistency": 0.0}
        tf = self.clip_model.encode_tokens(self.clip_model.tokenize([text]), True)
        fi = self.clip_model.engine.encode_image(torch.stack(list(generated_images)).to(self.clip_model.device), True)
        ti = (fi @ tf[0]).cpu().numpy().astype(np.float64)
        g = (fi @ fi.t()).cpu().numpy().astype(np.float64)
        iu = np.triu_indices(len(generated_images), 1)
        return {"text_image_consistency": float(ti.mean()),
                "inter_image_consistency": float(g[iu].mean()) if len(iu[0]) else 1.0,
                "text_image_std": float(ti.std()), "generation_count": len(generated_images)}

    def evaluate_generation_quality(self, text: str, generated_images: List[torch.Tensor],
                                    reference_image: Optional[torch.Tensor] = None) -> Dict[str, Any]:
        """:277-330."""
        if not generated_images:
            return {"message": "no generated images to evaluate"}
        ev: Dict[str, Any] = {"generation_count": len(generated_images),
                              "consistency_metrics": self.compute_generation_consistency(text, generated_images)}
        fi = self.clip_model.engine.encode_image(torch.stack(list(generated_images)).to(self.clip_model.device), True)
        if reference_image is not None:
            fr = self.clip_model.engine.encode_image(reference_image.unsqueeze(0).to(self.clip_model.device), True)
            s = (fi @ fr[0]).cpu().numpy().astype(np.float64)
            ev["reference_similarity"] = {"mean": float(s.mean()), "std": float(s.std()), "max": float(s.max()), "min": float(s.min())}
        if len(generated_images) >= 2:
            g = (fi @ fi.t()).cpu().numpy().astype(np.float64)
            iu = np.triu_indices(len(generated_images), 1)
            ev["diversity_score"] = float(np.clip(1.0 - g[iu].mean(), 0.0, 1.0))                      # :321-346
        else:
            ev["diversity_score"] = 0.0
        cm = ev["consistency_metrics"]
        inter = 1.0 - abs(cm.get("inter_image_consistency", 0.0) - 0.7)                                # :348-369, ideal value 0.7
        ev["overall_quality"] = float(np.clip(0.5 * cm.get("text_image_consistency", 0.0) + 0.3 * inter +
                                              0.2 * ev["diversity_score"], 0.0, 1.0))
        return ev

    def get_statistics(self) -> Dict[str, Any]:
        s = dict(self.generation_stats)
        tot = s["successful_generations"] + s["failed_generations"]
        s["success_rate"] = s["successful_generations"] / tot if tot else 0.0
        return s
