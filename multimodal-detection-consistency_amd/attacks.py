"""PGD and Hubness attack inner loops on the GPU.

Mirror of ``src/attacks/pgd_attack.py`` (``PGDAttackConfig`` :19-58, ``PGDAttacker`` :60-640,
``create_pgd_attacker`` :643): same constructor, ``attack`` / ``batch_attack`` / stats / cache surface, same
result dictionaries.  What runs per step is different:

* forward + input gradient = ``tvc_encode_image_grad`` + ``tvc_encode_image_backward`` (hand-written HIP; only
  d/d pixels exists, no weight gradient is ever formed);
* the loss ``mean_b cos(f_b, t_b)`` (:472-476,486-492) has the closed-form embedding gradient ``+-t_b / B`` on
  unit rows, so no autograd graph is built at all -- the kernel's own L2-normalise backward projects it;
* momentum, sign step, eps projection and clamp (:500-521) are ONE kernel (``tvc_pgd_step``), in place.

``CLIPModel.encode_image_tensor(x, requires_grad=True)`` offers the same gradient behind ``torch.autograd`` for
callers that write the loop themselves (Hubness / C&W, ``src/attacks/hubness_attack.py:586``).

On purpose not carried over: nn.DataParallel (:131-133; one process per GPU instead -- shard the image list across
ranks), the GradScaler branch (:452-471; the tower is bf16-MFMA / fp32-accumulate already, there is nothing to
scale), ``gradient_clip_value`` (:497-498 clips a tensor whose ``.grad`` is then never read).  The random start
draws from a seeded CPU generator (the reference uses the global device RNG, :437-442), so a run is reproducible
and equals ``oracle/synth_pgd.py`` on the same seed.
"""
from __future__ import annotations

import logging
import time
from dataclasses import dataclass, field
import random
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import torch


@dataclass
class PGDAttackConfig:
    """src/attacks/pgd_attack.py:19-58 (same names and defaults; multi-GPU / AMP fields kept for compatibility)."""
    random_seed: int = 42
    device: str = "cuda"
    epsilon: float = 8.0 / 255.0
    alpha: float = 2.0 / 255.0
    num_steps: int = 10
    targeted: bool = False
    clip_min: float = 0.0
    clip_max: float = 1.0
    momentum: float = 0.9
    use_momentum: bool = True
    batch_size: int = 32
    enable_cache: bool = True
    cache_size: int = 1000
    enable_multi_gpu: bool = False
    gpu_ids: Optional[List[int]] = None
    batch_size_per_gpu: int = 8
    num_workers: int = 4
    gradient_accumulation_steps: int = 1
    mixed_precision: bool = False
    pin_memory: bool = True
    gradient_clip_value: float = 0.0
    # batches of `batch_size` in flight at once (not in the reference): each on its own HIP stream and its own engine handle
    # (same weights), so that the launches of one batch fill the compute units the other's leave idle -- a batch of 32 images
    # is 132 output tiles in half of its GEMMs, on 256 compute units.  Every batch's arithmetic is what it is alone.
    concurrent_batches: int = 2


class PGDAttacker:
    def __init__(self, clip_model, config: Optional[PGDAttackConfig] = None):
        self.config = config or PGDAttackConfig()
        self.clip_model = clip_model
        self.engine = clip_model.engine
        self.device = clip_model.device
        self.device_ids = [self.device.index or 0]
        self._gen = torch.Generator().manual_seed(self.config.random_seed)
        self.cache: Optional[Dict[str, Any]] = {} if self.config.enable_cache else None
        self.reset_stats()

    # ---- helpers ---------------------------------------------------------------------------------
    def _to_batch(self, images) -> torch.Tensor:
        if isinstance(images, torch.Tensor):
            x = images if images.dim() == 4 else images.unsqueeze(0)
        else:
            x = torch.stack([im if isinstance(im, torch.Tensor) else self.clip_model.preprocess(im) for im in images])
        return x.to(self.device, torch.float32).contiguous()

    def _text_unit(self, texts: Sequence[str]) -> torch.Tensor:
        toks = self.clip_model.tokenize(list(texts))
        return self.clip_model.encode_tokens(toks, True)               # unit rows, on the device (:424-425)

    def _random_start(self, clean: torch.Tensor) -> torch.Tensor:
        c = self.config
        adv = clean.clone()
        if c.num_steps > 1:                                                                         # :437-442
            noise = (torch.rand(clean.shape, generator=self._gen) * 2 - 1) * c.epsilon
            adv = torch.clamp(adv + noise.to(clean.device), c.clip_min, c.clip_max)
        return adv.contiguous()

    def _engine_pool(self, n: int):
        """`n` engines (the model's own first) and as many side streams; the extra handles reference the same tower
        weights (uploaded once more: 0.6 GB for ViT-L/14) and own their workspaces / kept activations."""
        if not hasattr(self, "_pool"):
            self._pool, self._streams = [self.engine], []
        while len(self._pool) < n:
            from .engine import TVCEngine
            self._pool.append(TVCEngine(self.engine.arch, self.engine._w_host[0], None, device=str(self.device)))
        while len(self._streams) < n:
            self._streams.append(torch.cuda.Stream(self.device))
        return self._pool[:n], self._streams[:n]

    def close(self) -> None:
        """Release the extra engine handles of `concurrent_batches` (the model's own engine is the model's to close)."""
        for e in getattr(self, "_pool", [])[1:]:
            e.close()
        if hasattr(self, "_pool"):
            self._pool = self._pool[:1]

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 -- interpreter shutdown
            pass

    def _steps_concurrent(self, jobs: Sequence[Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]]) -> List[torch.Tensor]:
        """`_steps` for several batches at once, batch j on stream j / engine j; the random starts are drawn in batch
        order, so the outputs equal the sequential ones bit for bit."""
        if len(jobs) == 1:
            return [self._steps(*jobs[0])]
        c = self.config
        engines, streams = self._engine_pool(len(jobs))
        state = []
        for clean, text_f, target_f in jobs:
            adv = self._random_start(clean)
            mom = torch.zeros_like(adv) if c.use_momentum else None
            targeted = bool(c.targeted and target_f is not None)
            g_out = ((-target_f if targeted else text_f) / clean.shape[0]).contiguous()
            state.append((clean, adv, mom, g_out))
        main = torch.cuda.current_stream()
        for st in streams:
            st.wait_stream(main)
        for _ in range(c.num_steps):
            for (clean, adv, mom, g_out), eng, st in zip(state, engines, streams):
                with torch.cuda.stream(st):
                    eng.encode_image_grad(adv, True)
                    grad = eng.encode_image_backward(g_out)
                    eng.pgd_step(adv, clean, grad, mom, c.epsilon, c.alpha, c.momentum, c.clip_min, c.clip_max, c.targeted)
        for st in streams:
            main.wait_stream(st)
        return [adv for _, adv, _, _ in state]

    def _steps(self, clean: torch.Tensor, text_f: torch.Tensor, target_f: Optional[torch.Tensor],
               history: Optional[Dict[str, list]] = None) -> torch.Tensor:
        """The iteration of :452-521 / :238-302 on a device batch."""
        c = self.config
        B = clean.shape[0]
        adv = self._random_start(clean)
        mom = torch.zeros_like(adv) if c.use_momentum else None
        targeted = bool(c.targeted and target_f is not None)
        # d(loss)/d(unit embedding): loss = mean_b cos(f_b, text_b)  or  -mean_b cos(f_b, target_b)
        g_out = ((-target_f if targeted else text_f) / B).contiguous()
        for _ in range(c.num_steps):
            f = self.engine.encode_image_grad(adv, True)
            grad = self.engine.encode_image_backward(g_out)
            if history is not None:
                history["loss_history"].append(float((f * g_out).sum()))
            self.engine.pgd_step(adv, clean, grad, mom, c.epsilon, c.alpha, c.momentum, c.clip_min, c.clip_max, c.targeted)
            if history is not None:                                                                  # :293-301
                cur = self.engine.encode_image(adv, True)
                history["similarity_history"].append(float((cur * text_f).sum(-1).mean()))
                history["iterations"] += 1
        return adv

    # ---- single image (:144-340) --------------------------------------------------------------------
    def attack(self, image, text: str, target_text: Optional[str] = None) -> Dict[str, Any]:
        key = self._get_cache_key(image, text, target_text)
        if self.cache and key in self.cache:
            return self.cache[key]
        clean = self._to_batch(image if isinstance(image, torch.Tensor) else [image])
        text_f = self._text_unit([text])
        target_f = self._text_unit([target_text]) if target_text else None
        info = {"iterations": 0, "loss_history": [], "similarity_history": [], "perturbation_norm": 0.0}
        adv = self._steps(clean, text_f, target_f, info)
        info["perturbation_norm"] = float(torch.norm(adv - clean))                                   # :307-308
        success = self._evaluate_attack_success(adv, text_f, target_f)
        self._update_stats(info, success)
        result = {"adversarial_image": adv, "original_image": clean, "perturbation": adv - clean, "success": success,
                  "attack_info": info, "config": self.config}
        if self.cache is not None and len(self.cache) < self.config.cache_size:
            self.cache[key] = result
        return result

    def _pgd_attack(self, image: torch.Tensor, text_features: torch.Tensor,
                    target_features: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Dict]:
        info = {"iterations": 0, "loss_history": [], "similarity_history": [], "perturbation_norm": 0.0}
        unit = lambda t: None if t is None else torch.nn.functional.normalize(t.to(self.device, torch.float32), dim=-1)
        clean = self._to_batch(image)
        adv = self._steps(clean, unit(text_features), unit(target_features), info)
        info["perturbation_norm"] = float(torch.norm(adv - clean))
        return adv, info

    def _evaluate_attack_success(self, adversarial_image: torch.Tensor, text_features: torch.Tensor,
                                 target_features: Optional[torch.Tensor] = None) -> bool:
        """:312-340."""
        f = self.engine.encode_image(adversarial_image, True)
        cos = lambda t: float(torch.nn.functional.cosine_similarity(f, t.to(f.device)).mean())
        if self.config.targeted and target_features is not None:
            return cos(target_features) > cos(text_features)
        return cos(text_features) < 0.5

    # ---- batches (:342-563) ----------------------------------------------------------------------------
    def batch_attack(self, images, texts: List[str], target_texts: Optional[List[str]] = None) -> List[Dict[str, Any]]:
        t0 = time.time()
        x = self._to_batch(images)
        out: List[Dict[str, Any]] = []
        bs = self.config.batch_size
        for i in range(0, x.shape[0], bs):
            out.extend(self._batch_pgd_attack(x[i:i + bs], texts[i:i + bs], target_texts[i:i + bs] if target_texts else None))
        ok = sum(1 for r in out if r["success"])
        self.attack_stats["total_attacks"] += len(out)
        self.attack_stats["successful_attacks"] += ok
        logging.info("batch PGD: %d/%d succeeded in %.2f s", ok, len(out), time.time() - t0)
        return out

    def perturb(self, images: torch.Tensor, texts: Sequence[str], target_texts: Optional[Sequence[str]] = None) -> torch.Tensor:
        """Device tensor in, adversarial device tensor out (no per-sample dictionaries): what an evaluation run
        feeds straight into ``pipeline.detect``."""
        x = self._to_batch(images)
        bs = self.config.batch_size
        nc = max(1, int(self.config.concurrent_batches))
        outs = []
        starts = list(range(0, x.shape[0], bs))
        for g0 in range(0, len(starts), nc):
            jobs = []
            for i in starts[g0:g0 + nc]:
                tf = self._text_unit(texts[i:i + bs])
                gf = self._text_unit(target_texts[i:i + bs]) if target_texts else None
                jobs.append((x[i:i + bs].contiguous(), tf, gf))
            outs.extend(self._steps_concurrent(jobs))
        return torch.cat(outs)

    def _batch_pgd_attack(self, batch_images: torch.Tensor, batch_texts: List[str],
                          batch_targets: Optional[List[str]] = None) -> List[Dict[str, Any]]:
        c = self.config
        clean = self._to_batch(batch_images)
        text_f = self._text_unit(batch_texts)
        target_f = self._text_unit(batch_targets) if batch_targets else None
        adv = self._steps(clean, text_f, target_f)
        final = self.engine.encode_image(adv, True)                                                   # :524-526
        targeted = bool(c.targeted and target_f is not None)
        sims = (final * (target_f if targeted else text_f)).sum(-1).cpu().tolist()
        pert = adv - clean
        linf = pert.abs().flatten(1).max(dim=1).values.cpu().tolist()
        adv_c, clean_c, pert_c = adv.cpu(), clean.cpu(), pert.cpu()
        results = []
        for i in range(clean.shape[0]):
            success = sims[i] > 0.5 if targeted else sims[i] < 0.3                                    # :530-540
            results.append({"adversarial_image": adv_c[i], "original_image": clean_c[i], "perturbation": pert_c[i],
                            "success": success,
                            "attack_info": {"iterations": c.num_steps, "final_similarity": sims[i],
                                            "perturbation_norm": linf[i], "targeted": c.targeted},
                            "config": c})
        return results

    # ---- cache / stats (:565-640) -----------------------------------------------------------------------
    def _get_cache_key(self, image, text: str, target_text: Optional[str] = None) -> str:
        if isinstance(image, torch.Tensor):
            ih = hash(f"{tuple(image.shape)}_{image.device}_{float(image.sum())}")
        else:
            ih = hash(str(image))
        return f"pgd_{ih}_{hash(text)}_{hash(target_text) if target_text else 0}_{self.config.epsilon}_{self.config.num_steps}"

    def _update_stats(self, attack_info: Dict, success: bool) -> None:
        s = self.attack_stats
        s["total_attacks"] += 1
        s["successful_attacks"] += int(bool(success))
        n = s["total_attacks"]
        s["average_perturbation"] = (s["average_perturbation"] * (n - 1) + attack_info["perturbation_norm"]) / n
        s["average_iterations"] = (s["average_iterations"] * (n - 1) + attack_info["iterations"]) / n

    def get_attack_stats(self) -> Dict[str, Any]:
        s = dict(self.attack_stats)
        s["success_rate"] = s["successful_attacks"] / s["total_attacks"] if s["total_attacks"] else 0.0
        return s

    def reset_stats(self) -> None:
        self.attack_stats = {"total_attacks": 0, "successful_attacks": 0, "average_perturbation": 0.0,
                             "average_iterations": 0.0}

    def clear_cache(self) -> None:
        if self.cache:
            self.cache.clear()


def create_pgd_attacker(clip_model, config: Optional[PGDAttackConfig] = None) -> PGDAttacker:
    return PGDAttacker(clip_model, config or PGDAttackConfig())


# ================================================================================================
# Hubness attack (src/attacks/hubness_attack.py): push ONE image towards many text queries at once.
#   loss = -mean_q cos(f(x), t_q) = -(f_hat . mean_q t_hat_q)            (:656-676)
# so the embedding gradient is the constant vector -mean_q t_hat_q (/ B in a batch): again no autograd
# graph, and the "bank" of 10-200 queries collapses to one row per image before the loop starts.
# ================================================================================================
@dataclass
class HubnessAttackConfig:
    """src/attacks/hubness_attack.py:40-99 (same names and defaults)."""
    clip_model: str = "openai/clip-vit-base-patch32"
    device: str = "cuda"
    epsilon: float = 16.0 / 255.0
    num_iterations: int = 500
    step_size: float = 0.02
    k_neighbors: int = 10
    num_target_queries: int = 100
    hubness_weight: float = 1.0
    success_threshold: float = 0.84
    learning_rate: float = 0.02
    momentum: float = 0.9
    weight_decay: float = 1e-4
    attack_mode: str = "universal"
    target_concepts: List[str] = field(default_factory=list)
    norm_constraint: str = "linf"
    clamp_min: float = 0.0
    clamp_max: float = 1.0
    random_start: bool = True
    random_seed: int = 42
    enable_multi_gpu: bool = True
    gpu_ids: Optional[List[int]] = None
    batch_size: int = 64
    batch_size_per_gpu: int = 16
    num_workers: int = 4
    gradient_accumulation_steps: int = 1
    mixed_precision: bool = True
    pin_memory: bool = True
    enable_cache: bool = True
    cache_size: int = 1000
    dataset_size: int = 25000
    query_pool_size: int = 1000

    def __post_init__(self):
        if self.target_concepts is None:
            self.target_concepts = []

    @classmethod
    def from_dict(cls, config_dict: Dict[str, Any]) -> "HubnessAttackConfig":
        """:100-128: reads ``config_dict['attacks']['hubness']``."""
        h = config_dict.get("attacks", {}).get("hubness", {})
        keys = ("clip_model", "epsilon", "num_iterations", "step_size", "k_neighbors", "num_target_queries",
                "hubness_weight", "success_threshold", "learning_rate", "momentum", "weight_decay", "attack_mode",
                "target_concepts", "norm_constraint", "clamp_min", "clamp_max", "random_start", "random_seed",
                "enable_cache", "cache_size", "dataset_size", "query_pool_size")
        return cls(**{k: h[k] for k in keys if k in h})


COMMON_QUERIES = ("a photo of a cat", "a dog playing", "a beautiful landscape", "a person walking", "a car on the road",
                  "a bird flying", "a flower in the garden", "a building in the city", "food on a plate",
                  "a sunset over the ocean")                                                              # :769-774


class HubnessAttack:
    """Mirror of ``HubnessAttack`` (:131-787).  ``clip_model`` may be injected (the reference builds its own from
    ``config.clip_model``, :435-462; done here too when none is given).  ``norm_constraint`` 'linf' (default) or 'l2'
    (the batch core's L2 step and projection, :378-386, as one fused kernel: ``tvc_l2_step``)."""

    def __init__(self, config: Optional[HubnessAttackConfig] = None, clip_model=None):
        self.config = config or HubnessAttackConfig()
        if self.config.norm_constraint not in ("linf", "l2"):
            raise ValueError(f"norm_constraint must be 'linf' or 'l2' (got {self.config.norm_constraint!r})")
        if clip_model is None:
            from .clip import CLIPConfig, CLIPModel
            clip_model = CLIPModel(CLIPConfig(model_name=self.config.clip_model, device=self.config.device))
        self.clip_model = clip_model
        self.engine = clip_model.engine
        self.device = clip_model.device
        self.device_ids = [self.device.index or 0]
        self.attack_stats = {"total_attacks": 0, "successful_attacks": 0, "average_iterations": 0,
                             "average_hubness_score": 0.0, "average_attack_time": 0.0}
        self.cache: Optional[Dict[str, Any]] = {} if self.config.enable_cache else None
        self._rng = random.Random(self.config.random_seed)
        self._gen = torch.Generator().manual_seed(self.config.random_seed)
        self.text_features = None
        self.image_features = None

    # ---- reference database (:189-204) ----------------------------------------------------------------
    def build_reference_database(self, reference_images, reference_texts: List[str]) -> None:
        self.image_features = self.clip_model.encode_image(reference_images)
        self.text_features = self.clip_model.encode_text(reference_texts)

    def _generate_random_queries(self, n: Optional[int] = None) -> List[str]:
        """:766-777 (the batch core passes a count, :289, which the reference's signature does not take)."""
        n = min(n or self.config.num_target_queries, len(COMMON_QUERIES))
        return self._rng.sample(list(COMMON_QUERIES), n)

    def compute_hubness(self, image_features: torch.Tensor, text_features: torch.Tensor, k: int = 10) -> float:
        """:464-498: share of the text queries whose top-1 image is image 0.  One all-pairs cosine launch."""
        from .metrics import SimilarityCalculator
        img = image_features.reshape(-1, image_features.shape[-1])
        C = SimilarityCalculator.batch_cosine_similarity(text_features.reshape(-1, img.shape[-1]), img, engine=self.engine)
        return float((C.argmax(axis=1) == 0).mean())

    # ---- the loop (:549-654) on a device batch -----------------------------------------------------------
    def _optimise(self, clean: torch.Tensor, q_mean: torch.Tensor):
        """clean [B,3,S,S], q_mean [B, D] = mean of each image's unit query rows -> (best image, best loss [B])."""
        c = self.config
        B = clean.shape[0]
        if c.random_start:                                       # :573-576: perturbation ~ U(-eps, eps)
            noise = (torch.rand(clean.shape, generator=self._gen) * 2 - 1) * c.epsilon
            adv = torch.clamp(clean + noise.to(clean.device), c.clamp_min, c.clamp_max).contiguous()
        else:
            adv = torch.clamp(clean, c.clamp_min, c.clamp_max).contiguous()
        g_out = (-q_mean / B).contiguous()                       # d(mean_b loss_b) / d(unit embedding)
        best_loss = torch.full((B,), float("inf"), device=clean.device)
        best = adv.clone()
        for _ in range(c.num_iterations):
            f = self.engine.encode_image_grad(adv, True)
            loss = -(f * q_mean).sum(-1)                         # per image, on the device: no host sync in the loop
            better = loss < best_loss                            # :626-628 (the image the loss was computed on)
            best_loss = torch.where(better, loss, best_loss)
            best = torch.where(better.view(-1, 1, 1, 1), adv, best)
            grad = self.engine.encode_image_backward(g_out)
            if c.norm_constraint == "l2":
                # :378-386: unit-L2 gradient step, projection onto the eps L2 ball, clamp (descent, as the L-inf form)
                self.engine.l2_step(adv, clean, grad, c.epsilon, c.step_size, c.clamp_min, c.clamp_max, True)
            else:
                # :617-637: p -= step * sign(grad); clamp to the eps ball; clamp the image  == one descent pgd_step
                self.engine.pgd_step(adv, clean, grad, None, c.epsilon, c.step_size, 0.0, c.clamp_min, c.clamp_max, True)
        return best, best_loss

    def _unit_queries(self, queries: Sequence[str]) -> torch.Tensor:
        return self.clip_model.encode_tokens(self.clip_model.tokenize(list(queries)), True)

    def _perform_attack(self, image_tensor: torch.Tensor, text_features: torch.Tensor, text_queries: List[str]) -> Dict[str, Any]:
        clean = image_tensor.to(self.device, torch.float32)
        clean = (clean if clean.dim() == 4 else clean.unsqueeze(0)).contiguous()
        tq = torch.nn.functional.normalize(text_features.to(self.device, torch.float32), dim=-1)
        best, best_loss = self._optimise(clean, tq.mean(0, keepdim=True))
        final = self.engine.encode_image(best, True)
        hub = self.compute_hubness(final.unsqueeze(0), tq, self.config.k_neighbors)        # :640-646 (one image: 1.0)
        pert = best - clean
        return {"adversarial_image": best.squeeze(0).cpu(), "original_image": clean.squeeze(0).cpu(),
                "perturbation": pert.squeeze(0).cpu(), "hubness_score": hub,
                "perturbation_norm": float(pert.abs().max()), "final_loss": float(best_loss[0]),
                "iterations": self.config.num_iterations, "success": hub > self.config.success_threshold,
                "text_queries": text_queries}

    def create_adversarial_hub(self, image, text_queries: List[str]) -> Dict[str, Any]:
        """:500-547."""
        t0 = time.time()
        key = self._generate_cache_key(image, text_queries)
        if self.cache is not None and key in self.cache:
            return self.cache[key]
        x = image if isinstance(image, torch.Tensor) else self.clip_model.preprocess(image)
        result = self._perform_attack(x, self._unit_queries(text_queries), text_queries)
        self._update_attack_stats(result, time.time() - t0)
        if self.cache is not None and len(self.cache) < self.config.cache_size:
            self.cache[key] = result
        return result

    def attack(self, image, text: Optional[str] = None) -> Dict[str, Any]:
        """:703-728."""
        return self.create_adversarial_hub(image, self._generate_random_queries() if text is None else [text])

    def attack_single(self, image, text: str) -> Dict[str, Any]:
        """:730-764 (the key names run_experiments.py expects; errors are reported, not raised, as there)."""
        try:
            r = self.attack(image, text)
            return {"success": r.get("success", False), "hubness": r.get("hubness_score", 0.0),
                    "similarity_change": r.get("perturbation_norm", 0.0), "iterations": r.get("iterations", 0),
                    "adversarial_image": r.get("adversarial_image"), "original_image": r.get("original_image"),
                    "perturbation": r.get("perturbation"), "final_loss": r.get("final_loss", float("inf"))}
        except Exception as e:                                   # noqa: BLE001 -- mirrors :757-764
            logging.error("hubness attack_single failed: %s", e)
            return {"success": False, "hubness": 0.0, "similarity_change": 0.0, "iterations": 0, "error": str(e)}

    def batch_attack(self, images, texts: List[str]) -> List[Dict[str, Any]]:
        """:206-267.  Every image gets its own random query set (:286-291)."""
        t0 = time.time()
        if isinstance(images, torch.Tensor):
            x = images
        else:
            x = torch.stack([im if isinstance(im, torch.Tensor) else self.clip_model.preprocess(im) for im in images])
        out: List[Dict[str, Any]] = []
        bs = self.config.batch_size
        for i in range(0, x.shape[0], bs):
            out.extend(self._batch_attack_core(x[i:i + bs].to(self.device, torch.float32).contiguous(), texts[i:i + bs]))
        n, ok = len(out), sum(1 for r in out if r["success"])
        s = self.attack_stats
        s["total_attacks"] += n
        s["successful_attacks"] += ok
        s["average_attack_time"] = (s["average_attack_time"] * (s["total_attacks"] - n) + (time.time() - t0)) / max(s["total_attacks"], 1)
        return out

    def _batch_attack_core(self, batch_images: torch.Tensor, batch_texts: List[str]) -> List[Dict[str, Any]]:
        """:269-424 with the single-image loop's DESCENT on the loss.  (The reference's batch core steps
        ``+ step_size * grad.sign()`` on the same negative-similarity loss, :357, i.e. away from the queries --
        against its own comment and its single-image loop, :617; and it cannot run at all: it calls
        ``_generate_random_queries`` with an argument the method does not take, :289.)"""
        B = batch_images.shape[0]
        queries = [self._generate_random_queries(self.config.num_target_queries) for _ in range(B)]
        qf = self._unit_queries([q for qs in queries for q in qs])
        nq = len(queries[0])
        q_mean = qf.view(B, nq, -1).mean(1)
        best, _ = self._optimise(batch_images, q_mean)
        final = self.engine.encode_image(best, True)
        scores = (final * q_mean).sum(-1).cpu().tolist()                                   # mean similarity (:392-393)
        linf = (best - batch_images).abs().flatten(1).max(dim=1).values.cpu().tolist()
        best_c, clean_c = best.cpu(), batch_images.cpu()
        return [{"success": scores[i] > self.config.success_threshold, "hubness_score": scores[i],
                 "perturbation_strength": linf[i], "adversarial_image": best_c[i], "original_image": clean_c[i],
                 "target_queries": queries[i], "iterations": self.config.num_iterations} for i in range(B)]

    # ---- cache / stats (:678-701, 779-786) ---------------------------------------------------------------
    def _generate_cache_key(self, image, text_queries: List[str]) -> str:
        import hashlib
        ih = f"{tuple(image.shape)}_{float(image.sum())}" if isinstance(image, torch.Tensor) else str(image)
        return hashlib.md5((ih + "|".join(text_queries) + f"{self.config.epsilon}_{self.config.num_iterations}").encode()).hexdigest()

    def _update_attack_stats(self, result: Dict[str, Any], attack_time: float) -> None:
        s = self.attack_stats
        s["total_attacks"] += 1
        s["successful_attacks"] += int(bool(result["success"]))
        n = s["total_attacks"]
        s["average_iterations"] = (s["average_iterations"] * (n - 1) + result["iterations"]) / n
        s["average_hubness_score"] = (s["average_hubness_score"] * (n - 1) + result["hubness_score"]) / n
        s["average_attack_time"] = (s["average_attack_time"] * (n - 1) + attack_time) / n

    def get_attack_stats(self) -> Dict[str, Any]:
        s = dict(self.attack_stats)
        s["success_rate"] = s["successful_attacks"] / s["total_attacks"] if s["total_attacks"] else 0.0
        return s


class HubnessAttackPresets:
    """:789-838."""
    @staticmethod
    def weak_attack() -> HubnessAttackConfig:
        return HubnessAttackConfig(epsilon=8.0 / 255.0, num_iterations=100, learning_rate=0.01, k_neighbors=5, num_target_queries=50)

    @staticmethod
    def strong_attack() -> HubnessAttackConfig:
        return HubnessAttackConfig(epsilon=32.0 / 255.0, num_iterations=1000, learning_rate=0.05, k_neighbors=20, num_target_queries=200)

    @staticmethod
    def targeted_attack(target_concepts: List[str]) -> HubnessAttackConfig:
        return HubnessAttackConfig(attack_mode="targeted", target_concepts=target_concepts, epsilon=16.0 / 255.0,
                                   num_iterations=500, learning_rate=0.02, k_neighbors=10, num_target_queries=100)

    @staticmethod
    def paper_standard() -> HubnessAttackConfig:
        return HubnessAttackConfig(epsilon=16.0 / 255.0, num_iterations=500, learning_rate=0.02, k_neighbors=10,
                                   num_target_queries=100, dataset_size=25000, query_pool_size=1000)


def create_hubness_attacker(config: Optional[HubnessAttackConfig] = None, clip_model=None) -> HubnessAttack:
    return HubnessAttack(config or HubnessAttackConfig(), clip_model=clip_model)
