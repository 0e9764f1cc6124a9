"""PGD attack inner loop on the GPU.

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
from dataclasses import dataclass
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
        outs = []
        for i in range(0, x.shape[0], bs):
            tf = self._text_unit(texts[i:i + bs])
            gf = self._text_unit(target_texts[i:i + bs]) if target_texts else None
            outs.append(self._steps(x[i:i + bs].contiguous(), tf, gf))
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
