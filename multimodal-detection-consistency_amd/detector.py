"""Detectors of both reference implementations over the batched HIP path.

* ``AdversarialDetector`` / ``DetectorConfig`` -- ``src/detector.py:172-904``
  ("src" polarity: high score => adversarial, ``:399``).
* ``MultiModalDefenseDetector`` / ``DetectionConfig`` / ``ConsistencyChecker`` --
  ``experiments/defenses/detector.py``, ``consistency_checker.py`` ("exp"
  polarity: low score => adversarial, ``consistency_checker.py:93``).

The reference encodes the same image N+1 times and synchronises per variant
(``src/detector.py:461-471``); here a batch of B queries is ONE image encode,
ONE text encode of B*(N+1) rows, one bank search and one consistency launch.
The only per-query host work left is what the reference itself keeps on the
host: the stateful adaptive threshold / confidence of ``ConsistencyChecker``.
"""
from __future__ import annotations

import logging
import math
import threading
import time
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from .clip import CLIPConfig, CLIPModel
from .engine import ConsistencyConfig
from .variants import as_generator, batch_variants

logger = logging.getLogger(__name__)

SRC_WEIGHTS = {"text_variants": 0.4, "sd_reference": 0.4, "consistency": 0.2}   # src/detector.py:666-670


# ------------------------------------------------------------------ records
def unpack_records(rec: torch.Tensor, n_variants: int) -> Dict[str, np.ndarray]:
    """Device record tensor (layout: include/tvc.h) -> host arrays (ONE D2H copy)."""
    r = rec.detach().cpu().numpy()
    N = n_variants
    M = _lib.TVC_REC_MAXREF
    return {
        "original_similarity": r[:, 0].astype(np.float64), "variant_mean": r[:, 1].astype(np.float64),
        "variant_std": r[:, 2].astype(np.float64), "score_text_variants": r[:, 3].astype(np.float64),
        "score_consistency": r[:, 4].astype(np.float64), "score_src": r[:, 5].astype(np.float64),
        "retrieval_consistency": r[:, 6].astype(np.float64), "retrieval_std": r[:, 7].astype(np.float64),
        "n_references": r[:, 8].astype(np.int64), "cross_modal_variance": r[:, 9].astype(np.float64),
        "overall_exp": r[:, 10].astype(np.float64),
        "variant_similarities": r[:, 12:12 + N].astype(np.float64),
        "reference_indices": np.ascontiguousarray(r[:, 12 + N:12 + N + M]).view(np.int32).astype(np.int64),
        "reference_similarities": r[:, 12 + N + M:12 + N + 2 * M].astype(np.float64),
    }


# ------------------------------------------------------------- src polarity
@dataclass
class DetectorConfig:
    """src/detector.py:172-212 (same field names and defaults)."""
    clip_model: str = "ViT-B/32"
    device: str = "cuda"
    detection_methods: Optional[List[str]] = None
    use_text_variants: bool = True
    num_text_variants: int = 5
    text_similarity_threshold: float = 0.85
    use_sd_reference: bool = True
    num_reference_images: int = 3
    reference_similarity_threshold: float = 0.75
    consistency_threshold: float = 0.8
    consistency_weight: float = 0.5
    detection_threshold: float = 0.5
    adaptive_threshold: bool = True
    threshold_percentile: float = 95.0
    score_aggregation: str = "weighted_mean"
    enable_cache: bool = False            # reference default True; hashing device images forces a D2H copy
    cache_size: int = 1000
    batch_size: int = 32
    strict: bool = True                   # False: swallow errors into neutral results like src/detector.py:428-439

    def __post_init__(self):
        if self.detection_methods is None:
            self.detection_methods = ["text_variants", "sd_reference", "consistency"]


def aggregate_scores(scores: Dict[str, float], method: str = "weighted_mean") -> float:
    """src/detector.py:643-682."""
    if not scores:
        return 0.0
    vals = list(scores.values())
    if method == "mean":
        return float(np.mean(vals))
    if method == "max":
        return float(np.max(vals))
    if method == "min":
        return float(np.min(vals))
    if method == "weighted_mean":
        ws = tw = 0.0
        for name, s in scores.items():
            w = SRC_WEIGHTS.get(name, 1.0)
            ws += s * w
            tw += w
        return ws / tw if tw > 0 else 0.0
    return float(np.mean(vals))


class AdversarialDetector:
    """Drop-in for ``src/detector.py:217`` (``detect_adversarial`` / ``batch_detect``)
    plus the ``detect`` name the efficiency harness calls
    (``experiments/run_experiments.py:3253``)."""

    def __init__(self, config: Optional[DetectorConfig] = None, clip_model: Optional[CLIPModel] = None,
                 text_augmenter=None, sd_generator=None):
        self.config = config or DetectorConfig()
        self.clip_model = clip_model
        self.text_augmenter = text_augmenter
        self.sd_generator = sd_generator      # object with generate_reference_images(text, num_images) or None
        self.detection_cache: Dict[str, Dict] = {}
        self.detection_stats = {"total_detections": 0, "cache_hits": 0, "detection_time": 0.0,
                                "method_usage": {m: 0 for m in self.config.detection_methods}}
        self._lock = threading.Lock()

    # -- lazily built components (src/detector.py:253-343) ----------------
    def _get_clip_model(self) -> CLIPModel:
        if self.clip_model is None:
            self.clip_model = CLIPModel(CLIPConfig(model_name=self.config.clip_model, device=self.config.device))
        return self.clip_model

    def _variants(self, text: str) -> List[str]:
        if not self.config.use_text_variants:
            return []
        return as_generator(self.text_augmenter, self.config.num_text_variants)(text)

    # -- batched core --------------------------------------------------------
    def _src_cfg(self, text_variants_on: bool) -> ConsistencyConfig:
        """Weights of the device-side weighted mean (src/detector.py:666-670); weight 0 = method off."""
        return ConsistencyConfig(w_text_variants=SRC_WEIGHTS["text_variants"] if text_variants_on else 0.0,
                                 w_consistency=SRC_WEIGHTS["consistency"])

    def detect_tokens(self, images: torch.Tensor, tokens: torch.Tensor) -> Dict[str, np.ndarray]:
        """images [B,3,S,S], tokens int [B, N+1, ctx] (row 0 = original text) ->
        host arrays of the record fields (see ``unpack_records``) plus
        ``aggregated_score`` / ``is_adversarial`` for the methods
        {text_variants, consistency}."""
        clip = self._get_clip_model()
        B, N1, ctx = tokens.shape
        fi = clip.engine.encode_image(images.to(clip.device, torch.float32), True)
        ft = clip.engine.encode_text(tokens.reshape(B * N1, ctx).to(clip.device, torch.int32), True, group=N1)
        rec = clip.engine.consistency(fi, ft.view(B, N1, -1), self._src_cfg(self.config.use_text_variants))
        out = unpack_records(rec, N1 - 1)
        out["aggregated_score"] = out["score_src"]
        out["is_adversarial"] = out["score_src"] > self.config.detection_threshold      # src/detector.py:399
        return out

    def batch_detect(self, images, texts: Sequence[str], methods: Optional[List[str]] = None,
                     variants: Optional[Sequence[Sequence[str]]] = None,
                     reference_images: Optional[Sequence[Sequence[Any]]] = None,
                     keep_features: Optional[Dict[str, torch.Tensor]] = None,
                     image_rows=None) -> List[Dict[str, Any]]:
        """src/detector.py:711-734, truly batched.  ``keep_features`` (a dict) receives the device rows
        ``image`` [n, D] and ``text`` [n, D] (the ORIGINAL texts) so that a caller needing them again
        (the pipeline's retrieval step) does not encode the same texts twice.  Queries are grouped by their variant count so
        every group is one launch; ALL device work (image tower, text tower, reference-image tower,
        consistency kernels) is enqueued before the first device-to-host copy, and the host
        tokenises while the GPU runs the image tower.  ``image_rows``: the ``join`` of a ``CLIPModel.encode_image_beside``
        call the caller has ALREADY enqueued for these images (the pipeline starts the image tower before it generates the
        text variants on the host); ``images`` is then not encoded again."""
        methods = methods or self.config.detection_methods
        t0 = time.time()
        clip = self._get_clip_model()
        eng = clip.engine
        if isinstance(images, torch.Tensor) and images.dim() == 3:
            images = images.unsqueeze(0)
        n = len(texts)
        if image_rows is None:
            x, _ = clip._images_to_device(images if isinstance(images, torch.Tensor) else list(images))
            if x.shape[0] != n:
                raise ValueError("number of images and texts differ")
            join_fi = clip.encode_image_beside(x, True)     # enqueued first, on the side stream: beside the host work AND the text tower
        else:
            join_fi = image_rows
        fi = None
        # a requested method is scored whenever its component exists (src/detector.py:375,382), also when
        # the component yields nothing for a query (0.0 + 'error', :457-458, :524-525)
        tv_on = "text_variants" in methods and self.config.use_text_variants
        sd_on = "sd_reference" in methods and (reference_images is not None or self.sd_generator is not None)
        if variants is None:
            variants = batch_variants(self.text_augmenter, self.config.num_text_variants, texts) if tv_on \
                else [[] for _ in texts]
        groups: Dict[int, List[int]] = {}
        for i, v in enumerate(variants):
            groups.setdefault(len(v), []).append(i)
        cfg = self._src_cfg(tv_on)
        pending = []                                        # (N, ids, device records)
        for N, ids in groups.items():
            flat: List[str] = []
            for i in ids:
                flat.append(texts[i])
                flat.extend(variants[i])
            ft = clip.encode_tokens(clip.tokenize(flat), True, group=N + 1).view(len(ids), N + 1, -1)
            if fi is None:
                fi = join_fi()
            sel = fi if len(ids) == n else fi[torch.as_tensor(ids, device=fi.device)].contiguous()
            pending.append((N, ids, eng.consistency(sel, ft, cfg)))
            if keep_features is not None:
                if "text" not in keep_features:
                    keep_features["text"] = torch.empty((n, ft.shape[-1]), dtype=ft.dtype, device=ft.device)
                    keep_features["image"] = fi
                keep_features["text"][torch.as_tensor(ids, device=ft.device)] = ft[:, 0]
        # SD-reference method (src/detector.py:503-557): references from the caller, or from the generator
        # (sd_ref.SDReferenceGenerator over sd_model.StableDiffusionModel: tvc_sd_generate).  ONE encode of all
        # reference images, one consistency launch per distinct count (cos(image, ref_j) = record words 0, 12..).
        sd_pending, sd_counts = [], [0] * n
        sd_errors: Dict[int, str] = {}                      # query -> message: the method scores 0.0 + 'error' (:555-557)
        if fi is None:
            fi = join_fi()
        if sd_on:
            fr = None
            if reference_images is None and hasattr(self.sd_generator, "reference_features"):
                # a generator of this package: every prompt x seed of the batch in the SAME UNet launches, references
                # preprocessed on the device and encoded in one image-tower launch (sd_ref.SDReferenceGenerator).
                # A failure of the generation (no model, out of memory, ...) is the sd_reference method's failure only:
                # src/detector.py:555-557 returns 0.0 + {'error'} for it and the other methods carry on
                try:
                    if getattr(self.sd_generator, "clip_model", None) is None:
                        self.sd_generator.clip_model = clip
                    fr, sd_counts = self.sd_generator.reference_features(list(texts), self.config.num_reference_images)
                    sd_counts = list(sd_counts)
                    if len(sd_counts) != n or sum(sd_counts) != (0 if fr is None else fr.shape[0]):
                        raise RuntimeError("reference_features returned inconsistent counts")
                except Exception as e:                      # noqa: BLE001
                    logger.error("SD reference generation failed: %s", e)
                    fr, sd_counts = None, [0] * n
                    sd_errors = {i: str(e) for i in range(n)}
            else:
                per_q = []
                for i in range(n):
                    try:
                        if reference_images is not None:
                            refs = reference_images[i]
                        else:
                            refs = self.sd_generator.generate_reference_images(
                                texts[i], num_images=self.config.num_reference_images).get("images", [])
                    except Exception as e:                  # noqa: BLE001 -- :555-557, this query's sd method only
                        logger.error("SD reference generation failed for query %d: %s", i, e)
                        refs, sd_errors[i] = [], str(e)
                    per_q.append(list(refs) if refs is not None else [])
                sd_counts = [len(r) for r in per_q]
                all_refs = [im for r in per_q for im in r]
                if all_refs:
                    xr, _ = clip._images_to_device(all_refs)
                    fr = eng.encode_image(xr, True)
            if fr is not None and fr.shape[0]:
                offs = np.concatenate([[0], np.cumsum(sd_counts)])
                by_count: Dict[int, List[int]] = {}
                for i, c in enumerate(sd_counts):
                    if c:
                        by_count.setdefault(c, []).append(i)
                for J, ids in by_count.items():
                    rows = torch.as_tensor(np.concatenate([np.arange(offs[i], offs[i] + J) for i in ids]), device=fr.device)
                    qsel = torch.as_tensor(ids, device=fi.device)
                    sd_pending.append((J, ids, eng.consistency(fi[qsel].contiguous(), fr[rows].view(len(ids), J, -1), cfg)))

        # ---- device -> host, then plain-Python result construction (lists, no per-field numpy scalar)
        scores: List[Dict[str, float]] = [dict() for _ in range(n)]
        details: List[Dict[str, Any]] = [dict() for _ in range(n)]
        cons_rows: List[Any] = [None] * n
        for N, ids, rec_dev in pending:
            r = rec_dev.cpu().numpy().astype(np.float64)
            s0, mean, sd, tv, cs = (r[:, c].tolist() for c in range(5))
            sv = r[:, 12:12 + N].tolist()
            for j, i in enumerate(ids):
                if tv_on:
                    if N > 0:
                        scores[i]["text_variants"] = tv[j]
                        details[i]["text_variants"] = {
                            "original_similarity": s0[j], "variant_similarities": sv[j],
                            "mean_variant_similarity": mean[j], "std_variant_similarity": sd[j],
                            "consistency_score": 1.0 - abs(s0[j] - mean[j]), "variability_score": 1.0 - sd[j],
                            "num_variants": N}
                    else:
                        scores[i]["text_variants"] = 0.0                                 # :457-458
                        details[i]["text_variants"] = {"error": "no text variants"}
                cons_rows[i] = (cs[j], s0[j])
        if sd_on:
            for i in range(n):
                if sd_counts[i] == 0:
                    scores[i]["sd_reference"] = 0.0                                      # :524-525
                    details[i]["sd_reference"] = {"error": sd_errors.get(i, "no reference images")}
            for J, ids, rec_dev in sd_pending:
                r = rec_dev.cpu().numpy().astype(np.float64)
                sims = np.concatenate([r[:, 0:1], r[:, 12:12 + J - 1]], axis=1)
                mean, mx, sd = sims.mean(1).tolist(), sims.max(1).tolist(), sims.std(1).tolist()
                sl = sims.tolist()
                for j, i in enumerate(ids):
                    scores[i]["sd_reference"] = 1.0 - mean[j]
                    details[i]["sd_reference"] = {"reference_similarities": sl[j], "mean_similarity": mean[j],
                                                  "max_similarity": mx[j], "std_similarity": sd[j], "num_references": J}
        if "consistency" in methods:
            for i in range(n):
                cs, s0 = cons_rows[i]
                scores[i]["consistency"] = cs
                details[i]["consistency"] = {"image_text_similarity": s0, "consistency_score": s0}
        dt = time.time() - t0
        per_q_dt = dt / max(n, 1)
        thr, how = self.config.detection_threshold, self.config.score_aggregation
        results = []
        for i in range(n):
            agg = aggregate_scores(scores[i], how)
            results.append({"is_adversarial": bool(agg > thr), "aggregated_score": float(agg),
                            "detection_scores": scores[i], "detection_details": details[i],
                            "detection_time": per_q_dt, "methods_used": methods, "threshold": thr})
        with self._lock:
            self.detection_stats["total_detections"] += n
            self.detection_stats["detection_time"] += dt
            for m in methods:
                if m in self.detection_stats["method_usage"]:
                    self.detection_stats["method_usage"][m] += n
        return results

    def detect_adversarial(self, image, text: str, methods: Optional[List[str]] = None, **kw) -> Dict[str, Any]:
        """src/detector.py:345-439."""
        methods = methods or self.config.detection_methods
        try:
            key = None
            if self.config.enable_cache:
                key = self._cache_key(image, text, methods)
                if key in self.detection_cache:
                    self.detection_stats["cache_hits"] += 1
                    return self.detection_cache[key]
            res = self.batch_detect([image] if not isinstance(image, torch.Tensor) else image, [text], methods, **{
                k: [v] for k, v in kw.items()})[0]
            if key is not None:
                if len(self.detection_cache) >= self.config.cache_size:
                    del self.detection_cache[next(iter(self.detection_cache))]
                self.detection_cache[key] = res
            return res
        except Exception as e:     # src/detector.py:428-439 swallows; default here is to raise
            if self.config.strict:
                raise
            return {"is_adversarial": False, "aggregated_score": 0.0, "detection_scores": {},
                    "detection_details": {}, "detection_time": 0.0, "methods_used": methods,
                    "threshold": self.config.detection_threshold, "error": str(e)}

    detect = detect_adversarial

    @staticmethod
    def _cache_key(image, text, methods) -> str:
        """src/detector.py:684-709."""
        if isinstance(image, torch.Tensor):
            ih = hash(image.detach().cpu().numpy().tobytes())
        else:
            ih = hash(np.array(image).tobytes())
        return f"{hash(text)}_{ih}_{hash(tuple(sorted(methods)))}"

    def get_detection_stats(self) -> Dict[str, Any]:
        return dict(self.detection_stats)


def create_adversarial_detector(config: Optional[DetectorConfig] = None, **kw) -> AdversarialDetector:
    """src/detector.py:892."""
    return AdversarialDetector(config, **kw)


# ------------------------------------------------------------- exp polarity
class ConsistencyChecker:
    """experiments/defenses/consistency_checker.py:31-272.  Host-side and stateful
    (``threshold_history``), as in the reference: decisions depend on call order."""

    _NAMES = ("original_similarity", "text_variant_consistency", "retrieval_consistency", "generative_consistency")

    def __init__(self, threshold: float = 0.5, adaptive_threshold: bool = True, voting_strategy: str = "weighted",
                 weights: Optional[Dict[str, float]] = None):
        self.base_threshold = threshold
        self.adaptive_threshold = adaptive_threshold
        self.voting_strategy = voting_strategy
        self.weights = weights or {n: 0.25 for n in self._NAMES}
        self.detection_history: List[Dict] = []
        self.threshold_history: List[float] = []

    def _compute_overall_score(self, s: Dict[str, float]) -> float:
        if self.voting_strategy == "simple":
            v = [s.get(n, 0) for n in self._NAMES if s.get(n, 0) > 0]
            return float(np.mean(v)) if v else 0.0
        if self.voting_strategy == "weighted":
            ws = tw = 0.0
            for n, w in self.weights.items():
                if n in s and s[n] > 0:
                    ws += s[n] * w
                    tw += w
            return ws / tw if tw != 0 else 0.0
        if self.voting_strategy == "adaptive":
            rel = {"original_similarity": 1.0,
                   "text_variant_consistency": 1.0 / (1.0 + s.get("text_variant_std", 1.0)),
                   "retrieval_consistency": 1.0 / (1.0 + s.get("retrieval_std", 1.0)),
                   "generative_consistency": 1.0 / (1.0 + s.get("generative_std", 1.0))}
            tot = sum(rel.values())
            if tot > 0:
                rel = {k: v / tot for k, v in rel.items()}
            ws = tw = 0.0
            for n in self._NAMES:
                v = s.get(n, 0)
                if v > 0:
                    ws += v * rel[n]
                    tw += rel[n]
            return ws / tw if tw != 0 else 0.0
        raise ValueError(f"unknown voting strategy: {self.voting_strategy}")

    # The three helpers below are plain-Python restatements of the reference's numpy expressions (np.mean / np.std /
    # np.clip on 2-10 element lists) with numpy's own summation order, so the decisions stay bit-for-bit the reference's
    # (tests/test_oracle_golden.py) at a tenth of the host time: 512 decisions cost 66 ms through numpy scalars.
    @staticmethod
    def _mean10(h: List[float]) -> float:
        """np.mean of a list of up to 10 floats: numpy sums fewer than 8 elements left to right, 8 or more as eight
        strided partial sums combined pairwise, then the tail left to right."""
        n = len(h)
        if n < 8:
            t = 0.0
            for x in h:
                t += x
            return t / n
        t = ((h[0] + h[1]) + (h[2] + h[3])) + ((h[4] + h[5]) + (h[6] + h[7]))
        for x in h[8:]:
            t += x
        return t / n

    def _get_adaptive_threshold(self, s: Dict[str, float]) -> float:
        thr = self.base_threshold
        if s.get("cross_modal_variance", 0) > 0.1:
            thr += 0.1
        if (0.0 + s.get("text_variant_std", 0) + s.get("retrieval_std", 0) + s.get("generative_std", 0)) / 3 > 0.2:
            thr += 0.05
        if len(self.threshold_history) > 10:
            thr = 0.7 * thr + 0.3 * self._mean10(self.threshold_history[-10:])
        return float(min(max(thr, 0.1), 0.9)) if thr == thr else float(thr)

    def _compute_confidence(self, overall: float, thr: float, s: Dict[str, float]) -> float:
        dist = abs(overall - thr) / thr
        v = [s.get(n, 0) for n in self._NAMES if s.get(n, 0) > 0]
        if len(v) > 1:
            n = len(v)
            t = 0.0
            for x in v:
                t += x
            m = t / n
            t = 0.0
            for x in v:
                t += (x - m) * (x - m)
            cons = 1.0 - math.sqrt(t / n)
        else:
            cons = 0.5
        var = 1.0 - min(s.get("cross_modal_variance", 0), 1.0)
        c = (0.0 + dist + cons + var) / 3
        return float(min(max(c, 0.0), 1.0)) if c == c else float(c)

    def make_decision(self, consistency_scores: Dict[str, float], return_details: bool = False) -> Dict[str, Any]:
        overall = self._compute_overall_score(consistency_scores)
        thr = self._get_adaptive_threshold(consistency_scores) if self.adaptive_threshold else self.base_threshold
        is_adv = overall < thr
        conf = self._compute_confidence(overall, thr, consistency_scores)
        self.detection_history.append({"overall_score": overall, "threshold": thr, "is_adversarial": is_adv,
                                       "confidence": conf})
        self.threshold_history.append(thr)
        res = {"is_adversarial": bool(is_adv), "confidence": conf, "overall_score": float(overall),
               "threshold": float(thr)}
        if return_details:
            res["details"] = {"individual_scores": consistency_scores, "overall_score": overall, "threshold": thr}
        return res

    def reset(self) -> None:
        self.detection_history.clear()
        self.threshold_history.clear()


@dataclass
class DetectionConfig:
    """experiments/defenses/detector.py:20-44."""
    use_text_variants: bool = True
    text_variant_count: int = 5
    use_retrieval_ref: bool = True
    retrieval_top_k: int = 10
    retrieval_weight: float = 0.3
    use_generative_ref: bool = True
    generation_count: int = 3
    generation_weight: float = 0.4
    consistency_threshold: float = 0.5
    adaptive_threshold: bool = True
    voting_strategy: str = "weighted"
    device: str = "cuda"
    debug_mode: bool = False
    # retrieval (experiments/defenses/retrieval_ref.py:20-32)
    reference_count: int = 5
    similarity_threshold: float = 0.3


class MultiModalDefenseDetector:
    """Drop-in for ``experiments/defenses/detector.py:46``: injected ``clip_model``,
    optional text-variant generator, bank of reference image features
    (``features.npy`` rows, already registered on the clip engine with
    ``set_reference_bank``)."""

    def __init__(self, clip_model: CLIPModel, qwen_model=None, sd_model=None,
                 config: Optional[DetectionConfig] = None, text_generator=None, retrieval_generator=None,
                 generative_generator=None):
        self.clip_model = clip_model
        self.config = config or DetectionConfig()
        self.text_variant_generator = text_generator if text_generator is not None else qwen_model
        self.retrieval_generator = retrieval_generator
        # object with generate_references(text) -> list of image tensors (experiments/defenses/generative_ref.py:71).
        # A diffusion model (sd_model.StableDiffusionModel, or anything with the reference's `generate`) is wrapped
        # in the GenerativeReferenceGenerator mirror, as experiments/defenses/detector.py:99-105 does.
        if generative_generator is None and sd_model is not None:
            if hasattr(sd_model, "generate_references"):
                generative_generator = sd_model
            elif hasattr(sd_model, "generate_batch") or hasattr(sd_model, "generate"):
                from .sd_ref import GenerativeConfig, GenerativeReferenceGenerator
                generative_generator = GenerativeReferenceGenerator(
                    sd_model, clip_model, GenerativeConfig(generation_count=self.config.generation_count))
        self.generative_generator = generative_generator
        # this detector's own bank slot on the (shared) engine; an injected RetrievalReferenceGenerator
        # brings its registered features.npy rows with it
        self.bank_name = getattr(retrieval_generator, "bank_name", None) or f"defense:{id(self):x}"
        self.consistency_checker = ConsistencyChecker(threshold=self.config.consistency_threshold,
                                                      adaptive_threshold=self.config.adaptive_threshold,
                                                      voting_strategy=self.config.voting_strategy)

    def __del__(self):          # a bank this detector registered itself goes back to the shared engine
        try:
            if getattr(self.retrieval_generator, "bank_name", None) != self.bank_name:
                self.clip_model.engine.release_bank(self.bank_name)
        except Exception:
            pass

    def set_reference_bank(self, features: torch.Tensor) -> None:
        """features [R, D] L2-normalised (``features.npy``, retrieval_ref.py:99)."""
        self.clip_model.engine.set_bank(features.to(self.clip_model.device), name=self.bank_name)

    def _cons_cfg(self) -> ConsistencyConfig:
        c = self.config
        return ConsistencyConfig(reference_count=c.reference_count, similarity_threshold=c.similarity_threshold,
                                 retrieval_top_k=c.retrieval_top_k)

    def _variants(self, text: str) -> List[str]:
        if not self.config.use_text_variants:
            return []
        return as_generator(self.text_variant_generator, self.config.text_variant_count)(text)

    def scores_from_tokens(self, images: torch.Tensor, tokens: torch.Tensor) -> Dict[str, np.ndarray]:
        """Batched core on pre-tokenised input: images [B,3,S,S], tokens [B,N+1,ctx]."""
        clip = self.clip_model
        B, N1, ctx = tokens.shape
        fi = clip.engine.encode_image(images.to(clip.device, torch.float32), True)
        ft = clip.engine.encode_text(tokens.reshape(B * N1, ctx).to(clip.device, torch.int32), True, group=N1)
        use_bank = self.config.use_retrieval_ref and clip.engine.bank_size(self.bank_name) > 0
        rec = clip.engine.detect_embeddings(fi, ft.view(B, N1, -1), self._cons_cfg(), use_bank=use_bank, robust=True,
                                            bank=self.bank_name)
        return unpack_records(rec, N1 - 1)

    def batch_detect(self, images: torch.Tensor, texts: Sequence[str], return_details: bool = False,
                     variants: Optional[Sequence[Sequence[str]]] = None) -> List[Dict[str, Any]]:
        """experiments/defenses/detector.py:327-351, batched; decisions are taken
        in input order (the checker is stateful)."""
        clip = self.clip_model
        eng = clip.engine
        n = len(texts)
        x, _ = clip._images_to_device(images)
        join_fi = clip.encode_image_beside(x, True)         # enqueued first, on the side stream: beside the host work AND the text tower
        fi = None
        if variants is None:
            variants = batch_variants(self.text_variant_generator, self.config.text_variant_count, texts) \
                if self.config.use_text_variants else [[] for _ in texts]
        per_query: List[Optional[Dict[str, float]]] = [None] * n
        extra: List[Optional[Dict]] = [None] * n
        groups: Dict[int, List[int]] = {}
        for i, v in enumerate(variants):
            groups.setdefault(len(v), []).append(i)
        use_bank = self.config.use_retrieval_ref and eng.bank_size(self.bank_name) > 0
        for N, ids in groups.items():
            flat: List[str] = []
            for i in ids:
                flat.append(texts[i])
                flat.extend(variants[i])
            ft = clip.encode_tokens(clip.tokenize(flat), True, group=N + 1).view(len(ids), N + 1, -1)
            if fi is None:
                fi = join_fi()
            sel = fi if len(ids) == n else fi[torch.as_tensor(ids, device=fi.device)].contiguous()
            rec = unpack_records(eng.detect_embeddings(sel, ft, self._cons_cfg(), use_bank, robust=True,
                                                       bank=self.bank_name), N)
            s0 = rec["original_similarity"].tolist()
            vm, vs = rec["variant_mean"].tolist(), rec["variant_std"].tolist()
            rc, rs = rec["retrieval_consistency"].tolist(), rec["retrieval_std"].tolist()
            xv, nref = rec["cross_modal_variance"].tolist(), rec["n_references"].tolist()
            ridx, rsim = rec["reference_indices"].tolist(), rec["reference_similarities"].tolist()
            for j, i in enumerate(ids):
                per_query[i] = {
                    "original_similarity": s0[j],
                    "text_variant_consistency": vm[j] if N > 0 else s0[j],
                    "text_variant_std": vs[j] if N > 0 else 0.0,
                    "retrieval_consistency": rc[j], "retrieval_std": rs[j],
                    "generative_consistency": 0.0, "generative_std": 0.0,       # filled below when a generator is injected
                    "cross_modal_variance": xv[j]}
                k = nref[j]
                extra[i] = {"retrieval_references": ridx[j][:k], "retrieval_similarities": rsim[j][:k]}
        # ---- generative references (experiments/defenses/detector.py:206-226,268-280): generated for the first three
        # of (original + variants), cut to generation_count; ALL of them encoded in one image-tower launch, the
        # cosines with the query image from the K4 kernel (record words 0, 12..), statistics on the host
        gen_refs: List[list] = [[] for _ in range(n)]
        if self.config.use_generative_ref and self.generative_generator is not None:
            for i in range(n):
                got: list = []
                for t in ([texts[i]] + list(variants[i]))[:3]:
                    got.extend(self.generative_generator.generate_references(t))
                gen_refs[i] = got[:self.config.generation_count]
            flat_imgs = [im for g in gen_refs for im in g]
            if flat_imgs:
                xr, _ = clip._images_to_device(flat_imgs)
                fr = eng.encode_image(xr, True)
                counts = [len(g) for g in gen_refs]
                offs = np.concatenate([[0], np.cumsum(counts)])
                by_count: Dict[int, List[int]] = {}
                for i, c in enumerate(counts):
                    if c:
                        by_count.setdefault(c, []).append(i)
                for J, ids in by_count.items():
                    rows = torch.as_tensor(np.concatenate([np.arange(offs[i], offs[i] + J) for i in ids]), device=fr.device)
                    qsel = fi[torch.as_tensor(ids, device=fi.device)].contiguous()
                    r = eng.consistency(qsel, fr[rows].view(len(ids), J, -1), self._cons_cfg()).cpu().numpy().astype(np.float64)
                    sims = np.concatenate([r[:, 0:1], r[:, 12:12 + J - 1]], axis=1)
                    for j, i in enumerate(ids):
                        pq = per_query[i]
                        pq["generative_consistency"] = float(sims[j].mean())
                        pq["generative_std"] = float(sims[j].std())
                        valid = [v for v in (pq["original_similarity"], pq["text_variant_consistency"],
                                             pq["retrieval_consistency"], pq["generative_consistency"]) if v > 0]
                        pq["cross_modal_variance"] = float(np.var(valid)) if len(valid) >= 2 else 0.0     # :295-300
        out = []
        for i in range(n):
            d = self.consistency_checker.make_decision(per_query[i], return_details=return_details)
            res = {"is_adversarial": d["is_adversarial"], "confidence": d["confidence"],
                   "consistency_score": d["overall_score"]}
            if return_details:
                res["details"] = {"text_variants": [texts[i]] + list(variants[i]),
                                  "retrieval_references": extra[i]["retrieval_references"],
                                  "retrieval_similarities": extra[i]["retrieval_similarities"],
                                  "generative_references": gen_refs[i], "consistency_scores": per_query[i],
                                  "detection_details": d}
            out.append(res)
        return out

    def detect(self, image: torch.Tensor, text: str, return_details: bool = False, **kw) -> Dict[str, Any]:
        """experiments/defenses/detector.py:117-170."""
        return self.batch_detect(image, [text], return_details, **{k: [v] for k, v in kw.items()})[0]

    def get_statistics(self) -> Dict[str, Any]:
        return {"config": dict(self.config.__dict__),
                "components": {"text_variant_generator": self.text_variant_generator is not None,
                               "retrieval_generator": self.clip_model.engine.bank_size(self.bank_name) > 0,
                               "generative_generator": self.generative_generator is not None, "consistency_checker": True}}
