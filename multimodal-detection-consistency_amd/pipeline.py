"""``MultiModalDetectionPipeline`` (``src/pipeline.py:256``) over the batched HIP
path, with the call surface the existing runners use:

* ``process_single(image, text)`` -> ``PipelineResult`` (``src/pipeline.py:333``,
  caller ``src/evaluation/experiment_evaluator.py:243``)
* ``process(image, text)`` (``experiments/run_experiments.py:3326``)
* ``process_batch(images, texts)`` (``src/pipeline.py:536``)
* ``detect(images=, texts=, return_details=)`` ->
  ``{'predictions', 'scores', 'details'}`` (``experiments/runners/run_detection.py:172-203``)
* ``evaluate_pipeline(test_data)`` (``src/pipeline.py:605``)

Differences kept deliberate: ``process_batch`` returns results in INPUT order
(the reference's thread pool returns completion order, ``src/pipeline.py:562-565``)
and is one batched launch instead of a 4-thread pool of single queries; errors
raise unless ``strict=False``.
"""
from __future__ import annotations

import threading
import time
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .clip import CLIPConfig, CLIPModel
from .detector import AdversarialDetector, DetectorConfig
from .metrics import DetectionEvaluator
from .retrieval import MultiModalRetriever, RetrievalConfig
from .variants import as_generator, batch_variants


@dataclass
class PipelineConfig:
    """src/pipeline.py:31-74 (same field names)."""
    enable_text_augment: bool = True
    enable_retrieval: bool = True
    enable_sd_reference: bool = True
    enable_detection: bool = True
    enable_parallel: bool = True
    max_workers: int = 4
    batch_size: int = 32
    enable_cache: bool = True
    cache_dir: Optional[str] = None
    save_intermediate_results: bool = False
    output_dir: Optional[str] = None
    enable_profiling: bool = False
    profile_steps: bool = True
    text_augment_config: Any = None
    retrieval_config: Optional[RetrievalConfig] = None
    sd_reference_config: Any = None
    detector_config: Optional[DetectorConfig] = None
    strict: bool = True

    def __post_init__(self):
        if self.retrieval_config is None:
            self.retrieval_config = RetrievalConfig()
        if self.detector_config is None:
            self.detector_config = DetectorConfig()


@dataclass
class PipelineResult:
    """src/pipeline.py:77-131."""
    original_image: Any = None
    original_text: str = ""
    text_variants: List[str] = field(default_factory=list)
    text_augment_time: float = 0.0
    retrieved_images: List[Any] = field(default_factory=list)
    retrieved_texts: List[str] = field(default_factory=list)
    retrieval_scores: List[float] = field(default_factory=list)
    retrieval_time: float = 0.0
    reference_images: List[Any] = field(default_factory=list)
    reference_generation_time: float = 0.0
    is_adversarial: bool = False
    detection_score: float = 0.0
    detection_details: Dict[str, Any] = field(default_factory=dict)
    detection_time: float = 0.0
    total_time: float = 0.0
    pipeline_steps: List[str] = field(default_factory=list)
    errors: List[str] = field(default_factory=list)

    @property
    def adversarial_score(self) -> float:
        return self.detection_score

    def to_dict(self) -> Dict[str, Any]:
        return {k: getattr(self, k) for k in (
            "original_text", "text_variants", "text_augment_time", "retrieved_texts", "retrieval_scores",
            "retrieval_time", "reference_generation_time", "is_adversarial", "detection_score",
            "detection_details", "detection_time", "total_time", "pipeline_steps", "errors")}


class PipelineProfiler:
    """src/pipeline.py:179-253: wall-clock per step (count / total / mean / std / min / max)."""

    def __init__(self):
        self._lock = threading.Lock()
        self.times: Dict[str, List[float]] = {}

    def add(self, step: str, seconds: float) -> None:
        with self._lock:
            self.times.setdefault(step, []).append(seconds)

    def summary(self) -> Dict[str, Dict[str, float]]:
        with self._lock:
            return {s: {"count": len(t), "total": float(np.sum(t)), "mean": float(np.mean(t)),
                        "std": float(np.std(t)), "min": float(np.min(t)), "max": float(np.max(t))}
                    for s, t in self.times.items() if t}


class MultiModalDetectionPipeline:
    def __init__(self, config: Optional[PipelineConfig] = None, clip_model: Optional[CLIPModel] = None,
                 text_augmenter=None, sd_generator=None):
        self.config = config or PipelineConfig()
        dc = self.config.detector_config
        self.clip_model = clip_model or CLIPModel(CLIPConfig(model_name=dc.clip_model, device=dc.device))
        self.text_augmenter = text_augmenter
        self.sd_generator = sd_generator
        self.detector = AdversarialDetector(dc, clip_model=self.clip_model, text_augmenter=text_augmenter,
                                            sd_generator=sd_generator) if self.config.enable_detection else None
        self.retriever = MultiModalRetriever(self.config.retrieval_config, clip_model=self.clip_model) \
            if self.config.enable_retrieval else None
        self.profiler = PipelineProfiler() if self.config.enable_profiling else None
        self.metrics_calculator = DetectionEvaluator()
        self.pipeline_stats = {"total_processed": 0, "successful_processed": 0, "failed_processed": 0,
                               "total_time": 0.0,
                               "component_usage": {"text_augment": 0, "retrieval": 0, "sd_reference": 0, "detection": 0}}
        self._lock = threading.Lock()

    # ------------------------------------------------------------------ core
    def _default_steps(self) -> List[str]:
        c = self.config
        return [s for s, on in (("text_augment", c.enable_text_augment), ("retrieval", c.enable_retrieval),
                                ("sd_reference", c.enable_sd_reference), ("detection", c.enable_detection)) if on]

    def process_batch(self, images, texts: Sequence[str], steps: Optional[List[str]] = None) -> List[PipelineResult]:
        if len(images) != len(texts):
            raise ValueError("number of images and texts differ")          # src/pipeline.py:550-551
        steps = steps or self._default_steps()
        n = len(texts)
        t_start = time.time()
        results = [PipelineResult(original_text=t, pipeline_steps=list(steps)) for t in texts]
        dc = self.config.detector_config
        methods = [m for m in dc.detection_methods if m != "sd_reference" or "sd_reference" in steps]
        image_rows = None
        try:
            variants = None
            # the image tower needs nothing of the text side: it is enqueued (on the model's side stream) BEFORE the host
            # generates the text variants, so that the variant generator's time runs under it instead of in front of it
            if "detection" in steps and self.detector is not None and self.detector._get_clip_model() is self.clip_model:
                x, _ = self.clip_model._images_to_device(images if isinstance(images, torch.Tensor) else list(images))
                if x.shape[0] != n:
                    raise ValueError("number of images and texts differ")
                image_rows = self.clip_model.encode_image_beside(x, True)
            if "text_augment" in steps:
                t0 = time.time()
                variants = batch_variants(self.text_augmenter, dc.num_text_variants, texts)
                dt = (time.time() - t0) / max(n, 1)
                for r, v in zip(results, variants):
                    r.text_variants, r.text_augment_time = v, dt
                self._count("text_augment", n, dt * n)
            do_retrieval = "retrieval" in steps and self.retriever is not None and self.retriever._bank_is == "image"
            do_detection = "detection" in steps and self.detector is not None
            # retrieval needs the L2-normalised rows of the ORIGINAL texts, which the detection step encodes
            # anyway: when both run (and normalise alike) the rows are handed over instead of encoded twice
            share = do_retrieval and do_detection and self.retriever.config.normalize_features and \
                self.retriever.clip_model is self.clip_model
            if do_retrieval and not share:
                t0 = time.time()
                got = self.retriever.batch_retrieve_images_by_texts(list(texts), top_k=5)     # src/pipeline.py:450-453
                self._fill_retrieval(results, got, (time.time() - t0) / max(n, 1), n)
            ref_images = None
            if "sd_reference" in steps and self.sd_generator is not None:
                t0 = time.time()
                ref_images = [self.sd_generator.generate_reference_images(t, num_images=3).get("images", [])
                              for t in texts]                                                  # src/pipeline.py:495-498
                dt = (time.time() - t0) / max(n, 1)
                for r, im in zip(results, ref_images):
                    r.reference_images, r.reference_generation_time = im, dt
                self._count("sd_reference", n, dt * n)
            if "detection" in steps and self.detector is not None:
                t0 = time.time()
                if variants is None and "text_variants" in methods:
                    methods = [m for m in methods if m != "text_variants"]
                kept: Optional[Dict[str, torch.Tensor]] = {} if share else None
                det = self.detector.batch_detect(images, list(texts), methods=methods,
                                                 variants=variants if "text_variants" in methods else [[] for _ in texts],
                                                 reference_images=ref_images, keep_features=kept, image_rows=image_rows)
                dt = (time.time() - t0) / max(n, 1)
                if share:
                    t1 = time.time()
                    got = self.retriever.batch_retrieve_images_by_features(kept["text"], top_k=5)  # src/pipeline.py:450-453
                    self._fill_retrieval(results, got, (time.time() - t1) / max(n, 1), n)
                for r, d in zip(results, det):
                    r.is_adversarial = d["is_adversarial"]
                    r.detection_score = d["aggregated_score"]
                    r.detection_details = d["detection_details"]
                    r.detection_time = dt
                self._count("detection", n, dt * n)
        except Exception as e:
            if image_rows is not None:
                image_rows()                 # the current stream waits for the image tower before its input can be released
            if self.config.strict:
                raise
            for r in results:
                r.errors.append(str(e))
        total = time.time() - t_start
        for r in results:
            r.total_time = total / max(n, 1)
        with self._lock:
            self.pipeline_stats["total_processed"] += n
            ok = sum(1 for r in results if not r.errors)
            self.pipeline_stats["successful_processed"] += ok
            self.pipeline_stats["failed_processed"] += n - ok
            self.pipeline_stats["total_time"] += total
        return results

    def _fill_retrieval(self, results, got, dt: float, n: int) -> None:
        for r, (paths, scores) in zip(results, got):
            r.retrieved_images, r.retrieval_scores = paths, scores
            r.retrieved_texts = [r.original_text] * len(paths)                                # src/pipeline.py:468
            r.retrieval_time = dt
        self._count("retrieval", n, dt * n)

    def _count(self, step: str, n: int, seconds: float) -> None:
        with self._lock:
            self.pipeline_stats["component_usage"][step] += n
        if self.profiler:
            self.profiler.add(step, seconds)

    def process_single(self, image, text: str, steps: Optional[List[str]] = None) -> PipelineResult:
        img = image.unsqueeze(0) if isinstance(image, torch.Tensor) and image.dim() == 3 else \
            (image if isinstance(image, torch.Tensor) else [image])
        r = self.process_batch(img, [text], steps)[0]
        if not isinstance(image, torch.Tensor):
            r.original_image = image
        return r

    process = process_single       # experiments/run_experiments.py:3326

    def detect(self, images, texts: Sequence[str], return_details: bool = False) -> Dict[str, List[Any]]:
        """experiments/runners/run_detection.py:172-203."""
        res = self.process_batch(images, list(texts))
        out = {"predictions": [bool(r.is_adversarial) for r in res], "scores": [float(r.detection_score) for r in res]}
        out["details"] = [r.to_dict() if return_details else {} for r in res]
        return out

    def evaluate_pipeline(self, test_data: Sequence[Tuple[Any, str, bool]], steps: Optional[List[str]] = None,
                          batch_size: Optional[int] = None) -> Dict[str, Any]:
        """src/pipeline.py:605-680: detection metrics + mean step times; batched."""
        bs = batch_size or self.config.batch_size
        results: List[PipelineResult] = []
        for i in range(0, len(test_data), bs):
            chunk = test_data[i:i + bs]
            imgs = [c[0] for c in chunk]
            if all(isinstance(x, torch.Tensor) for x in imgs):
                imgs = torch.stack([x if x.dim() == 3 else x[0] for x in imgs])
            results.extend(self.process_batch(imgs, [c[1] for c in chunk], steps))
        labels = np.array([int(c[2]) for c in test_data])
        scores = np.array([r.detection_score for r in results])
        m = self.metrics_calculator.compute_detection_metrics(scores, labels) if len(set(labels.tolist())) > 1 else None
        total = sum(r.total_time for r in results)
        return {"detection_metrics": m,
                "predictions": [r.is_adversarial for r in results], "scores": scores.tolist(),
                "time_stats": {"mean_total_time": float(np.mean([r.total_time for r in results])),
                               "mean_detection_time": float(np.mean([r.detection_time for r in results])),
                               "mean_text_augment_time": float(np.mean([r.text_augment_time for r in results])),
                               "mean_retrieval_time": float(np.mean([r.retrieval_time for r in results])),
                               "mean_reference_time": float(np.mean([r.reference_generation_time for r in results]))},
                "throughput": len(results) / total if total > 0 else 0.0,     # experiment_evaluator.py:269
                "profiling": self.profiler.summary() if self.profiler else {}}

    def get_pipeline_stats(self) -> Dict[str, Any]:
        return dict(self.pipeline_stats)


DefensePipeline = MultiModalDetectionPipeline          # src/pipeline.py:805


def create_detection_pipeline(config: Optional[PipelineConfig] = None, **kw) -> MultiModalDetectionPipeline:
    """src/pipeline.py:808."""
    return MultiModalDetectionPipeline(config, **kw)


def create_defense_pipeline(config: Optional[PipelineConfig] = None, **kw) -> MultiModalDetectionPipeline:
    """src/pipeline.py:824."""
    return MultiModalDetectionPipeline(config, **kw)
