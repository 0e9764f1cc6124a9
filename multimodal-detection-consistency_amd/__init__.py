"""MI355X-native text-variant-consistency (TVC) hot path.

Drop-in for the ``src/pipeline.py -> retrieval.py / detector.py / ref_bank.py``
path of Zhang-Xin-Duke/multimodal-detection-consistency: the same call surface
(``pipeline.detect()``, ``Detector``, ``ReferenceBank`` ...) over hand-written
HIP kernels for gfx950 reached through the C-ABI in ``include/tvc.h``.
There is no CPU fallback: without ``libtvc_hip.so`` and a GPU every compute call
raises ``TVCError``.
"""
from . import _lib, attacks, sharding, synth
from ._lib import TVCError, build
from .attacks import (HubnessAttack, HubnessAttackConfig, HubnessAttackPresets, PGDAttackConfig, PGDAttacker,
                      create_hubness_attacker, create_pgd_attacker)
from .arch import ARCHS, ClipArch, Tower, get_arch
from .clip import CLIPConfig, CLIPModel
from .detector import (AdversarialDetector, ConsistencyChecker, DetectionConfig, DetectorConfig,
                       MultiModalDefenseDetector, aggregate_scores, create_adversarial_detector, unpack_records)
from .engine import ConsistencyConfig, TVCEngine
from .metrics import DetectionEvaluator, DetectionMetrics, SimilarityCalculator
from .pipeline import (DefensePipeline, MultiModalDetectionPipeline, PipelineConfig, PipelineProfiler,
                       PipelineResult, create_defense_pipeline, create_detection_pipeline)
from .ref_bank import ReferenceBank, ReferenceBankConfig, ReferenceItem, create_reference_bank
from .retrieval import (MultiModalRetriever, RetrievalConfig, RetrievalRefConfig, RetrievalReferenceGenerator,
                        create_retriever, extract_features)
from .sd_arch import SDArch, make_sd_weights
from .sd_model import SDKernels, SDModelConfig, StableDiffusionModel, create_sd_model
from .sd_ref import (GenerativeConfig, GenerativeReferenceGenerator, GenerationResult, QualityFilter, QualityMetrics, SDReferenceConfig, SDReferenceGenerator,
                     create_sd_reference_generator)
from .text_variants import TextVariantConfig, TextVariantGenerator
from .variants import TemplateVariantGenerator

__all__ = [n for n in dir() if not n.startswith("_")]
