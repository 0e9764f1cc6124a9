"""MI355X-native text-variant-consistency (TVC) hot path.

Drop-in for the ``src/pipeline.py -> retrieval.py / detector.py / ref_bank.py``
path of Zhang-Xin-Duke/multimodal-detection-consistency: the same call surface
(``pipeline.detect()``, ``Detector``, ``ReferenceBank`` ...) over hand-written
HIP kernels for gfx950 reached through the C-ABI in ``include/tvc.h``.
There is no CPU fallback: without ``libtvc_hip.so`` and a GPU every compute call
raises ``TVCError``.
"""
from . import _lib, synth
from ._lib import TVCError, build
from .arch import ARCHS, ClipArch, Tower, get_arch
from .engine import ConsistencyConfig, TVCEngine

__all__ = ["TVCError", "build", "ARCHS", "ClipArch", "Tower", "get_arch", "ConsistencyConfig", "TVCEngine"]
