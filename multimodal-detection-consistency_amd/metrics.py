"""``SimilarityCalculator`` (``src/utils/metrics.py:109-164``) over the HIP cosine
kernel, and ``DetectionEvaluator.compute_detection_metrics`` (``:286-329``), which
stays sklearn on the host exactly as in the reference."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Union

import numpy as np
import torch

from .engine import TVCEngine, _ptr, _stream

_default_engine: Optional[TVCEngine] = None


def _engine(engine: Optional[TVCEngine]) -> TVCEngine:
    global _default_engine
    if engine is not None:
        return engine
    if _default_engine is None:
        _default_engine = TVCEngine()          # consistency-only handle; raises without a GPU
    return _default_engine


class SimilarityCalculator:
    @staticmethod
    def batch_cosine_similarity(x: Union[np.ndarray, torch.Tensor], y: Union[np.ndarray, torch.Tensor],
                                engine: Optional[TVCEngine] = None) -> np.ndarray:
        """[N, D] x [M, D] -> [N, M] cosines (src/utils/metrics.py:144-164); D is
        zero-padded to a multiple of 64 (cosines are unchanged)."""
        eng = _engine(engine)
        xt = torch.as_tensor(x, dtype=torch.float32)
        yt = torch.as_tensor(y, dtype=torch.float32)
        N, D = xt.shape
        M = yt.shape[0]
        Dp = (D + 63) // 64 * 64
        if Dp != D:
            xt = torch.nn.functional.pad(xt, (0, Dp - D))
            yt = torch.nn.functional.pad(yt, (0, Dp - D))
        xd, yd = xt.to(eng.device).contiguous(), yt.to(eng.device).contiguous()
        out = torch.empty((N, M), dtype=torch.float32, device=eng.device)
        with eng._lock, torch.cuda.device(eng.device):
            eng._check(eng.lib.tvc_cosine_matrix(eng.handle, _ptr(xd), N, _ptr(yd), M, Dp, _ptr(out), _stream()))
        return out.cpu().numpy()

    @staticmethod
    def cosine_similarity(x, y, engine: Optional[TVCEngine] = None) -> float:
        """src/utils/metrics.py:116-141: scalar cosine, 0.0 for a zero vector."""
        xa = np.asarray(x.detach().cpu() if isinstance(x, torch.Tensor) else x, dtype=np.float32).reshape(1, -1)
        ya = np.asarray(y.detach().cpu() if isinstance(y, torch.Tensor) else y, dtype=np.float32).reshape(1, -1)
        if not xa.any() or not ya.any():
            return 0.0
        return float(SimilarityCalculator.batch_cosine_similarity(xa, ya, engine)[0, 0])


@dataclass
class DetectionMetrics:
    auc: float
    accuracy: float
    precision: float
    recall: float
    f1_score: float
    fpr_at_95_tpr: float
    threshold: float
    confusion_matrix: np.ndarray


class DetectionEvaluator:
    @staticmethod
    def compute_detection_metrics(scores: np.ndarray, labels: np.ndarray, pos_label: int = 1) -> DetectionMetrics:
        """src/utils/metrics.py:286-329: roc_auc_score + Youden-J threshold."""
        from sklearn.metrics import (accuracy_score, confusion_matrix, f1_score, precision_score, recall_score,
                                     roc_auc_score, roc_curve)
        scores = np.asarray(scores, dtype=np.float64)
        labels = np.asarray(labels).astype(int)
        fpr, tpr, thr = roc_curve(labels, scores, pos_label=pos_label)
        auc = roc_auc_score(labels, scores)
        j = int(np.argmax(tpr - fpr))
        pred = (scores >= thr[j]).astype(int)
        i95 = int(np.argmin(np.abs(tpr - 0.95)))       # src/utils/metrics.py:343-344: nearest TPR
        return DetectionMetrics(
            auc=float(auc), accuracy=float(accuracy_score(labels, pred)),
            precision=float(precision_score(labels, pred, pos_label=pos_label, zero_division=0)),
            recall=float(recall_score(labels, pred, pos_label=pos_label, zero_division=0)),
            f1_score=float(f1_score(labels, pred, pos_label=pos_label, zero_division=0)),
            fpr_at_95_tpr=float(fpr[i95]), threshold=float(thr[j]),
            confusion_matrix=confusion_matrix(labels, pred))
