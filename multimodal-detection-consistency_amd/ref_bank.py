"""``ReferenceBank`` (``src/ref_bank.py:86``) with the similarity scan on the GPU.

The reference rebuilds an fp64 matrix from a Python list on EVERY query
(``src/ref_bank.py:475``) and scans it with numpy.  Here the vectors are kept as
a device matrix of L2-normalised rows (re-registered lazily after additions) and
``query_similar`` is one ``tvc_bank_search`` call (fp32 bank -> split-bf16
planes, fp32-grade cosines).  In scope: ``add_reference`` (admission check
included), ``query_similar``, ``_compute_similarities``, size / FIFO-LRU-random
eviction bookkeeping.  Out of scope (host bookkeeping off the q/s path,
SURVEY.md 2.1 #4): KMeans / DBSCAN clustering and JSON persistence.
"""
from __future__ import annotations

import time
from collections import deque
from dataclasses import dataclass, field
from threading import Lock
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from .engine import TVCEngine


@dataclass
class ReferenceBankConfig:
    """src/ref_bank.py:24-44."""
    max_size: int = 10000
    similarity_threshold: float = 0.9
    clustering_method: str = "none"
    num_clusters: int = 100
    update_strategy: str = "fifo"
    persistence_enabled: bool = False
    save_path: str = "./cache/ref_bank"
    auto_clustering: bool = False
    clustering_interval: int = 1000
    feature_dim: int = 512

    def __post_init__(self):
        if self.update_strategy not in ("fifo", "lru", "random", "similarity"):
            raise ValueError(f"unsupported update strategy: {self.update_strategy}")


@dataclass
class ReferenceItem:
    """src/ref_bank.py:47-83."""
    vector: np.ndarray
    metadata: Dict[str, Any]
    timestamp: float
    access_count: int = 0
    cluster_id: Optional[int] = None
    similarity_scores: Dict[str, float] = field(default_factory=dict)


class ReferenceBank:
    def __init__(self, config: Optional[ReferenceBankConfig] = None, engine: Optional[TVCEngine] = None,
                 rng: Optional[np.random.Generator] = None):
        self.config = config or ReferenceBankConfig()
        self.engine = engine or TVCEngine()
        self.references: List[ReferenceItem] = []
        self.access_order: deque = deque()
        self._lock = Lock()
        self._dirty = True
        self._rng = rng or np.random.default_rng()
        self.bank_name = f"ref_bank:{id(self):x}"       # own bank slot on the (possibly shared) engine
        self.stats = {"total_added": 0, "total_removed": 0, "total_queries": 0}

    def __len__(self) -> int:
        return len(self.references)

    def __del__(self):          # give the bank slot back to the (possibly shared) engine
        try:
            self.engine.release_bank(self.bank_name)
        except Exception:
            pass

    # -- device copy ----------------------------------------------------------
    def _padded_dim(self) -> int:
        return (self.config.feature_dim + 63) // 64 * 64

    def _sync_device(self) -> None:
        if not self._dirty:
            return
        D, Dp = self.config.feature_dim, self._padded_dim()
        V = np.zeros((len(self.references), Dp), dtype=np.float64)
        for i, r in enumerate(self.references):
            V[i, :D] = r.vector
        n = np.linalg.norm(V, axis=1, keepdims=True)
        V = V / np.where(n == 0, 1.0, n)          # cosine = dot of unit rows; the +1e-8 of :482 is < 1e-9 here
        self.engine.set_bank(torch.from_numpy(V.astype(np.float32)).to(self.engine.device), name=self.bank_name)
        self._dirty = False

    def _query_rows(self, q: np.ndarray) -> torch.Tensor:
        D, Dp = self.config.feature_dim, self._padded_dim()
        q = np.asarray(q, dtype=np.float64).reshape(-1, D)
        n = np.linalg.norm(q, axis=1, keepdims=True)
        out = np.zeros((q.shape[0], Dp), dtype=np.float32)
        out[:, :D] = q / np.where(n == 0, 1.0, n)
        return torch.from_numpy(out).to(self.engine.device)

    # -- API ------------------------------------------------------------------
    def _compute_similarities(self, query_vector: np.ndarray) -> np.ndarray:
        """src/ref_bank.py:462-484 for ALL references (debug / parity use): the
        cosine matrix kernel instead of the top-k search."""
        if not self.references:
            return np.array([])
        from .metrics import SimilarityCalculator
        V = np.stack([r.vector for r in self.references]).astype(np.float32)
        return SimilarityCalculator.batch_cosine_similarity(np.asarray(query_vector, np.float32).reshape(1, -1), V,
                                                            engine=self.engine)[0].astype(np.float64)

    def _is_too_similar(self, vector: np.ndarray) -> bool:
        """src/ref_bank.py:340-362: the reference samples <= 100 refs with the global
        numpy RNG (non-deterministic admission); here the check is against ALL
        references (max similarity from the search), which is the sampled check's
        limit and is deterministic."""
        if not self.references:
            return False
        self._sync_device()
        _, sim, _ = self.engine.bank_search_robust(self._query_rows(vector), 1, want_moments=False, bank=self.bank_name)
        return bool(sim[0, 0].item() > self.config.similarity_threshold)

    def add_reference(self, vector: np.ndarray, metadata: Dict[str, Any]) -> bool:
        """src/ref_bank.py:123-170."""
        with self._lock:
            if self._is_too_similar(vector):
                return False
            if len(self.references) >= self.config.max_size:
                self._remove_reference()
            self.references.append(ReferenceItem(np.array(vector, dtype=np.float64), dict(metadata), time.time()))
            self.stats["total_added"] += 1
            self._dirty = True
            return True

    def add_references(self, vectors: np.ndarray, metadatas: Optional[List[Dict[str, Any]]] = None) -> None:
        """Bulk load without the admission check (e.g. a stored ``references.json``)."""
        with self._lock:
            for i, v in enumerate(vectors):
                self.references.append(ReferenceItem(np.array(v, dtype=np.float64),
                                                     dict(metadatas[i]) if metadatas else {}, time.time()))
            self.stats["total_added"] += len(vectors)
            self._dirty = True

    def _remove_reference(self) -> None:
        """src/ref_bank.py:364-400 (fifo / lru / random; 'similarity' falls back to fifo)."""
        if not self.references:
            return
        s = self.config.update_strategy
        idx = 0
        if s == "lru" and self.access_order:
            idx = self.access_order.popleft()
            idx = idx if idx < len(self.references) else 0
        elif s == "random":
            idx = int(self._rng.integers(len(self.references)))
        self.references.pop(idx)
        self.access_order = deque(i if i < idx else i - 1 for i in self.access_order if i != idx)
        self.stats["total_removed"] += 1
        self._dirty = True

    def query_similar(self, query_vector: np.ndarray, top_k: int = 10,
                      similarity_threshold: Optional[float] = None) -> List[Tuple[ReferenceItem, float]]:
        """src/ref_bank.py:172-224: sims >= threshold, sorted descending, top_k."""
        with self._lock:
            if not self.references:
                return []
            thr = similarity_threshold or self.config.similarity_threshold       # :191 (0.0 falls through)
            self._sync_device()
            # the reference accepts any top_k (src/ref_bank.py:172,203); more than the bank holds cannot come
            # back, and beyond the kernel's TVC_MAX_TOPK the engine raises -- never a silent truncation
            k = max(1, min(top_k, len(self.references)))
            idx, sim, _ = self.engine.bank_search_robust(self._query_rows(query_vector), k, thr, want_moments=False,
                                                         bank=self.bank_name)
            idx, sim = idx[0].cpu().numpy(), sim[0].cpu().numpy().astype(np.float64)
            out = []
            for i, s in zip(idx, sim):
                if i < 0 or s < thr:
                    continue
                item = self.references[int(i)]
                item.access_count += 1
                if int(i) in self.access_order:
                    self.access_order.remove(int(i))
                self.access_order.append(int(i))
                out.append((item, float(s)))
            self.stats["total_queries"] += 1
            return out

    def get_statistics(self) -> Dict[str, Any]:
        return {**self.stats, "size": len(self.references), "max_size": self.config.max_size}


def create_reference_bank(config: Optional[ReferenceBankConfig] = None, **kw) -> ReferenceBank:
    """src/ref_bank.py:727."""
    return ReferenceBank(config, **kw)
