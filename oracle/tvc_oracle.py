"""CPU restatement (numpy, fp64 accumulate) of the reference's TVC arithmetic.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Each function cites
the reference file:line (paths relative to ``/root/reference``) it follows.
The functions take *embeddings* (the outputs of the CLIP towers); the towers
themselves are restated in ``oracle/clip_oracle.py``.

Two score polarities exist in the reference and both are restated:

* ``src`` polarity  -- ``src/detector.py``: high score => adversarial
  (``aggregated_score > detection_threshold``, ``src/detector.py:399``).
* ``exp`` polarity  -- ``experiments/defenses``: low score => adversarial
  (``overall_score < threshold``, ``experiments/defenses/consistency_checker.py:93``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


# --------------------------------------------------------------------------
# cosine helpers
# --------------------------------------------------------------------------
def cosine(a: np.ndarray, b: np.ndarray, eps: float = 1e-8) -> float:
    """``torch.cosine_similarity(a, b, dim=-1).item()`` semantics.

    experiments/defenses/detector.py:240,248,262 -- torch clamps each norm to
    ``eps`` (1e-8).  ``get_text_image_similarity`` (src/detector.py:461) is a
    call into the absent ``src.models`` wrapper; it is restated as the same
    cosine.
    """
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    na = max(float(np.linalg.norm(a)), eps)
    nb = max(float(np.linalg.norm(b)), eps)
    return float(np.dot(a, b) / (na * nb))


def batch_cosine_similarity(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """src/utils/metrics.py:144-164 (numpy branch): divide by row norms, dot."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    xn = x / np.linalg.norm(x, axis=1, keepdims=True)
    yn = y / np.linalg.norm(y, axis=1, keepdims=True)
    return np.dot(xn, yn.T)


def l2_normalize(x: np.ndarray) -> np.ndarray:
    """``x / x.norm(dim=-1, keepdim=True)`` --
    experiments/defenses/retrieval_ref.py:243, scripts/build_faiss_indices.py:108-109."""
    x = np.asarray(x, dtype=np.float64)
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


# --------------------------------------------------------------------------
# src polarity: src/detector.py
# --------------------------------------------------------------------------
def text_variant_score(original_similarity: float,
                       variant_similarities: Sequence[float]) -> Tuple[float, Dict]:
    """src/detector.py:473-497 (``_detect_by_text_variants`` arithmetic).

    numpy ``mean``/``std`` with ddof=0 (src/detector.py:475-476).
    """
    sv = np.asarray(variant_similarities, dtype=np.float64)
    if sv.size == 0:
        # src/detector.py:457-458
        return 0.0, {'error': 'no variants'}
    mean_v = sv.mean()
    std_v = sv.std()
    consistency_score = 1.0 - abs(original_similarity - mean_v)       # :479
    variability_score = 1.0 - std_v                                    # :482
    detection_score = 1.0 - (consistency_score * 0.7 + variability_score * 0.3)  # :485
    details = {
        'original_similarity': float(original_similarity),
        'variant_similarities': sv.tolist(),
        'mean_variant_similarity': float(mean_v),
        'std_variant_similarity': float(std_v),
        'consistency_score': float(consistency_score),
        'variability_score': float(variability_score),
        'num_variants': int(sv.size),
    }
    return float(detection_score), details


def consistency_score(image_text_similarity: float) -> Tuple[float, Dict]:
    """src/detector.py:573-586 (``_detect_by_consistency``): ``1 - cos``."""
    c = float(image_text_similarity)
    return float(1.0 - c), {'image_text_similarity': c, 'consistency_score': c}


def sd_reference_score(reference_similarities: Sequence[float]) -> Tuple[float, Dict]:
    """src/detector.py:536-553 (``_detect_by_sd_reference`` arithmetic):
    ``1 - mean`` of the query-image vs reference-image cosines."""
    s = np.asarray(reference_similarities, dtype=np.float64)
    if s.size == 0:
        return 0.0, {'error': 'no references'}                         # :524-525
    details = {
        'reference_similarities': s.tolist(),
        'mean_similarity': float(s.mean()),
        'max_similarity': float(s.max()),
        'std_similarity': float(s.std()),
        'num_references': int(s.size),
    }
    return float(1.0 - s.mean()), details


SRC_WEIGHTS = {'text_variants': 0.4, 'sd_reference': 0.4, 'consistency': 0.2}  # src/detector.py:666-670


def aggregate_scores(scores: Dict[str, float], method: str = 'weighted_mean') -> float:
    """src/detector.py:643-682 (``_aggregate_scores``)."""
    if not scores:
        return 0.0
    vals = list(scores.values())
    if method == 'mean':
        return float(np.mean(vals))
    if method == 'max':
        return float(np.max(vals))
    if method == 'min':
        return float(np.min(vals))
    if method == 'weighted_mean':
        ws = 0.0
        tw = 0.0
        for name, s in scores.items():
            w = SRC_WEIGHTS.get(name, 1.0)
            ws += s * w
            tw += w
        return ws / tw if tw > 0 else 0.0
    return float(np.mean(vals))


def detect_adversarial_src(image_feat: np.ndarray,
                           text_feats: np.ndarray,
                           methods: Sequence[str] = ('text_variants', 'consistency'),
                           sd_ref_feats: Optional[np.ndarray] = None,
                           detection_threshold: float = 0.5,
                           score_aggregation: str = 'weighted_mean',
                           has_text_augmenter: bool = True,
                           has_sd_generator: Optional[bool] = None) -> Dict:
    """src/detector.py:345-410 (``detect_adversarial``) on embeddings.

    ``text_feats`` is ``[N+1, D]``: row 0 the original text, rows 1.. the
    variants (src/detector.py:461-471).  ``sd_ref_feats`` (``[J, D]``) stands
    for the encoded SD reference images (src/detector.py:528-534); producing
    them is out of scope.

    A method is scored whenever it is requested AND its component exists
    (``self._get_text_augmenter() is not None`` :375, ``self._get_sd_generator() is not None``
    :382).  A component that exists but yields nothing (no variants :457-458, no images
    :524-525) still contributes its 0.0 score to the aggregation: with the default weights
    ``aggregated = (0.4 * 0 + 0.2 * cs) / 0.6``.  ``has_sd_generator`` defaults to
    "``sd_ref_feats`` was given".
    """
    text_feats = np.asarray(text_feats)
    s0 = cosine(image_feat, text_feats[0])
    scores, details = {}, {}
    if has_sd_generator is None:
        has_sd_generator = sd_ref_feats is not None
    if 'text_variants' in methods and has_text_augmenter:
        sv = [cosine(image_feat, t) for t in text_feats[1:]]
        scores['text_variants'], details['text_variants'] = text_variant_score(s0, sv)   # 0.0 + error when empty
    if 'sd_reference' in methods and has_sd_generator:
        sims = [cosine(image_feat, r) for r in sd_ref_feats] if sd_ref_feats is not None else []
        scores['sd_reference'], details['sd_reference'] = sd_reference_score(sims)       # 0.0 + error when empty
    if 'consistency' in methods:
        scores['consistency'], details['consistency'] = consistency_score(s0)
    agg = aggregate_scores(scores, score_aggregation)
    return {
        'is_adversarial': bool(agg > detection_threshold),              # :399
        'aggregated_score': float(agg),
        'detection_scores': scores,
        'detection_details': details,
        'threshold': detection_threshold,
    }


# --------------------------------------------------------------------------
# bank retrieval: experiments/defenses/retrieval_ref.py, src/retrieval.py,
# src/ref_bank.py
# --------------------------------------------------------------------------
@dataclass
class RetrievalConfig:
    """experiments/defenses/retrieval_ref.py:20-32 (fields the path reads)."""
    reference_count: int = 5
    similarity_threshold: float = 0.3
    enable_reranking: bool = True
    rerank_top_k: int = 20


def numpy_retrieve(bank: np.ndarray, q: np.ndarray, search_k: int) -> Tuple[np.ndarray, np.ndarray]:
    """experiments/defenses/retrieval_ref.py:268-290 (``_numpy_retrieve``).

    ``sims = bank @ q``; argpartition for the top ``search_k``; argsort desc.
    Returns ``(indices, sims[indices])``.
    """
    sims = np.dot(bank, np.asarray(q).reshape(-1, 1)).ravel()
    search_k = min(search_k, sims.shape[0])
    if search_k <= 0:
        return np.zeros(0, np.int64), np.zeros(0, sims.dtype)
    top = np.argpartition(sims, -search_k)[-search_k:]
    top = top[np.argsort(sims[top])[::-1]]
    return top.astype(np.int64), sims[top]


def retrieve_references(bank: np.ndarray, q_normed: np.ndarray,
                        cfg: RetrievalConfig = RetrievalConfig()) -> List[Dict]:
    """experiments/defenses/retrieval_ref.py:173-236 (``retrieve_references``)
    minus the text encode (``_encode_text`` :238-244 = L2-normalised fp32 row).
    """
    if bank.shape[0] == 0:
        return []                                                       # :195-197
    search_k = cfg.rerank_top_k if cfg.enable_reranking else cfg.reference_count  # :274
    idx, sims = numpy_retrieve(bank, q_normed, search_k)
    refs = [{'index': int(i), 'similarity': float(s)} for i, s in zip(idx, sims)]
    if cfg.enable_reranking and len(refs) > cfg.reference_count:        # :206-207
        refs = sorted(refs, key=lambda r: r['similarity'], reverse=True)   # :298
    refs = [r for r in refs if r['similarity'] >= cfg.similarity_threshold]  # :210-213
    return refs[:cfg.reference_count]                                   # :216


def search_index_exact(features: np.ndarray, q: np.ndarray, top_k: int) -> Tuple[np.ndarray, np.ndarray]:
    """src/retrieval.py:669-673 (exact branch of ``_search_index``): sklearn
    ``cosine_similarity`` (row-normalise both, dot) then ``argsort[::-1][:k]``.
    Also the semantics of ``faiss.IndexFlatIP.search`` on L2-normalised rows
    (src/retrieval.py:652-656) up to tie order.
    """
    sims = batch_cosine_similarity(np.asarray(q).reshape(1, -1), features)[0]
    idx = np.argsort(sims)[::-1][:top_k]
    return idx.astype(np.int64), sims[idx]


def ref_bank_similarities(ref_vectors: np.ndarray, query_vector: np.ndarray) -> np.ndarray:
    """src/ref_bank.py:475-484 (``_compute_similarities``), note the ``+1e-8``."""
    ref_vectors = np.asarray(ref_vectors, dtype=np.float64)
    query_vector = np.asarray(query_vector, dtype=np.float64)
    query_norm = np.linalg.norm(query_vector)
    ref_norms = np.linalg.norm(ref_vectors, axis=1)
    dots = np.dot(ref_vectors, query_vector)
    return dots / (ref_norms * query_norm + 1e-8)


def ref_bank_query_similar(ref_vectors: np.ndarray, query_vector: np.ndarray,
                           top_k: int = 10,
                           similarity_threshold: Optional[float] = None,
                           config_threshold: float = 0.9) -> Tuple[np.ndarray, np.ndarray]:
    """src/ref_bank.py:186-216 (``query_similar``): threshold, argsort desc, top-k.

    ``threshold = similarity_threshold or config`` (:191) -- a 0.0 argument
    falls through to the config value, as in the reference.
    """
    if len(ref_vectors) == 0:
        return np.zeros(0, np.int64), np.zeros(0)
    thr = similarity_threshold or config_threshold
    sims = ref_bank_similarities(ref_vectors, query_vector)
    valid = np.where(sims >= thr)[0]
    if valid.size == 0:
        return np.zeros(0, np.int64), np.zeros(0)
    order = valid[np.argsort(sims[valid])[::-1]]
    top = order[:top_k]
    return top.astype(np.int64), sims[top]


# --------------------------------------------------------------------------
# exp polarity: experiments/defenses/detector.py + consistency_checker.py
# --------------------------------------------------------------------------
def deduplicate_references(ref_feats: np.ndarray, dup_threshold: float = 0.95) -> List[int]:
    """experiments/defenses/detector.py:302-325 (``_deduplicate_references``):
    greedy, keeps the first of any pair with cosine > 0.95.  Returns positions
    into ``ref_feats`` that survive.  The reference re-encodes reference
    *images*; here the bank rows are those encodings."""
    n = len(ref_feats)
    if n == 0:
        return []
    keep = [0]
    for i in range(1, n):
        dup = False
        for j in keep:
            if cosine(ref_feats[i], ref_feats[j]) > dup_threshold:
                dup = True
                break
        if not dup:
            keep.append(i)
    return keep


def generate_retrieval_references(bank: np.ndarray, text_feats_normed: np.ndarray,
                                  retrieval_top_k: int = 10,
                                  cfg: RetrievalConfig = RetrievalConfig()) -> List[int]:
    """experiments/defenses/detector.py:184-204 (``_generate_retrieval_references``):
    retrieve per text variant (original first), concatenate, dedupe, cut to
    ``retrieval_top_k``.  Returns bank indices."""
    all_idx: List[int] = []
    for t in text_feats_normed:
        all_idx.extend(r['index'] for r in retrieve_references(bank, t, cfg))
    if not all_idx:
        return []
    keep = deduplicate_references(bank[np.asarray(all_idx)])
    return [all_idx[p] for p in keep][:retrieval_top_k]


def cross_modal_variance(*sims: float) -> float:
    """experiments/defenses/detector.py:295-300."""
    valid = [s for s in sims if s > 0]
    if len(valid) < 2:
        return 0.0
    return float(np.var(valid))


def compute_consistency_scores_exp(image_feat: np.ndarray,
                                   text_feats: np.ndarray,
                                   retrieval_ref_feats: Optional[np.ndarray] = None,
                                   generative_ref_feats: Optional[np.ndarray] = None) -> Dict[str, float]:
    """experiments/defenses/detector.py:228-293 (``_compute_consistency_scores``)."""
    scores: Dict[str, float] = {}
    text_feats = np.asarray(text_feats)
    s0 = cosine(image_feat, text_feats[0])                              # :238-241
    scores['original_similarity'] = s0
    if text_feats.shape[0] > 1:                                         # :244-252
        sv = [cosine(image_feat, t) for t in text_feats[1:]]
        scores['text_variant_consistency'] = float(np.mean(sv))
        scores['text_variant_std'] = float(np.std(sv))
    else:                                                               # :253-255
        scores['text_variant_consistency'] = s0
        scores['text_variant_std'] = 0.0
    if retrieval_ref_feats is not None and len(retrieval_ref_feats):    # :258-266
        sr = [cosine(image_feat, r) for r in retrieval_ref_feats]
        scores['retrieval_consistency'] = float(np.mean(sr))
        scores['retrieval_std'] = float(np.std(sr))
    else:
        scores['retrieval_consistency'] = 0.0
        scores['retrieval_std'] = 0.0
    if generative_ref_feats is not None and len(generative_ref_feats):  # :272-280
        sg = [cosine(image_feat, r) for r in generative_ref_feats]
        scores['generative_consistency'] = float(np.mean(sg))
        scores['generative_std'] = float(np.std(sg))
    else:
        scores['generative_consistency'] = 0.0
        scores['generative_std'] = 0.0
    scores['cross_modal_variance'] = cross_modal_variance(               # :286-291
        s0, scores['text_variant_consistency'],
        scores['retrieval_consistency'], scores['generative_consistency'])
    return scores


@dataclass
class ConsistencyCheckerOracle:
    """experiments/defenses/consistency_checker.py:31-272 (stateful)."""
    threshold: float = 0.5
    adaptive_threshold: bool = True
    voting_strategy: str = 'weighted'
    weights: Dict[str, float] = field(default_factory=lambda: {
        'original_similarity': 0.25, 'text_variant_consistency': 0.25,
        'retrieval_consistency': 0.25, 'generative_consistency': 0.25})
    threshold_history: List[float] = field(default_factory=list)

    _NAMES = ('original_similarity', 'text_variant_consistency',
              'retrieval_consistency', 'generative_consistency')

    def overall(self, s: Dict[str, float]) -> float:
        if self.voting_strategy == 'simple':                            # :130-145
            valid = [s.get(n, 0) for n in self._NAMES if s.get(n, 0) > 0]
            return float(np.mean(valid)) if valid else 0.0
        if self.voting_strategy == 'weighted':                          # :147-160
            ws = tw = 0.0
            for n, w in self.weights.items():
                if n in s and s[n] > 0:
                    ws += s[n] * w
                    tw += w
            return ws / tw if tw != 0 else 0.0
        if self.voting_strategy == 'adaptive':                          # :162-212
            rel = {'original_similarity': 1.0,
                   'text_variant_consistency': 1.0 / (1.0 + s.get('text_variant_std', 1.0)),
                   'retrieval_consistency': 1.0 / (1.0 + s.get('retrieval_std', 1.0)),
                   'generative_consistency': 1.0 / (1.0 + s.get('generative_std', 1.0))}
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
        raise ValueError(self.voting_strategy)

    def adaptive(self, s: Dict[str, float]) -> float:
        """:214-242"""
        thr = self.threshold
        if s.get('cross_modal_variance', 0) > 0.1:
            thr += 0.1
        avg_std = np.mean([s.get('text_variant_std', 0), s.get('retrieval_std', 0),
                           s.get('generative_std', 0)])
        if avg_std > 0.2:
            thr += 0.05
        if len(self.threshold_history) > 10:
            thr = 0.7 * thr + 0.3 * np.mean(self.threshold_history[-10:])
        return float(np.clip(thr, 0.1, 0.9))

    def confidence(self, overall: float, thr: float, s: Dict[str, float]) -> float:
        """:244-272"""
        dist = abs(overall - thr) / thr
        valid = [s.get(n, 0) for n in self._NAMES if s.get(n, 0) > 0]
        cons = 1.0 - np.std(valid) if len(valid) > 1 else 0.5
        var = 1.0 - min(s.get('cross_modal_variance', 0), 1.0)
        return float(np.clip(np.mean([dist, cons, var]), 0.0, 1.0))

    def make_decision(self, s: Dict[str, float]) -> Dict:
        """:74-117"""
        overall = self.overall(s)
        thr = self.adaptive(s) if self.adaptive_threshold else self.threshold
        is_adv = overall < thr                                           # :93
        conf = self.confidence(overall, thr, s)
        self.threshold_history.append(thr)                               # :105
        return {'is_adversarial': bool(is_adv), 'confidence': conf,
                'overall_score': float(overall), 'threshold': float(thr)}


# --------------------------------------------------------------------------
# batched driver used by parity tests: one call per query, reference order
# --------------------------------------------------------------------------
def detect_batch(image_feats: np.ndarray, text_feats: np.ndarray,
                 bank: Optional[np.ndarray] = None,
                 retrieval_cfg: RetrievalConfig = RetrievalConfig(),
                 retrieval_top_k: int = 10,
                 detection_threshold: float = 0.5,
                 checker: Optional[ConsistencyCheckerOracle] = None,
                 src_methods: Sequence[str] = ('text_variants', 'consistency'),
                 has_text_augmenter: bool = True) -> Dict[str, np.ndarray]:
    """Run both polarities for every query, in input order (the order matters
    for the stateful checker, consistency_checker.py:105,235).

    image_feats [B, D], text_feats [B, N+1, D] (any norm; cosines normalise),
    bank [R, D] L2-normalised rows or None.
    """
    B, N1, D = text_feats.shape
    checker = checker or ConsistencyCheckerOracle()
    out = {k: np.zeros(B) for k in (
        'original_similarity', 'variant_mean', 'variant_std', 'score_src',
        'retrieval_consistency', 'retrieval_std', 'cross_modal_variance',
        'overall_exp', 'threshold_exp', 'confidence_exp')}
    out['variant_similarities'] = np.zeros((B, N1 - 1))
    out['is_adv_src'] = np.zeros(B, bool)
    out['is_adv_exp'] = np.zeros(B, bool)
    out['retrieval_indices'] = np.full((B, retrieval_top_k), -1, np.int64)
    for b in range(B):
        r = detect_adversarial_src(image_feats[b], text_feats[b], methods=src_methods,
                                   detection_threshold=detection_threshold, has_text_augmenter=has_text_augmenter)
        out['score_src'][b] = r['aggregated_score']
        out['is_adv_src'][b] = r['is_adversarial']
        refs_idx: List[int] = []
        if bank is not None and len(bank):
            tn = l2_normalize(text_feats[b]).astype(bank.dtype)
            refs_idx = generate_retrieval_references(bank, tn, retrieval_top_k, retrieval_cfg)
        ref_feats = bank[np.asarray(refs_idx, np.int64)] if refs_idx else None
        s = compute_consistency_scores_exp(image_feats[b], text_feats[b], ref_feats, None)
        d = checker.make_decision(s)
        out['original_similarity'][b] = s['original_similarity']
        out['variant_mean'][b] = s['text_variant_consistency']
        out['variant_std'][b] = s['text_variant_std']
        if N1 > 1:
            out['variant_similarities'][b] = [cosine(image_feats[b], t) for t in text_feats[b, 1:]]
        out['retrieval_consistency'][b] = s['retrieval_consistency']
        out['retrieval_std'][b] = s['retrieval_std']
        out['cross_modal_variance'][b] = s['cross_modal_variance']
        out['overall_exp'][b] = d['overall_score']
        out['threshold_exp'][b] = d['threshold']
        out['confidence_exp'][b] = d['confidence']
        out['is_adv_exp'][b] = d['is_adversarial']
        out['retrieval_indices'][b, :len(refs_idx)] = refs_idx
    return out


def detection_metrics(scores: np.ndarray, labels: np.ndarray) -> Dict[str, float]:
    """src/utils/metrics.py:286-329 (``compute_detection_metrics``): sklearn
    ``roc_auc_score`` + Youden-J threshold; label 1 = adversarial."""
    from sklearn.metrics import roc_auc_score, roc_curve, accuracy_score, f1_score
    scores = np.asarray(scores, dtype=np.float64)
    labels = np.asarray(labels).astype(int)
    fpr, tpr, thr = roc_curve(labels, scores, pos_label=1)
    auc = roc_auc_score(labels, scores)
    j = int(np.argmax(tpr - fpr))
    pred = (scores >= thr[j]).astype(int)
    return {'auc': float(auc), 'threshold': float(thr[j]),
            'accuracy': float(accuracy_score(labels, pred)),
            'f1_score': float(f1_score(labels, pred, zero_division=0))}
