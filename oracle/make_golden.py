#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's own importable files.

TEST INFRASTRUCTURE ONLY.  Run in the BUILD container (where /root/reference is
mounted read-only); the reference never travels to the GPU box, only the small
vectors written here do.  Three reference files import cleanly by path
(SURVEY.md section 8c): src/ref_bank.py, src/utils/metrics.py,
experiments/defenses/consistency_checker.py.  They are executed as they are;
nothing is copied from them.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
"""
import importlib.util
import json
import os
import sys
from pathlib import Path

import numpy as np

sys.dont_write_bytecode = True
REF = Path(os.environ.get("TVC_REFERENCE", "/root/reference"))
OUT = Path(__file__).resolve().parents[1] / "tests" / "golden"


def load(name: str, rel: str):
    spec = importlib.util.spec_from_file_location(name, REF / rel)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def golden_ref_bank():
    rb = load("ref_ref_bank", "src/ref_bank.py")
    cfg = rb.ReferenceBankConfig(persistence_enabled=False, auto_clustering=False, clustering_method="none",
                                 similarity_threshold=0.8, save_path=str(REF / "cache" / "ref_bank"))
    bank = rb.ReferenceBank(cfg)                       # reads the reference's own 20 x 512 data fixture
    V = np.stack([r.vector for r in bank.references])
    assert V.shape == (20, 512), V.shape
    rng = np.random.default_rng(1234)
    queries = []
    for i in range(12):
        if i < 4:
            q = rng.standard_normal(512)                               # unrelated query: nothing above threshold
        else:
            q = V[rng.integers(20)] + rng.uniform(0.2, 1.2) * np.linalg.norm(V[0]) / np.sqrt(512) * rng.standard_normal(512)
        queries.append(q)
    Q = np.stack(queries)
    sims = np.stack([bank._compute_similarities(q) for q in Q])        # src/ref_bank.py:462-484
    res = {}
    for thr_name, thr in (("default", None), ("t05", 0.5), ("t0", 0.0)):
        idx = np.full((len(Q), 10), -1, np.int64)
        val = np.zeros((len(Q), 10))
        for i, q in enumerate(Q):
            got = bank.query_similar(q, top_k=10, similarity_threshold=thr)     # :172-224
            for j, (item, s) in enumerate(got):
                idx[i, j] = next(k for k, r in enumerate(bank.references) if r is item)
                val[i, j] = s
        res[f"idx_{thr_name}"] = idx
        res[f"sim_{thr_name}"] = val
    pair = np.array([bank._cosine_similarity(Q[i], V[i]) for i in range(len(Q))])   # :486-503
    np.savez_compressed(OUT / "ref_bank.npz", vectors=V, queries=Q, similarities=sims, pair_cosine=pair,
                        config_threshold=np.array(cfg.similarity_threshold), **res)
    print("ref_bank.npz", V.shape, Q.shape, int((res["idx_default"] >= 0).sum()), "hits at default threshold")


def golden_metrics():
    import torch
    m = load("ref_metrics", "src/utils/metrics.py")
    rng = np.random.default_rng(99)
    x = rng.standard_normal((9, 512)) * 3.0
    y = rng.standard_normal((14, 512)) * 0.25
    cos_np = m.SimilarityCalculator.batch_cosine_similarity(x, y)                       # :160-164
    cos_t = m.SimilarityCalculator.batch_cosine_similarity(torch.from_numpy(x).float(), torch.from_numpy(y).float())
    one = np.array([m.SimilarityCalculator.cosine_similarity(x[i], y[i]) for i in range(9)])   # :116-141
    zero = m.SimilarityCalculator.cosine_similarity(np.zeros(512), y[0])
    labels = (rng.random(400) < 0.5).astype(int)
    scores = rng.standard_normal(400) * 0.2 + labels * 0.25
    dm = m.DetectionEvaluator.compute_detection_metrics(scores, labels)                 # :286-329
    np.savez_compressed(OUT / "metrics.npz", x=x, y=y, cos_numpy=cos_np, cos_torch=cos_t, cos_pairs=one,
                        cos_zero=np.array(zero), labels=labels, scores=scores,
                        auc=np.array(dm.auc), threshold=np.array(dm.threshold), accuracy=np.array(dm.accuracy),
                        precision=np.array(dm.precision), recall=np.array(dm.recall), f1=np.array(dm.f1_score),
                        fpr_at_95_tpr=np.array(dm.fpr_at_95_tpr), confusion=np.asarray(dm.confusion_matrix))
    print("metrics.npz auc", dm.auc)


def golden_checker():
    cc = load("ref_checker", "experiments/defenses/consistency_checker.py")
    rng = np.random.default_rng(7)
    T = 40
    names = ("original_similarity", "text_variant_consistency", "text_variant_std", "retrieval_consistency",
             "retrieval_std", "generative_consistency", "generative_std", "cross_modal_variance")
    S = np.zeros((T, len(names)))
    S[:, 0] = rng.uniform(-0.1, 0.9, T)
    S[:, 1] = S[:, 0] + rng.normal(0, 0.1, T)
    S[:, 2] = rng.uniform(0, 0.4, T)
    S[:, 3] = np.where(rng.random(T) < 0.3, 0.0, rng.uniform(0.1, 0.9, T))      # 0 = module disabled
    S[:, 4] = rng.uniform(0, 0.4, T)
    S[:, 5] = np.where(rng.random(T) < 0.6, 0.0, rng.uniform(0.1, 0.9, T))
    S[:, 6] = rng.uniform(0, 0.3, T)
    S[:, 7] = rng.uniform(0, 0.2, T)
    out = {"scores": S, "names": np.array(names)}
    for strat in ("weighted", "simple", "adaptive"):
        for adaptive in (True, False):
            chk = cc.ConsistencyChecker(threshold=0.5, adaptive_threshold=adaptive, voting_strategy=strat)
            rows = []
            for t in range(T):
                d = chk.make_decision(dict(zip(names, S[t].tolist())))       # consistency_checker.py:74-117
                rows.append([d["overall_score"], d["threshold"], d["confidence"], float(d["is_adversarial"])])
            out[f"{strat}_{'adaptive' if adaptive else 'fixed'}"] = np.array(rows)
    np.savez_compressed(OUT / "consistency_checker.npz", **out)
    print("consistency_checker.npz", T, "decisions x 6 configurations")


if __name__ == "__main__":
    OUT.mkdir(parents=True, exist_ok=True)
    golden_ref_bank()
    golden_metrics()
    golden_checker()
