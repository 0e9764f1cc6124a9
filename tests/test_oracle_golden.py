"""CPU: the oracle restatement vs golden vectors produced by the reference's OWN
importable code (oracle/make_golden.py: src/ref_bank.py, src/utils/metrics.py,
experiments/defenses/consistency_checker.py executed in the build container),
including the reference's 20 x 512 ``cache/ref_bank/references.json`` data fixture."""
from pathlib import Path

import numpy as np
import pytest

from oracle import tvc_oracle as O

G = Path(__file__).parent / "golden"


def test_ref_bank_similarities_match_reference():
    g = np.load(G / "ref_bank.npz")
    V, Q = g["vectors"], g["queries"]
    assert V.shape == (20, 512)
    for i, q in enumerate(Q):
        np.testing.assert_allclose(O.ref_bank_similarities(V, q), g["similarities"][i], rtol=0, atol=1e-12)


@pytest.mark.parametrize("name,thr", [("default", None), ("t05", 0.5), ("t0", 0.0)])
def test_ref_bank_query_similar_matches_reference(name, thr):
    g = np.load(G / "ref_bank.npz")
    V, Q = g["vectors"], g["queries"]
    cfg_thr = float(g["config_threshold"])
    hits = 0
    for i, q in enumerate(Q):
        idx, sim = O.ref_bank_query_similar(V, q, top_k=10, similarity_threshold=thr, config_threshold=cfg_thr)
        want = g[f"idx_{name}"][i]
        n = int((want >= 0).sum())
        assert idx.tolist() == want[:n].tolist()
        np.testing.assert_allclose(sim, g[f"sim_{name}"][i][:n], atol=1e-12)
        hits += n
    assert hits > 0


def test_metrics_cosine_matches_reference():
    g = np.load(G / "metrics.npz")
    got = O.batch_cosine_similarity(g["x"], g["y"])
    np.testing.assert_allclose(got, g["cos_numpy"], atol=1e-12)
    np.testing.assert_allclose(got, g["cos_torch"], atol=5e-7)          # the reference's torch branch is fp32
    pairs = [O.cosine(g["x"][i], g["y"][i]) for i in range(9)]
    np.testing.assert_allclose(pairs, g["cos_pairs"], atol=1e-12)


def test_detection_metrics_match_reference():
    g = np.load(G / "metrics.npz")
    m = O.detection_metrics(g["scores"], g["labels"])
    assert abs(m["auc"] - float(g["auc"])) < 1e-12
    assert abs(m["threshold"] - float(g["threshold"])) < 1e-12
    assert abs(m["accuracy"] - float(g["accuracy"])) < 1e-12
    assert abs(m["f1_score"] - float(g["f1"])) < 1e-12


@pytest.mark.parametrize("strategy", ["weighted", "simple", "adaptive"])
@pytest.mark.parametrize("adaptive", [True, False])
def test_consistency_checker_matches_reference(strategy, adaptive):
    """Bit-for-bit on overall score / threshold / decision given identical score
    dicts and call order (the checker is stateful)."""
    g = np.load(G / "consistency_checker.npz")
    names = [str(n) for n in g["names"]]
    want = g[f"{strategy}_{'adaptive' if adaptive else 'fixed'}"]
    chk = O.ConsistencyCheckerOracle(threshold=0.5, adaptive_threshold=adaptive, voting_strategy=strategy)
    for t, row in enumerate(g["scores"]):
        d = chk.make_decision(dict(zip(names, row.tolist())))
        assert d["overall_score"] == want[t, 0]
        assert d["threshold"] == want[t, 1]
        assert abs(d["confidence"] - want[t, 2]) < 1e-15
        assert float(d["is_adversarial"]) == want[t, 3]


def test_text_variant_arithmetic_hand_values():
    """src/detector.py:479-485 on a hand-computed case."""
    s, d = O.text_variant_score(0.30, [0.20, 0.40])
    # mean 0.30, std 0.10 -> consistency 1.0, variability 0.9 -> 1 - (0.7 + 0.27) = 0.03
    assert abs(s - 0.03) < 1e-12 and abs(d["std_variant_similarity"] - 0.1) < 1e-12
    assert O.aggregate_scores({"text_variants": 0.03, "consistency": 0.70}) == pytest.approx((0.03 * 0.4 + 0.7 * 0.2) / 0.6)
    assert O.aggregate_scores({}) == 0.0
    assert O.cross_modal_variance(0.5, 0.0, -0.1, 0.0) == 0.0          # < 2 valid (> 0) entries


def test_retrieve_references_semantics():
    rng = np.random.default_rng(0)
    bank = rng.standard_normal((500, 64)).astype(np.float32)
    bank /= np.linalg.norm(bank, axis=1, keepdims=True)
    q = bank[17] + 0.05 * rng.standard_normal(64).astype(np.float32)
    q /= np.linalg.norm(q)
    refs = O.retrieve_references(bank, q)
    assert refs[0]["index"] == 17 and all(r["similarity"] >= 0.3 for r in refs) and len(refs) <= 5
    sims = bank @ q
    assert [r["index"] for r in refs] == [i for i in np.argsort(-sims)[:5] if sims[i] >= 0.3]
    assert O.retrieve_references(bank[:0], q) == []                     # empty bank
    idx, s = O.search_index_exact(bank, q, 7)
    assert idx.tolist() == np.argsort(-sims)[:7].tolist()
