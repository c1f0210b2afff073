"""The index oracle (oracle/vsr_index_oracle.c: pgvector's IVFFlat and HNSW restated) against the reference's own
expectations: index-order known answers (pgvector/test/expected/hnsw_vector.out:3-90, ivfflat_vector.out) and the recall
thresholds of its TAP tests (t/012_hnsw_vector_build_recall.pl:94, t/005_ivfflat_query_recall.pl:31-41)."""
import json
import os

import numpy as np
import pytest

from oracle.oracle import HnswIndex, IvfIndex


@pytest.fixture(scope="module")
def known(golden_dir):
    with open(os.path.join(golden_dir, "pgvector_known_answers.json")) as f:
        return json.load(f)


def test_index_order_known_answers(oracle, known):
    """ORDER BY val <op> '[3,3,3]' through an index on the 4-row table of hnsw_vector.sql / ivfflat_vector.sql."""
    o = known["ordering"]
    rows = np.asarray(o["rows"], dtype=np.float32)
    q = np.asarray(o["query"], dtype=np.float32)
    for metric in ("l2", "ip"):
        h = HnswIndex(oracle, metric, rows, m=16, ef_construction=64)
        got, _, _, _ = h.search(q, 40)
        assert rows[got].tolist() == o[metric]
        ivf = IvfIndex(oracle, metric, rows, lists=1)
        idx, _ = ivf.search(q, 4, probes=1)
        assert rows[idx].tolist() == o[metric]
    # cosine opclass: rows are normalised on the way in and the zero vector is not indexed (hnswutils.c:167,401-425)
    nz = rows[np.linalg.norm(rows, axis=1) > 0]
    unit = np.stack([oracle.l2_normalize(r) for r in nz])
    h = HnswIndex(oracle, "cosine", unit)
    got, _, _, _ = h.search(oracle.l2_normalize(q), 40)
    assert nz[got].tolist() == o["cosine_index"]


def test_hnsw_recall_threshold_of_the_reference_tap_test(oracle):
    """t/012_hnsw_vector_build_recall.pl: 10000 rows of random() * random() in 3-d, 20 queries, LIMIT 20, default
    m = 16 / ef_construction = 64 / ef_search = 40: recall >= 0.99 for <-> (>= 0.97 for <#>)."""
    rng = np.random.default_rng(12)
    x = (rng.random((10000, 3)) * rng.random((10000, 3))).astype(np.float32)
    qs = rng.random((20, 3)).astype(np.float32)
    for metric, floor in (("l2", 0.99), ("ip", 0.97)):
        h = HnswIndex(oracle, metric, x, m=16, ef_construction=64, seed=5)
        correct = total = 0
        for q in qs:
            got, _, _, _ = h.search(q, 40)
            exact, _ = oracle.filtered_topk(metric, x, q, 20)
            correct += len(set(got[:20].tolist()) & set(exact.tolist()))
            total += 20
        assert correct / total >= floor, (metric, correct / total)


def test_hnsw_duplicates_share_an_element(oracle):
    """hnswbuild.c:309-355: identical vectors become heap TIDs of one element (at most 10), and a scan returns all of them."""
    rng = np.random.default_rng(3)
    base = rng.integers(0, 50, (300, 8)).astype(np.float32)
    x = np.concatenate([base, base[:40], base[:40], base[:5]] + [base[:1]] * 12)      # row 0's vector 16 times
    h = HnswIndex(oracle, "l2", x, m=8, ef_construction=32, seed=2)
    assert h.n_elem < len(x)
    ex = h.export()
    assert ex["tid_count"].max() == 10 and ex["tid_count"].sum() == len(x)
    got, dist, _, _ = h.search(base[0], 40)
    zero = got[dist == 0]
    assert sorted(zero.tolist()) == sorted(np.flatnonzero((x == base[0]).all(1)).tolist())


def test_ivfflat_self_query_recall(oracle):
    """t/005_ivfflat_query_recall.pl: with the default lists = 100 and probes = 1 a row is its own nearest neighbour."""
    rng = np.random.default_rng(7)
    x = rng.random((20000, 3)).astype(np.float32)
    ivf = IvfIndex(oracle, "l2", x, lists=100, seed=3)
    assert np.bincount(ivf.assign, minlength=100).min() >= 0 and ivf.assign.max() < 100
    for i in rng.integers(0, len(x), 20):
        idx, dist = ivf.search(x[i], 1, probes=1)
        assert (x[idx[0]] == x[i]).all() and dist[0] == 0
    # probes = lists is the exact scan
    q = rng.random(3).astype(np.float32)
    a, da = ivf.search(q, 50, probes=100)
    b, db = oracle.filtered_topk("l2", x, q, 50)
    assert a.tolist() == b.tolist() and da.tolist() == db.tolist()
    # spherical k-means for the inner-product / cosine opclasses: unit centres
    unit = x / np.linalg.norm(x, axis=1, keepdims=True)
    ivf2 = IvfIndex(oracle, "cosine", unit, lists=20, seed=4)
    np.testing.assert_allclose(np.linalg.norm(ivf2.centers, axis=1), 1.0, atol=1e-5)


def test_predicate_aware_walk_of_the_oracle(oracle):
    """search_layer_pa (the restatement of K4's predicate-aware walk): with every row allowed it is HnswSearchLayer itself;
    with few rows allowed everything it returns besides the entry point is allowed, and its recall against the exact
    filtered scan beats filtering the plain walk's results at the same ef_search."""
    rng = np.random.default_rng(5)
    n, dim, k, ef = 6000, 16, 10, 40
    x = np.clip(np.rint(np.abs(rng.normal(0, 45, (n, dim)))), 0, 255).astype(np.float32)
    h = HnswIndex(oracle, "l2", x, m=8, ef_construction=32, seed=2)
    allowed = (rng.random(n) < 0.1).astype(np.uint8)
    doc = np.arange(n, dtype=np.int32) + 1
    blk = np.arange(n, dtype=np.int64) + 1
    hits_pa = hits_plain = 0
    for i in range(20):
        q = x[rng.integers(0, n)] + rng.integers(-2, 3, dim).astype(np.float32)
        r_all, d_all, _, nv_all = h.search_predicate_aware(q, ef, np.ones(n, dtype=np.uint8))
        r_pl, d_pl, _, nv_pl = h.search(q, ef)
        np.testing.assert_array_equal(r_all, r_pl)
        np.testing.assert_array_equal(d_all, d_pl)
        assert nv_all == nv_pl
        r_pa, d_pa, _, _ = h.search_predicate_aware(q, ef, allowed)
        assert (allowed[r_pa] != 0).sum() >= r_pa.size - 1            # (the entry point of layer 0 may be a row that is not allowed)
        assert (np.diff(d_pa) >= 0).all()
        exact, _ = oracle.filtered_topk("l2", x, q, k, doc, blk, allowed)
        hits_pa += len(set(r_pa[allowed[r_pa] != 0][:k].tolist()) & set(exact.tolist()))
        hits_plain += len(set(r_pl[allowed[r_pl] != 0][:k].tolist()) & set(exact.tolist()))
    assert hits_pa > hits_plain and hits_pa >= 0.8 * 20 * k, (hits_pa, hits_plain)
