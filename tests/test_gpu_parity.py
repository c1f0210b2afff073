"""GPU parity: libvsrbac (HIP, through the C ABI) vs the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): returned row ids bit-exact at full recall, fp32 distances within 1e-4.
Integer-valued (SIFT-like) data has exact fp32 sums in any order, so ids AND distances must be identical
(tie groups included, tie rule = (distance, document_id, block_id)).  Real-valued data is summation-order
ambiguous in the reference itself (-fassociative-math) and is checked with
helpers.assert_valid_topk at 1e-4.
"""
import json
import math
import os

from types import SimpleNamespace

import numpy as np
import pytest

from helpers import assert_valid_topk, sift_like

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def ctx():
    import vsrbac
    c = vsrbac.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def known(golden_dir):
    with open(os.path.join(golden_dir, "pgvector_known_answers.json")) as f:
        return json.load(f)


def _ids(n, rows_per_doc):
    return (np.arange(n) + 1).astype(np.int64), (np.arange(n) // rows_per_doc + 1).astype(np.int32)


def _expect_exact(oracle, res, qi, metric, x, q, k, doc, blk, mask=None):
    idx, dist = oracle.filtered_topk(metric, x, q, k, doc, blk, mask)
    m = res.counts[qi]
    assert m == idx.size, (m, idx.size)
    np.testing.assert_array_equal(res.rows[qi, :m], idx)
    np.testing.assert_array_equal(res.block_ids[qi, :m], blk[idx])
    np.testing.assert_array_equal(res.doc_ids[qi, :m], doc[idx])
    np.testing.assert_array_equal(res.dist[qi, :m], dist.astype(np.float32))
    assert (res.block_ids[qi, m:] == -1).all() and np.isinf(res.dist[qi, m:]).all()


# ---------------------------------------------------------------------------------------------
# pgvector known answers through the GPU path
# ---------------------------------------------------------------------------------------------
def test_pair_distance_known_answers(ctx, known):
    import vsrbac
    metric = {"l2_distance": "l2", "negative_inner_product": "ip", "cosine_distance": "cosine", "l1_distance": "l1"}
    for fn, a, b, want in known["distances"]:
        if fn == "inner_product":          # SQL function inner_product = -(<#>)
            fn, want = "negative_inner_product", (-want if not isinstance(want, str) else want)
            if want == "Infinity":
                want = "-Infinity"
        if isinstance(want, str) and want.startswith("ERROR:"):
            with pytest.raises(vsrbac.VsrError) as e:
                ctx.pair_distances(metric[fn], [a], b)
            assert "ERROR:  " + str(e.value) == want
            continue
        got = ctx.pair_distances(metric[fn], [a], b)[0]
        if want == "Infinity":
            assert math.isinf(got) and got > 0
        elif want == "-Infinity":
            assert math.isinf(got) and got < 0
        elif want == "NaN":
            assert math.isnan(got)
        else:
            assert got == want


def test_opclass_support_functions(ctx, oracle, known):
    """vector_norm / l2_normalize / vector_spherical_distance on the device: pgvector's regress expectations
    (vector_type.out:337-371, 537-565) and the oracle's restatement of vector.c:692-711,756-808 on random rows."""
    import vsrbac
    for v, want in known["norms"]:
        assert ctx.vector_norms([v])[0] == want
    assert float(np.float32(ctx.vector_norms([[3e37, 4e37]])[0])) == float(np.float32(5e37))
    np.testing.assert_array_equal(ctx.l2_normalize([[3, 4]])[0], np.asarray([0.6, 0.8], dtype=np.float32))
    np.testing.assert_array_equal(ctx.l2_normalize([[3, 0]])[0], np.asarray([1, 0], dtype=np.float32))
    np.testing.assert_array_equal(ctx.l2_normalize([[0, 0.1]])[0], np.asarray([0, 1], dtype=np.float32))
    np.testing.assert_array_equal(ctx.l2_normalize([[0, 0]])[0], np.asarray([0, 0], dtype=np.float32))
    np.testing.assert_array_equal(ctx.l2_normalize([[3e38]])[0], np.asarray([1], dtype=np.float32))
    rng = np.random.default_rng(91)
    x = rng.normal(size=(50, 300)).astype(np.float32)
    got_n = ctx.vector_norms(x)
    got_u = ctx.l2_normalize(x)
    for i in range(50):
        assert abs(got_n[i] - oracle.vector_norm(x[i])) <= 1e-12 * got_n[i]
        np.testing.assert_allclose(got_u[i], oracle.l2_normalize(x[i]), rtol=2e-7, atol=0)
    y = got_u[::-1].copy()
    got_s = ctx.spherical_distances(got_u, y)
    for i in range(50):
        assert abs(got_s[i] - oracle.pair("spherical_distance", got_u[i], y[i])) <= 1e-6
    assert ctx.spherical_distances([[1.0, 0.0]], [1.0, 0.0])[0] == 0.0           # clamp, then acos(1) / pi
    assert ctx.spherical_distances([[1.0, 0.0]], [-1.0, 0.0])[0] == 1.0
    with pytest.raises(vsrbac.VsrError) as e:
        ctx.spherical_distances([[1.0, 0.0]], [1.0])
    assert str(e.value) == "different vector dimensions 2 and 1"


def test_ordering_known_answers(ctx, known):
    o = known["ordering"]
    rows = np.asarray(o["rows"], dtype=np.float32)
    corpus = ctx.load_corpus(rows)
    for metric in ("l2", "ip", "l1"):
        res = corpus.search(o["query"], 4, metric)
        assert rows[res.rows[0]].tolist() == o[metric]
    res = corpus.search(o["query"], 4, "cosine")
    assert rows[res.rows[0, :3]].tolist() == o["cosine_index"]
    assert math.isnan(res.dist[0, 3]) and res.rows[0, 3] == 0       # zero vector: NaN, sorted last
    corpus.free()


def test_dimension_mismatch_message(ctx):
    import vsrbac
    corpus = ctx.load_corpus(np.zeros((4, 2), dtype=np.float32))
    with pytest.raises(vsrbac.VsrError) as e:
        corpus.search([[1.0]], 1)
    assert str(e.value) == "different vector dimensions 2 and 1"     # vector.c:60-67
    assert e.value.status == 2
    corpus.free()


# ---------------------------------------------------------------------------------------------
# config 1 shape: 10k x 128, k = 10, no filter; plus k = 100
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k", [1, 10, 100])
def test_sift10k_unfiltered_bit_exact(ctx, oracle, k):
    rng = np.random.default_rng(1)
    n = 10_000
    x = sift_like(rng, n)
    blk, doc = _ids(n, 100)
    corpus = ctx.load_corpus(x, blk, doc)
    qrows = rng.integers(0, n, 8)
    res = corpus.search(x[qrows], k, "l2")
    for i, qr in enumerate(qrows):
        _expect_exact(oracle, res, i, "l2", x, x[qr], k, doc, blk)
        assert res.rows[i, 0] == qr or res.dist[i, 0] == 0
    corpus.free()


def test_ties_follow_doc_block_order(ctx, oracle):
    """Many exact ties (duplicated rows, shuffled identity): order must be (dist, document_id, block_id)."""
    rng = np.random.default_rng(2)
    base = sift_like(rng, 50)
    x = np.repeat(base, 40, axis=0)                      # 2000 rows, each vector 40 times
    perm = rng.permutation(x.shape[0])
    x = x[perm]
    blk = rng.permutation(x.shape[0]).astype(np.int64) + 1        # identities unrelated to row order
    doc = rng.integers(1, 30, x.shape[0]).astype(np.int32)
    corpus = ctx.load_corpus(x, blk, doc)
    res = corpus.search(base[:5], 100, "l2")
    for i in range(5):
        _expect_exact(oracle, res, i, "l2", x, base[i], 100, doc, blk)
    corpus.free()


def test_screening_gap_inside_error_bound_falls_back_to_exact(ctx, oracle):
    """Shared passes run on the matrix cores (K2: MFMA screening keeps 2k candidates, K5r re-ranks exactly).  When the
    2k-th candidate ties with the k-th result the re-rank cannot prove exactness: the query must be flagged and re-run
    on the exact path, and the answer must still follow the (distance, document_id, block_id) order."""
    rng = np.random.default_rng(12)
    base = 2.0 * sift_like(rng, 8)                       # even integers up to 510: exact in bf16, but not 0..255, so the
                                                         # int8 planes (exact screening, nothing to flag) do not apply
    x = np.repeat(base, 300, axis=0)                     # every vector 300 times: ties far beyond 2k = 200
    x = x[rng.permutation(x.shape[0])]
    n = x.shape[0]
    blk = (rng.permutation(n) + 1).astype(np.int64)
    doc = rng.integers(1, 20, n).astype(np.int32)
    corpus = ctx.load_corpus(x, blk, doc)
    before, _ = ctx.screening_check(0)
    res = corpus.search(base[:6], 100, "l2")             # 6 unfiltered queries share one pass
    after, _ = ctx.screening_check(0)
    for i in range(6):
        _expect_exact(oracle, res, i, "l2", x, base[i], 100, doc, blk)
    assert after > before, "the tie-saturated queries should have been flagged by the re-rank"
    # with screening disabled the same call never flags
    ctx.set_screening(False)
    res2 = corpus.search(base[:6], 100, "l2")
    ctx.set_screening(True)
    assert ctx.screening_check(0)[0] == after
    np.testing.assert_array_equal(res2.rows, res.rows)
    corpus.free()


def test_device_api_flags_cannot_be_ignored_and_exact_variant_reruns(ctx, oracle):
    """vsr_search_device is asynchronous and cannot re-run anything itself: a flagged query reports a NEGATIVE count
    (-1 - rows), so a caller that never calls vsr_screening_check still cannot publish its rows as a result.
    vsr_search_device_exact waits, re-runs the flagged queries on the exact path and patches the outputs."""
    import torch
    rng = np.random.default_rng(12)
    base = 2.0 * sift_like(rng, 8)                       # even integers up to 510: exact in bf16, but not 0..255, so the
                                                         # int8 planes (exact screening, nothing to flag) do not apply
    x = np.repeat(base, 300, axis=0)                     # every vector 300 times: ties far beyond 2k = 200
    x = x[rng.permutation(x.shape[0])]
    n = x.shape[0]
    blk = (rng.permutation(n) + 1).astype(np.int64)
    doc = rng.integers(1, 20, n).astype(np.int32)
    corpus = ctx.load_corpus(x, blk, doc)
    dev = torch.device("cuda", 0)
    nq, k = 6, 100
    q = torch.from_numpy(base[:nq].copy()).to(dev)
    o_blk = torch.empty((nq, k), dtype=torch.int64, device=dev)
    o_doc = torch.empty((nq, k), dtype=torch.int32, device=dev)
    o_row = torch.empty((nq, k), dtype=torch.int64, device=dev)
    o_dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    o_cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    args = (o_blk.data_ptr(), o_doc.data_ptr(), o_row.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr())
    corpus.search_device(q.data_ptr(), nq, k, "l2", None, *args)
    _, flags = ctx.screening_check(nq)                   # synchronises
    cnt = o_cnt.cpu().numpy()
    assert flags.any(), "the tie-saturated queries should have been flagged"
    assert ((cnt < 0) == (flags != 0)).all(), (cnt, flags)
    assert (cnt[flags != 0] == -1 - k).all()
    n_rerun = corpus.search_device_exact(q.data_ptr(), nq, k, "l2", None, *args)
    assert n_rerun == int((flags != 0).sum())
    cnt = o_cnt.cpu().numpy()
    assert (cnt == k).all()
    res = SimpleNamespace(block_ids=o_blk.cpu().numpy(), doc_ids=o_doc.cpu().numpy(), rows=o_row.cpu().numpy(),
                          dist=o_dist.cpu().numpy(), counts=cnt)
    for i in range(nq):
        _expect_exact(oracle, res, i, "l2", x, base[i], k, doc, blk)
    corpus.free()


# ---------------------------------------------------------------------------------------------
# RBAC: fixtures produced by the reference's generators (tests/golden/make_rbac_fixture.py)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,rows_per_doc", [("rbac_tree_small.json", 20), ("rbac_random_small.json", 7)])
def test_rbac_prefilter_and_postfilter(ctx, oracle, golden_dir, name, rows_per_doc):
    import vsrbac
    with open(os.path.join(golden_dir, name)) as f:
        fx = json.load(f)
    rng = np.random.default_rng(3)
    ndocs = fx["params"]["num_docs"]
    n = ndocs * rows_per_doc
    x = sift_like(rng, n)
    blk, doc = _ids(n, rows_per_doc)
    corpus = ctx.load_corpus(x, blk, doc)
    corpus.load_rbac(fx["user_roles"], fx["permissions"])
    users = list(range(1, fx["num_users"] + 1))
    qrows = rng.integers(0, n, len(users))
    masks = [oracle.user_row_mask(u, fx["user_roles"], fx["permissions"], doc) for u in users]
    for mode in (vsrbac.RANGES, vsrbac.BITMAP):
        filters = [corpus.filter_for_user(u, mode) for u in users]
        for u, f, m in zip(users, filters, masks):
            assert f.allowed_rows == int(m.sum()) == rows_per_doc * len(fx["visible_docs"][str(u)])
        res = corpus.search(x[qrows], 100, "l2", filters)          # one batch, mixed users (shared passes)
        for i in range(len(users)):
            _expect_exact(oracle, res, i, "l2", x, x[qrows[i]], 100, doc, blk, masks[i])
        one = corpus.search(x[qrows[:3]], 10, "l2", filters[:3])   # and small k, few queries
        for i in range(3):
            _expect_exact(oracle, one, i, "l2", x, x[qrows[i]], 10, doc, blk, masks[i])
    # a user the tables do not know sees nothing
    res = corpus.search(x[:1], 10, "l2", [corpus.filter_for_user(10_000_000, vsrbac.RANGES)])
    assert res.counts[0] == 0 and (res.block_ids[0] == -1).all()
    corpus.free()


def test_bytemask_and_document_filters(ctx, oracle, golden_dir):
    import vsrbac
    with open(os.path.join(golden_dir, "rbac_tree_small.json")) as f:
        fx = json.load(f)
    rng = np.random.default_rng(4)
    rows_per_doc = 10
    n = fx["params"]["num_docs"] * rows_per_doc
    x = sift_like(rng, n)
    blk, doc = _ids(n, rows_per_doc)
    corpus = ctx.load_corpus(x, blk, doc)
    corpus.load_rbac(fx["user_roles"], fx["permissions"])
    q = x[rng.integers(0, n, 4)]
    # arbitrary predicate: byte-per-row mask (ACORN / logical-partition convention)
    mask = (rng.random(n) < 0.05).astype(np.uint8)
    for mode in (vsrbac.RANGES, vsrbac.BITMAP):
        f = corpus.filter_from_bytemask(mask, mode)
        assert f.allowed_rows == int(mask.sum())
        res = corpus.search(q, 50, "l2", f)
        for i in range(4):
            _expect_exact(oracle, res, i, "l2", x, q[i], 50, doc, blk, mask)
        f.free()
    # dynamic partition = a document set; pure (no per-row test) and impure (with the user's permission)
    part_docs = rng.choice(np.arange(1, fx["params"]["num_docs"] + 1), 80, replace=False)
    in_part = np.isin(doc, part_docs).astype(np.uint8)
    f = corpus.filter_from_documents(part_docs)
    res = corpus.search(q, 30, "l2", f)
    for i in range(4):
        _expect_exact(oracle, res, i, "l2", x, q[i], 30, doc, blk, in_part)
    user = 7
    um = oracle.user_row_mask(user, fx["user_roles"], fx["permissions"], doc)
    f2 = corpus.filter_from_documents(part_docs, user_id=user)
    res = corpus.search(q, 30, "l2", f2)
    for i in range(4):
        _expect_exact(oracle, res, i, "l2", x, q[i], 30, doc, blk, in_part & um)
    corpus.free()


# ---------------------------------------------------------------------------------------------
# every kernel shape (dimension classes), every metric, real-valued data -> tolerance
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dim", [1, 3, 9, 16, 17, 50, 64, 100, 128, 129, 200, 256, 300, 512, 768, 960, 1024, 1536, 2000, 4100])
def test_dimension_classes_l2(ctx, dim):
    rng = np.random.default_rng(dim)
    n, k = 1500, 20
    x = rng.normal(size=(n, dim)).astype(np.float32)
    q = rng.normal(size=(3, dim)).astype(np.float32)
    corpus = ctx.load_corpus(x)
    res = corpus.search(q, k, "l2")
    for i in range(3):
        ref = np.sqrt(((x.astype(np.float64) - q[i].astype(np.float64)) ** 2).sum(1))
        assert_valid_topk(res.rows[i, :res.counts[i]], res.dist[i, :res.counts[i]], ref, k, TOL)
    corpus.free()


def _ref_all(metric, x, q):
    x64, q64 = x.astype(np.float64), q.astype(np.float64)
    if metric == "l2":
        return np.sqrt(((x64 - q64) ** 2).sum(1))
    if metric == "ip":
        return -(x64 @ q64)
    if metric == "l1":
        return np.abs(x64 - q64).sum(1)
    with np.errstate(invalid="ignore", divide="ignore"):
        sim = (x64 @ q64) / np.sqrt((x64 ** 2).sum(1) * (q64 ** 2).sum())
    return 1.0 - np.clip(sim, -1, 1)


@pytest.mark.parametrize("metric", ["l2", "ip", "cosine", "l1"])
@pytest.mark.parametrize("dim", [128, 768])
def test_metrics_real_valued(ctx, oracle, metric, dim):
    rng = np.random.default_rng(11)
    n, k = 4000, 100
    x = rng.normal(size=(n, dim)).astype(np.float32)
    if metric == "cosine":
        x /= np.linalg.norm(x, axis=1, keepdims=True)             # config 3: L2-normalised rows
        x[5] = 0                                                  # one zero vector -> NaN distance, sorted last
    mask = (rng.random(n) < 0.5).astype(np.uint8)
    q = x[rng.integers(10, n, 6)] + 0.01 * rng.normal(size=(6, dim)).astype(np.float32)
    corpus = ctx.load_corpus(x)
    f = corpus.filter_from_bytemask(mask)
    res = corpus.search(q, k, metric, f)
    for i in range(6):
        ref = _ref_all(metric, x, q[i])
        m = res.counts[i]
        assert m == k
        assert_valid_topk(res.rows[i, :m], res.dist[i, :m], ref, k, TOL, candidates=np.flatnonzero(mask))
        # and the oracle's own values agree within the same tolerance
        oidx, odist = oracle.filtered_topk(metric, x, q[i], k, mask=mask)
        np.testing.assert_allclose(res.dist[i, :m], odist, rtol=TOL, atol=TOL)
    corpus.free()


def test_cosine_nan_rows_sort_last(ctx, oracle):
    x = np.asarray([[0, 0, 0, 0], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 0], [1, 1, 0, 0]], dtype=np.float32)
    corpus = ctx.load_corpus(x)
    res = corpus.search([[1, 0, 0, 0]], 5, "cosine")
    idx, dist = oracle.filtered_topk("cosine", x, [1, 0, 0, 0], 5)
    np.testing.assert_array_equal(res.rows[0], idx)
    assert np.isnan(res.dist[0, 3:]).all() and np.isnan(dist[3:]).all()
    np.testing.assert_allclose(res.dist[0, :3], dist[:3], atol=1e-6)
    corpus.free()


def test_nonfinite_and_overflowing_rows_under_shared_passes(ctx, oracle):
    """pgvector rejects NaN / Inf elements on input (vector.c:101-113) but accepts finite rows whose distance overflows
    (vector_type.out:387-391).  The loader accepts both; such a corpus must stay off the matrix-core screening (its
    error bound is not finite) and batched queries must still follow the oracle: Inf distances after the finite ones,
    NaN last."""
    rng = np.random.default_rng(21)
    n, dim = 2000, 100                                   # dim 100: ragged last stage of the K2 staging image
    x = sift_like(rng, n, dim)
    x[17, 3] = np.inf
    x[1200, :] = 3e38                                    # finite row, |row|^2 overflows
    x[1900, 99] = -np.inf
    blk, doc = _ids(n, 50)
    corpus = ctx.load_corpus(x, blk, doc)
    q = x[rng.integers(20, 1000, 5)].copy()
    before, _ = ctx.screening_check(0)
    for k in (10, n):                                    # k = n: the Inf / NaN tail is part of the answer
        res = corpus.search(q, k, "l2")
        for i in range(5):
            idx, dist = oracle.filtered_topk("l2", x, q[i], k, doc, blk)
            m = res.counts[i]
            assert m == idx.size
            np.testing.assert_array_equal(res.rows[i, :m], idx)
            np.testing.assert_array_equal(res.dist[i, :m], dist.astype(np.float32))
    assert ctx.screening_check(0)[0] == before           # never screened, so never flagged
    corpus.free()


# ---------------------------------------------------------------------------------------------
# k range, empty / ragged inputs
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k", [1, 2, 100, 128, 129, 500, 1024, 2048])
def test_k_range(ctx, oracle, k):
    rng = np.random.default_rng(5)
    n = 9000
    x = sift_like(rng, n)
    blk, doc = _ids(n, 100)
    corpus = ctx.load_corpus(x, blk, doc)
    q = x[rng.integers(0, n, 2)]
    res = corpus.search(q, k, "l2")
    for i in range(2):
        _expect_exact(oracle, res, i, "l2", x, q[i], k, doc, blk)
    corpus.free()


def test_many_partial_lists_take_two_select_levels(ctx, oracle):
    """48 unfiltered queries over 300k rows: every pass is cut into ~146 workgroups, so a query owns far more partial
    lists than one wave-per-query selection holds (fan-in 20 at k = 100) and K5 runs in two levels; the seeded main
    launch, the XCD-aware workgroup order and a reusable filter array (all None here) ride along."""
    rng = np.random.default_rng(41)
    n, nq, k = 300_000, 48, 100
    x = sift_like(rng, n)
    blk, doc = _ids(n, 100)
    corpus = ctx.load_corpus(x, blk, doc)
    qrows = rng.integers(0, n, nq)
    q = x[qrows] + rng.integers(-3, 4, (nq, x.shape[1])).astype(np.float32)      # still integer-valued: exact sums
    res = corpus.search(q, k, "l2")
    for i in range(0, nq, 5):
        _expect_exact(oracle, res, i, "l2", x, q[i], k, doc, blk)
    # the same batch again through one class-decomposed role filter per query (shared tile lists + bitmap-free ranges)
    perms = [(1, d) for d in range(1, n // 100 + 1, 3)] + [(2, d) for d in range(2, n // 100 + 1, 3)]
    corpus.load_rbac([(u, 1 + u % 2) for u in range(1, 9)] + [(8, 2)], perms)   # user 8 holds both roles
    users = rng.integers(1, 9, nq)
    farr = corpus.pack_filters([corpus.filter_for_user(int(u)) for u in users])
    res = corpus.search(q, k, "l2", farr)
    urole = {u: {1 + u % 2} for u in range(1, 9)}
    urole[8].add(2)
    docs_of = {1: set(d for r, d in perms if r == 1), 2: set(d for r, d in perms if r == 2)}
    for i in range(0, nq, 7):
        vis = set().union(*[docs_of[r] for r in urole[int(users[i])]])
        mask = np.isin(doc, np.fromiter(vis, dtype=np.int32)).astype(np.uint8)
        _expect_exact(oracle, res, i, "l2", x, q[i], k, doc, blk, mask)
    corpus.free()


def test_k_larger_than_rows_and_empty(ctx, oracle):
    import vsrbac
    rng = np.random.default_rng(6)
    x = sift_like(rng, 37)
    blk, doc = _ids(37, 5)
    corpus = ctx.load_corpus(x, blk, doc)
    res = corpus.search(x[:2], 100, "l2")
    for i in range(2):
        _expect_exact(oracle, res, i, "l2", x, x[i], 100, doc, blk)
    none = corpus.filter_from_bytemask(np.zeros(37, np.uint8), vsrbac.BITMAP)
    res = corpus.search(x[:2], 10, "l2", none)
    assert (res.counts == 0).all() and (res.block_ids == -1).all() and np.isinf(res.dist).all()
    with pytest.raises(vsrbac.VsrError):
        corpus.search(x[:1], 0, "l2")
    with pytest.raises(vsrbac.VsrError) as e:
        corpus.search(x[:1], 5000, "l2")
    assert e.value.status == 6
    corpus.free()
    empty = ctx.load_corpus(np.zeros((0, 8), dtype=np.float32))
    res = empty.search(np.zeros((1, 8), np.float32), 5, "l2")
    assert res.counts[0] == 0
    empty.free()


def test_unsorted_identities_are_reordered(ctx, oracle):
    """Rows arrive in arbitrary (document, block) order; results report the caller's row index."""
    rng = np.random.default_rng(8)
    n = 3000
    x = sift_like(rng, n, 32)
    doc = rng.integers(1, 40, n).astype(np.int32)
    blk = rng.permutation(n).astype(np.int64)
    corpus = ctx.load_corpus(x, blk, doc)
    res = corpus.search(x[:4], 64, "l2")
    for i in range(4):
        _expect_exact(oracle, res, i, "l2", x, x[i], 64, doc, blk)
    corpus.free()


# ---------------------------------------------------------------------------------------------
# BASELINE config 3 shape: 768-d cosine, L2-normalised rows, tree RBAC (class passes), k = 100
# (scaled to 200k rows so that the oracle finishes in seconds; the kernels are the full-size ones)
# ---------------------------------------------------------------------------------------------
def test_config3_768d_cosine_tree_rbac(ctx, oracle):
    from vsrbac.datasets import gaussian_corpus, tree_rbac
    from vsrbac.harness import Deployment
    n, dim, k = 200_000, 768, 100
    x, blk, doc = gaussian_corpus(n, dim, seed=3, normalize=True, blocks_per_doc=10)
    ndocs = int(doc.max())
    rbac = tree_rbac(num_users=200, num_roles=40, num_docs=ndocs, seed=3)
    corpus = ctx.load_corpus(x, blk, doc)
    corpus.load_rbac(rbac.user_roles, rbac.permissions)
    rng = np.random.default_rng(31)
    users = rng.integers(1, 201, 48)
    q = x[rng.integers(0, n, 48)] + 0.05 * rng.normal(size=(48, dim)).astype(np.float32)
    filters = [corpus.filter_for_user(int(u)) for u in users]
    res = corpus.search(q, k, "cosine", filters)                    # 48 queries: class passes on the matrix cores
    for i in range(0, 48, 6):
        mask = oracle.user_row_mask(int(users[i]), rbac.user_roles, rbac.permissions, doc)
        ref = _ref_all("cosine", x, q[i])
        m = res.counts[i]
        assert m == k
        assert_valid_topk(res.rows[i, :m], res.dist[i, :m], ref, k, TOL, candidates=np.flatnonzero(mask))
        oidx, odist = oracle.filtered_topk("cosine", x, q[i], k, doc, blk, mask)
        np.testing.assert_allclose(res.dist[i, :m], odist, rtol=TOL, atol=TOL)
        assert len(set(res.rows[i, :m].tolist()) & set(oidx.tolist())) >= k - 1     # ids equal up to a boundary near-tie
    for i in range(48):                                             # properties on every query
        assert res.counts[i] == k and (np.diff(res.dist[i]) >= -1e-6).all()
        assert np.isin(res.doc_ids[i], rbac.visible_docs(int(users[i]))).all()
    corpus.free()


def test_config5_batched_queries_arbitrary_predicate(ctx, oracle):
    """1000 batched queries, 768-d inner product, an arbitrary per-query-group byte mask (ACORN-style predicate:
    acorn_benchmark/src/benchmark_utils.cpp:342-396) AND-ed with nothing else: exact filtered top-k per query."""
    n, dim, k, nq = 60_000, 768, 100, 1000
    rng = np.random.default_rng(32)
    x = rng.normal(size=(n, dim)).astype(np.float32)
    q = rng.normal(size=(nq, dim)).astype(np.float32)
    corpus = ctx.load_corpus(x)
    masks = [(rng.random(n) < p).astype(np.uint8) for p in (0.5, 0.1, 0.02, 0.9)]
    fs = [corpus.filter_from_bytemask(m) for m in masks]
    filters = [fs[i % 4] for i in range(nq)]
    res = corpus.search(q, k, "ip", filters)
    for i in range(0, nq, 97):
        ref = _ref_all("ip", x, q[i])
        cand = np.flatnonzero(masks[i % 4])
        m = res.counts[i]
        assert m == min(k, cand.size)
        assert_valid_topk(res.rows[i, :m], res.dist[i, :m], ref, k, TOL, candidates=cand)
    assert (res.counts == k).all()
    corpus.free()


# ---------------------------------------------------------------------------------------------
# BASELINE config 2 at full size: SIFT1M-like, k = 100, role-partition prefilter (one GPU)
# ---------------------------------------------------------------------------------------------
def test_config2_sift1m_role_prefilter(ctx, oracle):
    import vsrbac
    from vsrbac.datasets import sift_like_corpus, tree_rbac
    n = 1_000_000
    x, blk, doc = sift_like_corpus(n, 128, seed=20251121)
    rbac = tree_rbac(num_users=1000, num_roles=100, num_docs=n // 100, seed=20251121)
    corpus = ctx.load_corpus(x, blk, doc)
    corpus.load_rbac(rbac.user_roles, rbac.permissions)
    rng = np.random.default_rng(9)
    users = rng.integers(1, 1001, 64)
    qrows = rng.integers(0, n, 64)
    for mode in (vsrbac.RANGES, vsrbac.BITMAP):
        filters = [corpus.filter_for_user(int(u), mode) for u in users]
        res = corpus.search(x[qrows], 100, "l2", filters)
        for i in range(0, 64, 8):                                    # oracle on a sample: ~0.1 s per query
            mask = oracle.user_row_mask(int(users[i]), rbac.user_roles, rbac.permissions, doc)
            _expect_exact(oracle, res, i, "l2", x, x[qrows[i]], 100, doc, blk, mask)
        # size-independent properties on every query: sorted, permitted, self-match first, full count
        for i in range(64):
            assert res.counts[i] == 100
            d = res.dist[i]
            assert (np.diff(d) >= 0).all()
            vis = rbac.visible_docs(int(users[i]))
            assert np.isin(res.doc_ids[i], vis).all()
            recomputed = np.sqrt(((x[res.rows[i]].astype(np.float64) - x[qrows[i]].astype(np.float64)) ** 2).sum(1))
            np.testing.assert_array_equal(d, recomputed.astype(np.float32))
    corpus.free()


# ---------------------------------------------------------------------------------------------
# randomized differential test: many small shapes through every planner branch, exact against the oracle
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(40))
def test_randomized_shapes_exact(ctx, oracle, seed):
    """Integer-valued rows small enough that fp32 sums are exact for every metric and dimension drawn (31^2 * 768 < 2^24),
    so ids and values must equal the oracle bit for bit whatever kernels the planner picks: single-query K1, shared
    passes on K2 (one or two query groups, registers or streamed B fragments) or K1m (L1), bitmaps or ranges or class
    decomposition, one or two selection levels, seeded or not."""
    rng = np.random.default_rng(1000 + seed)
    for _ in range(4):
        n = int(rng.choice([60, 700, 5000, 30000]))
        dim = int(rng.choice([4, 16, 64, 100, 128, 200, 320, 768]))
        k = int(rng.choice([1, 5, 10, 64, 100, 200]))
        nq = int(rng.choice([1, 2, 7, 16, 33, 100]))
        metric = str(rng.choice(["l2", "ip", "l1"]))
        x = np.clip(np.rint(np.abs(rng.normal(0, 6, (n, dim)))), 0, 31).astype(np.float32)
        rows_per_doc = int(rng.choice([1, 7, 50]))
        blk, doc = _ids(n, rows_per_doc)
        corpus = ctx.load_corpus(x, blk, doc)
        q = np.clip(np.rint(np.abs(rng.normal(0, 6, (nq, dim)))), 0, 31).astype(np.float32)
        kind = str(rng.choice(["none", "mask", "rbac"]))
        masks = [None] * nq
        filters = None
        if kind == "mask":
            base = [(rng.random(n) < p).astype(np.uint8) for p in (0.03, 0.5)]
            fs = [corpus.filter_from_bytemask(m) for m in base]
            pick = rng.integers(0, 2, nq)
            filters = [fs[j] for j in pick]
            masks = [base[j] for j in pick]
        elif kind == "rbac":
            ndocs = int(doc.max())
            nroles, nusers = 5, 9
            perms = sorted({(int(r), int(d)) for r in range(1, nroles + 1)
                            for d in rng.choice(np.arange(1, ndocs + 1), size=max(1, ndocs // 3), replace=False)})
            ur = sorted({(u, int(r)) for u in range(1, nusers + 1) for r in rng.choice(np.arange(1, nroles + 1), size=int(rng.integers(1, 3)), replace=False)})
            corpus.load_rbac(ur, perms)
            users = rng.integers(1, nusers + 1, nq)
            mode = vsrbac_mode(rng)
            filters = [corpus.filter_for_user(int(u), mode) for u in users]
            masks = [oracle.user_row_mask(int(u), ur, perms, doc) for u in users]
        res = corpus.search(q, k, metric, filters)
        for i in range(0, nq, max(1, nq // 6)):
            _expect_exact(oracle, res, i, metric, x, q[i], k, doc, blk, masks[i])
        corpus.free()


@pytest.mark.parametrize("dim,nq", [(64, 40), (100, 300), (128, 16), (128, 32), (128, 48), (128, 130), (128, 330),
                                     (160, 70), (192, 128), (200, 40), (256, 50)])
def test_wide_shared_passes_exact(ctx, oracle, dim, nq):
    """K2w (vsr_mfmaw.h): one 64-row tile per workgroup against up to 128 queries.  Shapes that hit every wave role
    (1 / 2 / >= 3 query groups per pass, one or two groups per wave), several balanced passes per filter part (330
    queries -> 3 x 110), 1..3 stages per tile (d = 64..192; d >= 193 stays on K2), ranges and bitmaps, a class-decomposed
    role filter and the threshold-seeded main launch (n large enough).  Integer-valued rows: bit-exact against the oracle."""
    import vsrbac
    rng = np.random.default_rng(7000 + dim * 1000 + nq)
    n, k = 70_000, 100
    x = np.clip(np.rint(np.abs(rng.normal(0, 6, (n, dim)))), 0, 31).astype(np.float32)
    blk, doc = _ids(n, 50)
    corpus = ctx.load_corpus(x, blk, doc)
    ndocs = int(doc.max())
    # three nested roles (a chain like the tree generator's ancestors) + one disjoint role
    own = np.array_split(rng.permutation(np.arange(1, ndocs + 1)), 4)
    perms = [(1, int(d)) for d in own[0]] + [(2, int(d)) for d in np.concatenate(own[:2])] + \
            [(3, int(d)) for d in np.concatenate(own[:3])] + [(4, int(d)) for d in own[3]]
    ur = [(u, 1 + (u % 4)) for u in range(1, 41)]
    corpus.load_rbac(ur, perms)
    users = rng.integers(1, 41, nq)
    q = np.clip(np.rint(np.abs(rng.normal(0, 6, (nq, dim)))), 0, 31).astype(np.float32)
    before, _ = ctx.screening_check(0)
    for mode in (vsrbac.RANGES, vsrbac.BITMAP):
        filters = [corpus.filter_for_user(int(u), mode) for u in users]
        res = corpus.search(q, k, "l2", filters)
        for i in range(0, nq, max(1, nq // 12)):
            mask = oracle.user_row_mask(int(users[i]), ur, perms, doc)
            _expect_exact(oracle, res, i, "l2", x, q[i], k, doc, blk, mask)
    res = corpus.search(q, k, "ip")                                   # unfiltered: every query shares one part
    for i in range(0, nq, max(1, nq // 8)):
        _expect_exact(oracle, res, i, "ip", x, q[i], k, doc, blk)
    corpus.free()


def test_one_query_per_call_is_one_launch_and_exact(ctx, oracle):
    """nq == 1 (the reference harness's call shape): K1 + in-kernel merge tree + outputs in ONE launch.  Two merge levels
    (600k rows -> hundreds of workgroup lists), ranges and bitmaps, k from 1 to 512 (8192 keys in the last merge at k = 256), a tie-saturated corpus (the radix
    select has to break ties by row id), inner product, a filter with fewer than k rows, an empty filter, device and host
    queries; k > 512 and cosine take the general path and still answer exactly."""
    import torch
    import vsrbac
    rng = np.random.default_rng(77)
    n, dim = 600_000, 128
    base = sift_like(rng, 3000)
    x = base[rng.integers(0, len(base), n)]              # every vector ~200 times: ties everywhere
    blk, doc = _ids(n, 100)
    corpus = ctx.load_corpus(x, blk, doc)
    ndocs = int(doc.max())
    perms = [(1, int(d)) for d in range(1, ndocs + 1, 3)] + [(2, 7)] + [(3, int(d)) for d in range(1, ndocs + 1)]
    ur = [(1, 1), (2, 2), (3, 3), (4, 4)]                 # user 2 sees one document (100 rows), user 4 nothing
    corpus.load_rbac(ur, perms)
    q = base[5] + rng.integers(0, 3, dim).astype(np.float32)
    for user, mode in ((1, vsrbac.RANGES), (1, vsrbac.BITMAP), (3, vsrbac.RANGES), (2, vsrbac.BITMAP), (4, vsrbac.RANGES)):
        f = corpus.filter_for_user(user, mode)
        mask = oracle.user_row_mask(user, ur, perms, doc)
        for k in (1, 100, 256, 512):
            res = corpus.search(q[None, :], k, "l2", [f])
            if k <= 256:      # (k = 512 fuses only when the pass has few workgroups: two merge levels hold 8192 keys each)
                assert "in-kernel merge" in ctx.last_scan_kernel(), ctx.last_scan_kernel()
            _expect_exact(oracle, res, 0, "l2", x, q, k, doc, blk, mask)
    res = corpus.search(q[None, :], 100, "ip")
    assert "in-kernel merge" in ctx.last_scan_kernel()
    _expect_exact(oracle, res, 0, "ip", x, q, 100, doc, blk)
    res = corpus.search(q[None, :], 600, "l2")            # k > 512: staging + K1 + K5
    assert "in-kernel merge" not in ctx.last_scan_kernel()
    _expect_exact(oracle, res, 0, "l2", x, q, 600, doc, blk)
    res = corpus.search(q[None, :], 50, "cosine")
    assert "in-kernel merge" not in ctx.last_scan_kernel()
    oidx, odist = oracle.filtered_topk("cosine", x, q, 50, doc, blk)
    np.testing.assert_allclose(res.dist[0, :50], odist, rtol=TOL, atol=TOL)
    # device-resident query, asynchronous call, twice in a row (the arrival counters must be back at zero)
    dev = torch.device("cuda", 0)
    k = 100
    outs = (torch.empty((1, k), dtype=torch.int64, device=dev), torch.empty((1, k), dtype=torch.int32, device=dev),
            torch.empty((1, k), dtype=torch.int64, device=dev), torch.empty((1, k), dtype=torch.float32, device=dev),
            torch.empty((1,), dtype=torch.int32, device=dev))
    d_q = torch.from_numpy(q[None, :].copy()).to(dev)
    f = corpus.filter_for_user(1, vsrbac.RANGES)
    mask = oracle.user_row_mask(1, ur, perms, doc)
    for _ in range(2):
        corpus.search_device(d_q.data_ptr(), 1, k, "l2", [f], *(t.data_ptr() for t in outs))
        ctx.synchronize()
        got = SimpleNamespace(block_ids=outs[0].cpu().numpy(), doc_ids=outs[1].cpu().numpy(), rows=outs[2].cpu().numpy(),
                              dist=outs[3].cpu().numpy(), counts=outs[4].cpu().numpy())
        _expect_exact(oracle, got, 0, "l2", x, q, k, doc, blk, mask)
    corpus.free()


def test_one_query_per_call_on_int8_planes(ctx, oracle):
    """nq == 1 over a SIFT-like corpus: the fused launch reads the int8 planes (scan8_fused_kernel: a quarter of the bytes
    per row) and returns the bytes the fp32 kernel returns.  Host queries are checked by the library; device queries
    only under the u8 hint, verified in the kernel: a query that is not integer-valued in 0..255 is FLAGGED (negative
    count) and the exact variant re-runs it on the fp32 rows."""
    import torch
    import vsrbac
    rng = np.random.default_rng(909)
    for n, dim in ((200_000, 128), (30_000, 96)):
        x = sift_like(rng, n, dim)
        blk, doc = _ids(n, 37)                                          # ragged tiles: 16 + 16 + 5
        corpus = ctx.load_corpus(x, blk, doc)
        mask = (rng.random(int(doc.max()) + 1) < 0.4)[doc].astype(np.uint8)
        for mode in (vsrbac.RANGES, vsrbac.BITMAP, None):
            f = None if mode is None else corpus.filter_from_bytemask(mask, mode)
            for k in (1, 10, 100):
                q = x[rng.integers(0, n)].copy()
                q[:4] = rng.integers(0, 256, 4)
                res = corpus.search(q[None, :], k, "l2", None if f is None else [f])
                assert "scan8" in ctx.last_scan_kernel(), ctx.last_scan_kernel()
                _expect_exact(oracle, res, 0, "l2", x, q, k, doc, blk, None if f is None else mask)
        q = x[7].copy()
        res = corpus.search(q[None, :], 10, "ip")
        assert "scan8" not in ctx.last_scan_kernel()                   # an L2 path only
        q[3] = 0.5
        res = corpus.search(q[None, :], 10, "l2")                       # not u8-exact: the fp32 rows
        assert "scan8" not in ctx.last_scan_kernel()
        _expect_exact(oracle, res, 0, "l2", x, q, 10, doc, blk)
        # device-resident queries
        dev = torch.device("cuda", 0)
        k = 20
        outs = (torch.empty((1, k), dtype=torch.int64, device=dev), torch.empty((1, k), dtype=torch.int32, device=dev),
                torch.empty((1, k), dtype=torch.int64, device=dev), torch.empty((1, k), dtype=torch.float32, device=dev),
                torch.empty((1,), dtype=torch.int32, device=dev))
        ptrs = tuple(t.data_ptr() for t in outs)
        good = x[11].copy()
        d_good, d_bad = torch.from_numpy(good[None, :].copy()).to(dev), torch.from_numpy(q[None, :].copy()).to(dev)
        corpus.search_device(d_good.data_ptr(), 1, k, "l2", None, *ptrs)
        ctx.synchronize()
        assert "scan8" not in ctx.last_scan_kernel()                   # nothing is assumed without the hint
        ctx.set_query_hint(True)
        corpus.search_device(d_good.data_ptr(), 1, k, "l2", None, *ptrs)
        ctx.synchronize()
        assert "scan8" in ctx.last_scan_kernel() and int(outs[4].cpu()[0]) == k
        got = SimpleNamespace(block_ids=outs[0].cpu().numpy(), doc_ids=outs[1].cpu().numpy(), rows=outs[2].cpu().numpy(),
                              dist=outs[3].cpu().numpy(), counts=outs[4].cpu().numpy())
        _expect_exact(oracle, got, 0, "l2", x, good, k, doc, blk)
        corpus.search_device(d_bad.data_ptr(), 1, k, "l2", None, *ptrs)  # the promise is broken
        _, flags = ctx.screening_check(1)
        assert flags[0] and int(outs[4].cpu()[0]) < 0
        ctx.set_query_hint(True)
        assert corpus.search_device_exact(d_bad.data_ptr(), 1, k, "l2", None, *ptrs) == 1
        got = SimpleNamespace(block_ids=outs[0].cpu().numpy(), doc_ids=outs[1].cpu().numpy(), rows=outs[2].cpu().numpy(),
                              dist=outs[3].cpu().numpy(), counts=outs[4].cpu().numpy())
        _expect_exact(oracle, got, 0, "l2", x, q, k, doc, blk)
        ctx.set_query_hint(False)
        corpus.free()


@pytest.mark.parametrize("seed", range(16))
def test_one_query_per_call_randomized(ctx, oracle, seed):
    """nq == 1, random shapes: the in-kernel merge's radix selects (per-workgroup top-k, two merge levels, rank placement of
    the final order) on corpora from a few dozen rows to 150 k, tie-saturated or plain, u8-valued (scan8 on the int8 planes)
    or not (K1 on the fp32 rows), ranges / bitmaps / no filter, k from 1 to 400, ragged documents."""
    import vsrbac
    rng = np.random.default_rng(4242 + seed)
    n = int(rng.choice([40, 700, 5_000, 40_000, 150_000]))
    dim = int(rng.choice([8, 32, 96, 128]))
    if rng.random() < 0.5:
        base = sift_like(rng, max(8, n // int(rng.choice([1, 50, 400]))), dim)     # every vector up to ~400 times: ties
        x = base[rng.integers(0, len(base), n)]
    else:
        x = sift_like(rng, n, dim)
    if seed % 4 == 3:
        x = x + np.float32(0.5)                                                     # not u8-valued: the fp32 rows
    blk, doc = _ids(n, int(rng.choice([1, 7, 37, 100])))
    corpus = ctx.load_corpus(x, blk, doc)
    mask = (rng.random(int(doc.max()) + 1) < rng.choice([0.05, 0.4, 0.95]))[doc].astype(np.uint8)
    for mode in (vsrbac.RANGES, vsrbac.BITMAP, None):
        f = None if mode is None else corpus.filter_from_bytemask(mask, mode)
        for k in sorted({1, int(rng.integers(2, 40)), int(rng.integers(40, 400))}):
            q = x[rng.integers(0, n)].copy()
            for metric in (("l2", "ip") if k < 40 and seed % 4 != 3 else ("l2",)):       # (x + 0.5: only the L2 sums stay exact)
                res = corpus.search(q[None, :], k, metric, None if f is None else [f])
                _expect_exact(oracle, res, 0, metric, x, q, k, doc, blk, None if f is None else mask)
    corpus.free()


def test_int8_planes_for_sift_like_queries(ctx, oracle):
    """A corpus of integers 0..255 (d <= 128) keeps int8 planes; L2 searches whose queries are such integers screen on
    them (v_mfma_i32_16x16x64_i8, exact).  Host queries are checked by the library; device-resident queries only under
    vsr_set_query_hint, verified per query on the device: a query that breaks the promise is flagged (negative count),
    the exact variant re-runs it, and the hint is dropped."""
    import torch
    import vsrbac
    rng = np.random.default_rng(4242)
    n, dim, k, nq = 80_000, 128, 100, 96
    x = sift_like(rng, n)
    blk, doc = _ids(n, 40)
    corpus = ctx.load_corpus(x, blk, doc)
    q = x[rng.integers(0, n, nq)].copy()
    q[:, :7] = rng.integers(0, 256, (nq, 7)).astype(np.float32)        # still integers 0..255, no longer corpus rows
    res = corpus.search(q, k, "l2")
    assert "int8" in ctx.last_scan_kernel(), ctx.last_scan_kernel()
    for i in range(0, nq, 7):
        _expect_exact(oracle, res, i, "l2", x, q[i], k, doc, blk)
    q2 = q.copy()
    q2[5, 3] = 17.5                                                    # one non-integer element: bf16 planes for this call
    res = corpus.search(q2, k, "l2")
    assert "int8" not in ctx.last_scan_kernel()
    _expect_exact(oracle, res, 5, "l2", x, q2[5], k, doc, blk)
    res = corpus.search(q, k, "ip")                                    # int8 planes are an L2 path only
    assert "int8" not in ctx.last_scan_kernel()
    # device-resident queries: nothing is assumed without the hint
    dev = torch.device("cuda", 0)
    outs = (torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), dtype=torch.int32, device=dev),
            torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev),
            torch.empty((nq,), dtype=torch.int32, device=dev))
    ptrs = tuple(t.data_ptr() for t in outs)
    d_q = torch.from_numpy(q).to(dev)
    corpus.search_device(d_q.data_ptr(), nq, k, "l2", None, *ptrs)
    ctx.synchronize()
    assert "int8" not in ctx.last_scan_kernel()
    ctx.set_query_hint(True)
    corpus.search_device(d_q.data_ptr(), nq, k, "l2", None, *ptrs)
    _, flags = ctx.screening_check(nq)
    assert "int8" in ctx.last_scan_kernel() and not flags.any()
    got = SimpleNamespace(block_ids=outs[0].cpu().numpy(), doc_ids=outs[1].cpu().numpy(), rows=outs[2].cpu().numpy(),
                          dist=outs[3].cpu().numpy(), counts=outs[4].cpu().numpy())
    for i in range(0, nq, 9):
        _expect_exact(oracle, got, i, "l2", x, q[i], k, doc, blk)
    d_q2 = torch.from_numpy(q2).to(dev)                                # the promise is broken by query 5
    corpus.search_device(d_q2.data_ptr(), nq, k, "l2", None, *ptrs)
    _, flags = ctx.screening_check(nq)
    cnt = outs[4].cpu().numpy()
    assert flags[5] and cnt[5] < 0 and int(flags.sum()) == 1
    ctx.set_query_hint(True)
    n_rerun = corpus.search_device_exact(d_q2.data_ptr(), nq, k, "l2", None, *ptrs)
    assert n_rerun == 1
    got = SimpleNamespace(block_ids=outs[0].cpu().numpy(), doc_ids=outs[1].cpu().numpy(), rows=outs[2].cpu().numpy(),
                          dist=outs[3].cpu().numpy(), counts=outs[4].cpu().numpy())
    for i in (4, 5, 6):
        _expect_exact(oracle, got, i, "l2", x, q2[i], k, doc, blk)
    corpus.search_device(d_q.data_ptr(), nq, k, "l2", None, *ptrs)     # the violation dropped the hint
    ctx.synchronize()
    assert "int8" not in ctx.last_scan_kernel()
    ctx.set_query_hint(False)
    corpus.free()


def vsrbac_mode(rng):
    import vsrbac
    return vsrbac.RANGES if rng.random() < 0.5 else vsrbac.BITMAP
