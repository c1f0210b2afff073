"""K2g (vsr_gemm.h): wide passes over LONG rows -- 256 x 256 tiles, both operands through LDS, one bf16 product per
element on the coarse planes, exact re-rank of the 4k survivors.

Parity bar as everywhere: integer-valued rows whose elements are exact in bf16 make the coarse product exact, so ids and
fp32 distances must equal the oracle's bit for bit; real-valued rows must give a valid top-k within 1e-4 of the float64
reference; a query whose coarse screening cannot be proven must come back exact anyway (tiered re-run) when the caller
uses vsr_search / vsr_search_device_exact, and must be FLAGGED (negative count) on the asynchronous API."""
import ctypes

import numpy as np
import pytest

from helpers import assert_valid_topk

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def ctx():
    import vsrbac
    c = vsrbac.Context(0)
    yield c
    c.close()


def _ids(n, rows_per_doc):
    return (np.arange(n) + 1).astype(np.int64), (np.arange(n) // rows_per_doc + 1).astype(np.int32)


def _expect_exact(oracle, res, qi, metric, x, q, k, doc, blk, mask=None):
    idx, dist = oracle.filtered_topk(metric, x, q, k, doc, blk, mask)
    m = res.counts[qi]
    assert m == idx.size, (qi, m, idx.size)
    np.testing.assert_array_equal(res.rows[qi, :m], idx)
    np.testing.assert_array_equal(res.dist[qi, :m], dist.astype(np.float32))
    assert (res.block_ids[qi, m:] == -1).all() and np.isinf(res.dist[qi, m:]).all()


def _search_async(ctx, corpus, q, k, metric, filters=None):
    """vsr_search_device: no re-run tier behind it -- what comes back is what the launched kernels produced; flagged
    queries carry negative counts."""
    import torch
    from vsrbac.engine import SearchResult
    dev = torch.device("cuda", 0)
    nq = len(q)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    d_q = torch.from_numpy(np.ascontiguousarray(q)).to(dev)
    o_blk = torch.empty((nq, k), dtype=torch.int64, device=dev)
    o_doc = torch.empty((nq, k), dtype=torch.int32, device=dev)
    o_row = torch.empty((nq, k), dtype=torch.int64, device=dev)
    o_dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    o_cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    keep = corpus.search_device(p(d_q), nq, k, metric, filters, p(o_blk), p(o_doc), p(o_row), p(o_dist), p(o_cnt))
    ctx.synchronize()
    del keep
    return SearchResult(o_blk.cpu().numpy(), o_doc.cpu().numpy(), o_row.cpu().numpy(), o_dist.cpu().numpy(), o_cnt.cpu().numpy())


@pytest.mark.parametrize("dim,nq,n", [(320, 129, 50_001), (768, 300, 60_000), (512, 700, 50_123), (1000, 257, 50_000)])
def test_gemm_path_is_exact_on_integer_rows(ctx, oracle, dim, nq, n):
    """Unfiltered batches and big shared filters (one part seen by every query) of more than 128 queries take K2g: ragged
    row counts, 1..3 passes per part (129 -> 1 pass of 144 slots, 300 -> 2 x 160, 700 -> 3 x 240), 5..16 K-steps, ranges
    (tiles of 2..4 rows gathered into 256-row tiles), bitmaps (masked rows inside the tiles).  Rows and queries are small
    integers (exact in bf16: the coarse product is exact) laid out as 40 tight clusters in a sea of far rows, so that the
    gap behind the 4k survivors is far outside the coarse error bound: the asynchronous API must answer WITHOUT flagging,
    i.e. what is compared bit for bit with the oracle is K2g's own result, not a re-run's."""
    import vsrbac
    rng = np.random.default_rng(dim * 7 + nq)
    k, n_cl, per_cl = 100, 40, 330                    # per_cl < 4k: the survivor list reaches into the sea
    x = np.clip(np.rint(np.abs(rng.normal(0, 3, (n, dim)))), 0, 15).astype(np.float32)
    centres = np.clip(np.rint(np.abs(rng.normal(0, 3, (n_cl, dim)))), 0, 15).astype(np.float32)

    def jitter(v, m):
        out = np.repeat(v[None, :], m, axis=0)
        for r in range(m):
            at = rng.choice(dim, 24, replace=False)
            out[r, at] = np.clip(out[r, at] + rng.choice([-1.0, 1.0], 24), 0, 15)
        return out

    planted = rng.permutation(n)[: n_cl * per_cl].reshape(n_cl, per_cl)
    for c in range(n_cl):
        x[planted[c]] = jitter(centres[c], per_cl)
    blk, doc = _ids(n, 7)
    corpus = ctx.load_corpus(x, blk, doc)
    qc = rng.integers(0, n_cl, nq)
    q = np.concatenate([jitter(centres[c], 1) for c in qc])
    before, _ = ctx.screening_check(0)
    res = _search_async(ctx, corpus, q, k, "l2")
    assert "K2g" in ctx.last_scan_kernel(), ctx.last_scan_kernel()
    assert (res.counts == k).all(), res.counts[res.counts != k][:8]
    for i in range(0, nq, max(1, nq // 10)):
        _expect_exact(oracle, res, i, "l2", x, q[i], k, doc, blk)
    mask = (rng.random(n) < 0.6).astype(np.uint8)
    mask[: n // 5] = 0                                                # a long masked stretch: whole tiles without a row
    for mode in (vsrbac.RANGES, vsrbac.BITMAP):
        f = corpus.filter_from_bytemask(mask, mode)
        res = _search_async(ctx, corpus, q, k, "l2", [f] * nq)
        if mode == vsrbac.RANGES or dim <= 512:                       # (bitmap windows of a small corpus can leave the 256-row
            assert "K2g" in ctx.last_scan_kernel(), ctx.last_scan_kernel()    # sample too thin to seed: the planner then takes K2w)
        ok = res.counts >= 0                                          # (a cluster with < k permitted rows reaches into the sea: may flag)
        assert ok.mean() > 0.5
        for i in np.flatnonzero(ok)[:: max(1, int(ok.sum()) // 10)]:
            _expect_exact(oracle, res, int(i), "l2", x, q[i], k, doc, blk, mask)
        f.free()
    after, _ = ctx.screening_check(0)
    assert after >= before
    # through the host API every query is answered exactly, whatever tier proved it: inner product (no cluster structure
    # to rely on), a filter smaller than k and an empty one
    res = corpus.search(q, k, "ip")
    for i in range(0, nq, max(1, nq // 8)):
        _expect_exact(oracle, res, i, "ip", x, q[i], k, doc, blk)
    tiny = np.zeros(n, dtype=np.uint8)
    tiny[rng.choice(n, 37, replace=False)] = 1
    for m_ in (tiny, np.zeros(n, dtype=np.uint8)):
        f = corpus.filter_from_bytemask(m_, vsrbac.BITMAP)
        res = corpus.search(q, k, "l2", [f] * nq)
        for i in range(0, nq, max(1, nq // 5)):
            _expect_exact(oracle, res, i, "l2", x, q[i], k, doc, blk, m_)
        f.free()
    corpus.free()


@pytest.mark.parametrize("metric", ["cosine", "l2", "ip"])
def test_gemm_path_real_valued_rows(ctx, oracle, metric):
    """768-d Gaussian rows (configs 3 / 5's shape), 300 queries = corpus rows + noise: valid top-k within 1e-4 of the
    float64 reference for every sampled query, whatever tier finally answered it (the host API re-runs flagged ones)."""
    rng = np.random.default_rng(31)
    n, dim, nq, k = 120_000, 768, 300, 100
    x = rng.standard_normal((n, dim), dtype=np.float32)
    if metric == "cosine":
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    blk, doc = _ids(n, 10)
    corpus = ctx.load_corpus(x, blk, doc)
    q = x[rng.integers(0, n, nq)] + (0.05 if metric == "cosine" else 0.3) * rng.standard_normal((nq, dim)).astype(np.float32)
    ares = _search_async(ctx, corpus, q, k, metric)
    assert "K2g" in ctx.last_scan_kernel(), ctx.last_scan_kernel()
    flagged = float((ares.counts < 0).mean())
    print(f"K2g {metric}: {flagged:.3f} of the queries flagged by the coarse tier")
    assert flagged <= 0.05                                             # the bench's 768-d legs rely on this being rare
    res = corpus.search(q, k, metric)
    x64 = x.astype(np.float64)
    for i in range(0, nq, 25):
        q64 = q[i].astype(np.float64)
        if metric == "l2":
            ref = np.sqrt(((x64 - q64) ** 2).sum(1))
        elif metric == "ip":
            ref = -(x64 @ q64)
        else:
            ref = 1.0 - np.clip((x64 @ q64) / np.sqrt((x64 ** 2).sum(1) * (q64 ** 2).sum()), -1, 1)
        assert res.counts[i] == k
        assert_valid_topk(res.rows[i], res.dist[i], ref, k, TOL)
        if ares.counts[i] == k:                                        # proven by the coarse tier itself
            assert_valid_topk(ares.rows[i], ares.dist[i], ref, k, TOL)
    corpus.free()


def test_coarse_screen_flags_what_it_cannot_prove(ctx, oracle):
    """A cluster of near-duplicates around every query: the gaps between the k-th and the 4k-th neighbour are far inside
    the coarse planes' error bound, so the coarse tier must FLAG (asynchronous API: negative counts) and the tiers below
    must still deliver the exact answer (vsr_search_device_exact / vsr_search)."""
    import torch
    rng = np.random.default_rng(41)
    n, dim, nq, k = 60_000, 384, 200, 50
    centre = rng.standard_normal((1, dim)).astype(np.float32)
    x = rng.standard_normal((n, dim), dtype=np.float32)
    x[:5000] = centre + 1e-3 * rng.standard_normal((5000, dim)).astype(np.float32)     # 5000 rows within 0.03 of each other
    blk, doc = _ids(n, 10)
    corpus = ctx.load_corpus(x, blk, doc)
    q = (centre + 1e-3 * rng.standard_normal((nq, dim))).astype(np.float32)
    dev = torch.device("cuda", 0)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    d_q = torch.from_numpy(q).to(dev)
    o_blk = torch.empty((nq, k), dtype=torch.int64, device=dev)
    o_doc = torch.empty((nq, k), dtype=torch.int32, device=dev)
    o_row = torch.empty((nq, k), dtype=torch.int64, device=dev)
    o_dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    o_cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    corpus.search_device(p(d_q), nq, k, "l2", None, p(o_blk), p(o_doc), p(o_row), p(o_dist), p(o_cnt))
    ctx.synchronize()
    assert "K2g" in ctx.last_scan_kernel(), ctx.last_scan_kernel()
    cnt = o_cnt.cpu().numpy()
    _, flags = ctx.screening_check(nq)
    assert (cnt < 0).sum() == np.count_nonzero(flags) > 0            # unproven queries cannot be mistaken for results
    n_rerun = corpus.search_device_exact(p(d_q), nq, k, "l2", None, p(o_blk), p(o_doc), p(o_row), p(o_dist), p(o_cnt))
    assert n_rerun == np.count_nonzero(flags)
    rows, dist, cnt = o_row.cpu().numpy(), o_dist.cpu().numpy(), o_cnt.cpu().numpy()
    res = corpus.search(q, k, "l2")
    x64 = x.astype(np.float64)
    for i in range(0, nq, 20):
        ref = np.sqrt(((x64 - q[i].astype(np.float64)) ** 2).sum(1))
        assert cnt[i] == k and res.counts[i] == k
        assert_valid_topk(rows[i], dist[i], ref, k, TOL)
        assert_valid_topk(res.rows[i], res.dist[i], ref, k, TOL)
    corpus.free()
