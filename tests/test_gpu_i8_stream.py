"""K2i (vsr_i8s.h): the int8 main launch as per-wave LDS-DMA streams, against the oracle and against K2w.

Integer-valued rows (0..255): every fp32 sum of vector.c is exact, so ids, order and distances are compared bit for bit.
The planner picks the mask epilogue (and with it K2i) only when it expects few survivors per tile, i.e. at bench sizes;
VSR_FORCE_EPI=1 selects it for the small corpora of this file.  Covered: ragged list tiles (documents of 37 rows: 16 + 16
+ 5), RANGES and BITMAP filters (the permission bit tested per row inside the two-word window), passes of 1..64 queries
and more than 64 (several passes), workgroups whose waves get 0, 1 or many stages, an unfiltered identity list, thresholds
that stay open (a filter that fits the candidate buffer), and the same searches on K2w (the default) returning the same
bytes.  K2i is opt-in (VSR_K2I=1) while it is the slower of the two on the headline step."""
import numpy as np
import pytest

from helpers import sift_like

pytestmark = pytest.mark.gpu


def _ctx(monkeypatch, **env):
    import vsrbac
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    c = vsrbac.Context(0)
    for k in env:
        monkeypatch.delenv(k)
    return c


def _check(oracle, res, qs, x, q, k, doc, blk, masks):
    for i in qs:
        idx, dist = oracle.filtered_topk("l2", x, q[i], k, doc, blk, masks[i])
        m = res.counts[i]
        assert m == idx.size, (i, m, idx.size)
        np.testing.assert_array_equal(res.rows[i, :m], idx)
        np.testing.assert_array_equal(res.dist[i, :m], dist.astype(np.float32))
        assert (res.block_ids[i, m:] == -1).all()


@pytest.mark.parametrize("mode", ["ranges", "bitmap"])
def test_k2i_matches_oracle_and_k2w(oracle, monkeypatch, mode):
    import vsrbac
    rng = np.random.default_rng(515)
    n, dim, k = 150_000, 128, 50
    x = sift_like(rng, n)
    blk = (np.arange(n) + 1).astype(np.int64)
    doc = (np.arange(n) // 37 + 1).astype(np.int32)
    ndocs = int(doc.max())
    # three "roles": half of the documents, a tenth, and a thin one whose rows fit the candidate buffer (thresholds stay open)
    doc_sets = [rng.random(ndocs + 1) < p for p in (0.5, 0.1, 0.003)]
    row_masks = [s[doc].astype(np.uint8) for s in doc_sets]
    per_role = (150, 22, 3)                                            # > 64 queries: several passes; 22: two groups; 3: one
    role_of = np.concatenate([np.full(c, r) for r, c in enumerate(per_role)] + [np.full(9, -1)])   # -1: no filter
    rng.shuffle(role_of)
    nq = len(role_of)
    q = x[rng.integers(0, n, nq)].copy()
    q[:, :5] = rng.integers(0, 256, (nq, 5)).astype(np.float32)
    masks = [None if r < 0 else row_masks[r] for r in role_of]
    results = {}
    for label, env in (("k2i", {"VSR_FORCE_EPI": "1", "VSR_K2I": "1"}), ("k2i_wide", {"VSR_FORCE_EPI": "1", "VSR_K2I": "1", "VSR_K2I_WIDE": "1"}),
                       ("k2w", {"VSR_FORCE_EPI": "1"})):
        ctx = _ctx(monkeypatch, **env)
        corpus = ctx.load_corpus(x, blk, doc)
        fmode = vsrbac.RANGES if mode == "ranges" else vsrbac.BITMAP
        role_filters = [corpus.filter_from_bytemask(m, fmode) for m in row_masks]
        filters = [None if r < 0 else role_filters[r] for r in role_of]
        res = corpus.search(q, k, "l2", filters)
        name = ctx.last_scan_kernel()
        assert ("K2i" in name) == label.startswith("k2i"), name
        assert "int8" in name, name
        results[label] = res
        if label == "k2i":
            _check(oracle, res, list(range(0, nq, 5)) + [int(i) for i in np.flatnonzero(role_of == 2)], x, q, k, doc, blk, masks)
        corpus.free()
        ctx.close()
    for other in ("k2i_wide", "k2w"):                                  # 128-column passes (8 groups per wave); K2w's tiles
        a, b = results["k2i"], results[other]
        np.testing.assert_array_equal(a.counts, b.counts)
        np.testing.assert_array_equal(a.rows, b.rows)
        np.testing.assert_array_equal(a.dist, b.dist)


def test_k2i_short_streams_and_device_api(oracle, monkeypatch):
    """Workgroups with fewer list tiles than waves, a corpus smaller than one mapping chunk, and the device API with the
    u8 query hint: counts are never negative (int8 screening is exact: nothing to flag)."""
    import torch
    rng = np.random.default_rng(616)
    ctx = _ctx(monkeypatch, VSR_FORCE_EPI="1", VSR_K2I="1", VSR_MIN_ROWS_PER_BLOCK="16")
    ctx.set_query_hint(True)
    for n in (700, 5_000, 40_000):
        dim, k, nq = 96, 10, 40                                        # d = 96: rows padded to 128 int8 elements
        x = sift_like(rng, n, dim)
        blk = (np.arange(n) + 1).astype(np.int64)
        doc = (np.arange(n) // 7 + 1).astype(np.int32)
        corpus = ctx.load_corpus(x, blk, doc)
        q = x[rng.integers(0, n, nq)].copy()
        q[:, 0] = rng.integers(0, 256, nq)
        dq = torch.from_numpy(q).cuda()
        outs = (torch.empty((nq, k), dtype=torch.int64, device="cuda"), torch.empty((nq, k), dtype=torch.int32, device="cuda"),
                torch.empty((nq, k), dtype=torch.int64, device="cuda"), torch.empty((nq, k), dtype=torch.float32, device="cuda"),
                torch.empty((nq,), dtype=torch.int32, device="cuda"))
        corpus.search_device(dq.data_ptr(), nq, k, "l2", None, *(t.data_ptr() for t in outs))
        ctx.synchronize()
        name = ctx.last_scan_kernel()
        cnt = outs[4].cpu().numpy()
        assert (cnt >= 0).all()
        rows, dist = outs[2].cpu().numpy(), outs[3].cpu().numpy()
        for i in range(0, nq, 3):
            idx, d = oracle.filtered_topk("l2", x, q[i], k, doc, blk, None)
            np.testing.assert_array_equal(rows[i, :cnt[i]], idx)
            np.testing.assert_array_equal(dist[i, :cnt[i]], d.astype(np.float32))
        if n >= 40_000:
            assert "K2i" in name, name                                 # (smaller corpora may be planned onto K1 / K1m)
        corpus.free()
    ctx.close()
