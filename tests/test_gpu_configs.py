"""BASELINE.json configurations 3, 4 and 5 at their real sizes on one MI355X, through the C ABI.

At these sizes the oracle checks a sample of the queries (seconds each) and every query is checked through
size-independent properties: sorted, permitted, full count, distances recomputed from the returned rows.  The multi-GPU
configurations (C4, C5) additionally run as 8 row-range shards on the one GPU -- per-shard search, records stacked like the
all-gather delivers them, vsr_merge_topk_packed_device -- and must equal the unsharded result bit for bit.
"""
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


def _ranges_of(rbac, user, rows_per_doc):
    return [((int(d) - 1) * rows_per_doc, rows_per_doc) for d in rbac.visible_docs(int(user)).astype(np.int64)]


# ---------------------------------------------------------------------------------------------
# C4: SIFT10M-like 10M x 128, k = 100, row-level-security bitmap (post-filter mode), 8 shards
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def sift10m(ctx):
    from vsrbac.datasets import sift_like_corpus, tree_rbac
    n = 10_000_000
    x, blk, doc = sift_like_corpus(n, 128, seed=20251121)
    rbac = tree_rbac(num_users=1000, num_roles=100, num_docs=n // 100, seed=20251121)
    return n, x, blk, doc, rbac


def test_config4_sift10m_rls_bitmap(ctx, oracle, sift10m):
    import torch
    import vsrbac
    from vsrbac.datasets import sample_queries
    from vsrbac.sharded import shard_bounds
    n, x, blk, doc, rbac = sift10m
    k, nq = 100, 256
    qrow, quser = sample_queries(nq, n, 1000, seed=4)
    q = x[qrow] + np.float32(1.0)                                     # still integer-valued, not an exact corpus row
    corpus = ctx.load_corpus(x, blk, doc)
    corpus.load_rbac(rbac.user_roles, rbac.permissions)
    before = ctx.screening_check(0)[0]
    res = {}
    for mode in (vsrbac.BITMAP, vsrbac.RANGES):
        filters = [corpus.filter_for_user(int(u), mode) for u in quser]
        res[mode] = corpus.search(q, k, "l2", filters)
    r = res[vsrbac.BITMAP]
    np.testing.assert_array_equal(r.rows, res[vsrbac.RANGES].rows)    # post-filter and pre-filter agree
    np.testing.assert_array_equal(r.dist, res[vsrbac.RANGES].dist)
    assert ctx.screening_check(0)[0] == before, "no query of this workload may need the exact re-run"
    # oracle on a sample
    m = 8
    rows_o, dist_o, cnt_o = oracle.search_ranges("l2", x, q[:m], k, [_ranges_of(rbac, u, 100) for u in quser[:m]], doc, blk)
    np.testing.assert_array_equal(r.rows[:m], rows_o)
    np.testing.assert_array_equal(r.dist[:m], dist_o.astype(np.float32))
    # properties on every query
    for i in range(nq):
        assert r.counts[i] == k
        assert (np.diff(r.dist[i]) >= 0).all()
        assert np.isin(r.doc_ids[i], rbac.visible_docs(int(quser[i]))).all()
        d2 = ((x[r.rows[i]].astype(np.float64) - q[i].astype(np.float64)) ** 2).sum(1)
        np.testing.assert_array_equal(r.dist[i], np.sqrt(d2).astype(np.float32))
    # 8 row-range shards + packed merge == unsharded
    dev = torch.device("cuda", 0)
    parts = 8
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    rec = ctx.packed_result_bytes(nq, k)
    nk = nq * k
    packed = torch.empty((parts * rec,), dtype=torch.uint8, device=dev)
    d_q = torch.from_numpy(q).to(dev)
    cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    shards = []
    for s in range(parts):
        lo, hi = shard_bounds(n, parts, s, align=100)
        c = ctx.load_corpus(x[lo:hi], blk[lo:hi], doc[lo:hi], row_offset=lo)
        c.load_rbac(rbac.user_roles, rbac.permissions)
        fl = [c.filter_for_user(int(u), vsrbac.BITMAP) for u in quser]
        pk = packed[s * rec:(s + 1) * rec]
        c.search_device(p(d_q), nq, k, "l2", fl, p(pk[nk * 8:nk * 16]), p(pk[nk * 16:nk * 20]), None, p(pk[nk * 20:nk * 24]),
                        p(cnt), p(pk[0:nk * 8]))
        shards.append(c)
    ctx.synchronize()
    assert ctx.screening_check(0)[0] == before
    o_blk = torch.empty((nq, k), dtype=torch.int64, device=dev)
    o_doc = torch.empty((nq, k), dtype=torch.int32, device=dev)
    o_dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    o_cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    ctx.merge_topk_packed_device(p(packed), parts, nq, k, p(o_blk), p(o_doc), p(o_dist), None, p(o_cnt))
    ctx.synchronize()
    np.testing.assert_array_equal(o_blk.cpu().numpy(), r.block_ids)
    np.testing.assert_array_equal(o_doc.cpu().numpy(), r.doc_ids)
    np.testing.assert_array_equal(o_dist.cpu().numpy(), r.dist)
    assert (o_cnt.cpu().numpy() == k).all()
    for c in shards:
        c.free()
    corpus.free()


@pytest.mark.parametrize("n", [500_000, 3_000_000])      # K2w (one candidate buffer per query) / K2 (filters too big for it)
def test_seeded_thresholds_on_clustered_rows(ctx, oracle, sift10m, n):
    """Threshold seeding samples a fraction of the tiles.  Here the true top-k of every query sits in ONE document (100
    near-copies of the query planted in it), i.e. in a couple of tiles: whether or not the sample happens to hit them, the
    answer must be exact -- either the seed holds, or the query is flagged and the host API re-runs it."""
    import vsrbac
    _, x, blk, doc, rbac = sift10m
    xs = x[:n].copy()
    rng = np.random.default_rng(77)
    nq, k = 40, 100
    q = np.clip(np.rint(np.abs(rng.normal(0, 45, (nq, 128)))), 0, 255).astype(np.float32)
    docs = rng.choice(np.arange(50, n // 100 - 50), nq, replace=False)
    for i, d in enumerate(docs):                                       # document d = rows [100 (d-1), 100 d)
        noise = rng.integers(-1, 2, (100, 128)).astype(np.float32)
        xs[100 * (d - 1):100 * d] = np.clip(q[i] + noise, 0, 255)
    corpus = ctx.load_corpus(xs, blk[:n], doc[:n])
    res = corpus.search(q, k, "l2")                                    # unfiltered: one fat pass, seeded main launch
    for i in range(0, nq, 5):
        idx, dist = oracle.filtered_topk("l2", xs, q[i], k, doc[:n], blk[:n])
        np.testing.assert_array_equal(res.rows[i], idx)
        np.testing.assert_array_equal(res.dist[i], dist.astype(np.float32))
    for i in range(nq):                                                # every query: its planted document fills the answer
        assert (res.doc_ids[i] == docs[i]).all()
    # the device API on the same batch: any flagged query is reported, none is silently wrong
    import torch
    dev = torch.device("cuda", 0)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    d_q = torch.from_numpy(q).to(dev)
    o = {"blk": torch.empty((nq, k), dtype=torch.int64, device=dev), "doc": torch.empty((nq, k), dtype=torch.int32, device=dev),
         "row": torch.empty((nq, k), dtype=torch.int64, device=dev), "dist": torch.empty((nq, k), dtype=torch.float32, device=dev),
         "cnt": torch.empty((nq,), dtype=torch.int32, device=dev)}
    corpus.search_device(p(d_q), nq, k, "l2", None, p(o["blk"]), p(o["doc"]), p(o["row"]), p(o["dist"]), p(o["cnt"]))
    _, flags = ctx.screening_check(nq)
    rows = o["row"].cpu().numpy()
    for i in range(nq):
        if not flags[i]:
            np.testing.assert_array_equal(rows[i], res.rows[i])
    # screening (and with it seeding) disabled: exact and never flagged
    ctx.set_screening(False)
    before = ctx.screening_check(0)[0]
    corpus.search_device(p(d_q), nq, k, "l2", None, p(o["blk"]), p(o["doc"]), p(o["row"]), p(o["dist"]), p(o["cnt"]))
    total, flags = ctx.screening_check(nq)
    ctx.set_screening(True)
    assert total == before and not flags.any()
    np.testing.assert_array_equal(o["row"].cpu().numpy(), res.rows)
    corpus.free()


# ---------------------------------------------------------------------------------------------
# C3: Wikipedia-like 1M x 768 cosine, dynamic partitions (pure and impure), k = 100
# ---------------------------------------------------------------------------------------------
def test_config3_wikipedia1m_dynamic_partitions(ctx, oracle):
    from vsrbac.datasets import gaussian_corpus, tree_rbac
    from vsrbac.harness import Deployment
    n, dim, k = 1_000_000, 768, 100
    x, blk, doc = gaussian_corpus(n, dim, seed=33, normalize=True, blocks_per_doc=10)
    ndocs = int(doc.max())
    rbac = tree_rbac(num_users=200, num_roles=40, num_docs=ndocs, seed=33)
    dep = Deployment(ctx, x, blk, doc, rbac.user_roles, rbac.permissions, metric="cosine")
    # 12 partitions: documents dealt round-robin inside each role's own set, so that a combination's documents spread over
    # several partitions and every partition also holds documents the combination may NOT see (impure), plus two
    # partitions made of the whole visible set of one leaf role each (pure for that role)
    rng = np.random.default_rng(34)
    part_docs = {p: [] for p in range(12)}
    for d in range(1, ndocs + 1):
        part_docs[d % 10].append(d)
    leaf_roles = [r for r in rbac.role_docs if r not in set(rbac.parent.values())][:2]
    for j, r in enumerate(leaf_roles):
        part_docs[10 + j] = [int(d) for d in rbac.role_docs[r]]       # (those documents live only in the pure partition)
    pure_docs = set(part_docs[10]) | set(part_docs[11])
    for p in range(10):
        part_docs[p] = [d for d in part_docs[p] if d not in pure_docs]
    combs = {}
    for u in range(1, 201):
        roles = tuple(sorted(rbac.roles_of(u)))
        vis = set(int(d) for d in rbac.visible_docs(u))
        combs[roles] = [p for p, ds in part_docs.items() if vis & set(ds)]
    dep.load_partitions(part_docs, combs)
    users = rng.integers(1, 201, 24)
    qs = x[rng.integers(0, n, 24)] + 0.05 * rng.normal(size=(24, dim)).astype(np.float32)
    saw_pure = saw_impure = False
    x64 = x.astype(np.float64)
    xn2 = (x64 ** 2).sum(1)
    for i, (u, qv) in enumerate(zip(users, qs)):
        rows, secs = dep.dynamic_partition_search(int(u), qv, k, "system")
        assert len(rows) == k and secs > 0
        got_ids = [(r[1], r[0]) for r in rows]
        assert len(set(got_ids)) == k
        d = np.asarray([r[3] for r in rows])
        assert (np.diff(d) >= -1e-6).all()
        vis = set(int(v) for v in rbac.visible_docs(int(u)))
        assert all(r[1] in vis for r in rows)
        roles = tuple(sorted(rbac.roles_of(int(u))))
        for pid in combs[roles]:
            f = dep._partition_filter(pid, int(u))
            pure = f.allowed_rows == f.scanned_rows
            saw_pure |= pure
            saw_impure |= not pure
        if i % 6 == 0:                                                 # the oracle on a sample (~1 s per query at this size)
            mask = oracle.user_row_mask(int(u), rbac.user_roles, rbac.permissions, doc)
            oidx, odist = oracle.filtered_topk("cosine", x, qv, k, doc, blk, mask)
            q64 = qv.astype(np.float64)
            ref = 1.0 - np.clip((x64 @ q64) / np.sqrt(xn2 * (q64 ** 2).sum()), -1, 1)
            got_rows = np.asarray([r[0] - 1 for r in rows])            # block_id = row + 1
            assert_valid_topk(got_rows, d, ref, k, TOL, candidates=np.flatnonzero(mask))
            np.testing.assert_allclose(d, odist, rtol=TOL, atol=TOL)
            assert len(set(got_rows.tolist()) & set(oidx.tolist())) >= k - 1
    assert saw_pure and saw_impure
    dep.close()


# ---------------------------------------------------------------------------------------------
# C5: Wikipedia-like 768-d, 1000 batched queries (GEMM path), byte-mask predicate AND RBAC, cosine, 8 shards
# at its full size: 5M x 768 = 15.4 GB of rows on the one GPU
# ---------------------------------------------------------------------------------------------
def test_config5_batched_1000_queries_predicate_and_rbac(ctx, oracle):
    import torch
    import vsrbac
    from vsrbac.datasets import gaussian_corpus, tree_rbac
    from vsrbac.sharded import shard_bounds
    n, dim, k, nq = 5_000_000, 768, 100, 1000
    x, blk, doc = gaussian_corpus(n, dim, seed=55, normalize=True, blocks_per_doc=10)
    ndocs = int(doc.max())
    rbac = tree_rbac(num_users=100, num_roles=20, num_docs=ndocs, seed=55)
    rng = np.random.default_rng(56)
    users = rng.integers(1, 101, 4)                                    # 4 (user, predicate) groups of 250 queries
    preds = [(rng.random(n) < p).astype(np.uint8) for p in (0.5, 0.2, 0.05, 0.9)]
    masks = []
    for u, pr in zip(users, preds):                                    # ACORN-style predicate AND the user's RBAC visibility
        masks.append(pr & oracle.user_row_mask(int(u), rbac.user_roles, rbac.permissions, doc))
    q = x[rng.integers(0, n, nq)] + 0.05 * rng.normal(size=(nq, dim)).astype(np.float32)
    corpus = ctx.load_corpus(x, blk, doc)
    fs = [corpus.filter_from_bytemask(m) for m in masks]
    filters = [fs[i % 4] for i in range(nq)]
    res = corpus.search(q, k, "cosine", filters)
    def cosine_ref(qv):                                                # float64 reference, in row chunks (memory)
        q64 = qv.astype(np.float64)
        out = np.empty(n, dtype=np.float64)
        for a in range(0, n, 500_000):
            c64 = x[a:a + 500_000].astype(np.float64)
            out[a:a + 500_000] = 1.0 - np.clip((c64 @ q64) / np.sqrt((c64 ** 2).sum(1) * (q64 ** 2).sum()), -1, 1)
        return out

    for i in range(0, nq, 250):                                        # oracle + float64 reference on a sample
        cand = np.flatnonzero(masks[i % 4])
        m = res.counts[i]
        assert m == min(k, cand.size)
        assert_valid_topk(res.rows[i, :m], res.dist[i, :m], cosine_ref(q[i]), k, TOL, candidates=cand)
        oidx, odist = oracle.filtered_topk("cosine", x, q[i], k, doc, blk, masks[i % 4])
        np.testing.assert_allclose(res.dist[i, :m], odist, rtol=TOL, atol=TOL)
        assert len(set(res.rows[i, :m].tolist()) & set(oidx.tolist())) >= k - 1
    for i in range(nq):                                                # properties on every query
        m = res.counts[i]
        assert m == k and (np.diff(res.dist[i]) >= -1e-6).all()
        assert masks[i % 4][res.rows[i]].all()
    # 8 row-range shards of the same corpus + packed merge: the same rows, distances within the fp32 tolerance
    dev = torch.device("cuda", 0)
    parts = 8
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    rec = ctx.packed_result_bytes(nq, k)
    nk = nq * k
    packed = torch.empty((parts * rec,), dtype=torch.uint8, device=dev)
    d_q = torch.from_numpy(q).to(dev)
    cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    shards = []
    for s in range(parts):
        lo, hi = shard_bounds(n, parts, s, align=10)
        c = ctx.load_corpus(x[lo:hi], blk[lo:hi], doc[lo:hi], row_offset=lo)
        sf = [c.filter_from_bytemask(m[lo:hi]) for m in masks]
        fl = c.pack_filters([sf[i % 4] for i in range(nq)])
        pk = packed[s * rec:(s + 1) * rec]
        c.search_device(p(d_q), nq, k, "cosine", fl, p(pk[nk * 8:nk * 16]), p(pk[nk * 16:nk * 20]), None, p(pk[nk * 20:nk * 24]),
                        p(cnt), p(pk[0:nk * 8]))
        ctx.synchronize()
        _, flags = ctx.screening_check(nq)
        assert not flags.any()
        shards.append((c, sf))
    o_blk = torch.empty((nq, k), dtype=torch.int64, device=dev)
    o_doc = torch.empty((nq, k), dtype=torch.int32, device=dev)
    o_dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    o_cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    ctx.merge_topk_packed_device(p(packed), parts, nq, k, p(o_blk), p(o_doc), p(o_dist), None, p(o_cnt))
    ctx.synchronize()
    mb, md = o_blk.cpu().numpy(), o_dist.cpu().numpy()
    same = (mb == res.block_ids).mean()
    assert same > 0.999, same                                          # a boundary near-tie may swap two ids
    np.testing.assert_allclose(md, res.dist, rtol=TOL, atol=TOL)
    for c, sf in shards:
        c.free()
    corpus.free()
