"""GPU tests of the host mirror (reference search-function contract) and of the sharded merge path."""
import ctypes
import json
import os

import numpy as np
import pytest

from helpers import sift_like

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import vsrbac
    c = vsrbac.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def world(ctx, golden_dir):
    with open(os.path.join(golden_dir, "rbac_random_small.json")) as f:
        fx = json.load(f)
    rng = np.random.default_rng(21)
    rows_per_doc = 12
    n = fx["params"]["num_docs"] * rows_per_doc
    x = sift_like(rng, n)
    blk = (np.arange(n) + 1).astype(np.int64)
    doc = (np.arange(n) // rows_per_doc + 1).astype(np.int32)
    return fx, x, blk, doc


def _vec_text(v):
    return "[" + ",".join(repr(float(t)) for t in v) + "]"


def test_search_functions_match_oracle(ctx, oracle, world):
    """search_func(user_id, query_vector, topk, statistics_type) -> (rows, seconds) with rows =
    (block_id, document_id, block_content, distance), as prefilter_role.py:22-26 / row_level_security.py / search.py."""
    from vsrbac.harness import Deployment, run_search_experiment
    fx, x, blk, doc = world
    dep = Deployment(ctx, x, blk, doc, fx["user_roles"], fx["permissions"])
    rng = np.random.default_rng(22)
    queries = [{"user_id": int(u), "query_vector": _vec_text(x[r]), "topk": 10, "query_block_selectivity": 0.0}
               for u, r in zip(rng.integers(1, fx["num_users"] + 1, 12), rng.integers(0, len(x), 12))]

    def oracle_gt(user_id, qv, topk):
        from vsrbac.harness import parse_vector
        mask = oracle.user_row_mask(user_id, fx["user_roles"], fx["permissions"], doc)
        idx, d = oracle.filtered_topk("l2", x, parse_vector(qv), topk, doc, blk, mask)
        return [(int(blk[i]), int(doc[i]), None, float(dd)) for i, dd in zip(idx, d)]

    for fn in (dep.search_documents_role_partition, dep.search_documents_rls):
        for stats in ("sql", "system"):
            rows, secs = fn(queries[0]["user_id"], queries[0]["query_vector"], 10, stats)
            want = oracle_gt(queries[0]["user_id"], queries[0]["query_vector"], 10)
            assert [(r[0], r[1]) for r in rows] == [(w[0], w[1]) for w in want]
            assert [np.float32(r[3]) for r in rows] == [np.float32(w[3]) for w in want]
            assert secs > 0
        out = run_search_experiment(queries, fn, oracle_gt)
        assert out["avg_recall"] == 1.0 and out["qps"] > 0 and len(out["all_results"]) == len(queries)

    # dynamic partitions: documents split into 6 partitions; every combination maps to the partitions that
    # hold at least one document it may see (a valid CombRolePartitions table); some partitions are impure
    ndocs = fx["params"]["num_docs"]
    part_docs = {p: [d for d in range(1, ndocs + 1) if d % 6 == p] for p in range(6)}
    role_docs = {}
    for r, d in fx["permissions"]:
        role_docs.setdefault(r, set()).add(d)
    combs = {}
    for u in range(1, fx["num_users"] + 1):
        roles = tuple(sorted(r for uu, r in fx["user_roles"] if uu == u))
        vis = set().union(*[role_docs.get(r, set()) for r in roles]) if roles else set()
        combs[roles] = [p for p, ds in part_docs.items() if vis & set(ds)]
    dep.load_partitions(part_docs, combs)
    out = run_search_experiment(queries, dep.dynamic_partition_search, oracle_gt)
    assert out["avg_recall"] == 1.0
    rows, _ = dep.dynamic_partition_search(queries[3]["user_id"], queries[3]["query_vector"], 10)
    want = oracle_gt(queries[3]["user_id"], queries[3]["query_vector"], 10)
    assert [(r[0], r[1], np.float32(r[3])) for r in rows] == [(w[0], w[1], np.float32(w[3])) for w in want]
    dep.close()


def test_placed_partitions_equal_single_corpus(ctx, oracle, world):
    """whole partitions placed on GPUs by LPT (hot ones replicated); a query touches only the
    GPUs holding its combination's partitions and the merged rows equal the one-corpus dynamic-partition search and
    the oracle.  Three contexts of the one GPU stand in for three GPUs."""
    import vsrbac
    from vsrbac.harness import Deployment, parse_vector
    from vsrbac.placement import PlacedDeployment
    fx, x, blk, doc = world
    ndocs = fx["params"]["num_docs"]
    # overlapping partitions (impure + shared documents): partition p holds documents with d % 5 in {p, p + 1}
    part_docs = {p: [d for d in range(1, ndocs + 1) if d % 5 in (p, (p + 1) % 5)] for p in range(5)}
    role_docs = {}
    for r, d in fx["permissions"]:
        role_docs.setdefault(r, set()).add(d)
    combs = {}
    for u in range(1, fx["num_users"] + 1):
        roles = tuple(sorted(r for uu, r in fx["user_roles"] if uu == u))
        vis = set().union(*[role_docs.get(r, set()) for r in roles]) if roles else set()
        need, covered = [], set()
        for p, ds in part_docs.items():                      # a cover of the visible documents, like the planner's
            gain = (vis & set(ds)) - covered
            if gain:
                need.append(p)
                covered |= gain
        combs[roles] = need
    weights = {c: 1.0 + 9.0 * (i == 0) for i, c in enumerate(sorted(combs))}     # one hot combination
    ctxs = [ctx, vsrbac.Context(0), vsrbac.Context(0)]
    placed = PlacedDeployment(ctxs, x, blk, doc, fx["user_roles"], fx["permissions"], part_docs, combs, weights)
    assert set(placed.placement) == set(part_docs) and all(gs for gs in placed.placement.values())
    assert max(len(gs) for gs in placed.placement.values()) > 1, "the hot partitions should have been replicated"
    single = Deployment(ctx, x, blk, doc, fx["user_roles"], fx["permissions"])
    single.load_partitions(part_docs, combs)
    rng = np.random.default_rng(5)
    for u, r in zip(rng.integers(1, fx["num_users"] + 1, 10), rng.integers(0, len(x), 10)):
        qv = _vec_text(x[r])
        got, _ = placed.dynamic_partition_search(int(u), qv, 10)
        want, _ = single.dynamic_partition_search(int(u), qv, 10)
        assert [(g[0], g[1], np.float32(g[3])) for g in got] == [(w[0], w[1], np.float32(w[3])) for w in want]
        mask = oracle.user_row_mask(int(u), fx["user_roles"], fx["permissions"], doc)
        idx, d = oracle.filtered_topk("l2", x, parse_vector(qv), 10, doc, blk, mask)
        assert [(g[0], g[1]) for g in got] == [(int(blk[i]), int(doc[i])) for i in idx]
    single.close()
    placed.close()
    for c in ctxs[1:]:
        c.close()


@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("parts", [2, 3, 8])
def test_sharded_merge_equals_single_gpu(ctx, oracle, world, parts, shared):
    """Row-range shards with row_offset, per-shard vsr_search_device, stacked like an all-gather, merged with
    vsr_merge_topk_device: must equal the unsharded search and the oracle (multi-GPU path on one GPU)."""
    import torch
    import vsrbac
    from vsrbac.sharded import shard_bounds
    fx, x, blk, doc = world
    n, k, nq = len(x), 40, 6
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(23)
    users = rng.integers(1, fx["num_users"] + 1, nq)
    if shared:                                # all queries of one user: one shared pass per shard (K2 / K1m path)
        users[:] = users[0]
    q = x[rng.integers(0, n, nq)]
    d_q = torch.from_numpy(q).to(dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    shards = []
    g = {"keys": torch.empty((parts, nq, k), dtype=torch.int64, device=dev),
         "block": torch.empty((parts, nq, k), dtype=torch.int64, device=dev),
         "doc": torch.empty((parts, nq, k), dtype=torch.int32, device=dev),
         "dist": torch.empty((parts, nq, k), dtype=torch.float32, device=dev)}
    cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    for r in range(parts):
        lo, hi = shard_bounds(n, parts, r, align=12)
        c = ctx.load_corpus(x[lo:hi], blk[lo:hi], doc[lo:hi], row_offset=lo)
        c.load_rbac(fx["user_roles"], fx["permissions"])
        filters = [c.filter_for_user(int(u), vsrbac.RANGES) for u in users]
        c.search_device(p(d_q), nq, k, "l2", filters, p(g["block"][r]), p(g["doc"][r]), None, p(g["dist"][r]),
                        p(cnt), p(g["keys"][r]))
        shards.append(c)
    o_blk = torch.empty((nq, k), dtype=torch.int64, device=dev)
    o_doc = torch.empty((nq, k), dtype=torch.int32, device=dev)
    o_dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    o_cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    ctx.merge_topk_device(p(g["keys"]), p(g["block"]), p(g["doc"]), p(g["dist"]), parts, nq, k,
                          p(o_blk), p(o_doc), p(o_dist), None, p(o_cnt))
    # the same merge from packed per-shard records (what a single all-gather delivers)
    ctx.synchronize()                                   # the library runs on its own stream here
    rec = ctx.packed_result_bytes(nq, k)
    packed = torch.empty((parts * rec,), dtype=torch.uint8, device=dev)
    nk = nq * k
    for r in range(parts):
        base = r * rec
        packed[base:base + nk * 8] = g["keys"][r].contiguous().view(torch.uint8).flatten()
        packed[base + nk * 8:base + nk * 16] = g["block"][r].contiguous().view(torch.uint8).flatten()
        packed[base + nk * 16:base + nk * 20] = g["doc"][r].contiguous().view(torch.uint8).flatten()
        packed[base + nk * 20:base + nk * 24] = g["dist"][r].contiguous().view(torch.uint8).flatten()
    p_blk, p_doc, p_dist, p_cnt = (torch.empty_like(o_blk), torch.empty_like(o_doc), torch.empty_like(o_dist),
                                   torch.empty_like(o_cnt))
    torch.cuda.synchronize()
    ctx.merge_topk_packed_device(p(packed), parts, nq, k, p(p_blk), p(p_doc), p(p_dist), None, p(p_cnt))
    ctx.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(p_blk, o_blk) and torch.equal(p_doc, o_doc) and torch.equal(p_dist, o_dist) and torch.equal(p_cnt, o_cnt)
    for i in range(nq):
        mask = oracle.user_row_mask(int(users[i]), fx["user_roles"], fx["permissions"], doc)
        idx, d = oracle.filtered_topk("l2", x, q[i], k, doc, blk, mask)
        m = int(o_cnt[i])
        assert m == idx.size
        np.testing.assert_array_equal(o_blk[i, :m].cpu().numpy(), blk[idx])
        np.testing.assert_array_equal(o_doc[i, :m].cpu().numpy(), doc[idx])
        np.testing.assert_array_equal(o_dist[i, :m].cpu().numpy(), d.astype(np.float32))
    for c in shards:
        c.free()


def test_two_sessions_in_flight_over_one_corpus(ctx, oracle, world):
    """vsr_search_device_on: batches alternate between the corpus's own context and a second session (its own stream
    and workspaces) without any synchronisation in between; every batch must equal the oracle, flags stay per session."""
    import torch
    import vsrbac
    fx, x, blk, doc = world
    n, k, nq, rounds = len(x), 25, 40, 6
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(29)
    corpus = ctx.load_corpus(x, blk, doc)
    corpus.load_rbac(fx["user_roles"], fx["permissions"])
    other = vsrbac.Context(0)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    batches = []
    for r in range(rounds):
        users = rng.integers(1, fx["num_users"] + 1, nq)
        q = x[rng.integers(0, n, nq)] + rng.integers(-2, 3, (nq, x.shape[1])).astype(np.float32)
        d_q = torch.from_numpy(q).to(dev)
        out = {"blk": torch.empty((nq, k), dtype=torch.int64, device=dev), "doc": torch.empty((nq, k), dtype=torch.int32, device=dev),
               "row": torch.empty((nq, k), dtype=torch.int64, device=dev), "dist": torch.empty((nq, k), dtype=torch.float32, device=dev),
               "cnt": torch.empty((nq,), dtype=torch.int32, device=dev)}
        batches.append((users, q, d_q, out))
    torch.cuda.synchronize()                                          # inputs are in place; the sessions use their own streams
    for r, (users, q, d_q, out) in enumerate(batches):
        filters = corpus.pack_filters([corpus.filter_for_user(int(u), vsrbac.RANGES if r % 3 else vsrbac.BITMAP) for u in users])
        corpus.search_device(p(d_q), nq, k, "l2", filters, p(out["blk"]), p(out["doc"]), p(out["row"]), p(out["dist"]),
                             p(out["cnt"]), None, session=other if r % 2 else None)
    ctx.synchronize()
    other.synchronize()
    assert ctx.screening_check(0)[0] >= 0 and other.screening_check(0)[0] >= 0
    for users, q, d_q, out in batches:
        rows, dist, cnt = out["row"].cpu().numpy(), out["dist"].cpu().numpy(), out["cnt"].cpu().numpy()
        for i in range(0, nq, 3):
            mask = oracle.user_row_mask(int(users[i]), fx["user_roles"], fx["permissions"], doc)
            idx, d = oracle.filtered_topk("l2", x, q[i], k, doc, blk, mask)
            assert cnt[i] == idx.size
            np.testing.assert_array_equal(rows[i, :idx.size], idx)
            np.testing.assert_array_equal(dist[i, :idx.size], d.astype(np.float32))
    with pytest.raises(vsrbac.VsrError):                              # NULL outputs are still rejected on a session
        corpus.search_device(p(batches[0][2]), nq, k, "l2", None, None, None, None, None, None, None, session=other)
    corpus.free()
    other.close()
