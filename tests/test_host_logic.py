"""Host logic that needs no GPU: synthetic data generators, harness helpers, shard arithmetic, ordering keys."""
import json

import numpy as np
import pytest


def test_tree_rbac_semantics():
    """tree_based_rbac_data_generator.py:48-187: disjoint own sets, ancestor inheritance, one role per user."""
    from vsrbac.datasets import tree_rbac
    r = tree_rbac(num_users=1000, num_roles=100, num_docs=10_000, seed=3)
    assert len(r.role_docs) == 100 and len(r.user_roles) == 1000
    assert len(set(r.user_roles[:, 0])) == 1000                       # exactly one role per user
    own = {}
    for role, docs in r.role_docs.items():
        p = r.parent[role]
        own[role] = np.setdiff1d(docs, r.role_docs[p]) if p else docs
        if p:
            assert np.isin(r.role_docs[p], docs).all()                 # inherits every ancestor document
    allown = np.concatenate(list(own.values()))
    assert allown.size == 10_000 and np.unique(allown).size == 10_000  # own sets partition the documents
    depth = lambda x: 0 if x == 0 else 1 + depth(r.parent[x])
    for role, docs in r.role_docs.items():
        assert len(docs) in (100 * depth(role), 100 * depth(role) + 10_000 % 100)
    sel = len(r.permissions) / (100 * 10_000)
    assert 0.02 < sel < 0.05                                           # reference probe: 3.56 % mean selectivity
    again = tree_rbac(num_users=1000, num_roles=100, num_docs=10_000, seed=3)
    assert (again.permissions == r.permissions).all()


def test_random_rbac_multi_role():
    from vsrbac.datasets import random_rbac
    r = random_rbac(num_users=200, num_roles=20, num_docs=500, m_roles=3, m_perms=100, seed=1)
    per_user = np.bincount(r.user_roles[:, 0])[1:]
    assert per_user.min() >= 1 and per_user.max() <= 3 and per_user.max() > 1
    assert np.unique(r.permissions[:, 1]).size == 500                  # every document assigned at least once


def test_sift_like_rows_are_shard_independent():
    from vsrbac.datasets import sift_like_corpus, sift_like_rows, sift_like_rows_at
    x, blk, doc = sift_like_corpus(3000, 128, seed=5)
    assert x.dtype == np.float32 and x.min() >= 0 and x.max() <= 255 and (x == np.rint(x)).all()
    assert blk[0] == 1 and doc[0] == 1 and doc[100] == 2               # read_dataset_function.py:338-339
    np.testing.assert_array_equal(sift_like_rows(1000, 2000, 128, 5), x[1000:2000])
    np.testing.assert_array_equal(sift_like_rows_at([7, 2999, 7], 128, 5), x[[7, 2999, 7]])
    y, blk2, doc2 = sift_like_corpus(500, 128, seed=5, start=2500)
    np.testing.assert_array_equal(y, x[2500:])
    assert blk2[0] == 2501 and doc2[0] == 26


def test_shard_bounds_cover_and_align():
    from vsrbac.sharded import shard_bounds
    for n, world in ((10_000_000, 8), (1_000_003, 4), (700, 3), (5, 8)):
        cuts = [shard_bounds(n, world, r, 100 if n > 1000 else 1) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        for a, b in zip(cuts, cuts[1:]):
            assert a[1] == b[0] and a[0] <= a[1]
        if n > 1000:
            assert all(c[0] % 100 == 0 for c in cuts)


def test_monotone_keys_order_like_oracle(oracle):
    """Sorting by the 64-bit key must equal the oracle's (distance, NaN last, row) order."""
    from vsrbac.sharded import monotone_keys
    vals = np.asarray([3.5, -0.0, 0.0, -2.0, np.inf, -np.inf, np.nan, 1e-30, -1e-30, 3.5], dtype=np.float32)
    rows = np.arange(vals.size)
    order = np.argsort(monotone_keys(vals, rows), kind="stable")
    want = sorted(range(vals.size), key=lambda i: (np.isnan(vals[i]), float(vals[i]) + 0.0 if not np.isnan(vals[i]) else 0, i))
    assert order.tolist() == want


def test_harness_helpers_match_oracle(oracle, tmp_path):
    from vsrbac import harness
    np.testing.assert_array_equal(harness.parse_vector("[1, 2.5,-3]"), np.asarray([1, 2.5, -3], np.float32))
    with pytest.raises(ValueError):
        harness.parse_vector("1,2,3")
    rng = np.random.default_rng(0)
    rows = [(int(b), int(d), None, float(x)) for b, d, x in zip(rng.integers(1, 30, 200), rng.integers(1, 5, 200),
                                                                 rng.integers(0, 20, 200))]
    got = harness.merge_results(list(rows), 25)
    idx = oracle.merge_dedup([r[3] for r in rows], [r[1] for r in rows], [r[0] for r in rows], 25)
    assert got == [rows[i] for i in idx]                               # search.py:347-364 semantics
    gt = [(1, 1), (1, 2), (2, 3), (2, 4)]
    pr = [(1, 1), (2, 4), (9, 9)]
    assert harness.compute_recall(gt, pr) == oracle.recall(gt, pr) == 0.5
    # both ground-truth cache formats (common_function.py:831-849, compute_ground_truth.py:35-59)
    p1, p2 = tmp_path / "a.json", tmp_path / "b.json"
    p1.write_text(json.dumps([{"query": {"user_id": 1}, "ground_truth": [[5, 1, "txt", 0.1], [6, 1, "t", 0.2]]}]))
    p2.write_text(json.dumps([[[5, 1], [6, 1]]]))
    assert harness.load_ground_truth_cache(p1) == harness.load_ground_truth_cache(p2) == [[(5, 1), (6, 1)]]
    p3 = tmp_path / "q.json"
    p3.write_text(json.dumps([{"user_id": 3, "query_vector": "[1,2]", "topk": 10, "query_block_selectivity": 0.03}]))
    assert harness.load_query_dataset(p3)[0]["topk"] == 10


def test_run_search_experiment_protocol():
    """Two warm-up calls + one measured call per query; recall on (document_id, block_id) sets."""
    from vsrbac import harness
    calls = []

    def search_func(user_id, query_vector, topk, statistics_type):
        calls.append(user_id)
        return [(1, 10, None, 0.0), (2, 10, None, 1.0)], 0.004

    def gt(user_id, qv, topk):
        return [(1, 10, None, 0.0), (3, 10, None, 0.5)]

    out = harness.run_search_experiment([{"user_id": 7, "query_vector": [0.0], "topk": 2}] * 3, search_func, gt)
    assert len(calls) == 9
    assert out["avg_recall"] == 0.5 and abs(out["avg_query_time"] - 0.004) < 1e-12 and abs(out["qps"] - 250) < 1e-6
    assert set(out["all_results"][0]) == {"user_id", "query_vector", "recall", "query_time", "qps"}


def test_planner_output_to_tables():
    """The planner's in-memory dicts (sets, frozenset / tuple combinations, {partition: roles} trackers) become the two
    tables Deployment.load_partitions takes; unknown partitions are an error, not a silent hole in the result."""
    from vsrbac.harness import planner_output_to_tables
    pa = {0: {5, 1, 3}, 7: {2}, 9: set()}
    trackers = {frozenset({2, 1}): {0: {1, 2}, 7: {2}}, (3,): {9: {3}}, 4: {0: {4}}}
    docs, combs = planner_output_to_tables(pa, trackers)
    assert docs == {0: [1, 3, 5], 7: [2], 9: []}
    assert combs == {(1, 2): [0, 7], (3,): [9], (4,): [0]}
    import pytest
    with pytest.raises(ValueError, match="unknown partitions"):
        planner_output_to_tables(pa, {(1,): {42: {1}}})


def test_gpu_cost_model_and_placement():
    """f3: the planner's objective with the GPU's cost of a partition search (same argument list as the reference's
    compute_query_time, AnonySys_dynamic_partition.py:114) and LPT placement of partitions on GPUs."""
    from vsrbac.placement import compute_query_time_gpu, partition_heat, place_partitions
    comb_trackers = {(1,): {0, 1}, (2,): {1, 2}, (1, 2): {0, 1, 2, 3}}
    loads = {0: 1000, 1: 50000, 2: 2000, 3: 300}
    w = {(1,): 5.0, (2,): 1.0, (1, 2): 0.5}
    # the reference's positional call shape (sel_whole, topk, k, beta, a, b are HNSW tuning values: ignored here)
    t_all = compute_query_time_gpu(comb_trackers, loads, 0.1, 10, 0.5, 0.2, 1.0, 0.0, set(comb_trackers), w)
    t_one = compute_query_time_gpu(comb_trackers, loads, 0.1, 10, 0.5, 0.2, 1.0, 0.0, {(1,)}, w)
    assert 0 < t_one < t_all
    # moving the big partition out of a heavy combination's cover must lower the objective
    lighter = dict(comb_trackers)
    lighter[(1,)] = {0}
    assert compute_query_time_gpu(lighter, loads, comb_to_update=set(lighter), role_weights=w) < \
        compute_query_time_gpu(comb_trackers, loads, comb_to_update=set(comb_trackers), role_weights=w)
    heat = partition_heat(comb_trackers, loads, w)
    assert max(heat, key=heat.get) == 1
    placement, gpu_heat, gpu_rows = place_partitions(loads, heat, 4)
    assert set(placement) == set(loads) and all(len(set(g)) == len(g) >= 1 for g in placement.values())
    assert len(placement[1]) > 1, "the partition that dominates the load is replicated"
    assert sum(gpu_rows) == sum(loads[p] * len(g) for p, g in placement.items())
    assert max(gpu_heat) <= 0.75 * sum(heat.values()), gpu_heat       # no GPU carries (nearly) everything
    one, h1, _ = place_partitions(loads, heat, 1)
    assert all(g == [0] for g in one.values()) and abs(h1[0] - sum(heat.values())) < 1e-6
    import pytest
    with pytest.raises(ValueError):
        place_partitions(loads, heat, 2, mem_rows=10000)


def test_bench_roofline_arithmetic():
    """bench.py's roofline: unique rows priced in the layout the launched kernel reads (never more than HBM can deliver),
    the fp32-equivalent beside it, and the MFMA floor at the peak of the products' type."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    k_i8 = "vsr::mfma_wide_kernel<L2, NCH=1, SAMPLE=false, PL=int8> (K2w, int8 planes)"
    k_ho = "vsr::mfma_wide_kernel<L2, NCH=1, SAMPLE=false, HO=true> (K2w, bf16 hi-only planes)"
    k_hm = "vsr::mfma_wide_kernel<COSINE, NCH=12, SAMPLE=false, HO=false> (K2w, bf16 hi+mid planes)"
    k_k1 = "vsr::scan_kernel<L2, LPR=32, C=1, R=8, QI=1> (K1)"
    assert bench.kernel_layout(k_i8, 128)[0] == 132 and bench.kernel_layout(k_ho, 128)[0] == 260
    assert bench.kernel_layout(k_hm, 768)[0] == 768 * 4 + 4 and bench.kernel_layout(k_k1, 128)[0] == 512
    st = {"scan_ms": [0.0, 4.0], "scan_launches": [0, 10], "scan_pairs": [0, 10 * 360_000_000], "unique_rows": [0, 10 * 10_000_000],
          "scan_rows": [0, 10 * 13_000_000], "scan_bytes": [0, 0]}
    r = bench.roofline_of(st, 128, k_i8, 3)
    assert r["launch_ms"] == 0.4 and r["unique_bytes"] == 10_000_000 * 132 and r["bound"] == "hbm"
    assert abs(r["achieved"] - 1.32e9 / 0.4e-3 / 1e9) < 1 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-3
    assert r["fp32_equivalent"]["unique_bytes"] == 10_000_000 * 512
    assert r["frac"] <= 1.0 and r["mfma"]["frac"] < 0.1


def test_role_placement_covers_every_query_exactly_once():
    """bench.py's N > 1 placement: every role on one GPU with everything it can see; the ranks' query
    subsets partition every batch; the predicted load is balanced; the rows a rank holds are exactly what its roles see."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from vsrbac.datasets import sample_queries, tree_rbac
    n = 1_000_000
    rbac = tree_rbac(num_users=1000, num_roles=100, num_docs=n // 100, seed=20251121)
    role_of = {int(u): rs[0] for u, rs in rbac._user_roles_map.items()}
    assert all(len(rs) == 1 for rs in rbac._user_roles_map.values())
    users_of = {}
    for u, r in role_of.items():
        users_of[r] = users_of.get(r, 0) + 1
    weights = {r: users_of.get(r, 0) * len(d) for r, d in rbac.role_docs.items()}
    for parts in (2, 4, 8):
        where, load = bench.place_roles(rbac.parent, weights, parts)
        assert sorted(where) == sorted(rbac.role_docs) and set(where.values()) == set(range(parts))
        assert max(load) / (sum(load) / parts) < 1.08                     # a run misses its share by at most one role
        qrow, quser = sample_queries(1000, n, 1000, seed=7)
        seen = np.zeros(1000, dtype=int)
        total_rows = 0
        for g in range(parts):
            mine = np.array([where[role_of[int(u)]] == g for u in quser])
            seen += mine
            docs = np.unique(np.concatenate([rbac.role_docs[r] for r, gg in where.items() if gg == g]))
            total_rows += docs.size * 100
            for u in quser[mine][:20]:                                       # all a user may see is on the user's GPU
                assert np.isin(rbac.visible_docs(int(u)), docs).all()
        assert (seen == 1).all()
        assert total_rows < 1.3 * n                                          # replication of the classes near the root stays modest
        # the steps of the two scaling modes: weak = N x 1000 queries per step (about 1000 per rank), strong = the N = 1 step's
        # 1000 spread over the ranks; either way the ranks' shares partition the step
        for mult in (parts, 1):
            per_rank = [bench.draw_rank_queries(sample_queries, where, role_of, g, 1000, n, 20251121, mult, 2, 0) for g in range(parts)]
            for b in range(2):
                assert sum(len(per_rank[g][b][0]) for g in range(parts)) == 1000 * mult
                if mult == parts:
                    assert all(800 < len(per_rank[g][b][0]) < 1250 for g in range(parts))
            assert all(where[role_of[int(u)]] == g for g in range(parts) for u in per_rank[g][0][1])


def test_device_keys_of_index_results_match_the_library_key():
    """GpuShardEngine.device_keys (torch, used to merge per-shard index results) builds the library's ordering key:
    identical to sharded.monotone_keys for finite values, -0 == +0, NaN last among values, empty slots last of all."""
    import types
    import torch
    from vsrbac.sharded import GpuShardEngine, monotone_keys
    dist = np.array([0.0, -0.0, 1.5, -2.25, np.inf, -np.inf, np.nan, 3.0e38, 1e-40], dtype=np.float32)
    rows = np.arange(dist.size, dtype=np.int64) + 5
    got = GpuShardEngine.device_keys(types.SimpleNamespace(torch=torch), torch.from_numpy(dist), torch.from_numpy(rows), 1000)
    want = monotone_keys(dist, rows + 1000)
    np.testing.assert_array_equal(got.numpy().view(np.uint64), want)
    empty = GpuShardEngine.device_keys(types.SimpleNamespace(torch=torch), torch.tensor([np.inf]), torch.tensor([-1]), 7)
    assert int(empty.numpy().view(np.uint64)[0]) == 0xFFFFFFFFFFFFFFFF
    order = np.argsort(got.numpy().view(np.uint64))
    assert list(dist[order][:2]) == [-np.inf, -2.25] and np.isnan(dist[order][-1])
