"""GpuShardEngine + ShardedSearcher (the multi-GPU product classes) with two real ranks on ONE MI355X: both processes
open GPU 0, own one row-range shard each, search it through the C ABI and exchange their packed top-k records (gloo here,
RCCL on a multi-GPU node).  The batch is tie-saturated on purpose: the screening cannot prove exactness for its queries,
so every rank goes through the flagged-query re-run -- with a PACKED filter array, the form bench.py uses -- before the
exchange, and rank 0 compares the merged result with the oracle on the whole corpus."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _corpus():
    rng = np.random.default_rng(12)
    # even integers up to 510: exact in bf16 but outside 0..255, i.e. the bf16 screening (which flags ties), not the int8 planes
    base = 2.0 * np.clip(np.rint(np.abs(rng.normal(0, 45, (8, 128)))), 0, 255).astype(np.float32)
    x = np.repeat(base, 400, axis=0)                     # every vector 400 times: ties far beyond 2k = 200
    x = x[rng.permutation(x.shape[0])]
    n = x.shape[0]
    blk = (np.arange(n) + 1).astype(np.int64)
    doc = (np.arange(n) // 20 + 1).astype(np.int32)      # 160 documents of 20 rows
    perms = [(1, d) for d in range(1, 161, 2)] + [(2, d) for d in range(1, 161)]
    ur = [(u, 1 + u % 2) for u in range(1, 9)]
    return base, x, blk, doc, ur, perms


def _rank_main(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    import vsrbac
    from vsrbac.sharded import GpuShardEngine, ShardedSearcher, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    base, x, blk, doc, ur, perms = _corpus()
    n = x.shape[0]
    lo, hi = shard_bounds(n, world, rank, align=20)
    ctx = vsrbac.Context(0)
    corpus = ctx.load_corpus(x[lo:hi], blk[lo:hi], doc[lo:hi], row_offset=lo)
    corpus.load_rbac(ur, perms)
    engine = GpuShardEngine(ctx, corpus, torch.device("cuda", 0))
    searcher = ShardedSearcher(engine, world, rank, dist)
    users = [1, 2, 3, 4, 5, 6]
    filters = corpus.pack_filters([corpus.filter_for_user(u) for u in users])      # a packed C array, like bench.py
    before = ctx.screening_check(0)[0]
    b, d, dd, c = searcher.search(base[:6], 100, "l2", filters)
    flagged = ctx.screening_check(0)[0] - before
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out_path, blk=b.cpu().numpy(), doc=d.cpu().numpy(), dist=dd.cpu().numpy(), cnt=c.cpu().numpy(),
                 flagged=np.int64(flagged))
    dist.barrier()
    corpus.free()
    ctx.close()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_with_flagged_redo(oracle, tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "merged.npz")
    mp.spawn(_rank_main, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    base, x, blk, doc, ur, perms = _corpus()
    assert int(got["flagged"]) > 0, "the tie-saturated batch should have gone through the flagged-query re-run"
    for i, u in enumerate([1, 2, 3, 4, 5, 6]):
        mask = oracle.user_row_mask(u, ur, perms, doc)
        idx, dist = oracle.filtered_topk("l2", x, base[i], 100, doc, blk, mask)
        m = int(got["cnt"][i])
        assert m == idx.size
        np.testing.assert_array_equal(got["blk"][i, :m], blk[idx])
        np.testing.assert_array_equal(got["doc"][i, :m], doc[idx])
        np.testing.assert_array_equal(got["dist"][i, :m], dist.astype(np.float32))


def test_index_scans_shard_like_exact_scans(ctx_free_gpu, oracle):
    """Per-shard IVFFlat / HNSW indexes behind ShardedSearcher (the reference's one index per partition table + client-side
    merge): two shards on one GPU, the all-gather emulated by concatenating the ranks' packed records.  IVFFlat with
    probes = lists is exhaustive: the merged result equals the oracle's exact filtered top-k.  HNSW: the merged list equals
    the k best of what the two shard graphs return on their own (distance, then global row)."""
    import torch
    import vsrbac
    from vsrbac.sharded import GpuShardEngine, ShardedSearcher, shard_bounds
    rng = np.random.default_rng(31)
    n, dim, k, nq = 24_000, 64, 20, 12
    x = np.clip(np.rint(np.abs(rng.normal(0, 45, (n, dim)))), 0, 255).astype(np.float32)
    blk = (np.arange(n) + 1).astype(np.int64)
    doc = (np.arange(n) // 10 + 1).astype(np.int32)
    mask = (rng.random(int(doc.max()) + 1) < 0.5)[doc].astype(np.uint8)
    q = x[rng.integers(0, n, nq)] + rng.integers(0, 2, (nq, dim)).astype(np.float32)
    dev = torch.device("cuda", 0)
    engines, packs, per_shard = [], [], []
    for r in range(2):
        lo, hi = shard_bounds(n, 2, r, align=10)
        c = vsrbac.Context(0)
        corpus = c.load_corpus(x[lo:hi], blk[lo:hi], doc[lo:hi], row_offset=lo)
        e = GpuShardEngine(c, corpus, dev)
        ivf, _, _ = corpus.build_ivf(x[lo:hi], 16, "l2", seed=3)
        e.attach_index("ivf", ivf)
        e.attach_index("hnsw", corpus.build_hnsw(8, 32, "l2", seed=5))
        e.filters = [corpus.filter_from_bytemask(mask[lo:hi], vsrbac.BITMAP)] * nq
        engines.append((e, c, corpus, lo))
    for kind, params in (("ivf", {"probes": 16}), ("hnsw", {"ef_search": 64})):
        locs = [e.search_local_index(kind, q, k, "l2", e.filters, **params) for e, _, _, _ in engines]
        torch.cuda.synchronize()
        g = torch.cat([l["pack"] for l in locs])
        b, d, dd, cnt = engines[0][0].merge_packed(g, nq, k)
        torch.cuda.synchronize()
        b, dd, cnt = b.cpu().numpy(), dd.cpu().numpy(), cnt.cpu().numpy()
        for i in range(nq):
            if kind == "ivf":                                       # exhaustive probe: the exact answer
                idx, dist = oracle.filtered_topk("l2", x, q[i], k, doc, blk, mask)
                assert cnt[i] == idx.size
                np.testing.assert_array_equal(b[i, :cnt[i]], blk[idx])
                np.testing.assert_array_equal(dd[i, :cnt[i]], dist.astype(np.float32))
            else:                                                   # the k best of the two graphs' own answers
                cand = []
                for (e, c, corpus, lo) in engines:
                    res, _ = e.indexes["hnsw"].search(q[i][None, :], k, 64, "l2", [e.filters[i]])
                    cand += [(float(res.dist[0, j]), int(res.rows[0, j]) + lo, int(res.block_ids[0, j])) for j in range(int(res.counts[0]))]
                cand.sort()
                want = cand[:k]
                assert cnt[i] == len(want)
                np.testing.assert_array_equal(b[i, :cnt[i]], np.array([w[2] for w in want], dtype=np.int64))
                assert (np.diff(dd[i, :cnt[i]]) >= 0).all()
    # the collective path of ShardedSearcher with world = 1 takes the same entry point
    s1 = ShardedSearcher(engines[0][0], 1, 0, None)
    b1, d1, dd1, c1 = s1.search(q, k, "l2", engines[0][0].filters, index="ivf", probes=16)
    assert int(c1.cpu()[0]) > 0
    for e, c, corpus, lo in engines:
        for ix in e.indexes.values():
            ix.free()
        corpus.free()
        c.close()


@pytest.fixture
def ctx_free_gpu():
    yield None
