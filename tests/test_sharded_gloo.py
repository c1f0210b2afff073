"""The N > 1 path on CPU: world_size 2 over gloo.  The collective plumbing and shard arithmetic of
vsrbac.sharded run for real; the per-shard compute is a test double built on the CPU oracle (the product
engine needs a GPU), so the merged result must equal the unsharded oracle's."""
import os
import socket

import numpy as np
import pytest


class OracleShardEngine:
    """Stand-in for GpuShardEngine: same interface, oracle arithmetic, torch CPU tensors."""

    def __init__(self, x, doc, blk, lo, masks):
        import torch
        from oracle.oracle import Oracle
        self.torch, self.orc = torch, Oracle("strict")
        self.x, self.doc, self.blk, self.lo, self.masks = x, doc, blk, lo, masks

    def search_local(self, queries, k, metric, filters):
        from vsrbac.sharded import monotone_keys
        torch = self.torch
        nq = len(queries)
        keys = np.full((nq, k), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
        blk = np.full((nq, k), -1, dtype=np.int64)
        doc = np.full((nq, k), -1, dtype=np.int32)
        dist = np.full((nq, k), np.inf, dtype=np.float32)
        for i, q in enumerate(queries):
            idx, d = self.orc.filtered_topk(metric, self.x, q, k, self.doc, self.blk, self.masks[i])
            m = idx.size
            rank_val = (d.astype(np.float64) ** 2).astype(np.float32) if metric == "l2" else d.astype(np.float32)
            keys[i, :m] = monotone_keys(rank_val, idx + self.lo)
            blk[i, :m], doc[i, :m], dist[i, :m] = self.blk[idx], self.doc[idx], d.astype(np.float32)
        return {"keys": torch.from_numpy(keys.view(np.int64)), "block": torch.from_numpy(blk),
                "doc": torch.from_numpy(doc), "dist": torch.from_numpy(dist)}

    def search_local_index(self, kind, queries, k, metric, filters, probes=None, ef_search=None):
        """An "index" over this shard that looks at every `probes`-th row only (a stand-in for an approximate index: each
        rank's list is the exact answer over the rows its index reaches); keys as GpuShardEngine builds them for index
        results: the RETURNED distance and the global row."""
        from vsrbac.sharded import monotone_keys
        torch = self.torch
        nq = len(queries)
        keys = np.full((nq, k), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
        blk = np.full((nq, k), -1, dtype=np.int64)
        doc = np.full((nq, k), -1, dtype=np.int32)
        dist = np.full((nq, k), np.inf, dtype=np.float32)
        reach = np.zeros(len(self.x), dtype=np.uint8)
        reach[::max(1, int(probes or 1))] = 1
        for i, q in enumerate(queries):
            idx, d = self.orc.filtered_topk(metric, self.x, q, k, self.doc, self.blk, self.masks[i] & reach)
            m = idx.size
            keys[i, :m] = monotone_keys(d.astype(np.float32), idx + self.lo)
            blk[i, :m], doc[i, :m], dist[i, :m] = self.blk[idx], self.doc[idx], d.astype(np.float32)
        return {"keys": torch.from_numpy(keys.view(np.int64)), "block": torch.from_numpy(blk),
                "doc": torch.from_numpy(doc), "dist": torch.from_numpy(dist)}

    def finalize(self, local):
        return local["block"], local["doc"], local["dist"], None

    def merge(self, g, k):
        keys = g["keys"].numpy().view(np.uint64)                      # [P, nq, k]
        P, nq, _ = keys.shape
        flat = lambda t: np.moveaxis(t.numpy(), 0, 1).reshape(nq, P * k)
        fk = np.moveaxis(keys, 0, 1).reshape(nq, P * k)
        order = np.argsort(fk, axis=1, kind="stable")[:, :k]
        take = lambda a: np.take_along_axis(a, order, axis=1)
        cnt = (take(fk) != np.uint64(0xFFFFFFFFFFFFFFFF)).sum(1)
        return take(flat(g["block"])), take(flat(g["doc"])), take(flat(g["dist"])), cnt


def _worker(rank, world, port, n, k, q_out):
    import torch.distributed as dist
    from helpers import sift_like
    from vsrbac.sharded import ShardedSearcher, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(42)                                    # same data on every rank
    x = sift_like(rng, n, 32)
    doc = (np.arange(n) // 10 + 1).astype(np.int32)
    blk = (np.arange(n) + 1).astype(np.int64)
    masks = [(rng.random(n) < 0.2).astype(np.uint8) for _ in range(4)]
    queries = x[[3, 77, 500, 1234]]
    lo, hi = shard_bounds(n, world, rank, align=10)
    eng = OracleShardEngine(x[lo:hi], doc[lo:hi], blk[lo:hi], lo, [m[lo:hi] for m in masks])
    searcher = ShardedSearcher(eng, world, rank, dist)
    out = searcher.search(queries, k, "l2", None)
    out_ix = searcher.search(queries, k, "l2", None, index="ivf", probes=3)       # per-rank index scans, same collectives
    if rank == 0:
        q_out.put([np.asarray(o) for o in out] + [np.asarray(o) for o in out_ix])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gloo_matches_unsharded_oracle(oracle):
    import torch.multiprocessing as mp
    from helpers import sift_like
    n, k, world = 2000, 25, 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    blk, doc, dist, cnt, iblk, idoc, idist, icnt = q.get(timeout=150)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rng = np.random.default_rng(42)
    x = sift_like(rng, n, 32)
    docs = (np.arange(n) // 10 + 1).astype(np.int32)
    blks = (np.arange(n) + 1).astype(np.int64)
    masks = [(rng.random(n) < 0.2).astype(np.uint8) for _ in range(4)]
    for i, qi in enumerate([3, 77, 500, 1234]):
        idx, d = oracle.filtered_topk("l2", x, x[qi], k, docs, blks, masks[i])
        assert cnt[i] == idx.size
        np.testing.assert_array_equal(blk[i, :idx.size], blks[idx])
        np.testing.assert_array_equal(doc[i, :idx.size], docs[idx])
        np.testing.assert_array_equal(dist[i, :idx.size], d.astype(np.float32))
    # the index path: every rank's stand-in index reaches every 3rd row OF ITS SHARD; the merged list is the exact answer over
    # the union of what the ranks' indexes reach (the reference's per-partition indexes + client-side merge)
    reach = np.zeros(n, dtype=np.uint8)
    from vsrbac.sharded import shard_bounds
    for r in range(world):
        lo, hi = shard_bounds(n, world, r, align=10)
        reach[lo:hi:3] = 1
    for i, qi in enumerate([3, 77, 500, 1234]):
        idx, d = oracle.filtered_topk("l2", x, x[qi], k, docs, blks, masks[i] & reach)
        assert icnt[i] == idx.size
        np.testing.assert_array_equal(iblk[i, :idx.size], blks[idx])
        np.testing.assert_array_equal(idist[i, :idx.size], d.astype(np.float32))
