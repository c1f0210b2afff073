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
