"""Multi-GPU filtered k-NN: one process per GPU, corpus sharded by contiguous row range.

The reference has no distributed runtime; its only "sharding" is table-per-role / table-per-partition with a
client-side merge of the partial results (controller/dynamic_partition/search.py:347-364,
controller/baseline/prefilter/prefilter_role.py:174-189).  Here every rank owns rows [lo, hi) of the
(document_id, block_id)-ordered corpus, searches them with the same K1/K5 kernels, and the per-rank top-k lists
(k * 24 bytes per query) are exchanged with ONE all-gather per query batch over RCCL/xGMI and merged on the GPU
(vsr_merge_topk_device).  Ordering keys carry the global row, so the merged order equals the single-GPU order.

Index scans shard the same way (the reference's analogue: one hnsw / ivfflat index per partition table, searched one after
another and merged by the client, controller/dynamic_partition/search.py:54-58,347-364): every rank builds or loads an
index over ITS rows (`GpuShardEngine.attach_index`), `ShardedSearcher.search(..., index=...)` probes every rank's index
with the same parameters (ef_search / probes) and merges the per-rank lists with the same packed all-gather and merge
launch; the ordering key of an index result is (monotone bits of the returned distance << 32) | global row.

`ShardedSearcher` holds the rank arithmetic and the collective plumbing; the per-shard compute is an `engine`
object (GpuShardEngine in production).  Tests drive the same class under gloo with a CPU stand-in engine.
"""
import ctypes

import numpy as np



def shard_bounds(n_rows, world, rank, align=1):
    """Rows [lo, hi) of rank `rank`; boundaries rounded down to `align` rows so documents stay whole."""
    def cut(r):
        if r >= world:
            return n_rows
        b = r * n_rows // world
        return b - b % align
    return cut(rank), cut(rank + 1)


def monotone_keys(rank_values, global_rows):
    """The library's ordering key: (monotone fp32 bits << 32) | global row.  NaN sorts last, -0 == +0."""
    v = np.asarray(rank_values, dtype=np.float32) + np.float32(0.0)
    u = v.view(np.uint32).astype(np.uint64)
    u = np.where(np.isnan(v), np.uint64(0x7FC00000), u)
    neg = (u & np.uint64(0x80000000)) != 0
    m = np.where(neg, (~u) & np.uint64(0xFFFFFFFF), u | np.uint64(0x80000000))
    return (m << np.uint64(32)) | np.asarray(global_rows, dtype=np.uint64)


class ShardedSearcher:
    def __init__(self, engine, world=1, rank=0, dist=None, group=None):
        self.engine, self.world, self.rank, self.dist, self.group = engine, world, rank, dist, group

    def search(self, queries, k, metric="l2", filters=None, index=None, **index_params):
        """Every rank passes the same queries; returns (block_ids, doc_ids, dist, counts) of the global top-k.
        index = "ivf" / "hnsw": every rank scans its attached index (index_params: probes=... / ef_search=...) instead of
        its rows, and the merged list is the k best of what the ranks' indexes returned."""
        if index is None:
            local = self.engine.search_local(queries, k, metric, filters)  # dict of [nq, k] tensors
        else:
            local = self.engine.search_local_index(index, queries, k, metric, filters, **index_params)
        if self.world == 1:
            return self.engine.finalize(local)
        import torch
        if "pack" in local:                     # one collective: the rank's whole result as one packed record
            t = local["pack"]
            if t.is_cuda and self.dist.get_backend(self.group) == "gloo":
                # rehearsal on one GPU box (all ranks on GPU 0): gloo moves host memory, RCCL would move device memory
                gh = torch.empty((self.world * t.shape[0],), dtype=t.dtype)
                self.dist.all_gather_into_tensor(gh, t.cpu(), group=self.group)
                g = gh.to(t.device)
            else:
                g = torch.empty((self.world * t.shape[0],), dtype=t.dtype, device=t.device)
                self.dist.all_gather_into_tensor(g, t, group=self.group)
            return self.engine.merge_packed(g, local["keys"].shape[0], k)
        gathered = {}
        for name in ("keys", "block", "doc", "dist"):
            t = local[name].contiguous()
            g = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            self.dist.all_gather_into_tensor(g, t, group=self.group)       # concatenation along dim 0
            gathered[name] = g.view((self.world,) + tuple(t.shape))
        return self.engine.merge(gathered, k)


class GpuShardEngine:
    """Per-shard compute on one MI355X through the C ABI (device pointers; torch only owns the buffers)."""

    def __init__(self, ctx, corpus, device):
        import torch
        self.torch, self.ctx, self.corpus, self.device = torch, ctx, corpus, device
        ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
        self.reruns = 0                                   # flagged queries re-run so far

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr())

    def search_local(self, queries, k, metric, filters):
        torch = self.torch
        q = queries if torch.is_tensor(queries) else torch.from_numpy(np.ascontiguousarray(queries, np.float32))
        q = q.to(self.device, torch.float32).contiguous()
        nq = q.shape[0]
        nk = nq * k
        pack = torch.empty((self.ctx.packed_result_bytes(nq, k),), dtype=torch.uint8, device=self.device)
        out = {"keys": pack[0:nk * 8].view(torch.int64).view(nq, k),
               "block": pack[nk * 8:nk * 16].view(torch.int64).view(nq, k),
               "doc": pack[nk * 16:nk * 20].view(torch.int32).view(nq, k),
               "dist": pack[nk * 20:nk * 24].view(torch.float32).view(nq, k),
               "counts": torch.empty((nq,), dtype=torch.int32, device=self.device)}
        args = (self._p(out["block"]), self._p(out["doc"]), None, self._p(out["dist"]), self._p(out["counts"]),
                self._p(out["keys"]))
        # the exact variant of the device API: queries the screening could not prove exact are re-run on the exact path
        # and patched into the record before it leaves the rank (vsr_search_device_exact)
        self.reruns += self.corpus.search_device_exact(self._p(q), nq, k, metric, filters, *args)
        out["pack"] = pack
        return out

    def finalize(self, local):
        return local["block"], local["doc"], local["dist"], local["counts"]

    # ---- index scans over this rank's rows ---------------------------------------------------------
    def attach_index(self, kind, index):
        """kind: "ivf" (vsrbac.IvfIndex) or "hnsw" (vsrbac.HnswIndex) over this rank's corpus."""
        if kind not in ("ivf", "hnsw"):
            raise ValueError("index kind must be 'ivf' or 'hnsw'")
        if not hasattr(self, "indexes"):
            self.indexes = {}
        self.indexes[kind] = index

    def device_keys(self, dist, rows, row_offset):
        """The library's ordering key on the device: (monotone fp32 bits << 32) | global row; empty slots (row < 0) sort last."""
        torch = self.torch
        v = dist.to(torch.float32) + 0.0
        u = v.view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        u = torch.where(torch.isnan(v), torch.full_like(u, 0x7FC00000), u)
        neg = (u & 0x80000000) != 0
        m = torch.where(neg, (~u) & 0xFFFFFFFF, u | 0x80000000)
        keys = (m << 32) | ((rows + int(row_offset)) & 0xFFFFFFFF)
        return torch.where(rows < 0, torch.full_like(keys, -1), keys)      # (-1 = 0xFFFF...: KEY_EMPTY)

    def search_local_index(self, kind, queries, k, metric, filters, probes=None, ef_search=None):
        torch = self.torch
        index = getattr(self, "indexes", {}).get(kind)
        if index is None:
            raise ValueError(f"no {kind} index attached to this rank (attach_index)")
        q = queries if torch.is_tensor(queries) else torch.from_numpy(np.ascontiguousarray(queries, np.float32))
        q = q.to(self.device, torch.float32).contiguous()
        nq = q.shape[0]
        nk = nq * k
        pack = torch.empty((self.ctx.packed_result_bytes(nq, k),), dtype=torch.uint8, device=self.device)
        out = {"keys": pack[0:nk * 8].view(torch.int64).view(nq, k),
               "block": pack[nk * 8:nk * 16].view(torch.int64).view(nq, k),
               "doc": pack[nk * 16:nk * 20].view(torch.int32).view(nq, k),
               "dist": pack[nk * 20:nk * 24].view(torch.float32).view(nq, k),
               "counts": torch.empty((nq,), dtype=torch.int32, device=self.device)}
        rows = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        ptrs = (self._p(out["block"]), self._p(out["doc"]), self._p(rows), self._p(out["dist"]), self._p(out["counts"]))
        if kind == "ivf":
            index.search_device(self._p(q), nq, k, int(probes if probes is not None else 1), metric, filters, *ptrs)
        else:
            keep = index.search_device(self._p(q), nq, k, int(ef_search if ef_search is not None else 40), metric, filters, *ptrs)
            self.ctx.synchronize()                          # (one asynchronous launch: the filter array must outlive it)
            del keep
        # slots past a query's count hold -1 / +inf already (the ABI's convention); their keys sort last
        valid = torch.arange(k, device=self.device)[None, :] < out["counts"][:, None]
        rows = torch.where(valid, rows, torch.full_like(rows, -1))
        out["keys"].copy_(self.device_keys(out["dist"], rows, self.corpus.row_offset))
        out["pack"] = pack
        return out

    def merge_packed(self, g_pack, nq, k):
        torch = self.torch
        world = g_pack.numel() // self.ctx.packed_result_bytes(nq, k)
        blk = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        doc = torch.empty((nq, k), dtype=torch.int32, device=self.device)
        dist = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        cnt = torch.empty((nq,), dtype=torch.int32, device=self.device)
        self.ctx.merge_topk_packed_device(self._p(g_pack), world, nq, k, self._p(blk), self._p(doc), self._p(dist),
                                          None, self._p(cnt))
        return blk, doc, dist, cnt

    def merge(self, g, k):
        torch = self.torch
        world, nq = g["keys"].shape[0], g["keys"].shape[1]
        blk = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        doc = torch.empty((nq, k), dtype=torch.int32, device=self.device)
        dist = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        cnt = torch.empty((nq,), dtype=torch.int32, device=self.device)
        self.ctx.merge_topk_device(self._p(g["keys"]), self._p(g["block"]), self._p(g["doc"]), self._p(g["dist"]),
                                   world, nq, k, self._p(blk), self._p(doc), self._p(dist), None, self._p(cnt))
        return blk, doc, dist, cnt
