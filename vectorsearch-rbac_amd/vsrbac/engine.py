"""Object wrappers over the C ABI: Context (one GPU), Corpus (resident rows + identity + RBAC), Filter."""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import _ffi
from ._ffi import VsrError, check, load_library

L2, IP, COSINE, L1 = 0, 1, 2, 3
METRICS = {"l2": L2, "<->": L2, "ip": IP, "<#>": IP, "cosine": COSINE, "<=>": COSINE, "l1": L1, "<+>": L1}
RANGES, BITMAP = 0, 1

SearchResult = namedtuple("SearchResult", "block_ids doc_ids rows dist counts")


def _metric(m):
    return METRICS[m] if isinstance(m, str) else int(m)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One MI355X.  Fails loudly (VsrError) when there is no gfx950 device."""

    def __init__(self, device=0):
        self._lib = load_library()
        h = C.c_void_p()
        check(self._lib.vsr_open(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vsr_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, hbm = C.c_int(), C.c_int64()
        check(self._lib.vsr_device_info(self._h, name, 256, C.byref(cus), C.byref(hbm)))
        return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": hbm.value}

    def set_stream(self, stream_handle):
        """A hipStream_t handle (torch: `stream.cuda_stream`).  None: the context's own stream.  0 is what torch reports
        for the device's null stream: it is passed as VSR_STREAM_NULL, so that work the caller orders on that stream
        (events, copies, collectives) really is ordered against the searches."""
        if stream_handle is None:
            handle = 0
        else:
            handle = int(stream_handle) or 1
        check(self._lib.vsr_set_stream(self._h, C.c_void_p(handle)))

    def set_query_hint(self, u8_queries=True):
        """Promise that the device-resident queries of this context are integers 0..255 (SIFT): L2 searches over a
        corpus of such integers then screen on its int8 planes.  Verified on the device; a violation flags the query."""
        check(self._lib.vsr_set_query_hint(self._h, 1 if u8_queries else 0))

    def synchronize(self):
        check(self._lib.vsr_synchronize(self._h))

    def profiling(self, enable=True):
        """True / 1: HIP events around every launch class; 2: around the main scan launch only; False: off."""
        check(self._lib.vsr_profiling(self._h, int(enable)))

    def stats(self):
        st = _ffi.Stats()
        check(self._lib.vsr_stats_get(self._h, C.byref(st)))
        out = {}
        for f, _ in st._fields_:
            v = getattr(st, f)
            out[f] = list(v) if hasattr(v, "__len__") else v
        return out

    def stats_reset(self):
        check(self._lib.vsr_stats_reset(self._h))

    def last_scan_kernel(self):
        """Kernel instantiation the main scan launch of this session's last search resolved to."""
        buf = C.create_string_buffer(200)
        check(self._lib.vsr_last_scan_kernel(self._h, buf, 200))
        return buf.value.decode()

    def set_screening(self, enable=True):
        """Allow the MFMA screening + exact re-rank path (K2/K5r) for shared passes."""
        check(self._lib.vsr_set_screening(self._h, int(bool(enable))))

    def screening_check(self, nq=0):
        """(flagged queries since open, flags of the last vsr_search_device call).  Synchronises."""
        total = C.c_int64()
        flags = np.zeros(max(nq, 1), dtype=np.int32)
        check(self._lib.vsr_screening_check(self._h, C.byref(total), _ptr(flags) if nq else None, int(nq)))
        return total.value, flags[:nq]

    def tune(self, block_budget=-1, min_rows_per_block=0, max_queries_per_pass=0):
        check(self._lib.vsr_tune(self._h, block_budget, min_rows_per_block, max_queries_per_pass))

    def load_corpus(self, rows, block_ids=None, doc_ids=None, row_offset=0):
        return Corpus(self, rows, block_ids, doc_ids, row_offset)

    def pair_distances(self, metric, a, b):
        """Operator value for pairs (a[i], b[i]); b may be a single vector (broadcast).
        Raises VsrError('different vector dimensions %d and %d') like pgvector's CheckDims."""
        a = np.ascontiguousarray(np.atleast_2d(np.asarray(a, dtype=np.float32)))
        b = np.ascontiguousarray(np.asarray(b, dtype=np.float32))
        bcast = b.ndim == 1
        b2 = np.atleast_2d(b)
        out = np.empty(a.shape[0], dtype=np.float64)
        check(self._lib.vsr_pair_distances(self._h, _metric(metric), _ptr(a), _ptr(b2), a.shape[0],
                                           a.shape[1], b2.shape[1], int(bcast), _ptr(out)))
        return out

    def vector_norms(self, a):
        """vector_norm of every row (vector.c:756-769)."""
        a = np.ascontiguousarray(np.atleast_2d(np.asarray(a, dtype=np.float32)))
        out = np.empty(a.shape[0], dtype=np.float64)
        check(self._lib.vsr_vector_norms(self._h, _ptr(a), a.shape[0], a.shape[1], _ptr(out)))
        return out

    def l2_normalize(self, a):
        """l2_normalize of every row (vector.c:774-808); raises VsrError('value out of range: overflow')."""
        a = np.ascontiguousarray(np.atleast_2d(np.asarray(a, dtype=np.float32)))
        out = np.empty_like(a)
        check(self._lib.vsr_l2_normalize(self._h, _ptr(a), a.shape[0], a.shape[1], _ptr(out)))
        return out

    def spherical_distances(self, a, b):
        """vector_spherical_distance(a[i], b[i]) (vector.c:692-711); b may be a single vector."""
        a = np.ascontiguousarray(np.atleast_2d(np.asarray(a, dtype=np.float32)))
        b = np.ascontiguousarray(np.asarray(b, dtype=np.float32))
        bcast = b.ndim == 1
        b2 = np.atleast_2d(b)
        out = np.empty(a.shape[0], dtype=np.float64)
        check(self._lib.vsr_spherical_distances(self._h, _ptr(a), _ptr(b2), a.shape[0], a.shape[1], b2.shape[1], int(bcast),
                                                _ptr(out)))
        return out

    def ivf_kmeans(self, samples, lists, metric="l2", seed=1):
        """IVFFlat build, step 1 (ivfkmeans.c): k-means++ + Elkan's k-means over the sampled rows on the GPU.
        Returns (centers[lists, dim], iterations)."""
        s = np.ascontiguousarray(np.atleast_2d(np.asarray(samples, dtype=np.float32)))
        ns, dim = (0, s.shape[1]) if s.size == 0 else s.shape
        out = np.zeros((int(lists), dim), dtype=np.float32)
        it = C.c_int(0)
        check(self._lib.vsr_ivf_kmeans(self._h, _metric(metric), dim, _ptr(s), ns, int(lists), C.c_uint64(int(seed)), _ptr(out),
                                       C.byref(it)))
        return out, it.value

    def merge_topk_device(self, keys, block_ids, doc_ids, dist, n_parts, nq, k, out_block, out_doc, out_dist,
                          out_keys, out_counts):
        """Device pointers (ints).  Layout [n_parts][nq][k]."""
        check(self._lib.vsr_merge_topk_device(self._h, keys, block_ids, doc_ids, dist, n_parts, nq, k,
                                              out_block, out_doc, out_dist, out_keys, out_counts))


    def packed_result_bytes(self, nq, k):
        return self._lib.vsr_packed_result_bytes(int(nq), int(k))

    def merge_topk_packed_device(self, packed, n_parts, nq, k, out_block, out_doc, out_dist, out_keys, out_counts):
        """`packed`: device pointer to n_parts records of packed_result_bytes(nq, k) bytes (one all-gather)."""
        check(self._lib.vsr_merge_topk_packed_device(self._h, packed, n_parts, nq, k, out_block, out_doc, out_dist,
                                                     out_keys, out_counts))


class Filter:
    def __init__(self, corpus, handle, owned):
        self.corpus = corpus
        self._h = handle
        self._owned = owned

    @property
    def allowed_rows(self):
        return self.corpus._lib.vsr_filter_allowed_rows(self._h)

    @property
    def scanned_rows(self):
        return self.corpus._lib.vsr_filter_scanned_rows(self._h)

    def free(self):
        if self._owned and self._h:
            self.corpus._lib.vsr_filter_free(self._h)
        self._h = None

    def __del__(self):
        try:
            if self.corpus._h:
                self.free()
        except Exception:
            pass


class Corpus:
    """Rows resident in HBM, identified as (document_id, block_id) like the reference's documentblocks table."""

    def __init__(self, ctx, rows, block_ids=None, doc_ids=None, row_offset=0):
        self.ctx = ctx
        self._lib = ctx._lib
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim != 2:
            raise ValueError("rows must be [n, dim]")
        n, dim = rows.shape
        blk = None if block_ids is None else np.ascontiguousarray(block_ids, dtype=np.int64)
        doc = None if doc_ids is None else np.ascontiguousarray(doc_ids, dtype=np.int32)
        if (blk is not None and blk.size != n) or (doc is not None and doc.size != n):
            raise ValueError("block_ids / doc_ids must have one entry per row")
        h = C.c_void_p()
        check(self._lib.vsr_corpus_load(ctx._h, _ptr(rows), n, dim, _ptr(blk), _ptr(doc), int(row_offset),
                                        C.byref(h)))
        self._h = h
        self.n, self.dim = n, dim
        self.row_offset = int(row_offset)
        self._user_filters = {}

    def free(self):
        if getattr(self, "_h", None):
            self._lib.vsr_corpus_free(self._h)
            self._h = None

    def __del__(self):
        try:
            if self.ctx._h:
                self.free()
        except Exception:
            pass

    # ---- RBAC -------------------------------------------------------------------------------
    def load_rbac(self, user_roles, permissions):
        """user_roles: (user_id, role_id) pairs; permissions: (role_id, document_id) pairs."""
        ur = np.ascontiguousarray(np.asarray(user_roles, dtype=np.int32).reshape(-1, 2))
        pa = np.ascontiguousarray(np.asarray(permissions, dtype=np.int32).reshape(-1, 2))
        u, r = np.ascontiguousarray(ur[:, 0]), np.ascontiguousarray(ur[:, 1])
        pr, pd = np.ascontiguousarray(pa[:, 0]), np.ascontiguousarray(pa[:, 1])
        check(self._lib.vsr_rbac_load(self._h, _ptr(u), _ptr(r), len(ur), _ptr(pr), _ptr(pd), len(pa)))
        self._user_filters = {}

    def filter_for_user(self, user_id, mode=RANGES):
        key = (int(user_id), int(mode))
        f = self._user_filters.get(key)
        if f is None:
            h = C.c_void_p()
            check(self._lib.vsr_filter_for_user(self._h, int(user_id), int(mode), C.byref(h)))
            f = self._user_filters[key] = Filter(self, h, owned=False)
        return f

    def filter_for_roles(self, role_ids, mode=RANGES):
        r = np.ascontiguousarray(role_ids, dtype=np.int32)
        h = C.c_void_p()
        check(self._lib.vsr_filter_for_roles(self._h, _ptr(r), r.size, int(mode), C.byref(h)))
        return Filter(self, h, owned=False)

    def filter_from_bytemask(self, allowed, mode=BITMAP):
        m = np.ascontiguousarray(allowed, dtype=np.uint8)
        if m.size != self.n:
            raise ValueError("mask must have one byte per row")
        h = C.c_void_p()
        check(self._lib.vsr_filter_from_bytemask(self._h, _ptr(m), int(mode), C.byref(h)))
        return Filter(self, h, owned=True)

    def filter_from_documents(self, doc_ids, user_id=-1):
        d = np.ascontiguousarray(doc_ids, dtype=np.int32)
        h = C.c_void_p()
        check(self._lib.vsr_filter_from_documents(self._h, _ptr(d), d.size, int(user_id), C.byref(h)))
        return Filter(self, h, owned=True)

    # ---- indexes -----------------------------------------------------------------------------
    def load_ivf(self, centers, row_list):
        """IVFFlat index over this corpus: centres [lists, dim] and the list of every row (caller row order)."""
        return IvfIndex(self, centers, row_list)

    def build_ivf(self, rows, lists=100, metric="l2", seed=1, sample=None):
        """CREATE INDEX ... USING ivfflat on the GPU (ivfbuild.c:998-1019): sample max(lists * 50, 10000) of `rows` (this
        corpus's rows in the caller's order, host array), k-means on the sample (vsr_ivf_kmeans), every row into its
        nearest list (vsr_ivf_assign), list-ordered image (vsr_ivf_load).  Returns (IvfIndex, centers, row_list)."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        n = rows.shape[0]
        want = max(int(lists) * 50, 10000) if sample is None else int(sample)
        rng = np.random.default_rng(seed)
        pick = np.sort(rng.choice(n, size=min(n, want), replace=False)) if n else np.zeros(0, dtype=np.int64)
        centers, _ = self.ctx.ivf_kmeans(rows[pick], lists, metric, seed)
        row_list = self.ivf_assign(centers, metric)
        return IvfIndex(self, centers, row_list), centers, row_list

    def build_hnsw(self, m=16, ef_construction=64, metric="l2", seed=1):
        """CREATE INDEX ... USING hnsw on the GPU (vsr_hnsw_build): batched insertion, recall parity with the serial build."""
        h = C.c_void_p()
        check(self._lib.vsr_hnsw_build(self._h, int(m), int(ef_construction), _metric(metric), int(seed), C.byref(h)))
        return HnswIndex(self, h)

    def load_hnsw(self, graph):
        """HNSW graph over this corpus; `graph`: dict with m, entry, level, nbr0, tid_count, tids, up_slot, up_nbr,
        max_level (include/vsrbac.h, vsr_hnsw_load: what a dump of pgvector's in-memory build holds)."""
        return HnswIndex(self, graph)

    # ---- search ------------------------------------------------------------------------------
    def pack_filters(self, filters):
        """One filter per query as a reusable C array (build it once when the same batch shape repeats)."""
        arr = (C.c_void_p * len(filters))(*[(f._h if f is not None else None) for f in filters])
        arr._keep = list(filters)                    # the handles must outlive the array
        return arr

    def _filter_array(self, filters, nq):
        if filters is None:
            return None, None
        if isinstance(filters, C.Array):
            if len(filters) != nq:
                raise ValueError("one filter per query")
            return filters, filters
        if isinstance(filters, Filter):
            filters = [filters] * nq
        if len(filters) != nq:
            raise ValueError("one filter per query")
        arr = (C.c_void_p * nq)(*[(f._h if f is not None else None) for f in filters])
        return arr, filters

    def search(self, queries, k, metric="l2", filters=None):
        """Host buffers in and out.  Returns SearchResult of [nq, k] arrays (+ counts[nq])."""
        q = np.ascontiguousarray(np.atleast_2d(np.asarray(queries, dtype=np.float32)))
        nq, dim = q.shape
        farr, keep = self._filter_array(filters, nq)
        kk = max(int(k), 1)
        blk = np.full((nq, kk), -1, dtype=np.int64)
        doc = np.full((nq, kk), -1, dtype=np.int32)
        row = np.full((nq, kk), -1, dtype=np.int64)
        dist = np.full((nq, kk), np.inf, dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.int32)
        check(self._lib.vsr_search(self._h, _ptr(q), nq, dim, int(k), _metric(metric), farr, _ptr(blk), _ptr(doc),
                                   _ptr(row), _ptr(dist), _ptr(cnt)))
        del keep
        return SearchResult(blk, doc, row, dist, cnt)

    def search_device(self, d_queries, nq, k, metric, filters, d_block, d_doc, d_rows, d_dist, d_counts, d_keys=None,
                      dim=None, session=None):
        """Device pointers (ints); enqueues on the context's stream, does not synchronise.  `session`: another Context
        of the same GPU whose stream and workspaces the search uses (two batches in flight over one corpus)."""
        farr, keep = self._filter_array(filters, nq)
        check(self._lib.vsr_search_device_on(session._h if session is not None else None, self._h, d_queries, nq,
                                             self.dim if dim is None else dim, int(k), _metric(metric), farr, d_block,
                                             d_doc, d_rows, d_dist, d_counts, d_keys))
        return keep

    def ivf_assign(self, centers, metric="l2"):
        """The pass of the ivfflat build that touches every row (ivfbuild.c:404-445): the nearest centre of each row, in
        the caller's row order -- the `row_list` IvfIndex / vsr_ivf_load take."""
        c = np.ascontiguousarray(centers, dtype=np.float32)
        if c.ndim != 2 or c.shape[1] != self.dim:
            raise VsrError(_ffi.ERR_INVALID, f"centers must be (lists, {self.dim})")
        out = np.zeros(self.n, dtype=np.int32)
        check(self._lib.vsr_ivf_assign(self._h, _ptr(c), int(c.shape[0]), _metric(metric), _ptr(out)))
        return out

    def search_device_exact(self, d_queries, nq, k, metric, filters, d_block, d_doc, d_rows, d_dist, d_counts, d_keys=None,
                            dim=None, session=None):
        """search_device + wait + exact re-run of whatever the screening flagged (vsr_search_device_exact): returns the
        number of queries that were re-run; every row of the outputs is proven exact when it returns."""
        farr, keep = self._filter_array(filters, nq)
        n = C.c_int32(0)
        check(self._lib.vsr_search_device_exact(session._h if session is not None else None, self._h, d_queries, nq,
                                                self.dim if dim is None else dim, int(k), _metric(metric), farr, d_block,
                                                d_doc, d_rows, d_dist, d_counts, d_keys, C.byref(n)))
        del keep
        return int(n.value)


class IvfIndex:
    """pgvector's ivfflat scan (ivfscan.c) on the GPU: probe the nearest lists, scan them, keep the permitted top-k."""

    def __init__(self, corpus, centers, row_list):
        self.corpus, self._lib = corpus, corpus._lib
        c = np.ascontiguousarray(centers, dtype=np.float32)
        rl = np.ascontiguousarray(row_list, dtype=np.int32)
        if c.ndim != 2 or c.shape[1] != corpus.dim or rl.size != corpus.n:
            raise ValueError("centers must be [lists, dim] and row_list must have one entry per row")
        h = C.c_void_p()
        check(self._lib.vsr_ivf_load(corpus._h, _ptr(c), c.shape[0], _ptr(rl), C.byref(h)))
        self._h, self.lists = h, c.shape[0]

    def free(self):
        if getattr(self, "_h", None):
            self._lib.vsr_ivf_free(self._h)
            self._h = None

    def __del__(self):
        try:
            if self.corpus._h:
                self.free()
        except Exception:
            pass

    def search_device(self, d_queries, nq, k, probes, metric, filters, d_block, d_doc, d_rows, d_dist, d_counts):
        """Device pointers (ints); returns when every query is proven exact over its lists (vsr_ivf_search_device)."""
        farr, keep = self.corpus._filter_array(filters, nq)
        check(self._lib.vsr_ivf_search_device(self._h, d_queries, nq, self.corpus.dim, int(k), int(probes), _metric(metric),
                                              farr, d_block, d_doc, d_rows, d_dist, d_counts))
        del keep

    def probe(self, queries, probes, metric="l2"):
        q = np.ascontiguousarray(np.atleast_2d(np.asarray(queries, dtype=np.float32)))
        p = min(int(probes), self.lists)
        out = np.zeros((q.shape[0], p), dtype=np.int32)
        check(self._lib.vsr_ivf_probe(self._h, _ptr(q), q.shape[0], q.shape[1], int(probes), _metric(metric), _ptr(out)))
        return out

    def search(self, queries, k, probes, metric="l2", filters=None):
        q = np.ascontiguousarray(np.atleast_2d(np.asarray(queries, dtype=np.float32)))
        nq, dim = q.shape
        farr, keep = self.corpus._filter_array(filters, nq)
        kk = max(int(k), 1)
        blk = np.full((nq, kk), -1, dtype=np.int64)
        doc = np.full((nq, kk), -1, dtype=np.int32)
        row = np.full((nq, kk), -1, dtype=np.int64)
        dist = np.full((nq, kk), np.inf, dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.int32)
        check(self._lib.vsr_ivf_search(self._h, _ptr(q), nq, dim, int(k), int(probes), _metric(metric), farr, _ptr(blk),
                                       _ptr(doc), _ptr(row), _ptr(dist), _ptr(cnt)))
        del keep
        return SearchResult(blk, doc, row, dist, cnt)


class HnswIndex:
    """pgvector's hnsw scan (hnswscan.c) on the GPU: greedy descent, ef_search beam on layer 0, TIDs, filter, LIMIT."""

    def __init__(self, corpus, g):
        self.corpus, self._lib = corpus, corpus._lib
        if isinstance(g, C.c_void_p):                # a handle vsr_hnsw_build returned (Corpus.build_hnsw)
            self._h = g
            return
        a = lambda name, dt: np.ascontiguousarray(g[name], dtype=dt)
        level, nbr0, tc, tids = a("level", np.int32), a("nbr0", np.int32), a("tid_count", np.int32), a("tids", np.int64)
        up_slot, up_nbr = a("up_slot", np.int32), a("up_nbr", np.int32)
        n_upper = int((up_slot >= 0).sum())
        h = C.c_void_p()
        check(self._lib.vsr_hnsw_load(corpus._h, int(g["m"]), level.size, int(g["entry"]), _ptr(level), _ptr(nbr0), _ptr(tc),
                                      _ptr(tids), _ptr(up_slot), _ptr(up_nbr), n_upper, int(g["max_level"]), C.byref(h)))
        self._h = h

    def free(self):
        if getattr(self, "_h", None):
            self._lib.vsr_hnsw_free(self._h)
            self._h = None

    def __del__(self):
        try:
            if self.corpus._h:
                self.free()
        except Exception:
            pass

    def set_predicate_aware(self, on=True):
        """The layer-0 walk applies the query's filter itself (ACORN-1 style two-hop expansion) instead of leaving the
        filtering of the results to the caller's side (vsr_hnsw_set_predicate_aware)."""
        check(self._lib.vsr_hnsw_set_predicate_aware(self._h, 1 if on else 0))

    def info(self):
        """(elements, entry point, its level, highest level) of the graph."""
        v = [C.c_int32() for _ in range(4)]
        check(self._lib.vsr_hnsw_info(self._h, *[C.byref(x) for x in v]))
        return tuple(int(x.value) for x in v)

    def search_device(self, d_queries, nq, k, ef_search, metric, filters, d_block, d_doc, d_rows, d_dist, d_counts,
                      d_visited=None):
        """Device pointers (ints); ONE launch on the corpus context's stream, no synchronisation (vsr_hnsw_search_device)."""
        farr, keep = self.corpus._filter_array(filters, nq)
        check(self._lib.vsr_hnsw_search_device(self._h, d_queries, nq, self.corpus.dim, int(k), int(ef_search), _metric(metric),
                                               farr, d_block, d_doc, d_rows, d_dist, d_counts, d_visited))
        return keep

    def search(self, queries, k, ef_search=40, metric="l2", filters=None):
        """SearchResult plus, as a second value, the number of elements each query visited on layer 0."""
        q = np.ascontiguousarray(np.atleast_2d(np.asarray(queries, dtype=np.float32)))
        nq, dim = q.shape
        farr, keep = self.corpus._filter_array(filters, nq)
        kk = max(int(k), 1)
        blk = np.full((nq, kk), -1, dtype=np.int64)
        doc = np.full((nq, kk), -1, dtype=np.int32)
        row = np.full((nq, kk), -1, dtype=np.int64)
        dist = np.full((nq, kk), np.inf, dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.int32)
        vis = np.zeros(nq, dtype=np.int64)
        check(self._lib.vsr_hnsw_search(self._h, _ptr(q), nq, dim, int(k), int(ef_search), _metric(metric), farr, _ptr(blk),
                                        _ptr(doc), _ptr(row), _ptr(dist), _ptr(cnt), _ptr(vis)))
        del keep
        return SearchResult(blk, doc, row, dist, cnt), vis
