/*
 * vsrbac.h — C ABI of libvsrbac: RBAC-filtered k-NN on AMD Instinct MI355X (gfx950).
 *
 * Drop-in boundary for ONE path of rjzhb/VectorSearch-RBAC: "distance(query, corpus rows) AND
 * permission(user, row) -> top-k", which the reference runs inside PostgreSQL through pgvector
 * (file:line below are relative to the reference tree):
 *
 *   pgvector/src/vector.c:549-563,568-594   VectorL2SquaredDistance, l2_distance, vector_l2_squared_distance
 *   pgvector/src/vector.c:596-636           VectorInnerProduct, inner_product, vector_negative_inner_product
 *   pgvector/src/vector.c:638-685           VectorCosineSimilarity, cosine_distance
 *   pgvector/src/vector.c:714-739           VectorL1Distance, l1_distance
 *   pgvector/src/hnswscan.c:179-316         hnswgettuple   (first call runs the whole search; later calls pop TIDs)
 *   pgvector/src/ivfscan.c:339-389          ivfflatgettuple (same contract)
 *   controller/baseline/pg_row_security/row_level_security.py:54-65   RLS policy = the per-row permission test
 *   controller/baseline/prefilter/initialize_partitions.py:281-311    role tables  = pre-filter row sets
 *   controller/dynamic_partition/search.py:54-58,347-364              comb_role -> partitions, merge + dedup
 *
 * A PostgreSQL extension shim keeps pgvector's SQL surface (type vector, operators
 * <-> <#> <=> <+>, access methods hnsw / ivfflat, GUC names) and calls these entry points; the Python
 * harness mirror (vectorsearch-rbac_amd/vsrbac) binds them with ctypes.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++ or torch types; every function returns a vsr_status (0 = ok)
 *     unless stated; never throws, never exits.  vsr_last_error() gives the message of the last failure
 *     on the calling thread (dimension mismatch keeps pgvector's text, vector.c:60-67).
 *   - host pointers are borrowed for the duration of the call; the library owns all device memory.
 *   - "_device" variants take device pointers and enqueue on the context's stream without synchronising.
 *   - a context is bound to one GPU and one host thread at a time; open one context per process after
 *     fork (PostgreSQL backends), never share across fork.  A corpus, its filters and every session searching it
 *     (vsr_search_device_on) belong to ONE host thread at a time too: the planner keeps scratch marks in the filters.
 *   - there is no CPU fallback: without a usable gfx950 device vsr_open fails with VSR_ERR_NO_DEVICE.
 */
#ifndef VSRBAC_H
#define VSRBAC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSR_ABI_VERSION 2

typedef struct vsr_ctx vsr_ctx;
typedef struct vsr_corpus vsr_corpus;
typedef struct vsr_filter vsr_filter;

typedef enum {
    VSR_OK = 0,
    VSR_ERR_INVALID = 1,        /* bad argument */
    VSR_ERR_DIM_MISMATCH = 2,   /* "different vector dimensions %d and %d" (vector.c:60-67) */
    VSR_ERR_NO_DEVICE = 3,      /* no gfx950 GPU / HIP runtime unusable */
    VSR_ERR_HIP = 4,            /* a HIP call failed; see vsr_last_error() */
    VSR_ERR_OOM = 5,
    VSR_ERR_UNSUPPORTED = 6,    /* e.g. k > VSR_MAX_K */
    VSR_ERR_NO_RBAC = 7         /* filter requested before vsr_rbac_load */
} vsr_status;

/* operator of pgvector/sql/vector.sql:174-192; the value returned is the operator's float8 result */
typedef enum {
    VSR_METRIC_L2 = 0,          /* <->  sqrt(sum (a-b)^2)            */
    VSR_METRIC_IP = 1,          /* <#>  -sum a*b                     */
    VSR_METRIC_COSINE = 2,      /* <=>  1 - clamp(cos(a,b)), NaN for a zero vector (sorted last) */
    VSR_METRIC_L1 = 3           /* <+>  sum |a-b|                    */
} vsr_metric;

/* how a permission set is applied to the scan */
typedef enum {
    VSR_FILTER_RANGES = 0,      /* pre-filter: only the permitted row ranges are read (role / partition tables) */
    VSR_FILTER_BITMAP = 1       /* post-filter: whole-corpus scan order, per-row permission bit tested in the
                                   distance loop (row-level security); fully masked tiles are skipped */
} vsr_filter_mode;

#define VSR_MAX_K 2048

/* ---- context -------------------------------------------------------------------------------- */
int         vsr_abi_version(void);
const char* vsr_last_error(void);
const char* vsr_status_string(int status);

int vsr_open(int device_ordinal, vsr_ctx** out);
int vsr_close(vsr_ctx* ctx);
/* use the caller's HIP stream (hipStream_t as void*, e.g. torch's current stream); NULL = the context's own
 * non-blocking stream.  HIP's null stream has the handle 0 as well: name it as VSR_STREAM_NULL (= hipStreamLegacy). */
#define VSR_STREAM_NULL ((void*) 1)
int vsr_set_stream(vsr_ctx* ctx, void* hip_stream);
int vsr_synchronize(vsr_ctx* ctx);
int vsr_device_info(vsr_ctx* ctx, char* name, int name_len, int* compute_units, int64_t* hbm_bytes);

/* ---- corpus: resident row-major fp32 rows + row identity -------------------------------------- */
/* rows[n][dim]; block_ids[n] / doc_ids[n] identify a row as (document_id, block_id) like the reference's
 * documentblocks table (controller/initialize_main_tables.py:54-61); either may be NULL (block_id = row
 * index, document_id = 0).  Rows are re-ordered internally by (document_id, block_id); results report the
 * caller's row index.  row_offset is added to internal rows in raw keys (multi-GPU shards; 0 otherwise). */
int vsr_corpus_load(vsr_ctx* ctx, const float* rows, int64_t n, int dim,
                    const int64_t* block_ids, const int32_t* doc_ids, int64_t row_offset,
                    vsr_corpus** out);
int     vsr_corpus_free(vsr_corpus* corpus);
int64_t vsr_corpus_rows(const vsr_corpus* corpus);
int     vsr_corpus_dim(const vsr_corpus* corpus);

/* ---- RBAC tables (UserRoles, PermissionAssignment of controller/initialize_main_tables.py:17-72) ---- */
int vsr_rbac_load(vsr_corpus* corpus,
                  const int32_t* ur_user, const int32_t* ur_role, int64_t n_user_roles,
                  const int32_t* pa_role, const int32_t* pa_doc, int64_t n_permissions);

/* ---- filters ----------------------------------------------------------------------------------- */
/* rows visible to the user: EXISTS role in UserRoles(user): (role, document) in PermissionAssignment.
 * Cached per role combination and mode; owned by the corpus (do not free). */
int vsr_filter_for_user(vsr_corpus* corpus, int32_t user_id, int mode, vsr_filter** out);
int vsr_filter_for_roles(vsr_corpus* corpus, const int32_t* role_ids, int n_roles, int mode, vsr_filter** out);
/* byte-per-row mask in the caller's row order (uint8 allowed_mask[N] of global_hnsw_index.cpp:136-183,
 * char filter map of acorn_benchmark/src/benchmark_utils.cpp:342-396).  Caller frees with vsr_filter_free. */
int vsr_filter_from_bytemask(vsr_corpus* corpus, const uint8_t* allowed, int mode, vsr_filter** out);
/* a dynamic partition = a set of documents (load_result_to_database.py:207-240); user_id >= 0 adds the
 * per-row permission test of an "impure" partition (load_result_to_database.py:590-624), -1 = pure. */
int vsr_filter_from_documents(vsr_corpus* corpus, const int32_t* doc_ids, int64_t n_docs, int32_t user_id,
                              vsr_filter** out);
int     vsr_filter_free(vsr_filter* filter);
int64_t vsr_filter_allowed_rows(const vsr_filter* filter);   /* rows that pass the filter */
int64_t vsr_filter_scanned_rows(const vsr_filter* filter);   /* rows whose distance work is priced (N for bitmap mode) */

/* ---- search -------------------------------------------------------------------------------------- */
/* nq queries of `dim` floats; filters[i] applies to query i (filters == NULL or filters[i] == NULL: no filter).
 * Outputs are nq*k, row-major, ordered by (distance asc, NaN last, document_id asc, block_id asc); entries
 * past out_counts[i] hold id -1 and +Inf.  out_rows (caller row index) and out_doc_ids may be NULL. */
int vsr_search(vsr_corpus* corpus, const float* queries, int nq, int dim, int k, int metric,
               const vsr_filter* const* filters,
               int64_t* out_block_ids, int32_t* out_doc_ids, int64_t* out_rows,
               float* out_dist, int32_t* out_counts);

/* same, queries and outputs in device memory, enqueued on the context's stream, no synchronisation.
 * out_keys (nq*k, may be NULL) receives the raw ordering keys (monotone fp32 distance << 32 | global row)
 * that vsr_merge_topk_device consumes. */
int vsr_search_device(vsr_corpus* corpus, const float* d_queries, int nq, int dim, int k, int metric,
                      const vsr_filter* const* filters,
                      int64_t* d_out_block_ids, int32_t* d_out_doc_ids, int64_t* d_out_rows,
                      float* d_out_dist, int32_t* d_out_counts, uint64_t* d_out_keys);

/* same, in the stream and workspaces of `session`: another context opened on the corpus's device (NULL = the corpus's
 * own).  Two sessions let two batches over one corpus be in flight at once, so the small selection / re-rank kernels of
 * one batch run under the scan launch of the other (a serving loop alternates sessions; bench.py does).  Flags and
 * statistics (vsr_screening_check, vsr_stats_get) are per session.  vsr_corpus_free / vsr_filter_free wait for the
 * corpus's own context only: vsr_synchronize every other session that searched the corpus before freeing it. */
int vsr_search_device_on(vsr_ctx* session, vsr_corpus* corpus, const float* d_queries, int nq, int dim, int k, int metric,
                         const vsr_filter* const* filters,
                         int64_t* d_out_block_ids, int32_t* d_out_doc_ids, int64_t* d_out_rows,
                         float* d_out_dist, int32_t* d_out_counts, uint64_t* d_out_keys);

/* same as vsr_search_device_on, but the call returns only when every query is PROVEN exact: it waits for the search,
 * and queries the screening flagged (below) are re-run on the exact path and patched into the outputs, all on the
 * session's stream.  n_rerun (may be NULL) receives how many queries that took.  Synchronises the session's stream. */
int vsr_search_device_exact(vsr_ctx* session, vsr_corpus* corpus, const float* d_queries, int nq, int dim, int k, int metric,
                            const vsr_filter* const* filters,
                            int64_t* d_out_block_ids, int32_t* d_out_doc_ids, int64_t* d_out_rows,
                            float* d_out_dist, int32_t* d_out_counts, uint64_t* d_out_keys, int32_t* n_rerun);

/* Flagged queries.  Shared passes of L2 / inner-product / cosine searches may run on the matrix cores: a bf16 / fp32
 * MFMA screening with sampled thresholds keeps 2k candidates per query and an exact re-rank reports the k best (see
 * DESIGN.md, K2w / K5r).  The re-rank FLAGS a query when the screening's rounding error or a too-tight threshold could
 * have excluded a true result (rare: none in the benchmark's 3 M queries).  A flagged query cannot be mistaken for a
 * result: its out_counts entry is NEGATIVE (-1 - rows written).  vsr_search and vsr_search_device_exact re-run
 * flagged queries themselves.  After the asynchronous vsr_search_device(_on) the caller either tests the counts on the
 * device or calls vsr_screening_check: flagged_total = flagged queries since vsr_open, flags_last_call[i] != 0 =
 * query i of the last call must be re-run with screening disabled.  vsr_screening_check synchronises. */
/* Queries of a SIFT-like workload: a corpus whose elements are all integers 0..255 (d <= 128) also keeps int8 planes, and
 * L2 searches whose QUERIES are such integers too screen on them (a quarter of the fp32 bytes per row).  Host queries
 * (vsr_search) are checked by the library.  For device-resident queries the caller states it: u8_queries != 0 promises
 * that the queries of this context's vsr_search_device calls are integer-valued in 0..255.  The promise is verified on
 * the device: a query that breaks it is FLAGGED (re-run like any flagged query) and the hint is dropped. */
int vsr_set_query_hint(vsr_ctx* ctx, int u8_queries);
int vsr_set_screening(vsr_ctx* ctx, int enable);           /* default: enabled; 0 also disables threshold seeding, so
                                                              searches of that context are exact and never flag */
int vsr_screening_check(vsr_ctx* ctx, int64_t* flagged_total, int32_t* flags_last_call, int nq);

/* merge n_parts per-shard results (layout [n_parts][nq][k], as all-gathered from vsr_search_device) into the
 * global top-k: the client-side merge of search.py:347-364 done on the GPU. */
int vsr_merge_topk_device(vsr_ctx* ctx, const uint64_t* d_keys, const int64_t* d_block_ids,
                          const int32_t* d_doc_ids, const float* d_dist, int n_parts, int nq, int k,
                          int64_t* d_out_block_ids, int32_t* d_out_doc_ids, float* d_out_dist,
                          uint64_t* d_out_keys, int32_t* d_out_counts);

/* Same merge for PACKED per-shard results: one record per shard of vsr_packed_result_bytes(nq, k) = nq*k*24 bytes,
 * laid out {keys u64[nq][k], block_ids i64[nq][k], doc_ids i32[nq][k], dist f32[nq][k]} — i.e. the four output
 * pointers of vsr_search_device aimed into one buffer — so that ONE all-gather moves a rank's whole result. */
int64_t vsr_packed_result_bytes(int nq, int k);
int vsr_merge_topk_packed_device(vsr_ctx* ctx, const void* d_packed, int n_parts, int nq, int k,
                                 int64_t* d_out_block_ids, int32_t* d_out_doc_ids, float* d_out_dist,
                                 uint64_t* d_out_keys, int32_t* d_out_counts);

/* operator value for n explicit pairs a[i] (dim floats) vs b[i] (or one shared b when b_broadcast != 0):
 * what `SELECT a <-> b` evaluates per row (vector.c:568-578 etc.), batched.  Host pointers. */
int vsr_pair_distances(vsr_ctx* ctx, int metric, const float* a, const float* b, int64_t n_pairs,
                       int dim_a, int dim_b, int b_broadcast, double* out);

/* ---- IVFFlat list probe (pgvector/src/ivfscan.c:36-176, 339-389) ------------------------------------------ */
/* An index = `lists` centres (lists x dim floats) and the list of every corpus row (row_list[n], caller row order): what
 * IVFFlat's build leaves in its list pages (ivfbuild.c, ivfkmeans.c).  The library keeps a list-ordered image of the rows
 * beside the corpus, so that a probe reads contiguous memory.  vsr_ivf_search = ivfflatgettuple's first call + the
 * executor's LIMIT and RLS filter: the `probes` nearest lists of each query (GetScanLists; equal centre distances: the
 * lower list first) are scanned exhaustively (GetScanItems) and the k nearest PERMITTED rows come back, same output
 * conventions as vsr_search.  Metric L2 / IP / COSINE as the opclasses define them (cosine: pass unit queries; the
 * centres are unit vectors; rows rank by negative inner product inside the lists and report the operator's value). */
typedef struct vsr_ivf vsr_ivf;
int vsr_ivf_load(vsr_corpus* corpus, const float* centers, int lists, const int32_t* row_list, vsr_ivf** out);
int vsr_ivf_free(vsr_ivf* ivf);       /* before vsr_corpus_free of its corpus */
/* index build, step 1 (ivfbuild.c:404-445 ComputeCenters -> ivfkmeans.c:21-93,259-498): k-means++ seeding and Elkan's
 * k-means over the sampled rows (samples[n_samples][dim], the caller samples max(lists * 50, 10000) rows like
 * ivfbuild.c:421-445); L2 for vector_l2_ops, the spherical variant for the inner-product and cosine opclasses.  The
 * random draws come from a seeded xorshift64* stream (PostgreSQL's RandomDouble is not reproducible outside a backend).
 * out_centers[lists][dim]; out_iterations (may be NULL) = Elkan iterations run.  Host pointers; synchronises. */
int vsr_ivf_kmeans(vsr_ctx* ctx, int metric, int dim, const float* samples, int64_t n_samples, int lists, uint64_t seed,
                   float* out_centers, int* out_iterations);
/* index build, step 2, the pass over every row (ivfbuild.c:141-227): out_row_list[i] = nearest of `lists` centres for caller
 * row i under the opclass distance, ties to the lower list id -- what vsr_ivf_load takes.  Host pointers; synchronises. */
int vsr_ivf_assign(vsr_corpus* corpus, const float* centers, int lists, int metric, int32_t* out_row_list);
int vsr_ivf_probe(vsr_ivf* ivf, const float* queries, int nq, int dim, int probes, int metric, int32_t* out_lists /* nq*probes */);
int vsr_ivf_search(vsr_ivf* ivf, const float* queries, int nq, int dim, int k, int probes, int metric,
                   const vsr_filter* const* filters,
                   int64_t* out_block_ids, int32_t* out_doc_ids, int64_t* out_rows, float* out_dist, int32_t* out_counts);
/* same with queries and results resident on the device (the serving form: nothing but the probed list ids, nq x probes x
 * 4 bytes, crosses PCIe, because the planner that groups queries by list is host code).  Returns when every query is
 * proven exact over its lists, like vsr_search_device_exact.  d_out_doc_ids / d_out_rows may be NULL. */
int vsr_ivf_search_device(vsr_ivf* ivf, const float* d_queries, int nq, int dim, int k, int probes, int metric,
                          const vsr_filter* const* filters,
                          int64_t* d_out_block_ids, int32_t* d_out_doc_ids, int64_t* d_out_rows, float* d_out_dist,
                          int32_t* d_out_counts);

/* ---- HNSW graph search (pgvector/src/hnswscan.c:15-45,179-316; hnswutils.c:813-976) -------------------------- */
/* A graph as pgvector's in-memory build leaves it (hnswbuild.c:357-470): n_elem elements, each with a top level, up to 10
 * heap TIDs (tids: caller row indices, -1 padded; identical vectors share an element), 2m neighbours on layer 0
 * (nbr0[n_elem][2m], -1 padded) and m per upper layer (up_nbr[n_upper][max_level][m] for the elements with level >= 1,
 * addressed through up_slot[n_elem]).  vsr_hnsw_search = hnswgettuple's first call + the executor's filter and LIMIT:
 * greedy descent (ef = 1) from `entry`, HnswSearchLayer with ef_search on layer 0, then the TIDs of the result elements
 * nearest first, the permission test per row and the first k.  Equal distances are ordered by element id.  Like the
 * reference with hnsw.iterative_scan = off, a filtered query can return fewer than k rows.
 * out_visited (may be NULL): elements entered into the visited set by the layer-0 search, per query. */
typedef struct vsr_hnsw vsr_hnsw;
int vsr_hnsw_load(vsr_corpus* corpus, int m, int32_t n_elem, int32_t entry, const int32_t* level, const int32_t* nbr0,
                  const int32_t* tid_count, const int64_t* tids, const int32_t* up_slot, const int32_t* up_nbr,
                  int32_t n_upper, int32_t max_level, vsr_hnsw** out);
int vsr_hnsw_free(vsr_hnsw* index);   /* before vsr_corpus_free of its corpus */
int vsr_hnsw_search(vsr_hnsw* index, const float* queries, int nq, int dim, int k, int ef_search, int metric,
                    const vsr_filter* const* filters,
                    int64_t* out_block_ids, int32_t* out_doc_ids, int64_t* out_rows, float* out_dist, int32_t* out_counts,
                    int64_t* out_visited);

/* same, queries (nq x dim floats, row stride dim) and outputs in device memory, enqueued on the corpus context's stream in
 * ONE launch, no synchronisation.  On graphs beyond ~1M elements the visited set is a table in LDS sized by ef_search; a
 * query that outgrows it reports count -1 (never a partial result): vsr_hnsw_search re-runs such queries itself. */
/* index build (hnswbuild.c:357-470 in-memory build; hnswutils.c:1053-1346): batched insertion on the GPU over the corpus's rows
 * with m and ef_construction as the reloptions define them, levels from a seeded xorshift64* stream.  The
 * graph is not the serial build's graph -- neither is the reference's own parallel build's -- the guarantee is recall
 * (pgvector's test/t/012_hnsw_vector_build_recall.pl thresholds; tests/test_gpu_index.py).  Identical vectors are not merged
 * into one element (hnswbuild.c:329-351): every row is its own element.  Synchronises. */
int vsr_hnsw_build(vsr_corpus* corpus, int m, int ef_construction, int metric, uint64_t seed, vsr_hnsw** out);
int vsr_hnsw_info(const vsr_hnsw* index, int32_t* n_elem, int32_t* entry, int32_t* entry_level, int32_t* max_level);
/* Predicate-aware walk (off by default: pgvector filters what the index returns, hnswscan.c + the executor's RLS qual).  On:
 * the layer-0 search of later vsr_hnsw_search* calls applies the query's filter while it walks, ACORN-1 style -- the result
 * set and the candidate set hold permitted elements only, and an expansion also takes the permitted neighbours of its
 * neighbours that are not permitted -- which is what acorn_benchmark/src/acorn_search.cpp:144-181 gets from the ACORN
 * library (its source is not part of the reference tree: parity is with the index oracle's restatement of this walk and
 * with the exact filtered scan by recall).  Queries without a filter are searched as before. */
int vsr_hnsw_set_predicate_aware(vsr_hnsw* index, int on);
int vsr_hnsw_search_device(vsr_hnsw* index, const float* d_queries, int nq, int dim, int k, int ef_search, int metric,
                           const vsr_filter* const* filters,
                           int64_t* d_out_block_ids, int32_t* d_out_doc_ids, int64_t* d_out_rows, float* d_out_dist,
                           int32_t* d_out_counts, int64_t* d_out_visited);

/* opclass support functions for n vectors at once (host pointers): vector_norm (vector.c:756-769), l2_normalize
 * (vector.c:774-808; fails with "value out of range: overflow" like float_overflow_error) and
 * vector_spherical_distance (vector.c:692-711; unit vectors assumed, as IVFFlat's spherical k-means uses it) */
int vsr_vector_norms(vsr_ctx* ctx, const float* a, int64_t n, int dim, double* out);
int vsr_l2_normalize(vsr_ctx* ctx, const float* a, int64_t n, int dim, float* out);
int vsr_spherical_distances(vsr_ctx* ctx, const float* a, const float* b, int64_t n, int dim_a, int dim_b, int b_broadcast,
                            double* out);

/* ---- measurement --------------------------------------------------------------------------------- */
typedef struct {
    /* K1 launches by kernel class: [0] = one query per pass, [1] = up to 4 queries sharing a pass */
    int64_t scan_launches[2];
    double  scan_ms[2];         /* sum of HIP-event durations of those launches          */
    int64_t scan_bytes[2];      /* algorithmic bytes: rows*dim*4 + bitmap bytes + k*12   */
    int64_t scan_rows[2];       /* rows scanned (per shared pass)                        */
    int64_t select_launches;    /* K5 */
    double  select_ms;
    int64_t queries;
    double  search_ms;          /* profiling level 1: device time of whole searches, staging kernel to last output kernel */
    /* ABI 2: what the launches HAD to do, whatever the pass structure (the honest roofline inputs) */
    int64_t scan_pairs[2];      /* (row, query) pairs = sum over passes of rows * queries: flops = 2 * dim * pairs       */
    int64_t unique_rows[2];     /* rows read at least once per launch, summed over launches: distinct filter parts'
                                   rows, capped at the corpus size (exact when the parts are disjoint, e.g. classes)  */
    /* host side of the search entry points (always on): time spent inside them, and the part of it spent WAITING for
       the previous batch's staging block (back-pressure from the GPU, not work) */
    double  host_ms;
    double  host_wait_ms;
} vsr_stats;

int vsr_profiling(vsr_ctx* ctx, int enable);      /* HIP events on the launch stream: 1 = around every launch class (scan, sample, K5), 2 = around the main scan launch only, 0 = off */
int vsr_stats_get(vsr_ctx* ctx, vsr_stats* out);  /* synchronises, accumulates pending events */
int vsr_stats_reset(vsr_ctx* ctx);
/* name of the kernel instantiation the main scan launch of the session's last search resolved to ("" before any) */
int vsr_last_scan_kernel(vsr_ctx* ctx, char* name, int name_len);

/* launch-shape knobs (measurement only): blocks per launch budget, min rows per workgroup, queries per pass */
int vsr_tune(vsr_ctx* ctx, int block_budget, int min_rows_per_block, int max_queries_per_pass);

#ifdef __cplusplus
}
#endif
#endif /* VSRBAC_H */
