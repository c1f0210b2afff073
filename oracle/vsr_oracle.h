/*
 * vsr_oracle.h — CPU restatement of the reference's RBAC-filtered k-NN path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and there only as the checker / the timed CPU baseline.  The product
 * (libvsrbac.so) never links, loads or calls it and has no CPU fallback.
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference tree rjzhb/VectorSearch-RBAC @ 2025-11-21).
 *
 * Pinning: the distance functions are pinned by pgvector's own known answers
 * (pgvector/test/expected/vector_type.out:373-530), the ordering by
 * pgvector/test/expected/hnsw_vector.out:3-90, the RBAC predicate by fixtures
 * generated in-container from the reference's tree generator
 * (tests/golden/make_rbac_fixture.py).  See tests/test_oracle_golden.py.
 */
#ifndef VSR_ORACLE_H
#define VSR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_L2 = 0, ORC_IP = 1, ORC_COSINE = 2, ORC_L1 = 3 };

/* pgvector/src/vector.c:549-563 VectorL2SquaredDistance (fp32 accumulate) */
float  orc_l2_squared(int dim, const float *a, const float *b);
/* pgvector/src/vector.c:596-606 VectorInnerProduct */
float  orc_inner_product_f32(int dim, const float *a, const float *b);
/* pgvector/src/vector.c:638-655 VectorCosineSimilarity */
double orc_cosine_similarity(int dim, const float *a, const float *b);
/* pgvector/src/vector.c:714-724 VectorL1Distance */
float  orc_l1_f32(int dim, const float *a, const float *b);

/* SQL-level values (float8 results of the fmgr functions) */
double orc_l2_distance(int dim, const float *a, const float *b);               /* vector.c:568-578  <->  */
double orc_l2_squared_distance(int dim, const float *a, const float *b);       /* vector.c:584-594  opclass proc */
double orc_inner_product(int dim, const float *a, const float *b);             /* vector.c:611-621 */
double orc_negative_inner_product(int dim, const float *a, const float *b);    /* vector.c:626-636  <#>  */
double orc_cosine_distance(int dim, const float *a, const float *b);           /* vector.c:660-685  <=>  */
double orc_l1_distance(int dim, const float *a, const float *b);               /* vector.c:729-739  <+>  */
double orc_spherical_distance(int dim, const float *a, const float *b);        /* vector.c:692-711 */
double orc_vector_norm(int dim, const float *a);                               /* vector.c:756-769 */
/* vector.c:774-808; returns 0 ok, 1 on overflow (float_overflow_error) */
int    orc_l2_normalize(int dim, const float *a, float *out);

/* operator value for `metric` (what `vector <op> q AS distance` yields) */
double orc_distance(int metric, int dim, const float *a, const float *b);

/* CheckDims message, vector.c:60-67; writes "different vector dimensions %d and %d" */
int    orc_check_dims(int dim_a, int dim_b, char *msg, int msg_len);

/*
 * RBAC predicate (controller/baseline/pg_row_security/row_level_security.py:54-65):
 *   allowed(user,row) <=> EXISTS role in UserRoles(user): (role, row.document_id) in PermissionAssignment
 * Fills mask[i] in {0,1} for every row (byte-per-row convention of
 * logical_partition_benchmark/.../test_postfilter.cpp:169-195 and
 * acorn_benchmark/src/benchmark_utils.cpp:366-391).
 */
void orc_user_row_mask(int32_t user_id,
                       const int32_t *ur_user, const int32_t *ur_role, int64_t n_ur,
                       const int32_t *pa_role, const int32_t *pa_doc, int64_t n_pa,
                       const int32_t *row_doc, int64_t n_rows,
                       uint8_t *mask);

/*
 * Exact filtered top-k = the PostgreSQL ground-truth path
 * (basic_benchmark/common_function.py:671-759: seq scan, ORDER BY distance, cut to k),
 * restricted to rows with mask[i] != 0 (mask == NULL: all rows).
 * Order: distance asc (NaN last, as PostgreSQL float8 ordering), then document_id asc,
 * then block_id asc (the tie rule this build fixes; PostgreSQL's is unspecified).
 * Outputs row indices and SQL-level distances; returns the number of rows written (<= k).
 */
int64_t orc_filtered_topk(int metric, const float *rows, int64_t n_rows, int dim,
                          const int32_t *row_doc, const int64_t *row_block,
                          const uint8_t *mask, const float *q, int64_t k,
                          int64_t *out_rows, double *out_dist);

/*
 * Client-side merge (controller/dynamic_partition/search.py:347-364,
 * controller/baseline/prefilter/prefilter_role.py:174-189): stable sort of the
 * concatenated partial results by distance, dedup on (document_id, block_id), first k.
 * Entries are given as parallel arrays; `out_idx` receives indices into them.
 */
int64_t orc_merge_dedup(const double *dist, const int32_t *doc, const int64_t *block,
                        int64_t n, int64_t k, int64_t *out_idx);

/* recall = |GT ∩ pred| / |GT| over (document_id, block_id) sets,
 * basic_benchmark/common_function.py:1154-1160 */
double orc_recall(const int32_t *gt_doc, const int64_t *gt_block, int64_t n_gt,
                  const int32_t *pr_doc, const int64_t *pr_block, int64_t n_pr);

/*
 * Timed CPU baseline helper: runs orc_filtered_topk for `nq` queries, query i restricted to
 * the row ranges ranges[range_off[i] .. range_off[i+1]) given as (start,count) pairs
 * (the role-partition tables of controller/baseline/prefilter/initialize_partitions.py:281-311
 * expressed over the shared corpus).  One thread, like the harness
 * (prefilter_role.py:88-90: max_parallel_workers_per_gather = 0).
 * Returns rows written per query in out_counts; out arrays are nq*k.
 */
void orc_search_ranges(int metric, const float *rows, int64_t n_rows, int dim,
                       const int32_t *row_doc, const int64_t *row_block,
                       const float *queries, int64_t nq, int64_t k,
                       const int64_t *range_off, const int64_t *ranges,
                       int64_t *out_rows, double *out_dist, int64_t *out_counts);

#ifdef __cplusplus
}
#endif

/* ---- index paths (vsr_index_oracle.c): pgvector's IVFFlat and HNSW restated ---------------------------------------- */
int     orc_ivf_kmeans(int metric, int dim, const float *samples, int64_t ns, int lists, uint64_t seed, float *centers);
void    orc_ivf_assign(int metric, int dim, const float *rows, int64_t n, const float *centers, int lists, int32_t *assign);
int     orc_ivf_probe(int metric, int dim, const float *centers, int lists, const float *q, int probes, int32_t *out_lists);
void   *orc_hnsw_build(int metric, const float *rows, int64_t n, int dim, int m, int ef_construction, uint64_t seed);
void    orc_hnsw_free(void *h);
void    orc_hnsw_info(const void *h, int32_t *n_elem, int32_t *entry, int32_t *entry_level, int32_t *n_upper);
void    orc_hnsw_export(const void *h, int32_t *level, int32_t *nbr0, int32_t *tid_count, int64_t *tids, int32_t *up_slot,
                        int32_t *up_nbr, int32_t max_level);
int64_t orc_hnsw_search(const void *h, const float *q, int ef, int64_t *out_rows, double *out_dist, int32_t *out_elems,
                        int64_t *n_visited);
int64_t orc_hnsw_search_pa(const void *h, const float *q, int ef, const uint8_t *allowed_rows, int64_t *out_rows,
                           double *out_dist, int32_t *out_elems, int64_t *n_visited);

#endif
