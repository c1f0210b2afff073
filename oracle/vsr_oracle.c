/*
 * vsr_oracle.c — CPU restatement of the reference's RBAC-filtered k-NN path.
 * TEST INFRASTRUCTURE ONLY (see vsr_oracle.h): never linked into or called by the product.
 *
 * Arithmetic follows pgvector 0.8.1 as vendored by the reference:
 *   fp32 accumulation in the inner loops, the float8 (double) post-processing of the
 *   fmgr wrappers, NaN/Infinity behaviour as in pgvector/test/expected/vector_type.out.
 */
#define _GNU_SOURCE
#include "vsr_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- pgvector/src/vector.c:549-563 ---- */
float orc_l2_squared(int dim, const float *ax, const float *bx)
{
    float distance = 0.0f;
    for (int i = 0; i < dim; i++) {
        float diff = ax[i] - bx[i];
        distance += diff * diff;
    }
    return distance;
}

/* ---- pgvector/src/vector.c:596-606 ---- */
float orc_inner_product_f32(int dim, const float *ax, const float *bx)
{
    float distance = 0.0f;
    for (int i = 0; i < dim; i++)
        distance += ax[i] * bx[i];
    return distance;
}

/* ---- pgvector/src/vector.c:638-655 ---- */
double orc_cosine_similarity(int dim, const float *ax, const float *bx)
{
    float similarity = 0.0f, norma = 0.0f, normb = 0.0f;
    for (int i = 0; i < dim; i++) {
        similarity += ax[i] * bx[i];
        norma += ax[i] * ax[i];
        normb += bx[i] * bx[i];
    }
    /* "Use sqrt(a * b) over sqrt(a) * sqrt(b)" */
    return (double) similarity / sqrt((double) norma * (double) normb);
}

/* ---- pgvector/src/vector.c:714-724 ---- */
float orc_l1_f32(int dim, const float *ax, const float *bx)
{
    float distance = 0.0f;
    for (int i = 0; i < dim; i++)
        distance += fabsf(ax[i] - bx[i]);
    return distance;
}

/* vector.c:568-578 */
double orc_l2_distance(int dim, const float *a, const float *b)
{
    return sqrt((double) orc_l2_squared(dim, a, b));
}

/* vector.c:584-594 */
double orc_l2_squared_distance(int dim, const float *a, const float *b)
{
    return (double) orc_l2_squared(dim, a, b);
}

/* vector.c:611-621 */
double orc_inner_product(int dim, const float *a, const float *b)
{
    return (double) orc_inner_product_f32(dim, a, b);
}

/* vector.c:626-636 */
double orc_negative_inner_product(int dim, const float *a, const float *b)
{
    return (double) -orc_inner_product_f32(dim, a, b);
}

/* vector.c:660-685 */
double orc_cosine_distance(int dim, const float *a, const float *b)
{
    double similarity = orc_cosine_similarity(dim, a, b);
    /* Keep in range (NaN falls through both tests) */
    if (similarity > 1)
        similarity = 1.0;
    else if (similarity < -1)
        similarity = -1.0;
    return 1.0 - similarity;
}

/* vector.c:729-739 */
double orc_l1_distance(int dim, const float *a, const float *b)
{
    return (double) orc_l1_f32(dim, a, b);
}

/* vector.c:692-711 */
double orc_spherical_distance(int dim, const float *a, const float *b)
{
    double distance = (double) orc_inner_product_f32(dim, a, b);
    if (distance > 1)
        distance = 1;
    else if (distance < -1)
        distance = -1;
    return acos(distance) / M_PI;
}

/* vector.c:756-769 */
double orc_vector_norm(int dim, const float *ax)
{
    double norm = 0.0;
    for (int i = 0; i < dim; i++)
        norm += (double) ax[i] * (double) ax[i];
    return sqrt(norm);
}

/* vector.c:774-808 */
int orc_l2_normalize(int dim, const float *ax, float *rx)
{
    double norm = 0;
    for (int i = 0; i < dim; i++)
        norm += (double) ax[i] * (double) ax[i];
    norm = sqrt(norm);
    for (int i = 0; i < dim; i++)
        rx[i] = 0.0f;
    if (norm > 0) {
        for (int i = 0; i < dim; i++)
            rx[i] = ax[i] / norm;
        for (int i = 0; i < dim; i++)
            if (isinf(rx[i]))
                return 1;
    }
    return 0;
}

double orc_distance(int metric, int dim, const float *a, const float *b)
{
    switch (metric) {
    case ORC_L2:     return orc_l2_distance(dim, a, b);
    case ORC_IP:     return orc_negative_inner_product(dim, a, b);
    case ORC_COSINE: return orc_cosine_distance(dim, a, b);
    case ORC_L1:     return orc_l1_distance(dim, a, b);
    default:         return NAN;
    }
}

/* vector.c:60-67 */
int orc_check_dims(int dim_a, int dim_b, char *msg, int msg_len)
{
    if (dim_a != dim_b) {
        if (msg)
            snprintf(msg, (size_t) msg_len, "different vector dimensions %d and %d", dim_a, dim_b);
        return 1;
    }
    if (msg && msg_len > 0)
        msg[0] = 0;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * RBAC predicate — row_level_security.py:54-65
 * ---------------------------------------------------------------------------------------- */
static int cmp_i32(const void *a, const void *b)
{
    int32_t x = *(const int32_t *) a, y = *(const int32_t *) b;
    return (x > y) - (x < y);
}

void orc_user_row_mask(int32_t user_id,
                       const int32_t *ur_user, const int32_t *ur_role, int64_t n_ur,
                       const int32_t *pa_role, const int32_t *pa_doc, int64_t n_pa,
                       const int32_t *row_doc, int64_t n_rows,
                       uint8_t *mask)
{
    /* roles of the user (UserRoles WHERE user_id = current_user) */
    int32_t *roles = (int32_t *) malloc(sizeof(int32_t) * (size_t) (n_ur > 0 ? n_ur : 1));
    int64_t n_roles = 0;
    for (int64_t i = 0; i < n_ur; i++)
        if (ur_user[i] == user_id)
            roles[n_roles++] = ur_role[i];
    qsort(roles, (size_t) n_roles, sizeof(int32_t), cmp_i32);

    /* documents visible through any of those roles (PermissionAssignment JOIN UserRoles) */
    int32_t *docs = (int32_t *) malloc(sizeof(int32_t) * (size_t) (n_pa > 0 ? n_pa : 1));
    int64_t n_docs = 0;
    for (int64_t i = 0; i < n_pa; i++)
        if (bsearch(&pa_role[i], roles, (size_t) n_roles, sizeof(int32_t), cmp_i32))
            docs[n_docs++] = pa_doc[i];
    qsort(docs, (size_t) n_docs, sizeof(int32_t), cmp_i32);

    for (int64_t r = 0; r < n_rows; r++)
        mask[r] = bsearch(&row_doc[r], docs, (size_t) n_docs, sizeof(int32_t), cmp_i32) ? 1 : 0;

    free(roles);
    free(docs);
}

/* ------------------------------------------------------------------------------------------
 * Exact filtered top-k (ground-truth path, common_function.py:671-759)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    double  dist;
    int32_t doc;
    int64_t block;
    int64_t row;
} orc_cand;

/* PostgreSQL float8 ordering: NaN sorts after every non-NaN value (float8_cmp_internal). */
static inline int cmp_f8(double a, double b)
{
    int an = isnan(a), bn = isnan(b);
    if (an || bn)
        return an - bn;          /* both NaN -> 0, a NaN -> 1 (a after b) */
    return (a > b) - (a < b);
}

/* total order: (distance, document_id, block_id, row) */
static inline int cmp_cand(const orc_cand *a, const orc_cand *b)
{
    int c = cmp_f8(a->dist, b->dist);
    if (c) return c;
    if (a->doc != b->doc) return (a->doc > b->doc) - (a->doc < b->doc);
    if (a->block != b->block) return (a->block > b->block) - (a->block < b->block);
    return (a->row > b->row) - (a->row < b->row);
}

static int cmp_cand_qsort(const void *a, const void *b)
{
    return cmp_cand((const orc_cand *) a, (const orc_cand *) b);
}

/* bounded max-heap on cmp_cand: heap[0] is the worst of the kept k (PostgreSQL's top-N heapsort
 * does the same for ORDER BY ... LIMIT k) */
static void heap_sift_down(orc_cand *h, int64_t n, int64_t i)
{
    for (;;) {
        int64_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && cmp_cand(&h[l], &h[m]) > 0) m = l;
        if (r < n && cmp_cand(&h[r], &h[m]) > 0) m = r;
        if (m == i) return;
        orc_cand t = h[i]; h[i] = h[m]; h[m] = t;
        i = m;
    }
}

static void heap_sift_up(orc_cand *h, int64_t i)
{
    while (i > 0) {
        int64_t p = (i - 1) / 2;
        if (cmp_cand(&h[i], &h[p]) <= 0) return;
        orc_cand t = h[i]; h[i] = h[p]; h[p] = t;
        i = p;
    }
}

typedef struct {
    orc_cand *h;
    int64_t   n, k;
} orc_heap;

static inline void heap_offer(orc_heap *hp, const orc_cand *c)
{
    if (hp->n < hp->k) {
        hp->h[hp->n] = *c;
        heap_sift_up(hp->h, hp->n);
        hp->n++;
    } else if (hp->k > 0 && cmp_cand(c, &hp->h[0]) < 0) {
        hp->h[0] = *c;
        heap_sift_down(hp->h, hp->n, 0);
    }
}

/* ranking value: the SQL-level operator result (monotone in the fp32 sums) */
static inline double rank_value(int metric, int dim, const float *row, const float *q)
{
    return orc_distance(metric, dim, row, q);
}

static void scan_range(orc_heap *hp, int metric, const float *rows, int dim,
                       const int32_t *row_doc, const int64_t *row_block,
                       const uint8_t *mask, const float *q, int64_t start, int64_t count)
{
    for (int64_t r = start; r < start + count; r++) {
        if (mask && !mask[r])
            continue;
        orc_cand c;
        c.dist = rank_value(metric, dim, rows + (size_t) r * (size_t) dim, q);
        c.doc = row_doc ? row_doc[r] : 0;
        c.block = row_block ? row_block[r] : r;
        c.row = r;
        heap_offer(hp, &c);
    }
}

static int64_t heap_drain(orc_heap *hp, int64_t *out_rows, double *out_dist)
{
    qsort(hp->h, (size_t) hp->n, sizeof(orc_cand), cmp_cand_qsort);
    for (int64_t i = 0; i < hp->n; i++) {
        out_rows[i] = hp->h[i].row;
        out_dist[i] = hp->h[i].dist;
    }
    return hp->n;
}

int64_t orc_filtered_topk(int metric, const float *rows, int64_t n_rows, int dim,
                          const int32_t *row_doc, const int64_t *row_block,
                          const uint8_t *mask, const float *q, int64_t k,
                          int64_t *out_rows, double *out_dist)
{
    if (k <= 0 || n_rows <= 0)
        return 0;
    orc_heap hp;
    hp.h = (orc_cand *) malloc(sizeof(orc_cand) * (size_t) k);
    hp.n = 0;
    hp.k = k;
    scan_range(&hp, metric, rows, dim, row_doc, row_block, mask, q, 0, n_rows);
    int64_t n = heap_drain(&hp, out_rows, out_dist);
    free(hp.h);
    return n;
}

void orc_search_ranges(int metric, const float *rows, int64_t n_rows, int dim,
                       const int32_t *row_doc, const int64_t *row_block,
                       const float *queries, int64_t nq, int64_t k,
                       const int64_t *range_off, const int64_t *ranges,
                       int64_t *out_rows, double *out_dist, int64_t *out_counts)
{
    (void) n_rows;
    orc_heap hp;
    hp.h = (orc_cand *) malloc(sizeof(orc_cand) * (size_t) (k > 0 ? k : 1));
    hp.k = k;
    for (int64_t qi = 0; qi < nq; qi++) {
        hp.n = 0;
        for (int64_t j = range_off[qi]; j < range_off[qi + 1]; j++)
            scan_range(&hp, metric, rows, dim, row_doc, row_block, NULL,
                       queries + (size_t) qi * (size_t) dim, ranges[2 * j], ranges[2 * j + 1]);
        out_counts[qi] = heap_drain(&hp, out_rows + qi * k, out_dist + qi * k);
    }
    free(hp.h);
}

/* ------------------------------------------------------------------------------------------
 * merge + dedup — search.py:347-364, prefilter_role.py:174-189
 * Python's list.sort is stable: equal distances keep concatenation order.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    double  dist;
    int64_t idx;
} orc_mitem;

static int cmp_mitem(const void *a, const void *b)
{
    const orc_mitem *x = (const orc_mitem *) a, *y = (const orc_mitem *) b;
    int c = cmp_f8(x->dist, y->dist);
    if (c) return c;
    return (x->idx > y->idx) - (x->idx < y->idx);   /* stability */
}

int64_t orc_merge_dedup(const double *dist, const int32_t *doc, const int64_t *block,
                        int64_t n, int64_t k, int64_t *out_idx)
{
    if (n <= 0 || k <= 0)
        return 0;
    orc_mitem *it = (orc_mitem *) malloc(sizeof(orc_mitem) * (size_t) n);
    for (int64_t i = 0; i < n; i++) {
        it[i].dist = dist[i];
        it[i].idx = i;
    }
    qsort(it, (size_t) n, sizeof(orc_mitem), cmp_mitem);
    int64_t m = 0;
    for (int64_t i = 0; i < n && m < k; i++) {
        int64_t a = it[i].idx;
        int seen = 0;
        for (int64_t j = 0; j < m; j++) {          /* `seen` set of (document_id, block_id) */
            int64_t b = out_idx[j];
            if (doc[a] == doc[b] && block[a] == block[b]) { seen = 1; break; }
        }
        if (!seen)
            out_idx[m++] = a;
    }
    free(it);
    return m;
}

/* ------------------------------------------------------------------------------------------
 * recall — common_function.py:1154-1160 (sets of (document_id, block_id))
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t doc;
    int64_t block;
} orc_key;

static int cmp_key(const void *a, const void *b)
{
    const orc_key *x = (const orc_key *) a, *y = (const orc_key *) b;
    if (x->doc != y->doc) return (x->doc > y->doc) - (x->doc < y->doc);
    return (x->block > y->block) - (x->block < y->block);
}

static int64_t uniq_keys(orc_key *k, int64_t n)
{
    if (n == 0) return 0;
    qsort(k, (size_t) n, sizeof(orc_key), cmp_key);
    int64_t m = 1;
    for (int64_t i = 1; i < n; i++)
        if (cmp_key(&k[i], &k[m - 1]) != 0)
            k[m++] = k[i];
    return m;
}

double orc_recall(const int32_t *gt_doc, const int64_t *gt_block, int64_t n_gt,
                  const int32_t *pr_doc, const int64_t *pr_block, int64_t n_pr)
{
    orc_key *g = (orc_key *) malloc(sizeof(orc_key) * (size_t) (n_gt > 0 ? n_gt : 1));
    orc_key *p = (orc_key *) malloc(sizeof(orc_key) * (size_t) (n_pr > 0 ? n_pr : 1));
    for (int64_t i = 0; i < n_gt; i++) { g[i].doc = gt_doc[i]; g[i].block = gt_block[i]; }
    for (int64_t i = 0; i < n_pr; i++) { p[i].doc = pr_doc[i]; p[i].block = pr_block[i]; }
    int64_t ng = uniq_keys(g, n_gt), np = uniq_keys(p, n_pr);
    int64_t inter = 0;
    for (int64_t i = 0; i < ng; i++)
        if (bsearch(&g[i], p, (size_t) np, sizeof(orc_key), cmp_key))
            inter++;
    double r = ng > 0 ? (double) inter / (double) ng : NAN;   /* Python raises ZeroDivisionError */
    free(g);
    free(p);
    return r;
}
