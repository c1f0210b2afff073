/*
 * vsr_index_oracle.c — CPU restatement of pgvector's two index paths.  TEST INFRASTRUCTURE ONLY (see vsr_oracle.h):
 * it is the checker of the GPU list probe (K3) and graph search (K4) and the "pgvector CPU HNSW / IVFFlat" baseline that
 * bench.py times on the host; nothing under vectorsearch-rbac_amd/ may link or call it.
 *
 * Restated from (reference tree, file:line):
 *   IVFFlat   pgvector/src/ivfkmeans.c:21-93 (k-means++ InitCenters), :192-246 (ComputeNewCenters), :259-498 (ElkanKmeans),
 *             pgvector/src/ivfbuild.c:141-227 (every row goes to its nearest centre, opclass distance proc 1),
 *             pgvector/src/ivfscan.c:36-107 (GetScanLists: the `probes` nearest lists), :112-176 (GetScanItems)
 *   HNSW      pgvector/src/hnswutils.c:239-262 (level draw), :617-657 (comparators), :813-976 (HnswSearchLayer, Algorithm 2),
 *             :1024-1154 (CheckElementCloser, SelectNeighbors = Algorithm 4 keeping pruned connections),
 *             :1172-1220 (HnswUpdateConnection), :1270-1346 (HnswFindElementNeighbors, Algorithm 1),
 *             pgvector/src/hnswbuild.c:309-419 (duplicate vectors share an element with up to 10 heap TIDs, neighbours are
 *             linked back, entry point), pgvector/src/hnswscan.c:15-45 (GetScanItems: greedy descent, then ef_search)
 *             pgvector/src/hnsw.h:51,83-89 (HNSW_HEAPTIDS, layer m, ml, max level)
 *
 * What is NOT the reference's: the random source (pgvector draws from PostgreSQL's PRNG; here a seeded xorshift64*), and
 * the order among candidates of EQUAL distance (pgvector's pairing heaps leave it to insertion history; here candidates are
 * totally ordered by (distance, element id), which is one of the orders the reference can produce).  Pinned by
 * tests/test_index_oracle.py: the index-order expectations of pgvector/test/expected/hnsw_vector.out:3-90 and
 * ivfflat_vector.out, and the recall thresholds of pgvector/test/t/012_hnsw_vector_build_recall.pl:94 (>= 0.99 at
 * ef_search 40 on 10k x 3-d) and t/005_ivfflat_query_recall.pl:31-41 (a row is its own nearest neighbour).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "vsr_oracle.h"

/* ------------------------------------------------------------------------------------------ */
/* random source (seeded; stands in for RandomDouble / RandomInt of pgvector/src/vector.h)     */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint64_t s; } orc_rng;

static uint64_t rng_next(orc_rng *r)
{
    uint64_t x = r->s;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    r->s = x;
    return x * 0x2545F4914F6CDD1DULL;
}
static double rng_double(orc_rng *r) { return (double) (rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static void rng_seed(orc_rng *r, uint64_t seed) { r->s = seed * 0x9E3779B97F4A7C15ULL + 0x1234567ULL; if (!r->s) r->s = 1; rng_next(r); }

/* opclass distance proc 1: what the index ranks by (vector.sql:292-333) */
static double index_distance(int metric, int dim, const float *a, const float *b)
{
    switch (metric) {
    case ORC_L2: return orc_l2_squared_distance(dim, a, b);           /* vector_l2_squared_distance */
    case ORC_L1: return orc_l1_distance(dim, a, b);
    default:     return orc_negative_inner_product(dim, a, b);        /* ip and cosine (rows normalised by the caller) */
    }
}
/* k-means distance proc 3 (ivfflat.h IVFFLAT_KMEANS_DISTANCE_PROC): L2 for l2_ops, angular for ip / cosine */
static double kmeans_distance(int metric, int dim, const float *a, const float *b)
{
    return metric == ORC_L2 ? orc_l2_distance(dim, a, b) : orc_spherical_distance(dim, a, b);
}

/* ========================================================================================== */
/* IVFFlat                                                                                     */
/* ========================================================================================== */
static void norm_center(int dim, float *c)
{
    float tmp[dim > 0 ? dim : 1];
    if (orc_l2_normalize(dim, c, tmp) == 0) memcpy(c, tmp, sizeof(float) * (size_t) dim);
}

/* ivfkmeans.c:259-498.  samples [ns][dim]; centers [lists][dim] out.  metric ORC_L2: plain; else spherical (centres normalised) */
int orc_ivf_kmeans(int metric, int dim, const float *samples, int64_t ns, int lists, uint64_t seed, float *centers)
{
    orc_rng rng;
    rng_seed(&rng, seed);
    if (lists < 1 || dim < 1) return 1;
    if (ns == 0) {                                   /* RandomCenters, :124-147 */
        for (int64_t i = 0; i < (int64_t) lists * dim; i++) centers[i] = (float) rng_double(&rng);
        if (metric != ORC_L2) for (int j = 0; j < lists; j++) norm_center(dim, centers + (size_t) j * dim);
        return 0;
    }
    const int nc = lists;
    float *lower = (float *) malloc(sizeof(float) * (size_t) ns * nc);
    float *upper = (float *) malloc(sizeof(float) * (size_t) ns);
    float *weight = (float *) malloc(sizeof(float) * (size_t) ns);
    float *s = (float *) malloc(sizeof(float) * (size_t) nc);
    float *half = (float *) malloc(sizeof(float) * (size_t) nc * nc);
    float *newc = (float *) malloc(sizeof(float) * (size_t) nc * dim);
    float *newcdist = (float *) malloc(sizeof(float) * (size_t) nc);
    int *counts = (int *) malloc(sizeof(int) * (size_t) nc);
    int *closest = (int *) malloc(sizeof(int) * (size_t) ns);
#define S(j) (samples + (size_t) (j) * dim)
#define C(k) (centers + (size_t) (k) * dim)
    /* InitCenters (k-means++), :21-93 */
    memcpy(C(0), S((int64_t) (rng_next(&rng) % (uint64_t) ns)), sizeof(float) * (size_t) dim);
    for (int64_t j = 0; j < ns; j++) weight[j] = FLT_MAX;
    for (int i = 0; i < nc; i++) {
        double sum = 0.0;
        for (int64_t j = 0; j < ns; j++) {
            double d = kmeans_distance(metric, dim, S(j), C(i));
            lower[j * nc + i] = (float) d;
            d *= d;
            if (d < weight[j]) weight[j] = (float) d;
            sum += weight[j];
        }
        if (i + 1 == nc) break;
        double choice = sum * rng_double(&rng);
        int64_t j;
        for (j = 0; j < ns - 1; j++) {
            choice -= weight[j];
            if (choice <= 0) break;
        }
        memcpy(C(i + 1), S(j), sizeof(float) * (size_t) dim);
    }
    for (int64_t j = 0; j < ns; j++) {               /* :325-345 */
        float mind = FLT_MAX;
        int cc = 0;
        for (int k = 0; k < nc; k++)
            if (lower[j * nc + k] < mind) { mind = lower[j * nc + k]; cc = k; }
        upper[j] = mind;
        closest[j] = cc;
    }
    for (int iteration = 0; iteration < 500; iteration++) {          /* :348-497 */
        int changes = 0;
        for (int j = 0; j < nc; j++)
            for (int k = j + 1; k < nc; k++) {
                float d = (float) (0.5 * kmeans_distance(metric, dim, C(j), C(k)));
                half[j * nc + k] = d;
                half[k * nc + j] = d;
            }
        for (int j = 0; j < nc; j++) {
            float mind = FLT_MAX;
            for (int k = 0; k < nc; k++)
                if (j != k && half[j * nc + k] < mind) mind = half[j * nc + k];
            s[j] = mind;
        }
        const int rjreset = iteration != 0;
        for (int64_t j = 0; j < ns; j++) {
            if (upper[j] <= s[closest[j]]) continue;
            int rj = rjreset;
            for (int k = 0; k < nc; k++) {
                float dxcx;
                if (k == closest[j]) continue;
                if (upper[j] <= lower[j * nc + k]) continue;
                if (upper[j] <= half[closest[j] * nc + k]) continue;
                if (rj) {
                    dxcx = (float) kmeans_distance(metric, dim, S(j), C(closest[j]));
                    lower[j * nc + closest[j]] = dxcx;
                    upper[j] = dxcx;
                    rj = 0;
                } else
                    dxcx = upper[j];
                if (dxcx > lower[j * nc + k] || dxcx > half[closest[j] * nc + k]) {
                    float dxc = (float) kmeans_distance(metric, dim, S(j), C(k));
                    lower[j * nc + k] = dxc;
                    if (dxc < dxcx) {
                        closest[j] = k;
                        upper[j] = dxc;
                        changes++;
                    }
                }
            }
        }
        /* ComputeNewCenters, :192-246: sums in float4, empty centre -> random values */
        memset(newc, 0, sizeof(float) * (size_t) nc * dim);
        memset(counts, 0, sizeof(int) * (size_t) nc);
        for (int64_t j = 0; j < ns; j++) {
            float *x = newc + (size_t) closest[j] * dim;
            for (int k = 0; k < dim; k++) x[k] += S(j)[k];
            counts[closest[j]]++;
        }
        for (int j = 0; j < nc; j++) {
            float *x = newc + (size_t) j * dim;
            if (counts[j] > 0) {
                for (int k = 0; k < dim; k++) {
                    if (isinf(x[k])) x[k] = x[k] > 0 ? FLT_MAX : -FLT_MAX;
                    x[k] /= (float) counts[j];
                }
            } else
                for (int k = 0; k < dim; k++) x[k] = (float) rng_double(&rng);
            if (metric != ORC_L2) norm_center(dim, x);
        }
        for (int j = 0; j < nc; j++) newcdist[j] = (float) kmeans_distance(metric, dim, C(j), newc + (size_t) j * dim);
        for (int64_t j = 0; j < ns; j++)
            for (int k = 0; k < nc; k++) {
                float d = lower[j * nc + k] - newcdist[k];
                lower[j * nc + k] = d < 0 ? 0 : d;
            }
        for (int64_t j = 0; j < ns; j++) upper[j] += newcdist[closest[j]];
        memcpy(centers, newc, sizeof(float) * (size_t) nc * dim);
        if (changes == 0 && iteration != 0) break;
    }
#undef S
#undef C
    free(lower); free(upper); free(weight); free(s); free(half); free(newc); free(newcdist); free(counts); free(closest);
    return 0;
}

/* ivfbuild.c:141-227 (InsertTuple): the list of a row is its nearest centre under the index distance, first one on ties */
void orc_ivf_assign(int metric, int dim, const float *rows, int64_t n, const float *centers, int lists, int32_t *assign)
{
    for (int64_t i = 0; i < n; i++) {
        double mind = DBL_MAX;
        int best = 0;
        for (int k = 0; k < lists; k++) {
            double d = index_distance(metric, dim, rows + (size_t) i * dim, centers + (size_t) k * dim);
            if (d < mind) { mind = d; best = k; }
        }
        assign[i] = best;
    }
}

/* ivfscan.c:36-107 (GetScanLists): the `probes` nearest lists, nearest first; equal distances: the lower list id first */
int orc_ivf_probe(int metric, int dim, const float *centers, int lists, const float *q, int probes, int32_t *out_lists)
{
    if (probes > lists) probes = lists;
    double *d = (double *) malloc(sizeof(double) * (size_t) lists);
    char *taken = (char *) calloc((size_t) lists, 1);
    for (int k = 0; k < lists; k++) d[k] = index_distance(metric, dim, centers + (size_t) k * dim, q);
    for (int p = 0; p < probes; p++) {
        int best = -1;
        for (int k = 0; k < lists; k++)
            if (!taken[k] && (best < 0 || d[k] < d[best])) best = k;
        taken[best] = 1;
        out_lists[p] = best;
    }
    free(d);
    free(taken);
    return probes;
}

/* ========================================================================================== */
/* HNSW                                                                                        */
/* ========================================================================================== */
#define HNSW_HEAPTIDS 10                              /* hnsw.h:51 */

typedef struct { int32_t elem; float dist; } hcand;   /* HnswCandidate: distance is float4 (hnsw.h:159-164) */

typedef struct {
    int      metric, dim, m, efc;
    int64_t  n_rows;
    const float *rows;                                /* borrowed */
    int32_t  n_elem;
    int32_t *elem_row;                                /* representative row (first heap TID) of an element */
    int32_t *elem_level;
    int32_t *tid_count;                               /* heap TIDs of an element, <= 10 */
    int64_t (*tids)[HNSW_HEAPTIDS];
    hcand  **nbr;                                     /* nbr[e][lc]: neighbour array of level lc, capacity layer_m(lc) */
    int32_t **nbr_len;                                /* nbr_len[e][lc] */
    int32_t  entry;                                   /* element id or -1 */
    int      max_level_cap;
} orc_hnsw;

static int layer_m(int m, int lc) { return lc == 0 ? 2 * m : m; }                 /* hnsw.h:83 */
static const float *evec(const orc_hnsw *g, int32_t e) { return g->rows + (size_t) g->elem_row[e] * g->dim; }
static hcand *nbrs(const orc_hnsw *g, int32_t e, int lc, int32_t **len)
{
    int off = 0;
    for (int l = 0; l < lc; l++) off += layer_m(g->m, l);
    *len = &g->nbr_len[e][lc];
    return g->nbr[e] + off;
}

/* (distance, element id) total order: see the header comment */
static int cand_less(double da, int32_t ea, double db, int32_t eb) { return da < db || (da == db && ea < eb); }

typedef struct { double dist; int32_t elem; } scand;   /* HnswSearchCandidate: distance is double (hnsw.h:172-178) */
typedef struct { scand *a; int n, cap; int maxheap; } sheap;
static int sheap_before(const sheap *h, const scand *x, const scand *y)
{
    return h->maxheap ? cand_less(y->dist, y->elem, x->dist, x->elem) : cand_less(x->dist, x->elem, y->dist, y->elem);
}
static void sheap_push(sheap *h, scand v)
{
    if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 64; h->a = (scand *) realloc(h->a, sizeof(scand) * (size_t) h->cap); }
    int i = h->n++;
    h->a[i] = v;
    while (i > 0) {
        int p = (i - 1) / 2;
        if (!sheap_before(h, &h->a[i], &h->a[p])) break;
        scand t = h->a[i]; h->a[i] = h->a[p]; h->a[p] = t;
        i = p;
    }
}
static scand sheap_pop(sheap *h)
{
    scand top = h->a[0];
    h->a[0] = h->a[--h->n];
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = l + 1, b = i;
        if (l < h->n && sheap_before(h, &h->a[l], &h->a[b])) b = l;
        if (r < h->n && sheap_before(h, &h->a[r], &h->a[b])) b = r;
        if (b == i) break;
        scand t = h->a[i]; h->a[i] = h->a[b]; h->a[b] = t;
        i = b;
    }
    return top;
}

/* HnswSearchLayer, hnswutils.c:813-976.  ep / w: arrays of scand; w comes back ordered furthest first (as popping W does).
 * visited: caller-provided byte map over elements (cleared here), n_visited counts AddToVisited insertions. */
static int search_layer(const orc_hnsw *g, const float *q, const scand *ep, int n_ep, int ef, int lc, scand *w,
                        uint8_t *visited, int64_t *n_visited)
{
    sheap C = {0}, W = {0};
    W.maxheap = 1;
    int wlen = 0;
    memset(visited, 0, (size_t) g->n_elem);
    for (int i = 0; i < n_ep; i++) {
        visited[ep[i].elem] = 1;
        if (n_visited) (*n_visited)++;
        sheap_push(&C, ep[i]);
        sheap_push(&W, ep[i]);
        wlen++;
    }
    while (C.n > 0) {
        scand c = sheap_pop(&C);
        scand f = W.a[0];
        if (c.dist > f.dist) break;
        int32_t *len;
        hcand *nb = nbrs(g, c.elem, lc, &len);
        for (int i = 0; i < *len; i++) {
            const int32_t e = nb[i].elem;
            if (visited[e]) continue;                  /* HnswLoadUnvisitedFromMemory: AddToVisited, skip if found */
            visited[e] = 1;
            if (n_visited) (*n_visited)++;
            const int always = wlen < ef;
            f = W.a[0];
            const double ed = index_distance(g->metric, g->dim, q, evec(g, e));
            if (!(ed < f.dist || always)) continue;
            if (g->elem_level[e] < lc) continue;
            scand sc = {ed, e};
            sheap_push(&C, sc);
            sheap_push(&W, sc);
            wlen++;
            if (wlen > ef) (void) sheap_pop(&W);       /* "no need to decrement wlen" */
        }
    }
    int nw = 0;
    while (W.n > 0) w[nw++] = sheap_pop(&W);           /* furthest first */
    free(C.a);
    free(W.a);
    return nw;
}

/* The predicate-aware layer-0 walk of the GPU's K4 (vsr_hnsw.h, predicate_aware; ACORN-1 style -- the ACORN library the
 * reference calls at acorn_benchmark/src/acorn_search.cpp:144-181 is not part of the reference tree, so this restates OUR walk;
 * parity unpinned against ACORN itself).  Like search_layer on layer 0, except that an expansion of c takes, in list order,
 * c's unvisited neighbours that are allowed, then for every unvisited neighbour that is NOT allowed (marked visited too) its
 * allowed unvisited neighbours, 64 list positions at a time, until 256 candidates are collected (what a 64-position step marks
 * beyond the cap stays marked and is dropped).  allowed_rows: one byte per heap row; an element is allowed when one of its
 * heap rows is.  n_visited counts every element marked. */
#define ORC_PA_CAP 256
static int elem_allowed(const orc_hnsw *g, int32_t e, const uint8_t *allowed_rows)
{
    for (int t = 0; t < g->tid_count[e]; t++)
        if (allowed_rows[g->tids[e][t]]) return 1;
    return 0;
}

static int search_layer_pa(const orc_hnsw *g, const float *q, const scand *ep, int n_ep, int ef, scand *w, uint8_t *visited,
                           int64_t *n_visited, const uint8_t *allowed_rows)
{
    sheap C = {0}, W = {0};
    W.maxheap = 1;
    int wlen = 0;
    const int lm = layer_m(g->m, 0);
    int32_t cand[ORC_PA_CAP];
    int32_t *hop = (int32_t *) malloc(sizeof(int32_t) * (size_t) (lm + 1));
    memset(visited, 0, (size_t) g->n_elem);
    for (int i = 0; i < n_ep; i++) {
        visited[ep[i].elem] = 1;
        if (n_visited) (*n_visited)++;
        sheap_push(&C, ep[i]);
        sheap_push(&W, ep[i]);
        wlen++;
    }
    while (C.n > 0) {
        scand c = sheap_pop(&C);
        scand f = W.a[0];
        if (c.dist > f.dist) break;
        int32_t *len;
        hcand *nb = nbrs(g, c.elem, 0, &len);
        int cnt = 0, n_hop = 0;
        for (int i = 0; i < *len; i++) {
            const int32_t e = nb[i].elem;
            if (visited[e]) continue;
            visited[e] = 1;
            if (n_visited) (*n_visited)++;
            if (elem_allowed(g, e, allowed_rows)) cand[cnt++] = e;
            else hop[n_hop++] = e;
        }
        for (int h = 0; h < n_hop && cnt < ORC_PA_CAP; h++) {
            int32_t *len2;
            hcand *nb2 = nbrs(g, hop[h], 0, &len2);
            for (int j0 = 0; j0 < lm && cnt < ORC_PA_CAP; j0 += 64)
                for (int j = j0; j < j0 + 64 && j < *len2; j++) {
                    const int32_t e = nb2[j].elem;
                    if (!elem_allowed(g, e, allowed_rows) || visited[e]) continue;
                    visited[e] = 1;
                    if (n_visited) (*n_visited)++;
                    if (cnt < ORC_PA_CAP) cand[cnt++] = e;
                }
        }
        for (int i = 0; i < cnt; i++) {
            const int32_t e = cand[i];
            const int always = wlen < ef;
            f = W.a[0];
            const double ed = index_distance(g->metric, g->dim, q, evec(g, e));
            if (!(ed < f.dist || always)) continue;
            scand sc = {ed, e};
            sheap_push(&C, sc);
            sheap_push(&W, sc);
            wlen++;
            if (wlen > ef) (void) sheap_pop(&W);
        }
    }
    int nw = 0;
    while (W.n > 0) w[nw++] = sheap_pop(&W);
    free(C.a);
    free(W.a);
    free(hop);
    return nw;
}

/* CheckElementCloser, :1024-1046 */
static int check_closer(const orc_hnsw *g, const hcand *e, hcand *const *r, int nr)
{
    for (int i = 0; i < nr; i++) {
        float d = (float) index_distance(g->metric, g->dim, evec(g, e->elem), evec(g, r[i]->elem));
        if (d <= e->dist) return 0;
    }
    return 1;
}

/* SelectNeighbors, :1053-1154, without the closer cache (same result, recomputed).  c: candidates ordered furthest ->
 * nearest (the order HnswSearchLayer's w and list_sort produce); out: the selected neighbours in selection order;
 * *pruned: the candidate dropped (only meaningful for n > lm). */
static int select_neighbors(const orc_hnsw *g, hcand **c, int n, int lm, hcand **out, hcand **pruned)
{
    if (n <= lm) {
        for (int i = 0; i < n; i++) out[i] = c[i];
        if (pruned) *pruned = NULL;
        return n;
    }
    hcand **wd = (hcand **) malloc(sizeof(hcand *) * (size_t) n);
    int wdlen = 0, wdoff = 0, nr = 0, wl = n;
    while (wl > 0 && nr < lm) {
        hcand *e = c[--wl];                            /* llast(w): the nearest remaining */
        if (check_closer(g, e, out, nr)) out[nr++] = e;
        else wd[wdlen++] = e;
    }
    while (wdoff < wdlen && nr < lm) out[nr++] = wd[wdoff++];      /* keep pruned connections */
    if (pruned) *pruned = wdoff < wdlen ? wd[wdoff] : c[0];        /* linitial(w): w still holds c[0 .. wl) */
    free(wd);
    return nr;
}

static int cmp_cand_desc(const void *pa, const void *pb)           /* CompareCandidateDistances, :980-1000: nearest last */
{
    const hcand *a = *(hcand *const *) pa, *b = *(hcand *const *) pb;
    if (a->dist < b->dist) return 1;
    if (a->dist > b->dist) return -1;
    if (a->elem < b->elem) return 1;
    if (a->elem > b->elem) return -1;
    return 0;
}

/* HnswUpdateConnection, :1172-1220 */
static void update_connection(orc_hnsw *g, int32_t owner, int lc, int32_t new_elem, float distance)
{
    int32_t *len;
    hcand *nb = nbrs(g, owner, lc, &len);
    const int lm = layer_m(g->m, lc);
    hcand nh = {new_elem, distance};
    if (*len < lm) {
        nb[(*len)++] = nh;
        return;
    }
    hcand **c = (hcand **) malloc(sizeof(hcand *) * (size_t) (*len + 1));
    hcand **r = (hcand **) malloc(sizeof(hcand *) * (size_t) (*len + 1));
    for (int i = 0; i < *len; i++) c[i] = &nb[i];
    c[*len] = &nh;
    qsort(c, (size_t) (*len + 1), sizeof(hcand *), cmp_cand_desc);
    hcand *pruned = NULL;
    (void) select_neighbors(g, c, *len + 1, lm, r, &pruned);
    if (pruned && pruned != &nh)
        for (int i = 0; i < *len; i++)
            if (nb[i].elem == pruned->elem) { nb[i] = nh; break; }
    free(c);
    free(r);
}

static void hnsw_insert(orc_hnsw *g, int64_t row, orc_rng *rng, uint8_t *visited, scand *wbuf, hcand *lw, hcand **lwp, hcand **sel)
{
    const double ml = 1.0 / log((double) g->m);                                   /* hnsw.h:86 */
    int level = (int) (-log(rng_double(rng)) * ml);                               /* hnswutils.c:243 */
    if (level > g->max_level_cap) level = g->max_level_cap;
    const int32_t e = g->n_elem;                       /* tentatively the next element */
    g->elem_row[e] = (int32_t) row;
    g->elem_level[e] = level;
    g->tid_count[e] = 1;
    g->tids[e][0] = row;
    int total = 0;
    for (int l = 0; l <= level; l++) total += layer_m(g->m, l);
    g->nbr[e] = (hcand *) calloc((size_t) total, sizeof(hcand));
    g->nbr_len[e] = (int32_t *) calloc((size_t) level + 1, sizeof(int32_t));
    const float *q = g->rows + (size_t) row * g->dim;

    if (g->entry >= 0) {                               /* HnswFindElementNeighbors, :1270-1346 */
        const int32_t entry = g->entry;
        const int entry_level = g->elem_level[entry];
        scand ep[1] = {{index_distance(g->metric, g->dim, q, evec(g, entry)), entry}};
        scand *epv = ep;
        int n_ep = 1, lvl = level;
        /* the candidate being inserted must not see itself: it is not linked yet, so no special case is needed */
        g->n_elem = e;                                 /* search over the existing elements only */
        for (int lc = entry_level; lc >= level + 1; lc--) {
            int nw = search_layer(g, q, epv, n_ep, 1, lc, wbuf, visited, NULL);
            epv = wbuf;
            n_ep = nw;
        }
        if (lvl > entry_level) lvl = entry_level;
        scand *cur = (scand *) malloc(sizeof(scand) * (size_t) (g->efc + 2));
        memcpy(cur, epv, sizeof(scand) * (size_t) n_ep);
        for (int lc = lvl; lc >= 0; lc--) {
            const int lm = layer_m(g->m, lc);
            int nw = search_layer(g, q, cur, n_ep, g->efc, lc, wbuf, visited, NULL);
            for (int i = 0; i < nw; i++) { lw[i].elem = wbuf[i].elem; lw[i].dist = (float) wbuf[i].dist; lwp[i] = &lw[i]; }
            int ns = select_neighbors(g, lwp, nw, lm, sel, NULL);
            int32_t *len;
            hcand *nb = nbrs(g, e, lc, &len);
            for (int i = 0; i < ns; i++) nb[(*len)++] = *sel[i];                   /* AddConnections */
            memcpy(cur, wbuf, sizeof(scand) * (size_t) nw);
            n_ep = nw;
        }
        free(cur);
        g->n_elem = e;
    }
    /* UpdateGraphInMemory, hnswbuild.c:400-419: duplicate check first (:329-351) */
    {
        int32_t *len;
        hcand *nb = nbrs(g, e, 0, &len);
        for (int i = 0; i < *len; i++) {
            const int32_t d = nb[i].elem;
            if (memcmp(evec(g, d), q, sizeof(float) * (size_t) g->dim) != 0) break;   /* ordered by distance: exit early */
            if (g->tid_count[d] < HNSW_HEAPTIDS) {
                g->tids[d][g->tid_count[d]++] = row;
                free(g->nbr[e]);
                free(g->nbr_len[e]);
                g->nbr[e] = NULL;
                g->nbr_len[e] = NULL;
                return;                                 /* no new element */
            }
        }
    }
    g->n_elem = e + 1;
    for (int lc = level; lc >= 0; lc--) {              /* UpdateNeighborsInMemory, :368-395 */
        int32_t *len;
        hcand *nb = nbrs(g, e, lc, &len);
        const int cnt = *len;
        hcand copy[2 * 100 + 2];
        memcpy(copy, nb, sizeof(hcand) * (size_t) cnt);
        for (int i = 0; i < cnt; i++) update_connection(g, copy[i].elem, lc, e, copy[i].dist);
    }
    if (g->entry < 0 || level > g->elem_level[g->entry]) g->entry = e;
}

void *orc_hnsw_build(int metric, const float *rows, int64_t n, int dim, int m, int ef_construction, uint64_t seed)
{
    orc_hnsw *g = (orc_hnsw *) calloc(1, sizeof(orc_hnsw));
    g->metric = metric; g->dim = dim; g->m = m; g->efc = ef_construction; g->n_rows = n; g->rows = rows; g->entry = -1;
    /* HnswGetMaxLevel, hnsw.h:89 with BLCKSZ 8192: (8192 - 24 - 8 - 4 - 4) / 6 / m - 2, at most 255 */
    int cap = (8192 - 24 - 8 - 4 - 4) / 6 / m - 2;
    g->max_level_cap = cap > 255 ? 255 : cap;
    g->elem_row = (int32_t *) malloc(sizeof(int32_t) * (size_t) (n + 1));
    g->elem_level = (int32_t *) malloc(sizeof(int32_t) * (size_t) (n + 1));
    g->tid_count = (int32_t *) malloc(sizeof(int32_t) * (size_t) (n + 1));
    g->tids = malloc(sizeof(int64_t[HNSW_HEAPTIDS]) * (size_t) (n + 1));
    g->nbr = (hcand **) calloc((size_t) n + 1, sizeof(hcand *));
    g->nbr_len = (int32_t **) calloc((size_t) n + 1, sizeof(int32_t *));
    orc_rng rng;
    rng_seed(&rng, seed);
    uint8_t *visited = (uint8_t *) malloc((size_t) n + 1);
    scand *wbuf = (scand *) malloc(sizeof(scand) * (size_t) (ef_construction + 2));
    hcand *lw = (hcand *) malloc(sizeof(hcand) * (size_t) (ef_construction + 2));
    hcand **lwp = (hcand **) malloc(sizeof(hcand *) * (size_t) (ef_construction + 2));
    hcand **sel = (hcand **) malloc(sizeof(hcand *) * (size_t) (ef_construction + 2));
    for (int64_t i = 0; i < n; i++) hnsw_insert(g, i, &rng, visited, wbuf, lw, lwp, sel);
    free(visited); free(wbuf); free(lw); free(lwp); free(sel);
    return g;
}

void orc_hnsw_free(void *h)
{
    orc_hnsw *g = (orc_hnsw *) h;
    if (!g) return;
    for (int32_t e = 0; e < g->n_elem; e++) { free(g->nbr[e]); free(g->nbr_len[e]); }
    free(g->elem_row); free(g->elem_level); free(g->tid_count); free(g->tids); free(g->nbr); free(g->nbr_len);
    free(g);
}

/* sizes: elements, entry element, its level, elements with level >= 1 */
void orc_hnsw_info(const void *h, int32_t *n_elem, int32_t *entry, int32_t *entry_level, int32_t *n_upper)
{
    const orc_hnsw *g = (const orc_hnsw *) h;
    *n_elem = g->n_elem;
    *entry = g->entry;
    *entry_level = g->entry >= 0 ? g->elem_level[g->entry] : -1;
    int32_t up = 0;
    for (int32_t e = 0; e < g->n_elem; e++) up += g->elem_level[e] >= 1;
    *n_upper = up;
}

/* Flat export for the GPU graph search (K4):
 *   level[e]; nbr0[e][2m] (-1 padded); heap TIDs: tid_count[e], tids[e][10] (-1 padded);
 *   upper levels: up_slot[e] (-1: level 0 only), up_nbr[(slot * max_level + (lc - 1)) * m + j] (-1 padded) */
void orc_hnsw_export(const void *h, int32_t *level, int32_t *nbr0, int32_t *tid_count, int64_t *tids, int32_t *up_slot,
                     int32_t *up_nbr, int32_t max_level)
{
    const orc_hnsw *g = (const orc_hnsw *) h;
    int32_t slot = 0;
    for (int32_t e = 0; e < g->n_elem; e++) {
        level[e] = g->elem_level[e];
        int32_t *len;
        hcand *nb = nbrs(g, e, 0, &len);
        for (int j = 0; j < 2 * g->m; j++) nbr0[(size_t) e * 2 * g->m + j] = j < *len ? nb[j].elem : -1;
        tid_count[e] = g->tid_count[e];
        for (int j = 0; j < HNSW_HEAPTIDS; j++) tids[(size_t) e * HNSW_HEAPTIDS + j] = j < g->tid_count[e] ? g->tids[e][j] : -1;
        up_slot[e] = -1;
        if (g->elem_level[e] >= 1) {
            up_slot[e] = slot;
            for (int lc = 1; lc <= max_level; lc++)
                for (int j = 0; j < g->m; j++) {
                    int32_t v = -1;
                    if (lc <= g->elem_level[e]) {
                        hcand *nu = nbrs(g, e, lc, &len);
                        if (j < *len) v = nu[j].elem;
                    }
                    up_nbr[((size_t) slot * max_level + (lc - 1)) * g->m + j] = v;
                }
            slot++;
        }
    }
}

/* GetScanItems, hnswscan.c:15-45 + the TID emission of hnswgettuple, :278-311: greedy descent with ef = 1, then
 * HnswSearchLayer(ef_search) on layer 0; candidates nearest first, every heap TID of an element (newest first, :286-296).
 * Returns the number of rows written (<= ef * 10); out_dist = the index distance as float8; visited count in *n_visited. */
static int64_t hnsw_search_impl(const void *h, const float *q, int ef, int64_t *out_rows, double *out_dist, int32_t *out_elems,
                                int64_t *n_visited, const uint8_t *allowed_rows);

int64_t orc_hnsw_search(const void *h, const float *q, int ef, int64_t *out_rows, double *out_dist, int32_t *out_elems,
                        int64_t *n_visited)
{
    return hnsw_search_impl(h, q, ef, out_rows, out_dist, out_elems, n_visited, NULL);
}

/* the predicate-aware walk (search_layer_pa) behind the same descent and TID emission; the rows come back unfiltered: the
 * caller applies allowed_rows to them like the executor's qual (an entry point that is not allowed can be among them) */
int64_t orc_hnsw_search_pa(const void *h, const float *q, int ef, const uint8_t *allowed_rows, int64_t *out_rows,
                           double *out_dist, int32_t *out_elems, int64_t *n_visited)
{
    return hnsw_search_impl(h, q, ef, out_rows, out_dist, out_elems, n_visited, allowed_rows);
}

static int64_t hnsw_search_impl(const void *h, const float *q, int ef, int64_t *out_rows, double *out_dist, int32_t *out_elems,
                                int64_t *n_visited, const uint8_t *allowed_rows)
{
    const orc_hnsw *g = (const orc_hnsw *) h;
    if (n_visited) *n_visited = 0;
    if (g->entry < 0) return 0;
    uint8_t *visited = (uint8_t *) malloc((size_t) g->n_elem);
    scand *w = (scand *) malloc(sizeof(scand) * (size_t) (ef + 2));
    scand *ep = (scand *) malloc(sizeof(scand) * (size_t) (ef + 2));
    ep[0].dist = index_distance(g->metric, g->dim, q, evec(g, g->entry));
    ep[0].elem = g->entry;
    int n_ep = 1;
    for (int lc = g->elem_level[g->entry]; lc >= 1; lc--) {
        int nw = search_layer(g, q, ep, n_ep, 1, lc, w, visited, NULL);
        memcpy(ep, w, sizeof(scand) * (size_t) nw);
        n_ep = nw;
    }
    int nw = allowed_rows ? search_layer_pa(g, q, ep, n_ep, ef, w, visited, n_visited, allowed_rows)
                          : search_layer(g, q, ep, n_ep, ef, 0, w, visited, n_visited);
    int64_t out = 0;
    for (int i = nw - 1; i >= 0; i--) {                /* nearest first */
        const int32_t e = w[i].elem;
        if (out_elems) out_elems[nw - 1 - i] = e;
        for (int t = g->tid_count[e] - 1; t >= 0; t--) {
            out_rows[out] = g->tids[e][t];
            out_dist[out] = w[i].dist;
            out++;
        }
    }
    free(visited); free(w); free(ep);
    return out;
}
