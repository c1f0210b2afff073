/*
 * fake_vsrbac.c — TEST INFRASTRUCTURE: a CPU stand-in for the few libvsrbac entry points the sidecar calls, so that the
 * sidecar's protocol, residency and versioning can be exercised without a GPU (tests/test_sidecar_cpu.py).  Never shipped,
 * never linked into the product: the real library has no CPU path.
 */
#include "vsrbac.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct vsr_ctx { int device; };
struct vsr_corpus
{
	int64_t		n;
	int			dim;
	float	   *rows;
	int32_t    *doc;
	int64_t    *blk;
	int64_t		n_ur, n_pa;
	int32_t    *ur_user, *ur_role, *pa_role, *pa_doc;
};
struct vsr_filter { vsr_corpus *c; int32_t user; };
struct vsr_hnsw { vsr_corpus *c; };
struct vsr_ivf { vsr_corpus *c; };

static char last_error[256] = "";
static struct vsr_filter filter_slot[64];
static int	filter_next;

const char *vsr_last_error(void) { return last_error; }
int vsr_open(int device, vsr_ctx **out) { *out = calloc(1, sizeof(**out)); (*out)->device = device; return VSR_OK; }
int vsr_close(vsr_ctx *ctx) { free(ctx); return VSR_OK; }

int
vsr_corpus_load(vsr_ctx *ctx, const float *rows, int64_t n, int dim, const int64_t *blk, const int32_t *doc, int64_t off, vsr_corpus **out)
{
	vsr_corpus *c = calloc(1, sizeof(*c));

	(void) ctx; (void) off;
	c->n = n;
	c->dim = dim;
	c->rows = malloc(sizeof(float) * (size_t) (n ? n : 1) * dim);
	memcpy(c->rows, rows, sizeof(float) * (size_t) n * dim);
	c->doc = calloc((size_t) (n ? n : 1), sizeof(int32_t));
	c->blk = calloc((size_t) (n ? n : 1), sizeof(int64_t));
	for (int64_t i = 0; i < n; i++)
	{
		c->doc[i] = doc ? doc[i] : 0;
		c->blk[i] = blk ? blk[i] : i;
	}
	*out = c;
	return VSR_OK;
}

int
vsr_corpus_free(vsr_corpus *c)
{
	if (c)
	{
		free(c->rows); free(c->doc); free(c->blk); free(c->ur_user); free(c->ur_role); free(c->pa_role); free(c->pa_doc);
		free(c);
	}
	return VSR_OK;
}

static int32_t *dup32(const int32_t *a, int64_t n) { int32_t *r = malloc(sizeof(int32_t) * (size_t) (n ? n : 1)); memcpy(r, a, sizeof(int32_t) * (size_t) n); return r; }

int
vsr_rbac_load(vsr_corpus *c, const int32_t *uu, const int32_t *ur, int64_t n_ur, const int32_t *pr, const int32_t *pd, int64_t n_pa)
{
	free(c->ur_user); free(c->ur_role); free(c->pa_role); free(c->pa_doc);
	c->n_ur = n_ur; c->n_pa = n_pa;
	c->ur_user = dup32(uu, n_ur); c->ur_role = dup32(ur, n_ur); c->pa_role = dup32(pr, n_pa); c->pa_doc = dup32(pd, n_pa);
	return VSR_OK;
}

int
vsr_filter_for_user(vsr_corpus *c, int32_t user, int mode, vsr_filter **out)
{
	(void) mode;
	if (!c->ur_user) { snprintf(last_error, sizeof last_error, "vsr_filter_for_user: call vsr_rbac_load first"); return VSR_ERR_NO_RBAC; }
	filter_slot[filter_next % 64].c = c;
	filter_slot[filter_next % 64].user = user;
	*out = &filter_slot[filter_next++ % 64];
	return VSR_OK;
}

static int
allowed(const vsr_corpus *c, int32_t user, int32_t doc)
{
	for (int64_t i = 0; i < c->n_ur; i++)
		if (c->ur_user[i] == user)
			for (int64_t j = 0; j < c->n_pa; j++)
				if (c->pa_role[j] == c->ur_role[i] && c->pa_doc[j] == doc) return 1;
	return 0;
}

int
vsr_search(vsr_corpus *c, const float *q, int nq, int dim, int k, int metric, const vsr_filter *const *filters, int64_t *oblk,
		   int32_t *odoc, int64_t *orow, float *odist, int32_t *ocnt)
{
	(void) metric;
	if (dim != c->dim) { snprintf(last_error, sizeof last_error, "different vector dimensions %d and %d", c->dim, dim); return VSR_ERR_DIM_MISMATCH; }
	for (int qi = 0; qi < nq; qi++)
	{
		int			cnt = 0;

		for (int i = 0; i < k; i++) { oblk[qi * k + i] = -1; if (orow) orow[qi * k + i] = -1; if (odoc) odoc[qi * k + i] = -1; odist[qi * k + i] = INFINITY; }
		for (int64_t r = 0; r < c->n; r++)
		{
			double		s = 0;
			int			at;

			if (filters && filters[qi] && !allowed(c, filters[qi]->user, c->doc[r])) continue;
			for (int t = 0; t < dim; t++) { double d = (double) c->rows[r * dim + t] - q[qi * dim + t]; s += d * d; }
			s = sqrt(s);
			for (at = cnt; at > 0 && odist[qi * k + at - 1] > (float) s; at--) ;
			if (at >= k) continue;
			for (int m = (cnt < k ? cnt : k - 1); m > at; m--)
			{
				odist[qi * k + m] = odist[qi * k + m - 1]; oblk[qi * k + m] = oblk[qi * k + m - 1];
				if (orow) orow[qi * k + m] = orow[qi * k + m - 1];
			}
			odist[qi * k + at] = (float) s; oblk[qi * k + at] = c->blk[r]; if (orow) orow[qi * k + at] = r;
			if (cnt < k) cnt++;
		}
		ocnt[qi] = cnt;
	}
	return VSR_OK;
}

int vsr_hnsw_load(vsr_corpus *c, int m, int32_t n, int32_t e, const int32_t *l, const int32_t *nb, const int32_t *tc, const int64_t *t,
				  const int32_t *us, const int32_t *un, int32_t nu, int32_t ml, vsr_hnsw **out)
{ (void) m; (void) n; (void) e; (void) l; (void) nb; (void) tc; (void) t; (void) us; (void) un; (void) nu; (void) ml; *out = calloc(1, sizeof(**out)); (*out)->c = c; return VSR_OK; }
int vsr_hnsw_free(vsr_hnsw *h) { free(h); return VSR_OK; }
int vsr_hnsw_search(vsr_hnsw *h, const float *q, int nq, int dim, int k, int ef, int metric, const vsr_filter *const *f, int64_t *b,
					int32_t *d, int64_t *r, float *di, int32_t *cn, int64_t *vis)
{ (void) ef; (void) vis; return vsr_search(h->c, q, nq, dim, k, metric, f, b, d, r, di, cn); }
int vsr_ivf_load(vsr_corpus *c, const float *centers, int lists, const int32_t *row_list, vsr_ivf **out)
{ (void) centers; (void) lists; (void) row_list; *out = calloc(1, sizeof(**out)); (*out)->c = c; return VSR_OK; }
int vsr_ivf_free(vsr_ivf *v) { free(v); return VSR_OK; }
int vsr_ivf_search(vsr_ivf *v, const float *q, int nq, int dim, int k, int probes, int metric, const vsr_filter *const *f, int64_t *b,
				   int32_t *d, int64_t *r, float *di, int32_t *cn)
{ (void) probes; return vsr_search(v->c, q, nq, dim, k, metric, f, b, d, r, di, cn); }
