/*
 * vsr_sidecar.c — the resident GPU process of the PostgreSQL shim: one per GPU, owns the vsr_ctx and every corpus.
 *
 *     vsr_sidecar <socket path> [device ordinal]
 *
 * Serves the protocol of vsr_sidecar.h over a UNIX stream socket: a poll() loop, one request at a time (searches are
 * synchronous GPU calls; the library is used from this one thread, as include/vsrbac.h asks).  Corpora stay resident
 * until dropped, replaced by a newer version of the same key, or the process ends -- across backends and connections,
 * which is the point (the reference harness connects anew for every search: prefilter_role.py:86).
 *
 * Plain C over the C ABI of include/vsrbac.h: builds with `cc vsr_sidecar.c -lvsrbac` and needs no PostgreSQL.
 */
#include "vsr_sidecar.h"
#include "vsrbac.h"

#include <errno.h>
#include <poll.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <unistd.h>

#define MAX_CLIENTS 256
#define MAX_CORPORA 1024

typedef struct
{
	int			used;
	uint64_t	key,
				version,
				handle;
	vsr_corpus *corpus;
	vsr_hnsw   *hnsw;
	vsr_ivf    *ivf;
	int64_t		nrows;
	int			dim,
				has_rbac;
}			entry;

static vsr_ctx *ctx;
static entry corpora[MAX_CORPORA];
static uint64_t next_handle = 1;
static volatile sig_atomic_t stop;

static void
on_signal(int s)
{
	(void) s;
	stop = 1;
}

static int
io_all(int fd, void *buf, size_t n, int writing)
{
	char	   *p = (char *) buf;

	while (n > 0)
	{
		ssize_t		r = writing ? write(fd, p, n) : read(fd, p, n);

		if (r < 0 && errno == EINTR)
			continue;
		if (r <= 0)
			return -1;
		p += r;
		n -= (size_t) r;
	}
	return 0;
}

static void
drop_entry(entry * e)
{
	if (!e->used)
		return;
	if (e->hnsw)
		(void) vsr_hnsw_free(e->hnsw);	/* index structures before their corpus (include/vsrbac.h) */
	if (e->ivf)
		(void) vsr_ivf_free(e->ivf);
	if (e->corpus)
		(void) vsr_corpus_free(e->corpus);
	memset(e, 0, sizeof(*e));
}

static entry *
by_key(uint64_t key)
{
	for (int i = 0; i < MAX_CORPORA; i++)
		if (corpora[i].used && corpora[i].key == key)
			return &corpora[i];
	return NULL;
}

static entry *
by_handle(uint64_t handle)
{
	for (int i = 0; i < MAX_CORPORA; i++)
		if (corpora[i].used && corpora[i].handle == handle)
			return &corpora[i];
	return NULL;
}

static int
reply(int fd, int status, const void *payload, uint64_t bytes)
{
	vsr_sc_reply r;

	memset(&r, 0, sizeof(r));
	r.status = status;
	r.payload_bytes = status == 0 ? bytes : 0;
	if (status != 0)
		snprintf(r.msg, sizeof(r.msg), "%s", status == VSR_SC_NOTFOUND ? "no such corpus" : vsr_last_error());
	if (io_all(fd, &r, sizeof(r), 1))
		return -1;
	if (status == 0 && bytes && io_all(fd, (void *) payload, (size_t) bytes, 1))
		return -1;
	return 0;
}

static int
reply_msg(int fd, int status, const char *msg)
{
	vsr_sc_reply r;

	memset(&r, 0, sizeof(r));
	r.status = status;
	snprintf(r.msg, sizeof(r.msg), "%s", msg);
	return io_all(fd, &r, sizeof(r), 1);
}

static void
fill_info(const entry * e, vsr_sc_corpus_info * info)
{
	memset(info, 0, sizeof(*info));
	info->handle = e->handle;
	info->nrows = e->nrows;
	info->dim = e->dim;
	info->has_rbac = e->has_rbac;
	info->has_hnsw = e->hnsw != NULL;
	info->has_ivf = e->ivf != NULL;
}

/* one request whose payload (`n` bytes) is in `buf`; returns -1 when the connection must be closed, 1 for shutdown */
static int
serve(int fd, uint32_t op, char *buf, uint64_t n)
{
	switch (op)
	{
		case VSR_SC_PING:
			return reply(fd, 0, NULL, 0);
		case VSR_SC_SHUTDOWN:
			(void) reply(fd, 0, NULL, 0);
			return 1;
		case VSR_SC_CORPUS_LOOKUP:
		case VSR_SC_CORPUS_DROP:
			{
				vsr_sc_key *k = (vsr_sc_key *) buf;
				entry	   *e;
				vsr_sc_corpus_info info;

				if (n != sizeof(*k))
					return reply_msg(fd, VSR_ERR_INVALID, "malformed request");
				e = by_key(k->key);
				if (op == VSR_SC_CORPUS_DROP)
				{
					if (e)
						drop_entry(e);
					return reply(fd, 0, NULL, 0);
				}
				if (e && e->version != k->version)
				{
					drop_entry(e);	/* the heap or the RBAC tables changed since this copy was loaded */
					e = NULL;
				}
				if (!e)
					return reply(fd, VSR_SC_NOTFOUND, NULL, 0);
				fill_info(e, &info);
				return reply(fd, 0, &info, sizeof(info));
			}
		case VSR_SC_CORPUS_LOAD:
			{
				vsr_sc_load_req *l = (vsr_sc_load_req *) buf;
				size_t		rows_b,
							blk_b,
							doc_b;
				entry	   *e;
				vsr_sc_corpus_info info;
				int			rc;

				if (n < sizeof(*l) || l->nrows < 0 || l->dim < 1)
					return reply_msg(fd, VSR_ERR_INVALID, "malformed request");
				rows_b = sizeof(float) * (size_t) l->nrows * (size_t) l->dim;
				blk_b = l->has_blk ? sizeof(int64_t) * (size_t) l->nrows : 0;
				doc_b = l->has_doc ? sizeof(int32_t) * (size_t) l->nrows : 0;
				if (n != sizeof(*l) + rows_b + blk_b + doc_b)
					return reply_msg(fd, VSR_ERR_INVALID, "malformed request");
				if ((e = by_key(l->key)) != NULL)
					drop_entry(e);
				for (e = corpora; e < corpora + MAX_CORPORA && e->used; e++)
					;
				if (e == corpora + MAX_CORPORA)
					return reply_msg(fd, VSR_ERR_OOM, "too many resident corpora");
				memset(e, 0, sizeof(*e));
				rc = vsr_corpus_load(ctx, (const float *) (buf + sizeof(*l)), l->nrows, l->dim,
									 l->has_blk ? (const int64_t *) (buf + sizeof(*l) + rows_b) : NULL,
									 l->has_doc ? (const int32_t *) (buf + sizeof(*l) + rows_b + blk_b) : NULL, 0, &e->corpus);
				if (rc)
					return reply(fd, rc, NULL, 0);
				e->used = 1;
				e->key = l->key;
				e->version = l->version;
				e->handle = next_handle++;
				e->nrows = l->nrows;
				e->dim = l->dim;
				fill_info(e, &info);
				return reply(fd, 0, &info, sizeof(info));
			}
		case VSR_SC_RBAC_LOAD:
			{
				vsr_sc_rbac_req *r = (vsr_sc_rbac_req *) buf;
				entry	   *e;
				const int32_t *a;
				int			rc;

				if (n < sizeof(*r) || r->n_user_roles < 0 || r->n_permissions < 0 ||
					n != sizeof(*r) + 8 * ((size_t) r->n_user_roles + (size_t) r->n_permissions))
					return reply_msg(fd, VSR_ERR_INVALID, "malformed request");
				if ((e = by_handle(r->handle)) == NULL)
					return reply(fd, VSR_SC_NOTFOUND, NULL, 0);
				a = (const int32_t *) (buf + sizeof(*r));
				rc = vsr_rbac_load(e->corpus, a, a + r->n_user_roles, r->n_user_roles, a + 2 * r->n_user_roles,
								   a + 2 * r->n_user_roles + r->n_permissions, r->n_permissions);
				if (rc == 0)
					e->has_rbac = 1;
				return reply(fd, rc, NULL, 0);
			}
		case VSR_SC_HNSW_LOAD:
			{
				vsr_sc_hnsw_req *h = (vsr_sc_hnsw_req *) buf;
				entry	   *e;
				size_t		ne,
							want;
				const char *p = buf + sizeof(*h);
				const int32_t *level,
						   *nbr0,
						   *tid_count,
						   *up_slot,
						   *up_nbr;
				const int64_t *tids;
				int			rc;

				if (n < sizeof(*h) || h->n_elem < 0 || h->m < 1 || h->n_upper < 0 || h->max_level < 1)
					return reply_msg(fd, VSR_ERR_INVALID, "malformed request");
				ne = (size_t) h->n_elem;
				want = sizeof(*h) + ne * 4 + ne * 2 * (size_t) h->m * 4 + ne * 4 + ne * 80 + ne * 4 +
					(size_t) h->n_upper * (size_t) h->max_level * (size_t) h->m * 4;
				if (n != want)
					return reply_msg(fd, VSR_ERR_INVALID, "malformed request");
				if ((e = by_handle(h->handle)) == NULL)
					return reply(fd, VSR_SC_NOTFOUND, NULL, 0);
				level = (const int32_t *) p;
				p += ne * 4;
				nbr0 = (const int32_t *) p;
				p += ne * 2 * (size_t) h->m * 4;
				tid_count = (const int32_t *) p;
				p += ne * 4;
				tids = (const int64_t *) p;
				p += ne * 80;
				up_slot = (const int32_t *) p;
				p += ne * 4;
				up_nbr = (const int32_t *) p;
				if (e->hnsw)
				{
					(void) vsr_hnsw_free(e->hnsw);
					e->hnsw = NULL;
				}
				rc = vsr_hnsw_load(e->corpus, h->m, h->n_elem, h->entry, level, nbr0, tid_count, tids, up_slot, up_nbr, h->n_upper,
								   h->max_level, &e->hnsw);
				return reply(fd, rc, NULL, 0);
			}
		case VSR_SC_IVF_LOAD:
			{
				vsr_sc_ivf_req *v = (vsr_sc_ivf_req *) buf;
				entry	   *e;
				int			rc;

				if (n < sizeof(*v) || v->lists < 1)
					return reply_msg(fd, VSR_ERR_INVALID, "malformed request");
				if ((e = by_handle(v->handle)) == NULL)
					return reply(fd, VSR_SC_NOTFOUND, NULL, 0);
				if (n != sizeof(*v) + sizeof(float) * (size_t) v->lists * (size_t) e->dim + sizeof(int32_t) * (size_t) e->nrows)
					return reply_msg(fd, VSR_ERR_INVALID, "malformed request");
				if (e->ivf)
				{
					(void) vsr_ivf_free(e->ivf);
					e->ivf = NULL;
				}
				rc = vsr_ivf_load(e->corpus, (const float *) (buf + sizeof(*v)), v->lists,
								  (const int32_t *) (buf + sizeof(*v) + sizeof(float) * (size_t) v->lists * (size_t) e->dim), &e->ivf);
				return reply(fd, rc, NULL, 0);
			}
		case VSR_SC_SEARCH:
			{
				vsr_sc_search_req *s = (vsr_sc_search_req *) buf;
				entry	   *e;
				vsr_filter *f = NULL;
				const vsr_filter **fl = NULL;
				size_t		nk;
				char	   *out;
				vsr_sc_result *res;
				int32_t    *counts;
				int64_t    *rows,
						   *blk;
				float	   *dist;
				int			rc;

				if (n < sizeof(*s) || s->nq < 0 || s->k < 1 || s->k > VSR_MAX_K || s->dim < 1 ||
					n != sizeof(*s) + sizeof(float) * (size_t) s->nq * (size_t) s->dim)
					return reply_msg(fd, VSR_ERR_INVALID, "malformed request");
				if ((e = by_handle(s->handle)) == NULL)
					return reply(fd, VSR_SC_NOTFOUND, NULL, 0);
				if (s->filter_mode >= 0)
				{
					if ((rc = vsr_filter_for_user(e->corpus, s->user_id, s->filter_mode, &f)))
						return reply(fd, rc, NULL, 0);
					fl = (const vsr_filter **) malloc(sizeof(*fl) * (size_t) (s->nq > 0 ? s->nq : 1));
					for (int i = 0; i < s->nq; i++)
						fl[i] = f;
				}
				nk = (size_t) s->nq * (size_t) s->k;
				out = (char *) calloc(1, sizeof(*res) + VSR_SC_COUNTS_BYTES(s->nq) + nk * 20 + 16);
				res = (vsr_sc_result *) out;
				res->nq = s->nq;
				res->k = s->k;
				counts = (int32_t *) (out + sizeof(*res));
				rows = (int64_t *) (out + sizeof(*res) + VSR_SC_COUNTS_BYTES(s->nq));
				blk = rows + nk;
				dist = (float *) (blk + nk);
				if (s->index == 1 && e->hnsw)
					rc = vsr_hnsw_search(e->hnsw, (const float *) (buf + sizeof(*s)), s->nq, s->dim, s->k, s->param, s->metric, fl, blk,
										 NULL, rows, dist, counts, NULL);
				else if (s->index == 2 && e->ivf)
					rc = vsr_ivf_search(e->ivf, (const float *) (buf + sizeof(*s)), s->nq, s->dim, s->k, s->param, s->metric, fl, blk,
										NULL, rows, dist, counts);
				else if (s->index != 0)
				{
					free(out);
					free(fl);
					return reply_msg(fd, VSR_ERR_INVALID, "the index structure of this corpus has not been loaded");
				}
				else
					rc = vsr_search(e->corpus, (const float *) (buf + sizeof(*s)), s->nq, s->dim, s->k, s->metric, fl, blk, NULL, rows,
									dist, counts);
				rc = reply(fd, rc, out, sizeof(*res) + VSR_SC_COUNTS_BYTES(s->nq) + nk * 20);
				free(out);
				free(fl);
				return rc;
			}
		default:
			return reply_msg(fd, VSR_ERR_INVALID, "unknown request");
	}
}

int
main(int argc, char **argv)
{
	struct sockaddr_un sa;
	struct pollfd fds[MAX_CLIENTS + 1];
	int			nfds = 1,
				lfd,
				rc;

	if (argc < 2)
	{
		fprintf(stderr, "usage: %s <socket path> [device ordinal]\n", argv[0]);
		return 2;
	}
	signal(SIGPIPE, SIG_IGN);
	signal(SIGTERM, on_signal);
	signal(SIGINT, on_signal);
	if ((rc = vsr_open(argc > 2 ? atoi(argv[2]) : 0, &ctx)) != 0)
	{
		fprintf(stderr, "vsr_sidecar: %s\n", vsr_last_error());
		return 1;
	}
	lfd = socket(AF_UNIX, SOCK_STREAM, 0);
	memset(&sa, 0, sizeof(sa));
	sa.sun_family = AF_UNIX;
	if (lfd < 0 || strlen(argv[1]) >= sizeof(sa.sun_path))
	{
		fprintf(stderr, "vsr_sidecar: bad socket path\n");
		return 1;
	}
	strcpy(sa.sun_path, argv[1]);
	unlink(argv[1]);
	if (bind(lfd, (struct sockaddr *) &sa, sizeof(sa)) != 0 || chmod(argv[1], 0600) != 0 || listen(lfd, 64) != 0)
	{
		fprintf(stderr, "vsr_sidecar: cannot listen on %s: %s\n", argv[1], strerror(errno));
		return 1;
	}
	fds[0].fd = lfd;
	fds[0].events = POLLIN;
	printf("vsr_sidecar: listening on %s\n", argv[1]);
	fflush(stdout);
	while (!stop)
	{
		if (poll(fds, (nfds_t) nfds, 1000) <= 0)
			continue;
		if ((fds[0].revents & POLLIN) && nfds <= MAX_CLIENTS)
		{
			int			cfd = accept(lfd, NULL, NULL);

			if (cfd >= 0)
			{
				fds[nfds].fd = cfd;
				fds[nfds].events = POLLIN;
				fds[nfds].revents = 0;
				nfds++;
			}
		}
		for (int i = 1; i < nfds && !stop; i++)
		{
			vsr_sc_hdr	h;
			char	   *buf = NULL;
			int			r = -1;

			if (!(fds[i].revents & (POLLIN | POLLHUP | POLLERR)))
				continue;
			if (io_all(fds[i].fd, &h, sizeof(h), 0) == 0 && h.magic == VSR_SC_MAGIC && h.payload_bytes < ((uint64_t) 1 << 40) &&
				(buf = (char *) malloc((size_t) h.payload_bytes + 8)) != NULL &&
				(h.payload_bytes == 0 || io_all(fds[i].fd, buf, (size_t) h.payload_bytes, 0) == 0))
				r = serve(fds[i].fd, h.op, buf, h.payload_bytes);
			free(buf);
			if (r == 1)
				stop = 1;
			if (r != 0)
			{
				close(fds[i].fd);
				fds[i] = fds[--nfds];
				i--;
			}
		}
	}
	for (int i = 0; i < MAX_CORPORA; i++)
		drop_entry(&corpora[i]);
	(void) vsr_close(ctx);
	close(lfd);
	unlink(argv[1]);
	return 0;
}
