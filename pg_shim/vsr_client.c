/*
 * vsr_client.c — backend side of the sidecar protocol (vsr_sidecar.h).  Plain C over a UNIX socket: compiles without
 * PostgreSQL (tests/test_sidecar_cpu.py builds it with gcc and drives a sidecar linked against a stand-in engine).
 */
#include "vsr_sidecar.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/uio.h>
#include <sys/un.h>
#include <unistd.h>

struct vsr_sc_conn
{
	int			fd;
	char		err[256];
};

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

vsr_sc_conn *
vsr_sc_connect(const char *path, char *err, int err_len)
{
	struct sockaddr_un sa;
	vsr_sc_conn *c;
	int			fd = socket(AF_UNIX, SOCK_STREAM, 0);

	if (fd < 0 || strlen(path) >= sizeof(sa.sun_path))
	{
		if (err)
			snprintf(err, (size_t) err_len, "vsrbac sidecar: cannot create a socket for %s", path);
		if (fd >= 0)
			close(fd);
		return NULL;
	}
	memset(&sa, 0, sizeof(sa));
	sa.sun_family = AF_UNIX;
	strcpy(sa.sun_path, path);
	if (connect(fd, (struct sockaddr *) &sa, sizeof(sa)) != 0)
	{
		if (err)
			snprintf(err, (size_t) err_len, "vsrbac sidecar: cannot connect to %s: %s", path, strerror(errno));
		close(fd);
		return NULL;
	}
	c = (vsr_sc_conn *) calloc(1, sizeof(*c));
	c->fd = fd;
	return c;
}

void
vsr_sc_close(vsr_sc_conn * c)
{
	if (c)
	{
		close(c->fd);
		free(c);
	}
}

const char *
vsr_sc_error(const vsr_sc_conn * c)
{
	return c ? c->err : "vsrbac sidecar: not connected";
}

/* one request (header, fixed part, up to 6 array parts) and its reply header; the reply payload is left on the socket */
typedef struct
{
	const void *p;
	size_t		n;
}			part;

static int
call(vsr_sc_conn * c, uint32_t op, const part * parts, int nparts, vsr_sc_reply * rep)
{
	vsr_sc_hdr	h;

	h.magic = VSR_SC_MAGIC;
	h.op = op;
	h.payload_bytes = 0;
	for (int i = 0; i < nparts; i++)
		h.payload_bytes += parts[i].n;
	if (io_all(c->fd, &h, sizeof(h), 1))
		goto broken;
	for (int i = 0; i < nparts; i++)
		if (parts[i].n && io_all(c->fd, (void *) parts[i].p, parts[i].n, 1))
			goto broken;
	if (io_all(c->fd, rep, sizeof(*rep), 0))
		goto broken;
	if (rep->status != 0)
	{
		rep->msg[sizeof(rep->msg) - 1] = 0;
		snprintf(c->err, sizeof(c->err), "%s", rep->msg);
	}
	return rep->status;
broken:
	snprintf(c->err, sizeof(c->err), "vsrbac sidecar: connection lost (%s)", strerror(errno));
	return -1;
}

static int
drain(vsr_sc_conn * c, uint64_t n)
{
	char		buf[4096];

	while (n > 0)
	{
		size_t		m = n < sizeof(buf) ? (size_t) n : sizeof(buf);

		if (io_all(c->fd, buf, m, 0))
			return -1;
		n -= m;
	}
	return 0;
}

int
vsr_sc_ping(vsr_sc_conn * c)
{
	vsr_sc_reply rep;
	int			rc = call(c, VSR_SC_PING, NULL, 0, &rep);

	return rc ? rc : drain(c, rep.payload_bytes);
}

static int
info_reply(vsr_sc_conn * c, int rc, vsr_sc_reply * rep, vsr_sc_corpus_info * out)
{
	if (rc != 0)
		return rc < 0 ? rc : (drain(c, rep->payload_bytes), rc);
	if (rep->payload_bytes != sizeof(*out) || io_all(c->fd, out, sizeof(*out), 0))
	{
		snprintf(c->err, sizeof(c->err), "vsrbac sidecar: malformed reply");
		return -1;
	}
	return 0;
}

int
vsr_sc_corpus_lookup(vsr_sc_conn * c, uint64_t key, uint64_t version, vsr_sc_corpus_info * out)
{
	vsr_sc_key	k = {key, version};
	part		ps[1] = {{&k, sizeof(k)}};
	vsr_sc_reply rep;

	return info_reply(c, call(c, VSR_SC_CORPUS_LOOKUP, ps, 1, &rep), &rep, out);
}

int
vsr_sc_corpus_load(vsr_sc_conn * c, uint64_t key, uint64_t version, const float *rows, int64_t n, int dim, const int64_t *blk,
				   const int32_t *doc, vsr_sc_corpus_info * out)
{
	vsr_sc_load_req l;
	part		ps[4];
	vsr_sc_reply rep;

	memset(&l, 0, sizeof(l));
	l.key = key;
	l.version = version;
	l.nrows = n;
	l.dim = dim;
	l.has_blk = blk != NULL;
	l.has_doc = doc != NULL;
	ps[0].p = &l;
	ps[0].n = sizeof(l);
	ps[1].p = rows;
	ps[1].n = sizeof(float) * (size_t) n * (size_t) dim;
	ps[2].p = blk;
	ps[2].n = blk ? sizeof(int64_t) * (size_t) n : 0;
	ps[3].p = doc;
	ps[3].n = doc ? sizeof(int32_t) * (size_t) n : 0;
	return info_reply(c, call(c, VSR_SC_CORPUS_LOAD, ps, 4, &rep), &rep, out);
}

int
vsr_sc_rbac_load(vsr_sc_conn * c, uint64_t handle, const int32_t *ur_user, const int32_t *ur_role, int64_t n_ur,
				 const int32_t *pa_role, const int32_t *pa_doc, int64_t n_pa)
{
	vsr_sc_rbac_req r = {handle, n_ur, n_pa};
	part		ps[5] = {{&r, sizeof(r)}, {ur_user, sizeof(int32_t) * (size_t) n_ur}, {ur_role, sizeof(int32_t) * (size_t) n_ur},
	{pa_role, sizeof(int32_t) * (size_t) n_pa}, {pa_doc, sizeof(int32_t) * (size_t) n_pa}};
	vsr_sc_reply rep;
	int			rc = call(c, VSR_SC_RBAC_LOAD, ps, 5, &rep);

	return rc < 0 ? rc : (drain(c, rep.payload_bytes), rc);
}

int
vsr_sc_search(vsr_sc_conn * c, const vsr_sc_search_req * req, const float *queries, int32_t *counts, int64_t *rows, int64_t *blk,
			  float *dist)
{
	part		ps[2] = {{req, sizeof(*req)}, {queries, sizeof(float) * (size_t) req->nq * (size_t) req->dim}};
	vsr_sc_reply rep;
	vsr_sc_result res;
	size_t		nk = (size_t) req->nq * (size_t) req->k;
	int			rc = call(c, VSR_SC_SEARCH, ps, 2, &rep);

	if (rc != 0)
		return rc < 0 ? rc : (drain(c, rep.payload_bytes), rc);
	if (rep.payload_bytes != sizeof(res) + VSR_SC_COUNTS_BYTES(req->nq) + nk * 20 || io_all(c->fd, &res, sizeof(res), 0) ||
		res.nq != req->nq || res.k != req->k || io_all(c->fd, counts, (size_t) req->nq * 4, 0) ||
		drain(c, VSR_SC_COUNTS_BYTES(req->nq) - (size_t) req->nq * 4) || io_all(c->fd, rows, nk * 8, 0) ||
		io_all(c->fd, blk, nk * 8, 0) || io_all(c->fd, dist, nk * 4, 0))
	{
		snprintf(c->err, sizeof(c->err), "vsrbac sidecar: malformed search reply");
		return -1;
	}
	return 0;
}

int
vsr_sc_hnsw_load(vsr_sc_conn * c, const vsr_sc_hnsw_req * req, int unused, const int32_t *level, const int32_t *nbr0,
				 const int32_t *tid_count, const int64_t *tids, const int32_t *up_slot, const int32_t *up_nbr)
{
	size_t		ne = (size_t) req->n_elem;
	part		ps[7] = {{req, sizeof(*req)}, {level, ne * 4}, {nbr0, ne * 2 * (size_t) req->m * 4}, {tid_count, ne * 4},
	{tids, ne * 10 * 8}, {up_slot, ne * 4}, {up_nbr, (size_t) req->n_upper * (size_t) req->max_level * (size_t) req->m * 4}};
	vsr_sc_reply rep;
	int			rc;

	(void) unused;
	rc = call(c, VSR_SC_HNSW_LOAD, ps, 7, &rep);
	return rc < 0 ? rc : (drain(c, rep.payload_bytes), rc);
}

int
vsr_sc_ivf_load(vsr_sc_conn * c, const vsr_sc_ivf_req * req, int dim, int64_t nrows, const float *centers, const int32_t *row_list)
{
	part		ps[3] = {{req, sizeof(*req)}, {centers, sizeof(float) * (size_t) req->lists * (size_t) dim},
	{row_list, sizeof(int32_t) * (size_t) nrows}};
	vsr_sc_reply rep;
	int			rc = call(c, VSR_SC_IVF_LOAD, ps, 3, &rep);

	return rc < 0 ? rc : (drain(c, rep.payload_bytes), rc);
}

int
vsr_sc_corpus_drop(vsr_sc_conn * c, uint64_t key)
{
	vsr_sc_key	k = {key, 0};
	part		ps[1] = {{&k, sizeof(k)}};
	vsr_sc_reply rep;
	int			rc = call(c, VSR_SC_CORPUS_DROP, ps, 1, &rep);

	return rc < 0 ? rc : (drain(c, rep.payload_bytes), rc);
}

int
vsr_sc_shutdown(vsr_sc_conn * c)
{
	vsr_sc_reply rep;
	int			rc = call(c, VSR_SC_SHUTDOWN, NULL, 0, &rep);

	return rc < 0 ? rc : (drain(c, rep.payload_bytes), rc);
}
