/*
 * vsr_sidecar.h — wire protocol between PostgreSQL backends and the resident GPU process (vsr_sidecar).
 *
 * Why a sidecar: the reference harness opens a NEW connection for every search
 * (controller/baseline/prefilter/prefilter_role.py:86, row_level_security.py:105, dynamic_partition/search.py:36), and a
 * PostgreSQL backend is a forked single-threaded process that must not inherit a GPU context.  A corpus loaded by one
 * backend would die with it.  The sidecar is ONE process per GPU that owns the vsr_ctx, every resident corpus, their RBAC
 * tables, filters and index structures; backends keep nothing but a socket.
 *
 * Transport: UNIX stream socket (GUC vsrbac.sidecar = its path).  Every request is a vsr_sc_hdr followed by
 * `payload_bytes` of payload; every reply is a vsr_sc_reply followed by its payload.  Integers are host-endian (same
 * machine by construction).  The call set is the subset of include/vsrbac.h the shim uses, one opcode per entry point, with
 * handles replaced by 64-bit ids the sidecar hands out.  A corpus is named by a 64-bit key (database oid << 32 | index oid)
 * and carries the VERSION the loading backend computed for its heap (vsr_pg.c: relfilenode, block count, RBAC table
 * fingerprints); a lookup with another version drops the stale copy.
 */
#ifndef VSR_SIDECAR_H
#define VSR_SIDECAR_H

#include <stdint.h>

#define VSR_SC_MAGIC 0x43525356u	/* "VSRC" */
#define VSR_SC_NOTFOUND 100			/* reply status of a lookup that found nothing (or a stale version) */

typedef enum
{
	VSR_SC_PING = 1,
	VSR_SC_CORPUS_LOOKUP,			/* vsr_sc_key                         -> vsr_sc_corpus_info */
	VSR_SC_CORPUS_LOAD,				/* vsr_sc_load_req + rows [+ blk] [+ doc] -> vsr_sc_corpus_info   (vsr_corpus_load) */
	VSR_SC_RBAC_LOAD,				/* vsr_sc_rbac_req + four int32 arrays    -> (none)               (vsr_rbac_load) */
	VSR_SC_SEARCH,					/* vsr_sc_search_req + queries            -> vsr_sc_result        (vsr_search, filter = user) */
	VSR_SC_HNSW_LOAD,				/* vsr_sc_hnsw_req + graph arrays         -> (none)               (vsr_hnsw_load) */
	VSR_SC_IVF_LOAD,				/* vsr_sc_ivf_req + centres + row_list    -> (none)               (vsr_ivf_load) */
	VSR_SC_CORPUS_DROP,				/* vsr_sc_key                         -> (none) */
	VSR_SC_SHUTDOWN
}			vsr_sc_op;

typedef struct
{
	uint32_t	magic;
	uint32_t	op;
	uint64_t	payload_bytes;
}			vsr_sc_hdr;

typedef struct
{
	int32_t		status;				/* vsr_status, or VSR_SC_NOTFOUND */
	uint32_t	pad;
	uint64_t	payload_bytes;
	char		msg[240];			/* vsr_last_error() of the failing call */
}			vsr_sc_reply;

typedef struct
{
	uint64_t	key;
	uint64_t	version;
}			vsr_sc_key;

typedef struct
{
	uint64_t	handle;
	int64_t		nrows;
	int32_t		dim;
	int32_t		has_rbac;
	int32_t		has_hnsw;
	int32_t		has_ivf;
}			vsr_sc_corpus_info;

typedef struct
{
	uint64_t	key;
	uint64_t	version;
	int64_t		nrows;
	int32_t		dim;
	int32_t		has_blk;			/* payload: float rows[nrows*dim], then int64 blk[nrows] (has_blk), then int32 doc[nrows] (has_doc) */
	int32_t		has_doc;
	int32_t		pad;
}			vsr_sc_load_req;

typedef struct
{
	uint64_t	handle;
	int64_t		n_user_roles;		/* payload: int32 ur_user[], ur_role[], then int32 pa_role[], pa_doc[] */
	int64_t		n_permissions;
}			vsr_sc_rbac_req;

typedef struct
{
	uint64_t	handle;
	int32_t		nq;
	int32_t		dim;
	int32_t		k;
	int32_t		metric;				/* vsr_metric */
	int32_t		filter_mode;		/* -1: no filter; else vsr_filter_mode for the user below (vsr_filter_for_user) */
	int32_t		user_id;
	int32_t		index;				/* 0: exact search; 1: hnsw graph walk (param = ef_search); 2: ivfflat probe (param = probes) */
	int32_t		param;				/* payload: float queries[nq*dim] */
}			vsr_sc_search_req;

typedef struct						/* reply payload: this header, then int32 counts[nq] (padded to 8 bytes), int64 rows[nq*k],
									 * int64 blk[nq*k], float dist[nq*k] */
{
	int32_t		nq;
	int32_t		k;
}			vsr_sc_result;
#define VSR_SC_COUNTS_BYTES(nq) ((((size_t) (nq) + 1) / 2) * 8)

typedef struct
{
	uint64_t	handle;
	int32_t		m;
	int32_t		n_elem;
	int32_t		entry;
	int32_t		n_upper;
	int32_t		max_level;
	int32_t		pad;				/* payload: int32 level[n_elem], nbr0[n_elem*2m], tid_count[n_elem]; int64 tids[n_elem*10];
									 * int32 up_slot[n_elem], up_nbr[n_upper*max_level*m] */
}			vsr_sc_hnsw_req;

typedef struct
{
	uint64_t	handle;
	int32_t		lists;
	int32_t		pad;				/* payload: float centers[lists*dim], int32 row_list[nrows] */
}			vsr_sc_ivf_req;

/* ---- client side (pg_shim/vsr_client.c): plain C, no PostgreSQL dependency ---- */
typedef struct vsr_sc_conn vsr_sc_conn;

vsr_sc_conn *vsr_sc_connect(const char *socket_path, char *err, int err_len);
void		vsr_sc_close(vsr_sc_conn * c);
const char *vsr_sc_error(const vsr_sc_conn * c);	/* message of the last failed call on this connection */
int			vsr_sc_ping(vsr_sc_conn * c);
int			vsr_sc_corpus_lookup(vsr_sc_conn * c, uint64_t key, uint64_t version, vsr_sc_corpus_info * out);
int			vsr_sc_corpus_load(vsr_sc_conn * c, uint64_t key, uint64_t version, const float *rows, int64_t n, int dim,
							   const int64_t *blk, const int32_t *doc, vsr_sc_corpus_info * out);
int			vsr_sc_rbac_load(vsr_sc_conn * c, uint64_t handle, const int32_t *ur_user, const int32_t *ur_role, int64_t n_ur,
							 const int32_t *pa_role, const int32_t *pa_doc, int64_t n_pa);
int			vsr_sc_search(vsr_sc_conn * c, const vsr_sc_search_req * req, const float *queries, int32_t *counts, int64_t *rows,
						  int64_t *blk, float *dist);
int			vsr_sc_hnsw_load(vsr_sc_conn * c, const vsr_sc_hnsw_req * req, int dim2m_unused, const int32_t *level,
							 const int32_t *nbr0, const int32_t *tid_count, const int64_t *tids, const int32_t *up_slot,
							 const int32_t *up_nbr);
int			vsr_sc_ivf_load(vsr_sc_conn * c, const vsr_sc_ivf_req * req, int dim, int64_t nrows, const float *centers,
							const int32_t *row_list);
int			vsr_sc_corpus_drop(vsr_sc_conn * c, uint64_t key);
int			vsr_sc_shutdown(vsr_sc_conn * c);

#endif							/* VSR_SIDECAR_H */
