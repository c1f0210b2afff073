/*
 * vsr_pg.h — PostgreSQL extension shim over libvsrbac (include/vsrbac.h).
 *
 * The shim is pgvector with its two scan state machines swapped out: the SQL surface (sql/vector.sql, vector.control),
 * the `vector` type and its fmgr functions (src/vector.c), the index build and the access-method handlers
 * (src/hnsw.c:263-335, src/ivfflat.c) stay pgvector's own files, compiled unchanged.  Only
 *
 *     pgvector/src/hnswscan.c   ->  pg_shim/vsr_hnswscan.c   (hnswbeginscan / hnswrescan / hnswgettuple / hnswendscan)
 *     pgvector/src/ivfscan.c    ->  pg_shim/vsr_ivfscan.c    (ivfflatbeginscan / ... / ivfflatendscan)
 *
 * are replaced, plus pg_shim/vsr_pg.c (per-backend GPU context, corpus cache, RBAC tables, GUCs).  The Makefile builds
 * the result as `vector.so` with PGXS when `pg_config` exists.  It is NOT compiled in the authoring image (no
 * postgres.h there); the GPU library underneath is, and is what tests/ exercise.
 *
 * Single-pair operators (`SELECT a <-> b`, the Sort above a seq scan) keep calling pgvector's vector.c: one GPU launch
 * per pair would be slower than the 128-float loop, and libvsrbac deliberately contains no CPU arithmetic (its header
 * promises there is no CPU path).  Batched pair distances exist as vsr_pair_distances for callers that have many.
 */
#ifndef VSR_PG_H
#define VSR_PG_H

#include "postgres.h"

#include "access/genam.h"
#include "access/relscan.h"
#include "storage/itemptr.h"
#include "utils/rel.h"

#include "vsrbac.h"
#include "vsr_sidecar.h"

/* how the current user's permissions are applied (GUC vsrbac.mode) */
typedef enum
{
	VSR_PG_MODE_OFF,			/* no permission filter: the executor's RLS qual filters afterwards, as with stock pgvector */
	VSR_PG_MODE_PREFILTER,		/* VSR_FILTER_RANGES: only the user's rows are read (role / partition tables) */
	VSR_PG_MODE_POSTFILTER		/* VSR_FILTER_BITMAP: the RLS predicate as a per-row bit in the distance loop */
}			VsrPgMode;

extern int	vsr_pg_device;		/* GUC vsrbac.device */
extern int	vsr_pg_mode;		/* GUC vsrbac.mode */
extern bool vsr_pg_index_faithful;	/* GUC vsrbac.index_faithful: reproduce the index's own answer (vsr_indexload.c) */
extern bool vsr_pg_predicate_aware;	/* GUC vsrbac.predicate_aware: the in-backend HNSW walk filters while it walks (not through the sidecar) */
extern char *vsr_pg_sidecar;	/* GUC vsrbac.sidecar: socket of the resident GPU process ("" = in-process, per backend) */

/*
 * One resident corpus per index relation.  In-process mode: owned by this backend (corpus / graph / ivf).  Sidecar mode
 * (vsrbac.sidecar): owned by the sidecar and shared by every backend; this struct then only remembers its handle.
 * `version` fingerprints what the copy was built from (heap and RBAC tables: relfilenode and block count); a scan that
 * computes another fingerprint, a relcache invalidation of one of those relations (DDL, VACUUM, TRUNCATE, ANALYZE) or a
 * change of vsrbac.epoch drops the copy and rebuilds it under the current snapshot.
 */
typedef struct VsrPgCorpus
{
	Oid			indexoid;
	Oid			heapoid;
	Oid			rbac_oids[2];		/* userroles, permissionassignment (InvalidOid: absent) */
	uint64		version;
	int			epoch;				/* vsrbac.epoch at load time */
	bool		stale;				/* set by the relcache callback */
	uint64		sc_handle;			/* sidecar mode: the sidecar's handle (0: in-process) */
	bool		sc_has_hnsw, sc_has_ivf;
	vsr_corpus *corpus;
	int			dim;
	int64		nrows;
	ItemPointerData *tids;		/* caller row index (vsr_search's out_rows) -> heap TID */
	bool		has_rbac;
	vsr_hnsw   *graph;			/* vsrbac.index_faithful: pgvector's own graph / lists, loaded on first use */
	vsr_ivf    *ivf;
}			VsrPgCorpus;

/* scan state shared by the two access methods: the first gettuple runs the whole search, later calls pop */
typedef struct VsrPgScanOpaqueData
{
	VsrPgCorpus *pc;
	bool		first;
	int			nresults;
	int			next;
	int			probes_used;		/* ivfflat iterative scan: lists probed so far */
	ItemPointerData *result_tids;
	MemoryContext tmpCtx;
}			VsrPgScanOpaqueData;
typedef VsrPgScanOpaqueData *VsrPgScanOpaque;

/* vsr_pg.c */
extern void VsrCheck(int status);	/* ereport(ERROR) with vsr_last_error(); keeps pgvector's dimension text */
extern vsr_ctx *VsrBackendContext(void);	/* opened lazily, after fork, once per backend */
extern VsrPgCorpus *VsrCorpusForIndex(Relation index);	/* heap scan + vsr_corpus_load + RBAC tables on first use */
extern int	VsrMetricOf(Relation index);	/* opclass distance proc -> VSR_METRIC_* */
extern int32 VsrCurrentUserId(void);	/* current_user::int, the reference's RLS convention */
extern vsr_filter *VsrFilterForCurrentUser(VsrPgCorpus * pc);	/* NULL when vsrbac.mode = off or no RBAC tables */
extern void VsrRunSearch(IndexScanDesc scan, VsrPgScanOpaque so, int k_hint);	/* fills so->result_tids */
extern void VsrPgInit(void);	/* GUCs + invalidation callback; called by _PG_init (vsr_init.c) */
extern vsr_sc_conn *VsrSidecar(void);	/* connection of this backend to the sidecar, or NULL in in-process mode */
extern bool VsrNextTuple(IndexScanDesc scan, VsrPgScanOpaque so);

/* vsr_indexload.c */
extern vsr_hnsw *VsrLoadHnswGraph(Relation index, VsrPgCorpus * pc);	/* index pages -> vsr_hnsw_load */
extern vsr_ivf *VsrLoadIvfLists(Relation index, VsrPgCorpus * pc);	/* list pages -> vsr_ivf_load */
extern void VsrRunIndexSearch(IndexScanDesc scan, VsrPgScanOpaque so, bool is_hnsw, int ef_or_probes);

#endif							/* VSR_PG_H */
