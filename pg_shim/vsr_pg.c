/*
 * vsr_pg.c — per-backend GPU context, corpus cache, RBAC tables and GUCs of the PostgreSQL shim.
 *
 * Reference behaviour this file stands in for:
 *   pgvector/src/hnsw.c:86-104, ivfflat.c:41-53      GUC definitions (hnsw.ef_search, ivfflat.probes stay pgvector's)
 *   controller/initialize_main_tables.py:17-72       UserRoles(user_id, role_id), PermissionAssignment(role_id, document_id)
 *   controller/baseline/pg_row_security/row_level_security.py:41-69   role per user, policy on current_user::int
 *   controller/initialize_main_tables.py:54-61       documentblocks(block_id, document_id, block_content, vector)
 */
#include "vsr_pg.h"

#include "access/heapam.h"
#include "access/tableam.h"
#include "catalog/index.h"
#include "catalog/namespace.h"
#include "storage/bufmgr.h"
#include "executor/spi.h"
#include "executor/tuptable.h"
#include "miscadmin.h"
#include "storage/ipc.h"
#include "utils/builtins.h"
#include "utils/guc.h"
#include "utils/hsearch.h"
#include "utils/inval.h"
#include "utils/lsyscache.h"
#include "utils/memutils.h"
#include "utils/snapmgr.h"

#include "vector.h"				/* pgvector's: struct Vector, DatumGetVector */

int			vsr_pg_device = 0;
int			vsr_pg_mode = VSR_PG_MODE_POSTFILTER;
bool		vsr_pg_index_faithful = false;
bool		vsr_pg_predicate_aware = false;
char	   *vsr_pg_sidecar = NULL;
static int	vsr_pg_epoch = 0;	/* GUC vsrbac.epoch: any change drops every cached corpus (an explicit refresh) */
static vsr_sc_conn *sidecar_conn = NULL;

static const struct config_enum_entry vsr_pg_mode_options[] = {
	{"off", VSR_PG_MODE_OFF, false},
	{"prefilter", VSR_PG_MODE_PREFILTER, false},
	{"postfilter", VSR_PG_MODE_POSTFILTER, false},
	{NULL, 0, false}
};

static vsr_ctx *backend_ctx = NULL;
static HTAB *corpus_cache = NULL;

void
VsrCheck(int status)
{
	if (status == VSR_OK)
		return;
	/* vector.c:60-67 reports dimension mismatches as ERRCODE_DATA_EXCEPTION with exactly this text */
	ereport(ERROR,
			(errcode(status == VSR_ERR_DIM_MISMATCH ? ERRCODE_DATA_EXCEPTION :
					 status == VSR_ERR_OOM ? ERRCODE_OUT_OF_MEMORY : ERRCODE_EXTERNAL_ROUTINE_EXCEPTION),
			 errmsg("%s", vsr_last_error())));
}

static void
vsr_pg_shutdown(int code, Datum arg)
{
	if (corpus_cache != NULL)
	{
		/* index structures before their corpus, corpora before the context (include/vsrbac.h) */
		HASH_SEQ_STATUS seq;
		VsrPgCorpus *pc;

		hash_seq_init(&seq, corpus_cache);
		while ((pc = (VsrPgCorpus *) hash_seq_search(&seq)) != NULL)
		{
			if (pc->graph != NULL)
				(void) vsr_hnsw_free(pc->graph);
			if (pc->ivf != NULL)
				(void) vsr_ivf_free(pc->ivf);
			if (pc->corpus != NULL)
				(void) vsr_corpus_free(pc->corpus);
		}
		corpus_cache = NULL;
	}
	if (backend_ctx != NULL)
	{
		(void) vsr_close(backend_ctx);
		backend_ctx = NULL;
	}
	if (sidecar_conn != NULL)
	{
		vsr_sc_close(sidecar_conn);	/* the sidecar keeps the corpora: that is what it is for */
		sidecar_conn = NULL;
	}
}

vsr_sc_conn *
VsrSidecar(void)
{
	char		err[256];

	if (vsr_pg_sidecar == NULL || vsr_pg_sidecar[0] == '\0')
		return NULL;
	if (sidecar_conn == NULL)
	{
		sidecar_conn = vsr_sc_connect(vsr_pg_sidecar, err, sizeof(err));
		if (sidecar_conn == NULL)
			ereport(ERROR, (errcode(ERRCODE_CONNECTION_FAILURE), errmsg("%s", err)));
		on_proc_exit(vsr_pg_shutdown, (Datum) 0);
	}
	return sidecar_conn;
}

/* sidecar calls report through the same channel as library calls */
static void
VsrScCheck(int status)
{
	if (status == 0)
		return;
	if (status < 0)
	{
		/* the connection is gone: forget it, the next scan reconnects and finds (or reloads) the corpus */
		char		msg[256];

		snprintf(msg, sizeof(msg), "%s", vsr_sc_error(sidecar_conn));
		vsr_sc_close(sidecar_conn);
		sidecar_conn = NULL;
		ereport(ERROR, (errcode(ERRCODE_CONNECTION_FAILURE), errmsg("%s", msg)));
	}
	ereport(ERROR,
			(errcode(status == VSR_ERR_DIM_MISMATCH ? ERRCODE_DATA_EXCEPTION :
					 status == VSR_ERR_OOM ? ERRCODE_OUT_OF_MEMORY : ERRCODE_EXTERNAL_ROUTINE_EXCEPTION),
			 errmsg("%s", vsr_sc_error(sidecar_conn))));
}

/* drop what this backend holds of a cached corpus (the entry stays, marked empty) */
static void
vsr_pg_forget(VsrPgCorpus * pc)
{
	if (pc->graph != NULL)
		(void) vsr_hnsw_free(pc->graph);
	if (pc->ivf != NULL)
		(void) vsr_ivf_free(pc->ivf);
	if (pc->corpus != NULL)
		(void) vsr_corpus_free(pc->corpus);
	if (pc->tids != NULL)
		pfree(pc->tids);
	pc->graph = NULL;
	pc->ivf = NULL;
	pc->corpus = NULL;
	pc->tids = NULL;
	pc->sc_handle = 0;
	pc->sc_has_hnsw = pc->sc_has_ivf = false;
	pc->stale = false;
}

/*
 * Relcache invalidation (DDL, VACUUM, TRUNCATE, ANALYZE, CREATE INDEX on the heap or on an RBAC table): the cached copy
 * may no longer match; it is rebuilt by the next scan.  Plain INSERT / UPDATE / DELETE send no relcache invalidation:
 * those are caught by the block-count half of the version fingerprint when they extend a relation, and otherwise by
 * vsrbac.epoch ("Freshness").
 */
static void
vsr_pg_relcache_cb(Datum arg, Oid relid)
{
	HASH_SEQ_STATUS seq;
	VsrPgCorpus *pc;

	(void) arg;
	if (corpus_cache == NULL)
		return;
	hash_seq_init(&seq, corpus_cache);
	while ((pc = (VsrPgCorpus *) hash_seq_search(&seq)) != NULL)
		if (relid == InvalidOid || relid == pc->indexoid || relid == pc->heapoid || relid == pc->rbac_oids[0] ||
			relid == pc->rbac_oids[1])
			pc->stale = true;
}

/* what a cached copy was built from: relfilenode and size of the heap and of the two RBAC tables */
static uint64
vsr_pg_fingerprint(Relation heap, Oid *rbac_oids)
{
	uint64		h = UINT64CONST(0xcbf29ce484222325);
	const char *names[2] = {"userroles", "permissionassignment"};

#define MIX(v) do { h ^= (uint64) (v); h *= UINT64CONST(0x100000001b3); } while (0)
	MIX(RelationGetRelid(heap));
	MIX(heap->rd_rel->relfilenode);
	MIX(RelationGetNumberOfBlocks(heap));
	for (int i = 0; i < 2; i++)
	{
		Oid			oid = RelnameGetRelid(names[i]);

		rbac_oids[i] = oid;
		if (OidIsValid(oid))
		{
			Relation	r = table_open(oid, AccessShareLock);

			MIX(r->rd_rel->relfilenode);
			MIX(RelationGetNumberOfBlocks(r));
			table_close(r, AccessShareLock);
		}
		else
			MIX(0);
	}
#undef MIX
	return h;
}

vsr_ctx *
VsrBackendContext(void)
{
	/* never inherited across fork: a backend opens its own HIP context on first use */
	if (backend_ctx == NULL)
	{
		VsrCheck(vsr_open(vsr_pg_device, &backend_ctx));
		on_proc_exit(vsr_pg_shutdown, (Datum) 0);
	}
	return backend_ctx;
}

int
VsrMetricOf(Relation index)
{
	/* the opclass's distance support function (proc 1) names the operator family: vector.sql:292-333 */
	Oid			procid = index_getprocid(index, 1, 1);
	char	   *name = get_func_name(procid);

	if (strcmp(name, "vector_l2_squared_distance") == 0)
		return VSR_METRIC_L2;
	if (strcmp(name, "vector_negative_inner_product") == 0)
		return index->rd_support != NULL && OidIsValid(index_getprocid(index, 1, 2)) ? VSR_METRIC_COSINE : VSR_METRIC_IP;
	if (strcmp(name, "l1_distance") == 0 || strcmp(name, "vector_l1_distance") == 0)
		return VSR_METRIC_L1;
	ereport(ERROR, (errmsg("vsrbac: unsupported distance function %s", name)));
	return VSR_METRIC_L2;		/* not reached */
}

int32
VsrCurrentUserId(void)
{
	/* the reference creates one PostgreSQL role per user, named by the user id (row_level_security.py:41-52) */
	char	   *name = GetUserNameFromId(GetUserId(), false);
	char	   *end;
	long		v = strtol(name, &end, 10);

	return (*end == '\0' && end != name) ? (int32) v : -1;	/* not a numeric role: sees nothing under RBAC */
}

/* attribute number of a column of the heap, or 0 */
static AttrNumber
heap_attno(Relation heap, const char *name)
{
	TupleDesc	desc = RelationGetDescr(heap);

	for (int i = 0; i < desc->natts; i++)
		if (strcmp(NameStr(TupleDescAttr(desc, i)->attname), name) == 0)
			return (AttrNumber) (i + 1);
	return 0;
}

/* one int4 pair table through SPI: UserRoles / PermissionAssignment */
static int64
load_pairs(const char *sql, int32 **a, int32 **b, MemoryContext keep)
{
	int64		n = 0;

	if (SPI_execute(sql, true, 0) != SPI_OK_SELECT)
		return -1;				/* table absent: no RBAC */
	n = (int64) SPI_processed;
	*a = MemoryContextAlloc(keep, sizeof(int32) * Max(n, 1));
	*b = MemoryContextAlloc(keep, sizeof(int32) * Max(n, 1));
	for (int64 i = 0; i < n; i++)
	{
		bool		isnull;

		(*a)[i] = DatumGetInt32(SPI_getbinval(SPI_tuptable->vals[i], SPI_tuptable->tupdesc, 1, &isnull));
		(*b)[i] = DatumGetInt32(SPI_getbinval(SPI_tuptable->vals[i], SPI_tuptable->tupdesc, 2, &isnull));
	}
	return n;
}

VsrPgCorpus *
VsrCorpusForIndex(Relation index)
{
	Oid			indexoid = RelationGetRelid(index);
	bool		found;
	VsrPgCorpus *pc;
	Relation	heap;
	Oid			rbac_oids[2];
	uint64		version;
	vsr_sc_conn *sc = VsrSidecar();
	const uint64 sc_key = ((uint64) MyDatabaseId << 32) | (uint64) indexoid;

	if (corpus_cache == NULL)
	{
		HASHCTL		hc;

		memset(&hc, 0, sizeof(hc));
		hc.keysize = sizeof(Oid);
		hc.entrysize = sizeof(VsrPgCorpus);
		hc.hcxt = TopMemoryContext;
		corpus_cache = hash_create("vsrbac corpora", 16, &hc, HASH_ELEM | HASH_BLOBS | HASH_CONTEXT);
	}
	pc = hash_search(corpus_cache, &indexoid, HASH_ENTER, &found);
	if (!found)
	{
		memset(((char *) pc) + sizeof(Oid), 0, sizeof(VsrPgCorpus) - sizeof(Oid));
		pc->indexoid = indexoid;
	}

	/* is the copy we (or the sidecar) hold still the table the current snapshot sees? */
	heap = table_open(index->rd_index->indrelid, AccessShareLock);
	version = vsr_pg_fingerprint(heap, rbac_oids);
	if (found && (pc->corpus != NULL || pc->sc_handle != 0) && !pc->stale && pc->version == version && pc->epoch == vsr_pg_epoch)
	{
		table_close(heap, AccessShareLock);
		return pc;
	}
	vsr_pg_forget(pc);
	pc->heapoid = RelationGetRelid(heap);
	pc->rbac_oids[0] = rbac_oids[0];
	pc->rbac_oids[1] = rbac_oids[1];
	pc->version = version ^ (uint64) vsr_pg_epoch;
	pc->epoch = vsr_pg_epoch;

	/* sidecar mode: another backend may have loaded this very version already */
	if (sc != NULL)
	{
		vsr_sc_corpus_info info;
		int			rc = vsr_sc_corpus_lookup(sc, sc_key, pc->version, &info);

		if (rc == 0)
		{
			pc->sc_handle = info.handle;
			pc->dim = info.dim;
			pc->nrows = info.nrows;
			pc->has_rbac = info.has_rbac != 0;
			pc->sc_has_hnsw = info.has_hnsw != 0;
			pc->sc_has_ivf = info.has_ivf != 0;
		}
		else if (rc != VSR_SC_NOTFOUND)
			VsrScCheck(rc);
	}

	/*
	 * The heap in scan order under the active snapshot: the TIDs are needed by every backend (row index -> heap TID), the
	 * vectors only when somebody has to load the corpus.
	 */
	{
		AttrNumber	vec_att = index->rd_index->indkey.values[0];
		AttrNumber	blk_att = heap_attno(heap, "block_id");
		AttrNumber	doc_att = heap_attno(heap, "document_id");
		TableScanDesc hs = table_beginscan(heap, GetActiveSnapshot(), 0, NULL);
		TupleTableSlot *slot = table_slot_create(heap, NULL);
		const bool	need_rows = sc == NULL || pc->sc_handle == 0;
		int64		cap = 1 << 16,
					n = 0;
		int			dim = 0;
		float	   *rows = NULL;
		int64	   *blk = palloc(sizeof(int64) * cap);
		int32	   *doc = palloc(sizeof(int32) * cap);
		ItemPointerData *tids = MemoryContextAlloc(TopMemoryContext, sizeof(ItemPointerData) * cap);

		while (table_scan_getnextslot(hs, ForwardScanDirection, slot))
		{
			bool		isnull;
			Datum		d = slot_getattr(slot, vec_att, &isnull);
			Vector	   *v;

			if (isnull)
				continue;		/* NULL vectors are not indexed (hnswbuild.c:473-480) */
			v = DatumGetVector(d);
			if (dim == 0)
			{
				dim = v->dim;
				if (need_rows)
					rows = MemoryContextAllocHuge(CurrentMemoryContext, sizeof(float) * (Size) cap * dim);
			}
			if (n == cap)
			{
				cap *= 2;
				if (need_rows)
					rows = repalloc_huge(rows, sizeof(float) * (Size) cap * dim);
				blk = repalloc_huge(blk, sizeof(int64) * cap);
				doc = repalloc_huge(doc, sizeof(int32) * cap);
				tids = repalloc_huge(tids, sizeof(ItemPointerData) * cap);
			}
			if (need_rows)
				memcpy(rows + (Size) n * dim, v->x, sizeof(float) * dim);
			blk[n] = blk_att ? DatumGetInt64(slot_getattr(slot, blk_att, &isnull)) : n;
			doc[n] = doc_att ? DatumGetInt32(slot_getattr(slot, doc_att, &isnull)) : 0;
			tids[n] = slot->tts_tid;
			n++;
		}
		ExecDropSingleTupleTableSlot(slot);
		table_endscan(hs);
		table_close(heap, AccessShareLock);

		pc->tids = tids;
		if (pc->sc_handle != 0 && n != pc->nrows)
		{
			/* the resident copy was loaded under another snapshot of the same files: replace it */
			VsrScCheck(vsr_sc_corpus_drop(sc, sc_key));
			vsr_pg_forget(pc);
			pfree(blk);
			pfree(doc);
			return VsrCorpusForIndex(index);
		}
		pc->dim = dim;
		pc->nrows = n;
		if (need_rows)
		{
			pc->has_rbac = false;
			if (sc != NULL)
			{
				vsr_sc_corpus_info info;

				VsrScCheck(vsr_sc_corpus_load(sc, sc_key, pc->version, rows, n, dim > 0 ? dim : 1, blk_att ? blk : NULL,
											  doc_att ? doc : NULL, &info));
				pc->sc_handle = info.handle;
			}
			else
				VsrCheck(vsr_corpus_load(VsrBackendContext(), rows, n, dim > 0 ? dim : 1, blk_att ? blk : NULL, doc_att ? doc : NULL,
										 0, &pc->corpus));
			if (rows)
				pfree(rows);
		}
		pfree(blk);
		pfree(doc);
	}

	/* RBAC tables, when the schema of the reference is present and the heap carries document ids */
	if (!pc->has_rbac && OidIsValid(pc->rbac_oids[0]) && OidIsValid(pc->rbac_oids[1]) && SPI_connect() == SPI_OK_CONNECT)
	{
		int32	   *uu = NULL, *ur = NULL, *pr = NULL, *pd = NULL;
		int64		n_ur = load_pairs("SELECT user_id, role_id FROM userroles", &uu, &ur, CurrentMemoryContext);
		int64		n_pa = n_ur >= 0 ? load_pairs("SELECT role_id, document_id FROM permissionassignment", &pr, &pd, CurrentMemoryContext) : -1;

		if (n_ur >= 0 && n_pa >= 0)
		{
			if (sc != NULL)
				VsrScCheck(vsr_sc_rbac_load(sc, pc->sc_handle, uu, ur, n_ur, pr, pd, n_pa));
			else
				VsrCheck(vsr_rbac_load(pc->corpus, uu, ur, n_ur, pr, pd, n_pa));
			pc->has_rbac = true;
		}
		SPI_finish();
	}
	return pc;
}

vsr_filter *
VsrFilterForCurrentUser(VsrPgCorpus * pc)
{
	vsr_filter *f = NULL;

	if (vsr_pg_mode == VSR_PG_MODE_OFF || !pc->has_rbac || superuser())
		return NULL;
	VsrCheck(vsr_filter_for_user(pc->corpus, VsrCurrentUserId(),
								 vsr_pg_mode == VSR_PG_MODE_PREFILTER ? VSR_FILTER_RANGES : VSR_FILTER_BITMAP, &f));
	return f;
}

/*
 * The whole search, run by the first gettuple of a scan.  k_hint = hnsw.ef_search (hnswscan.c:44) or the number of rows
 * the probed lists hold; the executor's LIMIT pops at most that many.  The result is exact filtered top-k_hint.
 */
void
VsrRunSearch(IndexScanDesc scan, VsrPgScanOpaque so, int k_hint)
{
	VsrPgCorpus *pc = so->pc;
	Vector	   *q;
	const vsr_filter *filter;
	int			k = Min(Max(k_hint, 1), VSR_MAX_K);
	MemoryContext old = MemoryContextSwitchTo(so->tmpCtx);
	int64	   *blk = palloc(sizeof(int64) * k);
	int64	   *rowidx = palloc(sizeof(int64) * k);
	float	   *dist = palloc(sizeof(float) * k);
	int32		count = 0;

	if (scan->orderByData == NULL)
		elog(ERROR, "cannot scan index without order");	/* hnswscan.c:196-197 */
	if (scan->orderByData->sk_flags & SK_ISNULL)
	{
		so->nresults = 0;		/* ORDER BY vec <-> NULL: no rows from the index (hnswscan.c:84-89 returns a zero vector scan) */
		MemoryContextSwitchTo(old);
		return;
	}
	q = DatumGetVector(scan->orderByData->sk_argument);
	if (q->dim != pc->dim)
		ereport(ERROR, (errcode(ERRCODE_DATA_EXCEPTION),
						errmsg("different vector dimensions %d and %d", pc->dim, q->dim)));
	if (pc->sc_handle != 0)
	{
		vsr_sc_search_req req;

		memset(&req, 0, sizeof(req));
		req.handle = pc->sc_handle;
		req.nq = 1;
		req.dim = q->dim;
		req.k = k;
		req.metric = VsrMetricOf(scan->indexRelation);
		req.filter_mode = (vsr_pg_mode == VSR_PG_MODE_OFF || !pc->has_rbac || superuser()) ? -1 :
			vsr_pg_mode == VSR_PG_MODE_PREFILTER ? VSR_FILTER_RANGES : VSR_FILTER_BITMAP;
		req.user_id = VsrCurrentUserId();
		VsrScCheck(vsr_sc_search(VsrSidecar(), &req, q->x, &count, rowidx, blk, dist));
	}
	else
	{
		filter = VsrFilterForCurrentUser(pc);
		VsrCheck(vsr_search(pc->corpus, q->x, 1, q->dim, k, VsrMetricOf(scan->indexRelation), filter ? &filter : NULL,
							blk, NULL, rowidx, dist, &count));
	}
	so->result_tids = palloc(sizeof(ItemPointerData) * Max(count, 1));
	for (int i = 0; i < count; i++)
		so->result_tids[i] = pc->tids[rowidx[i]];
	so->nresults = count;
	so->next = 0;
	MemoryContextSwitchTo(old);
}

bool
VsrNextTuple(IndexScanDesc scan, VsrPgScanOpaque so)
{
	if (so->next >= so->nresults)
		return false;
	scan->xs_heaptid = so->result_tids[so->next++];
	scan->xs_recheck = false;	/* exact distances: hnswscan.c:308-310 */
	scan->xs_recheckorderby = false;
	return true;
}

void
VsrPgInit(void)
{
	DefineCustomIntVariable("vsrbac.device", "HIP device ordinal of the MI355X this backend searches on", NULL,
							&vsr_pg_device, 0, 0, 63, PGC_USERSET, 0, NULL, NULL, NULL);
	DefineCustomEnumVariable("vsrbac.mode", "How the current user's RBAC permissions are applied to index scans",
							 "off: unfiltered (RLS quals filter afterwards); prefilter: read only permitted rows; "
							 "postfilter: per-row permission bit in the distance loop", &vsr_pg_mode,
							 VSR_PG_MODE_POSTFILTER, vsr_pg_mode_options, PGC_USERSET, 0, NULL, NULL, NULL);
	DefineCustomBoolVariable("vsrbac.index_faithful",
							 "Answer index scans with the index's own graph walk / list probe on the GPU (same candidates and "
							 "recall as stock pgvector) instead of the exact filtered search", NULL, &vsr_pg_index_faithful,
							 false, PGC_USERSET, 0, NULL, NULL, NULL);
	DefineCustomBoolVariable("vsrbac.predicate_aware",
							 "With vsrbac.index_faithful: the HNSW walk applies the current user's permissions while it walks "
							 "(ACORN-style two-hop expansion) instead of filtering what the index returns", NULL,
							 &vsr_pg_predicate_aware, false, PGC_USERSET, 0, NULL, NULL, NULL);
	DefineCustomStringVariable("vsrbac.sidecar",
							   "UNIX socket of the resident GPU process (pg_shim/vsr_sidecar); empty: every backend loads its own copy",
							   "With a sidecar the corpus survives the backend: a new connection per search, as the reference harness "
							   "makes them, costs a socket connect instead of a corpus load", &vsr_pg_sidecar, "", PGC_USERSET, 0,
							   NULL, NULL, NULL);
	DefineCustomIntVariable("vsrbac.epoch", "Changing this value drops every cached corpus of this backend (and, through the "
							"version, of the sidecar): an explicit refresh after INSERT / UPDATE / DELETE", NULL,
							&vsr_pg_epoch, 0, 0, INT_MAX, PGC_USERSET, 0, NULL, NULL, NULL);
	MarkGUCPrefixReserved("vsrbac");
	CacheRegisterRelcacheCallback(vsr_pg_relcache_cb, (Datum) 0);
}
