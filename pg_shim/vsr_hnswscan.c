/*
 * vsr_hnswscan.c — replaces pgvector/src/hnswscan.c (file:line below refer to it).
 *
 * Same four entry points, same contract towards the executor: hnswgettuple's first call runs the whole search
 * (GetScanItems, :15-45, :206-218) and every call hands out one heap TID in distance order (:278-311) with
 * xs_recheck = xs_recheckorderby = false.  The search itself is libvsrbac's exact filtered k-NN over the resident
 * corpus instead of HnswSearchLayer over index pages; hnsw.ef_search keeps its meaning as the number of candidates the
 * scan can return (:44).  With hnsw.iterative_scan (:227-276) a scan that runs dry asks for twice as many.
 */
#include "vsr_pg.h"

#include "hnsw.h"				/* pgvector's: hnsw_ef_search, hnsw_iterative_scan, HNSW_SCAN_LOCK */
#include "storage/lmgr.h"
#include "utils/memutils.h"

IndexScanDesc
hnswbeginscan(Relation index, int nkeys, int norderbys)
{
	IndexScanDesc scan = RelationGetIndexScan(index, nkeys, norderbys);
	VsrPgScanOpaque so = (VsrPgScanOpaque) palloc0(sizeof(VsrPgScanOpaqueData));

	so->pc = VsrCorpusForIndex(index);
	so->first = true;
	so->tmpCtx = AllocSetContextCreate(CurrentMemoryContext, "vsrbac hnsw scan", ALLOCSET_DEFAULT_SIZES);	/* :138-140 */
	scan->opaque = so;
	return scan;
}

void
hnswrescan(IndexScanDesc scan, ScanKey keys, int nkeys, ScanKey orderbys, int norderbys)
{
	VsrPgScanOpaque so = (VsrPgScanOpaque) scan->opaque;

	so->first = true;
	so->nresults = so->next = 0;
	MemoryContextReset(so->tmpCtx);	/* :166 */
	if (keys && scan->numberOfKeys > 0)
		memmove(scan->keyData, keys, scan->numberOfKeys * sizeof(ScanKeyData));
	if (orderbys && scan->numberOfOrderBys > 0)
		memmove(scan->orderByData, orderbys, scan->numberOfOrderBys * sizeof(ScanKeyData));
}

bool
hnswgettuple(IndexScanDesc scan, ScanDirection dir)
{
	VsrPgScanOpaque so = (VsrPgScanOpaque) scan->opaque;

	Assert(ScanDirectionIsForward(dir));	/* :187 */
	if (so->first)
	{
		if (!IsMVCCSnapshot(scan->xs_snapshot))
			elog(ERROR, "non-MVCC snapshots are not supported with hnsw");	/* :203-204 */
		/* the share lock pgvector takes around its page walk (:213-218) keeps vacuum's ordering with scans */
		LockPage(scan->indexRelation, HNSW_SCAN_LOCK, ShareLock);
		if (vsr_pg_index_faithful)
			VsrRunIndexSearch(scan, so, true, hnsw_ef_search);	/* HnswSearchLayer over pgvector's own graph, on the GPU */
		else
			VsrRunSearch(scan, so, hnsw_ef_search);
		UnlockPage(scan->indexRelation, HNSW_SCAN_LOCK, ShareLock);
		so->first = false;
	}
	if (VsrNextTuple(scan, so))
		return true;
	if (hnsw_iterative_scan != HNSW_ITERATIVE_SCAN_OFF && so->nresults > 0 && so->nresults < VSR_MAX_K)
	{
		/*
		 * Iterative scan: the filter above the index discarded rows and the executor wants more (:227-276).  pgvector
		 * resumes its graph walk from the candidates it had discarded; here the search is re-run with twice the width
		 * and what was handed out already is skipped -- the exact search returns a prefix-stable total order, and the
		 * index-faithful walk with a wider beam contains the narrower beam's answer up to the reference's own
		 * "relaxed order" tolerance (hnsw.iterative_scan = relaxed_order, hnsw.c:90-92).  Bounded by
		 * hnsw.max_scan_tuples (:232-236) and by HNSW_MAX_EF_SEARCH.
		 */
		int			had = so->nresults;
		int			wider = Min(2 * had, VSR_MAX_K);

		if (had >= hnsw_max_scan_tuples)
			return false;
		MemoryContextReset(so->tmpCtx);
		if (vsr_pg_index_faithful)
			VsrRunIndexSearch(scan, so, true, Min(wider, HNSW_MAX_EF_SEARCH));
		else
			VsrRunSearch(scan, so, wider);
		so->next = Min(had, so->nresults);
		return VsrNextTuple(scan, so);
	}
	return false;
}

void
hnswendscan(IndexScanDesc scan)
{
	VsrPgScanOpaque so = (VsrPgScanOpaque) scan->opaque;

	MemoryContextDelete(so->tmpCtx);
	pfree(so);
	scan->opaque = NULL;
}
