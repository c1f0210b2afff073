/*
 * vsr_ivfscan.c — replaces pgvector/src/ivfscan.c (file:line below refer to it).
 *
 * ivfflatgettuple's first call (:339-372) picks the `ivfflat.probes` nearest lists (GetScanLists, :36-107), scans them
 * exhaustively into a tuplesort (GetScanItems, :112-176) and then streams TIDs in distance order (:375-388).  Here the
 * first call runs libvsrbac's search and later calls pop.  With an IVF view loaded for the corpus (vsr_ivf_*), the
 * search is restricted to the probed lists exactly like the reference; without one it is the exact scan, i.e. the answer
 * ivfflat converges to with probes = lists.
 */
#include "vsr_pg.h"

#include "ivfflat.h"			/* pgvector's: ivfflat_probes */
#include "utils/memutils.h"

IndexScanDesc
ivfflatbeginscan(Relation index, int nkeys, int norderbys)
{
	IndexScanDesc scan = RelationGetIndexScan(index, nkeys, norderbys);
	VsrPgScanOpaque so = (VsrPgScanOpaque) palloc0(sizeof(VsrPgScanOpaqueData));

	so->pc = VsrCorpusForIndex(index);
	so->first = true;
	so->tmpCtx = AllocSetContextCreate(CurrentMemoryContext, "vsrbac ivfflat scan", ALLOCSET_DEFAULT_SIZES);
	scan->opaque = so;
	return scan;
}

void
ivfflatrescan(IndexScanDesc scan, ScanKey keys, int nkeys, ScanKey orderbys, int norderbys)
{
	VsrPgScanOpaque so = (VsrPgScanOpaque) scan->opaque;

	so->first = true;
	so->nresults = so->next = 0;
	MemoryContextReset(so->tmpCtx);
	if (keys && scan->numberOfKeys > 0)
		memmove(scan->keyData, keys, scan->numberOfKeys * sizeof(ScanKeyData));
	if (orderbys && scan->numberOfOrderBys > 0)
		memmove(scan->orderByData, orderbys, scan->numberOfOrderBys * sizeof(ScanKeyData));
}

bool
ivfflatgettuple(IndexScanDesc scan, ScanDirection dir)
{
	VsrPgScanOpaque so = (VsrPgScanOpaque) scan->opaque;

	Assert(ScanDirectionIsForward(dir));	/* :347 */
	if (so->first)
	{
		if (!IsMVCCSnapshot(scan->xs_snapshot))
			elog(ERROR, "non-MVCC snapshots are not supported with ivfflat");	/* :360-361 */
		/* an ivfflat scan returns every row of the probed lists; the executor's LIMIT stops the popping */
		if (vsr_pg_index_faithful)
			VsrRunIndexSearch(scan, so, false, ivfflat_probes);	/* GetScanLists + GetScanItems over pgvector's own lists */
		else
			VsrRunSearch(scan, so, VSR_MAX_K);
		so->first = false;
		so->probes_used = ivfflat_probes;
	}
	if (VsrNextTuple(scan, so))
		return true;
	if (vsr_pg_index_faithful && ivfflat_iterative_scan != IVFFLAT_ITERATIVE_SCAN_OFF && so->nresults > 0 &&
		so->probes_used < Min(ivfflat_max_probes, IVFFLAT_MAX_LISTS))
	{
		/* ivfflat.iterative_scan (ivfscan.c:292-337): the probed lists ran dry, probe the next batch of lists */
		int			had = so->nresults;

		so->probes_used = Min(2 * so->probes_used, Min(ivfflat_max_probes, IVFFLAT_MAX_LISTS));
		MemoryContextReset(so->tmpCtx);
		VsrRunIndexSearch(scan, so, false, so->probes_used);
		so->next = Min(had, so->nresults);
		return VsrNextTuple(scan, so);
	}
	return false;
}

void
ivfflatendscan(IndexScanDesc scan)
{
	VsrPgScanOpaque so = (VsrPgScanOpaque) scan->opaque;

	MemoryContextDelete(so->tmpCtx);
	pfree(so);
	scan->opaque = NULL;
}
