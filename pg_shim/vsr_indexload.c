/*
 * vsr_indexload.c — index-faithful scans: pgvector's OWN on-disk index structure handed to libvsrbac.
 *
 * By default the shim answers an ordered scan with the exact filtered search (vsr_pg.c, VsrRunSearch).  With
 * vsrbac.index_faithful = on the scan reproduces the index's answer instead -- the same graph walk / list probe as
 * stock pgvector, same candidates, same recall -- on the GPU:
 *
 *   hnsw     the element and neighbour tuples of the index pages (hnsw.h:304-350; written by hnswbuild.c, read by
 *            hnswutils.c:751-780 HnswLoadNeighborTids) -> flat arrays -> vsr_hnsw_load; the scan is vsr_hnsw_search
 *   ivfflat  the list pages (ivfflat.h:222-248: centre + start page per list) and the TIDs of every list's entry pages
 *            -> centres + row -> list map -> vsr_ivf_load; the scan is vsr_ivf_search
 *
 * Like the rest of pg_shim/ this file is written against postgres.h + pgvector's headers and is NOT compiled in the
 * authoring image.  The arrays it builds are exactly what tests/test_gpu_index.py feeds the same entry points from the
 * CPU restatement of pgvector's build (oracle/vsr_index_oracle.c).
 */
#include "vsr_pg.h"

#include "access/generic_xlog.h"
#include "hnsw.h"
#include "ivfflat.h"
#include "storage/bufmgr.h"
#include "utils/hsearch.h"
#include "utils/memutils.h"

/* heap TID -> row of the resident corpus (the inverse of VsrPgCorpus::tids) */
typedef struct VsrTidRow
{
	ItemPointerData tid;		/* key */
	int64		row;
}			VsrTidRow;

static HTAB *
vsr_pg_tid_rows(VsrPgCorpus * pc)
{
	HASHCTL		ctl;
	HTAB	   *h;

	memset(&ctl, 0, sizeof(ctl));
	ctl.keysize = sizeof(ItemPointerData);
	ctl.entrysize = sizeof(VsrTidRow);
	ctl.hcxt = CurrentMemoryContext;
	h = hash_create("vsrbac tid -> row", pc->nrows, &ctl, HASH_ELEM | HASH_BLOBS | HASH_CONTEXT);
	for (int64 r = 0; r < pc->nrows; r++)
	{
		bool		found;
		VsrTidRow  *e = (VsrTidRow *) hash_search(h, &pc->tids[r], HASH_ENTER, &found);

		e->row = r;
	}
	return h;
}

/* index tuple position -> element number, in page order (= build order: hnswbuild.c writes elements in insertion order) */
typedef struct VsrElemNo
{
	ItemPointerData at;			/* key: (block, offset) of the element tuple */
	int32		no;
}			VsrElemNo;

/*
 * The HNSW graph of `index` as vsr_hnsw_load takes it.  Two passes over the element pages: number the live elements,
 * then translate every neighbour TID into an element number.
 */
vsr_hnsw *
VsrLoadHnswGraph(Relation index, VsrPgCorpus * pc)
{
	Buffer		buf;
	Page		page;
	HnswMetaPage meta;
	int			m,
				entry_level;
	BlockNumber entry_blk;
	OffsetNumber entry_off;
	HASHCTL		ctl;
	HTAB	   *elems,
			   *tidrows = vsr_pg_tid_rows(pc);
	int32		n_elem = 0,
				n_upper = 0,
				max_level = 1,
				entry = -1;
	int32	   *level,
			   *nbr0,
			   *tid_count,
			   *up_slot,
			   *up_nbr;
	int64	   *tids;
	vsr_hnsw   *graph = NULL;

	buf = ReadBuffer(index, HNSW_METAPAGE_BLKNO);
	LockBuffer(buf, BUFFER_LOCK_SHARE);
	meta = HnswPageGetMeta(BufferGetPage(buf));
	m = meta->m;
	entry_blk = meta->entryBlkno;
	entry_off = meta->entryOffno;
	entry_level = meta->entryLevel;
	UnlockReleaseBuffer(buf);
	if (!BlockNumberIsValid(entry_blk))
		return NULL;			/* empty index */

	memset(&ctl, 0, sizeof(ctl));
	ctl.keysize = sizeof(ItemPointerData);
	ctl.entrysize = sizeof(VsrElemNo);
	ctl.hcxt = CurrentMemoryContext;
	elems = hash_create("vsrbac hnsw elements", Max(pc->nrows, 16), &ctl, HASH_ELEM | HASH_BLOBS | HASH_CONTEXT);

	/* pass 1: number the elements, note the highest level */
	for (BlockNumber blk = HNSW_HEAD_BLKNO; BlockNumberIsValid(blk);)
	{
		OffsetNumber maxoff;

		buf = ReadBuffer(index, blk);
		LockBuffer(buf, BUFFER_LOCK_SHARE);
		page = BufferGetPage(buf);
		maxoff = PageGetMaxOffsetNumber(page);
		for (OffsetNumber off = FirstOffsetNumber; off <= maxoff; off = OffsetNumberNext(off))
		{
			HnswElementTuple etup = (HnswElementTuple) PageGetItem(page, PageGetItemId(page, off));
			VsrElemNo  *e;
			ItemPointerData at;
			bool		found;

			if (!HnswIsElementTuple(etup) || etup->deleted)
				continue;
			ItemPointerSet(&at, blk, off);
			e = (VsrElemNo *) hash_search(elems, &at, HASH_ENTER, &found);
			e->no = n_elem++;
			if (etup->level >= 1)
				n_upper++;
			max_level = Max(max_level, (int32) etup->level);
		}
		blk = HnswPageGetOpaque(page)->nextblkno;
		UnlockReleaseBuffer(buf);
	}
	max_level = Max(max_level, entry_level);

	level = (int32 *) palloc0(sizeof(int32) * Max(n_elem, 1));
	tid_count = (int32 *) palloc0(sizeof(int32) * Max(n_elem, 1));
	up_slot = (int32 *) palloc(sizeof(int32) * Max(n_elem, 1));
	nbr0 = (int32 *) palloc(sizeof(int32) * (Size) Max(n_elem, 1) * 2 * m);
	tids = (int64 *) palloc(sizeof(int64) * (Size) Max(n_elem, 1) * HNSW_HEAPTIDS);
	up_nbr = (int32 *) palloc(sizeof(int32) * (Size) Max(n_upper, 1) * max_level * m);
	memset(nbr0, 0xFF, sizeof(int32) * (Size) Max(n_elem, 1) * 2 * m);	/* -1 padded */
	memset(up_nbr, 0xFF, sizeof(int32) * (Size) Max(n_upper, 1) * max_level * m);
	for (int32 i = 0; i < n_elem; i++)
		up_slot[i] = -1;

	/* pass 2: heap TIDs -> corpus rows, neighbour TIDs -> element numbers */
	n_upper = 0;
	for (BlockNumber blk = HNSW_HEAD_BLKNO; BlockNumberIsValid(blk);)
	{
		OffsetNumber maxoff;

		buf = ReadBuffer(index, blk);
		LockBuffer(buf, BUFFER_LOCK_SHARE);
		page = BufferGetPage(buf);
		maxoff = PageGetMaxOffsetNumber(page);
		for (OffsetNumber off = FirstOffsetNumber; off <= maxoff; off = OffsetNumberNext(off))
		{
			HnswElementTuple etup = (HnswElementTuple) PageGetItem(page, PageGetItemId(page, off));
			ItemPointerData at;
			VsrElemNo  *me;
			Buffer		nbuf;
			Page		npage;
			HnswNeighborTuple ntup;
			int32		e;

			if (!HnswIsElementTuple(etup) || etup->deleted)
				continue;
			ItemPointerSet(&at, blk, off);
			me = (VsrElemNo *) hash_search(elems, &at, HASH_FIND, NULL);
			e = me->no;
			level[e] = etup->level;
			if (ItemPointerGetBlockNumber(&at) == entry_blk && ItemPointerGetOffsetNumber(&at) == entry_off)
				entry = e;
			/* newest TID first, as GetScanItems hands them out (hnswscan.c:278-311); -1 padded */
			for (int j = 0; j < HNSW_HEAPTIDS; j++)
			{
				tids[(Size) e * HNSW_HEAPTIDS + j] = -1;
				if (ItemPointerIsValid(&etup->heaptids[j]))
				{
					VsrTidRow  *tr = (VsrTidRow *) hash_search(tidrows, &etup->heaptids[j], HASH_FIND, NULL);

					if (tr != NULL)
						tids[(Size) e * HNSW_HEAPTIDS + tid_count[e]++] = tr->row;
				}
			}
			if (etup->level >= 1)
				up_slot[e] = n_upper++;

			/* neighbour tuple: (level + 2) * m index TIDs, layer lc starts at (level - lc) * m (hnswutils.c:775) */
			nbuf = ReadBuffer(index, ItemPointerGetBlockNumber(&etup->neighbortid));
			LockBuffer(nbuf, BUFFER_LOCK_SHARE);
			npage = BufferGetPage(nbuf);
			ntup = (HnswNeighborTuple) PageGetItem(npage, PageGetItemId(npage, ItemPointerGetOffsetNumber(&etup->neighbortid)));
			if (ntup->version == etup->version && ntup->count == (etup->level + 2) * m)
			{
				for (int lc = etup->level; lc >= 0; lc--)
				{
					int			lm = HnswGetLayerM(m, lc);
					int			start = (etup->level - lc) * m;
					int32	   *dst = lc == 0 ? nbr0 + (Size) e * 2 * m
						: up_nbr + ((Size) up_slot[e] * max_level + (lc - 1)) * m;

					for (int i = 0; i < lm; i++)
					{
						ItemPointer nt = &ntup->indextids[start + i];
						VsrElemNo  *ne;

						if (!ItemPointerIsValid(nt))
							break;		/* the list ends at the first invalid TID (hnswutils.c:800-801) */
						ne = (VsrElemNo *) hash_search(elems, nt, HASH_FIND, NULL);
						dst[i] = ne != NULL ? ne->no : -1;
					}
				}
			}
			UnlockReleaseBuffer(nbuf);
		}
		blk = HnswPageGetOpaque(page)->nextblkno;
		UnlockReleaseBuffer(buf);
	}

	if (entry >= 0 && pc->sc_handle != 0)
	{
		vsr_sc_hnsw_req req;

		memset(&req, 0, sizeof(req));
		req.handle = pc->sc_handle;
		req.m = m;
		req.n_elem = n_elem;
		req.entry = entry;
		req.n_upper = n_upper;
		req.max_level = max_level;
		if (vsr_sc_hnsw_load(VsrSidecar(), &req, 0, level, nbr0, tid_count, tids, up_slot, up_nbr) != 0)
			ereport(ERROR, (errcode(ERRCODE_EXTERNAL_ROUTINE_EXCEPTION), errmsg("%s", vsr_sc_error(VsrSidecar()))));
		pc->sc_has_hnsw = true;
	}
	else if (entry >= 0)
		VsrCheck(vsr_hnsw_load(pc->corpus, m, n_elem, entry, level, nbr0, tid_count, tids, up_slot, up_nbr, n_upper, max_level,
							   &graph));
	hash_destroy(elems);
	hash_destroy(tidrows);
	pfree(level);
	pfree(tid_count);
	pfree(up_slot);
	pfree(nbr0);
	pfree(tids);
	pfree(up_nbr);
	return graph;
}

/*
 * The lists of an ivfflat index: centres from the list pages, and for every heap row the list whose entry pages hold
 * its TID (ivfflat.h:222-248, ivfscan.c:112-176 walks the same pages at scan time).
 */
vsr_ivf *
VsrLoadIvfLists(Relation index, VsrPgCorpus * pc)
{
	Buffer		buf;
	Page		page;
	int			lists,
				dim;
	float	   *centers;
	int32	   *row_list;
	BlockNumber *start;
	int			nl = 0;
	HTAB	   *tidrows = vsr_pg_tid_rows(pc);
	vsr_ivf    *ivf = NULL;

	buf = ReadBuffer(index, IVFFLAT_METAPAGE_BLKNO);
	LockBuffer(buf, BUFFER_LOCK_SHARE);
	lists = IvfflatPageGetMeta(BufferGetPage(buf))->lists;
	dim = IvfflatPageGetMeta(BufferGetPage(buf))->dimensions;
	UnlockReleaseBuffer(buf);
	if (dim != pc->dim)
		elog(ERROR, "vsrbac: ivfflat index has %d dimensions, the resident corpus %d", dim, pc->dim);

	centers = (float *) palloc(sizeof(float) * (Size) lists * dim);
	start = (BlockNumber *) palloc(sizeof(BlockNumber) * lists);
	row_list = (int32 *) palloc0(sizeof(int32) * Max(pc->nrows, 1));

	/* list pages: one IvfflatListData (start page, centre) per list, in list order */
	for (BlockNumber blk = IVFFLAT_HEAD_BLKNO; BlockNumberIsValid(blk) && nl < lists;)
	{
		OffsetNumber maxoff;

		buf = ReadBuffer(index, blk);
		LockBuffer(buf, BUFFER_LOCK_SHARE);
		page = BufferGetPage(buf);
		maxoff = PageGetMaxOffsetNumber(page);
		for (OffsetNumber off = FirstOffsetNumber; off <= maxoff && nl < lists; off = OffsetNumberNext(off))
		{
			IvfflatList list = (IvfflatList) PageGetItem(page, PageGetItemId(page, off));

			memcpy(centers + (Size) nl * dim, list->center.x, sizeof(float) * dim);
			start[nl++] = list->startPage;
		}
		blk = IvfflatPageGetOpaque(page)->nextblkno;
		UnlockReleaseBuffer(buf);
	}

	/* entry pages of every list: the heap TID of each index tuple names the row */
	for (int l = 0; l < nl; l++)
		for (BlockNumber blk = start[l]; BlockNumberIsValid(blk);)
		{
			OffsetNumber maxoff;

			buf = ReadBuffer(index, blk);
			LockBuffer(buf, BUFFER_LOCK_SHARE);
			page = BufferGetPage(buf);
			maxoff = PageGetMaxOffsetNumber(page);
			for (OffsetNumber off = FirstOffsetNumber; off <= maxoff; off = OffsetNumberNext(off))
			{
				IndexTuple	itup = (IndexTuple) PageGetItem(page, PageGetItemId(page, off));
				VsrTidRow  *tr = (VsrTidRow *) hash_search(tidrows, &itup->t_tid, HASH_FIND, NULL);

				if (tr != NULL)
					row_list[tr->row] = l;
			}
			blk = IvfflatPageGetOpaque(page)->nextblkno;
			UnlockReleaseBuffer(buf);
		}

	if (pc->sc_handle != 0)
	{
		vsr_sc_ivf_req req;

		memset(&req, 0, sizeof(req));
		req.handle = pc->sc_handle;
		req.lists = nl;
		if (vsr_sc_ivf_load(VsrSidecar(), &req, dim, pc->nrows, centers, row_list) != 0)
			ereport(ERROR, (errcode(ERRCODE_EXTERNAL_ROUTINE_EXCEPTION), errmsg("%s", vsr_sc_error(VsrSidecar()))));
		pc->sc_has_ivf = true;
	}
	else
		VsrCheck(vsr_ivf_load(pc->corpus, centers, nl, row_list, &ivf));
	hash_destroy(tidrows);
	pfree(centers);
	pfree(start);
	pfree(row_list);
	return ivf;
}

/*
 * The index-faithful form of VsrRunSearch: the same result hand-out, the candidates are those of the index.
 * ef_or_probes: hnsw.ef_search / ivfflat.probes of the calling access method.
 */
void
VsrRunIndexSearch(IndexScanDesc scan, VsrPgScanOpaque so, bool is_hnsw, int ef_or_probes)
{
	VsrPgCorpus *pc = so->pc;
	MemoryContext old = MemoryContextSwitchTo(so->tmpCtx);
	Vector	   *q;
	const vsr_filter *filter = NULL;
	/* an hnsw scan can return ef_search items (hnswscan.c:44); ef_search goes to 5000, a result list to VSR_MAX_K */
	int			k = is_hnsw ? Min(ef_or_probes, VSR_MAX_K) : VSR_MAX_K;
	int64	   *blk = palloc(sizeof(int64) * k),
			   *rowidx = palloc(sizeof(int64) * k);
	float	   *dist = palloc(sizeof(float) * k);
	int32		count = 0;

	if (scan->orderByData == NULL)
		elog(ERROR, "cannot scan %s index without order", is_hnsw ? "hnsw" : "ivfflat");
	q = DatumGetVector(scan->orderByData->sk_argument);
	if (pc->sc_handle != 0)
	{
		/* sidecar mode: the graph / lists are loaded into the sidecar once (by whichever backend comes first) */
		vsr_sc_search_req req;

		if (is_hnsw ? !pc->sc_has_hnsw : !pc->sc_has_ivf)
		{
			if (is_hnsw)
				(void) VsrLoadHnswGraph(scan->indexRelation, pc);
			else
				(void) VsrLoadIvfLists(scan->indexRelation, pc);
		}
		memset(&req, 0, sizeof(req));
		req.handle = pc->sc_handle;
		req.nq = 1;
		req.dim = q->dim;
		req.k = k;
		req.metric = VsrMetricOf(scan->indexRelation);
		req.filter_mode = (vsr_pg_mode == VSR_PG_MODE_OFF || !pc->has_rbac || superuser()) ? -1 :
			vsr_pg_mode == VSR_PG_MODE_PREFILTER ? VSR_FILTER_RANGES : VSR_FILTER_BITMAP;
		req.user_id = VsrCurrentUserId();
		req.index = (is_hnsw ? pc->sc_has_hnsw : pc->sc_has_ivf) ? (is_hnsw ? 1 : 2) : 0;	/* empty index: the exact search */
		req.param = ef_or_probes;
		if (vsr_sc_search(VsrSidecar(), &req, q->x, &count, rowidx, blk, dist) != 0)
			ereport(ERROR, (errcode(ERRCODE_EXTERNAL_ROUTINE_EXCEPTION), errmsg("%s", vsr_sc_error(VsrSidecar()))));
		goto emit;
	}
	filter = VsrFilterForCurrentUser(pc);
	if (is_hnsw)
	{
		if (pc->graph == NULL)
			pc->graph = VsrLoadHnswGraph(scan->indexRelation, pc);
		if (pc->graph != NULL)
			VsrCheck(vsr_hnsw_set_predicate_aware(pc->graph, vsr_pg_predicate_aware ? 1 : 0));
		if (pc->graph != NULL)
			VsrCheck(vsr_hnsw_search(pc->graph, q->x, 1, q->dim, k, ef_or_probes, VsrMetricOf(scan->indexRelation),
									 filter ? &filter : NULL, blk, NULL, rowidx, dist, &count, NULL));
	}
	else
	{
		if (pc->ivf == NULL)
			pc->ivf = VsrLoadIvfLists(scan->indexRelation, pc);
		VsrCheck(vsr_ivf_search(pc->ivf, q->x, 1, q->dim, k, ef_or_probes, VsrMetricOf(scan->indexRelation),
								filter ? &filter : NULL, blk, NULL, rowidx, dist, &count));
	}
emit:
	so->result_tids = palloc(sizeof(ItemPointerData) * Max(count, 1));
	for (int i = 0; i < count; i++)
		so->result_tids[i] = pc->tids[rowidx[i]];
	so->nresults = count;
	so->next = 0;
	MemoryContextSwitchTo(old);
}
