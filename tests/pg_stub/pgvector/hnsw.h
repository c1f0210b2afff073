/* TEST INFRASTRUCTURE — the declarations of pgvector/src/hnsw.h the shim uses (hnsw.h:15-89, 111-121, 304-350) */
#ifndef PG_STUB_HNSW_H
#define PG_STUB_HNSW_H
#include "postgres.h"
#include "vector.h"
#define HNSW_METAPAGE_BLKNO 0
#define HNSW_HEAD_BLKNO 1
#define HNSW_SCAN_LOCK 1
#define HNSW_HEAPTIDS 10
#define HNSW_MAX_EF_SEARCH 5000
#define HNSW_ELEMENT_TUPLE_TYPE 1
extern int hnsw_ef_search, hnsw_iterative_scan, hnsw_max_scan_tuples;
typedef enum { HNSW_ITERATIVE_SCAN_OFF, HNSW_ITERATIVE_SCAN_RELAXED, HNSW_ITERATIVE_SCAN_STRICT } HnswIterativeScanMode;
typedef struct HnswMetaPageData { uint32 magicNumber, version, dimensions; uint16 m, efConstruction; BlockNumber entryBlkno; OffsetNumber entryOffno; int16 entryLevel; BlockNumber insertPage; } HnswMetaPageData;
typedef HnswMetaPageData *HnswMetaPage;
typedef struct HnswPageOpaqueData { BlockNumber nextblkno; uint16 unused, page_id; } HnswPageOpaqueData;
typedef HnswPageOpaqueData *HnswPageOpaque;
extern HnswMetaPage HnswPageGetMeta(Page page);
extern HnswPageOpaque HnswPageGetOpaque(Page page);
typedef struct HnswElementTupleData { uint8 type, level, deleted, version; ItemPointerData heaptids[HNSW_HEAPTIDS]; ItemPointerData neighbortid; uint16 unused; Vector data; } HnswElementTupleData;
typedef HnswElementTupleData *HnswElementTuple;
typedef struct HnswNeighborTupleData { uint8 type, version; uint16 count; ItemPointerData indextids[1]; } HnswNeighborTupleData;
typedef HnswNeighborTupleData *HnswNeighborTuple;
#define HnswIsElementTuple(t) ((t)->type == HNSW_ELEMENT_TUPLE_TYPE)
#define HnswGetLayerM(m, layer) ((layer) == 0 ? (m) * 2 : (m))
#endif
