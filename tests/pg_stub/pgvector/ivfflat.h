/* TEST INFRASTRUCTURE — the declarations of pgvector/src/ivfflat.h the shim uses (ivfflat.h:42-44, 84-91, 222-248) */
#ifndef PG_STUB_IVFFLAT_H
#define PG_STUB_IVFFLAT_H
#include "postgres.h"
#include "vector.h"
#define IVFFLAT_METAPAGE_BLKNO 0
#define IVFFLAT_HEAD_BLKNO 1
#define IVFFLAT_MAX_LISTS 32768
extern int ivfflat_probes, ivfflat_iterative_scan, ivfflat_max_probes;
typedef enum { IVFFLAT_ITERATIVE_SCAN_OFF, IVFFLAT_ITERATIVE_SCAN_RELAXED } IvfflatIterativeScanMode;
typedef struct IvfflatMetaPageData { uint32 magicNumber, version; uint16 dimensions, lists; } IvfflatMetaPageData;
typedef IvfflatMetaPageData *IvfflatMetaPage;
typedef struct IvfflatPageOpaqueData { BlockNumber nextblkno; uint16 unused, page_id; } IvfflatPageOpaqueData;
typedef IvfflatPageOpaqueData *IvfflatPageOpaque;
extern IvfflatMetaPage IvfflatPageGetMeta(Page page);
extern IvfflatPageOpaque IvfflatPageGetOpaque(Page page);
typedef struct IvfflatListData { BlockNumber startPage, insertPage; Vector center; } IvfflatListData;
typedef IvfflatListData *IvfflatList;
#endif
