/*
 * TEST INFRASTRUCTURE — minimal stand-in for the PostgreSQL server headers, exactly wide enough for `cc -fsyntax-only` over
 * the C files of pg_shim/ (tests/test_pg_shim_syntax.py).  It declares the names, types and macros the shim uses with plausible
 * shapes; it is NOT PostgreSQL, proves nothing about behaviour, and is never compiled into anything.  The authoring image
 * has no postgres.h; where pg_config exists the shim is built against the real headers by pg_shim/Makefile.
 */
#ifndef PG_STUB_POSTGRES_H
#define PG_STUB_POSTGRES_H

#include <limits.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int16_t int16;
typedef int32_t int32;
typedef int64_t int64;
typedef uint8_t uint8;
typedef uint16_t uint16;
typedef uint32_t uint32;
typedef uint64_t uint64;
typedef size_t Size;
typedef uintptr_t Datum;
typedef unsigned int Oid;
typedef int16 AttrNumber;
typedef uint32 BlockNumber;
typedef uint16 OffsetNumber;
typedef int Buffer;
typedef char *Page;
typedef void *Pointer;
#define InvalidOid ((Oid) 0)
#define OidIsValid(o) ((o) != InvalidOid)
#define InvalidBlockNumber ((BlockNumber) 0xFFFFFFFF)
#define BlockNumberIsValid(b) ((b) != InvalidBlockNumber)
#define FirstOffsetNumber ((OffsetNumber) 1)
#define OffsetNumberNext(o) ((OffsetNumber) (1 + (o)))
#define UINT64CONST(x) UINT64_C(x)
#define Max(a, b) ((a) > (b) ? (a) : (b))
#define Min(a, b) ((a) < (b) ? (a) : (b))
#define Assert(c) ((void) 0)
#define PGDLLEXPORT
#define PG_FUNCTION_ARGS void *fcinfo
#define NameStr(n) ((n).data)
#define DatumGetInt32(d) ((int32) (d))
#define DatumGetInt64(d) ((int64) (d))
#define DatumGetPointer(d) ((Pointer) (d))

typedef struct { char data[64]; } NameData;
typedef struct ItemPointerData { uint16 bi_hi, bi_lo; OffsetNumber ip_posid; } ItemPointerData;
typedef ItemPointerData *ItemPointer;
static inline void ItemPointerSet(ItemPointer p, BlockNumber b, OffsetNumber o) { p->bi_hi = (uint16) (b >> 16); p->bi_lo = (uint16) b; p->ip_posid = o; }
static inline BlockNumber ItemPointerGetBlockNumber(const ItemPointerData *p) { return ((BlockNumber) p->bi_hi << 16) | p->bi_lo; }
static inline OffsetNumber ItemPointerGetOffsetNumber(const ItemPointerData *p) { return p->ip_posid; }
static inline bool ItemPointerIsValid(const ItemPointerData *p) { return p->ip_posid != 0; }

/* memory contexts */
typedef struct MemoryContextData *MemoryContext;
extern MemoryContext CurrentMemoryContext, TopMemoryContext;
extern MemoryContext MemoryContextSwitchTo(MemoryContext c);
extern void *MemoryContextAlloc(MemoryContext c, Size n);
extern void *MemoryContextAllocHuge(MemoryContext c, Size n);
extern void MemoryContextReset(MemoryContext c);
extern void MemoryContextDelete(MemoryContext c);
extern MemoryContext AllocSetContextCreateInternal(MemoryContext parent, const char *name, Size a, Size b, Size c);
#define ALLOCSET_DEFAULT_SIZES 0, 8192, 8388608
#define AllocSetContextCreate AllocSetContextCreateInternal
extern void *palloc(Size n);
extern void *palloc0(Size n);
extern void *repalloc_huge(void *p, Size n);
extern void pfree(void *p);

/* error reporting */
#define ERROR 21
#define ERRCODE_DATA_EXCEPTION 1
#define ERRCODE_OUT_OF_MEMORY 2
#define ERRCODE_EXTERNAL_ROUTINE_EXCEPTION 3
#define ERRCODE_CONNECTION_FAILURE 4
extern int errcode(int code);
extern int errmsg(const char *fmt,...) __attribute__((format(printf, 1, 2)));
extern void pg_stub_ereport(int level, ...);
#define ereport(level, rest) pg_stub_ereport(level, rest)
#define errstart_args(...) __VA_ARGS__
extern void elog(int level, const char *fmt,...) __attribute__((format(printf, 2, 3)));

/* catalog / relations */
typedef struct FormData_pg_class { Oid relfilenode; float reltuples; } FormData_pg_class;
typedef struct FormData_pg_attribute { NameData attname; } FormData_pg_attribute;
typedef struct TupleDescData { int natts; FormData_pg_attribute attrs[1]; } *TupleDesc;
#define TupleDescAttr(desc, i) (&(desc)->attrs[i])
typedef struct { int16 values[1]; } int2vector;
typedef struct FormData_pg_index { Oid indrelid; int2vector indkey; } FormData_pg_index;
typedef struct RelationData
{
	Oid			rd_id;
	FormData_pg_class *rd_rel;
	FormData_pg_index *rd_index;
	TupleDesc	rd_att;
	void	   *rd_support;
}		   *Relation;
#define RelationGetRelid(r) ((r)->rd_id)
#define RelationGetDescr(r) ((r)->rd_att)
extern BlockNumber RelationGetNumberOfBlocks(Relation r);
typedef int LOCKMODE;
#define AccessShareLock 1
#define ShareLock 5
extern Relation table_open(Oid relid, LOCKMODE mode);
extern void table_close(Relation r, LOCKMODE mode);
extern Oid RelnameGetRelid(const char *name);
extern Oid index_getprocid(Relation index, AttrNumber att, uint16 procnum);
extern char *get_func_name(Oid funcid);
extern Oid GetUserId(void);
extern char *GetUserNameFromId(Oid roleid, bool noerr);
extern bool superuser(void);
extern Oid MyDatabaseId;

/* snapshots, scans, slots */
typedef struct SnapshotData *Snapshot;
extern Snapshot GetActiveSnapshot(void);
#define IsMVCCSnapshot(s) ((s) != NULL)
typedef struct TupleTableSlot { ItemPointerData tts_tid; } TupleTableSlot;
typedef struct TableScanDescData *TableScanDesc;
typedef enum { BackwardScanDirection = -1, NoMovementScanDirection = 0, ForwardScanDirection = 1 } ScanDirection;
#define ScanDirectionIsForward(d) ((d) == ForwardScanDirection)
extern TableScanDesc table_beginscan(Relation rel, Snapshot snap, int nkeys, void *keys);
extern bool table_scan_getnextslot(TableScanDesc scan, ScanDirection dir, TupleTableSlot *slot);
extern void table_endscan(TableScanDesc scan);
extern TupleTableSlot *table_slot_create(Relation rel, void *reglist);
extern void ExecDropSingleTupleTableSlot(TupleTableSlot *slot);
extern Datum slot_getattr(TupleTableSlot *slot, int attnum, bool *isnull);

/* index scans */
typedef struct ScanKeyData { int sk_flags; Datum sk_argument; } ScanKeyData;
typedef ScanKeyData *ScanKey;
#define SK_ISNULL 1
typedef struct IndexScanDescData
{
	Relation	indexRelation;
	Snapshot	xs_snapshot;
	int			numberOfKeys, numberOfOrderBys;
	ScanKey		keyData, orderByData;
	void	   *opaque;
	ItemPointerData xs_heaptid;
	bool		xs_recheck, xs_recheckorderby;
}		   *IndexScanDesc;
extern IndexScanDesc RelationGetIndexScan(Relation index, int nkeys, int norderbys);
typedef struct IndexTupleData { ItemPointerData t_tid; uint16 t_info; } IndexTupleData;
typedef IndexTupleData *IndexTuple;

/* pages and buffers */
#define BUFFER_LOCK_SHARE 1
extern Buffer ReadBuffer(Relation rel, BlockNumber blk);
extern void LockBuffer(Buffer buf, int mode);
extern void UnlockReleaseBuffer(Buffer buf);
extern Page BufferGetPage(Buffer buf);
typedef struct ItemIdData { unsigned lp_off:15, lp_flags:2, lp_len:15; } ItemIdData;
typedef ItemIdData *ItemId;
extern OffsetNumber PageGetMaxOffsetNumber(Page page);
extern ItemId PageGetItemId(Page page, OffsetNumber off);
extern void *PageGetItem(Page page, ItemId id);
extern void LockPage(Relation rel, BlockNumber blk, LOCKMODE mode);
extern void UnlockPage(Relation rel, BlockNumber blk, LOCKMODE mode);

/* hash tables */
typedef struct HTAB HTAB;
typedef struct HASHCTL { Size keysize, entrysize; MemoryContext hcxt; } HASHCTL;
typedef enum { HASH_FIND, HASH_ENTER, HASH_REMOVE } HASHACTION;
#define HASH_ELEM 1
#define HASH_BLOBS 2
#define HASH_CONTEXT 4
typedef struct HASH_SEQ_STATUS { int x; } HASH_SEQ_STATUS;
extern HTAB *hash_create(const char *name, long nelem, const HASHCTL *info, int flags);
extern void *hash_search(HTAB *h, const void *key, HASHACTION action, bool *found);
extern void hash_seq_init(HASH_SEQ_STATUS *s, HTAB *h);
extern void *hash_seq_search(HASH_SEQ_STATUS *s);
extern void hash_destroy(HTAB *h);

/* SPI */
#define SPI_OK_CONNECT 1
#define SPI_OK_SELECT 5
typedef struct HeapTupleData *HeapTuple;
typedef struct SPITupleTable { TupleDesc tupdesc; HeapTuple *vals; } SPITupleTable;
extern SPITupleTable *SPI_tuptable;
extern uint64 SPI_processed;
extern int SPI_connect(void);
extern int SPI_finish(void);
extern int SPI_execute(const char *sql, bool read_only, long count);
extern Datum SPI_getbinval(HeapTuple tuple, TupleDesc desc, int fnumber, bool *isnull);

/* GUCs, process exit, invalidation */
typedef enum { PGC_USERSET = 6 } GucContext;
struct config_enum_entry { const char *name; int val; bool hidden; };
extern void DefineCustomIntVariable(const char *name, const char *s, const char *l, int *v, int boot, int min, int max, GucContext c,
									int flags, void *check, void *assign, void *show);
extern void DefineCustomBoolVariable(const char *name, const char *s, const char *l, bool *v, bool boot, GucContext c, int flags,
									 void *check, void *assign, void *show);
extern void DefineCustomEnumVariable(const char *name, const char *s, const char *l, int *v, int boot,
									 const struct config_enum_entry *opts, GucContext c, int flags, void *check, void *assign, void *show);
extern void DefineCustomStringVariable(const char *name, const char *s, const char *l, char **v, const char *boot, GucContext c,
									   int flags, void *check, void *assign, void *show);
extern void MarkGUCPrefixReserved(const char *prefix);
extern void on_proc_exit(void (*fn) (int code, Datum arg), Datum arg);
extern void CacheRegisterRelcacheCallback(void (*fn) (Datum arg, Oid relid), Datum arg);

#endif
