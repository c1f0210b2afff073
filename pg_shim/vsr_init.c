/*
 * vsr_init.c — the module's _PG_init.
 *
 * pgvector's own _PG_init (pgvector/src/vector.c:47-55: BitvecInit, HalfvecInit, HnswInit, IvfflatInit) is the only init
 * hook a loadable module has, and vector.c is compiled unchanged -- so the Makefile renames it while compiling that one
 * file (-D_PG_init=vector_PG_init) and this file provides the real one: pgvector's four inits, then the shim's GUCs
 * (vsrbac.device, vsrbac.mode, vsrbac.index_faithful, vsrbac.sidecar, vsrbac.epoch) and its cache-invalidation callback.
 */
#include "postgres.h"

#include "fmgr.h"

#include "vsr_pg.h"

extern void vector_PG_init(void);	/* pgvector/src/vector.c:47-55, renamed by the Makefile */

PGDLLEXPORT void _PG_init(void);

void
_PG_init(void)
{
	vector_PG_init();
	VsrPgInit();
}
