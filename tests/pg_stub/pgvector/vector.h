/* TEST INFRASTRUCTURE — the few declarations of pgvector/src/vector.h the shim uses (vector.h:4-17) */
#ifndef PG_STUB_VECTOR_H
#define PG_STUB_VECTOR_H
#include "postgres.h"
typedef struct Vector { int32 vl_len_; int16 dim; int16 unused; float x[1]; } Vector;
extern Vector *pg_stub_detoast_vector(Datum d);
#define DatumGetVector(d) pg_stub_detoast_vector(d)
#endif
