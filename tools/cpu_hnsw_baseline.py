#!/usr/bin/env python3
"""Reported CPU baseline no. 2 (not used by bench.py): approximate search with the hnswlib copy the reference vendors
(oracle/_ref/libref_hnswlib.so, built by `make -C oracle ref` where the reference tree is present), with the index
parameters of the reference's role-partition experiment (m = 16, ef_construction = 64:
basic_benchmark/test_partition_prefilter_by_role.py:42-46) on ONE role partition of the SIFT-like corpus, swept over
ef_search until recall@k >= 0.95 against the exact oracle.  One thread, like the harness's single-connection queries.

  python tools/cpu_hnsw_baseline.py [--rows 100000] [--queries 200] [--k 100]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vectorsearch-rbac_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000, help="rows of the role partition (SIFT10M tree RBAC: 100k-400k)")
    ap.add_argument("--queries", type=int, default=200)
    ap.add_argument("--k", type=int, default=100)
    args = ap.parse_args()
    from vsrbac.datasets import sift_like_corpus
    from oracle.oracle import Oracle
    lib_path = os.path.join(ROOT, "oracle", "_ref", "libref_hnswlib.so")
    if not os.path.exists(lib_path):
        raise SystemExit("oracle/_ref/libref_hnswlib.so is missing: run `make -C oracle ref` where /root/reference exists")
    lib = C.CDLL(lib_path)
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
    i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
    lib.ref_hnsw_open.restype = C.c_void_p
    lib.ref_hnsw_open.argtypes = [C.c_int, C.c_int, f32p, C.c_int64, C.c_int, C.c_int]
    lib.ref_hnsw_query.restype = C.c_int
    lib.ref_hnsw_query.argtypes = [C.c_void_p, C.c_int, f32p, C.c_int, C.c_int, i64p, f32p, i32p]
    lib.ref_hnsw_close.argtypes = [C.c_void_p]

    n, nq, k, dim = args.rows, args.queries, args.k, 128
    x, blk, doc = sift_like_corpus(n, dim, seed=7)
    rng = np.random.default_rng(7)
    q = np.ascontiguousarray(x[rng.integers(0, n, nq)] + rng.integers(-2, 3, (nq, dim)).astype(np.float32))
    orc = Oracle("pgflags")
    t = time.perf_counter()
    exact = [set(orc.filtered_topk("l2", x, q[i], k)[0].tolist()) for i in range(nq)]
    t_exact = time.perf_counter() - t
    t = time.perf_counter()
    h = lib.ref_hnsw_open(0, dim, x, n, 16, 64)
    t_build = time.perf_counter() - t
    out = {"rows": n, "queries": nq, "k": k, "index": {"m": 16, "ef_construction": 64, "build_s": round(t_build, 1)},
           "exact_seq_scan_qps_1thread": round(nq / t_exact, 1), "sweep": []}
    ids = np.empty((nq, k), np.int64)
    dist = np.empty((nq, k), np.float32)
    cnt = np.empty(nq, np.int32)
    for ef in (100, 200, 400, 800, 1600, 3200, 6400):
        t = time.perf_counter()
        lib.ref_hnsw_query(h, ef, q, nq, k, ids, dist, cnt)
        dt = time.perf_counter() - t
        recall = float(np.mean([len(exact[i] & set(ids[i, :cnt[i]].tolist())) / k for i in range(nq)]))
        out["sweep"].append({"ef_search": ef, "recall": round(recall, 4), "qps_1thread": round(nq / dt, 1)})
        if recall >= 0.95 and "at_recall_0.95" not in out:
            out["at_recall_0.95"] = out["sweep"][-1]
    lib.ref_hnsw_close(h)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
