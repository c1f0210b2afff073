"""Second pin of the CPU oracle: the hnswlib copy the reference vendors, compiled in place (oracle/Makefile `ref` ->
oracle/_ref/libref_hnswlib.so, driver oracle/ref_hnswlib_driver.cpp) and run here as a witness.

  * BruteforceSearch + BaseFilterFunctor (hnswlib/bruteforce.h:107-135) is exact filtered k-NN: the oracle's
    orc_filtered_topk must return the same rows / values (L2Space = squared L2 sum, InnerProductSpace = 1 - dot).
  * HierarchicalNSW with the parameters of the reference's own comparison test
    (logical_partition_benchmark/benchmark/src/tests/test_hnsw_compare.cpp:71-79) on that test's dataset formula
    (:16-25) finds the exact neighbours, like the FAISS/hnswlib pair that test compares.

The library only exists where the reference tree was available at build time (this container); the tests skip without
it.  It is a checker, never the thing measured or shipped.
"""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import sift_like

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libref_hnswlib.so")

pytestmark = pytest.mark.skipif(not os.path.exists(REF_LIB), reason="oracle/_ref not built (no reference tree)")


@pytest.fixture(scope="module")
def ref():
    lib = C.CDLL(REF_LIB)
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
    i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
    lib.ref_bruteforce_topk.restype = C.c_int
    lib.ref_bruteforce_topk.argtypes = [C.c_int, C.c_int, f32p, C.c_int64, f32p, C.c_int, C.c_int, C.c_void_p, i64p, f32p,
                                        i32p]
    lib.ref_hnsw_topk.restype = C.c_int
    lib.ref_hnsw_topk.argtypes = [C.c_int, C.c_int, f32p, C.c_int64, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_int,
                                  C.c_void_p, i64p, f32p, i32p]
    return lib


def _run(fn, *args, nq, k):
    ids = np.empty((nq, k), np.int64)
    dist = np.empty((nq, k), np.float32)
    cnt = np.empty(nq, np.int32)
    assert fn(*args, ids, dist, cnt) == 0
    return ids, dist, cnt


def _compare_dataset(d, n):
    """generate_dataset of test_hnsw_compare.cpp:16-25."""
    i = np.arange(n, dtype=np.float64)[:, None]
    j = np.arange(d, dtype=np.float64)[None, :]
    return (i * 0.5 + j * 0.1).astype(np.float32)


@pytest.mark.parametrize("metric,code", [("l2", 0), ("ip", 1)])
def test_filtered_bruteforce_integer_data_identical(ref, oracle, metric, code):
    """Integer-valued rows: fp32 sums are exact in any order, so values must be identical; ids identical up to ties."""
    rng = np.random.default_rng(31)
    n, dim, k, nq = 4000, 128, 50, 8
    x = sift_like(rng, n, dim)
    mask = (rng.random(n) < 0.3).astype(np.uint8)
    q = np.ascontiguousarray(x[rng.integers(0, n, nq)])
    ids, dist, cnt = _run(ref.ref_bruteforce_topk, code, dim, x, n, q, nq, k, mask.ctypes.data_as(C.c_void_p), nq=nq, k=k)
    for i in range(nq):
        oidx, odist = oracle.filtered_topk(metric, x, q[i], k, mask=mask)
        assert cnt[i] == k == oidx.size
        assert mask[ids[i]].all()
        # the operator values: l2 = sqrt(sum), ip = -dot; hnswlib reports sum and 1 - dot
        want = odist ** 2 if metric == "l2" else 1.0 + odist
        np.testing.assert_array_equal(np.sort(dist[i].astype(np.float64)), np.sort(np.round(want)))
        # same rows wherever the value is not tied with the k-th one
        kth = dist[i].max()
        assert set(ids[i][dist[i] < kth]) == set(oidx[np.round(want) < kth])


def test_unfiltered_bruteforce_real_valued_within_tolerance(ref, oracle):
    rng = np.random.default_rng(32)
    n, dim, k, nq = 3000, 96, 20, 5
    x = rng.normal(size=(n, dim)).astype(np.float32)
    q = rng.normal(size=(nq, dim)).astype(np.float32)
    ids, dist, cnt = _run(ref.ref_bruteforce_topk, 0, dim, x, n, q, nq, k, None, nq=nq, k=k)
    for i in range(nq):
        oidx, odist = oracle.filtered_topk("l2", x, q[i], k)
        np.testing.assert_allclose(np.sqrt(dist[i].astype(np.float64)), odist, rtol=1e-5, atol=1e-5)
        np.testing.assert_array_equal(ids[i], oidx)          # gaps between neighbours are far above fp32 noise here


def test_reference_compare_test_dataset_hnsw_and_bruteforce(ref, oracle):
    """d = 8, nb = 64, nq = 10, k = 5, M = 16, efConstruction = 60, efSearch = 32 (test_hnsw_compare.cpp:71-79)."""
    d, nb, nq, k = 8, 64, 10, 5
    base, queries = _compare_dataset(d, nb), _compare_dataset(d, nq)
    for code, metric in ((0, "l2"), (1, "ip")):
        b_ids, b_dist, _ = _run(ref.ref_bruteforce_topk, code, d, base, nb, queries, nq, k, None, nq=nq, k=k)
        h_ids, h_dist, _ = _run(ref.ref_hnsw_topk, code, d, base, nb, 16, 60, 32, queries, nq, k, None, nq=nq, k=k)
        np.testing.assert_array_equal(h_ids, b_ids)
        for i in range(nq):
            oidx, odist = oracle.filtered_topk(metric, base, queries[i], k)
            np.testing.assert_array_equal(b_ids[i], oidx)
            want = odist ** 2 if metric == "l2" else 1.0 + odist
            np.testing.assert_allclose(b_dist[i], want, rtol=1e-4, atol=1e-4)     # the tolerance of that test (:27-32)
