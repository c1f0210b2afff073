"""The PostgreSQL shim's sidecar (pg_shim/vsr_sidecar.c + vsr_client.c) end to end on the CPU: both are plain C over the
C ABI, so they are compiled here with gcc -- the sidecar against tests/fake_vsrbac.c, a stand-in for the handful of
libvsrbac entry points it calls (the real library needs the GPU) -- and driven through the real socket protocol.  What is
checked is what the sidecar is for: a corpus loaded over one connection is still resident for the next one (the reference
harness connects anew for every search, prefilter_role.py:86), a changed version drops the stale copy, the RBAC filter is
applied on the sidecar's side, errors travel back with their text."""
import ctypes as C
import os
import subprocess
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "pg_shim")
CFLAGS = ["-std=c99", "-D_POSIX_C_SOURCE=200809L", "-Wall", "-Wextra", "-Werror", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + SHIM]


class Info(C.Structure):
    _fields_ = [("handle", C.c_uint64), ("nrows", C.c_int64), ("dim", C.c_int32), ("has_rbac", C.c_int32),
                ("has_hnsw", C.c_int32), ("has_ivf", C.c_int32)]


class SearchReq(C.Structure):
    _fields_ = [("handle", C.c_uint64), ("nq", C.c_int32), ("dim", C.c_int32), ("k", C.c_int32), ("metric", C.c_int32),
                ("filter_mode", C.c_int32), ("user_id", C.c_int32), ("index", C.c_int32), ("param", C.c_int32)]


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    d = tmp_path_factory.mktemp("sidecar")
    exe, lib = str(d / "vsr_sidecar"), str(d / "libvsrclient.so")
    subprocess.check_call(["gcc", *CFLAGS, os.path.join(SHIM, "vsr_sidecar.c"), os.path.join(ROOT, "tests", "fake_vsrbac.c"), "-lm", "-o", exe])
    subprocess.check_call(["gcc", *CFLAGS, "-shared", "-fPIC", os.path.join(SHIM, "vsr_client.c"), "-o", lib])
    return exe, lib, str(d / "sock")


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def test_sidecar_keeps_the_corpus_across_connections(built):
    exe, libpath, sock = built
    proc = subprocess.Popen([exe, sock], stdout=subprocess.PIPE)
    try:
        assert b"listening" in proc.stdout.readline()
        lib = C.CDLL(libpath)
        lib.vsr_sc_connect.restype = C.c_void_p
        lib.vsr_sc_connect.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        lib.vsr_sc_error.restype = C.c_char_p
        lib.vsr_sc_error.argtypes = [C.c_void_p]
        for fn in ("vsr_sc_ping", "vsr_sc_close", "vsr_sc_shutdown"):
            getattr(lib, fn).argtypes = [C.c_void_p]
        lib.vsr_sc_corpus_lookup.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(Info)]
        lib.vsr_sc_corpus_load.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                           C.POINTER(Info)]
        lib.vsr_sc_rbac_load.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64]
        lib.vsr_sc_search.argtypes = [C.c_void_p, C.POINTER(SearchReq), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.vsr_sc_corpus_drop.argtypes = [C.c_void_p, C.c_uint64]
        err = C.create_string_buffer(256)
        rng = np.random.default_rng(5)
        n, dim, k = 500, 16, 7
        x = rng.integers(0, 50, (n, dim)).astype(np.float32)
        blk = (np.arange(n) + 1).astype(np.int64)
        doc = (np.arange(n) // 5 + 1).astype(np.int32)
        key, version = (7 << 32) | 1234, 99

        c1 = lib.vsr_sc_connect(sock.encode(), err, 256)
        assert c1, err.value
        assert lib.vsr_sc_ping(c1) == 0
        info = Info()
        assert lib.vsr_sc_corpus_lookup(c1, key, version, C.byref(info)) == 100          # VSR_SC_NOTFOUND
        assert lib.vsr_sc_corpus_load(c1, key, version, _ptr(x), n, dim, _ptr(blk), _ptr(doc), C.byref(info)) == 0
        assert (info.nrows, info.dim, info.has_rbac) == (n, dim, 0)
        uu = np.array([1, 2, 2], dtype=np.int32)
        ur = np.array([10, 10, 20], dtype=np.int32)
        pr = np.concatenate([np.full(30, 10), np.full(40, 20)]).astype(np.int32)
        pd = np.concatenate([np.arange(1, 31), np.arange(41, 81)]).astype(np.int32)
        assert lib.vsr_sc_rbac_load(c1, info.handle, _ptr(uu), _ptr(ur), 3, _ptr(pr), _ptr(pd), len(pr)) == 0
        lib.vsr_sc_close(c1)                                                           # the backend goes away ...

        c2 = lib.vsr_sc_connect(sock.encode(), err, 256)                               # ... the next one finds the corpus resident
        info2 = Info()
        assert lib.vsr_sc_corpus_lookup(c2, key, version, C.byref(info2)) == 0
        assert (info2.handle, info2.nrows, info2.has_rbac) == (info.handle, n, 1)
        nq = 3                                                                          # (odd: the counts block is padded)
        q = x[[3, 77, 400]] + 0.25
        req = SearchReq(info2.handle, nq, dim, k, 0, 1, 2, 0, 0)                        # user 2: roles 10 and 20, post-filter mode
        counts = np.zeros(nq, dtype=np.int32)
        rows = np.zeros((nq, k), dtype=np.int64)
        oblk = np.zeros((nq, k), dtype=np.int64)
        dist = np.zeros((nq, k), dtype=np.float32)
        assert lib.vsr_sc_search(c2, C.byref(req), _ptr(q), _ptr(counts), _ptr(rows), _ptr(oblk), _ptr(dist)) == 0
        ok_docs = set(range(1, 31)) | set(range(41, 81))
        allowed = np.array([d in ok_docs for d in doc])
        for i in range(nq):
            d = np.sqrt(((x.astype(np.float64) - q[i]) ** 2).sum(1))
            d[~allowed] = np.inf
            want = np.argsort(d, kind="stable")[:k]
            assert counts[i] == k
            np.testing.assert_allclose(dist[i], d[want].astype(np.float32), rtol=1e-6)
            assert set(rows[i].tolist()) == set(want.tolist())
            np.testing.assert_array_equal(oblk[i], blk[rows[i]])
        req.filter_mode = -1                                                            # unfiltered: rows of forbidden documents appear
        assert lib.vsr_sc_search(c2, C.byref(req), _ptr(q), _ptr(counts), _ptr(rows), _ptr(oblk), _ptr(dist)) == 0
        assert rows[0, 0] == 3 and not allowed[rows].all()
        req.dim = dim + 1                                                               # errors come back with the library's text
        bad = np.zeros((nq, dim + 1), dtype=np.float32)
        assert lib.vsr_sc_search(c2, C.byref(req), _ptr(bad), _ptr(counts), _ptr(rows), _ptr(oblk), _ptr(dist)) == 2
        assert b"different vector dimensions 16 and 17" in lib.vsr_sc_error(c2)
        # the heap changed: a lookup with the new version drops the stale copy
        assert lib.vsr_sc_corpus_lookup(c2, key, version + 1, C.byref(info2)) == 100
        assert lib.vsr_sc_corpus_lookup(c2, key, version, C.byref(info2)) == 100
        assert lib.vsr_sc_shutdown(c2) == 0
        lib.vsr_sc_close(c2)
        assert proc.wait(timeout=10) == 0
    finally:
        if proc.poll() is None:
            proc.kill()
