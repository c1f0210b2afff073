"""ctypes binding of libvsrbac.so (include/vsrbac.h).  Loading never touches the GPU; vsr_open does."""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(os.path.dirname(_PKG), "lib", "libvsrbac.so")
_lib = None

OK, ERR_INVALID, ERR_DIM_MISMATCH, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_UNSUPPORTED, ERR_NO_RBAC = range(8)


class VsrError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


class Stats(C.Structure):
    _fields_ = [("scan_launches", C.c_int64 * 2), ("scan_ms", C.c_double * 2), ("scan_bytes", C.c_int64 * 2),
                ("scan_rows", C.c_int64 * 2), ("select_launches", C.c_int64), ("select_ms", C.c_double),
                ("queries", C.c_int64), ("search_ms", C.c_double),
                ("scan_pairs", C.c_int64 * 2), ("unique_rows", C.c_int64 * 2),
                ("host_ms", C.c_double), ("host_wait_ms", C.c_double)]


# every symbol include/vsrbac.h declares: name -> (restype, argtypes)
_vp, _i, _i64, _i32 = C.c_void_p, C.c_int, C.c_int64, C.c_int32
SYMBOLS = {
    "vsr_abi_version": (_i, []),
    "vsr_last_error": (C.c_char_p, []),
    "vsr_status_string": (C.c_char_p, [_i]),
    "vsr_open": (_i, [_i, C.POINTER(_vp)]),
    "vsr_close": (_i, [_vp]),
    "vsr_set_stream": (_i, [_vp, _vp]),
    "vsr_synchronize": (_i, [_vp]),
    "vsr_device_info": (_i, [_vp, C.c_char_p, _i, C.POINTER(_i), C.POINTER(_i64)]),
    "vsr_corpus_load": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _i64, C.POINTER(_vp)]),
    "vsr_corpus_free": (_i, [_vp]),
    "vsr_corpus_rows": (_i64, [_vp]),
    "vsr_corpus_dim": (_i, [_vp]),
    "vsr_rbac_load": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _i64]),
    "vsr_filter_for_user": (_i, [_vp, _i32, _i, C.POINTER(_vp)]),
    "vsr_filter_for_roles": (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "vsr_filter_from_bytemask": (_i, [_vp, _vp, _i, C.POINTER(_vp)]),
    "vsr_filter_from_documents": (_i, [_vp, _vp, _i64, _i32, C.POINTER(_vp)]),
    "vsr_filter_free": (_i, [_vp]),
    "vsr_filter_allowed_rows": (_i64, [_vp]),
    "vsr_filter_scanned_rows": (_i64, [_vp]),
    "vsr_search": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vsr_search_device": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vsr_search_device_on": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vsr_search_device_exact": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vsr_set_screening": (_i, [_vp, _i]),
    "vsr_set_query_hint": (_i, [_vp, _i]),
    "vsr_screening_check": (_i, [_vp, C.POINTER(_i64), _vp, _i]),
    "vsr_merge_topk_device": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "vsr_packed_result_bytes": (_i64, [_i, _i]),
    "vsr_merge_topk_packed_device": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "vsr_pair_distances": (_i, [_vp, _i, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "vsr_ivf_load": (_i, [_vp, _vp, _i, _vp, C.POINTER(_vp)]),
    "vsr_ivf_free": (_i, [_vp]),
    "vsr_ivf_assign": (_i, [_vp, _vp, _i, _i, _vp]),
    "vsr_ivf_kmeans": (_i, [_vp, _i, _i, _vp, _i64, _i, C.c_uint64, _vp, C.POINTER(_i)]),
    "vsr_ivf_probe": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "vsr_ivf_search": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vsr_ivf_search_device": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vsr_hnsw_load": (_i, [_vp, _i, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, C.POINTER(_vp)]),
    "vsr_hnsw_free": (_i, [_vp]),
    "vsr_hnsw_search": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vsr_hnsw_search_device": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vsr_hnsw_build": (_i, [_vp, _i, _i, _i, C.c_uint64, _vp]),
    "vsr_hnsw_info": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "vsr_hnsw_set_predicate_aware": (_i, [_vp, _i]),
    "vsr_vector_norms": (_i, [_vp, _vp, _i64, _i, _vp]),
    "vsr_l2_normalize": (_i, [_vp, _vp, _i64, _i, _vp]),
    "vsr_spherical_distances": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "vsr_profiling": (_i, [_vp, _i]),
    "vsr_stats_get": (_i, [_vp, C.POINTER(Stats)]),
    "vsr_stats_reset": (_i, [_vp]),
    "vsr_last_scan_kernel": (_i, [_vp, C.c_char_p, _i]),
    "vsr_tune": (_i, [_vp, _i, _i, _i]),
}


def library_path():
    return os.environ.get("VSRBAC_LIB", _LIB_PATH)


def load_library():
    """Load libvsrbac.so and bind every declared symbol.  Raises if the HIP extension is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise VsrError(ERR_NO_DEVICE, f"libvsrbac.so not found at {path}: build it with "
                                      f"`make -C vectorsearch-rbac_amd` (there is no CPU fallback)")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def abi_version():
    return load_library().vsr_abi_version()


def check(status):
    if status != OK:
        lib = load_library()
        msg = lib.vsr_last_error().decode(errors="replace") or lib.vsr_status_string(status).decode()
        raise VsrError(status, msg)
