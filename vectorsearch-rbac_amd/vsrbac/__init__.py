"""vsrbac — host side of the MI355X RBAC-filtered vector search path.

Python mirror of the reference's caller interface for this path (the `search_func` contract of
basic_benchmark/condition_config.py:12-38 and the run loop of
basic_benchmark/common_function.py:1321-1434) over the C ABI of include/vsrbac.h.
There is no CPU fallback: without libvsrbac.so and a gfx950 GPU every compute entry point raises.
"""
from ._ffi import VsrError, abi_version, library_path, load_library  # noqa: F401
from .engine import (BITMAP, COSINE, IP, L1, L2, METRICS, RANGES, Context, Corpus, Filter,  # noqa: F401
                     SearchResult)
