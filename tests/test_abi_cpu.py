"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol that
include/vsrbac.h declares, and refuses to compute without a gfx950 device (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "vsrbac.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vsr_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound():
    import vsrbac
    from vsrbac import _ffi
    declared = _declared_symbols()
    assert len(declared) >= 25
    lib = ctypes.CDLL(vsrbac.library_path())
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/vsrbac.h but not exported by libvsrbac.so"
    assert sorted(_ffi.SYMBOLS) == declared, "python binding and header disagree"
    assert vsrbac.abi_version() == 2


def test_header_compiles_as_plain_c(tmp_path):
    """The boundary is C: no C++/torch types may leak into the header."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "vsrbac.h"\nint main(void){ vsr_stats s; (void)s; return VSR_ABI_VERSION - 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                           "-o", str(tmp_path / "t.o")])


def test_no_cpu_fallback_when_gpu_missing():
    import torch
    import vsrbac
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(vsrbac.VsrError) as e:
        vsrbac.Context(0)
    assert e.value.status == 3 and "no CPU path" in str(e.value)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the product package may reference it."""
    pkg = os.path.join(ROOT, "vectorsearch-rbac_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle" not in text.lower() or f == "vsr_topk.h", os.path.join(base, f)
