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


def test_pg_shim_calls_only_declared_abi_functions():
    """pg_shim/ cannot be compiled in this image (no postgres.h); at least every libvsrbac entry point it calls must be
    one the header declares (and therefore one the library exports, see above), with the shim's own helpers defined."""
    import glob
    import re
    with open(os.path.join(ROOT, "include", "vsrbac.h")) as f:
        declared = set(re.findall(r"\b(vsr_[a-z0-9_]+)\s*\(", f.read()))
    text = ""
    for path in sorted(glob.glob(os.path.join(ROOT, "pg_shim", "*.[ch]"))):
        with open(path) as f:
            text += f.read()
    called = set(re.findall(r"\b(vsr_[a-z0-9_]+)\s*\(", text))
    own = set(re.findall(r"^(vsr_pg_[a-z0-9_]+)\s*\(", text, flags=re.M))          # the shim's own static functions
    own |= set(re.findall(r"^(vsr_sc_[a-z0-9_]+)\s*\(", text, flags=re.M))        # ... and its sidecar client (vsr_client.c)
    own |= {n for n in called if n.startswith("vsr_sc_") and n.endswith(("_req", "_info", "_key", "_hdr", "_reply", "_result"))}
    assert called - declared - own == set(), sorted(called - declared - own)
    helpers_called = set(re.findall(r"\b(Vsr[A-Z][A-Za-z0-9]+)\s*\(", text))
    helpers_defined = set(re.findall(r"^(Vsr[A-Z][A-Za-z0-9]+)\s*\(", text, flags=re.M))
    assert helpers_called <= helpers_defined, sorted(helpers_called - helpers_defined)


def test_kernel_parameter_blocks_are_value_initialised():
    """Every kernel parameter block (ScanParams, SelectParams, RerankParams, StageParams, HnswParams, ...) is declared
    `T x{};`.  The round-2 GPU memory fault was a `StageParams st;` at one call site whose newly added pointer fields
    (the int8 query planes) kept stack garbage and were dereferenced by stage_kernel."""
    import glob
    import re
    csrc = os.path.join(ROOT, "vectorsearch-rbac_amd", "csrc")
    names = set()
    for path in glob.glob(os.path.join(csrc, "*")):
        names |= set(re.findall(r"struct\s+(\w+Params)\b", open(path).read()))
    assert {"ScanParams", "SelectParams", "RerankParams", "StageParams", "HnswParams"} <= names
    bad = []
    for path in glob.glob(os.path.join(csrc, "*")):
        for ln, line in enumerate(open(path).read().splitlines(), 1):
            for nm in names:
                if re.search(rf"^\s*(?:vsr::)?{nm}\s+\w+\s*;", line):
                    bad.append(f"{os.path.basename(path)}:{ln}: {line.strip()}")
    assert not bad, bad
