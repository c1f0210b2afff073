"""Build-time guards on the generated code of the streaming kernels (no GPU needed: hipcc cross-compiles to assembly).

K2i (vsr_i8s.h) keeps three stages of LDS-DMA loads in flight per wave and waits for them with counted `s_waitcnt vmcnt(N)`.
One `s_waitcnt vmcnt(0)` the compiler adds on its own inside the stage loop drains that queue at every stage -- the kernel
still computes the right answer, only 2-3x slower, so nothing but a look at the assembly catches it.  The ways that happened
while the kernel was written are listed in its header; this test keeps them from coming back."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vectorsearch-rbac_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _asm(tmp_path, body):
    src = tmp_path / "tu.hip"
    src.write_text(body)
    out = tmp_path / "tu.s"
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-I" + CSRC,
                    "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", str(src), "-o", str(out)],
                   check=True, capture_output=True)
    return out.read_text().splitlines()


def _kernel(lines, mangled):
    start = next(i for i, l in enumerate(lines) if l.startswith(mangled + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start:end + 1]


@pytest.mark.skipif(not shutil.which(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("nqg,sample", [(4, False), (8, False), (4, True)])
def test_k2i_stage_loop_has_only_counted_waits(tmp_path, nqg, sample):
    flag = "true" if sample else "false"
    lines = _asm(tmp_path, '#include <hip/hip_runtime.h>\n#include "vsr_i8s.h"\n'
                           f'namespace vsr {{ template __global__ void i8_stream_kernel<{nqg}, {flag}>(const ScanParams); }}\n')
    k = _kernel(lines, f"_ZN3vsr16i8_stream_kernelILi{nqg}ELb{int(sample)}EEEvNS_10ScanParamsE")
    mfma = [i for i, l in enumerate(k) if "v_mfma_i32_16x16x64_i8" in l]
    assert len(mfma) == 4 * nqg, len(mfma)                               # 2 row blocks x NQG groups x 2 k-steps, no unrolled copies
    head = max(i for i, l in enumerate(k) if "Loop Header" in l and "Depth=1" in l and i < mfma[0])
    if sample:
        # the sample variant's stage has no rare branch: the whole loop body is the common path
        label = k[head].split(":")[0]
        tail = max(i for i, l in enumerate(k) if "Header=" + label.lstrip(".L") in l)      # the last block of the loop
    else:
        # the common path of a stage: from the loop header through the MFMAs and the candidate compares up to the scalar test
        # that skips the (rare) candidate handling
        tail = next(i for i in range(mfma[-1], len(k)) if "s_cmp_eq_u64" in k[i])
    waits = [l.strip() for l in k[head:tail] if "s_waitcnt" in l and "vmcnt" in l]
    assert waits, "the counted waits are gone"
    assert all(re.fullmatch(r"s_waitcnt vmcnt\((10|12|16)\)", w) for w in waits), waits
    meta = [l for l in lines if ".vgpr_spill_count" in l]
    assert all(l.strip().endswith(" 0") for l in meta), meta
