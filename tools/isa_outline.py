#!/usr/bin/env python3
"""Development tool: one-line outline of a kernel's ISA (runs of loads / LDS ops / MFMAs / waits) from a hipcc -S file.
  hipcc ... --cuda-device-only -S -o /tmp/l2.s csrc/vsr_scan_l2.hip ; python tools/isa_outline.py /tmp/l2.s <mangled symbol>
"""
import sys

def main():
    path, sym = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
    seq = []
    for l in lines[start + 1:]:
        l = l.strip()
        if l.startswith("s_endpgm"):
            break
        if not l or l.startswith(";") or l.startswith("."):
            if not l.endswith(":"):
                continue
        op = l.split()[0]
        if op.startswith("v_mfma"): t = "MFMA"
        elif op.startswith("global_load"): t = "GLOAD"
        elif op.startswith("global_store"): t = "GSTORE"
        elif op.startswith("global_atomic"): t = "GATOM"
        elif op.startswith("ds_write") or op.startswith("ds_store"): t = "DSW"
        elif op.startswith("ds_read") or op.startswith("ds_load"): t = "DSR"
        elif op.startswith("ds_"): t = "DSATOM"
        elif op.startswith("s_waitcnt"): t = "WAIT[" + " ".join(l.split()[1:]) + "]"
        elif op.startswith("s_barrier"): t = "BARRIER"
        elif op.startswith("s_cbranch") or op.startswith("s_branch"): t = "BR->" + l.split()[-1]
        elif l.endswith(":"): t = "\n" + l
        else: t = "."
        if seq and seq[-1][0] == t and not t.startswith("\n"):
            seq[-1][1] += 1
        else:
            seq.append([t, 1])
    print(" ".join(f"{t}x{n}" if n > 1 else t for t, n in seq))

if __name__ == "__main__":
    main()
