#!/usr/bin/env python3
"""development helper: per basic block of every kernel in a .s file, counts of the instruction kinds that matter
(usage: tools/isa_blocks.py file.s [kernel-substring])"""
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
want = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for si, s in enumerate(starts):
    if want not in lines[s]:
        continue
    e = starts[si + 1] if si + 1 < len(starts) else len(lines)
    body = lines[s:e]
    print(lines[s], len(body))
    blocks, cur, name = [], [], "entry"
    for l in body:
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append((name, cur))
            cur, name = [], l
        else:
            cur.append(l)
    blocks.append((name, cur))
    for name, b in blocks:
        c = lambda pat: sum(bool(re.search(pat, x)) for x in b)
        n_gload, n_valu = c(r"global_load_(?!lds)"), c(r"^\s+v_(?!mfma)")
        if c("v_mfma") or c("scratch_") or c("global_load_lds") or c("s_barrier"):
            print(f"  {name:12s} insts {len(b):5d} mfma {c('v_mfma'):4d} scratch {c('scratch_'):3d} glds {c('global_load_lds'):3d} "
                  f"gload {n_gload:3d} ds_read {c('ds_read'):3d} ds_write {c('ds_write'):3d} waitcnt {c('s_waitcnt'):3d} "
                  f"barrier {c('s_barrier'):2d} valu {n_valu:4d} atomics {c('atomic'):2d}")
