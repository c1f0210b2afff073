#!/usr/bin/env python3
"""Development tool: skeleton (loads, waits, LDS ops, branches, MFMA runs) of a kernel's MFMA loop from a hipcc -S file."""
import sys
lines = open(sys.argv[1]).read().split('\n')
sym = sys.argv[2] + ':'
before = int(sys.argv[3]) if len(sys.argv) > 3 else 200
st = next(i for i, l in enumerate(lines) if l.startswith(sym))
en = next(i for i in range(st, len(lines)) if 's_endpgm' in lines[i])
body = lines[st:en]
idx = [i for i, l in enumerate(body) if 'v_mfma' in l]
print(len(body), "lines,", len(idx), "mfma, first/last at", idx[0], idx[-1])
keep = ('v_mfma', 'global_load', 's_waitcnt', 'ds_write', 'ds_read', 's_barrier', 's_cbranch', 's_branch', 'global_store',
        'global_atomic', 'ds_add', '.LBB', 'scratch_')
res, run = [], 0
for i in range(max(0, idx[0] - before), min(len(body), idx[-1] + 40)):
    l = body[i].strip()
    if not any(k in l for k in keep):
        continue
    if 'v_mfma' in l:
        run += 1
        continue
    if run:
        res.append(f"   [MFMA x{run}]")
        run = 0
    res.append(f"{i}: {l[:90]}")
if run:
    res.append(f"   [MFMA x{run}]")
print('\n'.join(res))
