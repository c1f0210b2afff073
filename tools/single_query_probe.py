#!/usr/bin/env python3
"""Development tool: the harness's call shape (one RBAC-filtered query per call, back to back on one stream) on the bench
corpus; prints ms per query.  Under `rocprofv3 --kernel-trace` the trace shows the kernels of a call:
    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/single_query_probe.py
    python3 tools/single_query_probe.py --summarize out"""
import argparse, collections, csv, ctypes, glob, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vectorsearch-rbac_amd"))
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--k", type=int, default=100)
ap.add_argument("--calls", type=int, default=200)
ap.add_argument("--summarize")
a = ap.parse_args()
if a.summarize:
    f = glob.glob(a.summarize + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    tail = rows[-40:]
    t0 = int(tail[0]["Start_Timestamp"]); prev = t0
    for r in tail:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} gap {(s - prev) / 1e3:6.1f} grid={r['Grid_Size_X']:>8s} {r['Kernel_Name'][:70]}")
        prev = e
    sys.exit(0)
import torch, vsrbac
from vsrbac.datasets import sample_queries, sift_like_corpus, sift_like_rows_at, tree_rbac
seed = 20251121
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = vsrbac.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.set_query_hint(True)          # the queries are corpus rows: integers 0..255 (bench.py gives the same promise)
x, blk, doc = sift_like_corpus(a.rows, 128, seed=seed)
rbac = tree_rbac(num_users=1000, num_roles=100, num_docs=a.rows // 100, seed=seed)
c = ctx.load_corpus(x, blk, doc); c.load_rbac(rbac.user_roles, rbac.permissions)
qrow, quser = sample_queries(a.calls, a.rows, 1000, seed=seed)
q = torch.from_numpy(sift_like_rows_at(qrow, 128, seed)).to(dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
k = a.k
o = [torch.empty((1, k), dtype=torch.int64, device=dev), torch.empty((1, k), dtype=torch.int32, device=dev),
     torch.empty((1, k), dtype=torch.int64, device=dev), torch.empty((1, k), dtype=torch.float32, device=dev),
     torch.empty((1,), dtype=torch.int32, device=dev)]
fl = [c.pack_filters([c.filter_for_user(int(u), vsrbac.RANGES)]) for u in quser]
for i in range(8):
    c.search_device(p(q[i:i + 1]), 1, k, "l2", fl[i], *[p(t) for t in o])
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(a.calls):
    c.search_device(p(q[i:i + 1]), 1, k, "l2", fl[i], *[p(t) for t in o])
t_enq = time.perf_counter() - t
torch.cuda.synchronize()
dt = time.perf_counter() - t
vis = [int(len(rbac.visible_docs(int(u)))) * 100 for u in quser]
print(f"ms/query {dt / a.calls * 1e3:.4f}  host enqueue ms/query {t_enq / a.calls * 1e3:.4f}  kernel {ctx.last_scan_kernel()}  "
      f"visible rows mean {np.mean(vis):.0f} max {max(vis)}", flush=True)
