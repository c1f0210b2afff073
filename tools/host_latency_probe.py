#!/usr/bin/env python3
"""Development tool: vsr_search with host pointers, one RBAC-filtered query per call (the reference harness's call shape,
prefilter_role.py:128-172) on a 2M-row SIFT-like corpus; prints ms per call."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vectorsearch-rbac_amd"))
import vsrbac
from vsrbac.datasets import sift_like_corpus, tree_rbac, sample_queries
n = 2_000_000
x, blk, doc = sift_like_corpus(n, 128, seed=1)
rb = tree_rbac(num_users=1000, num_roles=100, num_docs=n // 100, seed=1)
ctx = vsrbac.Context(0)
c = ctx.load_corpus(x, blk, doc)
c.load_rbac(rb.user_roles, rb.permissions)
qrow, quser = sample_queries(300, n, 1000, seed=3)
fl = [c.filter_for_user(int(u), vsrbac.RANGES) for u in quser]
qs = [np.ascontiguousarray(x[qrow[i]][None, :]) for i in range(300)]
for i in range(20):
    c.search(qs[i], 100, "l2", [fl[i]])
t = time.perf_counter()
for i in range(300):
    c.search(qs[i], 100, "l2", [fl[i]])
print("host one query per call ms", round((time.perf_counter() - t) / 300 * 1e3, 4), ctx.last_scan_kernel())
