#!/usr/bin/env python3
"""Development tool: single-query call latency of vsr_search_device (unfiltered, k = 10) for 10k / 100k / 1M x 128 rows,
with and without the library's profiling events.   python tools/latency_probe.py"""
import sys, time, ctypes, numpy as np
sys.path.insert(0, "vectorsearch-rbac_amd")
import torch, vsrbac
from vsrbac.datasets import sift_like_corpus
dev = torch.device("cuda", 0)
ctx = vsrbac.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for n in (10_000, 100_000, 1_000_000):
    x, blk, doc = sift_like_corpus(n, 128)
    c = ctx.load_corpus(x, blk, doc)
    q = torch.from_numpy(x[:64]).to(dev)
    k = 10
    outs = [torch.empty((64, k), dtype=torch.int64, device=dev), torch.empty((64, k), dtype=torch.int32, device=dev), torch.empty((64, k), dtype=torch.int64, device=dev), torch.empty((64, k), dtype=torch.float32, device=dev), torch.empty((64,), dtype=torch.int32, device=dev)]
    for prof in (0, 1):
        ctx.profiling(prof)
        for _ in range(5): c.search_device(p(q[:1]), 1, k, "l2", None, *[p(o) for o in outs])
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(100): c.search_device(p(q[i % 64:i % 64 + 1]), 1, k, "l2", None, *[p(o) for o in outs])
        t_enq = time.perf_counter() - t
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(n, "profiling", prof, "ms/call", round(dt * 10, 4), "enqueue ms/call", round(t_enq * 10, 4), flush=True)
        ctx.stats()
    c.free()
