#!/usr/bin/env python3
"""Development probe: does the scan rate depend on how a filter's documents are laid out in HBM?
Same number of rows, 16 / 64 queries sharing the pass: contiguous documents vs every 25th / 100th document."""
import ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vectorsearch-rbac_amd")):
    sys.path.insert(0, p)
import torch, vsrbac
from vsrbac.datasets import sift_like_corpus
dev = torch.device("cuda", 0)
n, dim, k = 10_000_000, 128, 100
x, blk, doc = sift_like_corpus(n, dim)
ctx = vsrbac.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
corpus = ctx.load_corpus(x, blk, doc)
ndocs = n // 100
ptr = lambda t: ctypes.c_void_p(t.data_ptr())
for nq in (16, 64):
    q = torch.from_numpy(x[:nq].copy()).to(dev)
    outs = [torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), dtype=torch.int32, device=dev),
            torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev),
            torch.empty((nq,), dtype=torch.int32, device=dev)]
    for name, docs in (("contiguous", np.arange(1, 40001)), ("every25th", np.arange(1, ndocs + 1, 25)[:4000].repeat(1)),
                       ("every25th_x10", np.arange(1, ndocs + 1, 2.5).astype(np.int64)[:40000]),
                       ("every100th", np.arange(1, ndocs + 1, 100)[:1000])):
        f = corpus.filter_from_documents(np.unique(docs).astype(np.int32))
        fl = [f] * nq
        for _ in range(3):
            corpus.search_device(ptr(q), nq, k, "l2", fl, *[ptr(o) for o in outs])
        torch.cuda.synchronize()
        ctx.profiling(True); ctx.stats_reset()
        for _ in range(10):
            corpus.search_device(ptr(q), nq, k, "l2", fl, *[ptr(o) for o in outs])
        torch.cuda.synchronize()
        st = ctx.stats(); ctx.profiling(False)
        ms = st["scan_ms"][1] / max(1, st["scan_launches"][1])
        rows = f.scanned_rows
        print(json.dumps({"nq": nq, "layout": name, "rows": rows, "scan_ms": round(ms, 4), "GBs": round(rows * 512 / ms / 1e6, 1),
                          "kernel": ctx.last_scan_kernel()[-40:]}), flush=True)
