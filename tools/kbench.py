#!/usr/bin/env python3
"""Kernel-level sweeps for K1/K5 on one MI355X (development tool; not part of the bench contract).

Each case prints one JSON line: per-class K1 time, algorithmic GB/s, K5 time, wall per step.
  python tools/kbench.py --rows 10000000 --cases full1,full4,role1000
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vectorsearch-rbac_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--cases", default="full1,full4,full16,role1,role16,role1000,bitmap1000")
    ap.add_argument("--budgets", default="0")
    ap.add_argument("--minrows", default="256")
    ap.add_argument("--metric", default="l2")
    ap.add_argument("--gauss", action="store_true")
    args = ap.parse_args()
    import torch
    import vsrbac
    from vsrbac.datasets import gaussian_corpus, sample_queries, sift_like_corpus, tree_rbac

    dev = torch.device("cuda", 0)
    n, dim, k = args.rows, args.dim, args.k
    if args.gauss or dim != 128:
        x, blk, doc = gaussian_corpus(n, dim, normalize=(args.metric == "cosine"), blocks_per_doc=100)
    else:
        x, blk, doc = sift_like_corpus(n, dim)
    rbac = tree_rbac(1000, 100, n // 100)
    ctx = vsrbac.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    corpus = ctx.load_corpus(x, blk, doc)
    corpus.load_rbac(rbac.user_roles, rbac.permissions)
    qrow, quser = sample_queries(1000, n, 1000)
    qall = torch.from_numpy(x[qrow]).to(dev)

    for case in args.cases.split(","):
        kind = case.rstrip("0123456789")
        nq = int(case[len(kind):])
        if kind == "full":
            filters = None
        elif kind == "role":
            filters = [corpus.filter_for_user(int(u), vsrbac.RANGES) for u in quser[:nq]]
        elif kind == "bitmap":
            filters = [corpus.filter_for_user(int(u), vsrbac.BITMAP) for u in quser[:nq]]
        elif kind == "same":           # nq queries of ONE user: maximal pass sharing
            filters = [corpus.filter_for_user(int(quser[0]), vsrbac.RANGES)] * nq
        else:
            raise SystemExit(case)
        d_q = qall[:nq].contiguous()
        outs = [torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), dtype=torch.int32, device=dev),
                torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev),
                torch.empty((nq,), dtype=torch.int32, device=dev)]
        for budget in [int(b) for b in args.budgets.split(",")]:
            for minrows in [int(m) for m in args.minrows.split(",")]:
                ctx.tune(budget, minrows, 0)

                def step():
                    corpus.search_device(ptr(d_q), nq, k, args.metric, filters, *[ptr(o) for o in outs])

                for _ in range(2):
                    step()
                torch.cuda.synchronize()
                ctx.profiling(True)
                ctx.stats_reset()
                t = time.perf_counter()
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t) / args.steps
                st = ctx.stats()
                ctx.profiling(False)
                rec = {"case": case, "budget": budget, "minrows": minrows, "wall_ms": round(wall * 1e3, 4),
                       "qps": round(nq / wall, 1), "select_ms": round(st["select_ms"] / args.steps, 4),
                       "kernel": ctx.last_scan_kernel(), "flagged": ctx.screening_check(0)[0]}
                for c in (0, 1):
                    if st["scan_launches"][c]:
                        ms = st["scan_ms"][c] / st["scan_launches"][c]
                        by = st["scan_bytes"][c] / st["scan_launches"][c]
                        rec[f"scan{c}_ms"] = round(ms, 4)
                        rec[f"scan{c}_GBs"] = round(by / ms / 1e6, 1)
                        rec[f"scan{c}_MB"] = round(by / 1e6, 1)
                print(json.dumps(rec), flush=True)
    corpus.free()
    ctx.close()


if __name__ == "__main__":
    main()
