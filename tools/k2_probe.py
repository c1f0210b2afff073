#!/usr/bin/env python3
"""Tiny K2 probe: a handful of shared-pass searches on a small corpus, compared with numpy (development tool)."""
import os, sys, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vectorsearch-rbac_amd")]
faulthandler.dump_traceback_later(20, exit=True)
import numpy as np
import torch  # noqa: F401  (HIP runtime order)
import vsrbac
n = int(os.environ.get("N", "5000")); d = int(os.environ.get("D", "128")); nq = int(os.environ.get("NQ", "5")); k = 10
rng = np.random.default_rng(0)
x = np.clip(np.rint(np.abs(rng.normal(0, 45, (n, d)))), 0, 255).astype(np.float32)
ctx = vsrbac.Context(0)
c = ctx.load_corpus(x)
print("loaded", flush=True)
res = c.search(x[:nq], k, "l2")
print("searched", flush=True)
ref = ((x[None, :, :] - x[:nq, None, :]) ** 2).sum(-1)
want = np.argsort(ref, axis=1, kind="stable")[:, :k]
print("match", (res.rows == want).all(), ctx.screening_check(0), flush=True)
