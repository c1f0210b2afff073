#!/usr/bin/env python3
"""bench.py — QPS of RBAC-filtered exact k-NN on MI355X (BASELINE.json's metric), with the roofline of the dominant
scan kernel and a CPU baseline timed beside it.

  python bench.py [--gpus N --steps K --warmup W]          (N > 1: spawns its own ranks through torch.distributed.run)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config.workload): synthetic SIFT10M-like corpus (10M x 128 fp32, integer-valued, 100 rows per document),
tree RBAC (1000 users / 100 roles), k = 100, L2.  A step = one batch of `--queries` queries (uniform rows x
uniform users, a fresh batch per step from the seeded stream) through the whole hot path: staging, threshold sample,
shared-pass scan on the matrix cores (distance + permission + running top-k), selection, exact re-rank.  Two legs are
timed with the same K steps each:

  value / roofline      role-level filter applied as a PRE-filter: only the rows of the user's role partition are
                        scanned (BASELINE config 2's mode on the metric's SIFT10M corpus)
  postfilter{...}       the same users as row-level-security bitmaps tested in the distance loop (BASELINE config 4's
                        mode: RLS post-filter), whole-corpus scan order

With N > 1 the corpus is sharded by contiguous row range (strong scaling: total work fixed), every rank searches its
shard and the per-rank top-k lists are all-gathered over RCCL and merged on the GPU.  Inputs (corpus, filters,
queries) are resident in HBM when the timed region starts.

roofline: `achieved` = the rows the main scan launch covers, each counted ONCE, in the layout the kernel reads (K2w:
the bf16 screening planes, 2 or 4 bytes per element, + 4 bytes of |row|^2; K1 / K2: the fp32 rows) / the launch's
HIP-event duration, against 8 TB/s; `bound` is "mfma" only if the matrix-core floor (the products the kernel issues at
the dense peak of their type) is the larger one.  `fp32_equivalent` prices the same rows at SURVEY 8(d)'s 4 bytes per
element: a rate for comparison with an fp32 scan, not an HBM fraction.  `traffic` = HBM bytes per launch from a
rocprofv3 PMC pass of this command (profiles/), when one is on file for the workload.

No part of the timed path touches the CPU oracle; it is used only by the cpu_baseline leg (rank 0, N = 1) and for the
parity spot-check of the GPU results on the same sample.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "vectorsearch-rbac_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling ~6290
MFMA_F32_PEAK_TF = 157.3       # fp32 matrix peak (v_mfma_f32_16x16x4_f32), same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--queries", type=int, default=1000, help="queries per step")
    ap.add_argument("--legs", default="prefilter,postfilter", help="timed legs (the first one is the headline)")
    ap.add_argument("--batches", type=int, default=16, help="distinct query batches cycled through the steps")
    ap.add_argument("--sustained-s", type=float, default=2.0, help="extra untimed-by-the-driver leg: seconds of steps")
    ap.add_argument("--cpu-queries", type=int, default=1000, help="sample size of the CPU baseline leg")
    ap.add_argument("--cpu-reps", type=int, default=2, help="repetitions of the CPU sample (10-30 s of CPU work)")
    ap.add_argument("--traffic", default=os.path.join(ROOT, "profiles", "r3", "traffic.json"),
                    help="PMC-derived HBM bytes per launch (tools/pmc_traffic.py) for roofline.traffic")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bf16-line", action="store_true", help="skip the VSR_NO_INT8 sibling record of the headline leg")
    ap.add_argument("--sharding", default="auto", choices=["auto", "placement", "rows"],
                    help="N > 1: 'placement' = whole role partitions on GPUs, a query touches ONE GPU, no exchange (SURVEY 8e-ii); "
                         "'rows' = contiguous row ranges, every query on every GPU, all-gather + merge (SURVEY 8e-i); auto = placement")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1 under role placement: weak = N x 1000 queries per step (every GPU answers ~1000, the N = 1 step's "
                         "work; default), strong = the same 1000-query step as N = 1 spread over the GPUs")
    ap.add_argument("--wiki-rows", type=int, default=5_000_000, help="rows of the 768-d legs' corpus (0: skip the legs)")
    ap.add_argument("--wiki-steps", type=int, default=5)
    ap.add_argument("--index-rows", type=int, default=120_000, help="rows of the role partition the HNSW legs (CPU port and K4) index")
    ap.add_argument("--ivf-rows", type=int, default=1_000_000, help="rows of the IVFFlat leg's corpus (0: skip the leg)")
    ap.add_argument("--seed", type=int, default=20251121)
    return ap.parse_args()


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def self_launch(args):
    """--gpus N > 1 without a rank environment: spawn the N ranks as a CHILD (never re-exec a process that may touch
    the GPU), relay rank 0's JSON line through the inherited stdout, exit with the child's code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    sys.exit(subprocess.run(cmd).returncode)


MFMA_BF16_PEAK_TF = 2500.0      # dense bf16 MFMA peak (MI355X_MICROARCH.md); the fp32 MFMA peak is MFMA_F32_PEAK_TF
MFMA_I8_PEAK_TOPS = 5000.0      # dense int8 MFMA peak, same guide (ops, reported in the TFLOP/s fields)


def kernel_layout(kernel, dim):
    """(bytes per row the launch reads, MFMA flops per (row, query) pair it issues, MFMA peak) for the launched kernel."""
    if "K2g" in kernel:                  # coarse bf16 planes (hi only), rows padded to whole 64-element K-steps; one product
        d_pad = -(-dim // 64) * 64
        return d_pad * 2 + 4, 2 * d_pad, MFMA_BF16_PEAK_TF
    if "K2w" in kernel or "K2i" in kernel:
        if "int8" in kernel:             # int8 planes (x - 128), 128 elements per row; one product, exact
            d_pad = -(-dim // 128) * 128
            return d_pad + 4, 2 * d_pad, MFMA_I8_PEAK_TOPS
        if "HO=true" in kernel:          # hi-only bf16 planes, rows padded to whole 128-element stages; products xh*qh + xh*qm
            d_pad = -(-dim // 128) * 128
            return d_pad * 2 + 4, 2 * 2 * d_pad, MFMA_BF16_PEAK_TF
        d_pad = -(-dim // 64) * 64       # hi + mid planes; products xh*qh + xh*qm + xm*qh
        return d_pad * 4 + 4, 3 * 2 * d_pad, MFMA_BF16_PEAK_TF
    d_pad = -(-dim // 4) * 4             # fp32 rows (K1 / K1m: VALU; K2: fp32 MFMA)
    return d_pad * 4 + (4 if "K2" in kernel else 0), 2 * d_pad, MFMA_F32_PEAK_TF


def roofline_of(st, dim, kernel, n_sess, alone=None):
    """Floors of the dominant scan launch class from the library's statistics (HIP events on the launch stream).
    `achieved` counts every row the launch scans ONCE, in the layout the kernel actually reads (the screening planes of
    K2w are 2 or 4 bytes per element + 4 bytes of |row|^2): it cannot exceed what HBM delivers.  The same rows priced at
    SURVEY 8(d)'s 4 bytes per element are reported beside it as `fp32_equivalent` (a rate, not an HBM fraction)."""
    cls = int(np.argmax(st["scan_ms"]))
    launches = max(1, st["scan_launches"][cls])
    ms = st["scan_ms"][cls] / launches
    pairs = st["scan_pairs"][cls] / launches
    rows = st["unique_rows"][cls] / launches
    row_bytes, pair_flops, mfma_peak = kernel_layout(kernel, dim)
    unique_bytes = rows * row_bytes
    fp32_bytes = rows * dim * 4
    flops = pair_flops * pairs
    t_hbm, t_mfma = unique_bytes / (HBM_PEAK_GBS * 1e9), flops / (mfma_peak * 1e12)
    sec = ms * 1e-3

    def floors(seconds):
        if t_hbm >= t_mfma:
            return "hbm", unique_bytes / seconds / 1e9, HBM_PEAK_GBS, "GB/s"
        return "mfma", flops / seconds / 1e12, mfma_peak, "TFLOP/s"

    bound, achieved, peak, unit = floors(sec) if sec > 0 else ("hbm", 0.0, HBM_PEAK_GBS, "GB/s")
    r = {"bound": bound, "achieved": round(achieved, 2), "peak": peak, "unit": unit, "frac": round(achieved / peak, 4),
         "traffic": None, "kernel": kernel, "launch_ms": round(ms, 4), "launches": int(launches),
         "unique_rows": int(rows), "row_bytes": row_bytes, "unique_bytes": int(unique_bytes),
         "floor_ms": {"hbm_8TBs": round(t_hbm * 1e3, 4), f"mfma_{mfma_peak:g}TF": round(t_mfma * 1e3, 4)},
         "mfma": {"flops_issued": int(flops), "peak_tflops": mfma_peak,
                  "frac": round(flops / sec / 1e12 / mfma_peak, 4) if sec > 0 else 0.0},
         "fp32_equivalent": {"unique_bytes": int(fp32_bytes), "rate_gbs": round(fp32_bytes / sec / 1e9, 1) if sec > 0 else 0.0,
                             "flops": int(2.0 * dim * pairs),
                             "note": "the same rows at 4 bytes per element (SURVEY 8d) and 2*d flops per pair: what an "
                                     "fp32 scan would have to move / compute for this step; not an HBM fraction"},
         "pass_rows_per_launch": int(st["scan_rows"][cls] / launches),
         "all_scan_ms": [round(v, 3) for v in st["scan_ms"]]}
    if n_sess > 1:
        r["note"] = (f"{n_sess} batches in flight: a launch's event-timed duration includes the time it shares the GPU "
                     "with the other batch's kernels; `alone` is the same launch with one batch in flight")
    if alone and alone["scan_launches"][cls]:
        a_ms = alone["scan_ms"][cls] / alone["scan_launches"][cls]
        _, a_ach, a_peak, _ = floors(a_ms * 1e-3)
        r["alone"] = {"launch_ms": round(a_ms, 4), "achieved": round(a_ach, 2), "frac": round(a_ach / a_peak, 4),
                      "launches": int(alone["scan_launches"][cls])}
    return r


def start_cpu_hnsw_build(orc, x, rows_idx, seed):
    """pgvector's HNSW build restated (oracle/vsr_index_oracle.c, serial like a backend's in-memory build) on a host thread
    while the GPU legs run: the C call releases the GIL, so the build costs the bench no wall time."""
    import threading
    box = {}

    def run():
        from oracle.oracle import HnswIndex
        t = time.perf_counter()
        try:
            box["sub"] = np.ascontiguousarray(x[rows_idx])
            box["index"] = HnswIndex(orc, "l2", box["sub"], m=16, ef_construction=64, seed=seed)
        except Exception as exc:                                     # reported, never required
            box["error"] = repr(exc)
        box["build_s"] = time.perf_counter() - t

    box["thread"] = threading.Thread(target=run, daemon=True)
    box["thread"].start()
    return box


def host_threads():
    try:
        return max(1, min(64, len(os.sched_getaffinity(0))))
    except AttributeError:
        return max(1, min(64, os.cpu_count() or 1))


def hnsw_legs(args, torch, ctx, orc, box, qvec, k, dev):
    """cpu_baseline.hnsw and the `hnsw` GPU leg on the SAME graph: pgvector's HNSW with the index parameters of the
    reference's role-partition experiment (m = 16, ef_construction = 64; test_partition_prefilter_by_role.py:42-46) over
    one role partition.  ef_search is swept upwards until recall@k against the exact scan of the partition reaches 0.95."""
    import concurrent.futures
    box["thread"].join()
    if "error" in box:
        return {"error": box["error"]}, {"error": box["error"]}
    hidx, sub = box["index"], box["sub"]
    dim = sub.shape[1]
    hq = qvec[:100]
    exact = [set(orc.filtered_topk("l2", sub, hq[i], k)[0].tolist()) for i in range(len(hq))]
    recall_of = lambda got: float(np.mean([len(set(np.asarray(g).tolist()) & e) / max(1, len(e)) for g, e in zip(got, exact)]))
    sweep, cpu_rows = [], {}
    for ef in (40, 100, 200, 400, 800, 1600, 3200):
        th = time.perf_counter()
        got = [hidx.search(hq[i], ef)[0][:k] for i in range(len(hq))]
        hs = time.perf_counter() - th
        sweep.append({"ef_search": ef, "qps": round(len(hq) / hs, 1), "recall_at_k": round(recall_of(got), 4)})
        cpu_rows[ef] = got[:10]
        if sweep[-1]["recall_at_k"] >= 0.95:
            break
    ef_star = sweep[-1]["ef_search"]
    gq = qvec[:1000]
    threads = host_threads()
    cuts = np.linspace(0, len(gq), threads + 1).astype(int)
    ta = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(threads) as pool:
        list(pool.map(lambda se: [hidx.search(gq[i], ef_star) for i in range(se[0], se[1])],
                      [(int(a), int(b)) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]))
    all_s = time.perf_counter() - ta
    cpu = {"kind": "port", "cores": 1, "m": 16, "ef_construction": 64, "rows": int(len(sub)), "build_s": round(box["build_s"], 2),
           "sweep": sweep, "recall_target": 0.95, "reached": bool(sweep[-1]["recall_at_k"] >= 0.95),
           "all_cores": {"ef_search": ef_star, "value": round(len(gq) / all_s, 1), "unit": "queries/s", "threads": threads,
                         "queries": int(len(gq))},
           "sample": f"pgvector's HNSW restated (oracle/vsr_index_oracle.c; built on a host thread during the GPU legs), "
                     f"{len(sub)} rows of one role partition, 100 queries per ef_search on one core, recall against the exact "
                     f"scan of those rows; all_cores: 1000 queries fanned over host threads at the last ef_search"}
    # the same graph on the GPU (K4): queries and results resident, one launch per 1000 queries
    try:
        c2 = ctx.load_corpus(sub)
        gidx = c2.load_hnsw(hidx.export())
        nq = len(gq)
        d_q = torch.from_numpy(np.ascontiguousarray(gq)).to(dev)
        o_blk = torch.empty((nq, k), dtype=torch.int64, device=dev)
        o_doc = torch.empty((nq, k), dtype=torch.int32, device=dev)
        o_row = torch.empty((nq, k), dtype=torch.int64, device=dev)
        o_dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
        o_cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
        o_vis = torch.zeros((nq,), dtype=torch.int64, device=dev)
        gsweep = []
        for pt in sweep:
            ef = pt["ef_search"]
            call = lambda: gidx.search_device(ptr(d_q), nq, k, ef, "l2", None, ptr(o_blk), ptr(o_doc), ptr(o_row), ptr(o_dist),
                                              ptr(o_cnt), ptr(o_vis))
            call()
            ctx.synchronize()
            reps = 5
            th = time.perf_counter()
            for _ in range(reps):
                call()
            ctx.synchronize()
            gs = (time.perf_counter() - th) / reps
            rows_g, cnt_g = o_row.cpu().numpy(), o_cnt.cpu().numpy()
            if (cnt_g < 0).any():                                  # the LDS visited set overflowed: the host form re-runs those
                res, _ = gidx.search(gq, k, ef)
                rows_g, cnt_g = res.rows, res.counts
            same = all(np.array_equal(rows_g[i][:len(cpu_rows[ef][i])], cpu_rows[ef][i]) for i in range(10))
            visited = int(o_vis.sum().item())
            gbytes = visited * dim * 4
            gsweep.append({"ef_search": ef, "qps": round(nq / gs, 1), "ms_per_call": round(gs * 1e3, 4),
                           "recall_at_k": round(recall_of([rows_g[i][:int(cnt_g[i])] for i in range(len(hq))]), 4),
                           "same_rows_as_cpu_port": bool(same), "visited_per_query": round(visited / nq, 1),
                           "queries_rerun_with_global_visited_set": int((o_cnt.cpu().numpy() < 0).sum()),
                           "roofline": {"bound": "hbm", "achieved": round(gbytes / gs / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": round(gbytes / gs / 1e9 / HBM_PEAK_GBS, 4), "bytes": int(gbytes)},
                           "speedup_vs_cpu_one_core": round(nq / gs / pt["qps"], 1)})
        gpu = {"kernel": "vsr::hnsw_search_kernel (K4)", "rows": int(len(sub)), "queries_per_call": int(nq), "sweep": gsweep,
               "speedup_vs_cpu_all_cores_at_last_ef": round(gsweep[-1]["qps"] / cpu["all_cores"]["value"], 1),
               "note": "the graph of cpu_baseline.hnsw searched by K4 through vsr_hnsw_search_device (queries and results "
                       "resident, one launch per call).  roofline: the row bytes of every distance evaluation (visited "
                       "elements x d x 4; SURVEY 8d's gather figure) over the launch time -- a graph walk is a chain of "
                       "dependent gathers, latency- not bandwidth-bound, so the fraction is small by nature"}
        # BASELINE config 5's mode on the same graph: a predicate that admits ~10 % of the rows (every 10th document of 100
        # rows), the walk that applies it while it walks (vsr_hnsw_set_predicate_aware) beside the plain walk whose results are
        # filtered, both against the exact filtered scan of the oracle
        try:
            import vsrbac as _v
            pmask = (((np.arange(len(sub)) // 100) % 10) == 3).astype(np.uint8)
            pf = c2.filter_from_bytemask(pmask, _v.BITMAP)
            pfl = c2.pack_filters([pf] * nq)
            pexact = [set(orc.filtered_topk("l2", sub, hq[i], k, None, None, pmask)[0].tolist()) for i in range(len(hq))]
            prec = lambda got: float(np.mean([len(set(np.asarray(g).tolist()) & e) / max(1, len(e)) for g, e in zip(got, pexact)]))
            psweep = []
            for ef in (100, 400):
                pt = {"ef_search": ef}
                for aware in (False, True):
                    gidx.set_predicate_aware(aware)
                    call = lambda: gidx.search_device(ptr(d_q), nq, k, ef, "l2", pfl, ptr(o_blk), ptr(o_doc), ptr(o_row), ptr(o_dist),
                                                      ptr(o_cnt), ptr(o_vis))
                    call()
                    ctx.synchronize()
                    th = time.perf_counter()
                    for _ in range(3):
                        call()
                    ctx.synchronize()
                    gs = (time.perf_counter() - th) / 3
                    rows_g, cnt_g = o_row.cpu().numpy(), o_cnt.cpu().numpy()
                    if (cnt_g < 0).any():
                        res, _ = gidx.search(gq, k, ef, "l2", [pf] * nq)
                        rows_g, cnt_g = res.rows, res.counts
                    pt["predicate_aware" if aware else "plain_walk_then_filter"] = {
                        "qps": round(nq / gs, 1), "recall_at_k": round(prec([rows_g[i][:int(cnt_g[i])] for i in range(len(hq))]), 4),
                        "rows_returned_per_query": round(float(np.maximum(cnt_g, 0).mean()), 1)}
                psweep.append(pt)
            gidx.set_predicate_aware(False)
            gpu["predicate_aware"] = {"permitted_fraction": round(float(pmask.mean()), 3), "sweep": psweep,
                                      "note": "K4 with the permission bitmap applied inside the layer-0 walk (ACORN-1 style two-hop "
                                              "expansion; parity pinned by the index oracle's restatement, ACORN's own source is not in "
                                              "the reference tree) against the exact filtered scan; the plain walk returns what is left "
                                              "of its ef_search candidates after the filter (pgvector's behaviour)"}
        except Exception as exc:
            gpu["predicate_aware"] = {"error": repr(exc)}
        gidx.free()
        # CREATE INDEX on the GPU (vsr_hnsw_build: batched insertion) over the same rows: build time beside the CPU port's, and
        # the same ef_search sweep on the graph it leaves (another graph than the serial build's: recall is the parity measure)
        try:
            tb = time.perf_counter()
            bidx = c2.build_hnsw(16, 64, "l2", seed=args.seed)
            gb_s = time.perf_counter() - tb
            bsweep = []
            for pt in sweep:
                ef = pt["ef_search"]
                call = lambda: bidx.search_device(ptr(d_q), nq, k, ef, "l2", None, ptr(o_blk), ptr(o_doc), ptr(o_row), ptr(o_dist),
                                                  ptr(o_cnt), ptr(o_vis))
                call()
                ctx.synchronize()
                th = time.perf_counter()
                for _ in range(3):
                    call()
                ctx.synchronize()
                gs = (time.perf_counter() - th) / 3
                rows_g, cnt_g = o_row.cpu().numpy(), o_cnt.cpu().numpy()
                if (cnt_g < 0).any():
                    res, _ = bidx.search(gq, k, ef)
                    rows_g, cnt_g = res.rows, res.counts
                bsweep.append({"ef_search": ef, "qps": round(nq / gs, 1),
                               "recall_at_k": round(recall_of([rows_g[i][:int(cnt_g[i])] for i in range(len(hq))]), 4),
                               "recall_of_cpu_built_graph": pt["recall_at_k"]})
            gpu["gpu_built_graph"] = {"build_s": round(gb_s, 2), "cpu_port_build_s": cpu["build_s"],
                                      "build_speedup": round(cpu["build_s"] / max(gb_s, 1e-9), 1), "sweep": bsweep,
                                      "note": "vsr_hnsw_build: batches of at most 1/8 of the graph, one wave per new element; same m, "
                                              "ef_construction and level stream as the CPU port"}
            bidx.free()
        except Exception as exc:
            gpu["gpu_built_graph"] = {"error": repr(exc)}
        c2.free()
    except Exception as exc:
        gpu = {"error": repr(exc)}
    return cpu, gpu


def ivf_leg(args, torch, vsrbac, ctx, orc, x, blk, doc, qvec, k, dev, rows=None, lists=None, probes_list=(1, 4, 10, 32)):
    """The `ivf` leg: CREATE INDEX ... USING ivfflat on the GPU (k-means on the sample, every row into its list), then
    probes in {1, 4, 10, 32} through vsr_ivf_search_device; beside it pgvector's scan of the same lists on one core.
    rows / lists / probes_list: another index shape (the reference's own: lists = 100, ivfflat.probes = 5)."""
    n3 = min(int(rows if rows is not None else args.ivf_rows), len(x))
    lists = int(lists) if lists is not None else max(10, n3 // 1000)   # pgvector's guidance: rows / 1000 up to 1M rows
    x3 = x[:n3]
    dim = x3.shape[1]
    c3 = ctx.load_corpus(x3, blk[:n3], doc[:n3])
    tb = time.perf_counter()
    want = max(lists * 50, 10000)
    pick = np.sort(np.random.default_rng(args.seed).choice(n3, size=min(n3, want), replace=False))
    centers, iters = ctx.ivf_kmeans(x3[pick], lists, "l2", args.seed)
    t_km = time.perf_counter() - tb
    row_list = c3.ivf_assign(centers, "l2")
    t_as = time.perf_counter() - tb - t_km
    gidx = c3.load_ivf(centers, row_list)
    t_ld = time.perf_counter() - tb - t_km - t_as
    nq = min(1000, len(qvec))
    gq = np.ascontiguousarray(qvec[:nq])
    d_q = torch.from_numpy(gq).to(dev)
    o_blk = torch.empty((nq, k), dtype=torch.int64, device=dev)
    o_doc = torch.empty((nq, k), dtype=torch.int32, device=dev)
    o_row = torch.empty((nq, k), dtype=torch.int64, device=dev)
    o_dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    o_cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    m = 50
    exact = c3.search(gq[:m], k, "l2").rows
    # the CPU side scans list-ordered rows (GetScanItems reads the probed lists' pages, nothing else)
    order = np.argsort(row_list, kind="stable")
    start = np.concatenate([[0], np.cumsum(np.bincount(row_list, minlength=lists))]).astype(np.int64)
    x_lo = np.ascontiguousarray(x3[order])
    sweep = []
    for probes in probes_list:
        if probes > lists:
            break
        call = lambda: gidx.search_device(ptr(d_q), nq, k, probes, "l2", None, ptr(o_blk), ptr(o_doc), ptr(o_row), ptr(o_dist),
                                          ptr(o_cnt))
        call()
        ctx.profiling(2)
        ctx.stats_reset()
        reps = 3
        th = time.perf_counter()
        for _ in range(reps):
            call()
        gs = (time.perf_counter() - th) / reps
        st = ctx.stats()
        ctx.profiling(False)
        kern = ctx.last_scan_kernel()
        rows_g, dist_g = o_row.cpu().numpy(), o_dist.cpu().numpy()
        rec = float(np.mean([len(set(rows_g[i].tolist()) & set(exact[i].tolist())) / k for i in range(m)]))
        pl = gidx.probe(gq[:m], probes, "l2")
        tc = time.perf_counter()
        rows_o, dist_o, _ = orc.search_ranges("l2", x_lo, gq[:m], k, [[(int(start[l]), int(start[l + 1] - start[l])) for l in pl[i]]
                                                                      for i in range(m)])
        cs = time.perf_counter() - tc
        # equal distances are frequent on integer data and the two sides break ties at the k-th place differently (position in
        # the list-ordered copy vs (document, block) rank): the distances must be identical, the ids below the k-th distance too
        dist_o32 = dist_o.astype(np.float32)[:, :k]
        strict = lambda ids, dd: set(ids[(ids >= 0) & (dd < dd[-1])].tolist())
        same = bool(np.array_equal(dist_o32, dist_g[:m]) and
                    all(strict(order[np.maximum(rows_o[i], 0)] * (rows_o[i] >= 0) - (rows_o[i] < 0), dist_o32[i]) == strict(rows_g[i], dist_g[i])
                        for i in range(m)))
        roof = roofline_of(st, dim, kern, 1) if sum(st["scan_launches"]) else None
        sweep.append({"probes": probes, "qps": round(nq / gs, 1), "ms_per_call": round(gs * 1e3, 3), "recall_at_k": round(rec, 4),
                      "cpu_port_qps_one_core": round(m / cs, 1), "same_rows_and_distances_as_cpu_port": same,
                      "speedup_vs_cpu_one_core": round(nq / gs / (m / cs), 1), "roofline": roof})
    out = {"kernel": "vsr::ivf_probe_kernel (K3) + the list scan the planner picks", "rows": int(n3), "lists": int(lists),
           "queries_per_call": int(nq),
           "build": {"kmeans_s": round(t_km, 2), "kmeans_iterations": int(iters), "samples": int(len(pick)), "assign_s": round(t_as, 2),
                     "load_s": round(t_ld, 2), "note": "vsr_ivf_kmeans + vsr_ivf_assign + vsr_ivf_load (ivfbuild.c / ivfkmeans.c on the GPU)"},
           "sweep": sweep,
           "note": "vsr_ivf_search_device: queries and results resident; the probed list ids (nq x probes x 4 bytes) go to the host "
                   "planner, which groups queries by list so a list shared by many queries is streamed once.  roofline: the dominant "
                   "scan launch class of the call (event-timed), rows counted once.  CPU port: pgvector's GetScanItems over the same "
                   "lists (oracle, one core, 50 queries)"}
    gidx.free()
    c3.free()
    return out


def place_roles(parent, weights, n_bins):
    """Roles -> GPUs for the placement mode: the role tree in PRE-ORDER (children in id order: the order the generator
    consumed the roles in) cut into n_bins contiguous runs of nearly equal weight.  Neighbours in pre-order share their
    ancestors, so the permission classes near the root are replicated on few GPUs (an LPT packing of single roles balances
    as well but scatters siblings: 1.8 x the corpus resident at 8 GPUs against ~1.2 x), and a run's weight misses the
    average by at most one role.  Returns ({role: gpu}, per-GPU weight)."""
    kids = {}
    for r, p in parent.items():
        kids.setdefault(p, []).append(r)
    order, stack = [], [0]
    while stack:
        node = stack.pop()
        if node:
            order.append(node)
        stack.extend(sorted(kids.get(node, []), reverse=True))
    order = [r for r in order if r in weights] + sorted(r for r in weights if r not in parent)
    total = float(sum(weights[r] for r in order)) or 1.0
    where, load, acc, g = {}, [0.0] * n_bins, 0.0, 0
    for r in order:
        # the role goes to the next GPU once this one has reached its share (at least one role per GPU while roles last)
        while g < n_bins - 1 and acc + 0.5 * weights[r] > (g + 1) * total / n_bins:
            g += 1
        where[r] = g
        load[g] += weights[r]
        acc += weights[r]
    return where, load


def draw_rank_queries(sample_queries, where, role_of, me, nq, n, seed, mult, count, seed0):
    """`count` steps' queries of rank `me` under role placement: a step is mult x nq queries drawn exactly like the N = 1
    step's (mult draws of nq, consecutive seeds); the rank keeps those whose user's role it hosts.  mult = N: weak scaling
    (about nq per rank and step); mult = 1: strong scaling (the N = 1 step spread over the ranks).  Every query of a step
    belongs to exactly one rank."""
    made = []
    for b in range(count):
        qr, qu = [], []
        for j in range(mult):
            qrow, quser = sample_queries(nq, n, 1000, seed=seed + 1000 * (seed0 + b * mult + j))
            mine = np.flatnonzero(np.array([where[role_of[int(u)]] == me for u in quser]))
            qr.append(qrow[mine])
            qu.append(quser[mine])
        made.append((np.concatenate(qr), np.concatenate(qu)))
    return made


def placement_bench(args, torch, dist, vsrbac, rank, local_rank, world, dev, sim_world, rehearsal):
    """N > 1, role-partition placement (controller/dynamic_partition/search.py:54-58 searches exactly the
    partition tables of a user's role combination): every ROLE lives on one GPU together with everything it can see (its
    own permission class and its ancestors': the few classes near the root are replicated on the GPUs that host a
    descendant), so a query is answered by ONE GPU from its resident rows and there is no data-path collective at all.
    Weak scaling (default): a step is N x 1000 queries drawn like the N = 1 step's, each rank searches those of its roles
    (about 1000: the N = 1 step's work on every GPU), value = N x 1000 x steps / the slowest rank's time.  --scaling strong:
    the step is the same 1000-query batch as at N = 1.  The other mode is reported as a sibling record either way."""
    from vsrbac.datasets import sample_queries, sift_like_rows_at, tree_rbac
    n, dim, k, nq = args.rows, args.dim, args.k, args.queries
    parts = world if world > 1 else sim_world
    legs = [m for m in args.legs.split(",") if m]
    t0 = time.time()
    rbac = tree_rbac(num_users=1000, num_roles=100, num_docs=n // 100, seed=args.seed)
    role_of = {int(u): rs[0] for u, rs in rbac._user_roles_map.items()}        # tree RBAC: exactly one role per user
    users_of = {}
    for u, r in role_of.items():
        users_of[r] = users_of.get(r, 0) + 1
    # expected work a role attracts: its share of the queries x the rows each of them scans
    weights = {r: users_of.get(r, 0) * len(d) for r, d in rbac.role_docs.items()}
    where, load = place_roles(rbac.parent, weights, parts)
    me = rank if world > 1 else int(os.environ.get("VSR_BENCH_SIM_RANK", str(int(np.argmax(load)))))
    my_roles = sorted(r for r, g in where.items() if g == me)
    my_docs = np.unique(np.concatenate([rbac.role_docs[r] for r in my_roles])).astype(np.int64)
    rows_idx = (np.repeat((my_docs - 1) * 100, 100) + np.tile(np.arange(100), my_docs.size)).astype(np.int64)
    x = sift_like_rows_at(rows_idx, dim, args.seed)
    blk, doc = rows_idx + 1, (rows_idx // 100 + 1).astype(np.int32)
    nb = max(1, min(args.batches, args.steps + args.warmup))
    weak = args.scaling == "weak" and parts > 1

    def draw(mult, count, seed0):
        return draw_rank_queries(sample_queries, where, role_of, me, nq, n, args.seed, mult, count, seed0)

    # weak scaling (the default for N > 1): a step is N x 1000 queries, so every rank answers about 1000 -- the N = 1 step's
    # work per GPU; strong scaling (--scaling strong, and the sibling record of a weak run): the same 1000-query step as N = 1
    batches = draw(parts, nb, 0) if weak else draw(1, nb, 0)
    nq_step = nq * parts if weak else nq
    qvecs = [sift_like_rows_at(qr, dim, args.seed) if len(qr) else np.zeros((0, dim), np.float32) for qr, _ in batches]
    t_gen = time.time() - t0
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    ctx = vsrbac.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    corpus = ctx.load_corpus(x, blk, doc)
    corpus.load_rbac(rbac.user_roles, rbac.permissions)
    MODES = {"prefilter": vsrbac.RANGES, "postfilter": vsrbac.BITMAP}
    d_qs = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) if len(v) else torch.zeros((1, dim), device=dev) for v in qvecs]
    filt = {m: [corpus.pack_filters([corpus.filter_for_user(int(u), MODES[m]) for u in qu]) for _, qu in batches] for m in legs}
    t_load = time.time() - t0 - t_gen
    n_sess = max(1, min(8, int(os.environ.get("VSR_BENCH_SESSIONS", "4"))))
    sessions, mq = [ctx], max(1, max(len(qr) for qr, _ in batches))
    for _ in range(1, n_sess):
        cx = vsrbac.Context(local_rank)
        cx.set_stream(torch.cuda.Stream(device=dev).cuda_stream)
        sessions.append(cx)
    if os.environ.get("VSR_BENCH_NO_U8_HINT") != "1":
        for sess in sessions:
            sess.set_query_hint(True)
    outs = [{"blk": torch.empty((mq, k), dtype=torch.int64, device=dev), "doc": torch.empty((mq, k), dtype=torch.int32, device=dev),
             "row": torch.empty((mq, k), dtype=torch.int64, device=dev), "dist": torch.empty((mq, k), dtype=torch.float32, device=dev),
             "cnt": torch.empty((mq,), dtype=torch.int32, device=dev)} for _ in range(n_sess)]
    state = {"i": 0, "leg": legs[0]}

    def step():
        i = state["i"]
        state["i"] += 1
        b, o = i % nb, outs[i % n_sess]
        nqb = len(batches[b][0])
        if nqb:
            corpus.search_device(ptr(d_qs[b]), nqb, k, "l2", filt[state["leg"]][b], ptr(o["blk"]), ptr(o["doc"]), ptr(o["row"]),
                                 ptr(o["dist"]), ptr(o["cnt"]), None, session=sessions[i % n_sess])

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce(v, op):
        if world == 1:
            return v
        tt = torch.tensor([v], dtype=torch.float64, device=dev if not rehearsal else "cpu")
        dist.all_reduce(tt, op=op)
        return float(tt.item())

    def flagged():
        return int(reduce(float(sum(sess.screening_check(0)[0] for sess in sessions)), dist.ReduceOp.SUM if world > 1 else None))

    def timed_leg(leg, steps, warmup):
        state["leg"], state["i"] = leg, 0
        if os.environ.get("VSR_BENCH_NO_SETUP_PASS") != "1":     # untimed setup pass, as at N = 1 (main(): timed_leg)
            for _ in range(nb * n_sess):
                step()
            barrier()
            state["i"] = 0
            state["setup_pass_steps"] = nb * n_sess
        for _ in range(warmup):
            step()
        barrier()
        for sess in sessions:
            sess.profiling(2)
            sess.stats_reset()
        before = flagged()
        t1 = time.perf_counter()
        for _ in range(steps):
            step()
        t_enq = time.perf_counter() - t1
        barrier()
        mine = time.perf_counter() - t1
        dt = reduce(mine, dist.ReduceOp.MAX if world > 1 else None)
        st = None
        for sess in sessions:
            one = sess.stats()
            sess.profiling(False)
            if st is None:
                st = one
            else:
                for key in ("scan_launches", "scan_ms", "scan_bytes", "scan_rows", "scan_pairs", "unique_rows"):
                    st[key] = [a + b for a, b in zip(st[key], one[key])]
        return {"dt": dt, "dt_this_rank": mine, "t_enq": t_enq, "stats": st, "flagged": flagged() - before,
                "kernel": sessions[0].last_scan_kernel()}

    results = {leg: timed_leg(leg, args.steps, args.warmup) for leg in legs}
    if any(r["flagged"] for r in results.values()):
        raise SystemExit(f"screening flagged queries in the timed region: { {m: r['flagged'] for m, r in results.items()} }")
    head = results[legs[0]]
    sustained = None
    if args.sustained_s > 0:
        s_steps = max(args.steps, int(args.sustained_s / max(head["dt"] / args.steps, 1e-6)))
        s_steps = int(reduce(float(s_steps), dist.ReduceOp.MAX if world > 1 else None))      # the same count on every rank
        srun = timed_leg(legs[0], s_steps, 1)
        sustained = {"steps": s_steps, "seconds": round(srun["dt"], 3), "value": round(nq_step * s_steps / srun["dt"], 1),
                     "ms_per_step": round(srun["dt"] / s_steps * 1e3, 4)}
    # Sibling record, NOT the headline: the other scaling mode on the same ranks.  Under weak scaling (default) that is the
    # N = 1 step's own 1000 queries spread over the ranks (strong scaling: a rank's ~1000 / N-query call is mostly per-call
    # fixed cost, five launches for ~125 queries); under --scaling strong it is the N x 1000-query step.
    sibling = None
    if parts > 1 and os.environ.get("VSR_BENCH_NO_SIBLING") != "1":
        nbs = max(1, min(4, nb))
        other = draw(1, nbs, 500) if weak else draw(parts, nbs, 500)
        saved = (batches, d_qs, filt, outs, nb)
        batches = other
        nb = nbs
        d_qs = [torch.from_numpy(np.ascontiguousarray(sift_like_rows_at(qr, dim, args.seed))).to(dev) if len(qr)
                else torch.zeros((1, dim), device=dev) for qr, _ in other]
        filt = {legs[0]: [corpus.pack_filters([corpus.filter_for_user(int(u), MODES[legs[0]]) for u in qu]) for _, qu in other]}
        mq2 = max(1, max(len(qr) for qr, _ in other))
        outs = [{"blk": torch.empty((mq2, k), dtype=torch.int64, device=dev), "doc": torch.empty((mq2, k), dtype=torch.int32, device=dev),
                 "row": torch.empty((mq2, k), dtype=torch.int64, device=dev), "dist": torch.empty((mq2, k), dtype=torch.float32, device=dev),
                 "cnt": torch.empty((mq2,), dtype=torch.int32, device=dev)} for _ in range(n_sess)]
        try:
            sr = timed_leg(legs[0], args.steps, args.warmup)
            nq_other = nq if weak else nq * parts
            if sr["flagged"] == 0:
                sibling = {"scaling": "strong" if weak else "weak", "queries_per_step_all_ranks": int(nq_other),
                           "queries_per_step_this_rank": int(np.mean([len(q) for q, _ in other])),
                           "ms_per_step": round(sr["dt"] / args.steps * 1e3, 4), "value": round(nq_other * args.steps / sr["dt"], 1),
                           "unit": "queries/s",
                           "note": ("NOT the headline: the N = 1 step's 1000 queries spread over the ranks (total work fixed)" if weak
                                    else "NOT the headline: N x 1000 queries per step (per-GPU work fixed)")}
        finally:
            batches, d_qs, filt, outs, nb = saved
    # parity (default on): this rank's first queries of batch 0 against the oracle over the rows it holds -- by
    # construction ALL rows those users may see; every rank checks its own, rank 0 reports the conjunction
    from oracle.oracle import Oracle
    orc = Oracle("pgflags")
    m = min(8, len(batches[0][0]))
    ok = 1.0
    if m:
        o = outs[0]
        corpus.search_device(ptr(d_qs[0]), len(batches[0][0]), k, "l2", filt[legs[0]][0], ptr(o["blk"]), ptr(o["doc"]), ptr(o["row"]),
                             ptr(o["dist"]), ptr(o["cnt"]), None)
        torch.cuda.synchronize()
        pos = {int(d): i * 100 for i, d in enumerate(my_docs)}                  # document -> first local row
        ranges = [[(pos[int(d)], 100) for d in rbac.visible_docs(int(u)).astype(np.int64)] for u in batches[0][1][:m]]
        rows_o, dist_o, _ = orc.search_ranges("l2", x, qvecs[0][:m], k, ranges, doc, blk)
        ok = float((o["row"][:m].cpu().numpy() == rows_o).all() and (o["dist"][:m].cpu().numpy() == dist_o.astype(np.float32)).all())
    ok_all = reduce(ok, dist.ReduceOp.MIN if world > 1 else None)
    per_rank_q = [len(qr) for qr, _ in batches]
    mean_q = reduce(float(np.mean(per_rank_q)), dist.ReduceOp.SUM if world > 1 else None)
    if rank == 0:
        roof = roofline_of(head["stats"], dim, head["kernel"], n_sess)
        roof["note"] = "rank 0's main scan launches (its roles' rows); " + roof.get("note", "")
        out = {
            "metric": "QPS at recall@100, SIFT10M filtered-kNN (role RBAC), 1/2/4/8 MI355X",
            "value": round(nq_step * args.steps / head["dt"], 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(head["dt"] / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak" if weak else "strong", "vs_baseline": None,
            "dtype": ("int8 planes -> i32 (exact: bit-identical to the fp32 distances of vector.c)" if "int8" in head["kernel"]
                      else "bf16 planes -> f32 (screen), f32 exact re-rank" if "K2" in head["kernel"] else "f32"),
            "data": "synthetic",
            "config": {"workload": f"SIFT10M-like {n}x{dim} fp32 L2 k={k}, tree RBAC 1000 users/100 roles, role-partition "
                                   f"{legs[0]}, exact filtered top-k, {nq_step} queries/step over all ranks "
                                   f"({nb} distinct batches)" + (f": {nq} per GPU, the N = 1 step's work on every GPU" if weak else ""),
                       "rows": n, "dim": dim, "k": k, "queries_per_step": nq_step, "filter": legs[0],
                       "sharding": f"role placement x{parts}: every role (and all it can see) on one GPU, a query touches one "
                                   f"GPU, no exchange", "exchange": "none (no data-path collective)",
                       "recall": 1.0 if ok_all else None, "batches_in_flight": n_sess,
                       "setup_pass": {"steps": state.get("setup_pass_steps", 0),
                                      "note": "untimed, before the warm-up steps: every session searches every distinct batch once"}},
            "roofline": roof,
            "placement": {"roles_on_this_rank": len(my_roles), "rows_on_this_rank": int(len(rows_idx)),
                          "replicated_row_fraction_this_rank": round(len(rows_idx) / (n / parts) - 1.0, 3),
                          "predicted_load_max_over_mean": round(max(load) / (sum(load) / parts), 3),
                          "queries_per_step_this_rank": {"min": int(min(per_rank_q)), "mean": round(float(np.mean(per_rank_q)), 1),
                                                         "max": int(max(per_rank_q))},
                          "queries_per_step_all_ranks": round(mean_q, 1) if world > 1 else None},
            "multi_rank_parity": {"queries_per_rank": m, "ids_and_distances_identical": bool(ok_all),
                                  "checked": "every rank: its first queries of batch 0 against the oracle over the rows it holds"},
            "setup_s": {"generate": round(t_gen, 1), "load": round(t_load, 1)},
            "host_enqueue_ms_per_step": round(head["t_enq"] / args.steps * 1e3, 4),
            "screening_flagged_queries": 0,
        }
        for leg in legs[1:]:
            r = results[leg]
            out[leg] = {"value": round(nq_step * args.steps / r["dt"], 1), "unit": "queries/s", "ms_per_step": round(r["dt"] / args.steps * 1e3, 4),
                        "roofline": roofline_of(r["stats"], dim, r["kernel"], n_sess)}
        if sustained:
            out["sustained"] = sustained
        if sibling:
            out["strong_scaling" if weak else "weak_scaling"] = sibling
        if world == 1:
            out["sim_world"] = {"parts": parts, "simulated_rank": me, "note": "ONE rank's share of the N-GPU job on one GPU (the "
                                "rank with the largest predicted load unless VSR_BENCH_SIM_RANK says otherwise)"}
        print(json.dumps(out), flush=True)
    if not ok_all:
        raise SystemExit("parity check failed on a rank: GPU results differ from the oracle")
    corpus.free()
    for cx in sessions[1:]:
        cx.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)
    import torch
    import torch.distributed as dist
    import vsrbac
    from vsrbac.sharded import shard_bounds
    from vsrbac.datasets import sample_queries, sift_like_corpus, sift_like_rows_at, tree_rbac

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    # rehearsal on a 1-GPU box only: VSR_BENCH_REHEARSAL=1 puts every rank on GPU 0 and exchanges through gloo
    rehearsal = os.environ.get("VSR_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    # development (N = 1 only, with VSR_BENCH_SIM_WORLD): a one-rank RCCL communicator, so that the simulated exchange
    # goes through the real collective call on the exchange stream
    nccl_one = world == 1 and os.environ.get("VSR_BENCH_NCCL_ONE") == "1"
    if nccl_one:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29617")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    sim_world_early = int(os.environ.get("VSR_BENCH_SIM_WORLD", "0")) if world == 1 else 0
    if (world > 1 or sim_world_early > 1) and args.sharding in ("auto", "placement") and not nccl_one:
        return placement_bench(args, torch, dist, vsrbac, rank, local_rank, world, dev, sim_world_early, rehearsal)
    n, dim, k, nq = args.rows, args.dim, args.k, args.queries
    legs = [m for m in args.legs.split(",") if m]
    lo, hi = shard_bounds(n, world, rank, align=100)  # keep documents (100 rows) whole per shard
    t0 = time.time()
    x, blk, doc = sift_like_corpus(hi - lo, dim, seed=args.seed, start=lo)
    rbac = tree_rbac(num_users=1000, num_roles=100, num_docs=n // 100, seed=args.seed,
                     clustered=os.environ.get("VSR_BENCH_CLUSTERED") == "1")      # (development: what contiguous classes would buy)
    nb = max(1, min(args.batches, args.steps + args.warmup))
    batches = []                                      # fresh (rows, users) per step from the seeded stream, cycled
    for b in range(nb):
        qrow, quser = sample_queries(nq, n, 1000, seed=args.seed + 1000 * b)
        batches.append((qrow, quser))
    allrows = np.concatenate([qr for qr, _ in batches])
    allvec = sift_like_rows_at(allrows, dim, args.seed)   # query vectors = corpus rows (read_dataset_function.py:736-737)
    t_gen = time.time() - t0
    hnsw_box = None
    # every batch runs on an explicit stream (the null stream would order itself against all blocking streams)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    ctx = vsrbac.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    corpus = ctx.load_corpus(x, blk, doc, row_offset=lo)
    corpus.load_rbac(rbac.user_roles, rbac.permissions)
    MODES = {"prefilter": vsrbac.RANGES, "postfilter": vsrbac.BITMAP}
    d_qs = [torch.from_numpy(allvec[b * nq:(b + 1) * nq]).to(dev) for b in range(nb)]
    filt = {m: [corpus.pack_filters([corpus.filter_for_user(int(u), MODES[m]) for u in quser]) for _, quser in batches]
            for m in legs}
    t_load = time.time() - t0 - t_gen

    # one packed result record per rank {keys u64, block i64, doc i32, dist f32}[nq][k]: a single all-gather moves it
    rec = ctx.packed_result_bytes(nq, k)
    nk = nq * k
    # N > 1: the exchange + merge of batch i runs on a second stream while batch i+1 is being scanned, so the record and
    # the gathered buffer are double-buffered.  VSR_BENCH_SIM_WORLD=N (development, N=1 only) drives the same
    # choreography on one GPU: the "gather" is a device copy into part 0 of an otherwise empty N-part buffer.
    sim_world = int(os.environ.get("VSR_BENCH_SIM_WORLD", "0")) if world == 1 else 0
    parts = world if world > 1 else max(sim_world, 1)
    overlap = parts > 1 and not rehearsal
    # Two batches in flight: consecutive batches alternate between two sessions (contexts = stream + workspaces) over
    # the one resident corpus, so the selection / re-rank kernels of batch i run under the scan launch of batch i+1.
    # (three on one GPU; a shard's batches are short and mostly fixed cost, where a fourth session takes the simulated
    # 8-way shard step from 0.122 to 0.108 ms)
    n_sess = 1 if rehearsal else max(1, min(8, int(os.environ.get("VSR_BENCH_SESSIONS", "3" if parts == 1 else "4"))))
    # The exchange is grouped: G consecutive batches share ONE record (laid out as one batch of G * nq queries), one
    # all-gather and one merge launch -- fewer, larger collectives, and the host's per-step work (the pacing item once a
    # shard step is down to ~0.15 ms) shrinks to the search call and two event operations.
    G = max(1, int(os.environ.get("VSR_BENCH_EXCHANGE_EVERY", "4"))) if overlap else 1
    n_gbuf = 3 if overlap else max(1, n_sess)         # group records in flight: filling, being exchanged, being merged
                                                      # (no exchange: one record per batch in flight)
    nbuf = n_gbuf * G
    gq = G * nq
    rec_g = ctx.packed_result_bytes(gq, k)
    gk = gq * k

    def views(pack, j):                               # batch j's slices of a group record {keys, block, doc, dist}[G * nq][k]
        lo, hi = j * nk, (j + 1) * nk
        return (pack[0:gk * 8].view(torch.int64)[lo:hi].view(nq, k),     # raw u64 ordering keys
                pack[gk * 8:gk * 16].view(torch.int64)[lo:hi].view(nq, k),
                pack[gk * 16:gk * 20].view(torch.int32)[lo:hi].view(nq, k),
                pack[gk * 20:gk * 24].view(torch.float32)[lo:hi].view(nq, k))

    d_packs = [torch.empty((rec_g,), dtype=torch.uint8, device=dev) for _ in range(n_gbuf)]
    d_views = [views(d_packs[b // G], b % G) for b in range(nbuf)]
    d_rows = [torch.empty((nq, k), dtype=torch.int64, device=dev) for _ in range(nbuf)]
    d_cnts = [torch.empty((nq,), dtype=torch.int32, device=dev) for _ in range(nbuf)]
    s_main = torch.cuda.current_stream()
    sessions, s_scan = [ctx], [s_main]
    for _ in range(1, n_sess):
        sx = torch.cuda.Stream(device=dev)
        cx = vsrbac.Context(local_rank)
        cx.set_stream(sx.cuda_stream)
        sessions.append(cx)
        s_scan.append(sx)
    # SIFT queries are uint8 like the corpus (the query vectors are corpus rows, read_dataset_function.py:736-737): the
    # library is told so and verifies it per query on the device (vsr_set_query_hint; a violation would flag and abort)
    if os.environ.get("VSR_BENCH_NO_U8_HINT") != "1":
        for sess in sessions:
            sess.set_query_hint(True)
    if parts > 1:
        g_packs = [torch.full((parts * rec_g,), 0xFF, dtype=torch.uint8, device=dev) for _ in range(n_gbuf)]   # [parts] group records
        m_blk = torch.empty((gq, k), dtype=torch.int64, device=dev)
        m_doc = torch.empty((gq, k), dtype=torch.int32, device=dev)
        m_dist = torch.empty((gq, k), dtype=torch.float32, device=dev)
        m_cnt = torch.empty((gq,), dtype=torch.int32, device=dev)
        m_keys = torch.empty((gq, k), dtype=torch.int64, device=dev)
    if overlap:
        s_comm = torch.cuda.Stream(device=dev)
        mctx = vsrbac.Context(local_rank)                         # the merge runs on the exchange stream
        mctx.set_stream(s_comm.cuda_stream)
        ev_scan = [torch.cuda.Event() for _ in range(nbuf)]       # slot b of its group record holds the results of its batch
        ev_sent = [torch.cuda.Event() for _ in range(n_gbuf)]     # group record g has been read by the exchange
    state = {"i": 0, "overlap": overlap, "n_sess": n_sess, "leg": legs[0], "open": 0}

    def exchange_group(gb, filled):
        """All-gather + merge of group record gb (its first `filled` batches are new) on the exchange stream."""
        with torch.cuda.stream(s_comm):
            for j in range(filled):
                s_comm.wait_event(ev_scan[gb * G + j])
            if world > 1:
                dist.all_gather_into_tensor(g_packs[gb], d_packs[gb])     # RCCL over xGMI: G * nq * k * 24 bytes per rank
            elif nccl_one:      # development: the simulated exchange through a one-rank RCCL communicator (the real call path)
                dist.all_gather_into_tensor(g_packs[gb][0:rec_g], d_packs[gb])
            else:
                g_packs[gb][0:rec_g].copy_(d_packs[gb], non_blocking=True)
            ev_sent[gb].record(s_comm)
            mctx.merge_topk_packed_device(ptr(g_packs[gb]), parts, gq, k, ptr(m_blk), ptr(m_doc), ptr(m_dist),
                                          ptr(m_keys), ptr(m_cnt))

    def finish_groups():
        """A partly filled group record at the end of a run of steps is exchanged as it is."""
        if state["overlap"] and state["open"]:
            exchange_group(((state["i"] - 1) // G) % n_gbuf, state["open"])
            state["open"] = 0

    def step():
        i = state["i"]
        state["i"] += 1
        overlap, n_sess = state["overlap"], state["n_sess"]
        b = i % nbuf                                             # slot b % G of group record b // G
        if world > 1 and not overlap and not rehearsal:
            b = i % G                                            # serial fall-back: group record 0 only
        qb = i % nb                                              # this step's query batch
        keys_b, blk_b, doc_b, dist_b = d_views[b]
        sess, st = sessions[i % n_sess], s_scan[i % n_sess]
        if overlap and i >= nbuf:
            st.wait_event(ev_sent[b // G])                        # the exchange that read this group record before is done
        corpus.search_device(ptr(d_qs[qb]), nq, k, "l2", filt[state["leg"]][qb], ptr(blk_b), ptr(doc_b), ptr(d_rows[b]),
                             ptr(dist_b), ptr(d_cnts[b]), ptr(keys_b), session=sess)
        if world > 1 and rehearsal:
            torch.cuda.synchronize()
            hg = torch.empty((world * rec_g,), dtype=torch.uint8)
            dist.all_gather_into_tensor(hg, d_packs[0].cpu())
            g_packs[0].copy_(hg)
            ctx.merge_topk_packed_device(ptr(g_packs[0]), world, nq, k, ptr(m_blk), ptr(m_doc), ptr(m_dist), ptr(m_keys),
                                         ptr(m_cnt))
        elif overlap:
            ev_scan[b].record(st)
            state["open"] += 1
            if state["open"] == G:
                exchange_group(b // G, G)
                state["open"] = 0
        elif world > 1:                                           # serial fall-back: scan, exchange, merge one after the other
            torch.cuda.synchronize()                              # (the sessions' streams; no overlap wanted here)
            dist.all_gather_into_tensor(g_packs[0], d_packs[0])
            ctx.merge_topk_packed_device(ptr(g_packs[0]), parts, gq, k, ptr(m_blk), ptr(m_doc), ptr(m_dist),
                                         ptr(m_keys), ptr(m_cnt))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(v):
        if world == 1:
            return v
        tt = torch.tensor([v], dtype=torch.float64, device=dev if not rehearsal else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def flagged():
        total = sum(sess.screening_check(0)[0] for sess in sessions)
        if world > 1:
            ft = torch.tensor([total], dtype=torch.int64, device=dev if not rehearsal else "cpu")
            dist.all_reduce(ft, op=dist.ReduceOp.SUM)
            total = int(ft.item())
        return total

    def timed_leg(leg, steps, warmup):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks."""
        state["leg"] = leg
        state["i"] = 0
        try:
            # setup pass (untimed, before the W warm-up steps; VSR_BENCH_NO_SETUP_PASS=1 skips it): every session searches every
            # batch once.  The driver's W = 5 steps are 2 ms of GPU work behind seconds of host-side setup with the GPU idle;
            # measured, a 20-step region then runs at 0.39-0.41 ms per step against 0.345 after this pass -- the rate of the
            # `sustained` leg and of any serving process.  What the pass warms (VSR_BENCH_COLD_EXPERIMENT): not the
            # batches (48 steps over the warm-up's own five batches do as well), not the streams (200 empty launches each: no
            # effect), not the shader clock (100 ms of bf16 GEMMs: no effect); ~10 ms of this HBM-bound load itself is what it
            # takes (a plain 20 ms copy loop closes half of the gap): the memory side's power management ramps on sustained traffic.
            if os.environ.get("VSR_BENCH_NO_SETUP_PASS") != "1" and not state.get("skip_setup_pass"):
                for _ in range(nb * max(1, state["n_sess"])):
                    step()
                finish_groups()
                barrier()
                state["i"] = 0
                state["setup_pass_steps"] = nb * max(1, state["n_sess"])
            for _ in range(warmup):
                step()
            finish_groups()
            barrier()
        except Exception as exc:                      # never lose the run to the overlapped choreography
            if not (state["overlap"] or state["n_sess"] > 1):
                raise
            print(f"[bench] overlapped loop failed in warm-up ({exc!r}); falling back to one batch in flight, serial "
                  f"exchange", file=sys.stderr, flush=True)
            state["overlap"], state["n_sess"], state["open"] = False, 1, 0
            state["fell_back"] = repr(exc)                    # recorded in the line (config.choreography_fallback), not only on stderr
            torch.cuda.synchronize()
            for _ in range(warmup):
                step()
            barrier()
        for sess in sessions:
            sess.profiling(2)                         # events around the main scan launch only (the roofline kernel)
            sess.stats_reset()
        before = flagged()
        t1 = time.perf_counter()
        for _ in range(steps):
            step()
        finish_groups()
        t_enq = time.perf_counter() - t1              # host time to enqueue the whole run (must stay below dt)
        barrier()
        dt = reduce_max(time.perf_counter() - t1)
        st = None
        for sess in sessions:
            one = sess.stats()
            sess.profiling(False)
            if st is None:
                st = one
            else:
                for key in ("scan_launches", "scan_ms", "scan_bytes", "scan_rows", "scan_pairs", "unique_rows"):
                    st[key] = [a + b for a, b in zip(st[key], one[key])]
                for key in ("host_ms", "host_wait_ms"):
                    st[key] += one[key]
        sim_ok = None
        if sim_world > 1:      # development check of the overlapped choreography on one GPU: merged == local
            lb = (state["i"] - 1) % nbuf                 # the last batch: slot lb % G of the last group record
            last = d_views[lb]
            sl = slice((lb % G) * nq, (lb % G + 1) * nq)
            mk, mb, md = m_keys[sl], m_blk[sl], m_dist[sl]
            sim_ok = bool(state["overlap"] and torch.equal(mk, last[0]) and torch.equal(mb, last[1]) and torch.equal(md, last[3]))
            if not sim_ok:
                bad = ((mk != last[0]) | (mb != last[1]) | (md != last[3])).any(dim=1)
                gp = g_packs[lb // G]
                print(f"[bench] simulated exchange: {int(bad.sum())} of {nq} queries differ after the merge "
                      f"(first {bad.nonzero()[:4].flatten().tolist()}); gathered part 0 == record: "
                      f"{bool(torch.equal(gp[0:rec_g], d_packs[lb // G]))}; overlapped loop ran: {state['overlap']}",
                      file=sys.stderr, flush=True)
        return {"dt": dt, "t_enq": t_enq, "stats": st, "flagged": flagged() - before,
                "kernel": sessions[0].last_scan_kernel(), "sim_ok": sim_ok}

    def alone_stats(leg):
        """The same launch alone on the GPU: a few more batches, one in flight, same events (kernel quality, not value)."""
        if state["n_sess"] <= 1:
            return None
        ctx.profiling(2)
        ctx.stats_reset()
        for j in range(5):
            corpus.search_device(ptr(d_qs[j % nb]), nq, k, "l2", filt[leg][j % nb], ptr(d_views[0][1]), ptr(d_views[0][2]),
                                 ptr(d_rows[0]), ptr(d_views[0][3]), ptr(d_cnts[0]), ptr(d_views[0][0]))
        torch.cuda.synchronize()
        a = ctx.stats()
        ctx.profiling(False)
        return a

    def spot_check(leg, orc, m):
        """Batch 0 of the leg once more, then the first m queries against the oracle (ids AND distances)."""
        corpus.search_device(ptr(d_qs[0]), nq, k, "l2", filt[leg][0], ptr(d_views[0][1]), ptr(d_views[0][2]), ptr(d_rows[0]),
                             ptr(d_views[0][3]), ptr(d_cnts[0]), ptr(d_views[0][0]))
        torch.cuda.synchronize()
        _, fl = ctx.screening_check(nq)
        qrow, quser = batches[0]
        ranges = [[((d - 1) * 100, 100) for d in rbac.visible_docs(int(u)).astype(np.int64)] for u in quser[:m]]
        rows_o, dist_o, _ = orc.search_ranges("l2", x, allvec[:m], k, ranges, doc, blk)
        got_rows = d_rows[0][:m].cpu().numpy()
        got_dist = d_views[0][3][:m].cpu().numpy()
        same = (got_rows == rows_o).all(axis=1) & (got_dist == dist_o.astype(np.float32)).all(axis=1)
        hits = [len(set(a.tolist()) & set(b.tolist())) / max(1, len(b)) for a, b in zip(got_rows, rows_o)]
        return {"queries": int(m), "ids_and_distances_identical": bool(same.all()), "recall_at_k": float(np.mean(hits)),
                "flagged_in_batch": int(np.count_nonzero(fl))}

    # ---- the timed legs ----
    # first the headline leg WITHOUT the setup pass, exactly as the driver's flags alone would run it (W warm-up steps behind an
    # idle GPU): reported beside the headline as `cold_start`, so that the line shows what the setup pass changes
    cold = None
    if world == 1 and os.environ.get("VSR_BENCH_NO_SETUP_PASS") != "1":
        state["skip_setup_pass"] = True
        try:
            exp = os.environ.get("VSR_BENCH_COLD_EXPERIMENT", "")      # development: what the setup pass actually warms
            if exp == "gemm":                                          # ... the GPU's clocks? 100 ms of unrelated work
                ta = torch.randn((8192, 8192), device=dev, dtype=torch.bfloat16)
                t_end = time.perf_counter() + 0.1
                while time.perf_counter() < t_end:
                    (ta @ ta).sum().item()
                del ta
            elif exp == "copy":                                        # ... the memory clocks? 20 ms of plain HBM traffic
                tc = torch.empty((1 << 29,), device=dev, dtype=torch.uint8)
                td = torch.empty_like(tc)
                t_end = time.perf_counter() + 0.02
                while time.perf_counter() < t_end:
                    td.copy_(tc)
                    torch.cuda.synchronize()
                del tc, td
            elif exp == "noop":                                        # ... or the streams' queues? 200 empty launches per stream
                for sx in s_scan:
                    with torch.cuda.stream(sx):
                        tz = torch.empty((64,), device=dev)
                        for _ in range(200):
                            tz.zero_()
                torch.cuda.synchronize()
            elif exp == "same1":                                       # ... 8 calls of ONE batch per session
                for j in range(int(os.environ.get("VSR_BENCH_COLD_CALLS", "8")) * state["n_sess"]):
                    sess = sessions[j % state["n_sess"]]
                    corpus.search_device(ptr(d_qs[0]), nq, k, "l2", filt[legs[0]][0], ptr(d_views[0][1]), ptr(d_views[0][2]),
                                         ptr(d_rows[0]), ptr(d_views[0][3]), ptr(d_cnts[0]), ptr(d_views[0][0]), session=sess)
                torch.cuda.synchronize()
            elif exp == "same":                                        # ... or the sessions? 48 steps over the warm-up's own batches
                for j in range(48):
                    b = j % max(1, min(nb, args.warmup))
                    sess = sessions[j % state["n_sess"]]
                    corpus.search_device(ptr(d_qs[b]), nq, k, "l2", filt[legs[0]][b], ptr(d_views[0][1]), ptr(d_views[0][2]),
                                         ptr(d_rows[0]), ptr(d_views[0][3]), ptr(d_cnts[0]), ptr(d_views[0][0]), session=sess)
                torch.cuda.synchronize()
            rc0 = timed_leg(legs[0], args.steps, args.warmup)
            if not rc0["flagged"]:
                cold = {"value": round(nq * args.steps / rc0["dt"], 1), "unit": "queries/s",
                        "ms_per_step": round(rc0["dt"] / args.steps * 1e3, 4),
                        "note": "the headline leg run first and WITHOUT the untimed setup pass (config.setup_pass): W warm-up steps "
                                "behind seconds of host-side setup, then the K timed steps"}
        finally:
            state["skip_setup_pass"] = False
    results = {}
    for leg in legs:
        r = timed_leg(leg, args.steps, args.warmup)
        r["alone"] = alone_stats(leg)
        results[leg] = r
    # (the CPU port's HNSW build runs on a host thread from here on, under the remaining GPU legs -- not under the headline legs)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and int(os.environ.get("VSR_BENCH_SIM_WORLD", "0")) <= 1:
        from oracle.oracle import Oracle as _OracleEarly
        vis0 = rbac.visible_docs(int(batches[0][1][0])).astype(np.int64)
        part_rows = (((vis0 - 1) * 100)[:, None] + np.arange(100)[None, :]).reshape(-1)
        part_rows = part_rows[part_rows < len(x)]
        if len(part_rows) < args.index_rows:                       # a small role: top up with the rows that follow
            extra = np.setdiff1d(np.arange(min(len(x), args.index_rows * 2)), part_rows)[:args.index_rows - len(part_rows)]
            part_rows = np.sort(np.concatenate([part_rows, extra]))
        hnsw_box = start_cpu_hnsw_build(_OracleEarly("pgflags"), x, part_rows[:args.index_rows], args.seed)

    head = results[legs[0]]
    ablation = os.environ.get("VSR_BENCH_ABLATION") == "1"   # development: timing of deliberately wrong variant kernels
    if not ablation and (head["flagged"] != 0 or any(r["flagged"] for r in results.values())):
        # a flagged query is one whose exactness the screening could not prove: the serving loop must re-run it on the
        # exact path (vsr_search / GpuShardEngine do).  A bench line that claims recall 1.0 must not contain any.
        raise SystemExit(f"screening flagged queries in the timed region: { {m: r['flagged'] for m, r in results.items()} }")

    # ---- sustained leg (N = 1 and N > 1 alike): >= --sustained-s seconds of headline steps, clocks and thermals settled ----
    sustained = None
    if args.sustained_s > 0:
        per = head["dt"] / args.steps
        s_steps = max(args.steps, int(args.sustained_s / max(per, 1e-6)))
        s = timed_leg(legs[0], s_steps, 1)
        if s["flagged"] and not ablation:
            raise SystemExit(f"screening flagged {s['flagged']} queries in the sustained leg")
        sustained = {"steps": s_steps, "seconds": round(s["dt"], 3), "value": round(nq * s_steps / s["dt"], 1),
                     "ms_per_step": round(s["dt"] / s_steps * 1e3, 4),
                     "host_enqueue_ms_per_step": round(s["t_enq"] / s_steps * 1e3, 4)}

    def leg_record(leg, r):
        dt = r["dt"]
        roof = roofline_of(r["stats"], dim, r["kernel"], state["n_sess"], r["alone"])
        tag = f"{n}x{dim} k={k} q={nq} {leg} gpus={world}"
        ent = None
        try:      # HBM bytes per launch from a PMC pass of this same command (never measured inside the timed run)
            with open(args.traffic) as f:
                tr = json.load(f)
            ent = tr.get(tag)
            if ent:
                short = r["kernel"].split("<")[0].replace("vsr::", "")
                for name, v in ent["kernels"].items():
                    targs = name.split("<", 1)[1].split(">", 1)[0].replace(" ", "").split(",") if "<" in name else []
                    is_sample = short == "mfma_wide_kernel" and len(targs) > 2 and targs[2] == "true"
                    if short in name and not is_sample:                  # the main launch, not the SAMPLE=true pass
                        roof["traffic"] = int(v["hbm_bytes_per_launch"])
                        roof["traffic_source"] = os.path.relpath(args.traffic, ROOT) + ": " + tr.get("method", "")
                        roof["hbm_rate_measured"] = round(roof["traffic"] / (roof["launch_ms"] * 1e-3) / 1e9, 1)
        except (OSError, ValueError, KeyError, IndexError):
            pass
        # the same bytes against the WALL time of a step (every kernel of the step, all batches in flight): what the job
        # as a whole draws from HBM per second, independent of how many batches share the GPU
        roof["per_step"] = {"achieved": round(roof["unique_bytes"] / (dt / args.steps) / 1e9, 1), "unit": "GB/s",
                            "frac": round(roof["unique_bytes"] / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4)}
        try:      # ... and what the step's kernels MOVED over HBM per step (the PMC passes' bytes of every kernel of a batch)
            step_kernels = ("mfma_wide_kernel", "i8_stream_kernel", "seed_select_kernel", "select_rerank_kernel", "stage_kernel")
            moved = sum(v["hbm_bytes_per_launch"] for name, v in ent["kernels"].items() if any(sk in name for sk in step_kernels)
                        and v["dispatches"] >= 3) if ent else 0
            if moved:
                roof["per_step"]["hbm_traffic_bytes"] = int(moved)
                roof["per_step"]["hbm_traffic_gbs"] = round(moved / (dt / args.steps) / 1e9, 1)
                roof["per_step"]["hbm_traffic_frac"] = round(moved / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4)
                roof["per_step"]["note"] = ("achieved / frac: the algorithmic unique bytes against the wall time of a step; hbm_traffic_*: "
                                            "the PMC-measured HBM bytes of every kernel of one batch (stage, sample, seed, main, select) "
                                            "against the same wall time")
        except (NameError, KeyError, TypeError):
            pass
        roof["workload_tag"] = tag
        wait_ms = r["stats"]["host_wait_ms"] / args.steps
        return {"value": round(nq * args.steps / dt, 1), "unit": "queries/s", "ms_per_step": round(dt / args.steps * 1e3, 4),
                "host_enqueue_ms_per_step": round(r["t_enq"] / args.steps * 1e3, 4),
                # ... of which the host only WAITED for the GPU (staging block of an earlier batch still in use): the rest is
                # the work the host does per step (bench loop + planner + launches)
                "host_busy_ms_per_step": round(r["t_enq"] / args.steps * 1e3 - wait_ms, 4),
                "host_library_ms_per_step": round(r["stats"]["host_ms"] / args.steps - wait_ms, 4), "roofline": roof,
                "screening_flagged_queries": int(r["flagged"])}

    recs = {leg: leg_record(leg, r) for leg, r in results.items()}
    main_rec = recs[legs[0]]
    out = {
        "metric": "QPS at recall@100, SIFT10M filtered-kNN (role RBAC), 1/2/4/8 MI355X",
        "value": main_rec["value"], "unit": "queries/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": main_rec["ms_per_step"], "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None,
        "dtype": ("int8 planes -> i32 (exact: bit-identical to the fp32 distances of vector.c)" if "int8" in results[legs[0]]["kernel"]
                  else "bf16 planes -> f32 (screen), f32 exact re-rank" if "K2" in results[legs[0]]["kernel"] else "f32"),
        "data": "synthetic",
        "config": {"workload": f"SIFT10M-like {n}x{dim} fp32 L2 k={k}, tree RBAC 1000 users/100 roles, "
                               f"role-partition {legs[0]}, exact filtered top-k, {nq} queries/step ({nb} distinct batches)",
                   "rows": n, "dim": dim, "k": k, "queries_per_step": nq, "filter": legs[0],
                   "sharding": f"row-range x{world}", "recall": None, "batches_in_flight": state["n_sess"],
                   "setup_pass": {"steps": state.get("setup_pass_steps", 0),
                                  "note": "untimed, before the warm-up steps: every session searches every distinct batch once "
                                          "(~17 ms of the step's own HBM-bound load: the memory side's clocks ramp on sustained traffic; "
                                          "`cold_start` is the same leg without it); VSR_BENCH_NO_SETUP_PASS=1 skips it"}},
        "roofline": main_rec["roofline"],
        "setup_s": {"generate": round(t_gen, 1), "load": round(t_load, 1)},
        # what this rank keeps in HBM for the corpus: the fp32 rows (exact re-rank, K1), the hi-only bf16 screening planes and,
        # for u8-valued rows, the int8 planes the headline kernel reads; side arrays = ids, norms, document index
        "resident_bytes": {"fp32_rows": int(hi - lo) * dim * 4, "bf16_hi_planes": int(hi - lo) * dim * 2,
                           "int8_planes": int(hi - lo) * (dim + 4), "side_arrays": int(hi - lo) * 32},
        "host_enqueue_ms_per_step": main_rec["host_enqueue_ms_per_step"],
        "host_busy_ms_per_step": main_rec["host_busy_ms_per_step"],
        "host_library_ms_per_step": main_rec["host_library_ms_per_step"],
        "screening_flagged_queries": main_rec["screening_flagged_queries"],
    }
    if cold:
        out["cold_start"] = cold
    for leg in legs[1:]:
        out[leg] = recs[leg]
        out[leg]["workload"] = (f"same corpus, users and queries; filter = per-user row-level-security bitmap tested in "
                                f"the distance loop (BASELINE config 4's mode)" if leg == "postfilter" else leg)
    if sustained:
        out["sustained"] = sustained

    # ---- sibling record (N = 1): the headline leg's queries in batches four times as large -- what a caller that can wait for
    # 4000 queries gets (and the shape a rank of a weak-scaling job sees when its queue is deep): more queries share every pass
    if world == 1 and sim_world <= 1 and nb >= 4 and os.environ.get("VSR_BENCH_NO_LARGE_BATCH") != "1":
        try:
            mult = 4
            nbl = nb // mult
            lq = [torch.from_numpy(np.ascontiguousarray(allvec[g * mult * nq:(g + 1) * mult * nq])).to(dev) for g in range(nbl)]
            lf = [corpus.pack_filters([f for b in range(g * mult, (g + 1) * mult) for f in filt[legs[0]][b]._keep]) for g in range(nbl)]
            nql = mult * nq
            lo_ = [{"blk": torch.empty((nql, k), dtype=torch.int64, device=dev), "doc": torch.empty((nql, k), dtype=torch.int32, device=dev),
                    "row": torch.empty((nql, k), dtype=torch.int64, device=dev), "dist": torch.empty((nql, k), dtype=torch.float32, device=dev),
                    "cnt": torch.empty((nql,), dtype=torch.int32, device=dev)} for _ in range(state["n_sess"])]

            def lstep(i):
                o, g = lo_[i % state["n_sess"]], i % nbl
                corpus.search_device(ptr(lq[g]), nql, k, "l2", lf[g], ptr(o["blk"]), ptr(o["doc"]), ptr(o["row"]), ptr(o["dist"]),
                                     ptr(o["cnt"]), None, session=sessions[i % state["n_sess"]])
            for i in range(2 * state["n_sess"]):
                lstep(i)
            torch.cuda.synchronize()
            f0 = flagged()
            lsteps = max(6, args.steps // 2)
            tl0 = time.perf_counter()
            for i in range(lsteps):
                lstep(i)
            torch.cuda.synchronize()
            dtl = time.perf_counter() - tl0
            # parity: the first 1000 queries of large batch 0 are the headline leg's batch 0 -- same rows, same distances
            corpus.search_device(ptr(d_qs[0]), nq, k, "l2", filt[legs[0]][0], ptr(d_views[0][1]), ptr(d_views[0][2]), ptr(d_rows[0]),
                                 ptr(d_views[0][3]), ptr(d_cnts[0]), ptr(d_views[0][0]))
            lstep(0)
            torch.cuda.synchronize()
            same = bool((lo_[0]["row"][:nq] == d_rows[0]).all().item() and (lo_[0]["dist"][:nq] == d_views[0][3]).all().item())
            out["large_batch"] = {"queries_per_step": int(nql), "steps": int(lsteps), "ms_per_step": round(dtl / lsteps * 1e3, 4),
                                  "same_results_as_1000_query_calls": same,
                                  "value": round(nql * lsteps / dtl, 1), "unit": "queries/s", "flagged": int(flagged() - f0),
                                  "note": "NOT the headline: the same queries and filters in calls of 4000 (four headline batches "
                                          "concatenated), three calls in flight"}
        except Exception as exc:
            out.setdefault("leg_errors", {})["large_batch"] = repr(exc)

    # ---- sibling line (N = 1): the same headline leg with the int8 planes switched off (VSR_NO_INT8): hi-only bf16 planes,
    # 260 bytes per row instead of 132, fp32 accumulation, same exact results ----
    if world == 1 and sim_world <= 1 and not args.no_bf16_line:
        try:
            os.environ["VSR_NO_INT8"] = "1"
            try:
                c_bf = ctx.load_corpus(x, blk, doc, row_offset=lo)
            finally:
                del os.environ["VSR_NO_INT8"]
            c_bf.load_rbac(rbac.user_roles, rbac.permissions)
            filt_bf = {legs[0]: [c_bf.pack_filters([c_bf.filter_for_user(int(u), MODES[legs[0]]) for u in quser]) for _, quser in batches]}
            saved = (corpus, filt)
            corpus, filt = c_bf, filt_bf                      # (timed_leg / alone_stats / leg_record read these names)
            try:
                rb = timed_leg(legs[0], args.steps, args.warmup)
                rb["alone"] = alone_stats(legs[0])
                rec_bf = leg_record(legs[0], rb)
                rec_bf["roofline"]["traffic"] = None          # (the PMC file holds the int8 launch's bytes)
                rec_bf["dtype"] = "bf16 planes -> f32 accumulate (screen; exact for these integer rows), f32 exact re-rank"
                out["bf16_planes"] = rec_bf
            finally:
                corpus, filt = saved
                c_bf.free()
        except Exception as exc:          # a sibling leg never costs the run its headline line
            out.setdefault("leg_errors", {})["bf16_planes"] = repr(exc)
            print(f"[bench] leg bf16_planes failed: {exc!r}", file=sys.stderr, flush=True)

    d_keys, d_blk, d_doc, d_dist = d_views[0]         # slot 0 / session 0 from here on (everything above has drained)
    d_row, d_cnt = d_rows[0], d_cnts[0]
    # ---- latency mode (informational, N = 1): the harness's call shape, one query per call ----
    if world == 1 and sim_world <= 1:
        try:
            m1 = min(200, nq)
            singles = [corpus.pack_filters([filt[legs[0]][0]._keep[i]]) for i in range(m1)]
            for i in range(8):
                corpus.search_device(ptr(d_qs[0][i:i + 1]), 1, k, "l2", singles[i], ptr(d_blk), ptr(d_doc), ptr(d_row),
                                     ptr(d_dist), ptr(d_cnt), ptr(d_keys))
            torch.cuda.synchronize()
            tl = time.perf_counter()
            for i in range(m1):
                corpus.search_device(ptr(d_qs[0][i:i + 1]), 1, k, "l2", singles[i], ptr(d_blk), ptr(d_doc), ptr(d_row),
                                     ptr(d_dist), ptr(d_cnt), ptr(d_keys))
            torch.cuda.synchronize()
            tl = time.perf_counter() - tl
            out["single_query_mode"] = {"queries": m1, "ms_per_query": round(tl / m1 * 1e3, 4), "qps": round(m1 / tl, 1),
                                        "kernel": ctx.last_scan_kernel(),
                                        "note": "one query per call, back to back on one stream; not the headline"}

            # the brute-force distance kernel on its own (north_star: ">= 60 % HBM roofline on the brute-force distance
            # kernel"): K1, one unfiltered query over the whole corpus, HIP events on the launch stream.  Two forms, each priced
            # at the bytes ITS kernel reads: the fp32 rows (rows * d * 4 + k * 12; what any corpus gets) and, for
            # this u8-valued corpus under the query hint, the int8 planes (rows * (128 + 4)).
            hinted = os.environ.get("VSR_BENCH_NO_U8_HINT") != "1"
            for name, hint in (("brute_force_scan", False), ("brute_force_scan_int8_planes", True)):
                if hint and not hinted:
                    continue
                ctx.set_query_hint(hint)
                for i in range(3):
                    corpus.search_device(ptr(d_qs[0][i:i + 1]), 1, k, "l2", None, ptr(d_blk), ptr(d_doc), ptr(d_row), ptr(d_dist),
                                         ptr(d_cnt), ptr(d_keys))
                torch.cuda.synchronize()
                ctx.profiling(2)
                ctx.stats_reset()
                for i in range(10):
                    corpus.search_device(ptr(d_qs[0][i:i + 1]), 1, k, "l2", None, ptr(d_blk), ptr(d_doc), ptr(d_row), ptr(d_dist),
                                         ptr(d_cnt), ptr(d_keys))
                st1 = ctx.stats()
                ctx.profiling(False)
                if st1["scan_launches"][0]:
                    kern = ctx.last_scan_kernel()
                    on8 = "int8" in kern
                    bf_ms = st1["scan_ms"][0] / st1["scan_launches"][0]
                    bf_bytes = (hi - lo) * ((128 + 4) if on8 else dim * 4) + k * 12
                    out[name] = {"kernel": kern, "rows": int(hi - lo), "bytes": int(bf_bytes),
                                 "launch_ms": round(bf_ms, 4), "achieved_gbs": round(bf_bytes / bf_ms / 1e6, 1),
                                 "frac_of_8TBs": round(bf_bytes / bf_ms / 1e6 / HBM_PEAK_GBS, 4),
                                 "note": ("one query, no filter; bytes = rows * (128 B int8 plane + 4 B |row|^2) + k * 12: what this "
                                          "kernel reads" if on8 else
                                          "one query, no filter, fp32 rows (SURVEY 8d: rows * d * 4 + k * 12 bytes)") +
                                         "; the launch includes the in-kernel merge of the workgroups' lists"}
            ctx.set_query_hint(hinted)

            # the boundary's host-buffer form (vsr_search: queries in, results out over PCIe, synchronous, flagged queries
            # re-run inside the call), whole 1000-query batches; and the synchronous harness call, one query at a time
            hq_np = allvec[0:nq]
            hfil = list(filt[legs[0]][0]._keep)
            corpus.search(hq_np, k, "l2", hfil)
            th = time.perf_counter()
            for _ in range(5):
                corpus.search(hq_np, k, "l2", hfil)
            th = (time.perf_counter() - th) / 5
            t1q = time.perf_counter()
            for i in range(50):
                corpus.search(hq_np[i:i + 1], k, "l2", [hfil[i]])
            t1q = (time.perf_counter() - t1q) / 50
            out["host_buffers"] = {"ms_per_step": round(th * 1e3, 4), "qps": round(nq / th, 1),
                                   "one_query_per_call_ms": round(t1q * 1e3, 4),
                                   "note": "vsr_search with host pointers (PCIe inclusive, synchronous); never `value`"}
        except Exception as exc:          # a sibling leg never costs the run its headline line
            out.setdefault("leg_errors", {})["single_query_brute_force_host_buffers"] = repr(exc)
            print(f"[bench] leg single_query_brute_force_host_buffers failed: {exc!r}", file=sys.stderr, flush=True)

    # ---- 768-d legs (N = 1): BASELINE configs 3 / 5's row length, the path north_star names for MFMA ----
    # wiki768_unfiltered: `--wiki-rows` x 768 unit rows (synthetic, Gaussian), 1000 queries per batch, cosine, no filter: one
    #   filter part seen by every query = the batched-query x corpus GEMM (K2g: 256 x 256 tiles on the coarse bf16 planes)
    # wiki768_rbac: the first 1M of those rows, tree RBAC (1000 users / 100 roles), every query under its user's role
    #   pre-filter: ~100 permission classes, 10..330 queries each (K2w long rows, hi + mid planes)
    # Each leg: event-timed main launch (roofline, bound "mfma": products issued / dense bf16 peak), wall time per batch,
    # flagged queries (and the time with the tiered re-run when there are any), oracle spot check within 1e-4.
    if world == 1 and sim_world <= 1 and args.wiki_rows > 0:
        try:
            wd, wk = 768, 100
            wn = int(args.wiki_rows)
            tw = time.time()
            xw = np.empty((wn, wd), dtype=np.float32)
            gen = torch.Generator(device=dev)
            gen.manual_seed(args.seed + 768)
            for a in range(0, wn, 500_000):
                t = torch.randn((min(500_000, wn - a), wd), generator=gen, device=dev, dtype=torch.float32)
                t /= t.norm(dim=1, keepdim=True)
                xw[a:a + t.shape[0]] = t.cpu().numpy()
                del t
            rw_ = np.arange(wn, dtype=np.int64)
            blkw, docw = rw_ + 1, (rw_ // 10 + 1).astype(np.int32)
            rngw = np.random.default_rng(args.seed + 769)
            qw = xw[rngw.integers(0, wn, nq)] + 0.05 * rngw.standard_normal((nq, wd)).astype(np.float32)
            d_qw = torch.from_numpy(qw).to(dev)
            t_wgen = time.time() - tw
            o_blk = torch.empty((nq, wk), dtype=torch.int64, device=dev)
            o_doc = torch.empty((nq, wk), dtype=torch.int32, device=dev)
            o_row = torch.empty((nq, wk), dtype=torch.int64, device=dev)
            o_dist = torch.empty((nq, wk), dtype=torch.float32, device=dev)
            o_cnt = torch.empty((nq,), dtype=torch.int32, device=dev)

            def wiki_leg(cw, fl, label, workload, mask_of):
                call = lambda: cw.search_device(ptr(d_qw), nq, wk, "cosine", fl, ptr(o_blk), ptr(o_doc), ptr(o_row), ptr(o_dist),
                                                ptr(o_cnt), None, dim=wd)
                for _ in range(2):
                    call()
                torch.cuda.synchronize()
                total0 = ctx.screening_check(0)[0]
                ctx.profiling(2)
                ctx.stats_reset()
                t1 = time.perf_counter()
                for _ in range(args.wiki_steps):
                    call()
                torch.cuda.synchronize()
                dtw = (time.perf_counter() - t1) / args.wiki_steps
                stw = ctx.stats()
                ctx.profiling(False)
                kern = ctx.last_scan_kernel()
                flagged_n = int((o_cnt < 0).sum().item())
                assert ctx.screening_check(0)[0] - total0 == flagged_n * args.wiki_steps
                roof = roofline_of(stw, wd, kern, 1)
                rec = {"workload": workload, "value": round(nq / dtw, 1), "unit": "queries/s", "ms_per_step": round(dtw * 1e3, 4),
                       "steps": args.wiki_steps, "dtype": "bf16 planes -> f32 accumulate (screen), f32 exact re-rank",
                       "roofline": roof, "screening_flagged_queries": flagged_n}
                if flagged_n:      # the serving form: search, wait, re-run what was flagged one tier down, patch
                    cw.search_device_exact(ptr(d_qw), nq, wk, "cosine", fl, ptr(o_blk), ptr(o_doc), ptr(o_row), ptr(o_dist), ptr(o_cnt),
                                           None, dim=wd)
                    t2 = time.perf_counter()
                    for _ in range(args.wiki_steps):
                        cw.search_device_exact(ptr(d_qw), nq, wk, "cosine", fl, ptr(o_blk), ptr(o_doc), ptr(o_row), ptr(o_dist),
                                               ptr(o_cnt), None, dim=wd)
                    dte = (time.perf_counter() - t2) / args.wiki_steps
                    rec["with_tiered_rerun"] = {"ms_per_step": round(dte * 1e3, 4), "value": round(nq / dte, 1)}
                else:
                    cw.search_device_exact(ptr(d_qw), nq, wk, "cosine", fl, ptr(o_blk), ptr(o_doc), ptr(o_row), ptr(o_dist), ptr(o_cnt),
                                           None, dim=wd)
                if not args.no_cpu_baseline:
                    from oracle.oracle import Oracle
                    orc_w = Oracle("pgflags")
                    rows_g, dist_g = o_row.cpu().numpy(), o_dist.cpu().numpy()
                    worst, overlap = 0.0, []
                    tcs = time.perf_counter()
                    for i in range(3):
                        oidx, odist = orc_w.filtered_topk("cosine", cw._rows_host, qw[i], wk, cw._doc_host, cw._blk_host, mask_of(i))
                        worst = max(worst, float(np.abs(dist_g[i, :len(odist)] - odist).max()))
                        overlap.append(len(set(rows_g[i].tolist()) & set(oidx.tolist())))
                    rec["parity_spot_check"] = {"queries": 3, "max_abs_distance_error": worst, "tolerance": 1e-4,
                                                "ids_in_common_of_k": overlap, "within_tolerance": bool(worst <= 1e-4 and min(overlap) >= wk - 1),
                                                "oracle_qps_one_core": round(3 / (time.perf_counter() - tcs), 3)}
                return rec

            def host_view(cw, rows, doc_, blk_):                      # what the oracle checks against (host arrays, no copy)
                cw._rows_host, cw._doc_host, cw._blk_host = rows, doc_, blk_
                return cw

            tl0 = time.time()
            cw5 = host_view(ctx.load_corpus(xw, blkw, docw), xw, docw, blkw)
            t_wload = time.time() - tl0
            out["wiki768_unfiltered"] = wiki_leg(cw5, None, "unfiltered", f"{wn}x{wd} unit rows, cosine, k={wk}, {nq} unfiltered queries "
                                                 f"per batch (BASELINE config 5's shape on one GPU)", lambda i: None)
            out["wiki768_unfiltered"]["setup_s"] = {"generate": round(t_wgen, 1), "load": round(t_wload, 1)}
            out["wiki768_unfiltered"]["resident_bytes"] = {"fp32_rows": wn * wd * 4, "bf16_hi_mid_planes": wn * wd * 4, "bf16_coarse_planes": wn * wd * 2}
            cw5.free()
            n1 = min(wn, 1_000_000)
            x1, blk1, doc1 = xw[:n1], blkw[:n1], docw[:n1]
            cw1 = host_view(ctx.load_corpus(x1, blk1, doc1), x1, doc1, blk1)
            rb1 = tree_rbac(num_users=1000, num_roles=100, num_docs=int(doc1.max()), seed=args.seed + 3)
            cw1.load_rbac(rb1.user_roles, rb1.permissions)
            users1 = rngw.integers(1, 1001, nq)
            fl1 = cw1.pack_filters([cw1.filter_for_user(int(u), vsrbac.RANGES) for u in users1])
            from oracle.oracle import Oracle as _Orc
            _o = _Orc("pgflags")
            out["wiki768_rbac"] = wiki_leg(cw1, fl1, "rbac", f"{n1}x{wd} unit rows, cosine, k={wk}, tree RBAC 1000 users / 100 roles, "
                                           f"{nq} queries per batch under their users' role pre-filters (BASELINE config 3's shape)",
                                           lambda i: _o.user_row_mask(int(users1[i]), rb1.user_roles, rb1.permissions, doc1))
            cw1.free()
            del xw
        except Exception as exc:          # a sibling leg never costs the run its headline line
            out.setdefault("leg_errors", {})["wiki768"] = repr(exc)
            print(f"[bench] leg wiki768 failed: {exc!r}", file=sys.stderr, flush=True)

    # ---- CPU baseline (rank 0, N = 1): the oracle, pgvector's flags, one thread, bounded sample ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.oracle import Oracle
        orc = Oracle("pgflags")
        m = min(args.cpu_queries, nq)
        qrow, quser = batches[0]
        qvec = allvec[:nq]
        ranges = [[((d - 1) * 100, 100) for d in rbac.visible_docs(int(u)).astype(np.int64)] for u in quser[:m]]
        tc = time.perf_counter()
        for _ in range(max(1, args.cpu_reps)):
            orc.search_ranges("l2", x, qvec[:m], k, ranges, doc, blk)
        cpu_s = (time.perf_counter() - tc) / max(1, args.cpu_reps)
        # the same sample fanned over host threads, one query per thread at a time (the C call releases the GIL):
        # an all-cores figure beside the single-core one; never the headline
        import concurrent.futures
        threads = host_threads()
        cuts = np.linspace(0, m, threads + 1).astype(int)
        ta = time.perf_counter()
        with concurrent.futures.ThreadPoolExecutor(threads) as pool:
            list(pool.map(lambda se: orc.search_ranges("l2", x, qvec[se[0]:se[1]], k, ranges[se[0]:se[1]], doc, blk),
                          [(int(a), int(b)) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]))
        cpu_all_s = time.perf_counter() - ta
        out["cpu_baseline"] = {
            "value": round(m / cpu_s, 2), "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": f"first {m} queries of batch 0 x {max(1, args.cpu_reps)} repetitions, exact seq-scan of each "
                      f"user's role partition, {cpu_s * max(1, args.cpu_reps):.1f} s on 1 of {os.cpu_count()} host "
                      f"cores (oracle built with pgvector's flags)",
        }
        out["cpu_baseline"]["all_cores"] = {"value": round(m / cpu_all_s, 1), "unit": "queries/s", "threads": threads,
                                            "note": "same sample, queries fanned over host threads"}
        # CPU baseline no. 2 and the index legs: pgvector's HNSW on a whole role partition (CPU port; the same graph on K4),
        # and IVFFlat built and probed on the GPU beside the CPU port's scan of the same lists
        if hnsw_box is not None:
            try:
                out["cpu_baseline"]["hnsw"], out["hnsw"] = hnsw_legs(args, torch, ctx, orc, hnsw_box, qvec, k, dev)
            except Exception as exc:      # the baseline is reported, never required
                out["cpu_baseline"]["hnsw"] = {"error": repr(exc)}
        if args.ivf_rows > 0:
            try:
                out["ivf"] = ivf_leg(args, torch, vsrbac, ctx, orc, x, blk, doc, qvec, k, dev)
            except Exception as exc:
                out["ivf"] = {"error": repr(exc)}
            try:      # the reference's own ivfflat parameters on a corpus the size of a large role partition: lists = 100 (pgvector's
                      # default: initialize_partitions.py:404-409 creates the index without a WITH clause), nprobe = 5 (config_params.json:2)
                out["ivf_reference_params"] = ivf_leg(args, torch, vsrbac, ctx, orc, x, blk, doc, qvec, k, dev,
                                                      rows=min(300_000, args.ivf_rows), lists=100, probes_list=(5,))
            except Exception as exc:
                out["ivf_reference_params"] = {"error": repr(exc)}
        checks = {leg: spot_check(leg, orc, m) for leg in legs}
        out["parity_spot_check"] = checks[legs[0]]
        for leg in legs[1:]:
            out[leg]["parity_spot_check"] = checks[leg]
        out["config"]["recall"] = checks[legs[0]]["recall_at_k"]      # measured on the sample, not asserted
        if not all(c["ids_and_distances_identical"] for c in checks.values()):
            print(json.dumps(out), flush=True)
            raise SystemExit("parity spot check failed: GPU results differ from the oracle")
    out["config"]["choreography_fallback"] = state.get("fell_back")     # None: the batches-in-flight loop ran as planned
    if parts > 1:
        out["config"]["exchange"] = (f"one all-gather + merge per {G} batches on a second stream, overlapped with the "
                                     f"scans of the next batches" if state["overlap"] else
                                     "serial (rehearsal through host memory)" if rehearsal else "serial, one per batch")
    if sim_world > 1:
        out["sim_world"] = {"parts": parts, "merged_equals_local": bool(all(r["sim_ok"] for r in results.values()))}
    if rank == 0 and world > 1 and os.environ.get("VSR_BENCH_VERIFY", "1") != "0":
        # default on: the merged multi-rank result of a few queries against the oracle on the full corpus (rank 0 regenerates it)
        from oracle.oracle import Oracle
        orc = Oracle("pgflags")
        xf, blkf, docf = sift_like_corpus(n, dim, seed=args.seed)
        m = 8
        qb = (state["i"] - 1) % nb
        qrow, quser = batches[qb]
        ranges = [[((d - 1) * 100, 100) for d in rbac.visible_docs(int(u)).astype(np.int64)] for u in quser[:m]]
        rows_o, dist_o, _ = orc.search_ranges("l2", xf, allvec[qb * nq:qb * nq + m], k, ranges, docf, blkf)
        off = (((state["i"] - 1) % nbuf) % G) * nq      # the last batch's slot of its group record
        ok = bool((m_blk[off:off + m].cpu().numpy() == blkf[rows_o]).all() and
                  (m_dist[off:off + m].cpu().numpy() == dist_o.astype(np.float32)).all())
        out["multi_rank_parity"] = {"queries": m, "ids_and_distances_identical": ok}
    if rank == 0:
        print(json.dumps(out), flush=True)
    corpus.free()
    if overlap:
        mctx.close()
    for cx in sessions[1:]:
        cx.close()
    ctx.close()
    if world > 1 or nccl_one:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
