#!/usr/bin/env python3
"""bench.py — QPS of RBAC-filtered exact k-NN on MI355X (BASELINE.json's metric), with the K1 roofline
and a CPU baseline timed beside it.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config.workload): synthetic SIFT10M-like corpus (10M x 128 fp32, integer-valued, 100 rows per
document), tree RBAC (1000 users / 100 roles), k = 100, L2, role-level filter applied as a
pre-filter (only the rows of the user's role partition are scanned).  A step = one batch of `--queries`
queries (uniform rows x uniform users) through the whole hot path: K1 scan (distance + permission + running
top-k) and K5 select; with N > 1 the corpus is sharded by contiguous row range (strong scaling, total work
fixed), every rank searches its shard and the per-rank top-k lists are all-gathered over RCCL and merged.
Inputs (corpus, filters, queries) are resident in HBM when the timed region starts.

No part of the timed path touches the CPU oracle; it is used only by the cpu_baseline leg (rank 0, N = 1)
and for a parity spot-check of the GPU results on the same sample.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "vectorsearch-rbac_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling ~6290


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--queries", type=int, default=1000, help="queries per step")
    ap.add_argument("--mode", choices=["prefilter", "postfilter"], default="prefilter")
    ap.add_argument("--cpu-queries", type=int, default=1000, help="sample size of the CPU baseline leg")
    ap.add_argument("--cpu-reps", type=int, default=2, help="repetitions of the CPU sample (10-30 s of CPU work)")
    ap.add_argument("--traffic", default=os.path.join(ROOT, "profiles", "r1", "traffic.json"),
                    help="PMC-derived HBM bytes per launch (tools/pmc_traffic.py) for roofline.traffic")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=20251121)
    return ap.parse_args()


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import vsrbac
    from vsrbac.sharded import shard_bounds
    from vsrbac.datasets import sample_queries, sift_like_corpus, sift_like_rows_at, tree_rbac

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
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

    n, dim, k, nq = args.rows, args.dim, args.k, args.queries
    lo, hi = shard_bounds(n, world, rank, align=100)  # keep documents (100 rows) whole per shard
    t0 = time.time()
    x, blk, doc = sift_like_corpus(hi - lo, dim, seed=args.seed, start=lo)
    rbac = tree_rbac(num_users=1000, num_roles=100, num_docs=n // 100, seed=args.seed)
    qrow, quser = sample_queries(nq, n, 1000, seed=args.seed)
    qvec = sift_like_rows_at(qrow, dim, args.seed)    # query vectors = corpus rows (read_dataset_function.py:736-737)
    t_gen = time.time() - t0

    ctx = vsrbac.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    corpus = ctx.load_corpus(x, blk, doc, row_offset=lo)
    corpus.load_rbac(rbac.user_roles, rbac.permissions)
    mode = vsrbac.RANGES if args.mode == "prefilter" else vsrbac.BITMAP
    filters = corpus.pack_filters([corpus.filter_for_user(int(u), mode) for u in quser])
    t_load = time.time() - t0 - t_gen

    d_q = torch.from_numpy(qvec).to(dev)
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
    n_sess = 1 if rehearsal else max(1, min(4, int(os.environ.get("VSR_BENCH_SESSIONS", "2"))))
    nbuf = max(2 if overlap else 1, n_sess)           # one result record (and gathered buffer) per batch in flight

    def views(pack):
        return (pack[0:nk * 8].view(torch.int64).view(nq, k),            # raw u64 ordering keys
                pack[nk * 8:nk * 16].view(torch.int64).view(nq, k),
                pack[nk * 16:nk * 20].view(torch.int32).view(nq, k),
                pack[nk * 20:nk * 24].view(torch.float32).view(nq, k))

    d_packs = [torch.empty((rec,), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    d_views = [views(pk) for pk in d_packs]
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
    if parts > 1:
        g_packs = [torch.full((parts * rec,), 0xFF, dtype=torch.uint8, device=dev) for _ in range(nbuf)]   # [parts] records
        m_blk = torch.empty((nq, k), dtype=torch.int64, device=dev)
        m_doc = torch.empty((nq, k), dtype=torch.int32, device=dev)
        m_dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
        m_cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
        m_keys = torch.empty((nq, k), dtype=torch.int64, device=dev)
    if overlap:
        s_comm = torch.cuda.Stream(device=dev)
        mctx = vsrbac.Context(local_rank)                         # the merge runs on the exchange stream
        mctx.set_stream(s_comm.cuda_stream)
        ev_scan = [torch.cuda.Event() for _ in range(nbuf)]       # record b holds the results of its batch
        ev_sent = [torch.cuda.Event() for _ in range(nbuf)]       # record b has been read by the exchange
    step_no = [0]
    mode = {"overlap": overlap, "n_sess": n_sess}     # downgraded to the plain serial loop if the warm-up fails (below)

    def step():
        i = step_no[0]
        step_no[0] += 1
        overlap, n_sess = mode["overlap"], mode["n_sess"]
        b = i % nbuf
        keys_b, blk_b, doc_b, dist_b = d_views[b]
        sess, st = sessions[b % n_sess], s_scan[b % n_sess]
        if overlap and i >= nbuf:
            st.wait_event(ev_sent[b])                             # the batch that used record b before has left it
        corpus.search_device(ptr(d_q), nq, k, "l2", filters, ptr(blk_b), ptr(doc_b), ptr(d_rows[b]), ptr(dist_b),
                             ptr(d_cnts[b]), ptr(keys_b), session=sess)
        if world > 1 and rehearsal:
            torch.cuda.synchronize()
            hg = torch.empty((world * rec,), dtype=torch.uint8)
            dist.all_gather_into_tensor(hg, d_packs[0].cpu())
            g_packs[0].copy_(hg)
            ctx.merge_topk_packed_device(ptr(g_packs[0]), world, nq, k, ptr(m_blk), ptr(m_doc), ptr(m_dist), ptr(m_keys),
                                         ptr(m_cnt))
        elif overlap:
            ev_scan[b].record(st)
            with torch.cuda.stream(s_comm):
                s_comm.wait_event(ev_scan[b])
                if world > 1:
                    dist.all_gather_into_tensor(g_packs[b], d_packs[b])   # RCCL over xGMI: nq*k*24 bytes per rank
                else:
                    g_packs[b][0:rec].copy_(d_packs[b], non_blocking=True)
                ev_sent[b].record(s_comm)
                mctx.merge_topk_packed_device(ptr(g_packs[b]), parts, nq, k, ptr(m_blk), ptr(m_doc), ptr(m_dist),
                                              ptr(m_keys), ptr(m_cnt))
        elif world > 1:                                           # plain loop: scan, exchange, merge on one stream
            dist.all_gather_into_tensor(g_packs[b], d_packs[b])
            ctx.merge_topk_packed_device(ptr(g_packs[b]), parts, nq, k, ptr(m_blk), ptr(m_doc), ptr(m_dist),
                                         ptr(m_keys), ptr(m_cnt))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    try:
        for _ in range(args.warmup):
            step()
        barrier()
    except Exception as exc:                          # never lose the run to the overlapped choreography
        if not (mode["overlap"] or mode["n_sess"] > 1):
            raise
        print(f"[bench] overlapped loop failed in warm-up ({exc!r}); falling back to one batch in flight, serial "
              f"exchange", file=sys.stderr, flush=True)
        mode["overlap"], mode["n_sess"] = False, 1
        torch.cuda.synchronize()
        for _ in range(args.warmup):
            step()
        barrier()
    overlap, n_sess = mode["overlap"], mode["n_sess"]
    for sess in sessions:
        sess.profiling(2)                             # events around the main scan launch only (the roofline kernel)
        sess.stats_reset()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enq = time.perf_counter() - t1                  # host time to enqueue the whole run (must stay below dt)
    barrier()
    dt = time.perf_counter() - t1
    st = None
    flagged_total = 0
    for sess in sessions:
        one = sess.stats()
        sess.profiling(False)
        flagged_total += sess.screening_check(0)[0]   # K2 / seeding exactness flags over the whole run (expect 0)
        if st is None:
            st = one
        else:
            for key in ("scan_launches", "scan_ms", "scan_bytes", "scan_rows"):
                st[key] = [a + b for a, b in zip(st[key], one[key])]
    if world > 1:
        ft = torch.tensor([flagged_total], dtype=torch.int64, device=dev if not rehearsal else "cpu")
        dist.all_reduce(ft, op=dist.ReduceOp.SUM)
        flagged_total = int(ft.item())
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if not rehearsal else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- the same launch alone on the GPU: a few more batches, one in flight, same events (kernel quality, not value) ----
    alone = None
    if n_sess > 1:
        ctx.profiling(2)
        ctx.stats_reset()
        for _ in range(5):
            corpus.search_device(ptr(d_q), nq, k, "l2", filters, ptr(d_views[0][1]), ptr(d_views[0][2]), ptr(d_rows[0]),
                                 ptr(d_views[0][3]), ptr(d_cnts[0]), ptr(d_views[0][0]))
        torch.cuda.synchronize()
        alone = ctx.stats()
        ctx.profiling(False)

    # ---- roofline of the dominant K1 kernel class (HIP events on the launch stream) ----
    cls = int(np.argmax(st["scan_ms"]))
    launches = max(1, st["scan_launches"][cls])
    ms_avg = st["scan_ms"][cls] / launches
    bytes_per_launch = st["scan_bytes"][cls] / launches
    achieved = bytes_per_launch / (ms_avg * 1e-3) / 1e9 if ms_avg > 0 else 0.0
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
        "kernel": "vsr::mfma_scan_kernel<L2, NSTR=2, SAMPLE=false, NG=1> (K2 main pass)" if cls else
                  "vsr::scan_kernel<L2, LPR=32, C=1, R=8, QI=1>",
        "launch_ms": round(ms_avg, 4), "bytes_per_launch": int(bytes_per_launch),
        "launches": int(launches),
        "all_scan_ms": [round(v, 3) for v in st["scan_ms"]], "all_scan_bytes": [int(v) for v in st["scan_bytes"]],
    }

    roofline["wall_rate"] = round(sum(st["scan_bytes"]) / dt / 1e9, 1)      # pass bytes per second of wall time, GB/s
    if n_sess > 1:
        roofline["note"] = (f"{n_sess} batches in flight: a launch's event-timed duration includes the time it shares the GPU "
                            "with the other batch's kernels; `alone` is the same launch with one batch in flight")
        if alone and alone["scan_launches"][cls]:
            a_ms = alone["scan_ms"][cls] / alone["scan_launches"][cls]
            a_by = alone["scan_bytes"][cls] / alone["scan_launches"][cls]
            roofline["alone"] = {"launch_ms": round(a_ms, 4), "achieved": round(a_by / (a_ms * 1e-3) / 1e9, 1),
                                 "frac": round(a_by / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "launches": int(alone["scan_launches"][cls])}
    workload_tag = f"{n}x{dim} k={k} q={nq} {args.mode} gpus={world}"
    try:      # HBM bytes per launch from a PMC pass of this same command (never measured inside the timed run)
        with open(args.traffic) as f:
            tr = json.load(f)
        if tr.get("workload") == workload_tag:
            for name, v in tr["kernels"].items():
                main = ("mfma_scan_kernel" in name and "false" in name) if cls else ("vsr::scan_kernel" in name)
                if main:
                    roofline["traffic"] = int(v["hbm_bytes_per_launch"])
                    roofline["traffic_source"] = os.path.relpath(args.traffic, ROOT) + ": " + tr["method"]
    except (OSError, ValueError, KeyError):
        pass
    roofline["workload_tag"] = workload_tag

    out = {
        "metric": "QPS at recall@100, SIFT10M filtered-kNN (role RBAC), 1/2/4/8 MI355X",
        "value": round(nq * args.steps / dt, 1), "unit": "queries/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"SIFT10M-like {n}x{dim} fp32 L2 k={k}, tree RBAC 1000 users/100 roles, "
                               f"role-partition {args.mode}, exact (recall@{k} = 1.0), {nq} queries/step",
                   "rows": n, "dim": dim, "k": k, "queries_per_step": nq, "filter": args.mode,
                   "sharding": f"row-range x{world}", "recall": 1.0},
        "roofline": roofline,
        "setup_s": {"generate": round(t_gen, 1), "load": round(t_load, 1)},
        "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 4),
        "screening_flagged_queries": int(flagged_total),
    }

    d_keys, d_blk, d_doc, d_dist = d_views[0]         # slot 0 / session 0 from here on (everything above has drained)
    d_row, d_cnt = d_rows[0], d_cnts[0]
    # ---- latency mode (informational, N = 1): the harness's call shape, one query per call ----
    if world == 1 and sim_world <= 1:
        m1 = min(200, nq)
        singles = [corpus.pack_filters([filters._keep[i]]) for i in range(m1)]
        for i in range(8):
            corpus.search_device(ptr(d_q[i:i + 1]), 1, k, "l2", singles[i], ptr(d_blk), ptr(d_doc), ptr(d_row), ptr(d_dist),
                                 ptr(d_cnt), ptr(d_keys))
        torch.cuda.synchronize()
        tl = time.perf_counter()
        for i in range(m1):
            corpus.search_device(ptr(d_q[i:i + 1]), 1, k, "l2", singles[i], ptr(d_blk), ptr(d_doc), ptr(d_row), ptr(d_dist),
                                 ptr(d_cnt), ptr(d_keys))
        torch.cuda.synchronize()
        tl = time.perf_counter() - tl
        out["single_query_mode"] = {"queries": m1, "ms_per_query": round(tl / m1 * 1e3, 4), "qps": round(m1 / tl, 1),
                                    "note": "one query per call, back to back on one stream (K1 path); not the headline"}
        corpus.search_device(ptr(d_q), nq, k, "l2", filters, ptr(d_blk), ptr(d_doc), ptr(d_row), ptr(d_dist),
                             ptr(d_cnt), ptr(d_keys))       # leave the batch result in place for the parity spot check
        torch.cuda.synchronize()

    # ---- CPU baseline (rank 0, N = 1): the oracle, pgvector's flags, one thread, bounded sample ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.oracle import Oracle
        orc = Oracle("pgflags")
        m = min(args.cpu_queries, nq)
        ranges = []
        for u in quser[:m]:
            docs = rbac.visible_docs(int(u)).astype(np.int64)
            ranges.append([((d - 1) * 100, 100) for d in docs])
        tc = time.perf_counter()
        for _ in range(max(1, args.cpu_reps)):
            rows_o, dist_o, cnt_o = orc.search_ranges("l2", x, qvec[:m], k, ranges, doc, blk)
        cpu_s = (time.perf_counter() - tc) / max(1, args.cpu_reps)
        # the same sample fanned over host threads, one query per thread at a time (the C call releases the GIL):
        # an all-cores figure beside the single-core one; never the headline
        import concurrent.futures
        threads = max(1, min(16, os.cpu_count() or 1))
        cuts = np.linspace(0, m, threads + 1).astype(int)
        ta = time.perf_counter()
        with concurrent.futures.ThreadPoolExecutor(threads) as pool:
            list(pool.map(lambda se: orc.search_ranges("l2", x, qvec[se[0]:se[1]], k, ranges[se[0]:se[1]], doc, blk),
                          [(int(a), int(b)) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]))
        cpu_all_s = time.perf_counter() - ta
        got_rows = d_row[:m].cpu().numpy()
        got_dist = d_dist[:m].cpu().numpy()
        ok = bool((got_rows == rows_o).all() and (got_dist == dist_o.astype(np.float32)).all())
        out["cpu_baseline"] = {
            "value": round(m / cpu_s, 2), "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": f"first {m} queries of the step x {max(1, args.cpu_reps)} repetitions, exact seq-scan of each "
                      f"user's role partition, {cpu_s * max(1, args.cpu_reps):.1f} s on 1 of {os.cpu_count()} host "
                      f"cores (oracle built with pgvector's flags)",
        }
        out["cpu_baseline"]["all_cores"] = {"value": round(m / cpu_all_s, 1), "unit": "queries/s", "threads": threads,
                                            "note": "same sample, queries fanned over host threads"}
        out["parity_spot_check"] = {"queries": int(m), "ids_and_distances_identical": ok}
    out["config"]["batches_in_flight"] = n_sess
    if parts > 1:
        out["config"]["exchange"] = ("all-gather + merge of batch i overlapped with the scan of batch i+1 (second stream)"
                                     if overlap else "serial (rehearsal through host memory)")
    if sim_world > 1:      # development check of the overlapped choreography on one GPU
        last = d_views[(step_no[0] - 1) % nbuf]      # (the latency-mode loop below does not run in this mode)
        out["sim_world"] = {"parts": parts, "merged_equals_local": bool(torch.equal(m_keys, last[0]) and
                                                                        torch.equal(m_blk, last[1]) and
                                                                        torch.equal(m_dist, last[3]))}
    if rank == 0 and world > 1 and os.environ.get("VSR_BENCH_VERIFY") == "1":
        # rehearsal check: the merged multi-rank result of a few queries against the oracle on the full corpus
        from oracle.oracle import Oracle
        orc = Oracle("pgflags")
        xf, blkf, docf = sift_like_corpus(n, dim, seed=args.seed)
        m = 8
        ranges = [[((d - 1) * 100, 100) for d in rbac.visible_docs(int(u)).astype(np.int64)] for u in quser[:m]]
        rows_o, dist_o, _ = orc.search_ranges("l2", xf, qvec[:m], k, ranges, docf, blkf)
        ok = bool((m_blk[:m].cpu().numpy() == blkf[rows_o]).all() and
                  (m_dist[:m].cpu().numpy() == dist_o.astype(np.float32)).all())
        out["multi_rank_parity"] = {"queries": m, "ids_and_distances_identical": ok}
        if not ok:
            got_b, got_d = m_blk[:m].cpu().numpy(), m_dist[:m].cpu().numpy()
            bad = np.argwhere(got_b != blkf[rows_o])
            print("multi-rank mismatch at", bad[:5].tolist(), "got", got_b[0, :5].tolist(), got_d[0, :5].tolist(),
                  "want", blkf[rows_o][0, :5].tolist(), dist_o[0, :5].tolist(), "counts", m_cnt[:m].cpu().tolist(),
                  file=sys.stderr)
    if rank == 0:
        print(json.dumps(out), flush=True)
    corpus.free()
    if overlap:
        mctx.close()
    for cx in sessions[1:]:
        cx.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
