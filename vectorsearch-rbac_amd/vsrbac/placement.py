"""Planner integration: a GPU cost model the partition planner can optimise against, and the placement of
its partitions on the GPUs of one node.

The reference's planner (controller/dynamic_partition/hnsw/AnonySys_dynamic_partition.py) scores a candidate
partitioning with `compute_query_time` (:114-166): per role combination, sum over the partitions it must search of
`weight * log(n) * (a * ef_search + b)` -- the cost of an HNSW probe of a partition of n documents.  On MI355X the search
of a partition is an exact scan, so the same objective becomes `weight * (rows * bytes_per_row / scan_rate + fixed)`:
`compute_query_time_gpu` has the reference function's argument list (the HNSW-only tuning parameters are accepted and
ignored), so the planner's greedy loop (:425-672) can call it unchanged.

Placement is the alternative to row-range sharding that SURVEY §8(e)(ii) names: whole partitions are assigned to GPUs
(LPT bin packing on the expected scan time each partition attracts), partitions hot enough to dominate a GPU are
replicated, and a query touches only the GPUs that hold the partitions of its role combination
(controller/dynamic_partition/search.py:54-58 searches exactly those partition tables).  `PlacedDeployment` runs that
over one `Deployment` per GPU and merges like search.py:347-364 (dedup on (document_id, block_id): replicas and impure
partitions overlap).
"""
import heapq
import math
from collections import defaultdict

import numpy as np

from .harness import Deployment, merge_results, parse_vector

# measured on one MI355X (profiles/r2): what a partition scan costs
SCAN_RATE_BYTES_PER_US = 4.4e6        # K2w main launch alone, 4.4 TB/s of plane bytes (batched callers)
SINGLE_QUERY_RATE_BYTES_PER_US = 3.5e6   # K1, one query per call, role-partition sized passes
FIXED_US_PER_CALL = 60.0              # launch + in-kernel merge of a one-query call
FIXED_US_PER_PARTITION = 2.0          # one more pass inside a call (tile list, workgroup ramp)


def partition_scan_us(rows, dim, batched=False):
    """Expected device time to scan one partition of `rows` rows for one query (batched: its share of a shared pass)."""
    if batched:
        return rows * (2.0 * dim + 4) / SCAN_RATE_BYTES_PER_US
    return rows * 4.0 * dim / SINGLE_QUERY_RATE_BYTES_PER_US + FIXED_US_PER_PARTITION


def compute_query_time_gpu(comb_trackers, loads, sel_whole=None, topk=None, k=None, beta=None, a=None, b=None,
                           comb_to_update=None, role_weights=None, recall=None, dim=128, rows_per_doc=1.0,
                           batched=False):
    """Drop-in for AnonySys_dynamic_partition.compute_query_time (:114-166) with the GPU's cost of a partition search.
    `loads` counts documents per partition as in the reference; `rows_per_doc` converts to rows."""
    total = 0.0
    combs = comb_to_update if comb_to_update is not None else comb_trackers.keys()
    for comb in combs:
        weight = role_weights.get(comb, 0) if role_weights else 1
        if weight == 0 and role_weights:                      # single-role mode of the reference (:158-159)
            weight = role_weights.get(next(iter(comb)), 1) if comb else 0
        parts = [p for p in comb_trackers.get(comb, ()) if p in loads]
        if not parts:
            continue
        call = 0.0 if batched else FIXED_US_PER_CALL
        total += weight * (call + sum(partition_scan_us(loads[p] * rows_per_doc, dim, batched) for p in parts))
    return total


def partition_heat(comb_trackers, loads, role_weights=None, dim=128, rows_per_doc=1.0):
    """Expected scan time each partition attracts per unit of query weight: sum over the combinations that search it."""
    heat = defaultdict(float)
    for comb, parts in comb_trackers.items():
        w = role_weights.get(comb, 1.0) if role_weights else 1.0
        for p in parts:
            if p in loads:
                heat[p] += w * partition_scan_us(loads[p] * rows_per_doc, dim)
    return dict(heat)


def place_partitions(loads, heat, n_gpus, replicate_above=0.5, mem_rows=None):
    """LPT bin packing of partitions on GPUs by the scan time they attract; a partition whose heat exceeds
    `replicate_above` x the per-GPU average is replicated (its heat split) on the GPUs with the least load until each
    copy is below the bound.  Returns ({partition: [gpu, ...]}, per-GPU heat, per-GPU rows).
    mem_rows: optional cap on rows per GPU (288 GB of HBM hold ~500 M SIFT rows: rarely binding)."""
    n_gpus = int(n_gpus)
    total = sum(heat.get(p, 0.0) for p in loads)
    avg = total / max(1, n_gpus)
    copies = {}
    for p in loads:
        h = heat.get(p, 0.0)
        c = 1
        if n_gpus > 1 and avg > 0:
            c = min(n_gpus, max(1, int(math.ceil(h / (replicate_above * avg)))))
        copies[p] = c
    gpu_heat = [0.0] * n_gpus
    gpu_rows = [0] * n_gpus
    placement = {}
    order = sorted(loads, key=lambda p: (-heat.get(p, 0.0) / copies[p], p))     # longest processing time first
    for p in order:
        share = heat.get(p, 0.0) / copies[p]
        chosen = []
        cand = [(gpu_heat[g], g) for g in range(n_gpus)]
        heapq.heapify(cand)
        while len(chosen) < copies[p] and cand:
            _, g = heapq.heappop(cand)
            if mem_rows is not None and gpu_rows[g] + loads[p] > mem_rows:
                continue
            chosen.append(g)
        if not chosen:
            raise ValueError(f"partition {p} ({loads[p]} rows) fits no GPU under mem_rows = {mem_rows}")
        share = heat.get(p, 0.0) / len(chosen)
        for g in chosen:
            gpu_heat[g] += share
            gpu_rows[g] += loads[p]
        placement[p] = sorted(chosen)
    return placement, gpu_heat, gpu_rows


class PlacedDeployment:
    """The reference's dynamic-partition search (search.py:31-111) over partitions placed on several GPUs: one
    `Deployment` (resident corpus + RBAC tables) per GPU holding only the documents of its partitions; a query goes to
    the GPUs holding its combination's partitions (for a replicated partition: the replica with the least work queued in
    this call), and the per-GPU rows are merged with dedup.  `contexts`: one vsrbac.Context per GPU (several contexts of
    one GPU are fine: that is how the tests run it on a one-GPU box)."""

    def __init__(self, contexts, rows, block_ids, doc_ids, user_roles, permissions, partition_docs, comb_role_partitions,
                 role_weights=None, metric="l2", replicate_above=0.5):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        block_ids = np.asarray(block_ids, dtype=np.int64)
        doc_ids = np.asarray(doc_ids, dtype=np.int32)
        self.n_gpus = len(contexts)
        self.partition_docs = {int(p): np.asarray(d, dtype=np.int32) for p, d in partition_docs.items()}
        self.comb_role_partitions = {tuple(sorted(int(r) for r in c)): sorted(int(p) for p in ps)
                                     for c, ps in comb_role_partitions.items()}
        self.user_roles = defaultdict(set)
        for u, r in user_roles:
            self.user_roles[int(u)].add(int(r))
        doc_rows = defaultdict(int)
        for d in doc_ids:
            doc_rows[int(d)] += 1
        loads = {p: int(sum(doc_rows[int(d)] for d in docs)) for p, docs in self.partition_docs.items()}   # rows
        heat = partition_heat(self.comb_role_partitions, loads, role_weights, rows.shape[1])
        self.placement, self.gpu_heat, self.gpu_rows = place_partitions(loads, heat, self.n_gpus, replicate_above)
        self.deployments = []
        self._pool = None                                 # host threads, one per GPU a query touches (created on first use)
        for g, ctx in enumerate(contexts):
            mine = [p for p, gs in self.placement.items() if g in gs]
            docs = np.unique(np.concatenate([self.partition_docs[p] for p in mine])) if mine else np.zeros(0, np.int32)
            sel = np.isin(doc_ids, docs)
            if not sel.any():
                self.deployments.append(None)
                continue
            dset = set(int(d) for d in docs)
            perms = [(r, d) for r, d in permissions if int(d) in dset]
            dep = Deployment(ctx, rows[sel], block_ids[sel], doc_ids[sel], user_roles, perms, metric=metric)
            dep.load_partitions({p: self.partition_docs[p] for p in mine},
                                {c: [p for p in ps if p in mine] for c, ps in self.comb_role_partitions.items()})
            self.deployments.append(dep)

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        for d in self.deployments:
            if d is not None:
                d.close()

    def route(self, comb):
        """{gpu: [partition, ...]} for one query of role combination `comb`: every partition once, replicas balanced."""
        queued = defaultdict(float)
        out = defaultdict(list)
        for p in self.comb_role_partitions.get(comb, []):
            gs = self.placement[p]
            g = min(gs, key=lambda x: (queued[x], x))
            queued[g] += 1.0
            out[g].append(p)
        return dict(out)

    def dynamic_partition_search(self, user_id, query_vector, topk=5, statistics_type="sql"):
        q = parse_vector(query_vector)
        comb = tuple(sorted(self.user_roles.get(int(user_id), set())))
        plan = self.route(comb)
        if not plan:
            return [], 0.0
        def on_gpu(item):
            g, pids = item
            dep = self.deployments[g]
            filters = [dep._partition_filter(p, user_id) for p in pids]
            qs = np.repeat(q[None, :], len(pids), axis=0)
            res, t = dep._timed(statistics_type, lambda: dep.corpus.search(qs, topk, dep.metric, filters))
            rows = []
            for i in range(len(pids)):
                rows.extend(dep._rows(res, i))
            return rows, t

        # the GPUs of a query work side by side: one host thread per GPU (the C call releases the GIL), so the slowest GPU's
        # time is the query's time -- issued one after another (as this loop once did) the times would add up
        items = list(plan.items())
        if len(items) == 1:
            parts = [on_gpu(items[0])]
        else:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=len(self.deployments))
            parts = list(self._pool.map(on_gpu, items))
        all_rows = [r for rows, _ in parts for r in rows]
        return merge_results(all_rows, topk), max(t for _, t in parts)
