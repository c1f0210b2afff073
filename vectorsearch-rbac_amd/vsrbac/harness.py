"""Host-side mirror of the reference's benchmark harness for the filtered k-NN path.

The reference drives PostgreSQL through `search_func(user_id, query_vector, topk, statistics_type)
-> (rows, seconds)` (registry: basic_benchmark/condition_config.py:12-38) where a row is
`(block_id, document_id, block_content, distance)`; `run_search_experiment`
(basic_benchmark/common_function.py:1321-1434) calls it twice to warm up and once measured per query and
scores recall on `(document_id, block_id)` sets (:1154-1160, :1394-1396).  `Deployment` offers the same three
search functions over libvsrbac:

  search_documents_role_partition   controller/baseline/prefilter/prefilter_role.py:22-26,81-195
  search_documents_rls              controller/baseline/pg_row_security/row_level_security.py:98-160
  dynamic_partition_search          controller/dynamic_partition/search.py:31-111,347-364

Every search is exact (recall 1.0): there is no ef_search / probes knob to trade recall for time.
"""
import json
import time

import numpy as np

from .engine import BITMAP, RANGES
from .formats import read_shared_vectors, vector_from_text  # noqa: F401


def parse_vector(v):
    """Accepts pgvector's text form '[a,b,...]' (what the reference passes through SQL; parsed with vector_in's grammar,
    number parser and error texts, formats.vector_from_text) or a sequence of numbers."""
    if isinstance(v, (str, bytes)):
        return vector_from_text(v)
    return np.asarray(v, dtype=np.float32)


def planner_output_to_tables(partition_assignment, comb_role_trackers):
    """(partition_docs, comb_role_partitions) in the shape `Deployment.load_partitions` takes, from the planner's dicts:
    documents per partition sorted, one (combination -> partition ids) row set like the CombRolePartitions table."""
    partition_docs = {int(p): sorted(int(d) for d in docs) for p, docs in partition_assignment.items()}
    combs = {}
    for comb, parts in comb_role_trackers.items():
        key = tuple(sorted(int(r) for r in (comb if isinstance(comb, (tuple, list, set, frozenset)) else (comb,))))
        pids = sorted(int(p) for p in (parts.keys() if hasattr(parts, "keys") else parts))
        missing = [p for p in pids if p not in partition_docs]
        if missing:
            raise ValueError(f"combination {key} refers to unknown partitions {missing}")
        combs[key] = pids
    return partition_docs, combs


def load_query_dataset(path):
    """query_dataset.json: list of {user_id, query_vector, topk, query_block_selectivity}
    (services/read_dataset_function.py:705-710,1049-1062)."""
    with open(path) as f:
        return json.load(f)


def load_ground_truth_cache(path):
    """ground_truth_cache.json in either of the reference's two formats: a list of
    {"query": ..., "ground_truth": [[block_id, document_id, ...], ...]} dicts
    (basic_benchmark/common_function.py:831-849) or the "pointer" format, a list of
    [[block_id, document_id], ...] lists (basic_benchmark/compute_ground_truth.py:35-59).
    Returns, per query, a list of (block_id, document_id)."""
    with open(path) as f:
        data = json.load(f)
    out = []
    for entry in data:
        rows = entry.get("ground_truth", entry.get("results", [])) if isinstance(entry, dict) else entry
        out.append([(int(r[0]), int(r[1])) for r in rows])
    return out


def compute_recall(true_results, predicted_results):
    """|GT ∩ pred| / |GT| over (document_id, block_id) sets (common_function.py:1154-1160)."""
    true_set, predicted_set = set(true_results), set(predicted_results)
    return len(true_set & predicted_set) / len(true_set)


def merge_results(all_results, topk):
    """Stable sort by distance, first occurrence of (document_id, block_id) wins, stop at topk
    (controller/dynamic_partition/search.py:347-364)."""
    seen, unique = set(), []
    for row in sorted(all_results, key=lambda r: (np.isnan(r[3]), r[3])):
        key = (row[1], row[0])
        if key not in seen:
            seen.add(key)
            unique.append(row)
        if len(unique) == topk:
            break
    return unique


class Deployment:
    """A corpus + RBAC tables resident on one GPU, exposing the reference's search functions."""

    def __init__(self, ctx, rows, block_ids, doc_ids, user_roles, permissions, block_content=None, metric="l2"):
        """`metric`: the operator of the search SQL.  The reference's harness issues `vector <-> %s` (L2) everywhere
        (prefilter_role.py:128-172, search.py:86-97); BASELINE config 3 asks for cosine (`<=>`)."""
        self.ctx = ctx
        self.metric = metric
        self.corpus = ctx.load_corpus(rows, block_ids, doc_ids)
        self.corpus.load_rbac(user_roles, permissions)
        self.block_content = block_content          # optional: row index -> text (the table's block_content column)
        self.user_roles = {}
        for u, r in np.asarray(user_roles, dtype=np.int64).reshape(-1, 2):
            self.user_roles.setdefault(int(u), set()).add(int(r))
        self.partitions = {}                         # partition_id -> Filter factory inputs
        self.comb_role_partitions = {}               # sorted role tuple -> [partition_id]

    def close(self):
        self.corpus.free()

    # ---- result shaping --------------------------------------------------------------------
    def _rows(self, res, qi=0):
        m = int(res.counts[qi])
        out = []
        for j in range(m):
            r = int(res.rows[qi, j])
            content = self.block_content[r] if self.block_content is not None else None
            out.append((int(res.block_ids[qi, j]), int(res.doc_ids[qi, j]), content, float(res.dist[qi, j])))
        return out

    def _timed(self, statistics_type, fn):
        """'sql' -> device time of the search (the analogue of EXPLAIN ANALYZE's "Execution Time", which
        excludes connection and client overhead); 'system' -> wall clock around the call."""
        if statistics_type == "system":
            t = time.perf_counter()
            res = fn()
            return res, time.perf_counter() - t
        self.ctx.profiling(True)
        self.ctx.stats_reset()
        res = fn()
        st = self.ctx.stats()
        self.ctx.profiling(False)
        return res, st["search_ms"] * 1e-3

    # ---- ROLE pre-filter ---------------------------------------------------------------------
    def search_documents_role_partition(self, user_id, query_vector, topk=5, statistics_type="sql"):
        q = parse_vector(query_vector)
        f = self.corpus.filter_for_user(user_id, RANGES)     # union of the user's role partitions, each row once
        res, secs = self._timed(statistics_type, lambda: self.corpus.search(q, topk, self.metric, [f]))
        return self._rows(res), secs

    # ---- RLS post-filter ----------------------------------------------------------------------
    def search_documents_rls(self, user_id, query_vector, topk=5, statistics_type="sql"):
        q = parse_vector(query_vector)
        f = self.corpus.filter_for_user(user_id, BITMAP)
        res, secs = self._timed(statistics_type, lambda: self.corpus.search(q, topk, self.metric, [f]))
        return self._rows(res), secs

    # ---- dynamic partitions ---------------------------------------------------------------------
    def load_planner_output(self, partition_assignment, comb_role_trackers):
        """The in-memory result of the reference's partition planner, as its loader receives it
        (initialize_partitions_and_role_mappings, load_result_to_database.py:286-299): `partition_assignment`
        {partition_id: set(document_id)} and `comb_role_trackers` {role combination: {partition_id: set(roles)}}
        (AnonySys_dynamic_partition.py:312-320)."""
        self.load_partitions(*planner_output_to_tables(partition_assignment, comb_role_trackers))

    def load_partitions(self, partition_docs, comb_role_partitions):
        """partition_docs: {partition_id: [document_id,...]} (load_result_to_database.py:207-240);
        comb_role_partitions: {(role,...): [partition_id,...]} (CombRolePartitions, :293-299)."""
        self.partitions = {int(p): np.asarray(d, dtype=np.int32) for p, d in partition_docs.items()}
        self.comb_role_partitions = {tuple(sorted(int(r) for r in c)): [int(p) for p in ps]
                                     for c, ps in comb_role_partitions.items()}
        self._part_filters = {}

    def _partition_filter(self, pid, user_id):
        """Pure partitions (every document visible to the combination) carry no per-row test; impure ones
        get the user's permission bits (load_result_to_database.py:590-624)."""
        docs = self.partitions[pid]
        roles = self.user_roles.get(int(user_id), set())
        key = (pid, tuple(sorted(roles)))
        f = self._part_filters.get(key)
        if f is None:
            pure = self.corpus.filter_from_documents(docs, -1)
            restricted = self.corpus.filter_from_documents(docs, int(user_id))
            if restricted.allowed_rows == pure.allowed_rows:
                f = pure
                restricted.free()
            else:
                f = restricted
                pure.free()
            self._part_filters[key] = f
        return f

    def dynamic_partition_search(self, user_id, query_vector, topk=5, statistics_type="sql"):
        q = parse_vector(query_vector)
        comb = tuple(sorted(self.user_roles.get(int(user_id), set())))
        pids = sorted(set(self.comb_role_partitions.get(comb, [])))
        if not pids:
            return [], 0.0
        filters = [self._partition_filter(p, user_id) for p in pids]
        qs = np.repeat(q[None, :], len(pids), axis=0)
        res, secs = self._timed(statistics_type, lambda: self.corpus.search(qs, topk, self.metric, filters))
        all_rows = []
        for i in range(len(pids)):
            all_rows.extend(self._rows(res, i))
        return merge_results(all_rows, topk), secs

    # ---- ground truth (exact scan; the reference's PostgreSQL seq-scan path, common_function.py:671-759) ----
    def ground_truth(self, user_id, query_vector, topk):
        rows, _ = self.search_documents_rls(user_id, query_vector, topk, "system")
        return rows


def run_search_experiment(queries, search_func, ground_truth_func=None, queries_num=None, statistics_type="sql",
                          iterations=1, record_recall=True, warm_up=True):
    """Counterpart of basic_benchmark/common_function.py:1321-1434: per query two warm-up calls and one
    measured call; recall against the ground truth on (document_id, block_id) sets; returns the same
    aggregate keys (avg_recall, avg_query_time) plus qps = 1 / avg_query_time (:1414) and the per-query list."""
    todo = queries[:queries_num] if queries_num else queries
    all_results, total_recall, total_time = [], 0.0, 0.0
    for query in todo:
        user_id, qv, topk = query["user_id"], query["query_vector"], query.get("topk", 5)
        gt = ground_truth_func(user_id, qv, topk) if (record_recall and ground_truth_func) else None
        q_recall, q_time = 0.0, 0.0
        for _ in range(iterations):
            if warm_up:
                for _ in range(2):
                    search_func(user_id=user_id, query_vector=qv, topk=topk, statistics_type=statistics_type)
            results, query_time = search_func(user_id=user_id, query_vector=qv, topk=topk,
                                              statistics_type=statistics_type)
            if gt is not None:
                pred = set((r[1], r[0]) for r in results)
                true = set((g[1], g[0]) for g in gt)
                q_recall += compute_recall(true, pred) if true else 1.0
            q_time += query_time
        avg_r = q_recall / iterations if iterations else 0
        avg_t = q_time / iterations if iterations else 0
        all_results.append({"user_id": user_id, "query_vector": qv, "recall": avg_r, "query_time": avg_t,
                            "qps": 1 / avg_t if avg_t > 0 else 0})
        total_recall += avg_r
        total_time += avg_t
    n = len(todo)
    avg_recall = total_recall / n if n else 0
    avg_time = total_time / n if n else 0
    return {"avg_recall": avg_recall, "avg_query_time": avg_time, "qps": 1 / avg_time if avg_time > 0 else 0,
            "all_results": all_results}
