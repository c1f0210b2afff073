"""Seeded synthetic inputs shaped like the reference's datasets (no datasets exist offline).

* SIFT-like rows: integer-valued fp32 in [0,255], d = 128, `block_id = row + 1`,
  `document_id = row // 100 + 1` (services/read_dataset_function.py:27,336-339).
* Wikipedia-like rows: i.i.d. N(0,1), optionally L2-normalised (cosine config).
* Tree RBAC: the semantics of services/rbac_generator/tree_based_rbac_data_generator.py:22-217 restated
  with a seeded numpy Generator (the reference draws from the unseeded global `random`):
  roles are consumed in id order into a random tree of height h with b0..b1 children per node (:48-74);
  documents are shuffled and cut into (#nodes) equal disjoint sets, the last taking the remainder (:88-110);
  a role sees its own set and every ancestor's (:113-151); users are split evenly over the roles in
  pre-order with exactly one role each (:164-187).
* Queries: uniformly sampled corpus rows paired with uniformly sampled users
  (services/read_dataset_function.py:736-743).

Rows are generated in fixed 1M-row chunks seeded by (seed, chunk) so that any shard of the corpus can be
produced independently and is identical whatever the number of GPUs.
"""
import os
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field

import numpy as np

CHUNK = 1_000_000


def _sift_chunk(seed, chunk, rows, dim):
    rng = np.random.default_rng([int(seed), int(chunk)])
    x = rng.standard_normal((rows, dim), dtype=np.float32)
    np.abs(x, out=x)
    x *= 45.0
    np.rint(x, out=x)
    np.clip(x, 0, 255, out=x)
    return x


def sift_like_rows(start, stop, dim=128, seed=20251121, workers=None):
    """Rows [start, stop) of the SIFT-like corpus (chunks generated on a small thread pool)."""
    out = np.empty((stop - start, dim), dtype=np.float32)
    jobs = []
    pos = start
    while pos < stop:
        c = pos // CHUNK
        lo = c * CHUNK
        hi = min(stop, lo + CHUNK)
        jobs.append((c, lo, pos, hi))
        pos = hi

    def run(job):
        c, lo, pos, hi = job
        part = _sift_chunk(seed, c, hi - lo, dim)        # a prefix of the chunk's stream: same values
        out[pos - start:hi - start] = part[pos - lo:]

    if len(jobs) == 1:
        run(jobs[0])
    else:
        workers = workers or max(1, min(8, (os.cpu_count() or 2) // 2))
        with ThreadPoolExecutor(max_workers=workers) as pool:
            list(pool.map(run, jobs))
    return out


def sift_like_rows_at(indices, dim=128, seed=20251121, workers=None):
    """The SIFT-like rows at arbitrary row indices (query vectors are sampled corpus rows; a rank of the placement mode
    holds the rows of its roles' documents).  Chunks are generated on a small thread pool, each only as far as needed."""
    indices = np.asarray(indices, dtype=np.int64)
    out = np.empty((indices.size, dim), dtype=np.float32)
    if indices.size == 0:
        return out
    order = np.argsort(indices, kind="stable")
    sorted_idx = indices[order]
    chunk_of = sorted_idx // CHUNK
    cuts = np.flatnonzero(np.diff(chunk_of)) + 1
    starts = np.concatenate([[0], cuts])
    stops = np.concatenate([cuts, [sorted_idx.size]])

    def run(j):
        a, b = int(starts[j]), int(stops[j])
        c = int(chunk_of[a])
        local = sorted_idx[a:b] - c * CHUNK
        part = _sift_chunk(seed, c, int(local[-1]) + 1, dim)
        out[order[a:b]] = part[local]

    if len(starts) == 1:
        run(0)
    else:
        workers = workers or max(1, min(8, (os.cpu_count() or 2) // 2))
        with ThreadPoolExecutor(max_workers=workers) as pool:
            list(pool.map(run, range(len(starts))))
    return out


def sift_like_corpus(n, dim=128, seed=20251121, rows_per_doc=100, start=0):
    """(rows, block_ids, doc_ids) for rows [start, start + n)."""
    x = sift_like_rows(start, start + n, dim, seed)
    r = np.arange(start, start + n, dtype=np.int64)
    return x, r + 1, (r // rows_per_doc + 1).astype(np.int32)


def gaussian_corpus(n, dim, seed=20251121, normalize=False, blocks_per_doc=10):
    rng = np.random.default_rng([int(seed), 7])
    x = rng.standard_normal((n, dim), dtype=np.float32)
    if normalize:
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    r = np.arange(n, dtype=np.int64)
    return x, r + 1, (r // blocks_per_doc + 1).astype(np.int32)


@dataclass
class RBAC:
    user_roles: np.ndarray               # [n, 2] (user_id, role_id)
    permissions: np.ndarray              # [m, 2] (role_id, document_id)
    num_users: int
    role_docs: dict = field(default_factory=dict)       # role_id -> sorted document ids
    parent: dict = field(default_factory=dict)          # role_id -> parent role_id (0 = root)
    _user_roles_map: dict = field(default_factory=dict)

    def roles_of(self, user_id):
        return self._user_roles_map.get(int(user_id), [])

    def visible_docs(self, user_id):
        docs = [self.role_docs[r] for r in self.roles_of(user_id)]
        return np.unique(np.concatenate(docs)) if docs else np.zeros(0, dtype=np.int32)


def _finish(user_roles, role_docs, num_users, parent=None):
    perms = np.concatenate([np.stack([np.full(len(d), r, dtype=np.int32), np.asarray(d, dtype=np.int32)], 1)
                            for r, d in role_docs.items()]) if role_docs else np.zeros((0, 2), np.int32)
    ur = np.asarray(user_roles, dtype=np.int32).reshape(-1, 2)
    m = {}
    for u, r in ur:
        m.setdefault(int(u), []).append(int(r))
    return RBAC(ur, perms, num_users, {r: np.sort(np.asarray(d, dtype=np.int32)) for r, d in role_docs.items()},
                parent or {}, m)


def tree_rbac(num_users=1000, num_roles=100, num_docs=10_000, h=4, b0=3, b1=4, seed=20251121, clustered=False):
    """clustered (development, not the reference's generator): every role's own documents are one run of consecutive ids."""
    rng = np.random.default_rng([int(seed), 11])
    remaining = list(range(1, num_roles + 1))
    children = {0: []}
    parent = {}
    order = []                                   # pre-order of non-root nodes

    def add_children(node, level):
        if level >= h or not remaining:
            return
        for _ in range(min(int(rng.integers(b0, b1 + 1)), len(remaining))):
            if not remaining:
                break
            child = remaining.pop(0)
            children[node].append(child)
            children[child] = []
            parent[child] = node
            order.append(child)
            add_children(child, level + 1)

    add_children(0, 0)
    n_sets = len(order)
    docs = rng.permutation(np.arange(1, num_docs + 1, dtype=np.int32))
    if clustered:
        docs = np.arange(1, num_docs + 1, dtype=np.int32)
    size = num_docs // n_sets
    own = {}
    for i, role in enumerate(order):
        own[role] = docs[i * size:] if i == n_sets - 1 else docs[i * size:(i + 1) * size]
    role_docs = {}
    for role in order:                           # pre-order: the parent is always done first
        p = parent[role]
        role_docs[role] = own[role] if p == 0 else np.concatenate([role_docs[p], own[role]])
    user_roles = []
    for role, users in zip(order, np.array_split(np.arange(1, num_users + 1), n_sets)):
        user_roles.extend((int(u), role) for u in users)
    return _finish(user_roles, role_docs, num_users, parent)


def random_rbac(num_users=1000, num_roles=100, num_docs=10_000, m_roles=3, m_perms=2000, seed=20251121):
    """services/rbac_generator/random_rbac_data_generator.py:38-82: 1..m_roles roles per user, each role
    m_perms/2..m_perms random documents, every document assigned at least once."""
    rng = np.random.default_rng([int(seed), 13])
    user_roles = []
    for u in range(1, num_users + 1):
        for r in rng.choice(np.arange(1, num_roles + 1), int(rng.integers(1, m_roles + 1)), replace=False):
            user_roles.append((u, int(r)))
    role_docs = {}
    seen = np.zeros(num_docs + 1, dtype=bool)
    for r in range(1, num_roles + 1):
        k = int(rng.integers(m_perms // 2, m_perms + 1))
        d = rng.choice(np.arange(1, num_docs + 1, dtype=np.int32), min(k, num_docs), replace=False)
        role_docs[r] = d
        seen[d] = True
    for d in np.flatnonzero(~seen[1:]) + 1:
        r = int(rng.integers(1, num_roles + 1))
        role_docs[r] = np.append(role_docs[r], np.int32(d))
    return _finish(user_roles, role_docs, num_users)


def sample_queries(n_queries, n_rows, num_users, seed=20251121):
    """(query row index, user id) pairs: uniform rows x uniform users."""
    rng = np.random.default_rng([int(seed), 17])
    return rng.integers(0, n_rows, n_queries), rng.integers(1, num_users + 1, n_queries)
