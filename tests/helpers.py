"""Shared checkers for the parity tests."""
import numpy as np


def sift_like(rng, n, d=128):
    """SIFT-like rows: integer-valued fp32 in [0,255] — fp32 sums are exact."""
    return np.clip(np.rint(np.abs(rng.normal(0, 45, (n, d)))), 0, 255).astype(np.float32)


def assert_valid_topk(ids, dist, ref_all, k, tol, candidates=None):
    """`ids`/`dist` is a valid top-k of the reference distance vector `ref_all` within `tol`:
    values right for the returned ids, sorted, and nothing better left out.
    Used for real-valued data where the reference itself is summation-order ambiguous
    (pgvector builds with -fassociative-math); integer-valued data is
    compared bit-exact instead."""
    ids = np.asarray(ids)
    dist = np.asarray(dist, dtype=np.float64)
    ref_all = np.asarray(ref_all, dtype=np.float64)
    cand = np.arange(ref_all.size) if candidates is None else np.asarray(candidates)
    want = min(k, cand.size)
    assert ids.size == want, (ids.size, want)
    if want == 0:
        return
    assert len(set(ids.tolist())) == ids.size, "duplicate ids"
    assert np.isin(ids, cand).all(), "returned a filtered-out row"
    r = ref_all[ids]
    scale = np.maximum(1.0, np.abs(r))
    ok = (np.abs(dist - r) <= tol * scale) | (np.isnan(dist) & np.isnan(r))
    assert ok.all(), (dist[~ok], r[~ok])
    rr = np.where(np.isnan(r), np.inf, r)
    assert (np.diff(rr) >= -tol * scale[1:]).all(), "not sorted"
    rest = np.setdiff1d(cand, ids)
    if rest.size:
        rest_d = np.where(np.isnan(ref_all[rest]), np.inf, ref_all[rest])
        worst = rr.max()
        if np.isfinite(worst):
            assert rest_d.min() >= worst - tol * max(1.0, abs(worst)), (rest_d.min(), worst)
