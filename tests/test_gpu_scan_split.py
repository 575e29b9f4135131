"""The short-corpus form of the exact flat scan (k_flat_scan_split, qv_scan.hip): the eight waves of a workgroup share a tile, every
row's float64 chain becomes eight partial chains, and the float32 is taken from the certificate (or, when the interval does not decide
it, from the reference's single chain).  One query per call on collections in its range (>= 256 dimensions, 128 .. 160 k rows) over
vectors that make the certificate fail and the order matter: ties, duplicates, zero vectors, queries that ARE corpus rows, wild dynamic
range, tombstones.  Rows and float32 bits against the CPU oracle (exact.go:56-133 with the (distance, row) order)."""
import numpy as np
import pytest

import quiver_amd
from tests import _oracle as O

pytestmark = pytest.mark.gpu

METRICS = ["cosine", "l2", "dot", "l1", "l2sq_f64"]


def _vectors(rng, n, dim, style):
    if style == 0:
        x = rng.choice(np.array([-2.0, -1.0, -0.5, 0.0, 0.0, 0.5, 1.0, 3.0], np.float32), size=(n, dim))
    elif style == 1:
        x = (rng.standard_normal((n, dim)) * np.exp2(rng.integers(-20, 20, size=(n, dim)))).astype(np.float32)
    elif style == 2:
        c = rng.standard_normal((max(n // 50, 2), dim)).astype(np.float32)
        x = c[rng.integers(0, c.shape[0], n)] + (rng.standard_normal((n, dim)) * 1e-4).astype(np.float32)
        x[rng.integers(0, n, n // 8)] = x[rng.integers(0, n, n // 8)]
        x[rng.integers(0, n, 5)] = 0.0
    else:
        x = rng.standard_normal((n, dim)).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32)


@pytest.mark.parametrize("seed", range(30))
def test_split_scan_equals_the_oracle(seed):
    rng = np.random.default_rng(4400 + seed)
    metric = METRICS[seed % 5]
    dim = int(rng.choice([64, 100, 128, 256, 258, 300, 384, 768, 1000, 1536]))   # (below 128 dimensions only the shared-pass form applies)
    n = int(rng.choice([130, 700, 3000, 9000, 20000]))
    style = int(rng.integers(0, 4))
    rows = _vectors(rng, n, dim, style)
    idx = quiver_amd.DeviceIndex(dim, metric)
    idx.add(rows)
    alive = np.ones(n, np.uint8)
    if seed % 3 == 0:
        dead = rng.choice(n, n // 7, replace=False).astype(np.uint32)
        idx.remove(dead); alive[dead] = 0
    mid = quiver_amd.metric_id(metric)
    qs = np.concatenate([_vectors(rng, 5, dim, style), rows[rng.integers(0, n, 5)], np.zeros((1, dim), np.float32)])
    for k in (1, 10, 64):
        want = [O.exact_search(mid, rows, qs[i], k, alive=alive) for i in range(qs.shape[0])]
        for i in range(qs.shape[0]):
            r, d, c = idx.search(qs[i:i + 1], k)                       # one query per call: k_flat_scan_split
            ro, do = want[i]
            assert int(c[0]) == ro.size, (metric, dim, n, style, k, i)
            assert r[0, :ro.size].tolist() == ro.tolist(), (metric, dim, n, style, k, i)
            assert d[0, :ro.size].tobytes() == do.tobytes(), (metric, dim, n, style, k, i)
        for lo, hi in ((0, 2), (1, 4), (0, 5), (2, 10), (0, 11)):    # the queries of a shared pass: k_flat_scan_split_mq, groups of 4 / 8
            r, d, c = idx.search(qs[lo:hi], k)
            for i in range(lo, hi):
                ro, do = want[i]
                assert int(c[i - lo]) == ro.size, (metric, dim, n, style, k, lo, hi, i)
                assert r[i - lo, :ro.size].tolist() == ro.tolist(), (metric, dim, n, style, k, lo, hi, i)
                assert d[i - lo, :ro.size].tobytes() == do.tobytes(), (metric, dim, n, style, k, lo, hi, i)
    idx.close()


def test_split_scan_at_the_upper_end_of_its_range():
    """160 k rows x 256 (2500 tiles: ten per workgroup, consumers rotate over the eight waves)"""
    n, dim = 160_000, 256
    idx = quiver_amd.DeviceIndex(dim, "cosine")
    idx.add_synthetic(991, 0, n)
    rows = O.gen_rows(991, 0, n, dim)
    qs = np.concatenate([O.gen_rows(992, 0, 3, dim), rows[[0, n - 1, 77777]]])
    for i in range(qs.shape[0]):
        r, d, c = idx.search(qs[i:i + 1], 10)
        ro, do = O.exact_search(0, rows, qs[i], 10)
        assert r[0].tolist() == ro.tolist() and d[0].tobytes() == do.tobytes(), i
    idx.close()
