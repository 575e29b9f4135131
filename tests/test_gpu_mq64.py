"""Exact multi-query scan on the f64 matrix cores (quiver_amd/csrc/qv_mq64.hip) against the CPU oracle.

v_mfma_f64_16x16x4_f64 is a sequential chain of rounded FMAs over its 4 k-steps, so chaining one per 16-byte
chunk reproduces the reference's float64 loop (pkg/vectortypes/distances.go:17-22, :82-86) bit for bit.  The
library takes this path for >= 9 cosine/dot queries over long scans (>= 16 tiles per workgroup slot), hence
the row counts here.  Everything is compared through the C ABI: rows, order, float32 bits."""
import numpy as np
import pytest

import quiver_amd
from tests import _oracle as O

pytestmark = pytest.mark.gpu


def _check(idx, rows, alive, metric, qs, k):
    mid = quiver_amd.metric_id(metric)
    r, d, c = idx.search(qs, k, batched=False)              # qv_index_search: exact paths only
    for i in range(qs.shape[0]):
        ro, do = O.exact_search(mid, rows, qs[i], k, alive=alive)
        assert c[i] == ro.size
        assert r[i, :ro.size].tolist() == ro.tolist(), (metric, i)
        assert d[i, :ro.size].tobytes() == do.tobytes(), (metric, i)


@pytest.mark.parametrize("metric,dim,nq,k", [
    ("cosine", 32, 20, 10),        # two passes (16 + 4 queries)
    ("dot", 32, 16, 64),           # k = 64: full-width lists
    ("cosine", 100, 9, 7),         # dim4 = 25: three register blocks + one tail chunk, 7 query slots replicated
    ("cosine", 12, 31, 3),         # dim4 = 3: tail chunks only; two passes (16 + 15)
])
def test_mq64_equals_oracle(metric, dim, nq, k):
    n = 530_000                                              # 8282 tiles >= 16 per workgroup slot on 256 CUs
    rows = O.gen_rows(9001, 0, n, dim)
    if metric == "dot":
        rows *= np.linspace(0.5, 2.0, n, dtype=np.float32)[:, None]   # un-normalised rows
    rows[1000:1010] = rows[2000:2010]                        # equal distances: (distance, row) order
    rows[5] = 0.0                                            # zero vector (cosine: distance 1)
    idx = quiver_amd.DeviceIndex(dim, metric)
    idx.add(rows)
    dead = np.array([3, 64, 65, 1003, 2005, n - 1], dtype=np.uint32)
    idx.remove(dead)
    alive = np.ones(n, np.uint8); alive[dead] = 0
    qs = O.gen_rows(9002, 0, nq, dim)
    qs[1] = rows[2001]                                       # a query equal to a (duplicated) row
    if nq > 2:
        qs[2] = 0.0                                          # zero query
    _check(idx, rows, alive, metric, qs, k)


def test_mq64_ragged_last_tile_and_all_but_few_dead():
    n, dim = 524_288 + 37, 16                                # partial last tile
    rows = O.gen_rows(9003, 0, n, dim)
    idx = quiver_amd.DeviceIndex(dim, "cosine")
    idx.add(rows)
    keep = np.array([7, 70_000, 524_300, n - 1])
    dead = np.setdiff1d(np.arange(0, n, 3, dtype=np.uint32), keep.astype(np.uint32))
    idx.remove(dead)
    alive = np.ones(n, np.uint8); alive[dead] = 0
    _check(idx, rows, alive, "cosine", O.gen_rows(9004, 0, 12, dim), 10)


def test_long_cosine_scans_with_9_or_more_queries_run_on_the_f64_matrix_kernel():
    """the parity cases above must really be the MFMA kernel: QV_TRACE=1 names the scan kernel on stderr"""
    import os, subprocess, sys
    code = ("import numpy as np, quiver_amd\n"
            "idx = quiver_amd.DeviceIndex(16, 'cosine'); idx.add_synthetic(1, 0, 530000)\n"
            "q = np.random.default_rng(0).standard_normal((12, 16)).astype(np.float32)\n"
            "idx.search(q, 10); idx.search(q[:4], 10)\n")
    env = dict(os.environ, QV_TRACE="1")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert p.returncode == 0, p.stderr
    assert "k_flat_scan_mq64 (nq=12" in p.stderr, p.stderr
    assert "k_flat_scan_mq QB=4 (nq=4" in p.stderr, p.stderr
