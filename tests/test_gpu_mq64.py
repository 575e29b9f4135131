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


@pytest.fixture(autouse=True)
def exact_scans_only(monkeypatch):
    """qv_index_search hands batches of >= 9 queries over >= 8M query-rows to the filter + re-score path; these tests are about
    the exact multi-query scan, so every index made here chooses "no filter" (qv_index_set_filter(QV_FILTER_OFF))"""
    monkeypatch.setattr(quiver_amd.DeviceIndex, "default_filter", "off")


def _check(idx, rows, alive, metric, qs, k):
    mid = quiver_amd.metric_id(metric)
    r, d, c = idx.search(qs, k, batched=False)              # qv_index_search: exact paths only
    for i in range(qs.shape[0]):
        ro, do = O.exact_search(mid, rows, qs[i], k, alive=alive)
        assert c[i] == ro.size
        assert r[i, :ro.size].tolist() == ro.tolist(), (metric, i)
        assert d[i, :ro.size].tobytes() == do.tobytes(), (metric, i)


@pytest.mark.parametrize("metric,dim,nq,k", [
    ("cosine", 32, 20, 10),        # one 32-query pass, 12 query slots replicated
    ("dot", 32, 16, 64),           # k = 64: full-width lists; the 16-query shape
    ("cosine", 100, 9, 7),         # dim4 = 25: three register blocks + one tail chunk, 7 query slots replicated
    ("cosine", 12, 31, 3),         # dim4 = 3: tail chunks only
    ("cosine", 32, 40, 10),        # 32 on the matrix kernel + a last pass of 8 on k_flat_scan_mq (split in launch_flat_topk)
    ("dot", 64, 70, 5),            # two 32-query passes + 6 split off
    ("cosine", 32, 45, 10),        # two 32-query passes, the second with 13 queries
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
            "idx = quiver_amd.DeviceIndex(16, 'cosine', filter='off'); idx.add_synthetic(1, 0, 530000)\n"
            "q = np.random.default_rng(0).standard_normal((12, 16)).astype(np.float32)\n"
            "idx.search(q, 10); idx.search(q[:4], 10)\n")
    env = dict(os.environ, QV_TRACE="1")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert p.returncode == 0, p.stderr
    assert "k_flat_scan_mq64 (nq=12" in p.stderr, p.stderr
    assert "k_mq64_bounded held every candidate" in p.stderr, p.stderr
    assert "k_flat_scan_mq QB=4 (nq=4" in p.stderr, p.stderr


def test_a_bound_that_admits_everything_takes_the_fallback():
    """the companion of the two fallback parity tests below: with every sampled tile deleted the bounded pass must report
    an overflow (QV_TRACE=1 reads the flag back)"""
    import os, subprocess, sys
    code = ("import numpy as np, quiver_amd\n"
            "n = 530000\n"
            "idx = quiver_amd.DeviceIndex(16, 'dot', filter='off'); idx.add_synthetic(1, 0, n)\n"
            "nt = (n + 63) // 64; ns = min(256, nt // 8) // 8 * 8; step = nt // ns\n"
            "dead = np.concatenate([np.arange(t * step * 64, (t * step + 1) * 64, dtype=np.uint32) for t in range(ns)])\n"
            "idx.remove(dead)\n"
            "q = np.random.default_rng(0).standard_normal((12, 16)).astype(np.float32)\n"
            "idx.search(q, 10)\n")
    env = dict(os.environ, QV_TRACE="1")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert p.returncode == 0, p.stderr
    assert "k_mq64_bounded overflowed" in p.stderr, p.stderr


def _sample_tiles(n):
    """the tiles the sample pass of launch_flat_scan_mq64 reads on a 256-CU device: 256 tiles, evenly spaced"""
    n_tiles = (n + 63) // 64
    n_s = min(256, n_tiles // 8) // 8 * 8
    step = n_tiles // n_s
    return np.arange(n_s) * step


def test_mq64_bound_that_admits_everything_falls_back_to_the_list_kernel():
    """The sample pass's k-th key bounds the full pass.  Here every sampled tile holds far rows and every other row is close
    to the queries, so the bound admits ~all 530k rows, the candidate buffers overflow and the register-list kernel must redo
    the pass: results still equal the oracle's."""
    n, dim, nq, k = 530_000, 32, 20, 10
    rng = np.random.default_rng(77)
    centre = rng.standard_normal(dim).astype(np.float32)
    rows = (centre[None, :] + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)
    far = np.zeros(n, bool)
    for t in _sample_tiles(n):
        far[t * 64:(t + 1) * 64] = True
    rows[far] = -rows[far]                                       # antipodal: the worst cosine distances in the corpus
    idx = quiver_amd.DeviceIndex(dim, "cosine")
    idx.add(rows)
    qs = (centre[None, :] + 0.05 * rng.standard_normal((nq, dim))).astype(np.float32)
    _check(idx, rows, None, "cosine", qs, k)


def test_mq64_sample_with_fewer_than_k_live_rows():
    """all but 3 rows of the sampled tiles are deleted: no bound can be derived (fewer than k live keys), every row is a
    candidate, the fallback answers"""
    n, dim, nq, k = 530_000, 16, 12, 10
    rows = O.gen_rows(9011, 0, n, dim)
    idx = quiver_amd.DeviceIndex(dim, "dot")
    idx.add(rows)
    st = _sample_tiles(n)
    dead = np.concatenate([np.arange(t * 64, min(n, (t + 1) * 64), dtype=np.uint32) for t in st])
    dead = np.setdiff1d(dead, np.array([st[0] * 64 + 1, st[5] * 64 + 7, st[200] * 64 + 63], dtype=np.uint32))
    idx.remove(dead)
    alive = np.ones(n, np.uint8); alive[dead] = 0
    _check(idx, rows, alive, "dot", O.gen_rows(9012, 0, nq, dim), k)
