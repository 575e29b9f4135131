"""k above the 64-key wave list: scan -> one key per row -> radix SELECT (qv_select.hip).  The reference asks for such k
routinely: its negative-example branches fetch max(2k, 30) (pkg/hybrid/hybrid_index.go:516-522, pkg/hnsw/adapter.go:353-359)
and HybridIndex.BatchSearch takes any k (hybrid_index.go:677-811).  Bar as everywhere: rows in the oracle's order
((distance, row) ascending), distances bit for bit."""
import numpy as np
import pytest

from tests import _oracle as O

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _mk(dim, metric, rows=None, **kw):
    import quiver_amd as q
    idx = q.DeviceIndex(dim, metric, **kw)
    if rows is not None and len(rows):
        idx.add(rows)
    return idx


def _check(idx, metric, rows, qs, k, alive=None):
    r, d, c = idx.search(qs, k)
    for qi in range(len(qs)):
        er, ed = O.exact_search(metric, rows, qs[qi], k, alive=alive)
        assert int(c[qi]) == len(er)
        assert np.array_equal(r[qi, : len(er)], er), (metric, k, qi)
        assert np.array_equal(_bits(d[qi, : len(er)]), _bits(ed)), (metric, k, qi)
        assert np.all(r[qi, len(er):] == 0xFFFFFFFF) and np.all(np.isinf(d[qi, len(er):]))


@pytest.mark.parametrize("metric", range(9))
def test_select_every_metric(metric):
    rng = np.random.default_rng(40 + metric)
    rows = rng.standard_normal((30000, 24)).astype(np.float32)
    qs = rng.standard_normal((2, 24)).astype(np.float32)
    idx = _mk(24, metric, rows)
    for k in (65, 100, 1000):
        _check(idx, metric, rows, qs, k)


@pytest.mark.parametrize("k", [65, 66, 127, 128, 129, 257, 1000, 4095, 4096, 4097, 8192, 8193, 20000])
def test_select_k_sweep_unit_vectors(k):
    """unit vectors under cosine: every distance shares its exponent, so the first window decides little — the shape of the
    headline workload; 8193 and 20000 take the full ranking (radix sort) and must agree with the selection next to them"""
    rows = O.gen_rows(11, 0, 60000, 48)
    qs = O.gen_rows(12, 0, 2, 48)
    idx = _mk(48, "cosine", rows)
    _check(idx, 0, rows, qs, k)


@pytest.mark.parametrize("metric", [0, 1, 2, 4, 6])
def test_select_ties_beyond_the_sort_capacity(metric):
    """a handful of distinct distances over 40 000 rows: tens of thousands of keys tie on all 32 distance bits, far more than
    the selection's sort holds — the tie case takes the first k_rem tied rows in row order, as (distance, row) ascending asks"""
    rng = np.random.default_rng(70 + metric)
    rows = rng.integers(-1, 2, size=(40000, 3)).astype(np.float32)
    q = np.array([1, 0, -1], np.float32)
    idx = _mk(3, metric, rows)
    dead = np.nonzero(rng.random(40000) < 0.25)[0].astype(np.uint32)
    idx.remove(dead)
    alive = np.ones(40000, bool); alive[dead] = False
    for k in (65, 100, 3000, 8192):
        _check(idx, metric, rows, q[None, :], k, alive=alive)


def test_select_identical_rows_and_zero_vectors():
    rows = np.tile(np.array([[0.5, -1.0, 2.0, 0.25]], np.float32), (20000, 1))
    rows[1000:1500] = 0.0                                               # cosine: zero vectors sit at distance 1 (distances.go:25-27)
    q = np.array([0.5, -1.0, 2.0, 0.25], np.float32)
    for metric in (0, 1, 3):
        idx = _mk(4, metric, rows)
        for k in (100, 19600, 8192):
            _check(idx, metric, rows, q[None, :], k)


def test_select_nan_and_inf_rows():
    rng = np.random.default_rng(5)
    rows = rng.standard_normal((20000, 8)).astype(np.float32)
    rows[::7, 3] = np.nan
    rows[::11, 2] = np.inf
    q = rng.standard_normal(8).astype(np.float32)
    for metric in (0, 1, 2):
        idx = _mk(8, metric, rows)
        for k in (200, 8000):
            r, d, c = idx.search(q, k)
            er, ed = O.exact_search(metric, rows, q, k)
            assert np.array_equal(r[0], er) and np.array_equal(_bits(d[0]), _bits(ed)), (metric, k)


@pytest.mark.parametrize("nq", [2, 5, 9, 40])
def test_select_batches(nq):
    rows = O.gen_rows(21, 0, 50000, 64)
    qs = O.gen_rows(22, 0, nq, 64)
    for metric in (0, 3, 4):
        idx = _mk(64, metric, rows)
        for k in (100, 300):
            _check(idx, metric, rows, qs, k)


def test_select_masked_and_negative():
    rows = O.gen_rows(31, 0, 30000, 32)
    qs = O.gen_rows(32, 0, 2, 32)
    idx = _mk(32, "cosine", rows)
    mask = np.random.default_rng(3).random(30000) < 0.3
    r, d, c = idx.search_masked(qs, 500, mask)
    for qi in range(2):
        er, ed = O.exact_search(0, rows, qs[qi], 500, alive=mask)
        assert int(c[qi]) == 500 and np.array_equal(r[qi, :500], er) and np.array_equal(_bits(d[qi, :500]), _bits(ed))
    # hybrid_index.go:516-522: k = 50 with a negative example fetches 100
    import ctypes as C
    import quiver_amd as q
    neg = O.gen_rows(33, 0, 1, 32)[0]
    ro = np.empty(100, np.uint32); do = np.empty(100, np.float32); no = np.empty(100, np.float32); cnt = C.c_uint32(0)
    q._lib.check(q.lib().qv_index_search_negative(idx.handle, qs[0].ctypes.data, neg.ctypes.data, 100, ro.ctypes.data, do.ctypes.data, no.ctypes.data, C.byref(cnt)))
    er, ed = O.exact_search(0, rows, qs[0], 100)
    assert cnt.value == 100 and np.array_equal(ro, er) and np.array_equal(_bits(do), _bits(ed))
    assert np.array_equal(_bits(no), _bits(O.all_distances(0, rows[er], neg)))


def test_select_device_entry_point_pads_past_the_live_rows():
    """qv_index_search_device with k above the live size: min(k, live) results, the rest of each k-wide list padded"""
    import torch
    rows = O.gen_rows(41, 0, 300, 16)
    idx = _mk(16, "cosine", rows)
    idx.remove(np.arange(0, 300, 2, dtype=np.uint32))                  # 150 live
    alive = np.ones(300, bool); alive[::2] = False
    qs = O.gen_rows(42, 0, 3, 16)
    for k in (100, 200, 70):
        dq = torch.from_numpy(qs).cuda()
        dr = torch.empty((3, k), dtype=torch.int32, device="cuda"); dd = torch.empty((3, k), dtype=torch.float32, device="cuda")
        idx.search_device(dq.data_ptr(), 3, k, dr.data_ptr(), dd.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        r = dr.cpu().numpy().view(np.uint32); d = dd.cpu().numpy()
        for qi in range(3):
            er, ed = O.exact_search(0, rows, qs[qi], k, alive=alive)
            n = len(er)
            assert n == min(k, 150)
            assert np.array_equal(r[qi, :n], er) and np.array_equal(_bits(d[qi, :n]), _bits(ed))
            assert np.all(r[qi, n:] == 0xFFFFFFFF) and np.all(np.isinf(d[qi, n:]))


def test_select_one_million_rows_equals_full_ranking_prefix():
    """BASELINE configs[1]'s corpus: the selection's result is the first k of the full ranking (itself oracle-tested)"""
    import quiver_amd as q
    idx = q.DeviceIndex(768, "cosine"); idx.reserve(1_000_000); idx.add_synthetic(20260424, 0, 1_000_000)
    qs = O.gen_rows(20260425, 0, 2, 768)
    fr, fd, _ = idx.search(qs, 9000)                                   # > kMaxSelectK: the radix sort
    for k in (100, 1000, 4096, 8192):
        r, d, c = idx.search(qs, k)
        assert np.array_equal(r, fr[:, :k]) and np.array_equal(_bits(d), _bits(fd[:, :k]))
    corpus = O.gen_rows(20260424, 0, 1_000_000, 768)
    er, ed = O.exact_search(0, corpus, qs[0], 100)
    r, d, c = idx.search(qs[:1], 100)
    assert np.array_equal(r[0], er) and np.array_equal(_bits(d[0]), _bits(ed))
