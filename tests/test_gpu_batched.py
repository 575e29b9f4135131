"""Batched-query path (BASELINE configs[2]): fp32-MFMA filter + exact re-scoring must return
exactly what the exact scan returns — same rows, same order, same float32 bits — including
when the sample bound is loose (candidate overflow -> exact redo), with ties, tombstones and
zero vectors; and the multi-query exact scan must equal per-query scans."""
import numpy as np
import pytest

from tests import _oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["bf16x3", "fp32", "bf16x1"], autouse=True)
def filter_kernel(request, monkeypatch):
    """every case runs with the three filter kernels: bfloat16 x 3, the fp32 MFMA chain (BASELINE configs[2] as written) and
    the one-term bfloat16 filter; each index chooses through qv_index_set_filter (DeviceIndex.default_filter here)"""
    import quiver_amd
    monkeypatch.setattr(quiver_amd.DeviceIndex, "default_filter", request.param)
    return request.param


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _exact(idx, qs, k):
    """the exact multi-query scan, forced: qv_index_search hands batches of >= 9 queries to the
    filter + re-score path by itself, so the reference result is gathered 8 queries at a time"""
    outs = [idx.search(qs[i: i + 8], k) for i in range(0, len(qs), 8)]
    return tuple(np.concatenate([o[j] for o in outs]) for j in range(3))


def _eq(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(_bits(a[1]), _bits(b[1])) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("metric,nq", [("cosine", 256), ("dot_product", 256), ("euclidean", 256), ("squared_euclidean", 256),
                                       ("cosine", 100),      # pads to 128 queries: every wave fetches its own rows
                                       ("dot_product", 600), # pads to 768: three workgroups share a row group's walk
                                       ("euclidean", 40),    # pads to 64: one query block
                                       ("cosine", 300),      # 256 + a tail of 44 in a filter call of its own
                                       ("dot_product", 260)])  # 256 + a tail of 4 on the exact scan
def test_mfma_batched_equals_exact_scan(metric, nq):
    import quiver_amd as q
    n, dim = 300_000, 768
    idx = q.DeviceIndex(dim, metric)
    idx.add_synthetic(20260424, 0, n)
    qs = O.gen_rows(20260425, 0, nq, dim)
    exact = _exact(idx, qs, 10)
    batched = idx.search(qs, 10, batched=True)
    assert _eq(exact, batched)
    assert _eq(exact, idx.search(qs, 10))            # qv_index_search dispatches big batches to the same path
    # and against the CPU oracle for a few queries (bit-exact)
    corpus = O.gen_rows(20260424, 0, n, dim)
    for i in (0, 17 % nq, nq - 1):
        er, ed = O.exact_search(q.metric_id(metric), corpus, qs[i], 10)
        assert np.array_equal(batched[0][i], er) and np.array_equal(_bits(batched[1][i]), _bits(ed))
    # other k, fewer queries (not a multiple of 64)
    for k, m in ((1, 100), (64, 40), (33, 64), (10, 160), (7, 200)):      # 160 and 200 queries: padded from 3-4 to 4 blocks of 64
        if m > nq:
            continue
        assert _eq(_exact(idx, qs[:m], k), idx.search(qs[:m], k, batched=True))


@pytest.mark.parametrize("nq", [300, 260])
def test_batched_device_entry_splits_a_small_tail(nq):
    """qv_index_search_batched_device with 257-320 queries: the tail runs as a call of its own on the same stream"""
    import torch
    import quiver_amd as q
    n, dim, k = 300_000, 768, 10
    idx = q.DeviceIndex(dim, "cosine")
    idx.add_synthetic(20260424, 0, n)
    qs = O.gen_rows(20260425, 0, nq, dim)
    dq = torch.from_numpy(qs).cuda()
    dr = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    fl = torch.full((nq,), 7, dtype=torch.int32, device="cuda")
    idx.search_batched_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), fl.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(fl.abs().sum().item()) == 0
    exact = _exact(idx, qs, k)
    assert np.array_equal(dr.cpu().numpy().view(np.uint32), exact[0])
    assert np.array_equal(_bits(dd.cpu().numpy()), _bits(exact[1]))


def test_config2_dot_product_256x1Mx768_against_the_cpu_oracle():
    """BASELINE.json configs[2] as written — 256 queries x 1M x 768, dot-product, fp32 MFMA + fused candidate selection — at
    full size: three of the 256 queries against the CPU oracle over all 1M rows (rows and float32 bits), all 256 against the
    exact scan"""
    import os
    import quiver_amd as q
    from tests._par import exact_topk_synthetic
    n, dim, nq = 1_000_000, 768, 256
    idx = q.DeviceIndex(dim, "dot_product")
    idx.reserve(n)
    idx.add_synthetic(20260424, 0, n)
    qs = O.gen_rows(20260425, 0, nq, dim)
    batched = idx.search(qs, 10, batched=True)
    assert _eq(_exact(idx, qs, 10), batched)
    for i in (0, 101, 255):
        er, ed = exact_topk_synthetic(3, 20260424, n, dim, qs[i], 10, chunk=100_000, workers=min(16, os.cpu_count() or 8))
        assert np.array_equal(batched[0][i], er) and np.array_equal(_bits(batched[1][i]), _bits(ed)), i


def test_mfma_batched_with_unrepresentative_sample_ties_and_tombstones():
    """the first 32K rows (the sample) are all far from the queries, so the sample bound is
    loose and candidate buffers overflow -> those queries are redone by the exact scan;
    duplicated rows create exact ties that must come out in row order; dead rows never appear"""
    import quiver_amd as q
    dim, nq = 64, 64
    rng = np.random.default_rng(5)
    qs = rng.standard_normal((nq, dim)).astype(np.float32)
    far = -np.abs(rng.standard_normal((40_000, dim))).astype(np.float32) * np.sign(qs[0])      # anti-correlated with q0
    near = (qs[rng.integers(0, nq, 230_000)] + 0.3 * rng.standard_normal((230_000, dim))).astype(np.float32)
    rows = np.concatenate([far, near])
    rows[100_000] = rows[100_001] = rows[150_000]          # exact ties
    rows[200_000] = 0.0                                    # zero vector: cosine distance 1 by definition
    idx = q.DeviceIndex(dim, "cosine")
    idx.add(rows)
    dead = rng.choice(len(rows), 5000, replace=False).astype(np.uint32)
    idx.remove(dead)
    exact = _exact(idx, qs, 10)
    batched = idx.search(qs, 10, batched=True)
    assert _eq(exact, batched)
    assert not np.isin(batched[0], dead).any()
    alive = np.ones(len(rows), bool); alive[dead] = False
    for i in (0, 1, 63):
        er, ed = O.exact_search(0, rows, qs[i], 10, alive=alive)
        assert np.array_equal(batched[0][i], er) and np.array_equal(_bits(batched[1][i]), _bits(ed))


@pytest.mark.parametrize("metric", ["euclidean", "squared_euclidean", "dot_product"])
def test_mfma_batched_unnormalised_rows_and_queries(metric):
    """rows and queries of very different lengths: the error margins scale with |q||r|"""
    import quiver_amd as q
    rng = np.random.default_rng(11)
    n, dim, nq = 270_000, 96, 64
    rows = (rng.standard_normal((n, dim)) * rng.choice([0.01, 1.0, 30.0], size=(n, 1))).astype(np.float32)
    qs = (rng.standard_normal((nq, dim)) * rng.choice([0.1, 1.0, 10.0], size=(nq, 1))).astype(np.float32)
    rows[123] = qs[0]                                       # an exact hit: distance 0 (or 1 - |q|^2 for dot)
    idx = q.DeviceIndex(dim, metric)
    idx.add(rows)
    exact = _exact(idx, qs, 10)
    batched = idx.search(qs, 10, batched=True)
    assert _eq(exact, batched)
    for i in (0, 5, 63):
        er, ed = O.exact_search(q.metric_id(metric), rows, qs[i], 10)
        assert np.array_equal(batched[0][i], er) and np.array_equal(_bits(batched[1][i]), _bits(ed))


def test_batched_falls_back_when_not_applicable():
    import quiver_amd as q
    rows = O.gen_rows(3, 0, 5000, 32)
    qs = O.gen_rows(4, 0, 70, 32)
    for metric in ("cosine", "euclidean", "manhattan", "hnsw_cosine"):
        idx = q.DeviceIndex(32, metric)
        idx.add(rows)
        assert _eq(idx.search(qs, 5), idx.search(qs, 5, batched=True))          # small corpus / non-GEMM metric -> exact scan
        assert _eq(idx.search(qs, 200), idx.search(qs, 200, batched=True))      # k > 64 -> full-ranking path


@pytest.mark.parametrize("metric", range(9))
def test_multi_query_scan_equals_single_query_scans(metric):
    import quiver_amd as q
    rng = np.random.default_rng(metric)
    rows = rng.standard_normal((20_000, 100)).astype(np.float32)
    rows[7] = rows[8]
    idx = q.DeviceIndex(100, metric)
    idx.add(rows)
    idx.remove([3, 4, 5])
    for nq in (2, 5, 9, 16, 37):
        qs = rng.standard_normal((nq, 100)).astype(np.float32)
        many = idx.search(qs, 10)
        for i in range(nq):
            one = idx.search(qs[i], 10)
            assert np.array_equal(many[0][i], one[0][0]) and np.array_equal(_bits(many[1][i]), _bits(one[1][0]))
    alive = np.ones(20_000, bool); alive[[3, 4, 5]] = False
    er, ed = O.exact_search(metric, rows, qs[0], 10, alive=alive)
    assert np.array_equal(many[0][0], er) and np.array_equal(_bits(many[1][0]), _bits(ed))


def test_batched_device_pointer_entry_point():
    import torch
    import quiver_amd as q
    idx = q.DeviceIndex(128, "cosine")
    idx.add_synthetic(5, 0, 300_000)
    qs = O.gen_rows(6, 0, 96, 128)
    dq = torch.from_numpy(qs).cuda()
    dr = torch.empty((96, 10), dtype=torch.int32, device="cuda")
    dd = torch.empty((96, 10), dtype=torch.float32, device="cuda")
    fl = torch.ones((96,), dtype=torch.int32, device="cuda")
    idx.search_batched_device(dq.data_ptr(), 96, 10, dr.data_ptr(), dd.data_ptr(), fl.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(fl.sum().item()) == 0
    ex = _exact(idx, qs, 10)
    assert np.array_equal(dr.cpu().numpy().view(np.uint32), ex[0]) and np.array_equal(_bits(dd.cpu().numpy()), _bits(ex[1]))
    # k = 64: the bound sample grows with k (and N) so that the candidate buffers do not overflow into the exact redo
    dr64 = torch.empty((96, 64), dtype=torch.int32, device="cuda"); dd64 = torch.empty((96, 64), dtype=torch.float32, device="cuda")
    fl.fill_(1)
    idx.search_batched_device(dq.data_ptr(), 96, 64, dr64.data_ptr(), dd64.data_ptr(), fl.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(fl.sum().item()) == 0
    ex64 = _exact(idx, qs, 64)
    assert np.array_equal(dr64.cpu().numpy().view(np.uint32), ex64[0]) and np.array_equal(_bits(dd64.cpu().numpy()), _bits(ex64[1]))
    small = q.DeviceIndex(128, "manhattan")
    small.add_synthetic(5, 0, 1000)
    with pytest.raises(q.QvError) as e:
        small.search_batched_device(dq.data_ptr(), 96, 10, dr.data_ptr(), dd.data_ptr(), fl.data_ptr(), 0)
    assert e.value.code == -8


@pytest.mark.parametrize("metric,dim", [("cosine", 128), ("dot_product", 96), ("euclidean", 200), ("squared_euclidean", 256), ("cosine", 52)])
def test_mfma_batched_wide_dynamic_range(metric, dim):
    """element magnitudes spread over 2^-12 .. 2^12: the bfloat16 split keeps float32's exponent range and its margin is relative to
    |q||r|, so the filter must not lose a neighbour; dimensions exercise the eight-step rounds (128, 256), the plain loop (96) and
    chunk counts that are not a multiple of 4 (200, 52)"""
    import quiver_amd as q
    rng = np.random.default_rng(dim)
    n, nq = 150_000, 64                                    # 9.6 M query-rows: above the filter's 8 M crossover
    rows = (rng.standard_normal((n, dim)) * np.exp2(rng.integers(-12, 13, size=(n, dim)))).astype(np.float32)
    qs = (rng.standard_normal((nq, dim)) * np.exp2(rng.integers(-12, 13, size=(nq, dim)))).astype(np.float32)
    qs[0] = rows[777]; rows[5] = 0.0
    idx = q.DeviceIndex(dim, metric)
    idx.add(rows)
    exact = _exact(idx, qs, 10)
    assert _eq(exact, idx.search(qs, 10, batched=True))
    for i in (0, 31):
        er, ed = O.exact_search(q.metric_id(metric), rows, qs[i], 10)
        assert np.array_equal(exact[0][i], er) and np.array_equal(_bits(exact[1][i]), _bits(ed))


def test_device_entry_slices_very_large_batches():
    """more than 8192 queries through qv_index_search_batched_device go in slices on the caller's stream (bounded workspace)"""
    import torch
    import quiver_amd as q
    n, dim, nq, k = 120_000, 64, 9000, 5
    idx = q.DeviceIndex(dim, "dot_product")
    idx.add_synthetic(20260424, 0, n)
    qs = O.gen_rows(20260425, 0, nq, dim)
    dq = torch.from_numpy(qs).cuda()
    dr = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    er = torch.empty((nq, k), dtype=torch.int32, device="cuda"); ed = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    fl = torch.full((nq,), 7, dtype=torch.int32, device="cuda")
    sp = torch.cuda.current_stream().cuda_stream
    idx.search_batched_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), fl.data_ptr(), sp)
    idx.search_device(dq.data_ptr(), nq, k, er.data_ptr(), ed.data_ptr(), sp)
    torch.cuda.synchronize()
    ok = (fl == 0)
    assert int((~ok).sum().item()) <= 8
    assert torch.equal(dr[ok], er[ok]) and dd[ok].cpu().numpy().tobytes() == ed[ok].cpu().numpy().tobytes()


def test_corpus_stored_cluster_by_cluster_keeps_the_filter_path(filter_kernel):
    """the bound sample is spread over the corpus: with rows stored cluster after cluster (how data often arrives) the first rows alone
    would bound nothing for queries of the later clusters, every candidate buffer would overflow and every query fall back to the
    exact scan.  No query may be handed back, and the results equal the exact scan."""
    import torch
    import quiver_amd as q
    rng = np.random.default_rng(11)
    dim, n_clusters, per = 128, 30, 10_000
    centres = rng.standard_normal((n_clusters, dim))
    rows = np.concatenate([c + 0.3 * rng.standard_normal((per, dim)) for c in centres]).astype(np.float32)
    nq, k = 64, 10
    qs = (centres[rng.integers(0, n_clusters, nq)] + 0.3 * rng.standard_normal((nq, dim))).astype(np.float32)
    idx = q.DeviceIndex(dim, "cosine")
    idx.add(rows)
    dq = torch.from_numpy(qs).cuda()
    dr = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    fl = torch.full((nq,), 7, dtype=torch.int32, device="cuda")
    idx.search_batched_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), fl.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    exact = _exact(idx, qs, k)
    assert _eq(exact, idx.search(qs, k, batched=True))               # through the host entry point (which redoes what is handed back)
    if filter_kernel != "fp32":                                      # (the fp32 chain keeps round 1's bound: an exact scan of the first rows)
        assert int(fl.abs().sum().item()) == 0
        assert np.array_equal(dr.cpu().numpy().view(np.uint32), exact[0]) and np.array_equal(_bits(dd.cpu().numpy()), _bits(exact[1]))


def test_rows_near_flt_max_are_not_lost_by_the_filter():
    """rows whose elements sit near FLT_MAX overflow the filter's float32 sums: +inf for same-sign rows, inf - inf = NaN for mixed signs.
    Both must reach the exact re-score (which accumulates in float64): a query pointing along such a row finds it at distance ~0."""
    import quiver_amd as q
    rng = np.random.default_rng(7)
    n, dim, nq = 150_000, 128, 64
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    huge_mixed = np.arange(1000, 1050); huge_pos = np.arange(2000, 2010)
    rows[huge_mixed] = (3.0e38 * rng.choice([-1.0, 1.0], size=(50, dim)) * rng.uniform(0.5, 1.0, size=(50, dim))).astype(np.float32)
    rows[huge_pos] = (3.0e38 * rng.uniform(0.5, 1.0, size=(10, dim))).astype(np.float32)
    qs = rng.standard_normal((nq, dim)).astype(np.float32)
    qs[0] = 1.0                                                     # along the all-positive rows
    qs[1] = (rows[1007].astype(np.float64) / 1.0e38).astype(np.float32)   # along a mixed-sign huge row
    idx = q.DeviceIndex(dim, "cosine")
    idx.add(rows)
    exact = _exact(idx, qs, 10)
    assert exact[0][1][0] == 1007 and exact[0][0][0] in huge_pos
    assert _eq(exact, idx.search(qs, 10, batched=True))


@pytest.mark.parametrize("metric,dim", [("cosine", 768), ("dot_product", 256), ("euclidean", 128), ("squared_euclidean", 384), ("cosine", 96),
                                        ("cosine", 256), ("euclidean", 1024)])      # (k_bf16rows_filter for every metric: a multiple of 128 dimensions outside the query-resident kernel's three)
def test_bf16_row_plane_gives_the_same_results_and_follows_every_mutation(metric, dim):
    """QV_FLAG_BF16_ROWS: the one-term filter reads the index's bfloat16 copy of the rows (dimensions that are a multiple of 128;
    others keep converting the float32 rows).  The plane is refreshed by add (host / device / synthetic, across a regrow) and
    update; results stay identical to the exact scan."""
    import torch
    import quiver_amd as q
    rng = np.random.default_rng(dim)
    idx = q.DeviceIndex(dim, metric, bf16_rows=True)
    idx.add_synthetic(20260424, 0, 150_000)                                  # synthetic block
    host = rng.standard_normal((50_000, dim)).astype(np.float32)
    idx.add(host)                                                            # host rows: grows the arrays
    dev = torch.from_numpy(rng.standard_normal((30_037, dim)).astype(np.float32)).cuda()
    idx.add_device(dev.data_ptr(), dev.shape[0], torch.cuda.current_stream().cuda_stream)   # device rows, ragged last tile
    torch.cuda.synchronize()
    qs = rng.standard_normal((256, dim)).astype(np.float32)                # 256 queries: four query blocks share the rows (the kernel that reads the plane)
    qs[:8] = host[100:108] + 0.01 * rng.standard_normal((8, dim)).astype(np.float32)       # near the rows that get replaced below
    assert _eq(_exact(idx, qs, 10), idx.search(qs, 10, batched=True))
    for j in range(100, 108):                                                # replace rows the first queries are close to
        idx.update(150_000 + j, -host[j])
    idx.remove(np.arange(150_000 + 200, 150_000 + 260, dtype=np.uint32))
    idx.update(150_000 + 210, qs[9])                                         # revive a removed row as an exact match of query 9
    e = _exact(idx, qs, 10)
    assert _eq(e, idx.search(qs, 10, batched=True))
    assert e[0][9][0] == 150_000 + 210
    for m, kk in ((44, 7), (9, 10), (64, 1)):          # one query block: k_bf16rows_filter_q64 (queries resident in LDS) where the dimension allows
        assert _eq(_exact(idx, qs[:m], kk), idx.search(qs[:m], kk, batched=True))


@pytest.mark.parametrize("metric", ["cosine", "dot_product", "euclidean", "squared_euclidean"])
def test_rows_and_queries_at_the_worst_of_bfloat16_rounding(metric):
    """The one-term filter's margin is |q - qh||r| + |qh||r - rh| per query and row (IndexView::rres, k_row_residual), not a worst
    case over operands — so the corpus here IS the worst case: every element sits exactly half way between two bfloat16 values
    (the largest rounding loss there is, all of the same sign for a row), other rows are exact in bfloat16 (no loss at all: the
    tightest margin), others have elements below 2^-126 that the matrix core may flush, and rows are updated after ingest (the
    residual follows).  Queries likewise.  Every filter must still return what the exact scan returns."""
    import quiver_amd as q
    rng = np.random.default_rng(77)
    n, dim, nq, k = 70_000, 768, 64, 10

    def bf16_exact(x):                       # keep the top 16 bits: values exact in bfloat16
        return (x.astype(np.float32).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)

    def half_way(x):                         # bfloat16 value + exactly half a bfloat16 ulp, away from zero
        return ((x.astype(np.float32).view(np.uint32) & np.uint32(0xFFFF0000)) | np.uint32(0x00008000)).view(np.float32)

    centre = rng.standard_normal(dim).astype(np.float32)
    base = (centre[None, :] + 0.35 * rng.standard_normal((n, dim))).astype(np.float32)      # a cluster: many near-ties around every query
    rows = base.copy()
    rows[0::4] = half_way(base[0::4])
    rows[1::4] = bf16_exact(base[1::4])
    rows[2::8] *= np.float32(1e-3)           # mixed norms
    tiny = rng.integers(0, n, 200)
    rows[tiny, ::7] = np.float32(1e-41)      # denormal elements inside ordinary rows
    rows[5] = np.float32(3e-39)              # a whole row below the normal range
    qs = (centre[None, :] + 0.35 * rng.standard_normal((nq, dim))).astype(np.float32)
    qs[0::3] = half_way(qs[0::3])
    qs[1::3] = bf16_exact(qs[1::3])
    qs[7, ::5] = np.float32(1e-41)
    idx = q.DeviceIndex(dim, metric)
    idx.add(rows)
    assert _eq(_exact(idx, qs, k), idx.search(qs, k, batched=True))
    # ... and it was the filter that answered, not the exact redo of overflowed candidate buffers
    import torch
    dq = torch.from_numpy(qs).cuda()
    dr = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    fl = torch.zeros((nq,), dtype=torch.int32, device="cuda")
    idx.search_batched_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), fl.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print("queries sent back to the exact scan:", int(fl.sum().item()), "of", nq)
    assert int(fl.sum().item()) <= nq // 4
    # rows rewritten after ingest: the per-row residual must follow (a stale, smaller one would make the filter drop true neighbours)
    upd = np.arange(0, 4000, 3, dtype=np.uint32)
    newv = half_way(qs[rng.integers(0, nq, len(upd))] + 0.01 * rng.standard_normal((len(upd), dim)).astype(np.float32))
    for r, vrow in zip(upd[:64], newv[:64]):
        idx.update(int(r), vrow)
    exact = _exact(idx, qs, k)
    assert _eq(exact, idx.search(qs, k, batched=True))
    assert len(np.intersect1d(exact[0].ravel(), upd[:64])) > 0          # the rewritten rows are the nearest ones now


@pytest.mark.parametrize("scale", [1e-20, 1e-16, 3e-14])
@pytest.mark.parametrize("metric", ["cosine", "dot_product", "euclidean"])
def test_rows_too_small_for_the_filter_are_still_found(metric, scale):
    """Rows whose norm is so small that the matrix core flushes their operands or products (scores come out 0, or off by more than
    the margin allows for) say nothing to the filter: they must reach the exact pass whatever their score, and the exact pass must
    not trust that score either.  Under cosine a scaled-down copy of the query is its nearest neighbour."""
    import quiver_amd as q
    rng = np.random.default_rng(5)
    n, dim, nq, k = 40_000, 768, 32, 10
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    qs = rng.standard_normal((nq, dim)).astype(np.float32)
    for j in range(nq):                                         # three tiny near-copies of every query, spread over the corpus
        for t in range(3):
            rows[1000 * j + 37 * t + 5] = (qs[j] + 0.05 * (t + 1) * rng.standard_normal(dim)).astype(np.float32) * np.float32(scale)
    idx = q.DeviceIndex(dim, metric)
    idx.add(rows)
    exact = _exact(idx, qs, k)
    assert _eq(exact, idx.search(qs, k, batched=True))
    if metric == "cosine":
        assert all(1000 * j + 5 in exact[0][j] for j in range(nq))          # the tiny copies ARE the nearest rows
    # and tiny queries against ordinary rows
    tq = (qs * np.float32(scale)).astype(np.float32)
    assert _eq(_exact(idx, tq, k), idx.search(tq, k, batched=True))


@pytest.mark.parametrize("metric,dim", [("cosine", 1024), ("dot_product", 1536), ("euclidean", 512), ("squared_euclidean", 256), ("cosine", 640), ("cosine", 2048),
                                        # round 4: every other width goes through the eight-wave kernel's zero-padded form
                                        ("cosine", 64), ("dot_product", 100), ("euclidean", 128), ("cosine", 192), ("squared_euclidean", 200), ("cosine", 300),
                                        ("dot_product", 320), ("cosine", 960), ("euclidean", 1000), ("cosine", 17), ("dot_product", 4), ("cosine", 1)])
def test_filter_kernels_at_other_dimensions(metric, dim):
    """The eight-wave one-term kernel (and its sample mode) runs as it is whenever the dimension is a multiple of 128 with at least 16
    steps of 16; at any other width its zero-padded form takes the float32 rows (the K loop runs to the next multiple of 64
    dimensions, eight steps at least; a chunk past the row's end is replaced by zeros); 2048 is past the one-term rule (three terms
    by default): every shape against the exact scan, 256 queries (whole workgroups of eight waves) and 64 (one query block)."""
    import quiver_amd as q
    n = 70_000
    idx = q.DeviceIndex(dim, metric)
    idx.add_synthetic(20260424, 0, n)
    qs = O.gen_rows(20260425, 0, 256, dim)
    for nq in (256, 64):
        assert _eq(_exact(idx, qs[:nq], 10), idx.search(qs[:nq], 10, batched=True))


def test_concurrent_batches_on_one_index_are_independent():
    """BatchSearch callers run under the collection's read lock (collection.go:647): several batches at once on ONE index, from
    different host threads — each on a call context (stream, workspace) of its own, through every batch shape the router has: one
    query block, whole workgroups of 256, k above 64.  First calls on fresh contexts included (a fill that was not ordered against
    the context's stream broke exactly those once)."""
    import threading
    import quiver_amd as q
    n, dim = 150_000, 768
    idx = q.DeviceIndex(dim, "cosine")
    idx.add_synthetic(20260424, 0, n)
    qs = O.gen_rows(20260426, 0, 256, dim)
    shapes = [(256, 10), (40, 10), (256, 64), (100, 100), (9, 7), (256, 10), (64, 33), (130, 10)]
    want = [_exact(idx, qs[:m], k) for m, k in shapes]
    errs = []

    def worker(t):
        try:
            m, k = shapes[t]
            for _ in range(4):
                if not _eq(want[t], idx.search(qs[:m], k, batched=True)):
                    errs.append((t, m, k))
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex))

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(len(shapes))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert errs == []


def test_cluster_stored_corpus_at_768_dimensions_keeps_the_filter_path():
    """The same at the width and batch size where the eight-wave sample kernel runs (the test above is 128 dimensions x 64 queries, which
    it does not take).  Round 4 briefly selected the bound among per-group MINIMA of the sample: with 30 clusters stored one after the
    other a query's own cluster has ~8 sample groups, the 10th smallest minimum came from another cluster, and every one of the 256
    queries overflowed its candidate list.  Selecting among the rows' own bounds hands back a few per cent."""
    import torch
    import quiver_amd as q
    rng = np.random.default_rng(12)
    dim, n_clusters, per = 768, 30, 4_000
    centres = rng.standard_normal((n_clusters, dim)).astype(np.float32)
    rows = np.concatenate([c + 0.3 * rng.standard_normal((per, dim)).astype(np.float32) for c in centres])
    nq, k = 256, 10
    qs = (centres[rng.integers(0, n_clusters, nq)] + 0.3 * rng.standard_normal((nq, dim))).astype(np.float32)
    idx = q.DeviceIndex(dim, "cosine")
    idx.add(rows)
    dq = torch.from_numpy(qs).cuda()
    dr = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    fl = torch.full((nq,), 7, dtype=torch.int32, device="cuda")
    idx.search_batched_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), fl.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(fl.abs().sum().item()) <= nq // 4
    assert _eq(_exact(idx, qs, k), idx.search(qs, k, batched=True))
    for kk in (64, 100):                                                   # (larger k may overflow here: the results must not care)
        assert _eq(_exact(idx, qs[:64], kk), idx.search(qs[:64], kk, batched=True))


@pytest.mark.parametrize("bf16_rows", [False, True])
@pytest.mark.parametrize("metric,dim,nq", [("cosine", 768, 256), ("dot_product", 512, 600), ("euclidean", 384, 130), ("squared_euclidean", 768, 257),
                                           ("cosine", 512, 256), ("dot_product", 384, 512),
                                           ("cosine", 384, 256), ("euclidean", 512, 256), ("dot_product", 768, 256)])   # every (metric, dimension, plane) instantiation
def test_query_resident_filter_against_the_exact_scan(metric, dim, nq, bf16_rows):
    """k_qreg_filter (qv_qreg.hip; 384, 512 and 768 dimensions, whole workgroups of 256 queries): the queries' operands stay in
    registers and the rows arrive by LDS-DMA, from the float32 tiles or from the bfloat16 copy.  Ragged last tile, tombstones in
    the first and the last tile, a revived row, 1-3 workgroups per row walk (600 queries pad to 768), fewer live rows than k in
    nobody's way: rows, order and float32 bits of the exact scan."""
    import quiver_amd as q
    n = 64 * 700 + 37                                                       # 701 tiles: every workgroup of the 256 walks 2-3 of them
    idx = q.DeviceIndex(dim, metric, bf16_rows=bf16_rows)
    idx.add_synthetic(20260424, 0, n)
    qs = O.gen_rows(20260425, 0, nq, dim)
    rows = O.gen_rows(20260424, 0, n, dim)
    qs[3] = rows[n - 1]                                                     # an exact match in the ragged tile
    qs[5] = rows[0]
    assert _eq(_exact(idx, qs, 10), idx.search(qs, 10, batched=True))
    idx.remove(np.array([0, 1, 63, n - 1, n - 2, 64 * 350 + 7], dtype=np.uint32))
    idx.update(64 * 350 + 7, qs[7])                                         # revived as an exact match of query 7
    e = _exact(idx, qs, 10)
    assert _eq(e, idx.search(qs, 10, batched=True))
    assert e[0][7][0] == 64 * 350 + 7 and e[0][5][0] != 0
    assert _eq(_exact(idx, qs, 64), idx.search(qs, 64, batched=True))
    assert _eq(_exact(idx, qs[:200], 100), idx.search(qs[:200], 100, batched=True))      # the selection path's larger candidate lists


# ---------------------------------------------------------------- more than 64 results per query ---
# HybridIndex.BatchSearch takes any k (hybrid_index.go:677-811); its negative-example branch asks for max(2k, 30) (:516-522).
# Beyond the 64-key wave lists the filter + re-score path selects by radix selection (k_sample_hist, k_cand_*): the result must
# still be the exact scan's, row for row and bit for bit.

@pytest.mark.parametrize("metric,nq,k", [("cosine", 256, 100), ("cosine", 64, 65), ("dot_product", 100, 128), ("euclidean", 40, 300),
                                         ("squared_euclidean", 256, 129), ("cosine", 16, 1000), ("dot_product", 300, 2048), ("cosine", 32, 4096),
                                         ("dot_product", 16, 4096), ("euclidean", 24, 3000), ("squared_euclidean", 12, 2100)])   # (above 2048: the narrowing as separate kernels)
def test_batched_large_k_equals_exact_scan(metric, nq, k):
    import quiver_amd as q
    n, dim = 200_000, 256
    idx = q.DeviceIndex(dim, metric)
    idx.add_synthetic(20260424, 0, n)
    qs = O.gen_rows(20260425, 0, nq, dim)
    exact = _exact(idx, qs, k)                                       # 8 queries at a time: the exact scans (wide lists / one key per row + selection)
    batched = idx.search(qs, k, batched=True)
    assert _eq(exact, batched)
    assert _eq(exact, idx.search(qs, k))
    corpus = O.gen_rows(20260424, 0, n, dim)
    for i in (0, nq - 1):
        er, ed = O.exact_search(q.metric_id(metric), corpus, qs[i], k)
        assert np.array_equal(batched[0][i], er) and np.array_equal(_bits(batched[1][i]), _bits(ed))


@pytest.mark.parametrize("metric,dim,nq,k,rowmajor", [
    ("cosine", 100, 64, 100, True), ("dot_product", 100, 48, 200, True), ("euclidean", 256, 40, 300, True), ("squared_euclidean", 72, 64, 129, True),   # k_cand_exact_wave on the row-major copy (100, 72: a last slab of 1 chunk)
    ("cosine", 50, 64, 100, False), ("squared_euclidean", 100, 64, 80, False), ("dot_product", 36, 40, 70, False), ("euclidean", 50, 33, 65, False),    # ... on the tiles, padded rows (dim % 4 != 0) and short last slabs
    ("euclidean", 256, 256, 700, False), ("squared_euclidean", 100, 300, 1000, False), ("cosine", 72, 300, 1000, False), ("dot_product", 64, 40, 600, False),
    ("cosine", 2048, 32, 100, False), ("dot_product", 2048, 64, 500, False), ("euclidean", 1536, 24, 70, True)])                                         # wide rows: 64 / 48 slabs a row, the query's 8 / 6 KiB in LDS                               # k_tp_exact (the tile pass) for the other metrics, a short last slab
def test_batched_large_k_layouts_and_odd_dimensions(metric, dim, nq, k, rowmajor):
    """the exact passes of a large-k batch: a wave per 32 survivors (row-major copy or tiles) and, for many survivors, the pass over the tiles;
    120 000 rows: below the guessed bound's range, so the sample is half the corpus and its bound exact (the histogram kernels from k ~ 400)"""
    import quiver_amd as q
    n = 120_000
    idx = q.DeviceIndex(dim, metric, rowmajor=rowmajor)
    idx.add_synthetic(20260424, 0, n)
    qs = O.gen_rows(20260425, 0, nq, dim)
    batched = idx.search(qs, k, batched=True)
    assert _eq(_exact(idx, qs, k), batched)
    corpus = O.gen_rows(20260424, 0, n, dim)
    for i in (0, nq - 1):
        er, ed = O.exact_search(q.metric_id(metric), corpus, qs[i], k)
        assert np.array_equal(batched[0][i], er) and np.array_equal(_bits(batched[1][i]), _bits(ed))


@pytest.mark.parametrize("metric,k", [("cosine", 100), ("dot_product", 1000), ("euclidean", 300), ("squared_euclidean", 65)])
def test_batched_large_k_guessed_bound_on_a_large_corpus(metric, k):
    """from 131 072 rows the bound of a batch with k >= 16 is a guess from 32 768 sample rows or more, checked after the filter (batched_guess)"""
    import quiver_amd as q
    n, dim, nq = 300_000, 64, 96
    idx = q.DeviceIndex(dim, metric)
    idx.add_synthetic(20260424, 0, n)
    qs = O.gen_rows(20260425, 0, nq, dim)
    batched = idx.search(qs, k, batched=True)
    assert _eq(_exact(idx, qs, k), batched)
    corpus = O.gen_rows(20260424, 0, n, dim)
    for i in (0, nq - 1):
        er, ed = O.exact_search(q.metric_id(metric), corpus, qs[i], k)
        assert np.array_equal(batched[0][i], er) and np.array_equal(_bits(batched[1][i]), _bits(ed))


def test_batched_large_k_guessed_bound_with_most_rows_dead():
    """the guessed rank counts on k S / N live rows of the sample lying under the k-th distance; with 70 % of the rows removed far fewer do,
    the guess is too tight for most queries, and the hand-back answers them — the result is the exact scan's over the live rows either way"""
    import quiver_amd as q
    n, dim, nq, k = 300_000, 64, 40, 120
    idx = q.DeviceIndex(dim, "cosine")
    idx.add_synthetic(20260424, 0, n)
    rng = np.random.default_rng(9)
    dead = rng.choice(n, int(0.7 * n), replace=False).astype(np.uint32)
    idx.remove(dead)
    alive = np.ones(n, bool); alive[dead] = False
    qs = O.gen_rows(20260425, 0, nq, dim)
    got = idx.search(qs, k, batched=True)
    assert _eq(_exact(idx, qs, k), got)
    corpus = O.gen_rows(20260424, 0, n, dim)
    for i in (0, nq - 1):
        er, ed = O.exact_search(0, corpus, qs[i], k, alive=alive)
        assert np.array_equal(got[0][i], er) and np.array_equal(_bits(got[1][i]), _bits(ed))


def test_batched_large_k_guessed_bound_that_fails_is_handed_back():
    """clusters of 256 rows = two 128-row groups, stored one after the other; the sample takes every fourth group, so half of the clusters have
    HALF their rows in it (expected: 22 %) and the others none.  A query at the centre of an over-sampled cluster gets a bound under which
    fewer than k rows of the corpus lie: the check after the filter (H <= U) hands it back and the exact scan answers it.  Either way the
    answer is the exact scan's."""
    import torch
    import quiver_amd as q
    dim, k, per = 64, 200, 256
    rng = np.random.default_rng(5)
    centers = rng.standard_normal((1172, dim)).astype(np.float32)
    rows = np.concatenate([c + 0.02 * rng.standard_normal((per, dim)).astype(np.float32) for c in centers])       # 300 032 rows
    idx = q.DeviceIndex(dim, "cosine")
    idx.add(rows)
    qs = np.concatenate([centers[0:40:2], centers[1:41:2], rng.standard_normal((24, dim)).astype(np.float32)]).astype(np.float32)
    nq = len(qs)
    dq = torch.from_numpy(qs).cuda()
    dr = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    fl = torch.zeros((nq,), dtype=torch.int32, device="cuda")
    idx.search_batched_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), fl.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    flagged = fl.cpu().numpy() != 0
    assert flagged.any() and not flagged.all(), "this corpus no longer makes a guessed bound fail for some queries only (%d of %d handed back)" % (flagged.sum(), nq)
    got = idx.search(qs, k, batched=True)                                   # the host call redoes the handed-back queries
    assert _eq(_exact(idx, qs, k), got)
    hr, hd = dr.cpu().numpy().view(np.uint32), dd.cpu().numpy()
    for i in range(nq):
        if i in (0, 1, 20, 21, nq - 1) or not flagged[i]:
            if not flagged[i]: assert np.array_equal(hr[i], got[0][i]) and np.array_equal(_bits(hd[i]), _bits(got[1][i])), i
        if i in (0, 1, 20, 21, nq - 1):
            er, ed = O.exact_search(0, rows, qs[i], k)
            assert np.array_equal(got[0][i], er), i
            assert np.array_equal(_bits(got[1][i]), _bits(ed)), i


def test_batched_large_k_ties_tombstones_and_a_clustered_corpus():
    """duplicated rows (exact ties, resolved by row), dead rows, and a corpus stored cluster by cluster so that the sample's
    bound is loose for some queries (candidate overflow -> the caller's exact redo)"""
    import quiver_amd as q
    dim, nq, k = 64, 48, 200
    rng = np.random.default_rng(11)
    centers = rng.standard_normal((8, dim)).astype(np.float32)
    rows = np.concatenate([c + 0.05 * rng.standard_normal((12_000, dim)).astype(np.float32) for c in centers])
    rows[5000:5400] = rows[100]                                       # 400 identical rows
    idx = q.DeviceIndex(dim, "cosine")
    idx.add(rows)
    dead = rng.choice(len(rows), 3000, replace=False).astype(np.uint32)
    idx.remove(dead)
    alive = np.ones(len(rows), bool); alive[dead] = False
    qs = np.concatenate([rows[100:101], centers[3:4], rng.standard_normal((nq - 2, dim)).astype(np.float32)])
    got = idx.search(qs, k, batched=True)
    for i in range(nq):
        er, ed = O.exact_search(0, rows, qs[i], k, alive=alive)
        assert np.array_equal(got[0][i], er), i
        assert np.array_equal(_bits(got[1][i]), _bits(ed)), i


def test_batched_large_k_candidate_overflow_is_redone_exactly():
    """20 000 copies of one row sit at the same distance from a query: far more candidates than the batch's slots hold (16 k per query
    at k = 100), so the filter hands that query back (redo flag) and qv_index_search_batched answers it with the exact scan — whose
    selection then meets 20 000 keys tied on all 32 distance bits and takes the first of them in row order"""
    import quiver_amd as q
    dim, nq, k = 96, 40, 100
    rng = np.random.default_rng(21)
    rows = rng.standard_normal((120_000, dim)).astype(np.float32)
    rows[30_000:50_000] = rows[7]
    idx = q.DeviceIndex(dim, "cosine")
    idx.add(rows)
    qs = rng.standard_normal((nq, dim)).astype(np.float32)
    qs[3] = rows[7] + 0.01 * rng.standard_normal(dim).astype(np.float32)
    got = idx.search(qs, k, batched=True)
    for i in (0, 3, nq - 1):
        er, ed = O.exact_search(0, rows, qs[i], k)
        assert np.array_equal(got[0][i], er), i
        assert np.array_equal(_bits(got[1][i]), _bits(ed)), i
    assert got[0][3][0] == 7 and list(got[0][3][1:6]) == [30_000, 30_001, 30_002, 30_003, 30_004]
