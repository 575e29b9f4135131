"""Kernel instantiations that no other -m gpu test reaches (tools/kernel_coverage.py lists what libqv.so ships against what the
traced test run launched; `profiles/r06_kernel_coverage.txt`).  Each case below names the instantiation it is there for and checks
its results against the oracle or the exact scan, like every other test of the path."""
import numpy as np
import pytest

import quiver_amd
from quiver_amd.device_index import DeviceGraph
from tests import _oracle as O

pytestmark = pytest.mark.gpu

METRICS = ["cosine", "l2", "l2sq", "dot", "l1", "cosine_f32", "l2_f32", "dot_f32", "l2sq_f64"]
LAT = {"cosine", "l2", "dot", "l1", "l2sq_f64"}            # SplitOK: the latency form (a workgroup of eight waves per query) applies


def _knn(rows, metric, m):
    n = rows.shape[0]
    idx = quiver_amd.DeviceIndex(rows.shape[1], metric, rowmajor=True)
    idx.add(rows)
    nbr, _, _ = idx.search(rows, m + 1)
    links = np.zeros((n, m), np.uint32); deg = np.zeros(n, np.uint32)
    for i in range(n):
        l = [int(x) for x in nbr[i] if int(x) != i][:m]
        deg[i] = len(l); links[i, :len(l)] = l
    return idx, deg, links


@pytest.mark.parametrize("dim", [64, 100])                 # 64: the query through LDS (QLDS) and the round-6 hop front; 100: neither
@pytest.mark.parametrize("metric", METRICS)
def test_every_list_size_of_the_wave_traversal(metric, dim):
    """k_hnsw_search_wave<metric, 4, S, QLDS, W>: S = 2 / 4 / 5 / 8 / 9 list registers per lane for efSearch < 128 / < 256 / < 320 / < 512 /
    512, as a wave per query (W = 1: more than 768 queries in one call) and — for the metrics whose chain can be split — as a workgroup
    per query (W = 8: at most 256 queries).  Rows, float32 bits, counts, evaluation counts of the oracle's HNSW.Search."""
    n, m, k = 1400, 16, 10
    rows = O.gen_rows(9090 + dim, 0, n, dim)
    idx, deg, links = _knn(rows, metric, m)
    g = DeviceGraph(idx, np.zeros(n, np.int8), deg, links, entry=11)
    mid = quiver_amd.metric_id(metric)
    qs = O.gen_rows(9091, 0, 800, dim)
    for ef in (64, 128, 256, 400, 512):
        o = O.HNSW(mid, dim, M=m // 2, maxM0=m, efSearch=ef, maxLevel=1, seed=1)
        o.load_flat(rows, deg, links, 11)
        forms = [qs] + ([qs[:48]] if metric in LAT and dim % 32 == 0 else [])
        for batch in forms:
            r, d, c, ev = g.search(batch, k, ef, with_evals=True)
            for i in range(0, batch.shape[0], max(1, batch.shape[0] // 6)):
                ro, do, eo = o.search(batch[i], k, with_evals=True)
                assert c[i] == k, (ef, i)                  # (a connected single-level graph: the walk fills every result)
                assert r[i].tolist() == ro.tolist(), (ef, i)
                assert d[i].tobytes() == do.tobytes(), (ef, i)
                assert int(ev[i]) == eo - 1, (ef, i)


def test_float64_squared_l2_graph_paths_off_the_latency_form():
    """k_hnsw_search<QV_L2SQ_F64, 16, 1> (the exact-heap kernel as one wave per query: a dimension the latency form does not take, ties
    that flag queries) and k_graph_link_dists<QV_L2SQ_F64, 4> (qv_graph_make_buildable on an uploaded graph)."""
    n, dim, m, k, ef = 1200, 100, 16, 10, 64
    base = O.gen_rows(7171, 0, n // 2, dim)
    rows = np.ascontiguousarray(np.concatenate([base, base]))                 # every vector twice: equal distances on every hop
    idx, deg, links = _knn(rows, "l2sq_f64", m)
    g = DeviceGraph(idx, np.zeros(n, np.int8), deg, links, entry=5)
    o = O.HNSW(quiver_amd.metric_id("l2sq_f64"), dim, M=m // 2, maxM0=m, efSearch=ef, maxLevel=1, seed=1)
    o.load_flat(rows, deg, links, 5)
    qs = O.gen_rows(7172, 0, 900, dim)
    r, d, c, ev = g.search(qs, k, ef, with_evals=True)
    for i in range(0, 900, 75):
        ro, do, eo = o.search(qs[i], k, with_evals=True)
        assert r[i, :c[i]].tolist() == ro[:c[i]].tolist() and d[i, :c[i]].tobytes() == do[:c[i]].tobytes(), i
    assert g.stats()["search_redo"] > 0, "no query was flagged: the exact-heap kernel did not run"
    g.make_buildable(40)                                                       # scores the uploaded links on the device
    extra = O.gen_rows(7173, 0, 64, dim)
    first = idx.add(extra)
    g.insert(first, np.zeros(64, np.int8), batch_max=16)
    assert g.info()["n_nodes"] == n + 64
    r2, _, c2 = g.search(extra[:8], 1, ef)                                     # each new node is reachable and its own nearest neighbour
    assert [int(r2[i, 0]) for i in range(8) if c2[i]] == [first + i for i in range(8) if c2[i]]


@pytest.mark.parametrize("metric", ["dot_product", "euclidean", "squared_euclidean"])
def test_sharded_batch_redo_on_the_device_for_the_other_filter_metrics(metric):
    """k_flat_scan_redo<L2 / L2SQ / DOT> (tests/test_gpu_sharded_index.py has the cosine case): a corpus stored cluster by cluster makes
    the filter hand queries back; the sharded handle re-scans those on the device.  Equal to one index's exact scan."""
    import torch
    rng = np.random.default_rng(31)
    # clusters of 6 000 near-copies of one vector: every row of a query's cluster lies within the filter's margin of the k-th distance,
    # 6 000 candidates for 4 096 slots — the filter hands the query back whatever the sample said
    dim, n_clusters, per = 256, 12, 6_000
    centres = rng.standard_normal((n_clusters, dim)).astype(np.float32)
    rows = np.concatenate([c + 1e-4 * rng.standard_normal((per, dim)).astype(np.float32) for c in centres])
    nq, k = 256, 10
    qs = (centres[rng.integers(0, n_clusters, nq)] + 1e-4 * rng.standard_normal((nq, dim))).astype(np.float32)
    one = quiver_amd.DeviceIndex(dim, metric, filter="off")
    one.add(rows)
    er, ed, _ = one.search(qs, k)
    sh = quiver_amd.ShardedIndex(dim, metric, devices=[0, 0], peer_copy=True)
    gids = sh.add(rows)
    first = quiver_amd.DeviceIndex(dim, metric)
    first.add(rows[:sh.shard_info(0)["rows"]])
    dq = torch.from_numpy(qs).cuda()
    fr = torch.empty((nq, k), dtype=torch.int32, device="cuda"); fd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    fl = torch.zeros((nq,), dtype=torch.int32, device="cuda")
    first.search_batched_device(dq.data_ptr(), nq, k, fr.data_ptr(), fd.data_ptr(), fl.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int((fl != 0).sum().item()) >= 1, "this corpus no longer makes the %s filter hand anything back: the test exercises nothing" % metric
    dr = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    sh.search_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), torch.cuda.current_stream().cuda_stream)
    sh.sync(); torch.cuda.synchronize()
    assert np.array_equal(dr.cpu().numpy().view(np.uint32), gids[er])
    assert np.array_equal(dd.cpu().numpy().view(np.uint32), ed.view(np.uint32))
    sh.close()


def test_sharded_full_ranking_beyond_the_selection_range():
    """k_shard_keys: a sharded search with k above the radix selection's 8192 (a filtered Collection.Search asks k = Size(),
    collection.go:679-682) ranks the shards' gathered keys by a full sort."""
    n, dim = 20_000, 32
    rows = O.gen_rows(4545, 0, n, dim)
    one = quiver_amd.DeviceIndex(dim, "cosine"); one.add(rows)
    sh = quiver_amd.ShardedIndex(dim, "cosine", devices=[0, 0], peer_copy=True)
    gids = sh.add(rows)
    q = O.gen_rows(4546, 0, 1, dim)
    er, ed, _ = one.search(q, n)
    r, d, c = sh.search(q, n)
    assert int(c[0]) == n and np.array_equal(r[0], gids[er[0]]) and np.array_equal(d.view(np.uint32), ed.view(np.uint32))
    sh.close()


@pytest.mark.parametrize("metric", ["cosine", "dot_product", "euclidean", "squared_euclidean"])
def test_sample_bound_by_wave_lists_on_a_long_corpus(metric):
    """k_sample_bound<M>: up to 64 results per query the bound of the batched path comes from k_sample_select while k chunks of the sample
    fit its LDS; at k = 64 over 2.6 M rows they do not (the sample is 430 k rows) and the wave-list kernel takes over."""
    n, dim, nq, k = 2_600_000, 64, 16, 64
    idx = quiver_amd.DeviceIndex(dim, metric)
    idx.add_synthetic(777, 0, n)
    qs = O.gen_rows(778, 0, nq, dim)
    br, bd, _ = idx.search(qs, k, batched=True)
    idx.set_filter("off")
    er, ed, _ = idx.search(qs, k)
    assert np.array_equal(br, er) and np.array_equal(bd.view(np.uint32), ed.view(np.uint32))


@pytest.mark.parametrize("metric", ["euclidean", "squared_euclidean"])
def test_large_k_sample_bound_from_the_histogram_state(metric):
    """k_sample_bound_from_state<L2 / L2SQ>: a batch asking for more than 1024 results per query takes its sample bound from two
    histogram windows over the sample's bounds (tests/test_gpu_batched.py has cosine and dot)."""
    n, dim, nq, k = 80_000, 64, 16, 1500
    idx = quiver_amd.DeviceIndex(dim, metric)
    idx.add_synthetic(555, 0, n)
    qs = O.gen_rows(556, 0, nq, dim)
    br, bd, _ = idx.search(qs, k, batched=True)
    idx.set_filter("off")
    er, ed, _ = idx.search(qs, k)
    assert np.array_equal(br, er) and np.array_equal(bd.view(np.uint32), ed.view(np.uint32))
