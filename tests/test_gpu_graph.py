"""Device graph traversal (qv_graph_*) on graphs that were NOT built by Insert: an exact k-NN graph made
with the product's own flat scan.  The oracle runs HNSW.Search (pkg/hnsw/hnsw.go:602-713) on the
identical graph (qvo_hnsw_load_flat); rows, distance bits, counts and evaluation counts must agree."""
import numpy as np
import pytest

import quiver_amd
from tests import _oracle as O

pytestmark = pytest.mark.gpu


def _knn_graph(rows, metric, m):
    """exact m-NN lists (self excluded) through the product's flat scan"""
    n = rows.shape[0]
    idx = quiver_amd.DeviceIndex(rows.shape[1], metric, rowmajor=True)
    idx.add(rows)
    nbr, _, _ = idx.search(rows, m + 1)
    links = np.zeros((n, m), np.uint32); deg = np.zeros(n, np.uint32)
    for i in range(n):
        l = [int(x) for x in nbr[i] if int(x) != i][:m]
        deg[i] = len(l); links[i, :len(l)] = l
    return idx, deg, links


@pytest.mark.parametrize("metric,dim,n,m,ef,k", [
    ("cosine", 768, 3000, 32, 128, 10),
    ("cosine", 100, 2500, 16, 64, 10),       # dim4 = 25: partial DMA pieces
    ("l2", 64, 2000, 32, 200, 50),           # ef > 128: 4 list registers per lane
    ("dot", 36, 1500, 24, 40, 40),           # dim4 = 9: one full piece + a tail
    ("cosine_f32", 128, 2000, 32, 128, 10),
    ("l2sq", 20, 1000, 8, 16, 5),
    # dimensions that are a multiple of 32 take the query through LDS (hnsw_eval_rows_qlds): every query type and metric once
    ("dot", 96, 1500, 16, 64, 10),
    ("l1", 32, 1500, 16, 64, 10),
    ("dot_f32", 64, 1500, 16, 300, 10),      # ef >= 256: 5 list registers per lane
    ("l2_f32", 160, 1500, 16, 512, 10),      # ef = 512: 9 list registers per lane
    ("l2sq_f64", 224, 1200, 16, 64, 10),
    ("l2sq", 256, 1200, 16, 64, 10),
])
def test_knn_graph_traversal_identical_to_oracle(metric, dim, n, m, ef, k):
    mid = quiver_amd.metric_id(metric)
    rows = O.gen_rows(4242, 0, n, dim)
    idx, deg, links = _knn_graph(rows, metric, m)
    g = quiver_amd.DeviceGraph(idx, np.zeros(n, np.int8), deg, links, entry=7)
    o = O.HNSW(mid, dim, M=max(m // 2, 1), maxM0=m, efSearch=ef, maxLevel=1, seed=1)
    o.load_flat(rows, deg, links, 7)
    qs = O.gen_rows(4243, 0, 96, dim)
    r, d, c, ev = g.search(qs, k, ef, with_evals=True)
    for i in range(qs.shape[0]):
        ro, do, eo = o.search(qs[i], k, with_evals=True)
        if c[i] == k:                                     # filled by the graph search alone
            assert r[i].tolist() == ro.tolist(), i
            assert d[i].tobytes() == do.tobytes(), i
            assert int(ev[i]) == eo - 1, i               # the reference also evaluates the entry point once more up front (hnsw.go:637)
        else:                                             # under-filled: the caller tops up (hnsw.go:676-710); the prefix must agree
            assert c[i] < k
            assert r[i, :c[i]].tolist() == ro[:c[i]].tolist(), i


def test_knn_graph_with_duplicate_rows_goes_through_the_exact_heap_kernel():
    rows = O.gen_rows(5150, 0, 1200, 48)
    rows[600:] = rows[:600]                                # every vector twice: equal distances everywhere
    idx, deg, links = _knn_graph(rows, "cosine", 16)
    g = quiver_amd.DeviceGraph(idx, np.zeros(1200, np.int8), deg, links, entry=0)
    o = O.HNSW(0, 48, M=8, maxM0=16, efSearch=64, maxLevel=1, seed=1)
    o.load_flat(rows, deg, links, 0)
    qs = O.gen_rows(5151, 0, 32, 48)
    r, d, c, ev = g.search(qs, 10, 64, with_evals=True)
    for i in range(32):
        ro, do, eo = o.search(qs[i], 10, with_evals=True)
        n = min(int(c[i]), 10)
        assert r[i, :n].tolist() == ro[:n].tolist() and d[i, :n].tobytes() == do[:n].tobytes()


def test_device_pointer_form_equals_host_pointer_form():
    import torch
    n, dim, m, ef, k = 4000, 96, 24, 100, 10
    rows = O.gen_rows(777, 0, n, dim)
    idx, deg, links = _knn_graph(rows, "cosine", m)
    g = quiver_amd.DeviceGraph(idx, np.zeros(n, np.int8), deg, links, entry=11)
    qs = O.gen_rows(778, 0, 300, dim)
    r, d, c, ev = g.search(qs, k, ef, with_evals=True)
    dq = torch.from_numpy(qs).cuda()
    dr = torch.empty((300, k), dtype=torch.int32, device="cuda"); dd = torch.empty((300, k), dtype=torch.float32, device="cuda")
    dc = torch.empty(300, dtype=torch.int32, device="cuda"); de = torch.empty(300, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):                                      # repeated calls: fresh visited epochs each time
        g.search_device(dq.data_ptr(), 300, k, ef, dr.data_ptr(), dd.data_ptr(), dc.data_ptr(), de.data_ptr(), s)
    torch.cuda.synchronize()
    c2 = dc.cpu().numpy().view(np.uint32)
    ok = c2 != 0xFFFFFFFE                                   # tie-flagged queries are redone by the host-pointer form only
    assert ok.sum() >= 290
    assert np.array_equal(c2[ok], c[ok])
    assert np.array_equal(dr.cpu().numpy().view(np.uint32)[ok], r[ok])
    assert np.array_equal(dd.cpu().numpy().view(np.uint32)[ok], d.view(np.uint32)[ok])
    assert np.array_equal(de.cpu().numpy().view(np.uint32)[ok], ev[ok])


def test_committed_hnsw_fixture_multi_level_graph():
    """tests/golden/hnsw_3kx32_cosine.npz: a graph built by Insert with upper levels (and the reference's self-link quirk,
    so many level-0 searches under-fill).  Device traversal == fixture where the graph search filled k; elsewhere the
    reference tops up by brute force (hnsw.go:676-710), i.e. the fixture holds the exact top-k."""
    import os
    g = np.load(os.path.join(O.ROOT, "tests", "golden", "hnsw_3kx32_cosine.npz"))
    n, dim, k, ef = int(g["n"]), int(g["dim"]), 10, int(g["efSearch"])
    rows = O.gen_rows(int(g["corpus_seed"]), 0, n, dim)
    idx = quiver_amd.DeviceIndex(dim, "cosine", rowmajor=True)
    idx.add(rows)
    dg = quiver_amd.DeviceGraph(idx, g["levels"], g["l0_deg"], g["l0_links"], entry=int(g["entry"]), cur_level=int(g["cur_level"]),
                                up_off=g["up_off"], up_links=g["up_links"])
    qs = O.gen_rows(int(g["query_seed"]), 0, g["rows"].shape[0], dim)
    r, d, c, ev = dg.search(qs, k, ef, with_evals=True)
    filled = 0
    for i in range(qs.shape[0]):
        if c[i] == k:
            filled += 1
            assert np.array_equal(r[i], g["rows"][i]) and np.array_equal(d[i].view(np.uint32), g["dist"][i].view(np.uint32)), i
            assert int(ev[i]) == int(g["evals"][i]) - 1, i        # the reference evaluates the entry point once more up front (hnsw.go:637)
        else:
            er, ed, _ = idx.search(qs[i], k)
            assert np.array_equal(er[0], g["rows"][i]) and np.array_equal(ed[0].view(np.uint32), g["dist"][i].view(np.uint32)), i
    assert 0 < filled < qs.shape[0]                                # both branches exercised


def test_update_and_remove_reach_the_row_major_copy_the_traversal_reads():
    """qv_index_update rewrites both layouts; the graph walk (row-major copy, LDS-DMA) must see the new vector"""
    n, dim, m = 600, 40, 16
    rows = O.gen_rows(99, 0, n, dim)
    idx, deg, links = _knn_graph(rows, "cosine", m)
    q = O.gen_rows(100, 0, 1, dim)[0]
    far = int(np.argmax(O.all_distances(0, rows, q)))            # the row farthest from the query ...
    idx.update(far, q)                                           # ... becomes the query itself
    rows2 = rows.copy(); rows2[far] = q
    g = quiver_amd.DeviceGraph(idx, np.zeros(n, np.int8), deg, links, entry=far)
    o = O.HNSW(0, dim, M=m // 2, maxM0=m, efSearch=64, maxLevel=1, seed=1)
    o.load_flat(rows2, deg, links, far)
    r, d, c = g.search(q[None, :], 5, 64)
    ro, do = o.search(q, 5)
    assert c[0] == 5 and r[0].tolist() == ro.tolist() and d[0].tobytes() == do.tobytes()
    assert r[0, 0] == far and d[0, 0] == np.float32(O.distance(0, q, q))
    # the flat scan (tile layout) agrees
    fr, fd, _ = idx.search(q, 1)
    assert fr[0, 0] == far and fd[0, 0] == d[0, 0]


def test_graph_replicas_round_robin_equals_one_graph():
    """HNSW across GPUs = replicas only: one graph per device (two replicas co-located on device 0 here), a batch
    cut over them and walked concurrently; same answer, query for query, as one graph walking the whole batch"""
    import quiver_amd
    from quiver_amd.device_index import DeviceGraph, GraphReplicas, random_levels
    n, dim, k, ef = 4000, 48, 7, 40
    rows = O.gen_rows(901, 0, n, dim)
    levels = random_levels(n, 16, 5)
    reps = GraphReplicas(rows, levels, "cosine", devices=[0, 0, 0], m=8, max_m0=16, ef_construction=60)
    one_idx = quiver_amd.DeviceIndex(dim, "cosine", rowmajor=True)
    one_idx.add(rows)
    one = DeviceGraph.build(one_idx, levels, m=8, max_m0=16, ef_construction=60)
    for a, b in zip(reps.graphs[0].export(), one.export()):
        assert np.array_equal(a, b)                                   # the build is deterministic: replicas are copies
    qs = O.gen_rows(902, 0, 50, dim)
    r, d, c = reps.search(qs, k, ef)
    r1, d1, c1 = one.search(qs, k, ef)
    assert np.array_equal(c, c1)
    for i in range(50):
        assert np.array_equal(r[i, :c[i]], r1[i, :c1[i]]) and d[i, :c[i]].tobytes() == d1[i, :c1[i]].tobytes()
    r2, d2, c2 = reps.search(qs[:2], k, ef)                           # fewer queries than replicas
    assert np.array_equal(r2[:, :k], r[:2, :k])
    reps.close(); one.close(); one_idx.close()


@pytest.mark.parametrize("metric,dim,n,m,ef", [
    ("cosine", 768, 3000, 32, 128),
    ("cosine", 96, 2500, 24, 64),            # 3 pieces for 4 waves: one wave of the latency form has no columns
    ("l2", 160, 2000, 32, 200),
    ("l1", 32, 1500, 16, 64),
    ("l2sq_f64", 224, 1200, 16, 300),
    ("dot", 128, 1500, 16, 64),              # bound from the query's norm (k_hnsw_prep_queries) and the cached row norm
    ("cosine", 1536, 1200, 32, 64),          # 32 rows of this dimension do not fit the LDS: rounds of 16
    ("l2", 4096, 500, 16, 40),               # rounds of 8
])
def test_split_chains_both_kernel_forms_identical_to_oracle(metric, dim, n, m, ef):
    """The metrics whose float64 chain the latency form evaluates as several lanes' and waves' partial chains and certifies
    (SplitOK, qv_hnsw.hip): a batch of 320 queries takes the throughput form (a wave per query, every row one chain), batches
    of <= 64 the latency form (a workgroup per query, lat_eval_rows).  Both must give the oracle's rows, float32 bits and
    evaluation counts.
    Half of the queries ARE corpus rows: their distance to themselves is ~0, where the certification cannot hold and the
    row is walked again as one chain (the fallback)."""
    mid = quiver_amd.metric_id(metric)
    k = 10
    rows = O.gen_rows(5151, 0, n, dim)
    idx, deg, links = _knn_graph(rows, metric, m)
    g = quiver_amd.DeviceGraph(idx, np.zeros(n, np.int8), deg, links, entry=3)
    o = O.HNSW(mid, dim, M=max(m // 2, 1), maxM0=m, efSearch=ef, maxLevel=1, seed=1)
    o.load_flat(rows, deg, links, 3)
    qs = np.concatenate([O.gen_rows(5152, 0, 160, dim), rows[np.arange(160) * 7 % n]])
    r, d, c, ev = g.search(qs, k, ef, with_evals=True)                       # 320 > CUs: throughput form
    for lo in range(0, 320, 64):                                             # latency form
        r2, d2, c2, ev2 = g.search(qs[lo:lo + 64], k, ef, with_evals=True)
        assert np.array_equal(c2, c[lo:lo + 64]) and np.array_equal(ev2, ev[lo:lo + 64]), lo
        for i in range(64):
            assert r2[i, :c2[i]].tolist() == r[lo + i, :c2[i]].tolist(), (lo, i)
            assert d2[i, :c2[i]].tobytes() == d[lo + i, :c2[i]].tobytes(), (lo, i)
    for i in list(range(0, 320, 9)):
        ro, do, eo = o.search(qs[i], k, with_evals=True)
        assert c[i] <= k
        assert r[i, :c[i]].tolist() == ro[:c[i]].tolist(), i
        assert d[i, :c[i]].tobytes() == do[:c[i]].tobytes(), i
        if c[i] == k:
            assert int(ev[i]) == eo - 1, i


@pytest.mark.parametrize("metric,dim,m,ef,max_level,style", [
    ("cosine", 768, 16, 128, 1, 0),
    ("cosine", 64, 16, 64, 3, 2),            # multi-level: nodes repeated inside a level-0 list (the self-link quirk) count at their first occurrence
    ("l2", 96, 16, 200, 1, 0),
    ("l2sq", 128, 8, 40, 3, 1),              # exact ties everywhere: flagged queries go through the exact-heap kernel
    ("dot", 160, 16, 128, 1, 0),
    ("l1", 32, 16, 64, 3, 0),
    ("cosine_f32", 256, 16, 128, 1, 2),
    ("l2_f32", 64, 4, 16, 1, 0),
    ("dot_f32", 384, 16, 300, 1, 0),         # 5 list registers per lane
    ("l2sq_f64", 224, 16, 512, 1, 0),        # 9 list registers per lane
    ("cosine", 1536, 16, 128, 1, 0),
    ("cosine", 100, 16, 64, 1, 0),           # not a multiple of 32: the wave form WITHOUT the round-6 front (query streamed, open-addressed visited set)
    ("l2", 64, 32, 64, 1, 0),                # MaxM0 = 64 > 32 links: likewise
])
def test_wave_per_query_form_with_the_round6_front_identical_to_oracle(metric, dim, m, ef, max_level, style):
    """A call of more than 768 queries takes the wave-per-query form of the traversal (k_hnsw_search_wave<.., W = 1>), whose hop —
    on a row-major index with a dimension that is a multiple of 32 and at most 32 links per node — requests slab 0 of every link
    beside the visited test, keeps the visited set in buckets, the query resident in LDS as float32 words broadcast by v_readlane,
    and the next adjacency list a hop ahead (qv_hnsw.hip, HnswOpts::front).  Rows, float32 bits, counts and evaluation counts must
    equal the oracle's HNSW.Search (pkg/hnsw/hnsw.go:602-713) on the exported graph, and the latency form's (batches of 64)."""
    from quiver_amd.device_index import DeviceGraph, random_levels
    import zlib
    rng = np.random.default_rng(zlib.crc32(("%s/%d/%d" % (metric, dim, m)).encode()))
    n, k, nq = 2500, 10, 900
    if style == 1:
        rows = rng.choice(np.array([-2.0, -1.0, -0.5, 0.0, 0.5, 1.0, 3.0], np.float32), size=(n, dim)).astype(np.float32)
    elif style == 2:
        c = rng.standard_normal((60, dim)).astype(np.float32)
        rows = (c[rng.integers(0, 60, n)] + (rng.standard_normal((n, dim)) * 1e-3).astype(np.float32)).astype(np.float32)
        rows[rng.integers(0, n, n // 10)] = rows[rng.integers(0, n, n // 10)]
    else:
        rows = O.gen_rows(6161, 0, n, dim)
    rows = np.ascontiguousarray(rows)
    idx = quiver_amd.DeviceIndex(dim, metric, rowmajor=True)
    idx.add(rows)
    levels = random_levels(n, max_level, 5)
    g = DeviceGraph.build(idx, levels, m=m, max_m0=2 * m, ef_construction=60)
    info = g.info()
    lv, l0_deg, l0_links, up_off, up_links = g.export()
    o = O.HNSW(quiver_amd.metric_id(metric), dim, M=m, maxM0=2 * m, efConstruction=60, efSearch=ef, maxLevel=max_level, seed=5)
    o.load_graph(rows, lv, 2 * m, m, l0_deg, l0_links, up_off, up_links, info["entry"], info["cur_level"])
    qs = np.concatenate([O.gen_rows(6162, 0, nq - 300, dim) if style == 0 else rows[rng.integers(0, n, nq - 300)] + np.float32(0.01),
                         rows[rng.integers(0, n, 300)]]).astype(np.float32)
    r, d, c, ev = g.search(qs, k, ef, with_evals=True)                       # 900 > 768 queries: a wave per query
    for lo in range(0, nq, 128):                                             # <= 256 queries: the latency form (for its metrics)
        r2, d2, c2, ev2 = g.search(qs[lo:lo + 128], k, ef, with_evals=True)
        assert np.array_equal(c2, c[lo:lo + 128]) and np.array_equal(ev2, ev[lo:lo + 128]), lo
        for i in range(c2.shape[0]):
            assert r2[i, :c2[i]].tolist() == r[lo + i, :c2[i]].tolist() and d2[i, :c2[i]].tobytes() == d[lo + i, :c2[i]].tobytes(), (lo, i)
    filled = 0
    for i in range(0, nq, 11):
        ro, do, eo = o.search(qs[i], k, with_evals=True)
        assert c[i] <= k
        if c[i] == k:                                # filled by the graph search alone: rows, bits and evaluation counts
            assert r[i].tolist() == ro.tolist(), i
            assert d[i].tobytes() == do.tobytes(), i
            assert int(ev[i]) == eo - 1, i
            filled += 1
        # (under-filled — the level quirk's islands on a multi-level graph — the reference answers with the exact top-k, hnsw.go:676-710:
        # the host layer's job, tests/test_gpu_host.py; the graph results themselves are pinned by the latency form above)
    assert filled > 0 or max_level > 1


@pytest.mark.parametrize("metric,dim,ef", [("cosine", 64, 64), ("l2", 96, 128), ("dot", 128, 40), ("l2sq", 32, 64), ("cosine_f32", 64, 300), ("l1", 64, 64)])
def test_a_large_call_equals_the_latency_form_the_oracle_and_sees_updates(metric, dim, ef):
    """A call of thousands of queries (a wave per query, queries through the counter) against the same queries in calls of 256 (the
    latency form) and the oracle: rows, float32 bits, counts, evaluation counts; and rows rewritten afterwards (qv_index_update) must be
    seen by both.  (The measurement build's hub table — QV_HNSW_HUBS=1, qv_hnsw.hip "hubs" — is checked by this test too: it keeps
    copies of rows.)"""
    n, m, k, nq = 6000, 16, 10, 2304
    mid = quiver_amd.metric_id(metric)
    rows = O.gen_rows(3131, 0, n, dim)
    idx, deg, links = _knn_graph(rows, metric, m)
    g = quiver_amd.DeviceGraph(idx, np.zeros(n, np.int8), deg, links, entry=9)
    qs = np.concatenate([O.gen_rows(3132, 0, nq - 256, dim), rows[np.arange(256) * 17 % n]])

    def check(rows_now):
        o = O.HNSW(mid, dim, M=m // 2, maxM0=m, efSearch=ef, maxLevel=1, seed=1)
        o.load_flat(rows_now, deg, links, 9)
        r, d, c, ev = g.search(qs, k, ef, with_evals=True)                   # 2304 queries: hubs
        for lo in range(0, nq, 256):                                         # 256 queries: the latency form, no hubs
            r2, d2, c2, ev2 = g.search(qs[lo:lo + 256], k, ef, with_evals=True)
            assert np.array_equal(c2, c[lo:lo + 256]) and np.array_equal(ev2, ev[lo:lo + 256]), lo
            assert np.array_equal(r2, r[lo:lo + 256]) and np.array_equal(d2.view(np.uint32), d[lo:lo + 256].view(np.uint32)), lo
        for i in range(0, nq, 97):
            ro, do, eo = o.search(qs[i], k, with_evals=True)
            assert c[i] == k and r[i].tolist() == ro.tolist() and d[i].tobytes() == do.tobytes() and int(ev[i]) == eo - 1, i

    check(rows)
    rows2 = rows.copy()
    hot = np.unique(links[9])[:8]                                            # the entry point's neighbours: read by every traversal
    for j in hot:
        rows2[j] = -rows[j] if metric != "l2sq" else rows[j] * np.float32(1.5)
        idx.update(int(j), rows2[j])
    check(rows2)
