"""Concurrent single-query callers — the only traffic the reference's unchanged Go host produces: Collection.Search takes a
read lock and calls Index.Search(q, k) once per request (pkg/core/collection.go:647); DB.BatchSearch reaches a batch entry only
through a type assertion on the reference's own wrapper (pkg/core/db.go:726-727) and otherwise fans out one goroutine per query
(:805-828); HNSW.Search runs under a read lock, a goroutine per query (pkg/hnsw/hnsw.go:602-606, adapter.go:253-279).
libqv lets such callers share device passes (quiver_amd/csrc/qv_coalesce.h).  Here: whoever shares a pass with whom, every caller
gets the oracle's rows and bits."""
import threading

import numpy as np
import pytest

import quiver_amd
from tests import _callers, _oracle as O

pytestmark = pytest.mark.gpu


def _check_flat(metric_name, corpus, queries, res, k):
    mid = quiver_amd.metric_id(metric_name)
    for i in range(queries.shape[0]):
        if res["count"][i] == 0xFFFFFFFD:
            continue
        er, ed = O.exact_search(mid, corpus, queries[i], k)
        assert res["count"][i] == min(k, corpus.shape[0]), i
        assert np.array_equal(res["rows"][i, :len(er)], er), (i, res["rows"][i], er)
        assert np.array_equal(res["dist"][i, :len(er)].view(np.uint32), ed.view(np.uint32)), i


def test_64_threads_one_query_per_call_on_300k_x_768_equal_the_oracle():
    n, dim, k = 300_000, 768, 10
    idx = quiver_amd.DeviceIndex(dim, "cosine")
    idx.add_synthetic(20260424, 0, n)
    corpus = O.gen_rows(20260424, 0, n, dim)
    qs = O.gen_rows(20260425, 0, 96, dim)
    res = _callers.run("index", idx.handle, qs, k, threads=64, seconds=30.0, max_calls_per_thread=12)
    assert res["rc"] == 0, res["error"]
    assert res["errors"] == 0 and res["mismatches"] == 0 and res["calls"] == 64 * 12
    _check_flat("cosine", corpus, qs, res, k)
    st = _callers.coalesce_stats("index", idx.handle)
    assert st["solo"] + st["led"] + st["rode"] == res["calls"]
    assert st["rode"] > 0 and st["group_queries"] > st["groups"] > 0          # passes were shared
    # a lone caller afterwards runs solo, at once
    before = st["solo"]
    r, d, c = idx.search(qs[:1], k)
    assert _callers.coalesce_stats("index", idx.handle)["solo"] == before + 1
    er, ed = O.exact_search(0, corpus, qs[0], k)
    assert np.array_equal(r[0], er) and np.array_equal(d[0].view(np.uint32), ed.view(np.uint32))


@pytest.mark.parametrize("metric,dim,n", [("l2", 128, 120_000), ("dot", 96, 150_000), ("l1", 64, 200_000), ("cosine_f32", 256, 60_000)])
def test_callers_with_different_k_and_query_counts_share_passes(metric, dim, n):
    """groups carry the largest k of their members; every member gets the prefix it asked for (counts, padding included)"""
    idx = quiver_amd.DeviceIndex(dim, metric)
    corpus = O.gen_rows(99, 0, n, dim)
    idx.add(corpus)
    idx.remove(np.arange(0, n, 7, dtype=np.uint32))                             # tombstones travel through every path
    live = np.ones(n, bool); live[::7] = False
    mid = quiver_amd.metric_id(metric)
    qs = O.gen_rows(100, 0, 48, dim)
    out = {}
    errs = []

    def caller(t):
        try:
            rng = np.random.default_rng(t)
            for it in range(6):
                nq = int(rng.integers(1, 4)); k = int(rng.choice([1, 3, 10, 37, 64]))
                sel = rng.integers(0, qs.shape[0], nq)
                r, d, c = idx.search(qs[sel], k)
                out[(t, it)] = (sel, k, r, d, c)
        except Exception as e:                                                  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=caller, args=(t,)) for t in range(24)]
    [x.start() for x in th]; [x.join() for x in th]
    assert not errs, errs
    cache = {}
    for (t, it), (sel, k, r, d, c) in out.items():
        for j, qi in enumerate(sel):
            if (qi, k) not in cache:
                cache[(qi, k)] = O.exact_search(mid, corpus, qs[qi], k, alive=live)
            er, ed = cache[(qi, k)]
            assert c[j] == k and np.array_equal(r[j], er) and np.array_equal(d[j].view(np.uint32), ed.view(np.uint32)), (t, it, j)
    st = _callers.coalesce_stats("index", idx.handle)
    assert st["solo"] + st["led"] + st["rode"] == 24 * 6


def test_errors_keep_the_reference_order_and_wording_under_concurrency():
    """exact.go:96-106: empty -> ok with no results; k <= 0 -> "k must be positive"; failing callers do not disturb the others"""
    idx = quiver_amd.DeviceIndex(64, "cosine")
    corpus = O.gen_rows(5, 0, 200_000, 64)
    idx.add(corpus)
    qs = O.gen_rows(6, 0, 8, 64)
    bad, good = [], []

    def bad_caller():
        for _ in range(20):
            try:
                idx.search(qs[:1], 0)
            except quiver_amd.QvError as e:
                bad.append(str(e))

    def good_caller(t):
        for _ in range(10):
            good.append((t, idx.search(qs[t:t + 1], 5)))

    th = [threading.Thread(target=bad_caller)] + [threading.Thread(target=good_caller, args=(t,)) for t in range(8)]
    [x.start() for x in th]; [x.join() for x in th]
    assert bad == ["k must be positive"] * 20
    for t, (r, d, c) in good:
        er, ed = O.exact_search(0, corpus, qs[t], 5)
        assert np.array_equal(r[0], er) and np.array_equal(d[0].view(np.uint32), ed.view(np.uint32))


def _knn_graph(rows, metric, m):
    n = rows.shape[0]
    idx = quiver_amd.DeviceIndex(rows.shape[1], metric, rowmajor=True)
    idx.add(rows)
    nbr, _, _ = idx.search(rows, m + 1)
    links = np.zeros((n, m), np.uint32); deg = np.zeros(n, np.uint32)
    for i in range(n):
        l = [int(x) for x in nbr[i] if int(x) != i][:m]
        deg[i] = len(l); links[i, :len(l)] = l
    return idx, deg, links


def test_64_threads_one_query_per_call_on_a_100k_node_graph_equal_the_oracle_walk():
    from quiver_amd.device_index import DeviceGraph, random_levels
    n, dim, k, ef = 100_000, 64, 10, 64
    rows = O.gen_rows(31, 0, n, dim)
    idx = quiver_amd.DeviceIndex(dim, "cosine", rowmajor=True)
    idx.add(rows)
    levels = np.zeros(n, np.int8)                                               # MaxLevel = 1: one connected level-0 graph
    g = DeviceGraph.build(idx, levels, m=8, max_m0=16, ef_construction=64)
    lv, l0d, l0l, uo, ul = g.export()
    info = g.info()
    o = O.HNSW(0, dim, M=8, maxM0=16, efConstruction=64, efSearch=ef, maxLevel=1, seed=3)
    o.load_graph(rows, lv, 16, 8, l0d, l0l, uo, ul, info["entry"], info["cur_level"])
    qs = O.gen_rows(32, 0, 128, dim)
    res = _callers.run("graph", g.handle, qs, k, threads=64, seconds=60.0, max_calls_per_thread=16, ef=ef)
    assert res["rc"] == 0, res["error"]
    assert res["errors"] == 0 and res["mismatches"] == 0 and res["calls"] == 64 * 16
    for i in range(qs.shape[0]):
        c = int(res["count"][i])
        if c == 0xFFFFFFFD:
            continue
        ro, do = o.search(qs[i], k)
        assert c <= k
        assert res["rows"][i, :c].tolist() == ro[:c].tolist(), i
        assert res["dist"][i, :c].tobytes() == do[:c].tobytes(), i
    st = _callers.coalesce_stats("graph", g.handle)
    assert st["solo"] + st["led"] + st["rode"] == res["calls"]
    # the same answers from one batch call
    r, d, c = g.search(qs, k, ef)
    for i in range(qs.shape[0]):
        if res["count"][i] != 0xFFFFFFFD:
            assert int(c[i]) == int(res["count"][i]) and r[i, :c[i]].tolist() == res["rows"][i, :c[i]].tolist()


def test_concurrent_graph_callers_with_duplicate_rows_take_the_exact_heap_pass_each_in_their_own_context():
    """equal distances everywhere: every query is flagged by the wave kernel and redone by the exact-heap kernel, which keeps a
    visited bitmap per slot — contexts must not share them"""
    from quiver_amd.device_index import DeviceGraph
    rows = O.gen_rows(5150, 0, 1200, 48)
    rows[600:] = rows[:600]
    idx, deg, links = _knn_graph(rows, "cosine", 16)
    g = DeviceGraph(idx, np.zeros(1200, np.int8), deg, links, entry=0)
    o = O.HNSW(0, 48, M=8, maxM0=16, efSearch=64, maxLevel=1, seed=1)
    o.load_flat(rows, deg, links, 0)
    qs = O.gen_rows(5151, 0, 32, 48)
    res = _callers.run("graph", g.handle, qs, 10, threads=16, seconds=60.0, max_calls_per_thread=12, ef=64)
    assert res["rc"] == 0 and res["errors"] == 0 and res["mismatches"] == 0, res["error"]
    for i in range(32):
        ro, do = o.search(qs[i], 10)
        n = min(int(res["count"][i]), 10)
        assert res["rows"][i, :n].tolist() == ro[:n].tolist() and res["dist"][i, :n].tobytes() == do[:n].tobytes()


def test_shared_traversal_batches_in_which_only_some_queries_need_the_exact_heap_pass():
    """one row in eight appears twice: some queries meet equal distances (flagged by the wave kernel, redone by the exact-heap kernel)
    and most do not — in a shared batch the latter get their results after the first pass, the former after the second; all equal
    the oracle's walk"""
    from quiver_amd.device_index import DeviceGraph
    import torch
    qs = O.gen_rows(910, 0, 192, 64)
    dq = torch.from_numpy(qs).cuda()
    dr = torch.empty((192, 10), dtype=torch.int32, device="cuda"); dd = torch.empty((192, 10), dtype=torch.float32, device="cuda")
    dc = torch.empty(192, dtype=torch.int32, device="cuda")
    for n_dup in (8, 32, 2, 128):                                               # enough duplicated vectors that SOME queries meet a tie, not all
        rows = O.gen_rows(909, 0, 4000, 64)
        rows[4000 - n_dup:] = rows[:n_dup]
        idx, deg, links = _knn_graph(rows, "l2", 16)
        g = DeviceGraph(idx, np.zeros(4000, np.int8), deg, links, entry=11)
        # the device form says which queries are flagged on the way (count 0xFFFFFFFE)
        g.search_device(dq.data_ptr(), 192, 10, 48, dr.data_ptr(), dd.data_ptr(), dc.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        flagged = int((dc.cpu().numpy().view(np.uint32) == 0xFFFFFFFE).sum())
        if 0 < flagged < 192:
            break
    assert 0 < flagged < 192, flagged
    o = O.HNSW(quiver_amd.metric_id("l2"), 64, M=8, maxM0=16, efSearch=48, maxLevel=1, seed=1)
    o.load_flat(rows, deg, links, 11)
    res = _callers.run("graph", g.handle, qs, 10, threads=48, seconds=60.0, max_calls_per_thread=16, ef=48)
    assert res["rc"] == 0 and res["errors"] == 0 and res["mismatches"] == 0, res["error"]
    for i in range(192):
        ro, do = o.search(qs[i], 10)
        n = min(int(res["count"][i]), 10)
        assert n == min(len(ro), 10)
        assert res["rows"][i, :n].tolist() == ro[:n].tolist() and res["dist"][i, :n].tobytes() == do[:n].tobytes(), i


def test_concurrent_callers_on_a_sharded_handle_share_passes_and_equal_the_oracle():
    n, dim, k = 240_000, 128, 10
    sh = quiver_amd.ShardedIndex(dim, "cosine", devices=[0, 0, 0], peer_copy=True)
    sh.add_synthetic(777, 0, n)
    corpus = O.gen_rows(777, 0, n, dim)
    qs = O.gen_rows(778, 0, 64, dim)
    res = _callers.run("sharded", sh.handle, qs, k, threads=32, seconds=30.0, max_calls_per_thread=8)
    assert res["rc"] == 0 and res["errors"] == 0 and res["mismatches"] == 0, res["error"]
    base = [sh.shard_info(g)["base"] for g in range(3)]
    for i in range(qs.shape[0]):
        if res["count"][i] == 0xFFFFFFFD:
            continue
        er, ed = O.exact_search(0, corpus, qs[i], k)
        # global ids: shard base + local row; shard g holds the contiguous block [g n / 3, (g + 1) n / 3)
        got = [int(x) for x in res["rows"][i]]
        loc = []
        for x in got:
            gi = max(j for j in range(3) if base[j] <= x)
            loc.append(x - base[gi] + gi * (n // 3))
        assert loc == er.tolist(), i
        assert np.array_equal(res["dist"][i].view(np.uint32), ed.view(np.uint32)), i


def test_concurrent_batches_on_the_selection_path_equal_the_exact_scan():
    """eight threads, each sending batches of 16 to 1000 results per query to ONE index at the same time (every call its own context, stream and
    workspace: the candidate lists, the pairs of the tile pass, the selection's keys); every batch equals the exact scan of the same index"""
    n, dim = 200_000, 128
    idx = quiver_amd.DeviceIndex(dim, "cosine")
    idx.add_synthetic(20260424, 0, n)
    shapes = [(64, 16), (256, 100), (40, 1000), (300, 300), (9, 64), (128, 40), (256, 1000), (100, 129)]
    qsets = [O.gen_rows(20260500 + i, 0, nq, dim) for i, (nq, _) in enumerate(shapes)]
    got = [None] * len(shapes); errs = []

    def work(i):
        try:
            for _ in range(4):
                got[i] = idx.search(qsets[i], shapes[i][1], batched=True)
        except Exception as ex:                                            # noqa: BLE001
            errs.append((i, repr(ex)))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(shapes))]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not errs, errs
    idx.set_filter("off")
    for i, (nq, k) in enumerate(shapes):
        want = idx.search(qsets[i], k)
        assert np.array_equal(got[i][0], want[0]) and got[i][1].tobytes() == want[1].tobytes() and np.array_equal(got[i][2], want[2]), shapes[i]
    idx.close()
