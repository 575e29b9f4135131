"""Device-resident HNSW construction (qv_graph_create_empty / qv_graph_insert / qv_graph_build / qv_graph_export) against
the CPU oracle's restatement of hnsw.HNSW.Insert (pkg/hnsw/hnsw.go:266-468).

batch_max = 1 must reproduce the reference's SEQUENTIAL graph (the oracle's Insert loop, and the committed fixture
tests/golden/hnsw_3kx32_cosine.npz); larger batches must reproduce qvo_hnsw_insert_batch — the same snapshot-search /
in-order-link semantics with the reference's own re-scoring prune — link for link, in order."""
import os

import numpy as np
import pytest

import quiver_amd
from quiver_amd.device_index import DeviceGraph, graph_batch_size, random_levels
from tests import _oracle as O
from tests.test_oracle_hnsw import batch_schedule

pytestmark = pytest.mark.gpu


def _device_build(rows, metric, levels, m, max_m0, efc, batch_max, ramp_div):
    idx = quiver_amd.DeviceIndex(rows.shape[1], metric, rowmajor=True)
    idx.add(rows)
    g = DeviceGraph.build(idx, levels, m=m, max_m0=max_m0, ef_construction=efc, batch_max=batch_max, ramp_div=ramp_div)
    return idx, g


def _oracle_build(rows, mid, m, max_m0, efc, max_level, seed, batch_max, ramp_div):
    h = O.HNSW(mid, rows.shape[1], M=m, maxM0=max_m0, efConstruction=efc, maxLevel=max_level, seed=seed)
    done = 0
    for b in batch_schedule(rows.shape[0], batch_max, ramp_div):
        h.insert_batch(rows[done:done + b]); done += b
    return h


def _assert_same_graph(g: DeviceGraph, h: O.HNSW, m, max_m0):
    levels, l0_deg, l0_links, up_off, up_links = g.export()
    info = g.info()
    n = h.nodes()
    assert info["n_nodes"] == n
    assert (info["entry"], info["cur_level"]) == h.entry_point()
    want = h.export_flat(max_m0, m)
    assert np.array_equal(levels, want[0])
    bad = [i for i in range(n) if l0_deg[i] != want[1][i] or l0_links[i, :l0_deg[i]].tolist() != want[2][i, :want[1][i]].tolist()]
    assert not bad, "level-0 lists differ at nodes %s (first: got %s want %s)" % (
        bad[:5], l0_links[bad[0], :l0_deg[bad[0]]].tolist(), want[2][bad[0], :want[1][bad[0]]].tolist())
    assert np.array_equal(up_off[levels >= 1], want[3][levels >= 1])
    assert up_links.shape == want[4].shape
    for b in range(up_links.shape[0]):
        d = int(up_links[b, 0])
        assert d == int(want[4][b, 0]) and up_links[b, 1:1 + d].tolist() == want[4][b, 1:1 + d].tolist(), b


def test_schedule_rule_matches_the_test_helper():
    for bm, rd in ((1, 0), (64, 8), (4096, 16), (100, 0)):
        done = 0
        for b in batch_schedule(5000, bm, rd):
            assert b == min(graph_batch_size(done, bm, rd), 5000 - done)
            done += b


def test_level_law_is_the_oracles():
    h = O.HNSW(0, 4, seed=77, maxLevel=16)
    assert [h.random_level() for _ in range(5000)] == random_levels(5000, 16, 77).tolist()


def test_sequential_build_equals_committed_fixture():
    """batch_max = 1: the reference's graph, node for node — tests/golden/hnsw_3kx32_cosine.npz was built by the oracle's Insert"""
    f = np.load(os.path.join(O.ROOT, "tests", "golden", "hnsw_3kx32_cosine.npz"))
    n, dim, m = int(f["n"]), int(f["dim"]), int(f["M"])
    rows = O.gen_rows(int(f["corpus_seed"]), 0, n, dim)
    levels = random_levels(n, int(f["maxLevel"]), int(f["seed"]))
    assert np.array_equal(levels, f["levels"])
    idx, g = _device_build(rows, "cosine", levels, m, 2 * m, int(f["efConstruction"]), 1, 0)
    lv, l0_deg, l0_links, up_off, up_links = g.export()
    info = g.info()
    assert (info["entry"], info["cur_level"]) == (int(f["entry"]), int(f["cur_level"]))
    assert np.array_equal(l0_deg, f["l0_deg"])
    for i in range(n):
        assert l0_links[i, :l0_deg[i]].tolist() == f["l0_links"][i, :l0_deg[i]].tolist(), i
    assert np.array_equal(up_links, f["up_links"][:up_links.shape[0]])
    # and the traversal of the device-built graph gives the fixture's results
    qs = O.gen_rows(int(f["query_seed"]), 0, f["rows"].shape[0], dim)
    r, d, c, ev = g.search(qs, 10, int(f["efSearch"]), with_evals=True)
    for i in range(qs.shape[0]):
        if c[i] == 10:
            assert np.array_equal(r[i], f["rows"][i]) and np.array_equal(d[i].view(np.uint32), f["dist"][i].view(np.uint32))


@pytest.mark.parametrize("metric,dim,n,m,efc,max_level,batch_max,ramp_div", [
    ("cosine", 64, 1500, 8, 40, 16, 1, 0),          # sequential
    ("cosine", 64, 4000, 8, 60, 16, 64, 8),         # ramped batches
    ("l2", 32, 3000, 6, 30, 4, 256, 4),
    ("dot", 48, 2000, 16, 100, 16, 128, 16),        # M = 16 / MaxM0 = 32: the reference's defaults
    ("cosine_f32", 40, 2000, 8, 50, 8, 100, 0),     # no ramp: the second batch already has 100 nodes
    ("l2_f32", 24, 2500, 4, 20, 2, 512, 2),
    ("l2sq", 16, 1200, 8, 300, 16, 32, 8),          # efConstruction > 256: 8 list registers per lane
    ("cosine", 768, 1200, 16, 200, 16, 96, 8),      # the headline shape, small n
    ("l2", 30, 1500, 8, 40, 16, 64, 8),             # dim % 4 != 0: rows come from the tile layout, not by LDS-DMA
    ("cosine", 36, 1500, 32, 80, 16, 64, 8),        # M = 32 / MaxM0 = 64: full-width adjacency lists (the prune fills all 64 lanes)
    ("dot", 20, 1200, 16, 10, 16, 64, 8),           # efConstruction < MaxM0: the search returns fewer than a list holds
    ("l1", 24, 1200, 8, 64, 16, 64, 8),             # efConstruction = 64: the list registers' first spare notch
])
def test_batched_build_equals_oracle_batch_semantics(metric, dim, n, m, efc, max_level, batch_max, ramp_div):
    mid = quiver_amd.metric_id(metric)
    rows = O.gen_rows(31337, 0, n, dim)
    levels = random_levels(n, max_level, 5)
    idx, g = _device_build(rows, metric, levels, m, 2 * m, efc, batch_max, ramp_div)
    h = _oracle_build(rows, mid, m, 2 * m, efc, max_level, 5, batch_max, ramp_div)
    _assert_same_graph(g, h, m, 2 * m)
    # searching the built graph: device == oracle on the same graph
    qs = O.gen_rows(31338, 0, 48, dim)
    h.set_ef_search(64)
    r, d, c, ev = g.search(qs, 10, 64, with_evals=True)
    for i in range(qs.shape[0]):
        ro, do, eo = h.search(qs[i], 10, with_evals=True)
        if c[i] == 10:
            assert r[i].tolist() == ro.tolist() and d[i].tobytes() == do.tobytes() and int(ev[i]) == eo - 1, i


def test_duplicate_vectors_go_through_the_exact_heap_kernel_during_build():
    """every vector three times: equal distances in every construction search (heap order decides) and distance-0 ties in
    every prune (node index decides, hnsw.go:589-594)"""
    base = O.gen_rows(99, 0, 500, 24)
    rows = np.concatenate([base, base, base])
    levels = random_levels(rows.shape[0], 6, 3)
    idx, g = _device_build(rows, "l2", levels, 6, 12, 40, 32, 8)
    h = _oracle_build(rows, 1, 6, 12, 40, 6, 3, 32, 8)
    _assert_same_graph(g, h, 6, 12)
    assert g.stats()["build_redo"] > 0


def test_insert_in_two_calls_equals_one_call_and_graph_stays_searchable():
    n, dim = 3000, 32
    rows = O.gen_rows(8, 0, n, dim)
    levels = random_levels(n, 16, 21)
    idx = quiver_amd.DeviceIndex(dim, "cosine", rowmajor=True)
    idx.add(rows[:1800])
    g = DeviceGraph.empty(idx, 100, m=8, max_m0=16, ef_construction=50)       # capacity grows
    g.insert(0, levels[:1800], 64, 8)
    r0, d0, c0 = g.search(rows[:4], 3, 32)
    assert (r0[:, 0] < 1800).all()
    idx.add(rows[1800:])
    g.insert(1800, levels[1800:], 64, 8)
    h = O.HNSW(0, dim, M=8, maxM0=16, efConstruction=50, maxLevel=16, seed=21)
    done = 0
    for seg in (1800, n):
        while done < seg:
            b = min(graph_batch_size(done, 64, 8), seg - done)
            h.insert_batch(rows[done:done + b]); done += b
    _assert_same_graph(g, h, 8, 16)
    with pytest.raises(quiver_amd.QvError):
        g.insert(5, levels[:3], 64, 8)                                        # nodes are appended only


def test_build_needs_the_row_major_copy():
    idx = quiver_amd.DeviceIndex(16, "cosine")
    idx.add(O.gen_rows(1, 0, 10, 16))
    with pytest.raises(quiver_amd.QvError) as e:
        DeviceGraph.build(idx, np.zeros(10, np.int8))
    assert e.value.code == quiver_amd._lib.QV_ERR_UNSUPPORTED


def test_recall_of_the_built_graph():
    """a sanity check that the batched graph is a usable index (the reference pins HNSW by properties only): on
    low-dimensional Gaussian data the device-built graph returns the exact top-10 of stored vectors.  (Clustered data is
    no use here: the reference selects plain nearest-M neighbours, hnsw.go:583-599, which leaves clusters unconnected.)"""
    rng = np.random.default_rng(5)
    rows = rng.standard_normal((6000, 16)).astype(np.float32)
    levels = np.zeros(6000, np.int8)                                          # MaxLevel = 1: no self-link quirk in play
    idx, g = _device_build(rows, "l2", levels, 16, 32, 100, 256, 16)
    qs = rows[::60]
    r, d, c = g.search(qs, 10, 100)
    er, ed, _ = idx.search(qs, 10)
    hit = sum(len(set(r[i, :c[i]].tolist()) & set(er[i].tolist())) for i in range(qs.shape[0]))
    assert hit / (10 * qs.shape[0]) > 0.97


def test_uploaded_graph_can_be_extended_after_scoring_its_links():
    """qv_graph_create (a host-built graph, no link distances) + qv_graph_make_buildable + qv_graph_insert == building it all by
    batches: the first 2000 nodes come from the oracle's build, the next 1500 are inserted on the device"""
    n0, n1, dim, m, efc = 2000, 1500, 40, 8, 50
    rows = O.gen_rows(606, 0, n0 + n1, dim)
    levels = random_levels(n0 + n1, 16, 9)
    h = O.HNSW(1, dim, M=m, maxM0=2 * m, efConstruction=efc, maxLevel=16, seed=9)
    done = 0
    for b in batch_schedule(n0, 128, 8):
        h.insert_batch(rows[done:done + b]); done += b
    flat = h.export_flat(2 * m, m)
    ep, lvl = h.entry_point()
    idx = quiver_amd.DeviceIndex(dim, "l2", rowmajor=True)
    idx.add(rows)
    g = DeviceGraph(idx, flat[0], flat[1], flat[2], entry=ep, cur_level=lvl, up_off=flat[3], up_links=flat[4] if flat[4].shape[0] else None, max_m=m)
    with pytest.raises(quiver_amd.QvError, match="qv_graph_make_buildable"):
        g.insert(n0, levels[n0:], 128, 8)
    g.make_buildable(efc)
    g.insert(n0, levels[n0:], 128, 8)
    while done < n0 + n1:
        b = min(graph_batch_size(done, 128, 8), n0 + n1 - done)
        h.insert_batch(rows[done:done + b]); done += b
    _assert_same_graph(g, h, m, 2 * m)
