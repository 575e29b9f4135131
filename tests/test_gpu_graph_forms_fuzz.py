"""Differential fuzzing of the two forms of the device traversal (qv_hnsw.hip): seeded random graphs — device-BUILT, multi-level,
over vectors with ties, duplicates, zero vectors and wild dynamic range — searched once as a batch of 300 (a wave per query, one
chain per row) and in batches of <= 64 (the latency form: a workgroup per query, partial chains + certificate, batched
admissions, adjacency prefetch), and by the CPU oracle's HNSW.Search (pkg/hnsw/hnsw.go:602-713 restated) on the exported graph.
Rows, float32 bits, counts and evaluation counts must agree everywhere; ties send queries through the exact-heap kernel (which
has a latency form of its own)."""
import numpy as np
import pytest

import quiver_amd
from quiver_amd.device_index import DeviceGraph, random_levels
from tests import _oracle as O

pytestmark = pytest.mark.gpu

LAT_METRICS = ["cosine", "l2", "dot", "l1", "l2sq_f64"]           # SplitOK: the latency form applies
OTHER_METRICS = ["cosine_f32", "l2sq", "dot_f32", "l2_f32"]        # always a wave per query


def _vectors(rng, n, dim, style):
    if style == 0:                                   # few distinct values per component: exact ties everywhere
        x = rng.choice(np.array([-2.0, -1.0, -0.5, 0.0, 0.0, 0.5, 1.0, 3.0], np.float32), size=(n, dim))
    elif style == 1:                                 # wide dynamic range: cancellation in every accumulate
        x = (rng.standard_normal((n, dim)) * np.exp2(rng.integers(-12, 12, size=(n, dim)))).astype(np.float32)
    elif style == 2:                                 # clusters of near-duplicates and exact duplicates
        c = rng.standard_normal((max(n // 40, 2), dim)).astype(np.float32)
        x = c[rng.integers(0, c.shape[0], n)] + (rng.standard_normal((n, dim)) * 1e-3).astype(np.float32)
        x[rng.integers(0, n, n // 10)] = x[rng.integers(0, n, n // 10)]
    else:
        x = rng.standard_normal((n, dim)).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32)


@pytest.mark.parametrize("seed", range(64))
def test_random_graphs_both_forms_and_the_oracle(seed):
    rng = np.random.default_rng(8800 + seed)
    metric = (LAT_METRICS * 3 + OTHER_METRICS)[seed % 20 % 19] if seed % 20 < 19 else "cosine"
    dim = int(rng.choice([32, 64, 96, 128, 160, 256, 384, 768, 1536]))
    n = int(rng.integers(400, 5000))
    m = int(rng.choice([4, 8, 16, 32]))
    efc = int(rng.choice([16, 40, 100, 200]))
    ef = int(rng.choice([1, 10, 33, 64, 128, 200, 300, 512]))
    k = int(rng.choice([1, 5, 10, 37]))
    max_level = int(rng.choice([1, 1, 3, 16]))
    style = int(rng.integers(0, 4))
    rows = _vectors(rng, n, dim, style)
    idx = quiver_amd.DeviceIndex(dim, metric, rowmajor=True)
    idx.add(rows)
    levels = random_levels(n, max_level, seed + 1)
    g = DeviceGraph.build(idx, levels, m=m, max_m0=2 * m, ef_construction=efc, batch_max=int(rng.choice([1, 64, 4096])))
    info = g.info()
    lv, l0_deg, l0_links, up_off, up_links = g.export()
    o = O.HNSW(quiver_amd.metric_id(metric), dim, M=m, maxM0=2 * m, efConstruction=efc, efSearch=ef, maxLevel=max_level, seed=seed + 1)
    o.load_graph(rows, lv, 2 * m, m, l0_deg, l0_links, up_off, up_links, info["entry"], info["cur_level"])
    qs = np.concatenate([_vectors(rng, 150, dim, style), rows[rng.integers(0, n, 150)]])
    r, d, c, ev = g.search(qs, k, ef, with_evals=True)                        # 300 queries: a wave per query
    for lo in range(0, 300, 60):                                             # <= 64 queries: the latency form (for its metrics)
        r2, d2, c2, ev2 = g.search(qs[lo:lo + 60], k, ef, with_evals=True)
        assert np.array_equal(c2, c[lo:lo + 60]), (lo, metric, dim)
        assert np.array_equal(ev2, ev[lo:lo + 60]), (lo, metric, dim)
        for i in range(60):
            assert r2[i, :c2[i]].tolist() == r[lo + i, :c2[i]].tolist(), (lo, i, metric, dim)
            assert d2[i, :c2[i]].tobytes() == d[lo + i, :c2[i]].tobytes(), (lo, i, metric, dim)
    r1, d1, c1, ev1 = g.search(qs[7:8], k, ef, with_evals=True)               # and alone
    assert int(c1[0]) == int(c[7]) and r1[0, :c1[0]].tolist() == r[7, :c1[0]].tolist() and d1[0, :c1[0]].tobytes() == d[7, :c1[0]].tobytes()
    er, ed, _ = idx.search(qs, k)                                            # what HNSW.Search's top-up returns (hnsw.go:676-710)
    for i in range(0, 300, 13):
        ro, do, eo = o.search(qs[i], k, with_evals=True)
        ci = int(c[i])
        assert ci <= k
        if ci == k or ci == n:                       # filled by the graph search alone: rows, bits and evaluation counts
            assert r[i, :ci].tolist() == ro[:ci].tolist(), (i, metric, dim, style)
            assert d[i, :ci].tobytes() == do[:ci].tobytes(), (i, metric, dim, style)
            assert int(ev[i]) == eo - 1, (i, metric, dim, style)
        else:                                        # under-filled (the level quirk's islands): the reference answers with the exact top-k
            assert er[i].tolist() == ro.tolist() and ed[i].tobytes() == do.tobytes(), (i, metric, dim, style)
    g.close(); idx.close()
