"""The committed golden fixture must still be what the oracle produces (guards the
oracle against accidental change; CPU-only)."""
import os

import numpy as np

from tests import _oracle as O


def test_flat_10kx128_fixture_matches_oracle():
    g = np.load(os.path.join(O.ROOT, "tests", "golden", "flat_10kx128_cosine.npz"))
    rows = O.gen_rows(int(g["corpus_seed"]), 0, 10000, 128)
    qs = O.gen_rows(int(g["query_seed"]), 0, g["rows"].shape[0], 128)
    for i in range(0, g["rows"].shape[0], 5):
        r, d = O.exact_search(0, rows, qs[i], 10)
        assert np.array_equal(r, g["rows"][i])
        assert np.array_equal(d.view(np.uint32), g["dist"][i].view(np.uint32))


def test_hnsw_3kx32_fixture_matches_oracle():
    """graph shape (levels, links, entry) and Search output of the committed HNSW fixture"""
    g = np.load(os.path.join(O.ROOT, "tests", "golden", "hnsw_3kx32_cosine.npz"))
    n, dim = int(g["n"]), int(g["dim"])
    rows = O.gen_rows(int(g["corpus_seed"]), 0, n, dim)
    h = O.HNSW(0, dim, M=int(g["M"]), efConstruction=int(g["efConstruction"]), efSearch=int(g["efSearch"]), maxLevel=int(g["maxLevel"]), seed=int(g["seed"]))
    for r in rows:
        h.insert(r)
    assert h.entry_point() == (int(g["entry"]), int(g["cur_level"]))
    for node in range(0, n, 7):
        assert h.node_level(node) == int(g["levels"][node])
        l = h.links(node, 0)
        assert l.size == int(g["l0_deg"][node]) and np.array_equal(l, g["l0_links"][node, :l.size])
        for lv in range(1, int(g["levels"][node]) + 1):
            blk = g["up_links"][int(g["up_off"][node]) + lv - 1]
            l = h.links(node, lv)
            assert l.size == int(blk[0]) and np.array_equal(l, blk[1:1 + l.size])
    qs = O.gen_rows(int(g["query_seed"]), 0, g["rows"].shape[0], dim)
    for i in range(g["rows"].shape[0]):
        r, d, e = h.search(qs[i], 10, with_evals=True)
        assert np.array_equal(r, g["rows"][i][:r.size]) and np.array_equal(d.view(np.uint32), g["dist"][i][:d.size].view(np.uint32))
        assert e == int(g["evals"][i])
