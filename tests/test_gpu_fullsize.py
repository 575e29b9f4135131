"""BASELINE.json configs[3] and configs[4] at their stated sizes, inside `pytest -m gpu` (not only in bench.py):

  configs[3]  HNSW M=16 / MaxM0=32 / efConstruction=200 over 1M x 768, the graph built on the device (qv_graph_build),
              efSearch=128: 16 queries equal the CPU oracle's HNSW.Search walking the EXPORTED graph (rows, float32 bits,
              evaluation counts), and recall@10 against the exact top-10 on a corpus with neighbourhood structure
  configs[4]  flat cosine 10M x 768 as EIGHT row shards of 1.25M through qv_sharded_* (co-located on the one GPU of the test
              box, point-to-point exchange): the merged top-10 equals a single 10M-row index, and — for the query the
              single-index test checks against the complete CPU oracle — the oracle's top-10 itself
"""
import os

import numpy as np
import pytest

from tests import _oracle as O

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("max_level,min_recall", [(1, 0.99), (16, None)])
def test_config3_hnsw_1Mx768_built_on_device_equals_the_oracle_walk(max_level, min_recall):
    """max_level=1: one connected M=16/MaxM0=32 level-0 graph — recall@10 >= 0.99 at efSearch=128 on unit vectors of a
    16-dimensional subspace of R^768.  max_level=16 (the reference's default): its connectNode quirk (hnsw.go:463-467) makes
    islands and many queries end in the brute-force top-up (hnsw.go:676-710); reproduced, and still identical to the oracle."""
    from tests.bench.bench_hnsw_build import run
    e = run(rows=1_000_000, dim=768, metric="cosine", m=16, efc=200, max_level=max_level, efs=(128,), nq=1024, k=10, cpu_queries=16,
            corpus_seed=20260424, query_seed=20260425, intrinsic_dim=16)
    cpu = e["cpu_traversal_same_graph"]["by_ef"][0]
    assert cpu["ef_search"] == 128 and cpu["queries"] == 16
    assert cpu["identical_to_device"] is True                      # rows, float32 bits, evaluation counts (or the exact top-up)
    s = e["search"][0]
    assert s["ef_search"] == 128
    if min_recall is not None:
        assert s["underfilled_queries"] == 0
        assert s["search_complete"]["recall_at_10_vs_exact"] >= min_recall, s
    else:
        assert s["search_complete"]["recall_at_10_vs_exact"] > 0.0
    assert e["build"]["seconds"] < 120


@pytest.mark.timeout(900)
def test_config4_flat_cosine_10Mx768_eight_shards_equal_one_index_and_the_oracle():
    import quiver_amd
    from quiver_amd import DeviceIndex, ShardedIndex
    from tests._par import exact_topk_synthetic
    n, dim, seed, k, G = 10_000_000, 768, 20260424, 10, 8
    sh = ShardedIndex(dim, "cosine", devices=[0] * G, peer_copy=True)
    sh.reserve(n)
    sh.add_synthetic(seed, 0, n)                                   # shard g = generator rows [g*n/8, (g+1)*n/8): SURVEY.md 8e's contiguous blocks
    assert sh.size() == n and [sh.shard_info(g)["rows"] for g in range(G)] == [n // G] * G
    span = quiver_amd.lib().qv_sharded_span(G)
    bounds = [g * n // G for g in range(G + 1)]

    def to_corpus_row(x):
        return bounds[int(x) // span] + int(x) % span              # global id -> generator row

    qs = O.gen_rows(20260425, 4, 4, dim)                           # queries 4..7; query 7 is the one checked against the full oracle
    r, d, c = sh.search(qs, k)
    got = np.array([[to_corpus_row(x) for x in row] for row in r], dtype=np.uint32)
    one = DeviceIndex(dim, "cosine")
    one.reserve(n)
    for s in range(0, n, 2_000_000):
        one.add_synthetic(seed, s, 2_000_000)
    r1, d1, _ = one.search(qs, k)
    assert (c == k).all() and np.array_equal(got, r1) and np.array_equal(_bits(d), _bits(d1))
    # one query at a time (the single-query scan kernel on every shard) gives the same lists
    for i in range(4):
        ri, di, _ = sh.search(qs[i], k)
        assert np.array_equal(ri[0], r[i]) and np.array_equal(_bits(di[0]), _bits(d[i]))
    # a ranking deeper than the fused top-k (k = 1000 > 64): the radix-sort merge over 8 sorted runs
    rk, dk, ck = sh.search(qs[3], 1000)
    r1k, d1k, _ = one.search(qs[3], 1000)
    assert ck[0] == 1000 and np.array_equal(np.array([to_corpus_row(x) for x in rk[0]], np.uint32), r1k[0]) and np.array_equal(_bits(dk[0]), _bits(d1k[0]))
    one.close()
    # the complete CPU oracle for query 7 (the same check test_gpu_flat.py runs on the single index)
    er, ed = exact_topk_synthetic(0, seed, n, dim, qs[3], k, chunk=100_000, workers=min(32, os.cpu_count() or 8))
    assert np.array_equal(got[3], er) and np.array_equal(_bits(d[3]), _bits(ed))
    sh.close()
