"""Pins the CPU oracle: every literal known-answer test of the reference
(tests/golden/ref_kats.json) must pass through oracle/qv_oracle.c, and the C
restatement must agree bit for bit with the independent numpy restatement
(oracle/oracle_np.py) on random inputs."""
import json
import os
import sys

import numpy as np
import pytest

from tests import _oracle as O

sys.path.insert(0, os.path.join(O.ROOT, "oracle"))
import oracle_np as ONP  # noqa: E402

KATS = json.load(open(os.path.join(O.ROOT, "tests", "golden", "ref_kats.json")))


@pytest.mark.parametrize("kat", KATS["distance"], ids=lambda k: k["src"])
def test_distance_kats_c(kat):
    got = O.distance(kat["metric"], kat["a"], kat["b"])
    assert abs(float(got) - kat["want"]) <= kat["tol"], (kat, got)


@pytest.mark.parametrize("kat", KATS["distance"], ids=lambda k: k["src"])
def test_distance_kats_numpy(kat):
    got = ONP.distance(kat["metric"], kat["a"], kat["b"])
    assert abs(float(got) - kat["want"]) <= kat["tol"], (kat, got)


def test_length_mismatch_is_an_error():
    for kat in KATS["distance_length_mismatch"]:
        for m in kat["metrics"]:
            with pytest.raises(ValueError):
                O.distance(m, kat["a"], kat["b"])
            with pytest.raises(ValueError):
                ONP.distance(m, kat["a"], kat["b"])


def test_unknown_metric_defaults_to_cosine():
    kat = KATS["unknown_metric_defaults_to_cosine"]
    assert float(O.distance(99, kat["a"], kat["b"])) == kat["want"]
    assert float(ONP.distance(99, kat["a"], kat["b"])) == kat["want"]


@pytest.mark.parametrize("kat", KATS["exact_search"], ids=lambda k: k["src"])
def test_exact_search_kats(kat):
    rows = np.array(kat["rows"], dtype=np.float32)
    for search in (O.exact_search, lambda m, r, q, k: ONP.exact_search(m, r, np.array(q, np.float32), k)):
        r, d = search(kat["metric"], rows, kat["query"], kat["k"])
        ids = [kat["ids"][i] for i in r]
        assert len(ids) == min(kat["k"], len(kat["ids"]))
        if kat["exact_order"]:
            assert ids[: len(kat["want_ids"])] == kat["want_ids"]
        else:
            assert set(kat["want_ids"]) <= set(ids)
        assert all(d[i] <= d[i + 1] for i in range(len(d) - 1))  # exact_test.go:197-202
        if "want_all_dist" in kat:
            assert all(abs(float(x) - kat["want_all_dist"]) <= kat["dist_tol"] for x in d)


def test_exact_errors():
    for kat in KATS["exact_errors"]:
        rows = np.array(kat["rows"], dtype=np.float32).reshape(-1, kat["dim"])
        if kat["want"] == "empty_ok":
            r, d = O.exact_search(0, rows, kat["query"], kat["k"])
            assert len(r) == 0 and len(d) == 0
        elif "insert" in kat:
            continue  # insert-time validation lives in the host layer (tests/test_host_exact.py)
        else:
            with pytest.raises(ValueError) as e:
                O.exact_search(0, rows, kat["query"], kat["k"])
            assert str(e.value) == kat["want"]


@pytest.mark.parametrize("kat", KATS["negative_rerank"], ids=lambda k: k["src"])
def test_negative_rerank_kats(kat):
    rows = np.array(kat["rows"], dtype=np.float32)
    order = sorted(range(len(kat["ids"])), key=lambda i: kat["ids"][i])
    id_rank = np.empty(len(order), dtype=np.uint32)
    id_rank[order] = np.arange(len(order), dtype=np.uint32)
    r, d = O.exact_search_negative(kat["metric"], rows, kat["query"], kat["negative"], kat["weight"], kat["k"], id_rank=id_rank)
    assert len(r) == kat["want_count"]
    assert not np.isnan(d).any()
    if kat.get("want_all_equal"):
        assert d[0] == d[1] == d[2]
        assert [kat["ids"][i] for i in r] == sorted(kat["ids"])  # ties by id (hybrid_index.go:553-556)
    if "not_first" in kat:
        assert kat["ids"][r[0]] != kat["not_first"]


# ---- C restatement == numpy restatement, bit for bit -------------------------------

@pytest.mark.parametrize("metric", range(9))
@pytest.mark.parametrize("dim", [1, 3, 7, 64, 128, 768])
def test_c_equals_numpy_bitwise(metric, dim):
    rng = np.random.default_rng(1000 * metric + dim)
    rows = rng.standard_normal((257, dim)).astype(np.float32)
    rows[5] = 0.0                      # zero vector (cosine guard, distances.go:25-27)
    rows[6] = rows[7]                  # duplicate rows (ties)
    q = rng.standard_normal(dim).astype(np.float32)
    rows[8] = q                        # identical to the query
    rows[9] = -q
    c = O.all_distances(metric, rows, q)
    n = ONP.distances(metric, q, rows)
    assert c.dtype == np.float32 and n.dtype == np.float32
    assert np.array_equal(c.view(np.uint32), n.view(np.uint32)), np.nonzero(c != n)


def test_zero_query_cosine_is_one():
    rows = np.eye(4, dtype=np.float32)
    assert np.all(O.all_distances(0, rows, np.zeros(4, np.float32)) == 1.0)
    assert np.all(O.all_distances(5, rows, np.zeros(4, np.float32)) == 1.0)


def test_exact_search_matches_numpy_with_ties_and_tombstones():
    rng = np.random.default_rng(7)
    rows = rng.integers(-2, 3, size=(500, 4)).astype(np.float32)  # many exact ties
    q = np.array([1, 0, -1, 2], np.float32)
    alive = rng.random(500) > 0.2
    for metric in range(9):
        for k in (1, 10, 64, 400, 1000):
            r1, d1 = O.exact_search(metric, rows, q, k, alive=alive)
            r2, d2 = ONP.exact_search(metric, rows, q, k, alive=alive)
            assert np.array_equal(r1, r2)
            assert np.array_equal(d1.view(np.uint32), d2.view(np.uint32))
            assert len(r1) == min(k, int(alive.sum()))


def test_generator_c_equals_numpy_and_is_unit_norm():
    a = O.gen_rows(20260424, 12345, 64, 768)
    b = ONP.gen_rows(20260424, 12345, 64, 768)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.allclose(np.linalg.norm(a.astype(np.float64), axis=1), 1.0, atol=1e-6)
    # rows are a pure function of (seed, global row): shards of one corpus agree
    c = O.gen_rows(20260424, 12345 + 10, 5, 768)
    assert np.array_equal(a[10:15], c)
    assert not np.array_equal(O.gen_rows(20260425, 12345, 4, 768), a[:4])


def test_generator_is_roughly_isotropic():
    a = O.gen_rows(1, 0, 2000, 128).astype(np.float64)
    assert abs(a.mean()) < 1e-3
    g = a.T @ a / 2000 * 128          # ~ identity
    assert np.abs(g - np.eye(128)).max() < 0.15


def test_faithful_baseline_matches_oracle():
    rows = O.gen_rows(3, 0, 300, 32)
    f = O.Faithful(0, 32)
    for i, r in enumerate(rows):
        f.insert(f"v{i}", r)
    with pytest.raises(ValueError):
        f.insert("v3", rows[3])
    q = O.gen_rows(4, 0, 1, 32)[0]
    ids, d = f.search(q, 10)
    r2, d2 = O.exact_search(0, rows, q, 10)
    assert np.array_equal(d.view(np.uint32), d2.view(np.uint32))
    assert set(ids) == {f"v{i}" for i in r2}  # no ties in this draw
