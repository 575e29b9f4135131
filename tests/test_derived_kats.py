"""Discriminating known-answer vectors (tests/golden/derived_kats.json, derived by hand from pkg/vectortypes/distances.go:12-104,
pkg/hnsw/adapter.go:105-167 and pkg/index/arrow_hnsw.go:124-132; tests/golden/make_derived_kats.py recomputes each in exact
rational arithmetic).  The reference's own tests hold 27 distance vectors of dimension <= 3 at tolerance 1e-6, and none of them
tells float64 from float32 accumulation, "subtract in float32, then widen" from "widen, then subtract", a fused from an unfused
multiply-add, or finds a missing clamp.  These do: the expected value is a float32 BIT PATTERN, and for every vector the bits a
wrong restatement would produce are listed and differ.

CPU here: the C oracle, its numpy mirror and the host qv_distance_pair (the kernels' per-pair routine compiled for the CPU).
GPU (-m gpu): qv_distance_pairs, qv_distance_rows, the flat scans (one and several queries per pass) and the key-per-row path."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import _oracle as O

sys.path.insert(0, os.path.join(O.ROOT, "oracle"))
import oracle_np as ONP  # noqa: E402

DOC = json.load(open(os.path.join(O.ROOT, "tests", "golden", "derived_kats.json")))
KATS = DOC["distance"]


def _bits(x) -> int:
    return int(np.float32(x).view(np.uint32))


def _want(kat) -> int:
    return int(kat["want_bits"], 16)


def test_the_fixture_is_what_its_generator_derives(tmp_path):
    """the committed JSON equals a fresh run of the exact-arithmetic derivation (which also asserts that every wrong restatement
    listed really yields different bits, and that every bit pattern is stated in the hand-written derivation text)"""
    before = open(os.path.join(O.ROOT, "tests", "golden", "derived_kats.json")).read()
    env = dict(os.environ)
    subprocess.check_call([sys.executable, os.path.join(O.ROOT, "tests", "golden", "make_derived_kats.py")], env=env, stdout=subprocess.DEVNULL)
    assert open(os.path.join(O.ROOT, "tests", "golden", "derived_kats.json")).read() == before
    assert len(KATS) >= 12 and {k["metric"] for k in KATS} == set(range(9))
    for k in KATS:
        assert k["wrong_restatements"] and all(int(w, 16) != _want(k) for w in k["wrong_restatements"].values())
        assert all(np.float32(x) == x for x in k["a"] + k["b"])          # inputs are float32 values


@pytest.mark.parametrize("kat", KATS, ids=lambda k: k["name"])
def test_c_oracle(kat):
    assert _bits(O.distance(kat["metric"], kat["a"], kat["b"])) == _want(kat), kat["derivation"]
    assert _bits(O.distance(kat["metric"], kat["b"], kat["a"])) == _want(kat)            # every metric is bitwise symmetric


@pytest.mark.parametrize("kat", KATS, ids=lambda k: k["name"])
def test_numpy_mirror(kat):
    assert _bits(ONP.distance(kat["metric"], kat["a"], kat["b"])) == _want(kat), kat["derivation"]


@pytest.mark.parametrize("kat", KATS, ids=lambda k: k["name"])
def test_host_distance_pair(kat):
    from quiver_amd.device_index import distance_pair
    assert _bits(distance_pair(kat["metric"], kat["a"], kat["b"])) == _want(kat), kat["derivation"]


@pytest.mark.gpu
@pytest.mark.parametrize("kat", KATS, ids=lambda k: k["name"])
def test_device_kernels(kat):
    """the same bits from every kernel family that evaluates this metric: pairs, listed rows, the single-query scan, the
    multi-query scan (padding rows and extra queries around the vector), the key-per-row + selection path and the full ranking"""
    import quiver_amd as q
    from quiver_amd.device_index import distance_pairs
    a = np.array(kat["a"], np.float32); b = np.array(kat["b"], np.float32)
    m, want = kat["metric"], _want(kat)
    assert _bits(distance_pairs(m, a, b)[0]) == want
    dim = a.size
    rng = np.random.default_rng(7)
    filler = (rng.standard_normal((200, dim)) * 1e3).astype(np.float32)
    rows = np.vstack([filler[:77], b[None, :], filler[77:]])             # the vector is row 77 of 201
    idx = q.DeviceIndex(dim, m)
    idx.add(rows)
    assert _bits(idx.distance_rows(a, np.array([77], np.uint32))[0]) == want
    for k in (201, 10):
        for nq in (1, 3, 9):                                              # one query; QB = 4; QB = 16
            qs = np.vstack([a[None, :]] + [filler[i:i + 1] for i in range(nq - 1)])
            r, d, c = idx.search(qs, k)
            hit = np.nonzero(r[0] == 77)[0]
            if k == 201:
                assert hit.size == 1
            if hit.size:
                assert _bits(d[0, hit[0]]) == want, (k, nq)
    # key per row + radix selection (64 < k <= 8192) and the full ranking (k = N)
    big = np.vstack([rows] * 50)                                          # 10 050 rows, the vector 50 times
    idx2 = q.DeviceIndex(dim, m)
    idx2.add(big)
    for k in (100, 300, big.shape[0]):
        r, d, c = idx2.search(a, k)
        hits = [j for j in range(int(c[0])) if r[0, j] % 201 == 77]
        if k == big.shape[0]:
            assert len(hits) == 50
        assert all(_bits(d[0, j]) == want for j in hits), k
    er, ed = O.exact_search(m, big, a, 300)
    r, d, c = idx2.search(a, 300)
    assert np.array_equal(r[0], er) and np.array_equal(d[0].view(np.uint32), ed.view(np.uint32))
