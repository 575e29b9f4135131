"""qv_distance_pair — the DistanceFunc contract for ONE pair on the host (pkg/vectortypes/surface.go:8): the
reference's literal known-answer tests through the C ABI, and bit-for-bit agreement with the CPU oracle on random and
awkward inputs for all nine metrics.  Runs without a GPU: the entry point is the kernels' own per-pair routine compiled for
the CPU (qv_kernels.h pair_distance), not a fallback of the scan paths."""
import json
import os

import numpy as np
import pytest

from quiver_amd.device_index import distance_pair
from tests import _oracle as O

KATS = json.load(open(os.path.join(O.ROOT, "tests", "golden", "ref_kats.json")))


@pytest.mark.parametrize("kat", KATS["distance"], ids=lambda k: k["src"])
def test_reference_kats(kat):
    got = distance_pair(kat["metric"], kat["a"], kat["b"])
    assert abs(got - kat["want"]) <= kat["tol"], (kat, got)


def test_length_mismatch_is_the_wrappers_error():
    """the reference panics (distances.go:13-15); the wrapper that knows both lengths raises"""
    for kat in KATS["distance_length_mismatch"]:
        with pytest.raises(ValueError):
            distance_pair(0, kat["a"], kat["b"])


@pytest.mark.parametrize("metric", range(9))
@pytest.mark.parametrize("dim", [1, 3, 4, 63, 128, 768, 1000])
def test_bits_equal_the_oracle(metric, dim):
    rng = np.random.default_rng(metric * 1000 + dim)
    for scale in (1.0, 1e-20, 1e18):
        a = (rng.standard_normal(dim) * scale).astype(np.float32)
        b = (rng.standard_normal(dim) * scale).astype(np.float32)
        want = np.float32(O.distance(metric, a, b))
        got = np.float32(distance_pair(metric, a, b))
        assert got.view(np.uint32) == want.view(np.uint32) or (np.isnan(got) and np.isnan(want)), (metric, dim, scale, got, want)
    z = np.zeros(dim, np.float32)
    a = rng.standard_normal(dim).astype(np.float32)
    for x, y in ((z, a), (a, z), (z, z), (a, a), (a, -a)):
        want, got = np.float32(O.distance(metric, x, y)), np.float32(distance_pair(metric, x, y))
        assert got.view(np.uint32) == want.view(np.uint32), (metric, dim)


def test_unknown_metric_is_an_error_below_the_wrapper():
    import quiver_amd
    out = __import__("ctypes").c_float(0)
    a = np.ones(4, np.float32)
    assert quiver_amd.lib().qv_distance_pair(99, a.ctypes.data, a.ctypes.data, 4, out) == quiver_amd._lib.QV_ERR_INVALID_ARG
