"""Mirror of pkg/vectortypes for the hot path: the DistanceFunc contract
(surface.go:8 `func(a, b F32) float32`), the five metrics (distances.go:12-104), the
DistanceType lookup (types.go:36-49) and Surface (surface.go:11-44).  One pair is what a
DistanceFunc call is (78 ns in the reference, final_bench.txt:47): it goes to libqv's host entry
point qv_distance_pair — the kernels' own per-pair routine compiled for the CPU, bit-identical to the
scans — not through a device round trip; arrays of pairs go to the device (qv_distance_pairs).
There is no arithmetic in this file."""
from __future__ import annotations

import numpy as np

from ._lib import metric_id
from .device_index import distance_pair, distance_pairs  # noqa: F401

# DistanceType, types.go:15-26
Cosine, Euclidean, DotProduct, Manhattan = "cosine", "euclidean", "dot_product", "manhattan"


def _mk(metric: str, device: int = 0):
    mid = metric_id(metric)

    def f(a, b) -> np.float32:
        a = np.ascontiguousarray(a, dtype=np.float32).ravel()
        b = np.ascontiguousarray(b, dtype=np.float32).ravel()
        if a.size != b.size:
            # distances.go:13-15 panics; a Python caller gets the same message as an exception
            raise ValueError("vectors must have the same length")
        if a.size == 0:
            raise ValueError("vectors must not be empty")
        return np.float32(distance_pair(mid, a, b))

    f.metric = metric
    f.metric_id = mid
    return f


CosineDistance = _mk("cosine")                       # distances.go:12-40
EuclideanDistance = _mk("euclidean")                 # distances.go:43-55
SquaredEuclideanDistance = _mk("squared_euclidean")  # distances.go:60-72
DotProductDistance = _mk("dot_product")              # distances.go:77-90
ManhattanDistance = _mk("manhattan")                 # distances.go:93-104


def GetDistanceFuncByType(dist_type: str):
    """types.go:36-49; unknown -> cosine (types.go:46-47)"""
    return {Cosine: CosineDistance, Euclidean: EuclideanDistance, DotProduct: DotProductDistance,
            Manhattan: ManhattanDistance}.get(dist_type, CosineDistance)


def ComputeDistance(a, b, dist_type: str) -> np.float32:
    """types.go:68-75: returns an error (not a panic) on length mismatch"""
    if len(a) != len(b):
        raise ValueError("vectors must have the same length")
    return GetDistanceFuncByType(dist_type)(a, b)


class BasicSurface:
    """surface.go:32-44"""

    def __init__(self, dist_func):
        self.DistFunc = dist_func

    def Distance(self, a, b) -> np.float32:
        return self.DistFunc(a, b)


def CreateSurface(dist_func) -> BasicSurface:
    return BasicSurface(dist_func)


class ContraMap:
    """surface.go:16-30: apply a surface to another type through a mapping function"""

    def __init__(self, surface, contra_map):
        self.Surface, self.ContraMapFn = surface, contra_map

    def Distance(self, a, b) -> np.float32:
        return self.Surface.Distance(self.ContraMapFn(a), self.ContraMapFn(b))


CosineSurface = CreateSurface(CosineDistance)
EuclideanSurface = CreateSurface(EuclideanDistance)
SquaredEuclideanSurface = CreateSurface(SquaredEuclideanDistance)
DotProductSurface = CreateSurface(DotProductDistance)
ManhattanSurface = CreateSurface(ManhattanDistance)


def GetSurfaceByType(dist_type: str) -> BasicSurface:
    """types.go:52-65"""
    return {Cosine: CosineSurface, Euclidean: EuclideanSurface, DotProduct: DotProductSurface,
            Manhattan: ManhattanSurface}.get(dist_type, CosineSurface)


def batch_distances(metric, a, b) -> np.ndarray:
    """n pairs in one device call (what a Go caller holding many pairs should use)"""
    return distance_pairs(metric, a, b)
