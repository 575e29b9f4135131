"""Mirror of pkg/hybrid (exact.go, hybrid_index.go, adaptive.go, hnsw_adapter.go): Python
faces of the C++ host classes in quiver_amd/csrc/host (quiver::ExactIndex, HNSWAdapter,
HybridIndex).  Same method names, argument meaning and error strings as the Go types."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from ._host import GoError, Results, check, f32, hlib
from ._lib import metric_id

ExactIndexType, HNSWIndexType, HybridIndexType = "exact", "hnsw", "hybrid"   # types.go:16-24


def _metric_of(dist_func) -> int:
    """db.go:326-334 identifies the metric by function identity; here the mirrored
    vectortypes functions carry their metric id.  Arbitrary callables cannot be offloaded."""
    if dist_func is None:
        return 0                       # hybrid_index.go:46-48 default cosine
    if isinstance(dist_func, (int, str)):
        return metric_id(dist_func)
    mid = getattr(dist_func, "metric_id", None)
    if mid is None:
        raise TypeError("distance function is not one of the offloadable vectortypes/hnsw metrics")
    return mid


@dataclass
class BasicSearchResult:               # pkg/types/search.go:9-14
    ID: str
    Distance: float


def _res(pairs):
    return [BasicSearchResult(i, d) for i, d in pairs]


# Where new ExactIndex / HybridIndex objects keep their rows when the caller does not say: None = one device, or a dict
# {"Devices": [...], "PeerCopy": bool, "Bf16Rows": bool}.  The GPU tests run the host mirror over several placements through it.
DEFAULT_PLACEMENT = None


def _devs(devices):
    d = np.ascontiguousarray(list(devices), dtype=np.int32)
    if d.size == 0:
        raise ValueError("device list is empty")
    return d


class ExactIndex:
    """pkg/hybrid/exact.go:14-160.  `devices`: shard the rows over a device list (qv_sharded_*) instead of one
    `device`; peer_copy: point-to-point exchange (shards may then share a device); bf16_rows: QV_FLAG_BF16_ROWS."""

    def __init__(self, dist_func=None, device: int = 0, devices=None, peer_copy: bool = False, bf16_rows: bool = False):
        if devices is None and not peer_copy and not bf16_rows and DEFAULT_PLACEMENT:
            devices, peer_copy, bf16_rows = DEFAULT_PLACEMENT.get("Devices"), DEFAULT_PLACEMENT.get("PeerCopy", False), DEFAULT_PLACEMENT.get("Bf16Rows", False)
        if devices is None and not peer_copy and not bf16_rows:
            self._h = hlib().qvh_exact_new(_metric_of(dist_func), device)
        else:
            d = _devs(devices if devices is not None else [device])
            self._h = hlib().qvh_exact_new_placed(_metric_of(dist_func), d.ctypes.data, d.size, int(peer_copy), int(bf16_rows))

    def __del__(self):
        try:
            hlib().qvh_exact_free(self._h)
        except Exception:
            pass

    def Insert(self, id: str, vector) -> None:
        v = f32(vector).copy()
        check(hlib().qvh_exact_insert(self._h, id.encode(), v.ctypes.data, v.size))

    def Delete(self, id: str) -> None:
        check(hlib().qvh_exact_delete(self._h, id.encode()))

    def Search(self, query, k: int):
        q = f32(query)
        r = Results()
        check(hlib().qvh_exact_search(self._h, q.ctypes.data, q.size, k, r.h))
        return _res(r.list())

    def Size(self) -> int:
        return hlib().qvh_exact_size(self._h)

    def GetType(self) -> str:
        return ExactIndexType


@dataclass
class HNSWConfig:                      # types.go:47-70
    M: int = 16
    MaxM0: int = 32
    EfConstruction: int = 200
    EfSearch: int = 100


def DefaultHNSWConfig() -> HNSWConfig:
    return HNSWConfig()


@dataclass
class IndexConfig:                     # types.go:27-45
    DistanceFunc: object = None
    HNSWConfig: HNSWConfig = field(default_factory=HNSWConfig)
    ExactThreshold: int = 1000
    # not in the reference: its level RNG / exploration RNG are wall-clock / global seeded
    Seed: int = 1
    ExplorationFactor: float = 0.1     # DefaultAdaptiveConfig, types.go:93
    # not in the reference (one process on CPU cores): where the exact index keeps its rows — a device list shards them
    # (qv_sharded_*), the HNSW graph stays on the first device; PeerCopy lets shards share a device; Bf16Rows = QV_FLAG_BF16_ROWS
    Devices: object = None
    PeerCopy: bool = False
    Bf16Rows: bool = False


def DefaultIndexConfig() -> IndexConfig:
    return IndexConfig()


class HNSWAdapter:
    """pkg/hybrid/hnsw_adapter.go over pkg/hnsw/adapter.go"""

    def __init__(self, dist_func=None, config: Optional[HNSWConfig] = None, device: int = 0, seed: int = 1):
        c = config or HNSWConfig()
        self._h = hlib().qvh_adapter_new(_metric_of(dist_func), device, c.M, c.MaxM0, c.EfConstruction, c.EfSearch, seed)

    def __del__(self):
        try:
            hlib().qvh_adapter_free(self._h)
        except Exception:
            pass

    def Insert(self, id: str, vector) -> None:
        v = f32(vector)
        check(hlib().qvh_adapter_insert(self._h, id.encode(), v.ctypes.data, v.size))

    def Delete(self, id: str) -> None:
        check(hlib().qvh_adapter_delete(self._h, id.encode()))

    def Search(self, query, k: int):
        q = f32(query)
        r = Results()
        check(hlib().qvh_adapter_search(self._h, q.ctypes.data, q.size, k, r.h))
        return _res(r.list())

    def SearchWithNegative(self, query, negative, weight: float, k: int):
        q, n = f32(query), f32(negative)
        r = Results()
        check(hlib().qvh_adapter_search_negative(self._h, q.ctypes.data, q.size, n.ctypes.data, n.size, weight, k, r.h))
        return _res(r.list())

    def Size(self) -> int:
        return hlib().qvh_adapter_size(self._h)

    def GetType(self) -> str:
        return HNSWIndexType


@dataclass
class HybridSearchRequest:             # hybrid_index.go (request struct)
    Query: object = None
    K: int = 0
    ForceStrategy: str = ""
    IncludeStats: bool = False
    NegativeExample: object = None
    NegativeWeight: float = 0.0


@dataclass
class HybridSearchResponse:
    Results: list
    StrategyUsed: str


@dataclass
class BatchSearchRequest:
    Queries: list = None
    K: int = 0
    ForceStrategy: str = ""
    IncludeStats: bool = False


@dataclass
class BatchSearchResponse:
    Results: list
    StrategiesUsed: list


class HybridIndex:
    """pkg/hybrid/hybrid_index.go:15-811"""

    def __init__(self, config: Optional[IndexConfig] = None, device: int = 0):
        c = config or IndexConfig()
        h = c.HNSWConfig
        if c.Devices is None and not c.PeerCopy and not c.Bf16Rows and DEFAULT_PLACEMENT:
            c = IndexConfig(**{**c.__dict__, **{k: DEFAULT_PLACEMENT[k] for k in ("Devices", "PeerCopy", "Bf16Rows") if k in DEFAULT_PLACEMENT}})
        if c.Devices is None and not c.PeerCopy and not c.Bf16Rows:
            self._h = hlib().qvh_hybrid_new(_metric_of(c.DistanceFunc), device, h.M, h.MaxM0, h.EfConstruction, h.EfSearch,
                                            c.ExactThreshold, c.ExplorationFactor, c.Seed)
        else:
            d = _devs(c.Devices if c.Devices is not None else [device])
            self._h = hlib().qvh_hybrid_new_placed(_metric_of(c.DistanceFunc), d.ctypes.data, d.size, int(c.PeerCopy), int(c.Bf16Rows), h.M, h.MaxM0,
                                                   h.EfConstruction, h.EfSearch, c.ExactThreshold, c.ExplorationFactor, c.Seed)

    def __del__(self):
        try:
            hlib().qvh_hybrid_free(self._h)
        except Exception:
            pass

    def Insert(self, id: str, vector) -> None:
        v = f32(vector).copy()
        check(hlib().qvh_hybrid_insert(self._h, id.encode(), v.ctypes.data, v.size))

    def InsertBatch(self, vectors: dict) -> None:
        """core.BatchIndex (collection.go:93).  A Go map has no order; dict order is used."""
        if not vectors:
            return
        ids = list(vectors.keys())
        arrs = [f32(vectors[i]) for i in ids]
        lens = np.array([a.size for a in arrs], dtype=np.uint32)
        packed = np.ascontiguousarray(np.concatenate(arrs) if arrs else np.zeros(0, np.float32))
        cids = (C.c_char_p * len(ids))(*[i.encode() for i in ids])
        check(hlib().qvh_hybrid_insert_batch(self._h, cids, packed.ctypes.data, lens.ctypes.data, len(ids)))

    def Delete(self, id: str) -> None:
        check(hlib().qvh_hybrid_delete(self._h, id.encode()))

    def DeleteBatch(self, ids) -> None:
        ids = list(ids)
        if not ids:
            return
        cids = (C.c_char_p * len(ids))(*[i.encode() for i in ids])
        check(hlib().qvh_hybrid_delete_batch(self._h, cids, len(ids)))

    def Search(self, query, k: int):
        q = f32(query)
        r = Results()
        check(hlib().qvh_hybrid_search(self._h, q.ctypes.data, q.size, k, r.h))
        return _res(r.list())

    def SearchWithRequest(self, req: HybridSearchRequest) -> HybridSearchResponse:
        q = f32(req.Query)
        neg = f32(req.NegativeExample) if req.NegativeExample is not None else np.zeros(0, np.float32)
        r = Results()
        check(hlib().qvh_hybrid_search_request(self._h, q.ctypes.data, q.size, req.K, req.ForceStrategy.encode(),
                                               neg.ctypes.data if neg.size else None, neg.size, req.NegativeWeight, r.h))
        return HybridSearchResponse(_res(r.list()), r.strategy())

    def BatchSearch(self, req: BatchSearchRequest) -> BatchSearchResponse:
        qs = [f32(q) for q in (req.Queries or [])]
        if not qs:
            raise GoError("no queries provided")                        # hybrid_index.go:678-680
        n0 = qs[0].size
        for i, q in enumerate(qs):                                      # per-query dimension check, :707-713
            if q.size != n0:
                raise GoError(f"query {i} dimension mismatch: expected {n0}, got {q.size}")
        packed = np.ascontiguousarray(np.stack(qs))
        r = Results()
        check(hlib().qvh_hybrid_batch_search(self._h, packed.ctypes.data, n0, len(qs), req.K, req.ForceStrategy.encode(), r.h))
        out, used = r.many()
        return BatchSearchResponse([_res(o) for o in out], used)

    def Size(self) -> int:
        return hlib().qvh_hybrid_size(self._h)

    def GetType(self) -> str:
        return HybridIndexType

    def SelectStrategy(self, vector_count: int, dimension: int, k: int) -> str:
        return hlib().qvh_hybrid_select_strategy(self._h, vector_count, dimension, k).decode()

    def FluentSearch(self, query):
        return FluentHybridSearch(self, query)


class FluentHybridSearch:
    """hybrid_index.go:814-881 fluent builder"""

    def __init__(self, index: HybridIndex, query):
        self._i, self._req = index, HybridSearchRequest(Query=query, K=10)

    def WithK(self, k: int):
        self._req.K = k
        return self

    def WithForceStrategy(self, s: str):
        self._req.ForceStrategy = s
        return self

    def WithNegativeExample(self, v):
        self._req.NegativeExample = v
        if self._req.NegativeWeight == 0:
            self._req.NegativeWeight = 0.5
        return self

    def WithNegativeWeight(self, w: float):
        self._req.NegativeWeight = w
        return self

    def Execute(self) -> HybridSearchResponse:
        return self._i.SearchWithRequest(self._req)
