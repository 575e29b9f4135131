"""Mirror of pkg/persistence.Collection's vector-search surface (collection.go:46-283): the third flat-scan caller in the
reference besides hybrid.ExactIndex and core.Collection.  The reference computes one DistanceFunc call per
stored vector and orders the results with an O(N^2) selection sort (SortSearchResults, collection.go:270-278); here the
vectors live in a device index and Search is ONE qv_index_search call — `limit` results, or the full ranking when
limit <= 0 or limit >= N (the reference returns everything then, collection.go:256-258).  WAL / Parquet / facets stay in Go.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np

from ._lib import metric_id
from .device_index import DeviceIndex


class GoError(Exception):
    """an `error` value of the mirrored Go function (message = the Go error string)"""


@dataclass
class SearchResult:                    # collection.go:263-267
    ID: str
    Distance: float


def SortSearchResults(results: List[SearchResult]) -> None:
    """collection.go:270-278 (selection sort, ascending by Distance) — restated for its KAT; Search does not need it"""
    for i in range(len(results)):
        for j in range(i + 1, len(results)):
            if results[j].Distance < results[i].Distance:
                results[i], results[j] = results[j], results[i]


class Collection:
    def __init__(self, name: str, dimension: int, distance_func, device: int = 0):    # NewCollection, collection.go:46-59
        self.name, self.dimension = name, int(dimension)
        self._metric = None if distance_func is None else (distance_func if isinstance(distance_func, (int, str)) else getattr(distance_func, "metric_id"))
        self._idx = None if self._metric is None else DeviceIndex(self.dimension, metric_id(self._metric), device=device)
        self._row_of: Dict[str, int] = {}
        self._id_of: Dict[int, str] = {}
        self._free: List[int] = []
        self._meta: Dict[str, Dict[str, str]] = {}
        self._host: Dict[str, np.ndarray] = {}                       # distanceFunc == nil: the reference still stores, returns and deletes
                                                                     # vectors (collection.go:99-208); only Search errors.  No device index then.
        self.dirty = False

    def GetName(self) -> str:
        return self.name

    def GetDimension(self) -> int:
        return self.dimension

    def AddVector(self, id: str, vector, metadata: Optional[Dict[str, str]] = None) -> None:   # collection.go:99-148
        v = np.ascontiguousarray(vector, dtype=np.float32).ravel()
        if v.size != self.dimension:
            raise GoError(f"vector dimension mismatch: got {v.size}, expected {self.dimension}")
        if self._idx is not None:
            if id in self._row_of:                                   # c.vectors[id] = vecCopy: an existing id is overwritten
                self._idx.update(self._row_of[id], v)
            elif self._free:
                row = self._free.pop()
                self._idx.update(row, v)                             # revives the tombstoned row in place
                self._row_of[id], self._id_of[row] = row, id
            else:
                row = self._idx.add(v)
                self._row_of[id], self._id_of[row] = row, id
        else:
            self._host[id] = v.copy()
        if metadata is not None:
            self._meta[id] = dict(metadata)
        self.dirty = True

    def DeleteVector(self, id: str) -> None:                         # collection.go:151-182
        if self._idx is None:
            if id not in self._host:
                raise GoError(f"vector with ID {id} not found")
            del self._host[id]
            self._meta.pop(id, None)
            self.dirty = True
            return
        if id not in self._row_of:
            raise GoError(f"vector with ID {id} not found")
        row = self._row_of.pop(id)
        self._idx.remove([row])
        del self._id_of[row]
        self._free.append(row)
        self._meta.pop(id, None)
        self.dirty = True

    def GetVector(self, id: str):                                    # collection.go:185-208
        if id not in self._row_of and id not in self._host:
            raise GoError(f"vector with ID {id} not found")
        meta = self._meta.get(id)
        vec = self._host[id].copy() if self._idx is None else self._idx.get_row(self._row_of[id])
        return vec, (dict(meta) if meta is not None else None)

    def Count(self) -> int:                                          # collection.go:281-285
        return len(self._row_of) + len(self._host)

    def Search(self, query, limit: int) -> List[SearchResult]:       # collection.go:226-261
        if self._idx is None:
            raise GoError("distance function is not set")
        if query is None:
            raise GoError("query vector is nil")
        q = np.ascontiguousarray(query, dtype=np.float32).ravel()
        if q.size != self.dimension:
            raise GoError(f"query vector dimension mismatch: got {q.size}, expected {self.dimension}")
        n = len(self._row_of)
        if n == 0:
            return []
        k = limit if 0 < limit < n else n                            # :256-258: limit <= 0 or >= N returns every vector, ranked
        rows, dist, count = self._idx.search(q, k)
        return [SearchResult(self._id_of[int(rows[0, i])], float(dist[0, i])) for i in range(int(count[0]))]
