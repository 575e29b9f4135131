"""Mirror of the pkg/core.Collection SURFACE (collection.go:99-1108): Add / AddBatch / Get /
Delete / DeleteBatch / Update / Search / FluentSearch with the reference's validation order
and sentinel errors, around the `core.Index` seam (collection.go:78-96).  This is host
bookkeeping (ids, metadata JSON, filters), not compute: the index underneath is the
GPU-backed hybrid.HybridIndex (or anything with Insert/Delete/Search/Size).  Filtered
searches ask the index for a FULL ranking (searchK = Index.Size(), collection.go:679-682),
which is the k = N path of libqv."""
from __future__ import annotations

import json
import threading
from dataclasses import dataclass, field
from typing import Any, List, Optional

import numpy as np


class CoreError(Exception):
    pass


class ErrVectorNotFound(CoreError):          # collection.go:18
    def __init__(self, detail: str = ""):
        super().__init__("vector not found" + detail)


class ErrInvalidDimension(CoreError):        # collection.go:19
    def __init__(self, detail: str = ""):
        super().__init__("invalid vector dimension" + detail)


class ErrVectorAlreadyExist(CoreError):      # collection.go:20
    def __init__(self, detail: str = ""):
        super().__init__("vector with the same ID already exists" + detail)


class ErrInvalidMetadata(CoreError):         # collection.go:21
    def __init__(self, detail: str = ""):
        super().__init__("invalid metadata format" + detail)


# FilterOperator, collection.go:27-44
Equals, NotEquals, GreaterThan, GreaterThanOrEqual, LessThan, LessThanOrEqual, In, NotIn = "=", "!=", ">", ">=", "<", "<=", "in", "not_in"


@dataclass
class Filter:                                # pkg/types/search.go:66-73
    Field: str
    Operator: str
    Value: Any


@dataclass
class SearchOptions:                         # search.go:45-52
    IncludeVectors: bool = False
    IncludeMetadata: bool = False
    ExactSearch: bool = False


@dataclass
class SearchRequest:                         # search.go:75-86
    Vector: Any = None
    TopK: int = 0
    Filters: List[Filter] = field(default_factory=list)
    Options: SearchOptions = field(default_factory=SearchOptions)
    NamespaceID: str = ""


@dataclass
class SearchResultItem:
    ID: str
    Distance: float
    Score: float                             # 1 - Distance (collection.go:763)
    Vector: Any = None
    Metadata: Any = None


@dataclass
class SearchResultMetadata:
    TotalCount: int
    IndexSize: int
    IndexName: str


@dataclass
class SearchResponse:
    Results: List[SearchResultItem]
    Metadata: SearchResultMetadata
    Query: Any = None


@dataclass
class Vector:                                # vectortypes/types.go:29-33
    ID: str
    Values: Any
    Metadata: Any = None


def _as_float(v):                            # collection.go:575-597
    if isinstance(v, bool):
        return None
    if isinstance(v, (int, float)):
        return float(v)
    return None


def _values_equal(a, b) -> bool:             # collection.go:600-607
    fa, fb = _as_float(a), _as_float(b)
    if fa is not None and fb is not None:
        return abs(fa - fb) <= 1e-9
    return _go_str(a) == _go_str(b)


def _go_str(v) -> str:                       # fmt.Sprintf("%v", v) for JSON-decoded values
    if isinstance(v, bool):
        return "true" if v else "false"
    if v is None:
        return "<nil>"
    if isinstance(v, float) and v == int(v):
        return str(int(v))
    return str(v)


def _compare(a, b) -> int:                   # collection.go:609-632
    fa, fb = _as_float(a), _as_float(b)
    if fa is not None and fb is not None:
        return -1 if fa < fb else (1 if fa > fb else 0)
    sa, sb = _go_str(a), _go_str(b)
    return -1 if sa < sb else (1 if sa > sb else 0)


def matchesFilter(metadata: dict, f: Filter) -> bool:   # collection.go:530-573
    if f.Field not in metadata:
        return False
    value = metadata[f.Field]
    op = f.Operator
    if op == Equals:
        return _values_equal(value, f.Value)
    if op == NotEquals:
        return not _values_equal(value, f.Value)
    if op == GreaterThan:
        return _compare(value, f.Value) > 0
    if op == GreaterThanOrEqual:
        return _compare(value, f.Value) >= 0
    if op == LessThan:
        return _compare(value, f.Value) < 0
    if op == LessThanOrEqual:
        return _compare(value, f.Value) <= 0
    if op == In:
        return isinstance(f.Value, (list, tuple)) and any(_values_equal(value, v) for v in f.Value)
    if op == NotIn:
        if isinstance(f.Value, (list, tuple)):
            return not any(_values_equal(value, v) for v in f.Value)
        return True
    return False


def _validate_metadata(metadata, suffix: str = ""):
    """json.Unmarshal into map[string]interface{} (collection.go:159-167): must be a JSON object"""
    if metadata is None or len(metadata) == 0:
        return
    try:
        m = json.loads(metadata)
    except Exception as e:  # noqa: BLE001
        raise ErrInvalidMetadata(f"{suffix}: {e}")
    if not isinstance(m, dict):
        raise ErrInvalidMetadata(f"{suffix}: json: cannot unmarshal into Go value of type map[string]interface {{}}")


class Collection:
    def __init__(self, name: str, dimension: int, index):   # NewCollection, collection.go:121-130
        self.Name, self.Dimension, self.Index = name, dimension, index
        self.Metadata: dict = {}
        self.Vectors: dict = {}
        self._lock = threading.RLock()

    # ---- Add, collection.go:133-206
    def Add(self, id: str, vector, metadata=None) -> None:
        with self._lock:
            if id == "":
                raise CoreError("vector ID cannot be empty")                                  # :143-148
            if len(vector) != self.Dimension:
                raise ErrInvalidDimension(f": expected {self.Dimension}, got {len(vector)}")  # :151-156
            _validate_metadata(metadata)                                                      # :159-167
            if id in self.Vectors:
                raise ErrVectorAlreadyExist(f": {id}")                                        # :170-175
            self.Index.Insert(id, vector)                                                     # :178 (seam)
            self.Vectors[id] = vector                                                         # :186 (not copied, as in Go)
            self.Metadata[id] = metadata

    # ---- AddBatch, collection.go:209-331
    def AddBatch(self, vectors: List[Vector]) -> None:
        with self._lock:
            if len(vectors) == 0:
                raise CoreError("no vectors provided for batch insert")                       # :222-227
            for v in vectors:                                                                 # :231-264 validate everything first
                if v.ID == "":
                    raise CoreError("vector ID cannot be empty")
                if len(v.Values) != self.Dimension:
                    raise ErrInvalidDimension(f" for vector {v.ID}: expected {self.Dimension}, got {len(v.Values)}")
                _validate_metadata(v.Metadata, f" for vector {v.ID}")
                if v.ID in self.Vectors:
                    raise ErrVectorAlreadyExist(f": {v.ID}")
            if hasattr(self.Index, "InsertBatch"):                                            # :267 type-assert core.BatchIndex
                self.Index.InsertBatch({v.ID: v.Values for v in vectors})                     # :269-277
                for v in vectors:
                    self.Vectors[v.ID] = v.Values
                    self.Metadata[v.ID] = v.Metadata
                return
            for v in vectors:                                                                 # :305-321 fallback
                self.Index.Insert(v.ID, v.Values)
                self.Vectors[v.ID] = v.Values
                self.Metadata[v.ID] = v.Metadata

    def Get(self, id: str) -> Vector:                                                         # :334-353
        with self._lock:
            if id not in self.Vectors:
                raise ErrVectorNotFound()
            md = self.Metadata.get(id)
            return Vector(id, self.Vectors[id], md if md is not None else "{}")

    def Delete(self, id: str) -> None:                                                        # :356-372
        with self._lock:
            if id not in self.Vectors:
                raise ErrVectorNotFound()
            self.Index.Delete(id)
            del self.Vectors[id]
            self.Metadata.pop(id, None)

    def DeleteBatch(self, ids: List[str]) -> None:                                            # :375-414
        with self._lock:
            for id in ids:
                if id not in self.Vectors:
                    raise ErrVectorNotFound(f": {id}")
            if hasattr(self.Index, "DeleteBatch"):
                self.Index.DeleteBatch(ids)
            else:
                for id in ids:
                    self.Index.Delete(id)
            for id in ids:
                del self.Vectors[id]
                self.Metadata.pop(id, None)

    def Update(self, id: str, vector=None, metadata=None) -> None:                            # :417-465
        with self._lock:
            if id not in self.Vectors:
                raise ErrVectorNotFound()
            if vector is not None and len(vector) != self.Dimension:
                raise ErrInvalidDimension(f": expected {self.Dimension}, got {len(vector)}")
            _validate_metadata(metadata)
            if vector is not None:
                self.Index.Delete(id)                                                         # :438-444 delete + re-insert
                self.Index.Insert(id, vector)
                self.Vectors[id] = vector
            if metadata is not None and len(metadata) > 0:
                self.Metadata[id] = metadata

    def Count(self) -> int:                                                                   # :855-859
        with self._lock:
            return self.Index.Size()

    # ---- Search, collection.go:637-807
    def Search(self, request: SearchRequest) -> SearchResponse:
        with self._lock:
            if len(request.Vector) != self.Dimension:                                         # :651-656
                raise ErrInvalidDimension(f": expected {self.Dimension}, got {len(request.Vector)}")
            if request.TopK <= 0:                                                             # :659-664
                raise CoreError("top_k must be greater than 0")
            if self.Index.Size() == 0:                                                        # :665-676
                return SearchResponse([], SearchResultMetadata(0, 0, self.Name))
            searchK = request.TopK
            if len(request.Filters) > 0:
                searchK = self.Index.Size()                                                   # :679-682 FULL ranking
            basic = self.Index.Search(request.Vector, searchK)                                # :686 (seam)
            if len(request.Filters) > 0:                                                      # :704-752
                kept = []
                for r in basic:
                    mj = self.Metadata.get(r.ID)
                    if mj is None or len(mj) == 0:
                        continue
                    try:
                        m = json.loads(mj)
                    except Exception:  # noqa: BLE001
                        continue
                    if not isinstance(m, dict):
                        continue
                    if all(matchesFilter(m, f) for f in request.Filters):
                        kept.append(r)
                        if len(kept) >= request.TopK:
                            break
                basic = kept
            elif len(basic) > request.TopK:                                                   # :753-755
                basic = basic[: request.TopK]
            items = []
            for r in basic:                                                                   # :758-779
                d = np.float32(r.Distance)
                it = SearchResultItem(r.ID, float(d), float(np.float32(1.0) - d))
                if request.Options.IncludeVectors:
                    it.Vector = self.Vectors.get(r.ID)
                if request.Options.IncludeMetadata:
                    it.Metadata = self.Metadata.get(r.ID)
                items.append(it)
            resp = SearchResponse(items, SearchResultMetadata(len(items), self.Index.Size(), self.Name))
            if request.Options.IncludeVectors:
                resp.Query = request.Vector
            return resp

    def FluentSearch(self, vector) -> "FluentSearch":                                         # :886-904
        return FluentSearch(self, vector)


class FluentSearch:
    """collection.go:874-1108: errors are sticky and surface at Execute()"""

    def __init__(self, collection: Collection, vector):
        self.collection, self.vector, self.k = collection, vector, 10
        self.filters: List[Filter] = []
        self.options = SearchOptions(IncludeMetadata=True)
        self.namespaceID = ""
        self.valid, self.err = True, None
        if len(vector) != collection.Dimension:                                               # :898-901
            self.valid = False
            self.err = ErrInvalidDimension(f": expected {collection.Dimension}, got {len(vector)}")

    def _bad(self, msg):
        self.valid, self.err = False, CoreError(msg)
        return self

    def WithK(self, k: int):                                                                  # :932-945
        if not self.valid:
            return self
        if k <= 0:
            return self._bad("k must be greater than 0")
        self.k = k
        return self

    def WithNamespace(self, ns: str):
        if self.valid:
            self.namespaceID = ns
        return self

    def IncludeVectors(self, inc: bool):
        if self.valid:
            self.options.IncludeVectors = inc
        return self

    def IncludeMetadata(self, inc: bool):
        if self.valid:
            self.options.IncludeMetadata = inc
        return self

    def UseExactSearch(self):                                                                 # :978-985 (plumbed, never read: SURVEY 3.1)
        if self.valid:
            self.options.ExactSearch = True
        return self

    def _filter(self, field_: str, op: str, value):
        if not self.valid:
            return self
        if field_ == "":
            return self._bad("filter field cannot be empty")
        self.filters.append(Filter(field_, op, value))
        return self

    def Filter(self, field_: str, value):
        return self._filter(field_, Equals, value)

    def FilterNotEquals(self, field_: str, value):
        return self._filter(field_, NotEquals, value)

    def FilterGreaterThan(self, field_: str, value):
        return self._filter(field_, GreaterThan, value)

    def FilterLessThan(self, field_: str, value):
        return self._filter(field_, LessThan, value)

    def FilterIn(self, field_: str, values):
        if not self.valid:
            return self
        if field_ == "":
            return self._bad("filter field cannot be empty")
        if len(values) == 0:
            return self._bad("filter values cannot be empty")
        self.filters.append(Filter(field_, In, list(values)))
        return self

    def Execute(self) -> SearchResponse:                                                      # :1094-1108
        if not self.valid:
            raise self.err
        if self.vector is None:
            raise CoreError("query vector is nil")
        if self.k <= 0:
            raise CoreError("k must be greater than 0")
        if self.k > self.collection.Count():                                                  # :924-926 clamp
            self.k = self.collection.Count()
        return self.collection.Search(SearchRequest(Vector=self.vector, TopK=self.k, Filters=self.filters,
                                                    Options=self.options, NamespaceID=self.namespaceID))
