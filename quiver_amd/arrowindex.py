"""Arrow columnar load -> device (north star: "pkg/arrowindex columnar load -> device").

`index.ArrowHNSWIndex.Load` (index/arrow_hnsw.go:201-241) reads an Arrow IPC file with
schema {id: utf8, vector: FixedSizeList<float32>[dim]}; per record batch the list's child
array is ONE contiguous float32 buffer of rows*dim values (arrow_hnsw.go:222-225), which the
reference then walks row by row, widening each to float64 and inserting into its graph.
Here that child buffer is handed to libqv as it is — it already IS the row-major [rows][dim]
matrix `qv_index_add` ingests (one H2D copy per <= 256 MiB, one ingest kernel), no per-row
work on the host.  `Save` writes the same schema (arrow_hnsw.go:138-198).

`ArrowFlatIndex` keeps ArrowHNSWIndex's Add / Search / Save / Load / Len surface and its
distance definition: Search returns, for each hit, the squared L2 distance recomputed in
float64 over float64-widened vectors and rounded to float32 (arrow_hnsw.go:124-132) —
metric QV_L2SQ_F64.  The ranking is an exact scan (what the reference's graph search
degenerates to when len(nodes) <= m, graph.go:482-484, and a recall-1.0 superset of it
otherwise).  Arrow IPC reading/writing itself is pyarrow's job (storage only).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import pyarrow as pa
import pyarrow.ipc as ipc


def schema_for(dim: int) -> pa.Schema:                        # arrow_hnsw.go:153-156
    return pa.schema([pa.field("id", pa.string()), pa.field("vector", pa.list_(pa.float32(), dim))])


def batch_values(batch: pa.RecordBatch, dim: int) -> np.ndarray:
    """zero-copy view of a record batch's vectors as float32 [rows, dim] (arrow_hnsw.go:218-225)"""
    col = batch.column(1)
    if not pa.types.is_fixed_size_list(col.type) or col.type.list_size != dim or not pa.types.is_float32(col.type.value_type):
        raise ValueError(f"vector column must be FixedSizeList<float32>[{dim}], got {col.type}")
    if col.null_count:
        raise ValueError("null vectors are not supported")
    flat = col.flatten()                                       # accounts for the list array's offset
    vals = flat.to_numpy(zero_copy_only=True)
    return vals.reshape(len(col), dim)


def save_ipc(path: str, ids: Sequence[str], vectors: np.ndarray) -> None:
    vectors = np.ascontiguousarray(vectors, dtype=np.float32)
    n, dim = vectors.shape
    if len(ids) != n:
        raise ValueError("ids and vectors disagree")
    arr = pa.FixedSizeListArray.from_arrays(pa.array(vectors.reshape(-1), type=pa.float32()), dim)
    batch = pa.record_batch([pa.array(list(ids), type=pa.string()), arr], schema=schema_for(dim))
    with ipc.new_file(path, schema_for(dim)) as w:
        w.write_batch(batch)


def load_ipc(path: str, dim: int, sink) -> List[str]:
    """stream every record batch of an IPC file into `sink(values[rows, dim], ids)`; returns all ids"""
    all_ids: List[str] = []
    with pa.memory_map(path, "r") as src:
        r = ipc.open_file(src)
        for i in range(r.num_record_batches):                  # arrow_hnsw.go:215
            b = r.get_batch(i)
            ids = b.column(0).to_pylist()
            sink(batch_values(b, dim), ids)
            all_ids.extend(ids)
    return all_ids


class Result:                                                  # index/arrow_hnsw.go:19-23
    __slots__ = ("ID", "Distance")

    def __init__(self, id_: str, distance: float):
        self.ID, self.Distance = id_, distance

    def __repr__(self):
        return f"Result(ID={self.ID!r}, Distance={self.Distance!r})"


class ArrowFlatIndex:
    def __init__(self, dim: int, device: int = 0):             # NewArrowHNSWIndex(dim), arrow_hnsw.go:39-55
        from .device_index import DeviceIndex
        self.dim = dim
        self._idx = DeviceIndex(dim, "arrow_squared_euclidean", device=device)
        self.idToIdx: dict = {}
        self.idxToID: List[str] = []

    def _add_block(self, values: np.ndarray, ids: Sequence[str]) -> None:
        for id_ in ids:                                        # addRaw's duplicate check, arrow_hnsw.go:60-63
            if id_ in self.idToIdx:
                raise ValueError(f"vector with ID {id_} already exists")
        if len(set(ids)) != len(ids):
            raise ValueError("duplicate ids in one block")
        first = self._idx.add(values)                          # the contiguous child buffer, as is
        for j, id_ in enumerate(ids):
            self.idToIdx[id_] = first + j
            self.idxToID.append(id_)

    def Add(self, vec, id: str) -> None:                       # arrow_hnsw.go:83-91
        v = np.ascontiguousarray(vec.to_numpy(zero_copy_only=False) if isinstance(vec, pa.Array) else vec, dtype=np.float32)
        if v.size != self.dim:
            raise ValueError(f"dimension mismatch: got {v.size} want {self.dim}")
        self._add_block(v.reshape(1, self.dim), [id])

    def Search(self, query, k: int) -> List[Result]:           # arrow_hnsw.go:94-135
        if k <= 0:
            raise ValueError("k must be positive")
        if query is None:
            raise ValueError("invalid or nil query vector")
        q = np.ascontiguousarray(query.to_numpy(zero_copy_only=False) if isinstance(query, pa.Array) else query, dtype=np.float32)
        if q.size != self.dim:
            raise ValueError(f"dimension mismatch: got {q.size} want {self.dim}")
        rows, dist, count = self._idx.search(q, k)
        n = int(count[0])
        return [Result(self.idxToID[int(rows[0, i])], float(dist[0, i])) for i in range(n)]

    def Len(self) -> int:
        return self._idx.size()

    def Save(self, path: str) -> None:                         # arrow_hnsw.go:138-198
        n = self._idx.rows()
        vecs = self._idx.get_rows(np.arange(n, dtype=np.uint32)) if n else np.zeros((0, self.dim), np.float32)   # one device pass (qv_index_get_rows)
        save_ipc(path, self.idxToID, vecs)

    def Load(self, path: str) -> None:                         # arrow_hnsw.go:201-241
        load_ipc(path, self.dim, self._add_block)


class Graph:
    """Mirror of arrowindex.Graph's query surface (pkg/arrowindex/graph.go:162-200, 459-534, 897-918): int ids, float64 vectors,
    squared-L2 search.  Storage is the device index (metric QV_L2SQ_F64: float64 differences and accumulation, graph.go:749-794)
    instead of chunked arrow Float64 arrays, so vectors must be float32-representable — which is what reaches this type in
    the reference (ArrowHNSWIndex widens float32 rows, arrow_hnsw.go:66-80); anything else is refused rather than rounded.
    Search follows graph.go:467-534: dimension check and wording, empty graph -> no results, k clamped to the node count,
    and an EXHAUSTIVE ranking when len(nodes) <= m (:482-484).  Above m the reference walks its randomly-levelled graph
    (math/rand, unseeded: graph.go:945-952) with ef = max(efSearch, 2k) (:522) and returns an approximation of the same
    ranking; here every size gets the exact ranking (for len(nodes) > m a recall-1.0 stand-in, declared in DESIGN.md)."""

    def __init__(self, dim: int, m: int = 16, efConstruction: int = 200, efSearch: int = 100, chunkSize: int = 1024, device: int = 0):
        from .device_index import DeviceIndex
        self.dim, self.m, self.efConstruction, self.efSearch, self.chunkSize = dim, m, efConstruction, efSearch, chunkSize
        self._idx = DeviceIndex(dim, "arrow_squared_euclidean", device=device)
        self._ids: List[int] = []                                  # node index -> ID (nodes[i].ID)

    def AddBatch(self, items) -> None:                             # graph.go:203-274
        vecs = []
        for id_, vec in items:
            v64 = np.asarray(vec, dtype=np.float64).ravel()
            if v64.size != self.dim:
                raise ValueError(f"vector dimension mismatch for id {id_}: got {v64.size}, want {self.dim}")
            v32 = v64.astype(np.float32)
            if not np.array_equal(v32.astype(np.float64), v64):
                raise ValueError(f"vector of id {id_} is not float32-representable (device storage is float32)")
            vecs.append(v32)
        if vecs:
            self._idx.add(np.stack(vecs))
            self._ids.extend(int(id_) for id_, _ in items)

    def Add(self, id: int, vec) -> None:                           # graph.go:459-464
        self.AddBatch([(id, vec)])

    def Len(self) -> int:                                          # graph.go:915-918
        return len(self._ids)

    def GetVector(self, idx: int):                                 # graph.go:897-901
        return self._idx.get_row(idx).astype(np.float64) if 0 <= idx < len(self._ids) else None

    def Search(self, query, k: int, strict: bool = False) -> List[int]:   # graph.go:467-488
        """Stored vectors must be float32-representable (AddBatch refuses others).  A QUERY that is not — the reference's own
        table queries (0.1, 0.1), graph_test.go:10-28 — is rounded to the nearest float32 before the device call and a
        RuntimeWarning says so (never silently): distances then differ from the reference's by at most one float32 rounding of
        each query component, which can reorder only near-ties.  strict=True refuses instead, like AddBatch."""
        q = np.asarray(query, dtype=np.float64).ravel()
        if q.size != self.dim:
            raise ValueError(f"query dimension mismatch: got {q.size}, want {self.dim}")
        n = len(self._ids)
        if n == 0 or k <= 0:
            return []
        k = min(k, n)                                              # :478-480
        q32 = q.astype(np.float32)
        if not np.array_equal(q32.astype(np.float64), q):
            if strict:
                raise ValueError("query vector is not float32-representable (device arithmetic starts from float32 values)")
            import warnings
            warnings.warn("arrowindex.Graph.Search: query rounded to float32 for the device (max component change %.3g)"
                          % float(np.max(np.abs(q32.astype(np.float64) - q))), RuntimeWarning, stacklevel=2)
        rows, _, count = self._idx.search(q32, k)                  # exhaustiveSearch (:490-506): every node, nearest k
        return [self._ids[int(rows[0, i])] for i in range(int(count[0]))]
