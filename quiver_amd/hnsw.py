"""Mirror of pkg/hnsw/hnsw.go: the graph lives on the host (C++ quiver::HNSW), every
distance is a libqv device call — one qv_distance_rows batch per searchLayer hop
(hnsw.go:536-563)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from ._host import Results, check, f32, hlib
from ._lib import metric_id

DefaultM, DefaultEfConstruction, DefaultEfSearch, DefaultMaxLevel = 16, 200, 100, 16   # hnsw.go:16-25


@dataclass
class Config:                          # hnsw.go:27-41
    M: int = 0
    MaxM0: int = 0
    EfConstruction: int = 0
    EfSearch: int = 0
    MaxLevel: int = 0
    DistanceFunc: object = "hnsw_euclidean"
    Seed: int = 1                      # the reference: time.Now().UnixNano() (hnsw.go:248)


@dataclass
class Result:                          # hnsw.go:87-95
    VectorID: str
    Distance: float
    VectorIndex: int


def _mid(df) -> int:
    if isinstance(df, (int, str)):
        return metric_id(df)
    return getattr(df, "metric_id")


class HNSW:
    def __init__(self, config: Config = None, device: int = 0):
        c = config or Config()
        self._h = hlib().qvh_hnsw_new(_mid(c.DistanceFunc), device, c.M, c.MaxM0, c.EfConstruction, c.EfSearch, c.MaxLevel, c.Seed)

    def __del__(self):
        try:
            hlib().qvh_hnsw_free(self._h)
        except Exception:
            pass

    def Insert(self, id: str, vector) -> None:
        v = f32(vector)
        check(hlib().qvh_hnsw_insert(self._h, id.encode(), v.ctypes.data, v.size))

    def InsertBatch(self, ids, vectors, batch_max: int = 16384, ramp_div: int = 16) -> None:
        """n Inserts connected on the device (qv_graph_insert): searches against the graph before each batch, links applied
        in id order; batch_max = 1 is the sequential graph of n Insert calls"""
        vs = np.ascontiguousarray(vectors, dtype=np.float32)
        if vs.ndim != 2 or vs.shape[0] != len(ids):
            raise ValueError("vectors must be [len(ids), dim]")
        cids = (C.c_char_p * len(ids))(*[i.encode() for i in ids])
        check(hlib().qvh_hnsw_insert_batch(self._h, cids, vs.ctypes.data, vs.shape[1], vs.shape[0], batch_max, ramp_div))

    def built_on_device(self) -> bool:
        return bool(hlib().qvh_hnsw_built_on_device(self._h))

    def Delete(self, id: str) -> None:
        check(hlib().qvh_hnsw_delete(self._h, id.encode()))

    def Search(self, query, k: int):
        q = f32(query)
        r = Results()
        idx = np.zeros(max(k, 1) + 8, dtype=np.uint32)
        check(hlib().qvh_hnsw_search(self._h, q.ctypes.data, q.size, k, r.h, idx.ctypes.data))
        return [Result(i, d, int(idx[j])) for j, (i, d) in enumerate(r.list())]

    def SearchBatch(self, queries, k: int, with_evals: bool = False):
        """nq searches walked on the device (one wavefront per query); same results as Search()"""
        qs = np.ascontiguousarray(queries, dtype=np.float32)
        if qs.ndim == 1:
            qs = qs[None, :]
        nq, n = qs.shape
        r = Results()
        idx = np.zeros((nq, max(k, 1)), dtype=np.uint32)
        ev = np.zeros(nq, dtype=np.uint32)
        check(hlib().qvh_hnsw_search_batch(self._h, qs.ctypes.data, n, nq, k, r.h, idx.ctypes.data, ev.ctypes.data))
        many, _ = r.many()
        out = [[Result(i, d, int(idx[q, j])) for j, (i, d) in enumerate(many[q])] for q in range(nq)]
        return (out, ev) if with_evals else out

    def search_batch_raw(self, queries, k: int):
        """the device traversal alone: (rows [nq,k] uint32, dist [nq,k], count [nq], evals [nq], seconds in qv_graph_search)"""
        qs = np.ascontiguousarray(queries, dtype=np.float32)
        nq, n = qs.shape
        rows = np.empty((nq, k), np.uint32); dist = np.empty((nq, k), np.float32)
        cnt = np.empty(nq, np.uint32); ev = np.empty(nq, np.uint32)
        sec = C.c_double(0)
        check(hlib().qvh_hnsw_search_batch_raw(self._h, qs.ctypes.data, n, nq, k, rows.ctypes.data, dist.ctypes.data, cnt.ctypes.data, ev.ctypes.data, C.byref(sec)))
        return rows, dist, cnt, ev, float(sec.value)

    def device_fallbacks(self) -> int:
        return int(hlib().qvh_hnsw_device_fallbacks(self._h))

    def topups(self) -> int:
        return int(hlib().qvh_hnsw_topups(self._h))

    def Size(self) -> int:
        return hlib().qvh_hnsw_size(self._h)

    # introspection (graph-equality tests against the oracle)
    def nodes(self) -> int:
        return hlib().qvh_hnsw_nodes(self._h)

    def node_level(self, n: int) -> int:
        return hlib().qvh_hnsw_node_level(self._h, n)

    def links(self, n: int, level: int) -> np.ndarray:
        out = np.empty(4096, dtype=np.uint32)
        c = hlib().qvh_hnsw_links(self._h, n, level, out.ctypes.data, out.size)
        return out[: max(c, 0)].copy()

    def entry_point(self):
        ep, lv = C.c_uint32(0), C.c_int(0)
        hlib().qvh_hnsw_entry_point(self._h, C.byref(ep), C.byref(lv))
        return int(ep.value), int(lv.value)

    def set_ef_search(self, ef: int):
        hlib().qvh_hnsw_set_ef_search(self._h, ef)

    def distance_calls(self) -> int:
        return int(hlib().qvh_hnsw_distance_calls(self._h))

    def distance_evals(self) -> int:
        return int(hlib().qvh_hnsw_distance_evals(self._h))


def NewHNSW(config: Config) -> HNSW:   # hnsw.go:221-252
    return HNSW(config)
