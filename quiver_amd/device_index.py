"""DeviceIndex — a row-numbered, device-resident exact index: the Python face of one
``qv_index`` (one GPU, one shard).  String ids live one layer up (quiver_amd.hybrid)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib, metric_id


def _f32c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


class DeviceIndex:
    def __init__(self, dim: int, metric="cosine", device: int = 0, rowmajor: bool = False):
        self._h = C.c_void_p()
        self.dim = int(dim)
        self.metric = metric_id(metric)
        self.device = device
        check(lib().qv_index_create(C.byref(self._h), self.dim, self.metric, device, _lib.QV_FLAG_ROWMAJOR if rowmajor else 0))

    # ---- lifecycle ----
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().qv_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def reserve(self, rows: int):
        check(lib().qv_index_reserve(self._h, rows))

    # ---- mutation ----
    def add(self, rows) -> int:
        rows = _f32c(rows)
        if rows.ndim == 1:
            rows = rows[None, :]
        if rows.shape[1] != self.dim:
            raise ValueError(f"vector dimension mismatch: expected {self.dim}, got {rows.shape[1]}")
        first = C.c_uint32(0)
        check(lib().qv_index_add(self._h, rows.ctypes.data, rows.shape[0], C.byref(first)))
        return int(first.value)

    def add_device(self, d_ptr: int, n: int, stream: int = 0) -> int:
        first = C.c_uint32(0)
        check(lib().qv_index_add_device(self._h, d_ptr, n, C.byref(first), stream))
        return int(first.value)

    def add_synthetic(self, seed: int, gen_row0: int, n: int) -> int:
        first = C.c_uint32(0)
        check(lib().qv_index_add_synthetic(self._h, seed, gen_row0, n, C.byref(first)))
        return int(first.value)

    def remove(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint32).ravel()
        check(lib().qv_index_remove(self._h, rows.ctypes.data, rows.size))

    def update(self, row: int, vec):
        vec = _f32c(vec)
        if vec.size != self.dim:
            raise ValueError(f"vector dimension mismatch: expected {self.dim}, got {vec.size}")
        check(lib().qv_index_update(self._h, row, vec.ctypes.data))

    # ---- queries ----
    def rows(self) -> int:
        return int(lib().qv_index_rows(self._h))

    def size(self) -> int:
        return int(lib().qv_index_size(self._h))

    def __len__(self):
        return self.size()

    def get_row(self, row: int) -> np.ndarray:
        out = np.empty(self.dim, dtype=np.float32)
        check(lib().qv_index_get_row(self._h, row, out.ctypes.data))
        return out

    def search(self, queries, k: int, batched: bool = False):
        """-> (rows [nq,k] uint32, dist [nq,k] float32, count [nq] uint32)"""
        q = _f32c(queries)
        if q.ndim == 1:
            q = q[None, :]
        if q.shape[1] != self.dim:
            raise ValueError(f"query dimension mismatch: expected {self.dim}, got {q.shape[1]}")
        nq = q.shape[0]
        kk = max(int(k), 0)
        rows = np.full((nq, max(kk, 1)), 0xFFFFFFFF, dtype=np.uint32)
        dist = np.full((nq, max(kk, 1)), np.inf, dtype=np.float32)
        count = np.zeros(nq, dtype=np.uint32)
        fn = lib().qv_index_search_batched if batched else lib().qv_index_search
        check(fn(self._h, q.ctypes.data, nq, kk, rows.ctypes.data, dist.ctypes.data, count.ctypes.data))
        return rows[:, :kk], dist[:, :kk], count

    def search_device(self, d_queries: int, nq: int, k: int, d_rows_out: int, d_dist_out: int, stream: int = 0):
        check(lib().qv_index_search_device(self._h, d_queries, nq, k, d_rows_out, d_dist_out, stream))

    def search_masked(self, queries, k: int, mask):
        """exact top-k among the rows selected by `mask` (bool [rows] or packed uint64 words): filtered search without a full ranking"""
        q = _f32c(queries)
        if q.ndim == 1:
            q = q[None, :]
        if q.shape[1] != self.dim:
            raise ValueError("query dimension %d does not match index dimension %d" % (q.shape[1], self.dim))
        m = np.asarray(mask)
        words = (self.rows() + 63) // 64
        if m.dtype == np.bool_:
            if m.size != self.rows():
                raise ValueError("mask must have one entry per row")
            pad = np.zeros(words * 64, dtype=np.uint8); pad[:m.size] = m
            m = np.packbits(pad, bitorder="little").view(np.uint64)
        m = np.ascontiguousarray(m, dtype=np.uint64)
        if m.size != words:
            raise ValueError("mask must have ceil(rows/64) words")
        nq = q.shape[0]
        rows = np.empty((nq, max(k, 1)), dtype=np.uint32); dist = np.empty((nq, max(k, 1)), dtype=np.float32); cnt = np.empty(nq, dtype=np.uint32)
        check(lib().qv_index_search_masked(self._h, q.ctypes.data, nq, k, m.ctypes.data, rows.ctypes.data, dist.ctypes.data, cnt.ctypes.data))
        return rows, dist, cnt

    def search_batched_device(self, d_queries: int, nq: int, k: int, d_rows_out: int, d_dist_out: int, d_redo_flags: int, stream: int = 0):
        check(lib().qv_index_search_batched_device(self._h, d_queries, nq, k, d_rows_out, d_dist_out, d_redo_flags, stream))

    def distance_rows(self, query, rows) -> np.ndarray:
        q = _f32c(query)
        if q.size != self.dim:
            raise ValueError(f"query dimension mismatch: expected {self.dim}, got {q.size}")
        r = np.ascontiguousarray(rows, dtype=np.uint32).ravel()
        out = np.empty(r.size, dtype=np.float32)
        check(lib().qv_distance_rows(self._h, q.ctypes.data, r.ctypes.data, r.size, out.ctypes.data))
        return out

    def distance_rows_device(self, d_query: int, d_rows: int, n: int, d_out: int, stream: int = 0):
        check(lib().qv_distance_rows_device(self._h, d_query, d_rows, n, d_out, stream))


    def profile(self, enable: bool):
        check(lib().qv_index_profile(self._h, 1 if enable else 0))

    def profile_read(self):
        ms, n = C.c_double(0), C.c_uint64(0)
        check(lib().qv_index_profile_read(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)


class DeviceGraph:
    """An HNSW graph resident on the device over the rows of a DeviceIndex (qv_graph_* of include/qv.h):
    the whole of HNSW.Search (pkg/hnsw/hnsw.go:602-672) runs on the GPU, one wavefront per query.

    levels [n] int8 (-1 = tombstone), l0_deg [n], l0_links [n, max_m0]; upper levels as up_off [n] and
    up_links [n_blocks, 1 + max_m] (degree, links) — pass None for a single-layer graph."""

    def __init__(self, index: DeviceIndex, levels, l0_deg, l0_links, entry: int, cur_level: int = 0, up_off=None, up_links=None, max_m: int = 0):
        self.index = index
        levels = np.ascontiguousarray(levels, dtype=np.int8)
        l0_deg = np.ascontiguousarray(l0_deg, dtype=np.uint32)
        l0_links = np.ascontiguousarray(l0_links, dtype=np.uint32)
        n = levels.size
        if l0_deg.size != n or l0_links.ndim != 2 or l0_links.shape[0] != n:
            raise ValueError("levels, l0_deg and l0_links must describe the same nodes")
        if up_off is None:
            up_off = np.zeros(n, dtype=np.uint32); up_links = np.zeros((1, 1 + max(max_m, 1)), dtype=np.uint32); n_blocks = 0
        else:
            up_off = np.ascontiguousarray(up_off, dtype=np.uint32); up_links = np.ascontiguousarray(up_links, dtype=np.uint32); n_blocks = up_links.shape[0]
            max_m = up_links.shape[1] - 1
        self._g = C.c_void_p()
        check(lib().qv_graph_create(C.byref(self._g), index.handle, n, levels.ctypes.data, l0_links.shape[1], max(max_m, 1), l0_deg.ctypes.data,
                                    l0_links.ctypes.data, up_off.ctypes.data, up_links.ctypes.data, n_blocks, int(entry), int(cur_level)))

    def search(self, queries, k: int, ef_search: int, with_evals: bool = False):
        """rows [nq, k] uint32 (0xFFFFFFFF = unfilled), dist [nq, k] float32, count [nq] (< k: the graph search
        under-filled and the caller tops up like hnsw.go:676-710)"""
        q = _f32c(queries)
        if q.ndim == 1:
            q = q[None, :]
        if q.shape[1] != self.index.dim:
            raise ValueError("query dimension %d does not match index dimension %d" % (q.shape[1], self.index.dim))
        nq = q.shape[0]
        rows = np.empty((nq, max(k, 1)), dtype=np.uint32); dist = np.empty((nq, max(k, 1)), dtype=np.float32)
        count = np.empty(nq, dtype=np.uint32); evals = np.empty(nq, dtype=np.uint32)
        check(lib().qv_graph_search(self._g, q.ctypes.data, nq, k, ef_search, rows.ctypes.data, dist.ctypes.data, count.ctypes.data, evals.ctypes.data))
        return (rows, dist, count, evals) if with_evals else (rows, dist, count)

    def search_device(self, d_queries: int, nq: int, k: int, ef_search: int, d_rows_out: int, d_dist_out: int, d_count_out: int,
                      d_evals_out: int = 0, stream: int = 0):
        """device pointers in and out, enqueued on `stream`, no sync; count 0xFFFFFFFE = redo that query through search()"""
        check(lib().qv_graph_search_device(self._g, d_queries, nq, k, ef_search, d_rows_out, d_dist_out, d_count_out, d_evals_out or None, stream or None))

    def close(self):
        if getattr(self, "_g", None) is not None and self._g.value:
            lib().qv_graph_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_topk_device(d_dist_lists: int, d_row_lists: int, n_lists: int, k: int, d_rows_out: int, d_dist_out: int, stream: int = 0):
    check(lib().qv_merge_topk_device(d_dist_lists, d_row_lists, n_lists, k, d_rows_out, d_dist_out, stream))


def merge_topk_shards_device(d_packed_lists: int, d_bases: int, n_lists: int, nq: int, k: int, d_rows_out: int, d_dist_out: int, stream: int = 0):
    """merge of the packed per-shard buffers ([n_lists][nq][2][k]: k local rows, k distance bits); outputs [nq][k], rows global"""
    check(lib().qv_merge_topk_shards_device(d_packed_lists, d_bases, n_lists, nq, k, d_rows_out, d_dist_out, stream or None))


def distance_pairs(metric, a, b, device: int = 0) -> np.ndarray:
    a, b = _f32c(a), _f32c(b)
    if a.ndim == 1:
        a = a[None, :]
    if b.ndim == 1:
        b = b[None, :]
    if a.shape != b.shape:
        raise ValueError("vectors must have the same length")  # distances.go:13-15
    out = np.empty(a.shape[0], dtype=np.float32)
    check(lib().qv_distance_pairs(metric_id(metric), a.ctypes.data, b.ctypes.data, a.shape[0], a.shape[1], out.ctypes.data, device))
    return out


def device_info(device: int = 0):
    name = C.create_string_buffer(256)
    cus, mem = C.c_int(0), C.c_uint64(0)
    check(lib().qv_device_info(device, name, 256, C.byref(cus), C.byref(mem)))
    return {"name": name.value.decode(), "cus": cus.value, "hbm_bytes": mem.value}
