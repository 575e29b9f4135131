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
    default_filter = "auto"        # the filter kernel new indexes choose (set_filter); tests and bench.py compare the kernels through it

    def __init__(self, dim: int, metric="cosine", device: int = 0, rowmajor: bool = False, bf16_rows: bool = False, filter=None):
        self._h = C.c_void_p()
        self.dim = int(dim)
        self.metric = metric_id(metric)
        self.device = device
        check(lib().qv_index_create(C.byref(self._h), self.dim, self.metric, device, (_lib.QV_FLAG_ROWMAJOR if rowmajor else 0) | (_lib.QV_FLAG_BF16_ROWS if bf16_rows else 0)))
        f = filter if filter is not None else DeviceIndex.default_filter
        if f not in ("auto", 0):
            self.set_filter(f)

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

    def get_rows(self, rows) -> np.ndarray:
        """[n, dim]: the listed rows in one device pass (qv_index_get_rows)"""
        r = np.ascontiguousarray(rows, dtype=np.uint32).ravel()
        out = np.empty((r.size, self.dim), dtype=np.float32)
        check(lib().qv_index_get_rows(self._h, r.ctypes.data, r.size, out.ctypes.data))
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


    FILTERS = {"auto": 0, "fp32": 1, "bf16x3": 2, "bf16x1": 3, "off": 4}

    def set_filter(self, filter):
        """the batched path's filter kernel: "auto", "fp32" (fp32 MFMA chain), "bf16x3", "bf16x1", or "off" — exact scans only (qv_index_set_filter)"""
        check(lib().qv_index_set_filter(self._h, self.FILTERS.get(filter, filter)))

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
        self._g = C.c_void_p()
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

    @property
    def handle(self):
        return self._g

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

    # ---- device-resident construction (qv_graph_create_empty / qv_graph_insert / qv_graph_export) ----
    @classmethod
    def empty(cls, index: DeviceIndex, capacity: int, m: int = 16, max_m0: int = 0, ef_construction: int = 200) -> "DeviceGraph":
        """a graph to be built on the device over the rows of `index` (which must keep a row-major copy)"""
        self = cls.__new__(cls)
        self.index = index
        self._g = C.c_void_p()
        check(lib().qv_graph_create_empty(C.byref(self._g), index.handle, capacity, m, max_m0, ef_construction))
        return self

    @classmethod
    def build(cls, index: DeviceIndex, levels, m: int = 16, max_m0: int = 0, ef_construction: int = 200, batch_max: int = 16384,
              ramp_div: int = 16) -> "DeviceGraph":
        """hnsw.HNSW.Insert (hnsw.go:266-468) for rows 0..len(levels)-1 of `index`, in batches on the device"""
        levels = np.ascontiguousarray(levels, dtype=np.int8)
        self = cls.empty(index, levels.size, m, max_m0, ef_construction)
        self.insert(0, levels, batch_max, ramp_div)
        return self

    def make_buildable(self, ef_construction: int = 200):
        """score the links of an uploaded (host-built) graph once so that insert() can extend it on the device"""
        check(lib().qv_graph_make_buildable(self._g, ef_construction))

    def insert(self, first_row: int, levels, batch_max: int = 16384, ramp_div: int = 16):
        levels = np.ascontiguousarray(levels, dtype=np.int8)
        check(lib().qv_graph_insert(self._g, first_row, levels.size, levels.ctypes.data, batch_max, ramp_div))

    def info(self) -> dict:
        v = [C.c_uint32(0) for _ in range(5)]; lv = C.c_int(0)
        check(lib().qv_graph_info(self._g, *[C.byref(x) for x in v], C.byref(lv)))
        return {"n_nodes": v[0].value, "n_up_blocks": v[1].value, "max_m0": v[2].value, "max_m": v[3].value, "entry": v[4].value, "cur_level": lv.value}

    def stats(self) -> dict:
        sec = C.c_double(0); a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        check(lib().qv_graph_stats(self._g, C.byref(sec), C.byref(a), C.byref(b), C.byref(c)))
        return {"build_seconds": sec.value, "build_batches": a.value, "build_redo": b.value, "search_redo": c.value}

    def export(self):
        """(levels, l0_deg, l0_links [n, max_m0], up_off, up_links [blocks, 1 + max_m]) — the form qv_graph_create takes"""
        i = self.info()
        n, nb = i["n_nodes"], i["n_up_blocks"]
        levels = np.empty(n, dtype=np.int8); l0_deg = np.empty(n, dtype=np.uint32); l0_links = np.empty((n, i["max_m0"]), dtype=np.uint32)
        up_off = np.empty(n, dtype=np.uint32); up_links = np.zeros((nb, 1 + i["max_m"]), dtype=np.uint32)
        check(lib().qv_graph_export(self._g, levels.ctypes.data, l0_deg.ctypes.data, l0_links.ctypes.data, up_off.ctypes.data,
                                    up_links.ctypes.data if nb else None))
        return levels, l0_deg, l0_links, up_off, up_links

    def close(self):
        if getattr(self, "_g", None) is not None and self._g.value:
            lib().qv_graph_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GraphReplicas:
    """HNSW over the GPUs of a node: graph traversal is sequentially dependent (hnsw.go:471-580), so the graph does
    not shard by rows — REPLICAS ONLY: one complete index + graph per listed device, built from the same rows and levels (the
    device build is deterministic, so the replicas are identical), the queries of a batch cut into one contiguous slice per
    replica and walked concurrently.  A device may be listed more than once (two replicas on one GPU: how a 1-GPU box tests it)."""

    def __init__(self, rows, levels, metric="cosine", devices=(0,), m: int = 16, max_m0: int = 0, ef_construction: int = 200,
                 batch_max: int = 16384, ramp_div: int = 16):
        rows = _f32c(rows)
        self.dim = rows.shape[1]
        self.indexes, self.graphs = [], []
        for d in devices:
            idx = DeviceIndex(self.dim, metric, device=int(d), rowmajor=True)
            idx.add(rows)
            self.indexes.append(idx)
            self.graphs.append(DeviceGraph.build(idx, levels, m=m, max_m0=max_m0, ef_construction=ef_construction, batch_max=batch_max, ramp_div=ramp_div))

    def search(self, queries, k: int, ef_search: int):
        """-> (rows [nq, k], dist [nq, k], count [nq]): DeviceGraph.search of each query slice on its replica, concurrently"""
        from concurrent.futures import ThreadPoolExecutor
        q = _f32c(queries)
        if q.ndim == 1:
            q = q[None, :]
        nq, R = q.shape[0], len(self.graphs)
        cuts = [i * nq // R for i in range(R + 1)]
        jobs = [(g, q[cuts[i]:cuts[i + 1]]) for i, g in enumerate(self.graphs) if cuts[i + 1] > cuts[i]]
        with ThreadPoolExecutor(max_workers=max(len(jobs), 1)) as pool:           # ctypes drops the GIL: the replicas' calls overlap
            parts = list(pool.map(lambda job: job[0].search(job[1], k, ef_search), jobs))
        return tuple(np.concatenate([p[j] for p in parts]) for j in range(3))

    def close(self):
        for g in self.graphs:
            g.close()
        for i in self.indexes:
            i.close()
        self.graphs, self.indexes = [], []


class ShardedIndex:
    """One corpus over several GPUs behind ONE C-ABI handle (qv_sharded_* of include/qv.h): a shard (exact index) per device,
    one RCCL all-gather of the per-shard top-k per search, merge on the first device.  `devices` may repeat a device only
    with peer_copy=True (point-to-point exchange instead of the collective)."""

    def __init__(self, dim: int, metric="cosine", devices=(0,), rowmajor: bool = False, peer_copy: bool = False, bf16_rows: bool = False):
        self._h = C.c_void_p()
        self.dim = int(dim)
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        flags = (_lib.QV_FLAG_ROWMAJOR if rowmajor else 0) | (_lib.QV_SHARDED_PEER_COPY if peer_copy else 0) | (_lib.QV_FLAG_BF16_ROWS if bf16_rows else 0)
        check(lib().qv_sharded_create(C.byref(self._h), self.dim, metric_id(metric), devs, len(devices), flags))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().qv_sharded_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def shards(self) -> int:
        return int(lib().qv_sharded_shards(self._h))

    def size(self) -> int:
        return int(lib().qv_sharded_size(self._h))

    def shard_info(self, g: int) -> dict:
        dev = C.c_int(0); b, r, l = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        check(lib().qv_sharded_shard_info(self._h, g, C.byref(dev), C.byref(b), C.byref(r), C.byref(l)))
        return {"device": dev.value, "base": b.value, "rows": r.value, "live": l.value}

    def reserve(self, rows_total: int):
        check(lib().qv_sharded_reserve(self._h, rows_total))

    def add(self, rows) -> np.ndarray:
        """-> global row id of every added row"""
        rows = _f32c(rows)
        if rows.ndim == 1:
            rows = rows[None, :]
        if rows.shape[1] != self.dim:
            raise ValueError(f"vector dimension mismatch: expected {self.dim}, got {rows.shape[1]}")
        ids = np.empty(rows.shape[0], dtype=np.uint32)
        check(lib().qv_sharded_add(self._h, rows.ctypes.data, rows.shape[0], ids.ctypes.data))
        return ids

    def add_synthetic(self, seed: int, gen_row0: int, n: int):
        check(lib().qv_sharded_add_synthetic(self._h, seed, gen_row0, n))

    def remove(self, global_rows):
        r = np.ascontiguousarray(global_rows, dtype=np.uint32).ravel()
        check(lib().qv_sharded_remove(self._h, r.ctypes.data, r.size))

    def search(self, queries, k: int):
        q = _f32c(queries)
        if q.ndim == 1:
            q = q[None, :]
        if q.shape[1] != self.dim:
            raise ValueError(f"query dimension mismatch: expected {self.dim}, got {q.shape[1]}")
        nq, kk = q.shape[0], max(int(k), 0)
        rows = np.full((nq, max(kk, 1)), 0xFFFFFFFF, dtype=np.uint32); dist = np.full((nq, max(kk, 1)), np.inf, dtype=np.float32)
        count = np.zeros(nq, dtype=np.uint32)
        check(lib().qv_sharded_search(self._h, q.ctypes.data, nq, kk, rows.ctypes.data, dist.ctypes.data, count.ctypes.data))
        return rows[:, :kk], dist[:, :kk], count

    def search_device(self, d_queries: int, nq: int, k: int, d_rows_out: int, d_dist_out: int, stream: int = 0):
        check(lib().qv_sharded_search_device(self._h, d_queries, nq, k, d_rows_out, d_dist_out, stream or None))

    def _q(self, queries):
        q = _f32c(queries)
        if q.ndim == 1:
            q = q[None, :]
        if q.shape[1] != self.dim:
            raise ValueError(f"query dimension mismatch: expected {self.dim}, got {q.shape[1]}")
        return q

    def rows(self) -> int:
        return int(lib().qv_sharded_rows(self._h))

    def update(self, global_row: int, vec):
        vec = _f32c(vec)
        if vec.size != self.dim:
            raise ValueError(f"vector dimension mismatch: expected {self.dim}, got {vec.size}")
        check(lib().qv_sharded_update(self._h, int(global_row), vec.ctypes.data))

    def get_row(self, global_row: int) -> np.ndarray:
        out = np.empty(self.dim, dtype=np.float32)
        check(lib().qv_sharded_get_row(self._h, int(global_row), out.ctypes.data))
        return out

    def get_rows(self, global_rows) -> np.ndarray:
        r = np.ascontiguousarray(global_rows, dtype=np.uint32).ravel()
        out = np.empty((r.size, self.dim), dtype=np.float32)
        check(lib().qv_sharded_get_rows(self._h, r.ctypes.data, r.size, out.ctypes.data))
        return out

    def search_masked(self, queries, k: int, selected_global_rows):
        """exact top-k among the listed rows (global ids): the filtered search, qv_sharded_search_masked"""
        q = self._q(queries)
        sel = np.ascontiguousarray(selected_global_rows, dtype=np.uint32).ravel()
        nq, kk = q.shape[0], max(int(k), 0)
        rows = np.full((nq, max(kk, 1)), 0xFFFFFFFF, dtype=np.uint32); dist = np.full((nq, max(kk, 1)), np.inf, dtype=np.float32)
        count = np.zeros(nq, dtype=np.uint32)
        check(lib().qv_sharded_search_masked(self._h, q.ctypes.data, nq, kk, sel.ctypes.data, sel.size, rows.ctypes.data, dist.ctypes.data, count.ctypes.data))
        return rows[:, :kk], dist[:, :kk], count

    def search_negative(self, query, negative, k_fetch: int):
        """-> (rows [k_fetch], dist, neg_dist, count): qv_sharded_search_negative"""
        q, n = self._q(query), self._q(negative)
        kk = max(int(k_fetch), 0)
        rows = np.full(max(kk, 1), 0xFFFFFFFF, dtype=np.uint32); dist = np.full(max(kk, 1), np.inf, dtype=np.float32)
        nd = np.full(max(kk, 1), np.inf, dtype=np.float32); cnt = C.c_uint32(0)
        check(lib().qv_sharded_search_negative(self._h, q.ctypes.data, n.ctypes.data, kk, rows.ctypes.data, dist.ctypes.data, nd.ctypes.data, C.byref(cnt)))
        return rows[:kk], dist[:kk], nd[:kk], int(cnt.value)

    def distance_rows(self, query, global_rows) -> np.ndarray:
        q = self._q(query)
        r = np.ascontiguousarray(global_rows, dtype=np.uint32).ravel()
        out = np.empty(r.size, dtype=np.float32)
        check(lib().qv_sharded_distance_rows(self._h, q.ctypes.data, r.ctypes.data, r.size, out.ctypes.data))
        return out

    def set_filter(self, filter):
        check(lib().qv_sharded_set_filter(self._h, DeviceIndex.FILTERS.get(filter, filter)))

    def profile_read_shard(self, g: int):
        ms, n = C.c_double(0), C.c_uint64(0)
        check(lib().qv_sharded_profile_read_shard(self._h, g, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def sync(self):
        check(lib().qv_sharded_sync(self._h))

    def profile(self, enable: bool):
        check(lib().qv_sharded_profile(self._h, 1 if enable else 0))

    def profile_read(self) -> dict:
        a, b, c = C.c_double(0), C.c_double(0), C.c_double(0); n = C.c_uint64(0)
        check(lib().qv_sharded_profile_read(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        m = max(int(n.value), 1)
        return {"searches": int(n.value), "scan_ms": a.value / m, "exchange_ms": b.value / m, "merge_ms": c.value / m}


def sharded_plan_add(rows_per_shard, n: int) -> np.ndarray:
    have = np.ascontiguousarray(rows_per_shard, dtype=np.uint64)
    give = np.zeros(have.size, dtype=np.uint64)
    check(lib().qv_sharded_plan_add(have.ctypes.data, have.size, n, give.ctypes.data))
    return give


def graph_batch_size(nodes_linked: int, batch_max: int, ramp_div: int) -> int:
    return int(lib().qv_graph_batch_size(nodes_linked, batch_max, ramp_div))


def random_levels(n: int, max_level: int = 16, seed: int = 1) -> np.ndarray:
    """n draws of randomLevel (hnsw.go:716-738: p = 0.25 per extra level, at most min(MaxLevel, 10) draws, < MaxLevel) from a
    SplitMix64 stream — the level law stays on the host side of the boundary, where the reference keeps its RNG
    (hnsw.go:248 seeds it from the wall clock; a seed here makes builds repeatable)."""
    attempts = min(max_level, 10)
    out = np.empty(n, dtype=np.int8)
    filled, draw0 = 0, 0
    while filled < n:
        # a block of the draw stream u_j = f(seed + (j+1) * gamma); a node consumes draws up to its first failure (u >= 0.25)
        # or `attempts` successes, whichever comes first
        m = max(2 * (n - filled), 1024)
        j = np.arange(draw0 + 1, draw0 + m + 1, dtype=np.uint64)
        with np.errstate(over="ignore"):
            x = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + j * np.uint64(0x9E3779B97F4A7C15)
            x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            x = x ^ (x >> np.uint64(31))
        ok = ((x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)) < 0.25
        fails = np.flatnonzero(~ok)
        if fails.size == 0:
            raise RuntimeError("level draw stream has no failure in a block")
        runs = np.diff(np.concatenate(([-1], fails))) - 1          # successes before each failure
        if (runs >= attempts).any():                                # a run of >= `attempts` successes: walk the block draw by draw
            pos = 0
            while filled < n and pos < m:
                level = 0
                while level < attempts and pos < m and ok[pos]:
                    level += 1; pos += 1
                if level < attempts:
                    if pos >= m:
                        break                                       # node straddles the block: redo it in the next one
                    pos += 1                                        # the failing draw
                out[filled] = min(level, max_level - 1); filled += 1
                last_end = pos
            draw0 += last_end
            continue
        take = min(runs.size, n - filled)
        out[filled:filled + take] = np.minimum(runs[:take], max_level - 1)
        filled += take
        draw0 += int(fails[take - 1]) + 1
    return out


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


def runtime_info() -> str:
    """which HIP runtime and RCCL the process bound (qv_runtime_info)"""
    buf = C.create_string_buffer(1024)
    check(lib().qv_runtime_info(buf, 1024))
    return buf.value.decode()


def distance_pair(metric, a, b) -> float:
    """one DistanceFunc call on the host (qv_distance_pair): the kernels' per-pair arithmetic compiled for the CPU"""
    a, b = _f32c(a).ravel(), _f32c(b).ravel()
    if a.size != b.size:
        raise ValueError("vectors must have the same length")  # distances.go:13-15 (the reference panics)
    out = C.c_float(0)
    check(lib().qv_distance_pair(metric_id(metric), a.ctypes.data, b.ctypes.data, a.size, C.byref(out)))
    return float(np.float32(out.value))


def device_info(device: int = 0):
    name = C.create_string_buffer(256)
    cus, mem = C.c_int(0), C.c_uint64(0)
    check(lib().qv_device_info(device, name, 256, C.byref(cus), C.byref(mem)))
    return {"name": name.value.decode(), "cus": cus.value, "hbm_bytes": mem.value}
