"""ctypes binding of oracle/libqvoracle.so — TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never
by quiver_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# QV_ORACLE_LIB: another build of the same sources (tests/c/Makefile's oracle_asan: the sanitizer build, checker of the checker)
_SO = os.environ.get("QV_ORACLE_LIB") or os.path.join(ROOT, "oracle", "libqvoracle.so")

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    lib.qvo_distance.restype = C.c_float
    lib.qvo_distance.argtypes = [C.c_int, _f32p, _f32p, C.c_uint32]
    lib.qvo_gen_rows.restype = None
    lib.qvo_gen_rows.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, _f32p]
    lib.qvo_exact_search.restype = C.c_int64
    lib.qvo_exact_search.argtypes = [C.c_int, _f32p, C.c_void_p, C.c_uint32, C.c_uint32, _f32p, C.c_uint32, _u32p, _f32p]
    lib.qvo_all_distances.restype = None
    lib.qvo_all_distances.argtypes = [C.c_int, _f32p, C.c_uint32, C.c_uint32, _f32p, _f32p]
    lib.qvo_exact_search_negative.restype = C.c_int64
    lib.qvo_exact_search_negative.argtypes = [C.c_int, _f32p, C.c_void_p, C.c_uint32, C.c_uint32, _f32p, _f32p, C.c_float,
                                              C.c_uint32, C.c_void_p, _u32p, _f32p]
    lib.qvo_faithful_create.restype = C.c_void_p
    lib.qvo_faithful_create.argtypes = [C.c_int, C.c_uint32]
    lib.qvo_faithful_destroy.argtypes = [C.c_void_p]
    lib.qvo_faithful_insert.argtypes = [C.c_void_p, C.c_char_p, _f32p]
    lib.qvo_faithful_size.restype = C.c_uint32
    lib.qvo_faithful_size.argtypes = [C.c_void_p]
    lib.qvo_faithful_search.restype = C.c_int64
    lib.qvo_faithful_search.argtypes = [C.c_void_p, _f32p, C.c_uint32, C.POINTER(C.c_char_p), _f32p]
    lib.qvo_faithful_search_many.restype = C.c_double
    lib.qvo_faithful_search_many.argtypes = [C.c_void_p, C.c_uint32, _f32p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    lib.qvo_opt_create.restype = C.c_void_p
    lib.qvo_opt_create.argtypes = [_f32p, C.c_uint32, C.c_uint32, C.c_int]
    lib.qvo_opt_destroy.argtypes = [C.c_void_p]
    lib.qvo_opt_cosine_scan.restype = C.c_double
    lib.qvo_opt_cosine_scan.argtypes = [C.c_void_p, _f32p, C.c_uint32, C.c_uint32, _u32p, _f32p]
    lib.qvo_opt_simd_bits.restype = C.c_int
    lib.qvo_hnsw_create.restype = C.c_void_p
    lib.qvo_hnsw_create.argtypes = [C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64]
    lib.qvo_hnsw_destroy.argtypes = [C.c_void_p]
    lib.qvo_hnsw_insert.restype = C.c_int64
    lib.qvo_hnsw_insert.argtypes = [C.c_void_p, _f32p]
    lib.qvo_hnsw_delete.argtypes = [C.c_void_p, C.c_uint32]
    lib.qvo_hnsw_search.restype = C.c_int64
    lib.qvo_hnsw_search.argtypes = [C.c_void_p, _f32p, C.c_uint32, _u32p, _f32p, C.POINTER(C.c_uint64)]
    lib.qvo_hnsw_set_ef_search.argtypes = [C.c_void_p, C.c_int]
    lib.qvo_hnsw_size.restype = C.c_uint32
    lib.qvo_hnsw_size.argtypes = [C.c_void_p]
    lib.qvo_hnsw_nodes.restype = C.c_uint32
    lib.qvo_hnsw_nodes.argtypes = [C.c_void_p]
    lib.qvo_hnsw_entry_point.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    lib.qvo_hnsw_node_level.argtypes = [C.c_void_p, C.c_uint32]
    lib.qvo_hnsw_links.argtypes = [C.c_void_p, C.c_uint32, C.c_int, _u32p, C.c_uint32]
    lib.qvo_hnsw_random_level.argtypes = [C.c_void_p]
    lib.qvo_hnsw_load_flat.argtypes = [C.c_void_p, C.c_uint32, _f32p, _u32p, _u32p, C.c_uint32, C.c_uint32]
    lib.qvo_hnsw_load_flat.restype = C.c_int
    lib.qvo_hnsw_insert_batch.restype = C.c_int64
    lib.qvo_hnsw_insert_batch.argtypes = [C.c_void_p, _f32p, C.c_uint32]
    _i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")
    lib.qvo_hnsw_load_graph.restype = C.c_int
    lib.qvo_hnsw_load_graph.argtypes = [C.c_void_p, C.c_uint32, _f32p, _i8p, C.c_uint32, C.c_uint32, _u32p, _u32p, _u32p, _u32p,
                                        C.c_uint32, C.c_int]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def distance(metric: int, a, b) -> np.float32:
    a, b = _f32(a), _f32(b)
    if a.shape != b.shape:
        raise ValueError("vectors must have the same length")
    return np.float32(lib().qvo_distance(metric, a, b, a.size))


def gen_rows(seed: int, row0: int, n: int, dim: int) -> np.ndarray:
    out = np.empty((n, dim), dtype=np.float32)
    lib().qvo_gen_rows(seed, row0, n, dim, out)
    return out


def all_distances(metric: int, rows, query) -> np.ndarray:
    rows, query = _f32(rows), _f32(query)
    out = np.empty(rows.shape[0], dtype=np.float32)
    lib().qvo_all_distances(metric, rows, rows.shape[0], rows.shape[1], query, out)
    return out


def exact_search(metric: int, rows, query, k: int, alive=None):
    rows, query = _f32(rows), _f32(query)
    n, dim = rows.shape if rows.ndim == 2 else (0, query.size)
    if n and query.size != dim:
        raise ValueError(f"query dimension mismatch: expected {dim}, got {query.size}")
    ro = np.empty(max(k, 1), dtype=np.uint32)
    do = np.empty(max(k, 1), dtype=np.float32)
    ap = None
    if alive is not None:
        alive = np.ascontiguousarray(alive, dtype=np.uint8)
        ap = alive.ctypes.data_as(C.c_void_p)
    if n == 0:
        return ro[:0], do[:0]
    got = lib().qvo_exact_search(metric, rows, ap, n, dim, query, max(k, 0), ro, do)
    if got < 0:
        raise ValueError("k must be positive")
    return ro[:got].copy(), do[:got].copy()


def exact_search_negative(metric: int, rows, query, negative, weight: float, k: int, alive=None, id_rank=None):
    rows, query, negative = _f32(rows), _f32(query), _f32(negative)
    n, dim = rows.shape
    cap = max(2 * k, 30)
    ro = np.empty(cap, dtype=np.uint32)
    do = np.empty(cap, dtype=np.float32)
    ap = None
    if alive is not None:
        alive = np.ascontiguousarray(alive, dtype=np.uint8)
        ap = alive.ctypes.data_as(C.c_void_p)
    rp = None
    if id_rank is not None:
        id_rank = np.ascontiguousarray(id_rank, dtype=np.uint32)
        rp = id_rank.ctypes.data_as(C.c_void_p)
    got = lib().qvo_exact_search_negative(metric, rows, ap, n, dim, query, negative, weight, k, rp, ro, do)
    if got < 0:
        raise ValueError("k must be positive")
    return ro[:got].copy(), do[:got].copy()


class OptScan:
    """the optimised (NOT reference-faithful) CPU cosine scan: per-thread NUMA-local slices, cached norms, SIMD lanes"""

    def __init__(self, rows, threads: int):
        rows = _f32(rows)
        self.threads = threads
        self._h = lib().qvo_opt_create(rows, rows.shape[0], rows.shape[1], threads)

    def search(self, queries, k: int):
        """-> (wall seconds of the scans, rows [nq, k], dist [nq, k])"""
        qs = _f32(queries)
        if qs.ndim == 1:
            qs = qs[None, :]
        ro = np.empty((qs.shape[0], k), dtype=np.uint32); do = np.empty((qs.shape[0], k), dtype=np.float32)
        dt = lib().qvo_opt_cosine_scan(self._h, qs, qs.shape[0], k, ro, do)
        return float(dt), ro, do

    def __del__(self):
        try:
            lib().qvo_opt_destroy(self._h)
        except Exception:
            pass


class Faithful:
    """reference-faithful ExactIndex (CPU baseline)"""

    def __init__(self, metric: int, dim: int):
        self._h = lib().qvo_faithful_create(metric, dim)
        self.dim = dim

    def insert(self, id_: str, vec) -> None:
        if lib().qvo_faithful_insert(self._h, id_.encode(), _f32(vec)) != 0:
            raise ValueError(f"vector with ID {id_} already exists")

    def size(self) -> int:
        return lib().qvo_faithful_size(self._h)

    def search(self, query, k: int):
        ids = (C.c_char_p * max(k, 1))()
        do = np.empty(max(k, 1), dtype=np.float32)
        got = lib().qvo_faithful_search(self._h, _f32(query), k, ids, do)
        if got < 0:
            raise ValueError("k must be positive")
        return [ids[i].decode() for i in range(got)], do[:got].copy()

    def search_many(self, queries, k: int, threads: int):
        """nq faithful searches on `threads` threads (one whole search per thread at a time) -> (wall seconds, dist [nq, k])"""
        qs = _f32(queries)
        out = np.empty((qs.shape[0], k), dtype=np.float32)
        dt = lib().qvo_faithful_search_many(self._h, self.dim, qs, qs.shape[0], k, threads, out.ctypes.data_as(C.c_void_p))
        return float(dt), out

    def __del__(self):
        try:
            lib().qvo_faithful_destroy(self._h)
        except Exception:
            pass


class HNSW:
    """oracle restatement of pkg/hnsw/hnsw.go"""

    def __init__(self, metric: int, dim: int, M=16, maxM0=0, efConstruction=200, efSearch=100, maxLevel=16, seed=1):
        self._h = lib().qvo_hnsw_create(metric, dim, M, maxM0, efConstruction, efSearch, maxLevel, seed)
        self.dim = dim

    def insert(self, vec) -> int:
        return int(lib().qvo_hnsw_insert(self._h, _f32(vec)))

    def insert_batch(self, vecs) -> int:
        """n Inserts at once: searches against the graph before the batch, links applied in node order"""
        vecs = _f32(vecs)
        assert vecs.ndim == 2 and vecs.shape[1] == self.dim
        r = int(lib().qvo_hnsw_insert_batch(self._h, vecs, vecs.shape[0]))
        if r < 0:
            raise RuntimeError("insert_batch failed")
        return r

    def delete(self, node: int) -> int:
        return lib().qvo_hnsw_delete(self._h, node)

    def set_ef_search(self, ef: int):
        lib().qvo_hnsw_set_ef_search(self._h, ef)

    def search(self, query, k: int, with_evals=False):
        ro = np.empty(max(k, 1), dtype=np.uint32)
        do = np.empty(max(k, 1), dtype=np.float32)
        ne = C.c_uint64(0)
        got = lib().qvo_hnsw_search(self._h, _f32(query), max(k, 0), ro, do, C.byref(ne))
        if got == -1:
            raise ValueError("k must be positive")
        if got < 0:
            raise RuntimeError("search failed")
        if with_evals:
            return ro[:got].copy(), do[:got].copy(), int(ne.value)
        return ro[:got].copy(), do[:got].copy()

    def size(self) -> int:
        return lib().qvo_hnsw_size(self._h)

    def nodes(self) -> int:
        return lib().qvo_hnsw_nodes(self._h)

    def entry_point(self):
        ep, lv = C.c_uint32(0), C.c_int(0)
        lib().qvo_hnsw_entry_point(self._h, C.byref(ep), C.byref(lv))
        return int(ep.value), int(lv.value)

    def node_level(self, node: int) -> int:
        return lib().qvo_hnsw_node_level(self._h, node)

    def links(self, node: int, level: int) -> np.ndarray:
        out = np.empty(4096, dtype=np.uint32)
        n = lib().qvo_hnsw_links(self._h, node, level, out, out.size)
        if n < 0:
            return out[:0]
        return out[:n].copy()

    def random_level(self) -> int:
        return lib().qvo_hnsw_random_level(self._h)

    def load_flat(self, rows, deg, links, entry: int):
        """test scaffolding: install a ready-made single-layer graph (rows are borrowed, kept alive here)"""
        self._rows = np.ascontiguousarray(rows, dtype=np.float32)
        deg = np.ascontiguousarray(deg, dtype=np.uint32); links = np.ascontiguousarray(links, dtype=np.uint32)
        assert self._rows.shape == (deg.size, self.dim) and links.shape[0] == deg.size
        if lib().qvo_hnsw_load_flat(self._h, deg.size, self._rows, deg, links, links.shape[1], entry) != 0:
            raise RuntimeError("load_flat failed (index not empty / bad entry)")

    def load_graph(self, rows, levels, max_m0, max_m, l0_deg, l0_links, up_off, up_links, entry: int, cur_level: int):
        """test scaffolding: install a multi-level graph in qv_graph_export's flat form (rows borrowed, kept alive here)"""
        self._rows = np.ascontiguousarray(rows, dtype=np.float32)
        levels = np.ascontiguousarray(levels, dtype=np.int8)
        l0_deg = np.ascontiguousarray(l0_deg, dtype=np.uint32); l0_links = np.ascontiguousarray(l0_links, dtype=np.uint32)
        up_off = np.ascontiguousarray(up_off, dtype=np.uint32); up_links = np.ascontiguousarray(up_links, dtype=np.uint32)
        if up_links.size == 0:
            up_links = np.zeros(1 + max_m, dtype=np.uint32)
        n = levels.size
        assert self._rows.shape == (n, self.dim) and l0_deg.size == n and l0_links.size == n * max_m0 and up_off.size == n
        if lib().qvo_hnsw_load_graph(self._h, n, self._rows, levels, max_m0, max_m, l0_deg, l0_links.reshape(-1), up_off,
                                     up_links.reshape(-1), entry, cur_level) != 0:
            raise RuntimeError("load_graph failed (index not empty / bad entry)")

    def export_flat(self, max_m0: int, max_m: int):
        """the graph in qv_graph_export's flat form: (levels, l0_deg, l0_links[n][max_m0], up_off, up_links[blocks][1+max_m])"""
        n = self.nodes()
        levels = np.array([self.node_level(i) for i in range(n)], dtype=np.int8)
        l0_deg = np.zeros(n, dtype=np.uint32); l0_links = np.zeros((n, max_m0), dtype=np.uint32)
        up_off = np.zeros(n, dtype=np.uint32); blocks = []
        for i in range(n):
            if levels[i] < 0:
                continue
            l = self.links(i, 0); l0_deg[i] = l.size; l0_links[i, :l.size] = l
            if levels[i] >= 1:
                up_off[i] = len(blocks)
                for lv in range(1, levels[i] + 1):
                    l = self.links(i, lv); b = np.zeros(1 + max_m, dtype=np.uint32); b[0] = l.size; b[1:1 + l.size] = l
                    blocks.append(b)
        up_links = np.stack(blocks) if blocks else np.zeros((0, 1 + max_m), dtype=np.uint32)
        return levels, l0_deg, l0_links, up_off, up_links

    def __del__(self):
        try:
            lib().qvo_hnsw_destroy(self._h)
        except Exception:
            pass
