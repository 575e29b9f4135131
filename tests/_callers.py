"""ctypes wrapper of quiver_amd/lib/libqvcallers.so (tools/native/qv_callers.cpp): T native threads calling the C ABI's
host-pointer search with one query per call, closed loop — the traffic the reference's Go host produces
(pkg/core/collection.go:647, pkg/core/db.go:805-828, pkg/hnsw/hnsw.go:602-606).  Measurement / test infrastructure."""
import ctypes as C
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PATH = os.path.join(_ROOT, "quiver_amd", "lib", "libqvcallers.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        import quiver_amd
        quiver_amd.load_library()                                   # libqv first (same copy for both)
        _lib = C.CDLL(_PATH)
        _lib.qvc_run.restype = C.c_int
        _lib.qvc_run.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_uint64,
                                 C.c_void_p, C.c_void_p, C.c_void_p] + [C.POINTER(C.c_uint64)] * 3 + [C.POINTER(C.c_double)] * 4 + [C.c_char_p, C.c_size_t]
    return _lib


KIND = {"index": 0, "sharded": 1, "graph": 2}


def run(kind, handle, queries, k, threads, seconds=1.0, max_calls_per_thread=0, ef=0):
    """-> dict(rows, dist, count [per query: first result recorded], calls, mismatches, errors, seconds, qps, p50_us, p99_us, max_us)"""
    q = np.ascontiguousarray(queries, dtype=np.float32)
    nq, dim = q.shape
    rows = np.zeros((nq, k), np.uint32); dist = np.zeros((nq, k), np.float32); count = np.zeros(nq, np.uint32)
    calls, mism, errs = C.c_uint64(), C.c_uint64(), C.c_uint64()
    el, p50, p99, mx = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    msg = C.create_string_buffer(512)
    rc = lib().qvc_run(KIND[kind], handle, q.ctypes.data, nq, dim, k, ef, threads, float(seconds), int(max_calls_per_thread),
                       rows.ctypes.data, dist.ctypes.data, count.ctypes.data, C.byref(calls), C.byref(mism), C.byref(errs),
                       C.byref(el), C.byref(p50), C.byref(p99), C.byref(mx), msg, 512)
    return dict(rc=rc, error=msg.value.decode(), rows=rows, dist=dist, count=count, calls=calls.value, mismatches=mism.value, errors=errs.value,
                seconds=el.value, qps=calls.value / max(el.value, 1e-9), p50_us=p50.value, p99_us=p99.value, max_us=mx.value)


def coalesce_stats(kind, handle):
    import quiver_amd
    v = (C.c_uint64 * 8)()
    fn = quiver_amd.lib().qv_index_coalesce_stats if kind == "index" else quiver_amd.lib().qv_graph_coalesce_stats
    fn(handle, v)
    return dict(zip(("solo", "led", "rode", "groups", "group_queries", "lingers", "linger_ns", "group_pass_ns"), [int(x) for x in v]))
