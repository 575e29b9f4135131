"""ctypes binding of libqv.so (include/qv.h).  Fails loudly if the library is missing:
there is no Python/CPU fallback for any compute entry point."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QV_LIB_PATH") or os.path.join(_HERE, "lib", "libqv.so")   # QV_LIB_PATH: a measurement build of the library (tools/)

QV_OK = 0
QV_ERR_INVALID_ARG, QV_ERR_DIM_MISMATCH, QV_ERR_K_NOT_POSITIVE, QV_ERR_OUT_OF_RANGE = -1, -2, -3, -4
QV_ERR_NO_DEVICE, QV_ERR_DEVICE, QV_ERR_OOM, QV_ERR_UNSUPPORTED = -5, -6, -7, -8
QV_FLAG_ROWMAJOR = 1
QV_FLAG_BF16_ROWS = 2
QV_SHARDED_PEER_COPY = 1 << 32

# include/qv.h qv_metric
METRICS = {
    "cosine": 0, "euclidean": 1, "squared_euclidean": 2, "dot_product": 3, "manhattan": 4,
    "hnsw_cosine": 5, "hnsw_euclidean": 6, "hnsw_dot_product": 7, "arrow_squared_euclidean": 8,
    # short names (the qv_metric enumerators of include/qv.h)
    "l2": 1, "l2sq": 2, "dot": 3, "l1": 4, "cosine_f32": 5, "l2_f32": 6, "dot_f32": 7, "l2sq_f64": 8,
}


def metric_id(metric) -> int:
    if isinstance(metric, int):
        return metric
    # pkg/vectortypes/types.go:46-47: an unknown distance type falls back to cosine
    return METRICS.get(str(metric), 0)


class QvError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


_u32p = C.POINTER(C.c_uint32)
_f32p = C.POINTER(C.c_float)

# name -> (restype, argtypes); every symbol include/qv.h declares
PROTOTYPES = {
    "qv_index_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.c_int, C.c_int, C.c_uint64]),
    "qv_index_destroy": (None, [C.c_void_p]),
    "qv_index_reserve": (C.c_int, [C.c_void_p, C.c_uint64]),
    "qv_index_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, _u32p]),
    "qv_index_add_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, _u32p, C.c_void_p]),
    "qv_index_add_synthetic": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, _u32p]),
    "qv_index_remove": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "qv_index_update": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "qv_index_rows": (C.c_uint32, [C.c_void_p]),
    "qv_index_size": (C.c_uint32, [C.c_void_p]),
    "qv_index_dim": (C.c_uint32, [C.c_void_p]),
    "qv_index_metric": (C.c_int, [C.c_void_p]),
    "qv_index_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qv_index_coalesce_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "qv_graph_coalesce_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "qv_index_coalesce_early_rounds": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "qv_graph_coalesce_early_rounds": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "qv_index_search_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qv_index_search_masked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qv_index_search_negative": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, _u32p]),
    "qv_index_search_batched": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qv_index_search_batched_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qv_distance_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "qv_distance_rows_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "qv_distance_pair": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, _f32p]),
    "qv_distance_pairs": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int]),
    "qv_merge_topk_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qv_merge_topk_shards_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qv_index_set_filter": (C.c_int, [C.c_void_p, C.c_int]),
    "qv_sharded_set_filter": (C.c_int, [C.c_void_p, C.c_int]),
    "qv_index_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "qv_index_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "qv_graph_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]),
    "qv_graph_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qv_graph_search_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qv_graph_destroy": (None, [C.c_void_p]),
    "qv_graph_batch_size": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "qv_graph_create_empty": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "qv_graph_insert": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]),
    "qv_graph_make_buildable": (C.c_int, [C.c_void_p, C.c_uint32]),
    "qv_graph_build": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "qv_graph_info": (C.c_int, [C.c_void_p, _u32p, _u32p, _u32p, _u32p, _u32p, C.POINTER(C.c_int)]),
    "qv_graph_export": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qv_graph_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "qv_sharded_span": (C.c_uint32, [C.c_int]),
    "qv_sharded_plan_add": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p]),
    "qv_sharded_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.c_int, C.c_void_p, C.c_int, C.c_uint64]),
    "qv_sharded_destroy": (None, [C.c_void_p]),
    "qv_sharded_shards": (C.c_int, [C.c_void_p]),
    "qv_sharded_size": (C.c_uint64, [C.c_void_p]),
    "qv_sharded_dim": (C.c_uint32, [C.c_void_p]),
    "qv_sharded_shard_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), _u32p, _u32p, _u32p]),
    "qv_sharded_reserve": (C.c_int, [C.c_void_p, C.c_uint64]),
    "qv_sharded_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "qv_sharded_add_synthetic": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]),
    "qv_sharded_remove": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "qv_sharded_rows": (C.c_uint64, [C.c_void_p]),
    "qv_sharded_update": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "qv_sharded_get_row": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "qv_sharded_get_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "qv_sharded_search_masked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qv_sharded_search_negative": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, _u32p]),
    "qv_sharded_distance_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "qv_sharded_profile_read_shard": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "qv_sharded_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qv_sharded_search_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qv_sharded_sync": (C.c_int, [C.c_void_p]),
    "qv_sharded_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "qv_sharded_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "qv_index_get_row": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "qv_index_get_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "qv_last_error": (C.c_char_p, []),
    "qv_abi_version": (C.c_int, []),
    "qv_device_count": (C.c_int, []),
    "qv_runtime_info": (C.c_int, [C.c_char_p, C.c_size_t]),
    "qv_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
}

_lib = None


def load_library(path: str | None = None):
    """dlopen libqv.so and bind every prototype.  Raises (never falls back) when the
    library has not been built: run ``python -c 'import __graft_entry__ as g; g.build()'``."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError(f"{p} not found: build it with `make -C quiver_amd/csrc` (hipcc --offload-arch=gfx950); "
                          "quiver_amd has no CPU fallback")
    lib_ = C.CDLL(p)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib_, name)   # AttributeError = ABI drift, loud by design
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib_
    return lib_


def lib():
    return load_library()


def check(rc: int) -> None:
    if rc != QV_OK:
        msg = lib().qv_last_error()
        raise QvError(rc, msg.decode() if msg else f"qv error {rc}")
