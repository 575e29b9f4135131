"""ctypes binding of libqvhost.so — the C++ host-side mirror of the reference's Go callers
(quiver_amd/csrc/host/qvhost.h).  Pure plumbing: no logic lives here."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "lib", "libqvhost.so")
_lib = None

_vp, _cp, _u32, _i, _f = C.c_void_p, C.c_char_p, C.c_uint32, C.c_int, C.c_float
_SIGS = {
    "qvh_last_error": (_cp, []),
    "qvh_results_new": (_vp, []), "qvh_results_free": (None, [_vp]), "qvh_results_count": (_i, [_vp]),
    "qvh_results_id": (_cp, [_vp, _i]), "qvh_results_distance": (_f, [_vp, _i]), "qvh_results_strategy": (_cp, [_vp]),
    "qvh_results_many_count": (_i, [_vp]), "qvh_results_many_len": (_i, [_vp, _i]),
    "qvh_results_many_id": (_cp, [_vp, _i, _i]), "qvh_results_many_distance": (_f, [_vp, _i, _i]),
    "qvh_results_many_strategy": (_cp, [_vp, _i]),
    "qvh_exact_new": (_vp, [_i, _i]), "qvh_exact_new_placed": (_vp, [_i, _vp, _i, _i, _i]), "qvh_exact_free": (None, [_vp]),
    "qvh_exact_insert": (_i, [_vp, _cp, _vp, _u32]), "qvh_exact_delete": (_i, [_vp, _cp]),
    "qvh_exact_search": (_i, [_vp, _vp, _u32, _i, _vp]), "qvh_exact_size": (_i, [_vp]), "qvh_exact_device_rows": (_u32, [_vp]),
    "qvh_hnsw_insert_batch": (_i, [_vp, C.POINTER(_cp), _vp, _u32, _u32, _u32, _u32]), "qvh_hnsw_built_on_device": (_i, [_vp]),
    "qvh_hybrid_exact_device_rows": (_u32, [_vp]), "qvh_hybrid_hnsw_built_on_device": (_i, [_vp]),
    "qvh_hnsw_new": (_vp, [_i, _i, _i, _i, _i, _i, _i, C.c_uint64]), "qvh_hnsw_free": (None, [_vp]),
    "qvh_hnsw_insert": (_i, [_vp, _cp, _vp, _u32]), "qvh_hnsw_delete": (_i, [_vp, _cp]),
    "qvh_hnsw_search": (_i, [_vp, _vp, _u32, _i, _vp, _vp]),
    "qvh_hnsw_search_batch": (_i, [_vp, _vp, _u32, _u32, _i, _vp, _vp, _vp]), "qvh_hnsw_search_batch_raw": (_i, [_vp, _vp, _u32, _u32, _i, _vp, _vp, _vp, _vp, C.POINTER(C.c_double)]),
    "qvh_hnsw_device_fallbacks": (_u32, [_vp]), "qvh_hnsw_topups": (_u32, [_vp]),
    "qvh_hnsw_size": (_u32, [_vp]), "qvh_hnsw_nodes": (_u32, [_vp]), "qvh_hnsw_node_level": (_i, [_vp, _u32]),
    "qvh_hnsw_links": (_i, [_vp, _u32, _i, _vp, _u32]), "qvh_hnsw_entry_point": (None, [_vp, C.POINTER(_u32), C.POINTER(_i)]),
    "qvh_hnsw_set_ef_search": (None, [_vp, _i]),
    "qvh_hnsw_distance_calls": (C.c_uint64, [_vp]), "qvh_hnsw_distance_evals": (C.c_uint64, [_vp]),
    "qvh_adapter_new": (_vp, [_i, _i, _i, _i, _i, _i, C.c_uint64]), "qvh_adapter_free": (None, [_vp]),
    "qvh_adapter_insert": (_i, [_vp, _cp, _vp, _u32]), "qvh_adapter_delete": (_i, [_vp, _cp]),
    "qvh_adapter_search": (_i, [_vp, _vp, _u32, _i, _vp]),
    "qvh_adapter_search_negative": (_i, [_vp, _vp, _u32, _vp, _u32, _f, _i, _vp]), "qvh_adapter_size": (_i, [_vp]),
    "qvh_hybrid_new": (_vp, [_i, _i, _i, _i, _i, _i, _i, C.c_double, C.c_uint64]), "qvh_hybrid_free": (None, [_vp]),
    "qvh_hybrid_new_placed": (_vp, [_i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, C.c_double, C.c_uint64]),
    "qvh_hybrid_insert": (_i, [_vp, _cp, _vp, _u32]),
    "qvh_hybrid_insert_batch": (_i, [_vp, C.POINTER(_cp), _vp, _vp, _u32]),
    "qvh_hybrid_delete": (_i, [_vp, _cp]), "qvh_hybrid_delete_batch": (_i, [_vp, C.POINTER(_cp), _u32]),
    "qvh_hybrid_search": (_i, [_vp, _vp, _u32, _i, _vp]),
    "qvh_hybrid_search_request": (_i, [_vp, _vp, _u32, _i, _cp, _vp, _u32, _f, _vp]),
    "qvh_hybrid_batch_search": (_i, [_vp, _vp, _u32, _u32, _i, _cp, _vp]),
    "qvh_hybrid_size": (_i, [_vp]), "qvh_hybrid_select_strategy": (_cp, [_vp, _i, _i, _i]),
}


def hlib():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise ImportError(f"{HOST_LIB_PATH} not found: build it with `make -C quiver_amd/csrc`")
        from ._lib import load_library
        load_library()                       # libqv.so first (RTLD_GLOBAL not needed: rpath $ORIGIN)
        h = C.CDLL(HOST_LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


class GoError(Exception):
    """an `error` value returned by the mirrored Go function (message = the Go error string)"""


def check(rc: int):
    if rc != 0:
        raise GoError(hlib().qvh_last_error().decode())


def f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32).ravel()


class Results:
    def __init__(self):
        self.h = hlib().qvh_results_new()

    def __del__(self):
        try:
            hlib().qvh_results_free(self.h)
        except Exception:
            pass

    def list(self):
        L = hlib()
        return [(L.qvh_results_id(self.h, i).decode(), float(np.float32(L.qvh_results_distance(self.h, i))))
                for i in range(L.qvh_results_count(self.h))]

    def strategy(self) -> str:
        return hlib().qvh_results_strategy(self.h).decode()

    def many(self):
        L = hlib()
        out, used = [], []
        for q in range(L.qvh_results_many_count(self.h)):
            out.append([(L.qvh_results_many_id(self.h, q, i).decode(), float(np.float32(L.qvh_results_many_distance(self.h, q, i))))
                        for i in range(L.qvh_results_many_len(self.h, q))])
            used.append(L.qvh_results_many_strategy(self.h, q).decode())
        return out, used
