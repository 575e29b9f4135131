"""oracle_np — independent numpy restatement of the reference's distance arithmetic.

TEST INFRASTRUCTURE ONLY.  Nothing under quiver_amd/ may import this; it exists so
that two independently written restatements (this one and oracle/qv_oracle.c) can
be checked against each other and against the reference's known-answer tests
(tests/golden/ref_kats.json).

Sequential accumulation is expressed with ``np.cumsum`` (a strict left-to-right
recurrence, unlike ``np.sum`` which is pairwise), so element order matches the Go
loops in pkg/vectortypes/distances.go:12-104 and pkg/hnsw/adapter.go:105-167.
"""
from __future__ import annotations

import numpy as np

COSINE, L2, L2SQ, DOT, L1, COSINE_F32, L2_F32, DOT_F32, L2SQ_F64 = range(9)
METRIC_NAMES = {
    COSINE: "cosine", L2: "euclidean", L2SQ: "squared_euclidean", DOT: "dot_product", L1: "manhattan",
    COSINE_F32: "hnsw_cosine", L2_F32: "hnsw_euclidean", DOT_F32: "hnsw_dot_product", L2SQ_F64: "arrow_squared_euclidean",
}


def _seq_sum(x: np.ndarray, dtype) -> np.ndarray:
    """left-to-right sum of the last axis in ``dtype`` (0 for an empty axis)."""
    x = np.asarray(x, dtype=dtype)
    if x.shape[-1] == 0:
        return np.zeros(x.shape[:-1], dtype=dtype)
    return np.cumsum(x, axis=-1, dtype=dtype)[..., -1]


def distances(metric: int, query: np.ndarray, rows: np.ndarray) -> np.ndarray:
    """distance(query, row) for every row of ``rows`` [n, d]; float32 [n]."""
    q32 = np.asarray(query, dtype=np.float32)
    r32 = np.atleast_2d(np.asarray(rows, dtype=np.float32))
    if q32.shape[-1] != r32.shape[-1]:
        # distances.go:13-15 panics; adapter.go:106-108 returns ErrDimensionMismatch
        raise ValueError("vectors must have the same length")
    q64, r64 = q32.astype(np.float64), r32.astype(np.float64)
    with np.errstate(all="ignore"):
        if metric == COSINE:  # distances.go:12-40
            dot = _seq_sum(q64 * r64, np.float64)
            ma = _seq_sum(q64 * q64, np.float64)
            mb = _seq_sum(r64 * r64, np.float64)
            sim = dot / (np.sqrt(ma) * np.sqrt(mb))
            sim = np.where(sim > 1.0, 1.0, np.where(sim < -1.0, -1.0, sim))
            out = (1.0 - sim).astype(np.float32)
            return np.where((ma == 0) | (mb == 0), np.float32(1.0), out)
        if metric == L2:  # distances.go:43-55
            diff = (q32 - r32).astype(np.float64)
            return np.sqrt(_seq_sum(diff * diff, np.float64)).astype(np.float32)
        if metric == L2SQ:  # distances.go:60-72
            diff = q32 - r32
            return _seq_sum(diff * diff, np.float32)
        if metric == DOT:  # distances.go:77-90
            return (1.0 - _seq_sum(q64 * r64, np.float64)).astype(np.float32)
        if metric == L1:  # distances.go:93-104
            diff = (q32 - r32).astype(np.float64)
            return _seq_sum(np.abs(diff), np.float64).astype(np.float32)
        if metric == COSINE_F32:  # adapter.go:105-136
            dot = _seq_sum(q32 * r32, np.float32)
            na = _seq_sum(q32 * q32, np.float32)
            nb = _seq_sum(r32 * r32, np.float32)
            sa = np.sqrt(na.astype(np.float64)).astype(np.float32)
            sb = np.sqrt(nb.astype(np.float64)).astype(np.float32)
            sim = dot / (sa * sb)
            sim = np.where(sim > 1.0, np.float32(1.0), np.where(sim < -1.0, np.float32(-1.0), sim)).astype(np.float32)
            out = (np.float32(1.0) - sim).astype(np.float32)
            return np.where((na == 0) | (nb == 0), np.float32(1.0), out)
        if metric == L2_F32:  # adapter.go:139-151
            diff = q32 - r32
            return np.sqrt(_seq_sum(diff * diff, np.float32).astype(np.float64)).astype(np.float32)
        if metric == L2SQ_F64:  # index/arrow_hnsw.go:124-132
            d = q64 - r64
            return _seq_sum(d * d, np.float64).astype(np.float32)
        if metric == DOT_F32:  # adapter.go:154-165
            return (np.float32(1.0) - _seq_sum(q32 * r32, np.float32)).astype(np.float32)
    return distances(COSINE, query, rows)  # types.go:46-47 unknown -> cosine


def distance(metric: int, a, b) -> np.float32:
    return distances(metric, np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)[None, :])[0]


def exact_search(metric: int, rows: np.ndarray, query: np.ndarray, k: int, alive: np.ndarray | None = None):
    """pkg/hybrid/exact.go:92-133 with the declared (distance, row) order."""
    rows = np.atleast_2d(np.asarray(rows, dtype=np.float32))
    n = rows.shape[0]
    live = np.arange(n, dtype=np.uint32) if alive is None else np.nonzero(np.asarray(alive))[0].astype(np.uint32)
    if live.size == 0:
        return np.zeros(0, np.uint32), np.zeros(0, np.float32)
    if k <= 0:
        raise ValueError("k must be positive")
    d = distances(metric, query, rows[live])
    key = np.where(np.isnan(d), np.float32(np.inf), d)
    order = np.lexsort((live, key, np.isnan(d)))
    order = order[: min(k, live.size)]
    return live[order], d[order]


# ---------------------------------------------------------------- generator mirror ---
_M64 = (1 << 64) - 1


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def gen_rows(seed: int, row0: int, n: int, dim: int) -> np.ndarray:
    """numpy mirror of qvo_gen_rows (bit-identical)."""
    with np.errstate(over="ignore"):
        rows = (np.arange(n, dtype=np.uint64) + np.uint64(row0)) * np.uint64(0xD1342543DE82EF95)
        row_key = _splitmix64(np.uint64(seed & _M64) ^ rows)
        h = _splitmix64(row_key[:, None] + np.arange(dim, dtype=np.uint64)[None, :])
    s = ((h & np.uint64(0xFFFF)).astype(np.int64) + ((h >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64)
         + ((h >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64) + (h >> np.uint64(48)).astype(np.int64) - 131070)
    v = s.astype(np.float64)
    sumsq = (v * v).sum(axis=1)  # exact integers: order-free
    norm = np.where(sumsq > 0, np.sqrt(sumsq), 1.0)
    return (v / norm[:, None]).astype(np.float32)
