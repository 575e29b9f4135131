"""Thread-parallel helpers for the CPU oracle on corpora that do not fit one pass (TEST INFRASTRUCTURE ONLY).
The oracle is called through ctypes, which drops the GIL: plain threads scale with the cores, and nothing is forked or
spawned from a process that has initialised the GPU."""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor

import numpy as np


def _chunk_topk(args):
    metric, seed, row0, n, dim, query, k = args
    from tests import _oracle as O
    rows = O.gen_rows(seed, row0, n, dim)
    d = O.all_distances(metric, rows, query)
    order = np.lexsort((np.arange(n), d))[:k]                # (distance, row) ascending: the declared tie-break
    return (row0 + order).astype(np.uint32), d[order]


def exact_topk_synthetic(metric: int, seed: int, n_rows: int, dim: int, query, k: int, chunk: int = 100_000, workers: int = 16):
    """exact top-k of `query` over the synthetic corpus (seed, rows 0..n_rows) by the CPU oracle's scalar arithmetic,
    chunk by chunk on `workers` threads; -> (rows [k] uint32, dist [k] float32) under (distance, row) order"""
    q = np.ascontiguousarray(query, dtype=np.float32)
    jobs = [(metric, seed, s, min(chunk, n_rows - s), dim, q, k) for s in range(0, n_rows, chunk)]
    with ThreadPoolExecutor(max_workers=min(workers, len(jobs))) as pool:
        parts = list(pool.map(_chunk_topk, jobs))
    rows = np.concatenate([p[0] for p in parts]); dist = np.concatenate([p[1] for p in parts])
    order = np.lexsort((rows, dist))[:k]
    return rows[order], dist[order]
