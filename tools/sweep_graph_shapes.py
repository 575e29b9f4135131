"""HNSW build + traversal throughput at other dimensions / metrics (MaxLevel = 1 graphs of 200k nodes):
python tools/sweep_graph_shapes.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import quiver_amd
from quiver_amd.device_index import DeviceGraph, random_levels

N, NQ, K = 200_000, 8192, 10
for metric, dim in (("cosine", 64), ("cosine", 100), ("cosine", 128), ("cosine", 384), ("cosine", 768), ("cosine", 1536),
                    ("l2", 128), ("l2", 768), ("dot", 768), ("cosine_f32", 768), ("l2_f32", 128)):
    idx = quiver_amd.DeviceIndex(dim, metric, rowmajor=True)
    idx.reserve(N)
    idx.add_synthetic(20260424, 0, N)
    t = time.perf_counter()
    g = DeviceGraph.build(idx, random_levels(N, 1, 1), m=16, max_m0=32, ef_construction=200)
    tb = time.perf_counter() - t
    qg = quiver_amd.DeviceIndex(dim, metric)
    qg.add_synthetic(20260425, 0, NQ)
    hq = np.stack([qg.get_row(i) for i in range(NQ)])
    line = "%-10s dim %4d: build %.2f s (%.0f nodes/s)" % (metric, dim, tb, N / tb)
    for ef in (64, 256):
        g.search(hq, K, ef)
        t = time.perf_counter()
        r = g.search(hq, K, ef, with_evals=True)
        dt = time.perf_counter() - t
        ev = float(np.mean(r[3]))
        line += " | ef %d: %.0f QPS (host pointers), %.0f evals/query, %.0f GB/s gathered" % (ef, NQ / dt, ev, ev * dim * 4 * NQ / dt / 1e9)
    print(line, flush=True)
    del g, idx
