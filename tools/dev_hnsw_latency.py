"""Single-query latency of the device HNSW traversal against the exact flat scan on the same 1M x 768 corpus (host pointers):
python tools/dev_hnsw_latency.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, quiver_amd
from quiver_amd.device_index import DeviceGraph, random_levels
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = 768
idx = quiver_amd.DeviceIndex(D, "cosine", rowmajor=True); idx.reserve(N); idx.add_synthetic(20260424, 0, N)
g = DeviceGraph.build(idx, random_levels(N, 1, 1), m=16, max_m0=32, ef_construction=200)
qg = quiver_amd.DeviceIndex(D, "cosine"); qg.add_synthetic(20260425, 0, 64)
hq = np.stack([qg.get_row(i) for i in range(64)])
for ef in (64, 128, 512):
    g.search(hq[:1], 10, ef)
    t = time.perf_counter()
    for i in range(64):
        g.search(hq[i:i + 1], 10, ef)
    dt = (time.perf_counter() - t) / 64
    r, d, c, ev = g.search(hq, 10, ef, with_evals=True)
    print("HNSW one query per call, efSearch %d: %.2f ms per query (%.0f evaluations on average)" % (ef, dt * 1e3, ev.mean()), flush=True)
idx.search(hq[:1], 10)
t = time.perf_counter()
for i in range(64):
    idx.search(hq[i:i + 1], 10)
print("exact flat scan, one query per call: %.3f ms per query" % ((time.perf_counter() - t) / 64 * 1e3), flush=True)
