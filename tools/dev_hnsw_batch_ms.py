"""One qv_graph_search call of nq queries on the 1M x 768 graph, ms per call (median of 7), by nq:  python tools/dev_hnsw_batch_ms.py [nq,nq,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, quiver_amd
from quiver_amd.device_index import DeviceGraph, random_levels
from tests import _oracle as O
N, D = 1_000_000, 768
nqs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "64,256,384,512,768,1024,1536,2048,4096").split(",")]
idx = quiver_amd.DeviceIndex(D, "cosine", rowmajor=True); idx.reserve(N); idx.add_synthetic(20260424, 0, N)
g = DeviceGraph.build(idx, random_levels(N, 1, 1), m=16, max_m0=32, ef_construction=200)
qs = O.gen_rows(20260425, 0, max(nqs), D)
out = []
for nq in nqs:
    g.search(qs[:nq], 10, 128)
    ts = []
    for _ in range(7):
        t = time.perf_counter(); g.search(qs[:nq], 10, 128); ts.append(time.perf_counter() - t)
    ts.sort(); out.append("%d: %.2f" % (nq, ts[3] * 1e3))
print("ms per call of nq queries: " + "  ".join(out), flush=True)
