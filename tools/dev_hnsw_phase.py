"""Where a (nearly) lone traversal's time goes: four queries in one call on a 1M x 768 graph with a -DQV_HNSW_PROF build
(tools/build_variant.sh prof qv_hnsw.hip -DQV_HNSW_PROF; QV_LIB_PATH=quiver_amd/lib/libqv_prof.so python tools/dev_hnsw_phase.py [rows] [nq])"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, quiver_amd
from quiver_amd.device_index import DeviceGraph, random_levels
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 4
D = 768
idx = quiver_amd.DeviceIndex(D, "cosine", rowmajor=True); idx.reserve(N); idx.add_synthetic(20260424, 0, N)
t0 = time.perf_counter()
g = DeviceGraph.build(idx, random_levels(N, 1, 1), m=16, max_m0=32, ef_construction=200)
print("build %.2f s" % (time.perf_counter() - t0), g.stats() if hasattr(g, "stats") else "", flush=True)
qg = quiver_amd.DeviceIndex(D, "cosine"); qg.add_synthetic(20260425, 0, 64)
hq = np.stack([qg.get_row(i) for i in range(64)])
print("---- search ----", flush=True)
for ef in (128,):
    for rep in range(3):
        t = time.perf_counter(); r, d, c, ev = g.search(hq[:NQ], 10, ef, with_evals=True); dt = time.perf_counter() - t
        print("ef %d nq %d: %.2f ms, evals %s" % (ef, NQ, dt * 1e3, ev.tolist()), flush=True)
