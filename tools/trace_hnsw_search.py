import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, quiver_amd
from quiver_amd.device_index import DeviceGraph, random_levels
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
idx = quiver_amd.DeviceIndex(D, "cosine", rowmajor=True); idx.reserve(N); idx.add_synthetic(20260424, 0, N)
g = DeviceGraph.build(idx, random_levels(N, 1, 1), m=16, max_m0=32, ef_construction=200)
qg = quiver_amd.DeviceIndex(D, "cosine"); qg.add_synthetic(20260425, 0, 8192)
hq = np.stack([qg.get_row(i) for i in range(8192)])
for ef in (128, 512):
    g.search(hq, 10, ef)
    t = time.perf_counter(); g.search(hq, 10, ef); print("ef", ef, "host-pointer search", (time.perf_counter() - t) * 1e3, "ms", flush=True)
