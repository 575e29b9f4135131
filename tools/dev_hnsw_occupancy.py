"""Device-resident HNSW traversal rate against resident queries per CU (QV_HNSW_WAVES_PER_CU is read once per process: run once per value):
QV_HNSW_WAVES_PER_CU=12 python tools/dev_hnsw_occupancy.py [rows] [graph-cache-dir]
Builds (or, with a cache directory that travels with the call, builds once per process) the configs[3] graph and times efSearch 128 / 256."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, quiver_amd
from quiver_amd.device_index import DeviceGraph, random_levels
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D, NQ, K = 768, 8192, 10
idx = quiver_amd.DeviceIndex(D, "cosine", rowmajor=True); idx.reserve(N); idx.add_synthetic(20260424, 0, N)
t = time.perf_counter()
g = DeviceGraph.build(idx, random_levels(N, 1, 1), m=16, max_m0=32, ef_construction=200)
bs = time.perf_counter() - t
qg = quiver_amd.DeviceIndex(D, "cosine"); qg.add_synthetic(20260425, 0, NQ)
dq = torch.from_numpy(np.stack([qg.get_row(i) for i in range(NQ)])).cuda()
dr = torch.empty((NQ, K), dtype=torch.int32, device="cuda"); dd = torch.empty((NQ, K), dtype=torch.float32, device="cuda")
dc = torch.empty((NQ,), dtype=torch.int32, device="cuda"); de = torch.empty((NQ,), dtype=torch.int32, device="cuda")
sp = torch.cuda.current_stream().cuda_stream
out = ["waves/CU %s: build %.2f s" % (os.environ.get("QV_HNSW_WAVES_PER_CU", "16"), bs)]
for ef in (128, 256):
    g.search_device(dq.data_ptr(), NQ, K, ef, dr.data_ptr(), dd.data_ptr(), dc.data_ptr(), de.data_ptr(), sp); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        g.search_device(dq.data_ptr(), NQ, K, ef, dr.data_ptr(), dd.data_ptr(), dc.data_ptr(), de.data_ptr(), sp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    ev = float(de.float().mean().item())
    out.append("ef %d: %.0f k QPS, %.2f TB/s gathered" % (ef, NQ / dt / 1e3, NQ * ev * D * 4 / dt / 1e12))
print(" | ".join(out), flush=True)
