"""Does the traversal's rate depend on WHERE the row table lands?  One process; the index (tiles + row-major copy, 6 GB) and the graph are
created, measured (8192 queries, efSearch 128) and destroyed several times, with dummy allocations of changing size held in between so that
the allocator hands out different memory each time.  QV_GRAPH_CACHE must name a graph saved by tools/dev_hnsw_r06.py."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, quiver_amd
from quiver_amd.device_index import DeviceGraph
N, D, k, ef, nq = 1_000_000, 768, 10, 128, 8192
z = np.load(os.environ["QV_GRAPH_CACHE"])
qg = quiver_amd.DeviceIndex(D, "cosine"); qg.add_synthetic(20260425, 0, nq)
dq = torch.from_numpy(np.stack([qg.get_row(i) for i in range(nq)])).cuda(); qg.close()
dr = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
dc = torch.empty(nq, dtype=torch.int32, device="cuda"); de = torch.empty(nq, dtype=torch.int32, device="cuda")
hold = []
for trial, pad_gb in enumerate((0, 1.3, 0, 7.7, 2.1, 0.4, 13.0, 0)):
    if pad_gb: hold.append(torch.empty(int(pad_gb * 2**30), dtype=torch.uint8, device="cuda"))
    idx = quiver_amd.DeviceIndex(D, "cosine", rowmajor=True); idx.reserve(N); idx.add_synthetic(20260424, 0, N)
    g = DeviceGraph(idx, z["levels"], z["l0_deg"], z["l0_links"], int(z["entry"]), int(z["cur_level"]), z["up_off"], z["up_links"])
    g.search_device(dq.data_ptr(), nq, k, ef, dr.data_ptr(), dd.data_ptr(), dc.data_ptr(), de.data_ptr(), 0); torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); g.search_device(dq.data_ptr(), nq, k, ef, dr.data_ptr(), dd.data_ptr(), dc.data_ptr(), de.data_ptr(), 0); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"trial": trial, "dummy_gb_added_before": pad_gb, "ms": [round(t, 2) for t in ts], "qps": round(nq / (min(ts) * 1e-3))}), flush=True)
    g.close(); idx.close()
    if trial % 3 == 2 and hold: hold.pop(0)
