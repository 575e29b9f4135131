"""Round-6 traversal probe: the 1M x 768 MaxLevel=1 graph built once on the device, then device-resident searches at one efSearch
for a list of batch sizes (queries per call).  Prints queries/s, evaluations and gathered GB/s per batch size — the batch-size
sweep shows how much of a call is the tail of its last traversals.
  python tools/dev_hnsw_r06.py [nq,nq,...] [ef] [reps] [rows]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import quiver_amd
from quiver_amd.device_index import DeviceGraph, random_levels
nqs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8192").split(",")]
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
N = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000
D, k = 768, 10
idx = quiver_amd.DeviceIndex(D, "cosine", rowmajor=True); idx.reserve(N); idx.add_synthetic(20260424, 0, N)
# QV_GRAPH_CACHE=<file.npz>: the built graph is kept there and uploaded from there the next time (counter passes run every kernel
# serialised: thousands of construction launches would take minutes per pass)
cache = os.environ.get("QV_GRAPH_CACHE")
t0 = time.perf_counter()
if cache and os.path.exists(cache):
    z = np.load(cache)
    g = DeviceGraph(idx, z["levels"], z["l0_deg"], z["l0_links"], int(z["entry"]), int(z["cur_level"]), z["up_off"], z["up_links"])
    print("graph uploaded from %s in %.2f s" % (cache, time.perf_counter() - t0), flush=True)
else:
    g = DeviceGraph.build(idx, random_levels(N, 1, 1), m=16, max_m0=32, ef_construction=200)
    print("build %.2f s" % (time.perf_counter() - t0), flush=True)
    if cache:
        lv, l0d, l0l, uo, ul = g.export(); info = g.info()
        np.savez(cache, levels=lv, l0_deg=l0d, l0_links=l0l, up_off=uo, up_links=ul, entry=info["entry"], cur_level=info["cur_level"])
nmax = max(nqs)
qg = quiver_amd.DeviceIndex(D, "cosine"); qg.add_synthetic(20260425, 0, nmax)
dq = torch.empty((nmax, D), dtype=torch.float32, device="cuda")
CH = 8192
for s0 in range(0, nmax, CH):
    hq = np.stack([qg.get_row(i) for i in range(s0, min(nmax, s0 + CH))]); dq[s0:s0 + hq.shape[0]].copy_(torch.from_numpy(hq))
qg.close()
dr = torch.empty((nmax, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nmax, k), dtype=torch.float32, device="cuda")
dc = torch.empty(nmax, dtype=torch.int32, device="cuda"); de = torch.empty(nmax, dtype=torch.int32, device="cuda")
sp = torch.cuda.current_stream().cuda_stream
for nq in nqs:
    g.search_device(dq.data_ptr(), nq, k, ef, dr.data_ptr(), dd.data_ptr(), dc.data_ptr(), de.data_ptr(), sp)
    torch.cuda.synchronize()
    each = []                         # (stream 0 = the graph's own stream: torch events on the current stream would not see it)
    for _ in range(reps):
        t0 = time.perf_counter()
        g.search_device(dq.data_ptr(), nq, k, ef, dr.data_ptr(), dd.data_ptr(), dc.data_ptr(), de.data_ptr(), sp)
        torch.cuda.synchronize()
        each.append(time.perf_counter() - t0)
    t = sum(each) / reps
    ev = de[:nq].cpu().numpy().astype(np.int64); cnt = dc[:nq].cpu().numpy().view(np.uint32)
    print(json.dumps({"ef": ef, "nq": nq, "ms": round(t * 1e3, 3), "each_ms": [round(x * 1e3, 2) for x in each], "qps": round(nq / t), "evals_per_query": round(float(ev.mean()), 1),
                      "evals_p5_p50_p95_max": [int(np.percentile(ev, p)) for p in (5, 50, 95, 100)],
                      "gathered_GBps": round(float(ev.sum()) * D * 4 / t / 1e9, 1), "flagged": int((cnt == 0xFFFFFFFE).sum())}), flush=True)
