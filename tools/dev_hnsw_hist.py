"""Which rows does a batch of traversals read, and how often?  (measurement build: tools/build_variant.sh hist qv_hnsw.hip -DQV_HNSW_HIST,
QV_LIB_PATH=quiver_amd/lib/libqv_hist.so: the kernel adds one to hist[row] per evaluated row instead of writing evaluation counts)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, quiver_amd
from quiver_amd.device_index import DeviceGraph, random_levels
N, D, k, ef, nq = 1_000_000, 768, 10, 128, 8192
idx = quiver_amd.DeviceIndex(D, "cosine", rowmajor=True); idx.reserve(N); idx.add_synthetic(20260424, 0, N)
cache = os.environ.get("QV_GRAPH_CACHE")
z = np.load(cache)
g = DeviceGraph(idx, z["levels"], z["l0_deg"], z["l0_links"], int(z["entry"]), int(z["cur_level"]), z["up_off"], z["up_links"])
qg = quiver_amd.DeviceIndex(D, "cosine"); qg.add_synthetic(20260425, 0, nq)
dq = torch.from_numpy(np.stack([qg.get_row(i) for i in range(nq)])).cuda()
dr = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
dc = torch.empty(nq, dtype=torch.int32, device="cuda"); hist = torch.zeros(N, dtype=torch.int32, device="cuda")
g.search_device(dq.data_ptr(), nq, k, ef, dr.data_ptr(), dd.data_ptr(), dc.data_ptr(), hist.data_ptr(), 0)
torch.cuda.synchronize()
h = np.sort(hist.cpu().numpy().astype(np.int64))[::-1]
tot = int(h.sum())
cum = np.cumsum(h)
print(json.dumps({"evaluations": tot, "per_query": tot / nq, "rows_touched": int((h > 0).sum()),
                  "top_rows_share": {str(n): round(float(cum[n - 1]) / tot, 4) for n in (16, 128, 1024, 8192, 65536, 262144)},
                  "visits_of_the_hottest": h[:8].tolist(), "visits_at_rank": {str(n): int(h[n]) for n in (100, 1000, 10000, 100000, 500000)},
                  "in_degree_vs_visits": None}))
deg_in = np.bincount(z["l0_links"][np.arange(N)[:, None].repeat(32, 1) < 0].ravel(), minlength=N) if False else None
links = z["l0_links"]; deg = z["l0_deg"]
mask = np.arange(links.shape[1])[None, :] < deg[:, None]
indeg = np.bincount(links[mask].ravel().astype(np.int64), minlength=N)
hv = hist.cpu().numpy().astype(np.int64)
order = np.argsort(-indeg)
print(json.dumps({"in_degree_max_p99_median": [int(indeg.max()), int(np.percentile(indeg, 99)), int(np.median(indeg))],
                  "share_of_visits_going_to_the_top_in_degree_rows": {str(n): round(float(hv[order[:n]].sum()) / tot, 4) for n in (1024, 8192, 65536)},
                  "corr_in_degree_visits": round(float(np.corrcoef(indeg, hv)[0, 1]), 4)}))
