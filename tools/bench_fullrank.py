"""Full-ranking measurement: Collection.Search with a filter asks the index for k = Index.Size()
results (collection.go:679-682).  python tools/bench_fullrank.py [--rows 1000000] [--k 1000000]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quiver_amd
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000); ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--k", type=int, default=0); ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
k = a.k or a.rows
idx = quiver_amd.DeviceIndex(a.dim, "cosine"); idx.reserve(a.rows); idx.add_synthetic(20260424, 0, a.rows)
q = torch.randn(1, a.dim, device="cuda")
dr = torch.empty((1, k), dtype=torch.int32, device="cuda"); dd = torch.empty((1, k), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
idx.search_device(q.data_ptr(), 1, k, dr.data_ptr(), dd.data_ptr(), s); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    idx.search_device(q.data_ptr(), 1, k, dr.data_ptr(), dd.data_ptr(), s)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.reps
d = dd.cpu().numpy()[0]; r = dr.cpu().numpy()[0].view(np.uint32)
ok = bool(np.all(d[:-1] <= d[1:])) and len(np.unique(r)) == k
print(json.dumps({"workload": "full ranking k=%d of %dx%d cosine" % (k, a.rows, a.dim), "ms": dt * 1e3, "sorted_and_unique": ok}))

# the two ways to serve a filtered search (10 % of the rows match, top-10 wanted), host pointers both:
#   (a) what the reference does above the seam: full ranking down to the host, then walk it until 10 rows match
#   (b) qv_index_search_masked: the row bitmap goes into the scan, 10 results come back
qh = q.cpu().numpy()
mask = np.random.default_rng(1).random(a.rows) < 0.10
t0 = time.perf_counter()
for _ in range(3):
    fr, fd, _ = idx.search(qh, a.rows)
    first = []
    for x in fr[0]:
        if mask[x]:
            first.append(int(x))
            if len(first) == 10:
                break
t_full = (time.perf_counter() - t0) / 3
t0 = time.perf_counter()
for _ in range(3):
    mr, md, mc = idx.search_masked(qh, 10, mask)
t_mask = (time.perf_counter() - t0) / 3
print(json.dumps({"workload": "filtered top-10, 10%% of %dx%d rows match" % (a.rows, a.dim), "full_ranking_then_filter_ms": t_full * 1e3,
                  "row_bitmap_in_scan_ms": t_mask * 1e3, "same_result": first == mr[0, :10].tolist()}))
