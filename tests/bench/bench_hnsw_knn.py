"""HNSW traversal at BASELINE configs[3] scale (1M x 768, M=16 -> MaxM0=32 level-0 links, efSearch=128).

The reference builds its graph by sequential Insert calls (efConstruction searches each): hours at 1M on
one core and not a data-parallel path.  To measure the TRAVERSAL kernel at the configured size the graph
here is an exact 32-NN graph made by the product's own batched flat scan (fp32-MFMA filter + exact
re-score), i.e. a graph of the same shape and degree over the same kind of vectors.  The CPU oracle runs
HNSW.Search on the IDENTICAL graph (qvo_hnsw_load_flat) for a sample of the queries: results must be
bit-identical.  Recall is reported against the exact scan.

  python tests/bench/bench_hnsw_knn.py [--rows 1000000] [--dim 768] [--m 32] [--efs 128] [--nq 16384] [--cpu-queries 50]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch                      # before libqv: both must share one HIP runtime (torch bundles its own)
import quiver_amd
from tests import _oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1000000); ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--m", type=int, default=32); ap.add_argument("--efs", type=int, default=128)
ap.add_argument("--nq", type=int, default=16384); ap.add_argument("--k", type=int, default=10)
ap.add_argument("--cpu-queries", type=int, default=50); ap.add_argument("--metric", default="cosine")
ap.add_argument("--chunk", type=int, default=4096)
a = ap.parse_args()
mid = quiver_amd.metric_id(a.metric)
N, D, M = a.rows, a.dim, a.m

t0 = time.perf_counter()
rows = np.empty((N, D), np.float32)
for s in range(0, N, 100000):
    e = min(N, s + 100000); rows[s:e] = O.gen_rows(20260424, s, e - s, D)
t_gen = time.perf_counter() - t0

idx = quiver_amd.DeviceIndex(D, a.metric, rowmajor=True)
idx.reserve(N)
for s in range(0, N, 250000):
    idx.add(rows[s:min(N, s + 250000)])

t0 = time.perf_counter()
links = np.empty((N, M), np.uint32)
cols = np.arange(M)[None, :]
for s in range(0, N, a.chunk):
    e = min(N, s + a.chunk)
    nbr, _, _ = idx.search(rows[s:e], M + 1, batched=True)
    me = np.arange(s, e, dtype=np.uint32)[:, None]
    hit = nbr == me
    p = np.where(hit.any(axis=1), hit.argmax(axis=1), M)[:, None]     # where the node itself sits (M = absent: drop the last)
    links[s:e] = np.where(cols < p, nbr[:, :M], nbr[:, 1:M + 1])
t_knn = time.perf_counter() - t0
deg = np.full(N, M, np.uint32)

g = quiver_amd.DeviceGraph(idx, np.zeros(N, np.int8), deg, links, entry=0)
qs = O.gen_rows(20260425, 0, a.nq, D)
g.search(qs[:256], a.k, a.efs)                                        # warm-up
runs = {}
for n2 in sorted({min(a.nq, 4096), a.nq}):
    t0 = time.perf_counter()
    r, d, c, ev = g.search(qs[:n2], a.k, a.efs, with_evals=True)
    t = time.perf_counter() - t0
    runs[str(n2)] = {"qps": n2 / t, "batch_ms": t * 1e3, "evals_per_query": float(ev.mean()), "evals_per_s": float(ev.sum()) / t,
                     "gather_GBps": float(ev.sum()) * D * 4 / t / 1e9, "underfilled": int((c < a.k).sum())}

# the same batches with queries and results resident on the device (qv_graph_search_device): no PCIe in the timed region
dq = torch.from_numpy(qs).cuda()
dr = torch.empty((a.nq, a.k), dtype=torch.int32, device="cuda"); dd = torch.empty((a.nq, a.k), dtype=torch.float32, device="cuda")
dc = torch.empty(a.nq, dtype=torch.int32, device="cuda"); de = torch.empty(a.nq, dtype=torch.int32, device="cuda")
sp = torch.cuda.current_stream().cuda_stream
runs_dev = {}
for n2 in sorted({min(a.nq, 4096), a.nq}):
    g.search_device(dq.data_ptr(), n2, a.k, a.efs, dr.data_ptr(), dd.data_ptr(), dc.data_ptr(), de.data_ptr(), sp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        g.search_device(dq.data_ptr(), n2, a.k, a.efs, dr.data_ptr(), dd.data_ptr(), dc.data_ptr(), de.data_ptr(), sp)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 3
    evs = float(de[:n2].sum().item())
    runs_dev[str(n2)] = {"qps": n2 / t, "batch_ms": t * 1e3, "evals_per_s": evs / t, "gather_GBps": evs * D * 4 / t / 1e9,
                         "tie_flagged": int((dc[:n2].cpu().numpy().view(np.uint32) == 0xFFFFFFFE).sum())}

er, ed, _ = idx.search(qs, a.k, batched=True)
hit = sum(len(set(r[i, :min(int(c[i]), a.k)].tolist()) & set(er[i].tolist())) for i in range(a.nq))

o = O.HNSW(mid, D, M=max(M // 2, 1), maxM0=M, efSearch=a.efs, maxLevel=1, seed=1)
o.load_flat(rows, deg, links, 0)
t0 = time.perf_counter(); identical = True; cpu_evals = 0
for i in range(a.cpu_queries):
    ro, do, eo = o.search(qs[i], a.k, with_evals=True)
    cpu_evals += eo
    n = min(int(c[i]), a.k)
    identical &= r[i, :n].tolist() == ro[:n].tolist() and d[i, :n].tobytes() == do[:n].tobytes() and (n < a.k or int(ev[i]) == eo - 1)
t_cpu = time.perf_counter() - t0

print(json.dumps({
    "workload": "HNSW traversal on an exact %d-NN graph, %dx%d %s, efSearch=%d, k=%d" % (M, N, D, a.metric, a.efs, a.k),
    "graph": "exact k-NN graph built by the product's batched flat scan (single layer, entry = node 0); not the reference's insertion-built graph",
    "gen_rows_s_cpu": t_gen, "knn_graph_build_s_gpu": t_knn, "knn_queries_per_s": N / t_knn,
    "device_call_only": runs, "device_resident": runs_dev, "recall_at_%d_vs_exact" % a.k: hit / (a.nq * a.k),
    "cpu_oracle_qps_1core": a.cpu_queries / t_cpu if a.cpu_queries else None, "cpu_evals_per_query": cpu_evals / max(a.cpu_queries, 1),
    "results_identical_to_cpu_traversal": bool(identical), "cpu_queries_checked": a.cpu_queries}))
