"""HNSW measurement (BASELINE configs[3] shape, reduced N: the reference's sequential graph
build — efConstruction searches per insert — is not a data-parallel path).

  python tests/bench/bench_hnsw.py [--rows 20000] [--dim 768] [--efc 100] [--efs 128] [--nq 1000] [--max-level 16]

Builds the graph with the product host HNSW (every distance a libqv call), walks --nq queries on
the device (qv_graph_search, one wavefront per query), and reports QPS, distance evaluations/s,
gathered GB/s, recall@10 against the exact scan, and — beside it — the CPU oracle traversing the
IDENTICAL graph (same seed -> same graph, asserted) on one core."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quiver_amd
from quiver_amd import hnsw
from tests import _oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=20000); ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--m", type=int, default=16); ap.add_argument("--efc", type=int, default=100); ap.add_argument("--efs", type=int, default=128)
ap.add_argument("--nq", type=int, default=1000); ap.add_argument("--k", type=int, default=10)
ap.add_argument("--max-level", type=int, default=16); ap.add_argument("--metric", default="cosine")
ap.add_argument("--cpu-queries", type=int, default=100)
ap.add_argument("--skip-cpu", action="store_true")
ap.add_argument("--more-nq", type=int, nargs="*", default=[], help="extra batch sizes timed on the same graph")
a = ap.parse_args()
mid = quiver_amd.metric_id(a.metric)
rows = O.gen_rows(20260424, 0, a.rows, a.dim)
qs = O.gen_rows(20260425, 0, a.nq, a.dim)

t0 = time.perf_counter()
h = hnsw.HNSW(hnsw.Config(M=a.m, EfConstruction=a.efc, EfSearch=a.efs, MaxLevel=a.max_level, DistanceFunc=mid, Seed=7))
for i in range(a.rows):
    h.Insert("v%d" % i, rows[i])
t_build = time.perf_counter() - t0
calls, evals_build = h.distance_calls(), h.distance_evals()

h.SearchBatch(qs[:32], a.k)                              # warm-up + graph upload
t0 = time.perf_counter()
res, ev = h.SearchBatch(qs, a.k, with_evals=True)
t_gpu = time.perf_counter() - t0
topups = h.topups()

# the device call alone (qv_graph_search: upload, traversal kernel, download), without the id-string marshalling above
extra = {}
for n2 in [a.nq] + list(a.more_nq):
    q2 = O.gen_rows(20260426, 0, n2, a.dim)
    h.search_batch_raw(q2[:64], a.k)
    _, _, c2, e2, t2 = h.search_batch_raw(q2, a.k)
    extra[str(n2)] = {"qps": n2 / t2, "batch_ms": t2 * 1e3, "evals_per_s": float(e2.sum()) / t2, "gather_GBps": float(e2.sum()) * a.dim * 4 / t2 / 1e9,
                      "flagged_or_underfilled": int((c2 != a.k).sum())}

# exact top-k for recall
flat = quiver_amd.DeviceIndex(a.dim, mid); flat.add(rows)
er, ed, _ = flat.search(qs, a.k, batched=True)
hit = sum(len(set(r.VectorIndex for r in res[i]) & set(er[i].tolist())) for i in range(a.nq))

if a.skip_cpu:
    print(json.dumps({"workload": "HNSW %dx%d efS=%d MaxLevel=%d" % (a.rows, a.dim, a.efs, a.max_level), "device_call_only": extra,
                      "recall_at_10_vs_exact": hit / (a.nq * a.k)}))
    sys.exit(0)
# CPU oracle on the identical graph
t0 = time.perf_counter()
o = O.HNSW(mid, a.dim, M=a.m, efConstruction=a.efc, efSearch=a.efs, maxLevel=a.max_level, seed=7)
for i in range(a.rows):
    o.insert(rows[i])
t_cpu_build = time.perf_counter() - t0
same_graph = all(np.array_equal(h.links(n, 0), o.links(n, 0)) for n in range(0, a.rows, max(1, a.rows // 500)))
t0 = time.perf_counter()
cpu_evals = 0; identical = True
for i in range(a.cpu_queries):
    r, d, ne = o.search(qs[i], a.k, with_evals=True)
    cpu_evals += ne
    identical &= [x.VectorIndex for x in res[i]] == r.tolist() and np.array_equal(np.array([x.Distance for x in res[i]], np.float32).view(np.uint32), d.view(np.uint32))
t_cpu = time.perf_counter() - t0

print(json.dumps({
    "workload": "HNSW M=%d efC=%d efS=%d MaxLevel=%d, %dx%d %s, k=%d" % (a.m, a.efc, a.efs, a.max_level, a.rows, a.dim, a.metric, a.k),
    "build_s_gpu_distances": t_build, "build_distance_calls": calls, "build_distance_evals": evals_build,
    "gpu_qps": a.nq / t_gpu, "gpu_batch_ms": t_gpu * 1e3, "evals_per_query_device": float(np.mean(ev[ev > 0])) if (ev > 0).any() else 0.0,
    "evals_per_s": float(ev.sum()) / t_gpu, "gather_GBps": float(ev.sum()) * a.dim * 4 / t_gpu / 1e9,
    "underfilled_queries_topped_up_by_exact_scan": topups - 0, "recall_at_10_vs_exact": hit / (a.nq * a.k),
    "cpu_oracle_qps_1core": a.cpu_queries / t_cpu, "cpu_build_s": t_cpu_build, "cpu_evals_per_query": cpu_evals / a.cpu_queries,
    "device_call_only": extra, "graph_identical_to_cpu_graph": bool(same_graph), "results_identical_to_cpu_traversal": bool(identical)}))
