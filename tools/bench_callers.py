"""Concurrent single-query callers through the C ABI (native threads, tools/native/qv_callers.cpp): QPS and latency by caller
count on the flat index and on the device HNSW graph.  This is the traffic the reference's unchanged Go host produces
(pkg/core/collection.go:647; pkg/core/db.go:805-828; pkg/hnsw/hnsw.go:602-606).

    python tools/bench_callers.py [--rows 1000000] [--dim 768] [--graph-rows 1000000] [--seconds 2] [--out gpurun_out/callers.json]
QV_COALESCE=0 in the environment turns the sharing off (every call on its own, as before round 5)."""
import os; os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # the host's setting, before the first HIP call
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quiver_amd                                              # noqa: E402
from tests import _callers, _oracle as O                        # noqa: E402  (oracle: query generator + spot check only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--graph-rows", type=int, default=1_000_000)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--ef", type=int, default=128)
    ap.add_argument("--flat-callers", default="1,2,4,8,16,64,256")
    ap.add_argument("--graph-callers", default="1,8,64,256,1024")
    ap.add_argument("--sharded", type=int, default=0, help="also: a sharded handle of this many shards on device 0, the flat caller counts")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    rec = {"coalesce": os.environ.get("QV_COALESCE", "1"), "k": a.k, "flat": [], "graph": []}
    qs = O.gen_rows(20260425, 0, 1024, a.dim)
    if a.rows:
        idx = quiver_amd.DeviceIndex(a.dim, "cosine")
        idx.add_synthetic(20260424, 0, a.rows)
        ref_r, ref_d, _ = idx.search(qs[:256], a.k, batched=True)          # one batch call: what every caller must get
        for t in [int(x) for x in a.flat_callers.split(",")]:
            s0 = _callers.coalesce_stats("index", idx.handle)
            r = _callers.run("index", idx.handle, qs[:256], a.k, threads=t, seconds=a.seconds)
            s1 = _callers.coalesce_stats("index", idx.handle)
            seen = r["count"] != 0xFFFFFFFD
            same = bool(np.array_equal(r["rows"][seen], ref_r[seen]) and np.array_equal(r["dist"][seen].view(np.uint32), ref_d[seen].view(np.uint32)))
            groups = s1["groups"] - s0["groups"]
            e = dict(callers=t, qps=round(r["qps"], 1), p50_us=round(r["p50_us"], 1), p99_us=round(r["p99_us"], 1), max_us=round(r["max_us"], 1),
                     calls=r["calls"], errors=r["errors"], mismatches=r["mismatches"], same_as_batch_call=same,
                     solo=s1["solo"] - s0["solo"], groups=groups, mean_group=round((s1["group_queries"] - s0["group_queries"]) / max(groups, 1), 1),
                     mean_group_pass_us=round((s1["group_pass_ns"] - s0["group_pass_ns"]) / max(groups, 1) / 1e3, 1),
                     lingers=s1["lingers"] - s0["lingers"], mean_linger_us=round((s1["linger_ns"] - s0["linger_ns"]) / max(s1["lingers"] - s0["lingers"], 1) / 1e3, 1))
            rec["flat"].append(e)
            print("flat ", json.dumps(e), flush=True)
        rec["flat_shape"] = [a.rows, a.dim]
        idx.close()
    if a.rows and a.sharded:
        from quiver_amd.device_index import ShardedIndex
        sh = ShardedIndex(a.dim, "cosine", devices=[0] * a.sharded, peer_copy=a.sharded > 1)
        sh.add_synthetic(20260424, 0, a.rows)
        ref_r, ref_d, _ = sh.search(qs[:256], a.k)
        rec["sharded"] = []
        for t in [int(x) for x in a.flat_callers.split(",")]:
            r = _callers.run("sharded", sh.handle, qs[:256], a.k, threads=t, seconds=a.seconds)
            seen = r["count"] != 0xFFFFFFFD
            same = bool(np.array_equal(r["rows"][seen], ref_r[seen]) and np.array_equal(r["dist"][seen].view(np.uint32), ref_d[seen].view(np.uint32)))
            e = dict(callers=t, qps=round(r["qps"], 1), p50_us=round(r["p50_us"], 1), p99_us=round(r["p99_us"], 1), calls=r["calls"], errors=r["errors"],
                     mismatches=r["mismatches"], same_as_batch_call=same)
            rec["sharded"].append(e)
            print("sharded", json.dumps(e), flush=True)
        sh.close()
    if a.graph_rows:
        from quiver_amd.device_index import DeviceGraph
        gi = quiver_amd.DeviceIndex(a.dim, "cosine", rowmajor=True)
        gi.add_synthetic(20260424, 0, a.graph_rows)
        t0 = time.time()
        g = DeviceGraph.build(gi, np.zeros(a.graph_rows, np.int8), m=16, max_m0=32, ef_construction=200)
        rec["graph_build_s"] = round(time.time() - t0, 2)
        ref_r, ref_d, ref_c = g.search(qs, a.k, a.ef)
        # one batch call of nq queries, alone on the device: what a group of that size costs without any caller traffic
        rec["graph_batch_ms"] = {}
        for nq in (1, 8, 64, 160, 256, 1024):
            ts = []
            for _ in range(7):
                t1 = time.perf_counter(); g.search(qs[:nq], a.k, a.ef); ts.append((time.perf_counter() - t1) * 1e3)
            rec["graph_batch_ms"][nq] = round(sorted(ts)[3], 3)
        print("graph batch ms", rec["graph_batch_ms"], flush=True)
        # the same with 4 threads at once (the front's four lanes), 160 queries each
        import threading
        lat = []
        def worker():
            for _ in range(6):
                t1 = time.perf_counter(); g.search(qs[:160], a.k, a.ef); lat.append((time.perf_counter() - t1) * 1e3)
        th = [threading.Thread(target=worker) for _ in range(4)]
        [x.start() for x in th]; [x.join() for x in th]
        rec["graph_batch160_x4_threads_ms"] = round(sorted(lat)[len(lat) // 2], 3)
        print("graph 4 x 160 concurrently, ms per call", rec["graph_batch160_x4_threads_ms"], flush=True)
        for t in [int(x) for x in a.graph_callers.split(",")]:
            s0 = _callers.coalesce_stats("graph", g.handle)
            r = _callers.run("graph", g.handle, qs, a.k, threads=t, seconds=a.seconds, ef=a.ef)
            s1 = _callers.coalesce_stats("graph", g.handle)
            seen = r["count"] != 0xFFFFFFFD
            same = bool(np.array_equal(r["count"][seen], ref_c[seen]) and np.array_equal(r["rows"][seen], ref_r[seen]) and
                        np.array_equal(r["dist"][seen].view(np.uint32), ref_d[seen].view(np.uint32)))
            groups = s1["groups"] - s0["groups"]
            e = dict(callers=t, qps=round(r["qps"], 1), p50_us=round(r["p50_us"], 1), p99_us=round(r["p99_us"], 1), max_us=round(r["max_us"], 1),
                     calls=r["calls"], errors=r["errors"], mismatches=r["mismatches"], same_as_batch_call=same,
                     solo=s1["solo"] - s0["solo"], groups=groups, mean_group=round((s1["group_queries"] - s0["group_queries"]) / max(groups, 1), 1),
                     mean_group_pass_us=round((s1["group_pass_ns"] - s0["group_pass_ns"]) / max(groups, 1) / 1e3, 1),
                     lingers=s1["lingers"] - s0["lingers"], mean_linger_us=round((s1["linger_ns"] - s0["linger_ns"]) / max(s1["lingers"] - s0["lingers"], 1) / 1e3, 1))
            rec["graph"].append(e)
            print("graph", json.dumps(e), flush=True)
        rec["graph_shape"] = [a.graph_rows, a.dim, a.ef]
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
