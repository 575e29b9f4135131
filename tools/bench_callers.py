"""Concurrent single-query callers through the C ABI (native threads, tools/native/qv_callers.cpp): QPS and latency by caller
count on the flat index and on the device HNSW graph.  This is the traffic the reference's unchanged Go host produces
(pkg/core/collection.go:647; pkg/core/db.go:805-828; pkg/hnsw/hnsw.go:602-606).

    python tools/bench_callers.py [--rows 1000000] [--dim 768] [--graph-rows 1000000] [--seconds 2] [--out gpurun_out/callers.json]
QV_COALESCE=0 in the environment turns the sharing off (every call on its own, as before round 5)."""
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
                     solo=s1["solo"] - s0["solo"], groups=groups, mean_group=round((s1["group_queries"] - s0["group_queries"]) / max(groups, 1), 1))
            rec["flat"].append(e)
            print("flat ", json.dumps(e), flush=True)
        rec["flat_shape"] = [a.rows, a.dim]
        idx.close()
    if a.graph_rows:
        from quiver_amd.device_index import DeviceGraph
        gi = quiver_amd.DeviceIndex(a.dim, "cosine", rowmajor=True)
        gi.add_synthetic(20260424, 0, a.graph_rows)
        t0 = time.time()
        g = DeviceGraph.build(gi, np.zeros(a.graph_rows, np.int8), m=16, max_m0=32, ef_construction=200)
        rec["graph_build_s"] = round(time.time() - t0, 2)
        ref_r, ref_d, ref_c = g.search(qs, a.k, a.ef)
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
                     solo=s1["solo"] - s0["solo"], groups=groups, mean_group=round((s1["group_queries"] - s0["group_queries"]) / max(groups, 1), 1))
            rec["graph"].append(e)
            print("graph", json.dumps(e), flush=True)
        rec["graph_shape"] = [a.graph_rows, a.dim, a.ef]
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
