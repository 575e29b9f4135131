"""The reference-faithful CPU port (oracle/qv_oracle.c qvo_faithful_search: rows individually allocated behind a string-keyed hash map, scalar
float64 distance through a function pointer, full sort of all N records — what ExactIndex.Search does, exact.go:92-133) at the FULL size of
the headline workload, one thread: bench.py times it on the first 500 000 rows and scales by N, which ignores that the sort is not
linear.  This measures how much that flatters the CPU.   python tools/cpu_baseline_full.py [--rows 10000000] [--out profiles/...json]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import _oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000); ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--queries", type=int, default=3); ap.add_argument("--out", default="")
a = ap.parse_args()
f = O.Faithful(0, a.dim)
qs = O.gen_rows(20260425, 0, a.queries, a.dim)
rec = {"what": "qvo_faithful_search, 1 thread, k = 10, cosine, rows of the bench corpus (seed 20260424)", "dim": a.dim, "points": []}
done = 0
t_build = time.perf_counter()
for stop in [s for s in (500_000, 2_000_000, 5_000_000, 10_000_000) if s <= a.rows]:
    while done < stop:
        m = min(250_000, stop - done)
        rows = O.gen_rows(20260424, done, m, a.dim)
        for i in range(m):
            f.insert("v%d" % (done + i), rows[i])
        done += m
    f.search(qs[0], 10)
    t0 = time.perf_counter()
    for q in qs:
        ids, d = f.search(q, 10)
    dt = (time.perf_counter() - t0) / len(qs)
    p = {"rows": done, "s_per_query": dt, "rows_per_s": done / dt, "qps": 1.0 / dt}
    if rec["points"]:
        p["rows_per_s_relative_to_500k"] = p["rows_per_s"] / rec["points"][0]["rows_per_s"]
    rec["points"].append(p)
    print(json.dumps(p), flush=True)
rec["build_s"] = time.perf_counter() - t_build
if a.out:
    with open(a.out, "w") as fh:
        json.dump(rec, fh, indent=1)
