"""Where a small traversal batch's time goes (QV_TRACE=1 prints pass 1 / pass 2 of every qv_graph_search call)."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quiver_amd
from quiver_amd.device_index import DeviceGraph
from tests import _oracle as O

n, dim, ef, k = int(os.environ.get("ROWS", 1000000)), 768, 128, 10
gi = quiver_amd.DeviceIndex(dim, "cosine", rowmajor=True)
gi.add_synthetic(20260424, 0, n)
g = DeviceGraph.build(gi, np.zeros(n, np.int8), m=16, max_m0=32, ef_construction=200)
qs = O.gen_rows(20260425, 0, 1024, dim)
r, d, c, ev = g.search(qs, k, ef, with_evals=True)
print("evals: mean %.0f  p50 %.0f  p99 %.0f  max %d" % (ev.mean(), np.median(ev), np.percentile(ev, 99), ev.max()), flush=True)
for nq in (1, 8, 64, 160, 256, 1024):
    print("---- nq", nq, "max evals in batch", int(ev[:nq].max()), flush=True)
    for _ in range(3):
        t = time.perf_counter(); g.search(qs[:nq], k, ef); print("  call %.3f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
print("---- 4 threads x 160", flush=True)
def w(i):
    for _ in range(3):
        t = time.perf_counter(); g.search(qs[i * 160:(i + 1) * 160], k, ef); print("  thread %d call %.3f ms" % (i, (time.perf_counter() - t) * 1e3), flush=True)
th = [threading.Thread(target=w, args=(i,)) for i in range(4)]
[x.start() for x in th]; [x.join() for x in th]
