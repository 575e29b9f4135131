"""Shared passes on short collections: one host-pointer call of nq queries, ms per call (QV_SCAN_SPLIT_MQ_MAX=1: the multi-query kernels of before)
python tools/dev_split_mq.py [dim]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, quiver_amd
from tests import _oracle as O
D = int(sys.argv[1]) if len(sys.argv) > 1 else 768
qs = O.gen_rows(20260425, 0, 64, D)
for n in (10_000, 30_000, 100_000, 160_000):
    idx = quiver_amd.DeviceIndex(D, "cosine"); idx.add_synthetic(20260424, 0, n)
    out = []
    for nq in (1, 2, 4, 8, 16, 32, 64):
        for _ in range(10): idx.search(qs[:nq], 10)
        ts = []
        for _ in range(100):
            t = time.perf_counter(); idx.search(qs[:nq], 10); ts.append(time.perf_counter() - t)
        ts.sort(); out.append("%d: %.0f" % (nq, ts[50] * 1e6))
    print("rows %7d x %d, us per call of nq queries: %s" % (n, D, "  ".join(out)), flush=True)
    idx.close()
