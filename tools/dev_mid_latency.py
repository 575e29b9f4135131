"""Lone-query latency of the flat index on small and medium collections (host pointers, one query per call) beside the scan kernels' own
time and the HBM floor:  python tools/dev_mid_latency.py [dim] [rows,rows,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, quiver_amd
from tests import _oracle as O
D = int(sys.argv[1]) if len(sys.argv) > 1 else 768
sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "10000,16000,20000,30000,60000,100000,200000,300000,500000,1000000").split(",")]
qs = O.gen_rows(20260425, 0, 64, D)
for n in sizes:
    idx = quiver_amd.DeviceIndex(D, "cosine"); idx.add_synthetic(20260424, 0, n)
    for i in range(20): idx.search(qs[i % 64:i % 64 + 1], 10)
    lat = []
    for i in range(300):
        t = time.perf_counter(); idx.search(qs[i % 64:i % 64 + 1], 10); lat.append(time.perf_counter() - t)
    lat.sort()
    kern = float("nan")
    if hasattr(idx, "profile"):
        idx.profile(True)
        for i in range(50): idx.search(qs[i % 64:i % 64 + 1], 10)
        ms, launches = idx.profile_read(); idx.profile(False)
        kern = ms / max(launches, 1) * 1e3 if launches else float("nan")
    floor = n * D * 4 / 7.15e12 * 1e6
    print("rows %8d x %d: p50 %7.1f us  p10 %7.1f  scan kernels %7.1f us per launch   (stream floor %6.1f us)" % (n, D, lat[150] * 1e6, lat[30] * 1e6, kern, floor), flush=True)
    idx.close()
