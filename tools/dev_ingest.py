"""Ingest rates of the flat index: device-resident rows (qv_index_add_device), host rows (qv_index_add), with and without the
row-major copy: python tools/dev_ingest.py [rows] [dim]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, quiver_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
x = torch.randn((N, D), device="cuda", dtype=torch.float32)
hx = x.cpu().numpy()
sp = torch.cuda.current_stream().cuda_stream
for rm in (False, True):
    for metric in ("cosine", "l2"):
        idx = quiver_amd.DeviceIndex(D, metric, rowmajor=rm); idx.reserve(N + 1000)
        idx.add_device(x.data_ptr(), 1000, sp); torch.cuda.synchronize()
        t = time.perf_counter(); idx.add_device(x.data_ptr(), N, sp); torch.cuda.synchronize(); dt = time.perf_counter() - t
        print("add_device %s rowmajor=%s: %d x %d in %.2f ms = %.1f M rows/s, %.0f GB/s of rows read" % (metric, rm, N, D, dt * 1e3, N / dt / 1e6, N * D * 4 / dt / 1e9), flush=True)
        idx.close()
idx = quiver_amd.DeviceIndex(D, "cosine"); idx.reserve(N)
t = time.perf_counter(); idx.add(hx); dt = time.perf_counter() - t
print("add (host rows) cosine: %.1f ms = %.1f GB/s" % (dt * 1e3, N * D * 4 / dt / 1e9), flush=True)
