"""Throughput of the search entry points away from the benchmark shapes (odd query counts, other dimensions, other metrics):
python tools/sweep_shapes.py [flat|exact|filter|all].  Device-resident queries and results, k = 10."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import quiver_amd

what = sys.argv[1] if len(sys.argv) > 1 else "all"
k = 10
sp = torch.cuda.current_stream().cuda_stream


def queries(dim, metric, nq):
    qi = quiver_amd.DeviceIndex(dim, metric)
    qi.add_synthetic(20260425, 0, nq)
    return torch.from_numpy(np.stack([qi.get_row(i) for i in range(nq)])).cuda()


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


if what in ("flat", "all"):
    print("single-query exact scan, ~3 GB of rows per shape", flush=True)
    for metric in ("cosine", "l2", "dot", "l1", "l2sq"):
        for dim in ((64, 100, 128, 384, 768, 1000, 1536, 3072) if metric == "cosine" else (128, 768)):
            rows = 768_000_000 // dim
            idx = quiver_amd.DeviceIndex(dim, metric)
            idx.add_synthetic(20260424, 0, rows)
            dq = queries(dim, metric, 1)
            dr = torch.empty((1, k), dtype=torch.int32, device="cuda"); dd = torch.empty((1, k), dtype=torch.float32, device="cuda")
            ms = timed(lambda: idx.search_device(dq.data_ptr(), 1, k, dr.data_ptr(), dd.data_ptr(), sp), 20)
            print("  %-6s dim %4d rows %8d: %.3f ms  %.0f GB/s" % (metric, dim, rows, ms, rows * (dim * 4 + (8 if metric == "cosine" else 0)) / ms / 1e6), flush=True)
            del idx

if what in ("exact", "filter", "all"):
    dim, rows = 768, 1_000_000
    for metric in ("cosine", "l2"):
        idx = quiver_amd.DeviceIndex(dim, metric)
        idx.add_synthetic(20260424, 0, rows)
        allq = queries(dim, metric, 1024)
        for mode in ("exact", "filter"):
            if what not in (mode, "all"):
                continue
            print("%s, %s, 1M x 768: query count -> ms per batch (ms per 32 queries)" % (metric, "exact multi-query scan" if mode == "exact" else "bfloat16 filter + exact re-score (qv_index_search_batched_device; below its query minimum it is the exact scan)"), flush=True)
            for nq in (2, 8, 9, 16, 17, 31, 32, 33, 48, 64, 65, 100, 128, 129, 192, 255, 256, 257, 320, 384, 512, 600, 1000, 1024):
                dq = allq[:nq].contiguous()
                dr = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
                fl = torch.zeros((nq,), dtype=torch.int32, device="cuda")
                if mode == "exact":
                    ms = timed(lambda: idx.search_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), sp), 5)
                else:
                    try:
                        ms = timed(lambda: idx.search_batched_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), fl.data_ptr(), sp), 5)
                    except quiver_amd.QvError:
                        print("  %5d: not applicable (qv_index_search takes the exact scan)" % nq, flush=True)
                        continue
                print("  %5d: %8.3f  (%.3f)" % (nq, ms, ms / nq * 32), flush=True)
        del idx
