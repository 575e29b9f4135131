"""Batched-query measurement (BASELINE configs[2] shape: 256 queries x 1M x 768).
python tests/bench/bench_batched.py [--rows 1000000] [--nq 256] [--metric cosine] [--reps 5]"""
import argparse, json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import quiver_amd
from tests import _oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000); ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--nq", type=int, default=256); ap.add_argument("--k", type=int, default=10)
ap.add_argument("--metric", default="cosine"); ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--check", type=int, default=4)
ap.add_argument("--mfma", action="store_true", help="host-pointer qv_index_search_batched (MFMA filter + exact re-score)")
a = ap.parse_args()
idx = quiver_amd.DeviceIndex(a.dim, a.metric); idx.reserve(a.rows); idx.add_synthetic(20260424, 0, a.rows)
qs = O.gen_rows(20260425, 0, a.nq, a.dim)
dq = torch.from_numpy(qs).cuda()
dr = torch.empty((a.nq, a.k), dtype=torch.int32, device="cuda"); dd = torch.empty((a.nq, a.k), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
if a.mfma:
    R, D, C_ = idx.search(qs, a.k, batched=True)
    idx.profile(True)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        R, D, C_ = idx.search(qs, a.k, batched=True)
    dt = (time.perf_counter() - t0) / a.reps
    ms, n = idx.profile_read()
    dr = torch.from_numpy(R.view(np.int32).copy()).cuda(); dd = torch.from_numpy(D.copy()).cuda()
else:
    idx.search_device(dq.data_ptr(), a.nq, a.k, dr.data_ptr(), dd.data_ptr(), s); torch.cuda.synchronize()
    idx.profile(True)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        idx.search_device(dq.data_ptr(), a.nq, a.k, dr.data_ptr(), dd.data_ptr(), s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    ms, n = idx.profile_read()
ok = True
if a.check:
    corpus = O.gen_rows(20260424, 0, a.rows, a.dim)
    R, D = dr.cpu().numpy().view(np.uint32), dd.cpu().numpy()
    mid = quiver_amd.metric_id(a.metric)
    for i in list(range(a.check)) + [a.nq - 1]:
        er, ed = O.exact_search(mid, corpus, qs[i], a.k)
        ok &= bool(np.array_equal(R[i], er) and np.array_equal(D[i].view(np.uint32), ed.view(np.uint32)))
print(json.dumps({"path": "mfma-filter+exact-rescore (host pointers, PCIe + sync included)" if a.mfma else "exact multi-query scan (device pointers)", "workload": f"batched {a.nq} x {a.rows}x{a.dim} {a.metric} k={a.k}", "batch_ms": dt * 1e3, "qps": a.nq / dt,
                  "scan_kernel_ms": ms / max(n, 1), "gflop_equiv": 2.0 * a.nq * a.rows * a.dim / 1e9,
                  "tflops_equiv": 2.0 * a.nq * a.rows * a.dim / dt / 1e12, "bit_exact_vs_oracle": ok}))
