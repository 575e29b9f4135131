"""Timing of the fp32-MFMA filter + exact re-score batch (qv_index_search_batched_device) on 256 x 1M x 768:
python tools/dev_batched.py [metric] [nq] [rows] [dim] [k]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import quiver_amd

metric = sys.argv[1] if len(sys.argv) > 1 else "cosine"
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
dim = int(sys.argv[4]) if len(sys.argv) > 4 else 768
k = int(sys.argv[5]) if len(sys.argv) > 5 else 10
idx = quiver_amd.DeviceIndex(dim, metric, bf16_rows=os.environ.get("DEV_BF16_ROWS") == "1", filter=os.environ.get("DEV_FILTER") or None,
                             rowmajor=os.environ.get("DEV_ROWMAJOR") == "1")   # DEV_FILTER=fp32 | bf16x3 | bf16x1; DEV_ROWMAJOR=1: keep the row-major copy (the exact passes gather from it)
idx.add_synthetic(20260424, 0, rows)
qi = quiver_amd.DeviceIndex(dim, metric)
qi.add_synthetic(20260425, 0, nq)
hq = np.stack([qi.get_row(i) for i in range(nq)])
dq = torch.from_numpy(hq).cuda()
dr = torch.empty((nq, k), dtype=torch.int32, device="cuda")
dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
fl = torch.zeros((nq,), dtype=torch.int32, device="cuda")
sp = torch.cuda.current_stream().cuda_stream
idx.search_batched_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), fl.data_ptr(), sp)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = int(os.environ.get("DEV_REPS", "10"))     # (the clocks take ~8 launches to ramp after an idle gap: 50+ for a steady-state mean)
for _ in range(reps):
    idx.search_batched_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), fl.data_ptr(), sp)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
er = torch.empty((nq, k), dtype=torch.int32, device="cuda"); ed = torch.empty((nq, k), dtype=torch.float32, device="cuda")
idx.search_device(dq.data_ptr(), nq, k, er.data_ptr(), ed.data_ptr(), sp)
torch.cuda.synchronize()
same = bool(torch.equal(er, dr)) and er.cpu().numpy().tobytes() == dr.cpu().numpy().tobytes() and ed.cpu().numpy().tobytes() == dd.cpu().numpy().tobytes()
print("batched %s nq=%d rows=%d: %.3f ms/batch, %.1f TFLOP/s fp32-equivalent, flagged %d, identical to the exact scan: %s"
      % (metric, nq, rows, dt * 1e3, 2.0 * nq * rows * dim / dt / 1e12, int(fl.sum().item()), same), flush=True)
idx.search(hq, k, batched=True)
t0 = time.perf_counter()
for _ in range(3):
    idx.search(hq, k, batched=True)
print("  host pointers (qv_index_search_batched): %.3f ms/batch" % ((time.perf_counter() - t0) / 3 * 1e3), flush=True)
