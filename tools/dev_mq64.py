"""Timing of the exact multi-query scan (k_flat_scan_mq64) on 256 x 1M x 768, with a digest of the results so that
variants of the kernel can be compared run to run:  python tools/dev_mq64.py [metric] [nq] [rows]"""
import hashlib
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
dim, k = 768, 10
idx = quiver_amd.DeviceIndex(dim, metric)
idx.add_synthetic(20260424, 0, rows)
qi = quiver_amd.DeviceIndex(dim, metric)
qi.add_synthetic(20260425, 0, nq)
hq = np.stack([qi.get_row(i) for i in range(nq)])
dq = torch.from_numpy(hq).cuda()
dr = torch.empty((nq, k), dtype=torch.int32, device="cuda")
dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
sp = torch.cuda.current_stream().cuda_stream
idx.search_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), sp)
torch.cuda.synchronize()
idx.profile(True)
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    idx.search_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), sp)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
ms, n = idx.profile_read()
h = hashlib.sha256(dr.cpu().numpy().tobytes() + dd.cpu().numpy().tobytes()).hexdigest()[:16]
# the single-query exact scan (one lane per row, scalar f64 chain) as the on-device cross-check for a few queries
ok = True
for i in (0, 1, nq // 2, nq - 1):
    r1, d1, _ = idx.search(hq[i:i + 1], k)
    ok &= r1[0].tolist() == dr[i].cpu().numpy().view(np.uint32).tolist() and d1[0].tobytes() == dd[i].cpu().numpy().tobytes()
print("mq64 %s nq=%d rows=%d: batch %.3f ms, kernel %.3f ms (x%d), f64-equivalent %.1f TFLOP/s, digest %s, equals single-query scan: %s, env %s"
      % (metric, nq, rows, dt * 1e3, ms / max(n, 1), n, 2.0 * nq * rows * dim / dt / 1e12, h, ok,
         {k_: v for k_, v in os.environ.items() if k_.startswith("QV_")}), flush=True)
