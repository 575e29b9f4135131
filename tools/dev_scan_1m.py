"""one query per call over N x 768, device-resident, back to back on one stream: ms per query and the scan kernel's own time
(QV_SCAN_FUSE=0: scan + k_merge_lists as two launches; default: one launch, the last workgroup merges)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
import quiver_amd

n = int(os.environ.get("ROWS", 1000000)); dim, k = 768, int(os.environ.get("K", 10))
idx = quiver_amd.DeviceIndex(dim, "cosine")
idx.add_synthetic(20260424, 0, n)
qg = quiver_amd.DeviceIndex(dim, "cosine"); qg.add_synthetic(20260425, 0, 64)
qs = np.stack([qg.get_row(i) for i in range(64)]); d_q = torch.from_numpy(qs).cuda()
d_r = torch.empty((k,), dtype=torch.int32, device="cuda"); d_d = torch.empty((k,), dtype=torch.float32, device="cuda")
sp = torch.cuda.current_stream().cuda_stream
for j in range(50):
    idx.search_device(d_q.data_ptr() + (j % 64) * dim * 4, 1, k, d_r.data_ptr(), d_d.data_ptr(), sp)
torch.cuda.synchronize()
res = []
for rep in range(5):
    t = time.perf_counter()
    for j in range(500):
        idx.search_device(d_q.data_ptr() + (j % 64) * dim * 4, 1, k, d_r.data_ptr(), d_d.data_ptr(), sp)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t) / 500 * 1e3)
idx.profile(True)
for j in range(200):
    idx.search_device(d_q.data_ptr() + (j % 64) * dim * 4, 1, k, d_r.data_ptr(), d_d.data_ptr(), sp)
torch.cuda.synchronize()
ms, cnt = idx.profile_read()
b = n * (dim * 4 + 8)
print("fuse=%s rows=%d k=%d ms/query %s  best %.4f (%.3f of 8 TB/s)  kernel %.4f ms (%.3f)" % (os.environ.get("QV_SCAN_FUSE", "1"), n, k, ["%.4f" % x for x in res], min(res),
      b / (min(res) * 1e-3) / 8e12, ms / cnt, b / (ms / cnt * 1e-3) / 8e12), flush=True)
