"""The batched filter path on structured corpora (unit vectors on an r-dimensional subspace of R^768, optionally with near-duplicate
clusters): time, flagged queries and identity with the exact scan, per filter mode.  python tools/dev_batched_structured.py [r] [rows] [nq] [k] [modes, e.g. 3 or 3,2,1]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import quiver_amd

r = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 256
D = 768
k = int(sys.argv[4]) if len(sys.argv) > 4 else 10
modes = (sys.argv[5] if len(sys.argv) > 5 else "3,2,1").split(",")
gen = torch.Generator(device="cuda"); gen.manual_seed(20260424)
basis = torch.linalg.qr(torch.randn((D, r), generator=gen, device="cuda", dtype=torch.float64))[0].T.contiguous()


def lowrank(n, centers=None, spread=0.0):
    z = torch.randn((n, r), generator=gen, device="cuda", dtype=torch.float64)
    if centers is not None:          # clustered: a centre plus a small perturbation
        z = centers[torch.randint(0, centers.shape[0], (n,), generator=gen, device="cuda")] + spread * z
    x = z @ basis
    return (x / x.norm(dim=1, keepdim=True)).to(torch.float32).contiguous()


sp = torch.cuda.current_stream().cuda_stream
for label, centers, spread in (("subspace r=%d" % r, None, 0.0),
                               ("1000 tight clusters (spread 0.02) on the subspace", torch.randn((1000, r), generator=gen, device="cuda", dtype=torch.float64), 0.02)):
    idx = quiver_amd.DeviceIndex(D, "cosine")
    idx.reserve(N)
    for s0 in range(0, N, 250_000):
        x = lowrank(min(250_000, N - s0), centers, spread)
        idx.add_device(x.data_ptr(), x.shape[0], sp)
        torch.cuda.synchronize()
    dq = lowrank(nq, centers, spread)
    er = torch.empty((nq, k), dtype=torch.int32, device="cuda"); ed = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    idx.search_device(dq.data_ptr(), nq, k, er.data_ptr(), ed.data_ptr(), sp)
    torch.cuda.synchronize()
    for mode in modes:
        idx.set_filter(int(mode))
        dr = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        fl = torch.zeros((nq,), dtype=torch.int32, device="cuda")
        idx.search_batched_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), fl.data_ptr(), sp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            idx.search_batched_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), fl.data_ptr(), sp)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        ok = (fl == 0)
        same = bool(torch.equal(er[ok], dr[ok])) and ed[ok].cpu().numpy().tobytes() == dd[ok].cpu().numpy().tobytes()
        print("%-55s filter %s: %.3f ms/batch, flagged %d of %d, unflagged identical to the exact scan: %s" % (label, mode, dt * 1e3, int((~ok).sum().item()), nq, same), flush=True)
    del idx
