"""Randomised stress of qv_sharded_search_device's device-side redo: co-located shards over corpora whose tiny clusters make guessed bounds fail, host and
device calls against one index's exact scan.  python tools/stress_sharded_redo.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import quiver_amd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
runs = bad = flagged = 0
while time.time() < t_end:
    metric = str(rng.choice(["cosine", "dot_product", "euclidean", "squared_euclidean"]))
    dim = int(rng.choice([32, 64, 128, 256]))
    G = int(rng.choice([1, 2, 3]))
    per = int(rng.choice([140_000, 200_000]))
    n = G * per
    cl = int(rng.choice([128, 256, 384]))
    centers = rng.standard_normal((n // 2 // cl + 1, dim)).astype(np.float32)
    clustered = np.concatenate([c + 0.02 * rng.standard_normal((cl, dim)).astype(np.float32) for c in centers])[: n // 2]
    rows = np.concatenate([clustered, rng.standard_normal((n - len(clustered), dim)).astype(np.float32)])
    if rng.integers(2): rows = rows[::-1].copy()
    one = quiver_amd.DeviceIndex(dim, metric, filter="off"); one.add(rows)
    sh = quiver_amd.ShardedIndex(dim, metric, devices=[0] * G, peer_copy=G > 1)
    gids = sh.add(rows)
    pos = np.full(int(gids.max()) + 1, -1, np.int64); pos[gids] = np.arange(n)
    for _ in range(3):
        nq = int(rng.choice([9, 48, 100, 256]))
        k = int(rng.choice([16, 40, 64, 65, 100, 300, 1000]))
        qs = np.concatenate([centers[rng.integers(0, len(centers), size=nq // 2)], rng.standard_normal((nq - nq // 2, dim)).astype(np.float32)]).astype(np.float32)
        er, ed, _ = one.search(qs, k)
        r, d, c = sh.search(qs, k)
        dq = torch.from_numpy(qs).cuda()
        dr = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        sh.search_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), torch.cuda.current_stream().cuda_stream)
        sh.sync(); torch.cuda.synchronize()
        ok = (c == k).all() and np.array_equal(pos[r], er.astype(np.int64)) and d.tobytes() == ed.tobytes() \
            and np.array_equal(dr.cpu().numpy().view(np.uint32), r) and dd.cpu().numpy().tobytes() == d.tobytes()
        runs += 1
        if not ok:
            bad += 1
            print("MISMATCH metric=%s dim=%d shards=%d per=%d cluster=%d nq=%d k=%d" % (metric, dim, G, per, cl, nq, k), flush=True)
    sh.close(); one.close()
print("%d sharded batches (host + device call each), %d mismatches" % (runs, bad))
sys.exit(1 if bad else 0)
