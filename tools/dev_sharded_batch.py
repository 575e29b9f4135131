"""A batch of queries through qv_sharded_search_device with co-located shards: python tools/dev_sharded_batch.py [shards] [nq] [rows] [k]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import quiver_amd
from quiver_amd import ShardedIndex

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
dim = 768
k = int(sys.argv[4]) if len(sys.argv) > 4 else 10
sh = ShardedIndex(dim, "cosine", devices=[0] * G, peer_copy=G > 1)
sh.add_synthetic(20260424, 0, n)
qi = quiver_amd.DeviceIndex(dim, "cosine"); qi.add_synthetic(20260425, 0, nq)
dq = torch.from_numpy(np.stack([qi.get_row(i) for i in range(nq)])).cuda()
dr = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
sp = torch.cuda.current_stream().cuda_stream
for env in ("off", None):
    sh.set_filter("off" if env else "auto")
    sh.search_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), sp); torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(5):
        t1 = time.perf_counter()
        sh.search_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), sp)
        host += time.perf_counter() - t1                # the call itself: enqueue only since round 5 (hand-backs are redone on the device)
    torch.cuda.synchronize()
    print("%d co-located shards, %d queries x %d x %d: %s %.3f ms/batch, of which the host is inside the call for %.3f ms" % (
        G, nq, n, dim, "exact multi-query scan per shard" if env else "filter + re-score per shard     ", (time.perf_counter() - t0) / 5 * 1e3, host / 5 * 1e3), flush=True)
