"""Host-side cost of one qv_sharded_search_device: G shards CO-LOCATED on one GPU (peer-copy exchange), tiny shards so that the
GPU work is ~nothing and the time per call is the host's — what a real 8-GPU node pays per search on top of its 0.48 ms scan.
    python tools/dev_sharded_hostcost.py [G=8] [rows_per_shard=64] [calls=2000] [threads=1]
Prints µs per search (enqueue only, then including the final sync) and per shard."""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quiver_amd
from quiver_amd.device_index import ShardedIndex

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 64
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
threads = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dim, k = 768, 10
sh = ShardedIndex(dim, "cosine", devices=[0] * G, peer_copy=True)
sh.add_synthetic(20260424, 0, rows * G)
q = torch.randn(16, dim, device="cuda")
outs = [(torch.empty((1, k), dtype=torch.int32, device="cuda"), torch.empty((1, k), dtype=torch.float32, device="cuda")) for _ in range(threads)]
streams = [torch.cuda.Stream() for _ in range(threads)]
for t in range(threads):
    sh.search_device(q.data_ptr(), 1, k, outs[t][0].data_ptr(), outs[t][1].data_ptr(), streams[t].cuda_stream)
sh.sync(); torch.cuda.synchronize()

def work(t, n):
    for i in range(n):
        sh.search_device(q.data_ptr() + (i % 16) * dim * 4, 1, k, outs[t][0].data_ptr(), outs[t][1].data_ptr(), streams[t].cuda_stream)

t0 = time.perf_counter()
if threads == 1:
    work(0, calls)
else:
    th = [threading.Thread(target=work, args=(t, calls // threads)) for t in range(threads)]
    [x.start() for x in th]; [x.join() for x in th]
t1 = time.perf_counter()
sh.sync(); torch.cuda.synchronize()
t2 = time.perf_counter()
n = calls // threads * threads
print(json.dumps({"shards_co_located": G, "rows_per_shard": rows, "calls": n, "caller_threads": threads,
                  "enqueue_us_per_search": round((t1 - t0) / n * 1e6, 2), "with_final_sync_us_per_search": round((t2 - t0) / n * 1e6, 2),
                  "enqueue_us_per_search_per_shard": round((t1 - t0) / n * 1e6 / G, 2)}))
