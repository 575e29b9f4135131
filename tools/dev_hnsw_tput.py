"""Device-resident traversal throughput and host-pointer QPS on the 1M x 768 MaxLevel=1 graph (what bench.py's also.hnsw_1Mx768_maxlevel1 reports):
python tools/dev_hnsw_tput.py [efs]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.bench import bench_hnsw_build as B
efs = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "64,128,256").split(","))
r = B.run(rows=1_000_000, max_level=1, efs=efs, cpu_queries=0)
print("build", json.dumps(r["build"]))
for e in r["search"]:
    t = e["graph_traversal"]
    print("ef %d: device-resident %.0f QPS (%.2f ms per 8192), host pointers %.0f QPS, %.0f GB/s gathered, flagged %d" % (
        e["ef_search"], t["qps_device_resident"], t["batch_ms"], t["qps_host_pointers_incl_exact_heap_redo"], t["gathered_GBps"], t["flagged_for_exact_heap"]), flush=True)
