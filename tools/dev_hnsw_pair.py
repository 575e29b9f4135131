"""device-resident traversal rate at 1M x 768 (graph built once), QV_HNSW_PAIR variants in child processes"""
import json, os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tests.bench.bench_hnsw_build import run
    r = run(rows=int(os.environ.get("ROWS", 1000000)), efs=(64, 128, 200), nq=8192, cpu_queries=0, max_level=1)
    print("PAIR=%s build %.2f s redo %d | " % (os.environ.get("QV_HNSW_PAIR", "1"), r["build"]["seconds"], r["build"]["searches_redone_exact_heap"]) +
          " | ".join("ef %d: %.0f k QPS %.2f TB/s host %.0f k" % (x["ef_search"], x["graph_traversal"]["qps_device_resident"] / 1e3, x["graph_traversal"]["gathered_GBps"] / 1e3,
                                                                   x["graph_traversal"]["qps_host_pointers_incl_exact_heap_redo"] / 1e3) for x in r["search"]), flush=True)
else:
    for p in sys.argv[1:] or ["2", "1", "3"]:
        env = dict(os.environ); env["QV_HNSW_PAIR"] = p
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
        print(out.stdout.strip().split("\n")[-1] if out.stdout.strip() else out.stderr[-600:], flush=True)
