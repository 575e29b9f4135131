"""Selection cost against k: the reference asks for k > 64 routinely — the negative-example branches fetch max(2k, 30)
(hybrid_index.go:516-522, hnsw/adapter.go:353-359), BatchSearch takes any k (hybrid_index.go:677-811).

    python tools/bench_kselect.py [--rows 1000000] [--dim 768] [--ks 10,64,65,100,256,1000,4096] [--nqs 1,256] [--check]

One line of JSON per (nq, k): ms per call with device-resident queries and results (qv_index_search_device for one query,
the library's own routing — qv_index_search's — for a batch), and, with --check, whether rows and distance bits equal the
full ranking's first k (the k = N radix path, itself oracle-tested in tests/test_gpu_flat.py)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quiver_amd

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000); ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--ks", default="10,64,65,100,256,1000,4096"); ap.add_argument("--nqs", default="1,256")
ap.add_argument("--reps", type=int, default=10); ap.add_argument("--metric", default="cosine")
ap.add_argument("--check", action="store_true"); ap.add_argument("--bf16-rows", action="store_true")
a = ap.parse_args()
idx = quiver_amd.DeviceIndex(a.dim, a.metric, bf16_rows=a.bf16_rows); idx.reserve(a.rows); idx.add_synthetic(20260424, 0, a.rows)
s = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device="cuda"); gen.manual_seed(5)
for nq in [int(x) for x in a.nqs.split(",")]:
    q = torch.randn(nq, a.dim, device="cuda", generator=gen)
    q = q / q.norm(dim=1, keepdim=True)
    full = None
    for k in [int(x) for x in a.ks.split(",")]:
        if k > a.rows:
            continue
        dr = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        flags = torch.zeros(nq, dtype=torch.int32, device="cuda")

        def call():
            # the library's own routing for a batch (what qv_index_search does with host pointers), device-resident
            if nq >= 9:
                try:
                    idx.search_batched_device(q.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), flags.data_ptr(), s)
                    return "filter+rescore"
                except Exception:                                     # QV_ERR_UNSUPPORTED: the exact scans
                    pass
            idx.search_device(q.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), s)
            return "exact"
        reps = a.reps if (nq == 1 or k <= 64) else max(2, a.reps // 5)
        path = call(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        rec = {"rows": a.rows, "dim": a.dim, "metric": a.metric, "nq": nq, "k": k, "path": path, "ms": round(ms, 4), "redo_flags": int(flags.sum().item())}
        if a.check:
            if full is None:                                          # first kmax of the full ranking, one query at a time (k = N: the radix sort)
                kmax = max(int(x) for x in a.ks.split(",") if int(x) <= a.rows)
                fr = torch.empty((nq, kmax), dtype=torch.int32, device="cuda"); fd = torch.empty((nq, kmax), dtype=torch.float32, device="cuda")
                tr = torch.empty((1, a.rows), dtype=torch.int32, device="cuda"); td = torch.empty((1, a.rows), dtype=torch.float32, device="cuda")
                for i in range(min(nq, 8)):
                    idx.search_device(q[i:i + 1].data_ptr(), 1, a.rows, tr.data_ptr(), td.data_ptr(), s); torch.cuda.synchronize()
                    fr[i] = tr[0, :kmax]; fd[i] = td[0, :kmax]
                full = (fr, fd)
            m = min(nq, 8)
            rec["equals_full_ranking"] = bool(torch.equal(dr[:m], full[0][:m, :k]) and torch.equal(dd[:m].view(torch.int32), full[1][:m, :k].view(torch.int32)))
        print(json.dumps(rec), flush=True)
