"""BASELINE configs[3] as written: HNSW M=16 (MaxM0=32), efConstruction=200, MaxLevel=16 over N x 768 random unit vectors,
the graph INSERTION-BUILT on the device (qv_graph_build: pkg/hnsw/hnsw.go:266-468 in batches), then searched at
efSearch = 64 / 128 / 256 / 512 with queries and results resident on the device.

Reported per efSearch: QPS, distance evaluations, gathered GB/s, recall@k against the exact top-k (flat scan), and — on the
IDENTICAL graph (qv_graph_export -> qvo_hnsw_load_graph) — the CPU traversal's QPS on one core with a bit-for-bit
comparison of its results with the device's for the sampled queries.

  python tests/bench/bench_hnsw_build.py [--rows 1000000] [--dim 768] [--efs 64,128,256,512] [--nq 8192] [--cpu-queries 100]

`run()` is what bench.py calls for its `also.hnsw_build_and_search` entry.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np


def run(rows=1_000_000, dim=768, metric="cosine", m=16, efc=200, max_level=16, batch_max=16384, ramp_div=16, efs=(64, 128, 256, 512),
        nq=8192, k=10, cpu_queries=100, device=0, level_seed=1, corpus_seed=20260424, query_seed=20260425, cpu_build_rows=0, intrinsic_dim=0,
        callers=()):
    import torch                      # before libqv: both must share one HIP runtime (torch bundles its own)
    import quiver_amd
    from quiver_amd.device_index import DeviceGraph, random_levels
    torch.cuda.set_device(device)
    N, D = rows, dim
    out = {"workload": "HNSW M=%d MaxM0=%d efConstruction=%d MaxLevel=%d, %dx%d %s, k=%d (BASELINE configs[3])" % (m, 2 * m, efc, max_level, N, D, metric, k),
           "graph": "insertion-built on the device: qv_graph_build, batches of <= %d nodes (never more than 1/%d of the nodes already linked), "
                    "every construction search against the graph before its batch, links applied in node order "
                    "(hnsw.go:266-468; batch of one == Insert)" % (batch_max, ramp_div)}
    idx = quiver_amd.DeviceIndex(D, metric, device=device, rowmajor=True)
    idx.reserve(N)
    if intrinsic_dim:
        # Structured data: unit vectors on an `intrinsic_dim`-dimensional subspace of R^D (Gaussian coordinates under a random
        # orthonormal basis) — the regime embeddings live in and the one where a graph index finds neighbours; uniformly random
        # 768-d unit vectors (BASELINE's synthetic corpus) have no neighbourhood structure for any graph to follow.
        gen = torch.Generator(device="cuda"); gen.manual_seed(corpus_seed)
        basis = torch.linalg.qr(torch.randn((D, intrinsic_dim), generator=gen, device="cuda", dtype=torch.float64))[0].T.contiguous()
        def lowrank(n):
            z = torch.randn((n, intrinsic_dim), generator=gen, device="cuda", dtype=torch.float64)
            x = z @ basis
            return (x / x.norm(dim=1, keepdim=True)).to(torch.float32).contiguous()
        host_rows = np.empty((N, D), np.float32) if cpu_queries else None
        for s0 in range(0, N, 250_000):
            x = lowrank(min(250_000, N - s0))
            idx.add_device(x.data_ptr(), x.shape[0], torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            if host_rows is not None:
                host_rows[s0:s0 + x.shape[0]] = x.cpu().numpy()
        out["corpus"] = "unit vectors on a random %d-dimensional subspace of R^%d (Gaussian coordinates)" % (intrinsic_dim, D)
    else:
        idx.add_synthetic(corpus_seed, 0, N)
    levels = random_levels(N, max_level, level_seed)
    t0 = time.perf_counter()
    g = DeviceGraph.build(idx, levels, m=m, max_m0=2 * m, ef_construction=efc, batch_max=batch_max, ramp_div=ramp_div)
    t_build = time.perf_counter() - t0
    st, info = g.stats(), g.info()
    out["build"] = {"seconds": t_build, "nodes_per_s": N / t_build, "batches": st["build_batches"], "searches_redone_exact_heap": st["build_redo"],
                    "entry": info["entry"], "top_level": info["cur_level"], "upper_level_lists": info["n_up_blocks"]}

    # exact top-k of every query: the recall denominator (the flat scan is the oracle-checked exact path)
    dq = torch.empty((nq, D), dtype=torch.float32, device="cuda")
    if intrinsic_dim:
        dq.copy_(lowrank(nq)); hq = dq.cpu().numpy()
    else:
        qg = quiver_amd.DeviceIndex(D, metric, device=device)
        qg.add_synthetic(query_seed, 0, nq)
        hq = np.stack([qg.get_row(i) for i in range(nq)])
        dq.copy_(torch.from_numpy(hq))
        qg.close()
    er, ed, _ = idx.search(hq, k, batched=True)
    # 4 nq different queries for the larger call of the throughput measurement (the first nq are the ones above)
    big_q = None
    try:
        big_q = torch.empty((4 * nq, D), dtype=torch.float32, device="cuda")
        big_q[:nq].copy_(dq)
        if intrinsic_dim:
            big_q[nq:].copy_(lowrank(3 * nq))
        else:
            qg = quiver_amd.DeviceIndex(D, metric, device=device)
            qg.add_synthetic(query_seed, nq, 3 * nq)
            for s0 in range(0, 3 * nq, 4096):
                big_q[nq + s0:nq + min(3 * nq, s0 + 4096)].copy_(torch.from_numpy(qg.get_rows(np.arange(s0, min(3 * nq, s0 + 4096), dtype=np.uint32))))
            qg.close()
    except Exception:                                      # noqa: BLE001
        big_q = None

    dr = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    dc = torch.empty(nq, dtype=torch.int32, device="cuda"); de = torch.empty(nq, dtype=torch.int32, device="cuda")
    sp = torch.cuda.current_stream().cuda_stream
    sweep = []
    host = {}
    for ef in efs:
        g.search_device(dq.data_ptr(), nq, k, ef, dr.data_ptr(), dd.data_ptr(), dc.data_ptr(), de.data_ptr(), sp)
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            g.search_device(dq.data_ptr(), nq, k, ef, dr.data_ptr(), dd.data_ptr(), dc.data_ptr(), de.data_ptr(), sp)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / reps
        cnt = dc.cpu().numpy().view(np.uint32)
        flagged = int((cnt == 0xFFFFFFFE).sum())
        # four times the queries per call (all different): the call's last traversals, which leave CUs idle, weigh a quarter as much
        qps_4x = None
        if ef == 128 and big_q is not None:
            nb = big_q.shape[0]
            br = torch.empty((nb, k), dtype=torch.int32, device="cuda"); bd = torch.empty((nb, k), dtype=torch.float32, device="cuda")
            bc = torch.empty(nb, dtype=torch.int32, device="cuda"); be = torch.empty(nb, dtype=torch.int32, device="cuda")
            g.search_device(big_q.data_ptr(), nb, k, ef, br.data_ptr(), bd.data_ptr(), bc.data_ptr(), be.data_ptr(), sp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                g.search_device(big_q.data_ptr(), nb, k, ef, br.data_ptr(), bd.data_ptr(), bc.data_ptr(), be.data_ptr(), sp)
            torch.cuda.synchronize()
            qps_4x = nb / ((time.perf_counter() - t0) / 2)
            same_4x = bool(torch.equal(br[:nq], dr) and torch.equal(bd[:nq].view(torch.int32), dd.view(torch.int32)))
            del br, bd, bc, be
        # complete results (flagged queries redone by the exact-heap kernel) through the host-pointer form
        t0 = time.perf_counter()
        r, d, c, ev = g.search(hq, k, ef, with_evals=True)
        t_host = time.perf_counter() - t0
        host[ef] = (r, d, c, ev)
        under = np.flatnonzero(c < k)
        # HNSW.Search completes an under-filled graph search by brute force over every node (hnsw.go:676-710): the exact top-k.
        # On the device that is one batched flat-scan call for the under-filled queries.
        t_top = 0.0
        if under.size:
            t0 = time.perf_counter()
            idx.search(hq[under], k, batched=True)
            t_top = time.perf_counter() - t0
        hit_graph = sum(len(set(r[i, :min(int(c[i]), k)].tolist()) & set(er[i].tolist())) for i in range(nq))
        hit_full = sum(k if c[i] < k else len(set(r[i].tolist()) & set(er[i].tolist())) for i in range(nq))
        evs = float(ev.sum())
        sweep.append({"ef_search": ef, "graph_traversal": {"qps_device_resident": nq / t, "batch_ms": t * 1e3,
                                                           **({"qps_device_resident_4x_queries_per_call": qps_4x, "first_queries_of_the_4x_call_identical": same_4x} if qps_4x else {}), "qps_host_pointers_incl_exact_heap_redo": nq / t_host,
                                                           "evals_per_query": evs / nq, "evals_per_s": evs / t, "gathered_GBps": evs * D * 4 / t / 1e9,
                                                           "flagged_for_exact_heap": flagged, "recall_at_%d_graph_results_only" % k: hit_graph / (nq * k)},
                      "underfilled_queries": int(under.size), "top_up_exact_scan_ms": t_top * 1e3,
                      "search_complete": {"qps": nq / (t_host + t_top), "recall_at_%d_vs_exact" % k: hit_full / (nq * k),
                                          "what": "HNSW.Search as the reference defines it: graph traversal, then the exact top-k for queries the graph under-filled (hnsw.go:676-710)"}})
    out["search"] = sweep
    # the ceiling of `gathered_GBps` on THIS box: the bare random-row stream in the traversal's own shape (32 rows x 128-byte pieces per
    # slab by LDS-DMA, two slab buffers per wave, 12 waves per CU, a table of the corpus's size; no arithmetic, no bookkeeping) —
    # tools/native/qv_ubench.hip (a measurement library beside libqv, not part of it)
    try:
        import ctypes as C
        ub = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "quiver_amd", "lib", "libqvubench.so"))
        ub.qvu_gather_rate.restype = C.c_int
        ub.qvu_gather_rate.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_double)]
        table = torch.empty((N, D), dtype=torch.float32, device="cuda")
        tb = C.c_double(0.0)
        if ub.qvu_gather_rate(table.data_ptr(), N, D, 12, C.byref(tb)) == 0:
            out["bare_row_stream_GBps_this_box"] = tb.value * 1e3
            for e in sweep:
                e["graph_traversal"]["gathered_frac_of_bare_row_stream"] = e["graph_traversal"]["gathered_GBps"] / (tb.value * 1e3)
        del table
    except Exception as ex:                                  # noqa: BLE001  (a ceiling beside the measurement, never a reason to lose it)
        out["bare_row_stream_GBps_this_box"] = "unmeasured: %s" % ex
    if callers:
        # The traffic the reference's host produces: HNSW.Search under a read lock, one query per call, a goroutine per request
        # (hnsw.go:602-606, adapter.go:253-279).  Native threads through qv_graph_search with nq = 1 (tools/native/qv_callers.cpp); every
        # result is compared with the batch call's above.
        from tests import _callers
        ef = 128 if 128 in efs else efs[0]
        r0, d0, c0, _ = host[ef]
        rows_c = []
        for t in callers:
            cr = _callers.run("graph", g.handle, hq[:1024], k, threads=t, seconds=1.5, ef=ef)
            seen = cr["count"] != 0xFFFFFFFD
            same = bool(np.array_equal(cr["count"][seen], c0[:1024][seen]) and np.array_equal(cr["rows"][seen], r0[:1024][seen]) and
                        np.array_equal(cr["dist"][seen].view(np.uint32), d0[:1024][seen].view(np.uint32)))
            rows_c.append({"callers": t, "qps": cr["qps"], "p50_us": cr["p50_us"], "p99_us": cr["p99_us"], "errors": cr["errors"],
                           "mismatches": cr["mismatches"], "same_as_batch_call": same})
        out["concurrent_single_query_callers"] = {"ef_search": ef, "what": "T native threads, qv_graph_search with one query per call, closed loop", "by_callers": rows_c}

    if cpu_queries:
        from tests import _oracle as O                    # checker / CPU baseline only
        t0 = time.perf_counter()
        if intrinsic_dim:
            hrows = host_rows
        else:
            hrows = np.empty((N, D), np.float32)
            for s in range(0, N, 100_000):
                e = min(N, s + 100_000); hrows[s:e] = O.gen_rows(corpus_seed, s, e - s, D)
        t_gen = time.perf_counter() - t0
        lv, l0_deg, l0_links, up_off, up_links = g.export()
        mid = quiver_amd.metric_id(metric)
        cpu = []
        for ef in efs:
            o = O.HNSW(mid, D, M=m, maxM0=2 * m, efConstruction=efc, efSearch=ef, maxLevel=max_level, seed=level_seed)
            o.load_graph(hrows, lv, 2 * m, m, l0_deg, l0_links, up_off, up_links, info["entry"], info["cur_level"])
            r, d, c, ev = host[ef]
            identical = True; evals = 0
            t0 = time.perf_counter()
            res = [o.search(hq[i], k, with_evals=True) for i in range(cpu_queries)]
            t_cpu = time.perf_counter() - t0
            for i, (ro, do, eo) in enumerate(res):
                evals += eo
                if int(c[i]) == k:                                  # filled by the graph: rows, float32 bits and evaluation counts
                    identical &= r[i].tolist() == ro.tolist() and d[i].tobytes() == do.tobytes() and int(ev[i]) == eo - 1
                else:                                               # topped up: the exact top-k under (distance, node) order
                    identical &= er[i].tolist() == ro.tolist() and ed[i].tobytes() == do.tobytes()
            cpu.append({"ef_search": ef, "qps": cpu_queries / t_cpu, "cores": 1, "evals_per_query": evals / cpu_queries,
                        "identical_to_device": bool(identical), "queries": cpu_queries})
        out["cpu_traversal_same_graph"] = {"kind": "port", "what": "the oracle's HNSW.Search (hnsw.go:602-713 restated) walking the exported device-built graph",
                                           "gen_rows_s": t_gen, "by_ef": cpu}
        if cpu_build_rows:
            nb = min(cpu_build_rows, N)
            o = O.HNSW(mid, D, M=m, maxM0=2 * m, efConstruction=efc, maxLevel=max_level, seed=level_seed)
            t0 = time.perf_counter()
            for i in range(nb):
                o.insert(hrows[i])
            t_cb = time.perf_counter() - t0
            out["cpu_build_sequential"] = {"rows": nb, "seconds": t_cb, "nodes_per_s": nb / t_cb, "cores": 1,
                                           "what": "the oracle's Insert loop (hnsw.go:266-468 restated) over the first rows of the same corpus"}
    g.close(); idx.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000); ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--metric", default="cosine"); ap.add_argument("--m", type=int, default=16); ap.add_argument("--efc", type=int, default=200)
    ap.add_argument("--max-level", type=int, default=16); ap.add_argument("--batch-max", type=int, default=16384); ap.add_argument("--ramp-div", type=int, default=16)
    ap.add_argument("--efs", default="64,128,256,512"); ap.add_argument("--nq", type=int, default=8192); ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--cpu-queries", type=int, default=100); ap.add_argument("--cpu-build-rows", type=int, default=0)
    ap.add_argument("--intrinsic-dim", type=int, default=0, help="0 = BASELINE's random unit vectors; r = unit vectors on an r-dimensional subspace")
    a = ap.parse_args()
    print(json.dumps(run(a.rows, a.dim, a.metric, a.m, a.efc, a.max_level, a.batch_max, a.ramp_div, tuple(int(x) for x in a.efs.split(",")),
                         a.nq, a.k, a.cpu_queries, cpu_build_rows=a.cpu_build_rows, intrinsic_dim=a.intrinsic_dim)), flush=True)
