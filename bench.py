#!/usr/bin/env python3
"""bench.py — flat cosine scan QPS on MI355X (BASELINE.json metric), one JSON line.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): exact flat cosine top-10 over a synthetic unit-vector
corpus of --rows x 768 fp32 (default 10M x 768 = BASELINE.json configs[4]'s corpus,
the shape the north star's roofline target is quoted on; it fits one GPU, so the same
corpus is used at every N and the curve is strong scaling).  One STEP = one query
against the whole corpus: every rank scans its contiguous row shard
(qv_index_search_device: HIP flat-scan + fused top-k), the per-shard top-k are
all-gathered over RCCL and merged deterministically (qv_merge_topk_device).  Inputs
(corpus, queries) are resident in HBM before the timed region.  At N=1 the line also
carries configs[1] (1M x 768) measured in the same process ("also").

Extra objects: "roofline" (HBM, dominant kernel = k_flat_scan, from HIP events around
that kernel) and "cpu_baseline" (the CPU oracle in reference-faithful mode on a
bounded sample; the oracle is only the checker/baseline here, never the product).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec; ~6.3 TB/s measured copy)
CORPUS_SEED, QUERY_SEED = 20260424, 20260425


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=10_000_000, help="total corpus rows (sharded over --gpus)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", default="cosine")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the extra 1M x 768 measurement at N=1")
    ap.add_argument("--cpu-sample-rows", type=int, default=100_000)
    ap.add_argument("--cpu-sample-queries", type=int, default=8)
    return ap.parse_args()


def cpu_baseline(dim, k, sample_rows, sample_queries, total_rows):
    """Reference-faithful ExactIndex.Search on the host (oracle 'port'): rows behind a
    string-keyed hash map, scalar float64 distance per row, full sort of all N.
    1 thread (what ExactIndex.Search uses per query, exact.go:92-133)."""
    from tests import _oracle as O
    rows = O.gen_rows(CORPUS_SEED, 0, sample_rows, dim)
    f = O.Faithful(0, dim)
    for i in range(sample_rows):
        f.insert("v%d" % i, rows[i])
    qs = O.gen_rows(QUERY_SEED, 0, sample_queries, dim)
    f.search(qs[0], k)
    t0 = time.perf_counter()
    for q in qs:
        f.search(q, k)
    dt = time.perf_counter() - t0
    rows_per_s = sample_rows * sample_queries / dt
    return {
        "value": rows_per_s / total_rows, "unit": "queries/s",
        "cores": 1, "kind": "port",
        "sample": "%d queries x first %d rows of the same corpus, reference-faithful ExactIndex.Search restatement "
                  "(oracle/qv_oracle.c qvo_faithful_search), %.2f s; value = measured rows/s / %d rows" %
                  (sample_queries, sample_rows, dt, total_rows),
        "rows_per_s": rows_per_s,
    }


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (a.gpus, a.gpus))
        a.gpus = world
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import quiver_amd
    from quiver_amd.device_index import merge_topk_device, device_info

    dim, k, G = a.dim, a.k, world
    # contiguous row shards [g*N/G, (g+1)*N/G)
    base = rank * a.rows // G
    n_local = (rank + 1) * a.rows // G - base
    idx = quiver_amd.DeviceIndex(dim, a.metric, device=local_rank)
    idx.reserve(n_local)
    t_gen = time.perf_counter()
    done = 0
    while done < n_local:                       # generate in slabs (one kernel launch each)
        m = min(n_local - done, 2_000_000)
        idx.add_synthetic(CORPUS_SEED, base + done, m)
        done += m
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen

    from tests import _oracle as O                # queries come from the shared generator (host side, tiny)
    nq_pool = 256
    qs_host = O.gen_rows(QUERY_SEED, 0, nq_pool, dim)
    d_q = torch.from_numpy(qs_host).cuda()
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream

    total_steps = a.warmup + a.steps
    d_rows = torch.empty((total_steps + 1, k), dtype=torch.int32, device="cuda")      # local top-k, global row ids
    d_dist = torch.empty((total_steps + 1, k), dtype=torch.float32, device="cuda")
    g_rows = torch.empty((total_steps + 1, G, k), dtype=torch.int32, device="cuda") if G > 1 else None
    g_dist = torch.empty((total_steps + 1, G, k), dtype=torch.float32, device="cuda") if G > 1 else None
    f_rows = torch.empty((total_steps + 1, k), dtype=torch.int32, device="cuda")
    f_dist = torch.empty((total_steps + 1, k), dtype=torch.float32, device="cuda")
    qsz = dim * 4

    def local_scan(i, index):
        index.search_device(d_q.data_ptr() + (i % nq_pool) * qsz, 1, k, d_rows[i].data_ptr(), d_dist[i].data_ptr(), sp)
        if G > 1:
            d_rows[i].add_(base)                 # shard-local row -> global row

    def finish(i, works):
        """exchange + merge for step i (called one step late so it overlaps scan i+1)"""
        w1, w2 = works
        w1.wait(); w2.wait()
        merge_topk_device(g_dist[i].data_ptr(), g_rows[i].data_ptr(), G, k, f_rows[i].data_ptr(), f_dist[i].data_ptr(), sp)

    def run(first, count, index):
        pending = None
        for i in range(first, first + count):
            local_scan(i, index)
            if G > 1:
                if pending is not None:
                    finish(*pending)
                w1 = dist.all_gather_into_tensor(g_dist[i].view(-1), d_dist[i], async_op=True)
                w2 = dist.all_gather_into_tensor(g_rows[i].view(-1), d_rows[i], async_op=True)
                pending = (i, (w1, w2))
        if pending is not None:
            finish(*pending)

    def barrier():
        if G > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup, then EXACTLY K timed steps bracketed by barrier + synchronize ----
    run(0, a.warmup, idx)
    barrier()
    t0 = time.perf_counter()
    run(a.warmup, a.steps, idx)
    barrier()
    dt = time.perf_counter() - t0
    if G > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    qps = a.steps / dt

    # ---- roofline of the dominant kernel: HIP events around k_flat_scan, separate pass ----
    idx.profile(True)
    pass_steps = min(a.steps, 50)
    for j in range(pass_steps):
        idx.search_device(d_q.data_ptr() + (j % nq_pool) * qsz, 1, k, d_rows[total_steps].data_ptr(), d_dist[total_steps].data_ptr(), sp)
    torch.cuda.synchronize()
    scan_ms, launches = idx.profile_read()
    idx.profile(False)
    alg_bytes = n_local * dim * 4 + n_local * 8          # rows once + cached f64 row norms (cosine); SURVEY.md 8d
    kern_ms = scan_ms / max(launches, 1)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if launches else 0.0

    # ---- verification of what was timed (checker only) ----
    res_rows = (f_rows if G > 1 else d_rows)[a.warmup: a.warmup + min(a.steps, 8)].cpu().numpy().view(np.uint32)
    res_dist = (f_dist if G > 1 else d_dist)[a.warmup: a.warmup + min(a.steps, 8)].cpu().numpy()
    verified = True
    if rank == 0:
        for j in range(res_rows.shape[0]):
            q = qs_host[(a.warmup + j) % nq_pool]
            for t in range(k):
                row = int(res_rows[j, t])
                want = O.distance(0, q, O.gen_rows(CORPUS_SEED, row, 1, dim)[0]) if a.metric == "cosine" else None
                if want is not None and np.float32(want).view(np.uint32) != res_dist[j, t].view(np.uint32):
                    verified = False
            if not all(res_dist[j, t] <= res_dist[j, t + 1] for t in range(k - 1)):
                verified = False

    also = None
    if G == 1 and not a.no_also and a.rows != 1_000_000 and a.metric == "cosine":
        # configs[1]: flat cosine 1M x 768, single query, same process
        idx1 = quiver_amd.DeviceIndex(dim, a.metric, device=local_rank)
        idx1.reserve(1_000_000)
        idx1.add_synthetic(CORPUS_SEED, 0, 1_000_000)
        r1 = torch.empty((k,), dtype=torch.int32, device="cuda"); d1 = torch.empty((k,), dtype=torch.float32, device="cuda")
        for j in range(20):
            idx1.search_device(d_q.data_ptr() + (j % nq_pool) * qsz, 1, k, r1.data_ptr(), d1.data_ptr(), sp)
        torch.cuda.synchronize()
        steps1 = 500
        t1 = time.perf_counter()
        for j in range(steps1):
            idx1.search_device(d_q.data_ptr() + (j % nq_pool) * qsz, 1, k, r1.data_ptr(), d1.data_ptr(), sp)
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        idx1.profile(True)
        for j in range(100):
            idx1.search_device(d_q.data_ptr() + (j % nq_pool) * qsz, 1, k, r1.data_ptr(), d1.data_ptr(), sp)
        torch.cuda.synchronize()
        ms1, n1 = idx1.profile_read()
        b1 = 1_000_000 * dim * 4 + 1_000_000 * 8
        also = {"workload": "flat cosine 1Mx768 fp32, k=10, single query (BASELINE configs[1])",
                "qps": steps1 / dt1, "ms_per_query": dt1 / steps1 * 1e3,
                "scan_kernel_ms": ms1 / max(n1, 1), "hbm_gbs": b1 / (ms1 / max(n1, 1) * 1e-3) / 1e9,
                "hbm_frac": b1 / (ms1 / max(n1, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS}
        idx1.close()

    cpu = None
    if rank == 0 and not a.no_cpu_baseline:
        cpu = cpu_baseline(dim, k, a.cpu_sample_rows, a.cpu_sample_queries, a.rows)

    if rank == 0:
        info = device_info(local_rank)
        out = {
            "metric": "flat_cosine_qps_recall_1.0", "value": qps, "unit": "queries/s",
            "n_gpus": G, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64-accumulate over fp32 rows (reference arithmetic)", "data": "synthetic",
            "config": {"workload": "flat %s scan %dx%d fp32, k=%d, single query per step, recall 1.0 (exact)" % (a.metric, a.rows, dim, k),
                       "rows_total": a.rows, "rows_per_gpu": n_local, "dim": dim, "k": k,
                       "sharding": "contiguous row shards, per-shard top-k + RCCL all-gather + deterministic merge" if G > 1 else "single shard",
                       "device": info["name"], "cus": info["cus"], "corpus_gen_s": round(t_gen, 3)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_flat_scan", "kernel_ms": kern_ms, "launches_timed": launches,
                         "algorithmic_bytes_per_launch": alg_bytes},
            "cpu_baseline": cpu,
            "verified_against_oracle": bool(verified),
        }
        if also:
            out["also"] = also
        print(json.dumps(out), flush=True)
    if G > 1:
        dist.barrier()
        dist.destroy_process_group()
    idx.close()


if __name__ == "__main__":
    main()
