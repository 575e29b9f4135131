#!/usr/bin/env python3
"""bench.py — flat cosine scan QPS on MI355X (BASELINE.json metric), one JSON line.

  python bench.py --gpus N --steps K --warmup W            # N > 1 launches its own ranks (see below)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): exact flat cosine top-10 over a synthetic unit-vector corpus of --rows x 768 fp32 (default
10M x 768 = BASELINE.json configs[4]'s corpus, the shape the north star's roofline target is quoted on; it fits one GPU,
so the same corpus is used at every N and the curve is strong scaling).  One STEP = one query against the whole corpus:
every rank scans its contiguous row shard (qv_index_search_device: HIP flat scan + fused top-k), the per-shard top-k are
all-gathered over RCCL and merged deterministically (qv_merge_topk_shards_device) — the orchestration is
quiver_amd.sharded.ShardedFlatSearch, the same code the gloo tests cover; the exchange of step i overlaps the scan of
step i+1.  Corpus and queries are resident in HBM before the timed region.

Launching.  Under torch.distributed.run (RANK / WORLD_SIZE set) this file is one rank.  Run plainly with --gpus N > 1 it
starts N fresh rank processes itself (python -m torch.distributed.run ... bench.py, before this process has touched the
GPU) and passes their output and exit code through.  With fewer GPUs than ranks (a dry run of the N > 1 control path on a
1-GPU box) ranks share devices and the exchange runs over gloo, because RCCL needs one device per rank; the line then says
"exchange": "gloo (ranks share a device)".  --abi-sharded measures the same workload through ONE process and the C ABI's
qv_sharded_* handle (a shard per device, ncclAllGather inside libqv) instead of one process per GPU.

Extra objects on the line:
  "roofline"     HBM roofline of the dominant kernel k_flat_scan (HIP events around that kernel, separate pass; traffic
                 from the committed rocprofv3 PMC summary); at N > 1 also every GPU's fraction and the measured
                 all-gather + merge latency
  "cpu_baseline" the CPU oracle in reference-faithful mode on a bounded sample, 1 thread (what ExactIndex.Search uses) —
                 plus, under "others": the same with one query per core (what BatchSearch does), and an optimised
                 AVX-512 scan labelled NOT the reference; host model and core count (checker/baseline only — never the
                 product path)
  "also"         (N=1) the other BASELINE configs measured in the same process: configs[0] 10k x 128; configs[1] flat
                 cosine 1M x 768 single query; configs[2] 256 queries x 1M x 768 (exact f64-matrix scan, fp32-MFMA filter for
                 cosine and for dot-product, with MFMA rooflines); configs[3] HNSW M=16 efC=200 at 1M x 768: the graph
                 built on the device, efSearch sweep, recall@10, CPU traversal of the identical graph; the PCIe-inclusive
                 single-query rate of the host-pointer entry point.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

# more hardware queues than the runtime's default four (streams that share one run their kernels one after another): libqv asks for the
# same when it is loaded, but torch initialises HIP first in this process, and the runtime reads the variable once, at that moment
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec; ~6.3 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3    # v_mfma_f32_32x32x2_f32 dense peak (same guide)
MFMA_BF16_PEAK_TF = 2500.0  # v_mfma_f32_32x32x16_bf16 dense (same guide: ~2.5 PF, 16x the fp32 MFMA rate)
MFMA_F64_PEAK_TF = 77.5     # v_mfma_f64_16x16x4_f64 as measured with register-resident operands (tools/ubench/mfma_f64_rate.hip; the guide quotes 78.6)
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
    ap.add_argument("--no-also", action="store_true", help="skip the extra configs measured at N=1")
    ap.add_argument("--no-hnsw", action="store_true", help="skip the configs[3] HNSW build + search entries of `also`")
    ap.add_argument("--hnsw-rows", type=int, default=1_000_000)
    ap.add_argument("--force-exchange", action="store_true",
                    help="with one rank, still run the all-gather + merge per step (exercises the N>1 code path on a 1-GPU box)")
    ap.add_argument("--scan-streams", type=int, default=0, help="scan streams alternated between consecutive queries; 0 = auto")
    ap.add_argument("--cpu-sample-rows", type=int, default=500_000)
    ap.add_argument("--cpu-sample-queries", type=int, default=30)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the multi-core CPU baselines; 0 = all cores")
    ap.add_argument("--abi-sharded", action="store_true",
                    help="one process, qv_sharded_* over --gpus devices (RCCL all-gather inside libqv) instead of one process per GPU")
    ap.add_argument("--peer-copy", action="store_true", help="with --abi-sharded: point-to-point exchange (allows shards to share a device)")
    ap.add_argument("--preflight", action="store_true",
                    help="check the multi-GPU plumbing and exit (no corpus): device count, peer-access matrix, the HIP/RCCL pair libqv bound, "
                         "communicators created, one tiny all-gather, communicators destroyed; non-zero exit with the reason on failure")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------- launching
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_REAL_STDOUT = None


def quiet_stdout():
    """The one JSON line is all this program may write to stdout, but libraries write there too (RCCL prints a version banner
    when a communicator is created): point fd 1 at stderr for the run and keep the real stdout for emit()."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


MAX_LINE = 4000            # the driver keeps the last 8 KB of stdout: the contract line must fit with room to spare


def _r(x, nd=4):
    """floats rounded to `nd` significant digits (the sidecar keeps full precision)"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (nd, x))


def _pick(d, keys, nd=4):
    return {k: _r(d[k], nd) for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_also(also, budget):
    """One or two numbers per extra config (ms or QPS + fraction of its roofline, `same` = identical to the exact scan / oracle);
    everything else lives in the sidecar.  Entries are shed from the least important end until the text fits `budget` bytes."""
    c = {}
    for key, e in also.items():
        if not isinstance(e, dict):
            continue
        short = key.replace("batched_256x", "b256x").replace("_single_query", "").replace("hnsw_1Mx768_", "hnsw_1M_")
        if "error" in e:
            c[short] = {"error": str(e["error"])[:60]}
        elif key.startswith("hnsw_"):
            o = {}
            b = e.get("build") or {}
            if "seconds" in b:
                o["build_s"] = _r(b["seconds"], 3)
            srch = e.get("search") or []
            pick = next((x for x in srch if x.get("ef_search") == 128), srch[-1] if srch else None)
            if pick:
                g, sc = pick.get("graph_traversal", {}), pick.get("search_complete", {})
                o.update({"corpus": "16-d subspace of R^768" if key.endswith("_structured") else "iid random (BASELINE)",
                          "ef": pick.get("ef_search"), "qps": _r(g.get("qps_device_resident"), 4),
                          **({"qps_4x_per_call": _r(g["qps_device_resident_4x_queries_per_call"], 4)} if g.get("qps_device_resident_4x_queries_per_call") else {}),
                          # recall of the graph results alone, and of HNSW.Search as the reference defines it (with its exact top-up of under-filled queries)
                          "recall10_vs_exact": _r(g.get("recall_at_10_graph_results_only"), 4), "recall10_with_topup": _r(sc.get("recall_at_10_vs_exact"), 4),
                          "gather_frac": _r(g.get("gathered_GBps", 0.0) / HBM_PEAK_GBS, 3),
                          "of_bare_row_stream": _r(g.get("gathered_frac_of_bare_row_stream"), 3)})
            cpu = (e.get("cpu_traversal_same_graph") or {}).get("by_ef") or []
            cpick = next((x for x in cpu if pick and x.get("ef_search") == pick.get("ef_search")), None)
            if cpick:
                o.update({"cpu_qps": _r(cpick.get("qps"), 4), "same": cpick.get("identical_to_device")})
            cc = (e.get("concurrent_single_query_callers") or {}).get("by_callers")
            if cc:
                o["callers_qps"] = {str(x["callers"]): _r(x["qps"], 4) for x in cc}
            c[short] = o
        elif key == "batched_small_1Mx768":
            c["b_small_1Mx768_ms"] = {k2.replace("_queries_", "q_").replace("float32", "f32"): _r(v2.get("batch_ms"), 3) for k2, v2 in e.items() if isinstance(v2, dict)}
        elif key == "k_above_64":
            c["k_gt_64_ms"] = {k2[:-3]: _r(v2, 4) for k2, v2 in e.items() if k2.endswith("_ms")}
            c["k_gt_64_ms"]["same"] = all(v2 for k2, v2 in e.items() if k2.endswith("_same"))
        elif key == "short_collections_single_query":
            c["short_1q_p50_us"] = {k2: _r(v2.get("p50_us"), 3) for k2, v2 in e.items() if isinstance(v2, dict)}
        elif key == "pcie_inclusive_single_query":
            c["pcie_inclusive_qps"] = _r(e.get("qps"), 4)
        elif key == "concurrent_single_query_callers":
            c["callers_1M_qps"] = {str(x["callers"]): _r(x["qps"], 4) for x in e.get("by_callers", [])}
            c["callers_1M_qps"]["same"] = all(x.get("same_as_batch_call") and not x.get("errors") and not x.get("mismatches") for x in e.get("by_callers", []))
        else:
            o = {}
            for src, dst in (("batch_ms", "ms"), ("qps", "qps"), ("hbm_frac", "frac"), ("frac_of_f64_matrix_peak", "frac"), ("latency_us", "us"),
                             ("cpu_port_latency_us_1core", "cpu_us"), ("identical_to_exact_scan", "same"), ("identical_to_oracle", "same")):
                if e.get(src) is not None and not (src == "qps" and "batch_ms" in e):
                    o[dst] = _r(e[src], 4)
            if isinstance(e.get("plain_c_caller"), dict) and "p50_us" in e["plain_c_caller"]:
                o["c_us"] = _r(e["plain_c_caller"]["p50_us"], 4)
            rf = e.get("roofline")
            if isinstance(rf, dict):
                o.update({"bound": rf.get("bound"), "kernel_ms": _r(rf.get("kernel_ms"), 4), "frac": _r(rf.get("frac"), 3)})
                if "end_to_end_frac" in rf:
                    o["e2e"] = _r(rf["end_to_end_frac"], 3)
                if "frac_of_bare_mfma_loop" in rf and rf.get("bare_mfma_loop_clock_ghz"):
                    o.update({"of_bare_loop": _r(rf["frac_of_bare_mfma_loop"], 3), "clock_ghz": _r(rf["bare_mfma_loop_clock_ghz"], 3), "at_clock": _r(rf.get("frac_at_the_held_clock"), 3)})
            c[short] = o
    shed = [k for k in c if "10M" in k] + [k for k in c if k.endswith("_dot")] + ["b_small_1Mx768_ms"] + [k for k in c if "bf16x3" in k] + list(c)
    dropped = 0
    while len(json.dumps(c, separators=(",", ":"))) > budget and shed:
        if c.pop(shed.pop(0), None) is not None:
            dropped += 1
    if dropped:
        c["in_sidecar_only"] = dropped
    return c


def compact_line(out, sidecar=None):
    """The contract line: headline + roofline + cpu_baseline (two numbers per extra CPU leg) + a compact `also`.  <= MAX_LINE bytes."""
    line = {k: (_r(v, 6) if isinstance(v, float) else v) for k, v in out.items() if k not in ("also", "cpu_baseline", "roofline", "config")}
    cfg = dict(out.get("config") or {})
    for k in list(cfg):
        if isinstance(cfg[k], str) and len(cfg[k]) > 160:
            cfg[k] = cfg[k][:157] + "..."
    line["config"] = cfg
    rf = out.get("roofline")
    if isinstance(rf, dict):
        line["roofline"] = {k: (_r(v, 6) if isinstance(v, float) else v) for k, v in rf.items() if v is not None or k == "traffic"}
        pg = rf.get("per_gpu")
        if pg:
            line["roofline"]["per_gpu"] = [{"rank": g["rank"], "ms": _r(g["scan_kernel_ms"], 4), "frac": _r(g["hbm_frac"], 3)} for g in pg]
    else:
        line["roofline"] = rf
    cpu = out.get("cpu_baseline")
    if isinstance(cpu, dict):
        cc = {"value": _r(cpu["value"], 5), "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"], "sample": cpu["sample"][:200]}
        cc["others"] = {k: {"value": _r(v["value"], 5), "cores": v["cores"]} for k, v in (cpu.get("others") or {}).items()}
        h = cpu.get("host") or {}
        cc["host"] = _pick(h, ("cpu_model", "nproc", "usable_cores"))
        line["cpu_baseline"] = cc
    else:
        line["cpu_baseline"] = cpu
    if sidecar:
        line["full_record"] = sidecar

    def dumps():
        return json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(dumps()) > MAX_LINE - 200:                      # shrink the descriptive strings before anything measured
        for k in ("config", "cpu_baseline"):
            if isinstance(line.get(k), dict):
                line[k] = {kk: (vv[:80] if isinstance(vv, str) else vv) for kk, vv in line[k].items()}
    if out.get("also"):                                    # the extras get what is left; the sidecar has all of them
        line["also"] = compact_also(out["also"], MAX_LINE - len(dumps()) - 64)      # (the key, its braces and the shed counter come on top)
    text = dumps()
    # never lose the headline to the budget: shed the extras, then the per-GPU list, then the descriptive strings
    for drop in ("also", ("roofline", "per_gpu"), ("cpu_baseline", "others"), ("cpu_baseline", "sample"), ("cpu_baseline", "host"), "full_record"):
        if len(text) <= MAX_LINE:
            break
        if isinstance(drop, tuple):
            if isinstance(line.get(drop[0]), dict):
                line[drop[0]].pop(drop[1], None)
        else:
            line.pop(drop, None)
        text = dumps()
    if len(text) > MAX_LINE and isinstance(line.get("config"), dict):
        line["config"] = {kk: (vv[:24] if isinstance(vv, str) else vv) for kk, vv in line["config"].items()}
        text = dumps()
    return text


def write_sidecar(out):
    """The full record (every leg of `also`, samples, notes) next to the contract line: gpurun_out/bench_full.json (merged back
    from the GPU box), or $QV_BENCH_SIDECAR; on failure the record goes to stderr only."""
    path = os.environ.get("QV_BENCH_SIDECAR") or os.path.join(ROOT, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError:
        sys.stderr.write(json.dumps(out) + "\n")
        return None


def emit(out):
    sidecar = write_sidecar(out)
    line = (compact_line(out, sidecar) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line)


def self_launch(a):
    """--gpus N > 1 outside torch.distributed.run: start the N ranks as fresh child processes and pass their output through.
    Nothing in this process has initialised the GPU (torch.cuda.device_count() does not), and it only waits."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.exit(subprocess.run(cmd, env=env).returncode)


# ---------------------------------------------------------------------------------------------- CPU baselines
def host_info():
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:                                                   # cgroup v2 CPU quota of this container ("max" = none)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                                               # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    cores = max(1, int(min(usable, quota) if quota else usable))
    return {"nproc": os.cpu_count(), "cpu_model": model, "schedulable_cpus": usable, "cgroup_cpu_quota": quota, "usable_cores": cores}


def cpu_baseline(dim, k, sample_rows, sample_queries, total_rows, threads):
    """Reference-faithful ExactIndex.Search on the host (oracle 'port'): rows behind a string-keyed hash map, scalar float64
    distance per row, full sort of all N.  `value`: 1 thread (what ExactIndex.Search uses per query, exact.go:92-133).
    others: one query per core (BatchSearch, hybrid_index.go:703-705) and an optimised AVX-512 scan (NOT the reference)."""
    from tests import _oracle as O
    host = host_info()
    T = threads or host["usable_cores"]                    # what this process may actually run on (affinity mask, cgroup quota)
    rows = O.gen_rows(CORPUS_SEED, 0, sample_rows, dim)
    f = O.Faithful(0, dim)
    for i in range(sample_rows):
        f.insert("v%d" % i, rows[i])
    qs = O.gen_rows(QUERY_SEED, 0, max(sample_queries, 2 * T), dim)
    f.search(qs[0], k)
    t0 = time.perf_counter()
    for q in qs[:sample_queries]:
        f.search(q, k)
    dt = time.perf_counter() - t0
    rows_per_s = sample_rows * sample_queries / dt
    # the same searches, one per core at a time
    nq_t = max(sample_queries, 2 * T)
    dt_t, _ = f.search_many(qs[:nq_t], k, T)
    rps_t = sample_rows * nq_t / dt_t
    # NOT the reference: contiguous rows, cached norms, AVX-512 float32 lanes, rows split over the cores, partial top-k
    # the container may be allowed fewer cores than it can see (no readable quota): the measured speed-up of the run above says
    # how many really ran; spinning barriers with more threads than cores would measure the scheduler, not the scan
    eff = max(1.0, rps_t / rows_per_s)
    T_o = int(max(1, min(T if T < 32 else T // 2, round(eff))))
    opt = O.OptScan(rows, T_o)
    opt.search(qs[:2], k)
    dt_o, ro, _ = opt.search(qs[:sample_queries], k)
    rps_o = sample_rows * sample_queries / dt_o
    er, _ = O.exact_search(0, rows, qs[0], k)
    return {
        "value": rows_per_s / total_rows, "unit": "queries/s", "cores": 1, "kind": "port",
        "sample": "%d queries x first %d rows of the same corpus, reference-faithful ExactIndex.Search restatement "
                  "(oracle/qv_oracle.c qvo_faithful_search), %.2f s; value = measured rows/s / %d rows (measured at the full 10M rows once: "
                  "7.94 s per query = 0.1259 QPS, 1.7 %% below this scaling, profiles/r05_cpu_baseline_full.json)" %
                  (sample_queries, sample_rows, dt, total_rows),
        "rows_per_s": rows_per_s,
        "others": {
            "faithful_one_query_per_core": {
                "value": rps_t / total_rows, "unit": "queries/s", "cores": T, "kind": "port", "rows_per_s": rps_t,
                "speedup_over_one_thread": rps_t / rows_per_s,
                "sample": "%d queries over %d threads, each thread a whole faithful search at a time (BatchSearch's goroutine per "
                          "query, hybrid_index.go:703-705), same %d rows, %.2f s" % (nq_t, T, sample_rows, dt_t)},
            "optimised_scan_not_the_reference": {
                "value": rps_o / total_rows, "unit": "queries/s", "cores": T_o, "kind": "port", "rows_per_s": rps_o,
                "host_GBps": rps_o * dim * 4 / 1e9, "simd_bits": int(O.lib().qvo_opt_simd_bits()),
                "top%d_overlap_with_exact_first_query" % k: len(set(er.tolist()) & set(ro[0].tolist())) / k,
                "sample": "%d queries, rows split over %d pinned threads (NUMA-local slices), contiguous rows + cached norms + float32 FMA "
                          "lanes + partial top-k (oracle/qv_cpu_baselines.c qvo_opt_cosine_scan): what a tuned CPU scan does, not what the "
                          "reference does; %.3f s" % (sample_queries, T_o, dt_o)}},
        "host": host,
    }


def membership_check(O, mid, query, rows_chk, first_gen_row, rr_gen, dd):
    """The CPU oracle as a checker of MEMBERSHIP (recomputing the returned rows' distances cannot show a skipped row): with the
    oracle's distances of the corpus rows first_gen_row .. first_gen_row + len(rows_chk), every one of those rows is either among
    the returned results (rr_gen: their corpus row numbers, dd: their distances, ascending) or its (distance, row) key is not
    below the returned k-th key; returned rows inside the range carry the oracle's bits."""
    n = rows_chk.shape[0]
    od = O.all_distances(mid, rows_chk, query)
    gen = np.arange(n, dtype=np.int64) + first_gen_row
    better = (od < dd[-1]) | ((od == dd[-1]) & (gen < int(rr_gen[-1])))
    local = (rr_gen >= first_gen_row) & (rr_gen < first_gen_row + n)
    inside = np.zeros(n, bool)
    inside[(rr_gen[local] - first_gen_row).astype(np.int64)] = True
    ok = not np.any(better & ~inside)
    ok &= np.array_equal(od[(rr_gen[local] - first_gen_row).astype(np.int64)].view(np.uint32), np.ascontiguousarray(dd[local]).view(np.uint32))
    return bool(ok)


def short_runtime(text):
    """qv_runtime_info, shortened for the line: versions + the directory each library came from"""
    import re
    m = re.match(r"hip_runtime=(\S+) lib=(\S+); rccl=(\S+) lib=(\S+)", text)
    if not m:
        return text[:150]
    return "hip %s (%s), rccl %s (%s)" % (m.group(1), os.path.dirname(m.group(2)), m.group(3), os.path.dirname(m.group(4)))


def pmc_traffic(rows_per_gpu, dim):
    """HBM bytes per k_flat_scan launch from the newest committed rocprofv3 PMC summary (profiles/rNN_10Mx768_pmc.json), when it was
    taken on this exact per-GPU workload AND on the kernel sources as they are now (the file carries their SHA-256,
    tools/make_pmc_json.py): counters of an older kernel are not this kernel's traffic — then None, with the reason."""
    import glob
    if not (rows_per_gpu == 10_000_000 and dim == 768):
        return None, None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from make_pmc_json import source_hash
        now = source_hash()
    except Exception as ex:  # noqa: BLE001
        return None, "kernel sources unreadable: %s" % ex
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_10Mx768_pmc.json")), reverse=True)
    for f in files[:1]:
        try:
            d = json.load(open(f))
        except Exception:  # noqa: BLE001
            continue
        name = "profiles/" + os.path.basename(f)
        if d.get("kernel_sources_sha256") != now:
            return None, name + " is STALE: taken on other kernel sources (re-run tools/run_round_pmc.sh)"
        return d["hbm_bytes_per_launch"], name
    return None, None


# ---------------------------------------------------------------------------------------------- `also` (N = 1)
def also_entries(a, torch, quiver_amd, idx, d_q, qs_host, local_rank):
    """BASELINE.json's other configurations beside the headline (tests/bench/bench_also.py)"""
    from tests.bench.bench_also import also_entries as run
    return run(a, torch, quiver_amd, idx, d_q, qs_host, local_rank)


# ---------------------------------------------------------------------------------------------- preflight
def _say(msg):
    sys.stderr.write("[preflight] %s\n" % msg)
    sys.stderr.flush()


def preflight_devices(torch, need, rank=0):
    """device count and the peer-access matrix (rank 0 prints); raises SystemExit with the reason when `need` devices are not there"""
    ndev = torch.cuda.device_count()
    if rank == 0:
        _say("visible devices: %d (need %d); HSA_ENABLE_IPC_MODE_LEGACY=%s" % (ndev, need, os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")))
        if ndev > 1:
            rows = []
            for i in range(ndev):
                rows.append(" ".join("-" if i == j else ("1" if torch.cuda.can_device_access_peer(i, j) else "0") for j in range(ndev)))
            _say("peer access (row = from, column = to):\n    " + "\n    ".join(rows))
    return ndev


def preflight_one_runtime():
    """libqv's RCCL / HIP runtime and torch's are the SAME files: every librccl / libamdhip64 mapped into this process, by real path.
    Two copies of either (torch's bundled pair beside /opt/rocm's) is the mix that hung a collective in round 2 and ran unnoticed
    through a green single-GPU record in round 5: fail fast, with both paths.  Returns the one pair."""
    import re
    libs = {}
    with open("/proc/self/maps") as f:
        for line in f:
            m = re.search(r"(/\S*/(librccl|libamdhip64)\.so[^\s]*)", line)
            if m:
                libs.setdefault(m.group(2), set()).add(os.path.realpath(m.group(1)))
    twice = {k: sorted(v) for k, v in libs.items() if len(v) > 1}
    if twice:
        raise SystemExit("[preflight] FAILED: two copies of a runtime library in one process: %s (import torch before quiver_amd, or point "
                         "LD_LIBRARY_PATH at one installation)" % twice)
    return {k: next(iter(v)) for k, v in libs.items()}


def preflight_abi(a):
    """one process, qv_sharded_*: the handle over --gpus devices created (ncclCommInitAll, or peer access with --peer-copy), one
    search on a 64-row-per-shard corpus checked against one index, destroyed"""
    import torch
    import quiver_amd
    from quiver_amd.device_index import runtime_info
    t0 = time.perf_counter()
    ndev = preflight_devices(torch, a.gpus)
    _say("libqv bound: " + runtime_info())
    _say("one runtime pair in the process: %s" % preflight_one_runtime())
    if ndev < a.gpus and not a.peer_copy:
        raise SystemExit("[preflight] FAILED: --abi-sharded --gpus %d needs %d devices, %d visible (--peer-copy co-locates shards for a dry run)" % (a.gpus, a.gpus, ndev))
    devices = [g % max(ndev, 1) for g in range(a.gpus)]
    try:
        sh = quiver_amd.ShardedIndex(a.dim, a.metric, devices=devices, peer_copy=a.peer_copy)
        _say("qv_sharded_create over devices %s: ok (%s exchange), %.1f s" % (devices, "point-to-point" if a.peer_copy else "RCCL all-gather", time.perf_counter() - t0))
        sh.add_synthetic(CORPUS_SEED, 0, 64 * a.gpus)
        one = quiver_amd.DeviceIndex(a.dim, a.metric, device=devices[0])
        one.add_synthetic(CORPUS_SEED, 0, 64 * a.gpus)
        q = one.get_row(3)
        r, d, _ = sh.search(q, a.k)
        r1, d1, _ = one.search(q, a.k)
        span = quiver_amd.lib().qv_sharded_span(a.gpus)
        back = np.array([(int(x) // span) * 64 + int(x) % span for x in r[0]], dtype=np.uint32)
        if not (np.array_equal(back, r1[0]) and np.array_equal(d.view(np.uint32), d1.view(np.uint32))):
            raise SystemExit("[preflight] FAILED: the %d-shard handle and one index disagree on a 64-rows-per-shard corpus" % a.gpus)
        sh.close(); one.close()
    except quiver_amd.QvError as ex:
        raise SystemExit("[preflight] FAILED: %s" % ex)
    _say("one search through %d shards equals one index; handle destroyed; %.1f s in all" % (a.gpus, time.perf_counter() - t0))


def preflight_ranks(torch, dist, rank, world, device_id, backend, full):
    """one rank of `world`: the process group is up (the caller made it); one tiny all-gather with the rank numbers, checked.
    full (--preflight): also print libqv's runtime pair; the caller destroys the group and exits"""
    t0 = time.perf_counter()
    dev = "cuda" if backend == "nccl" else "cpu"          # (gloo: host tensors, as quiver_amd.sharded does when ranks share a device)
    x = torch.full((4,), float(rank), device=dev)
    out = torch.empty((world * 4,), device=dev)
    dist.all_gather_into_tensor(out, x)
    if dev == "cuda":
        torch.cuda.synchronize()
    want = torch.arange(world, device=dev, dtype=torch.float32).repeat_interleave(4)
    if not torch.equal(out, want):
        raise SystemExit("[preflight] FAILED on rank %d: all-gather over %s returned %s" % (rank, backend, out.tolist()))
    if rank == 0:
        _say("process group (%s, %d ranks) up, one all-gather correct, %.2f s" % (backend, world, time.perf_counter() - t0))
        if full:
            from quiver_amd.device_index import runtime_info
            _say("libqv bound: " + runtime_info())
            _say("one runtime pair in the process: %s" % preflight_one_runtime())


# ---------------------------------------------------------------------------------------------- one process, C-ABI sharding
def run_abi_sharded(a):
    # torch first, here as in the one-process-per-GPU mode: librccl / libamdhip64 resolve by soname to whatever the process loaded first, so
    # both launch modes of this file run on the SAME pair (PyTorch's bundled one) and their N > 1 lines compare like with like
    # (config.runtime names it; qv_sharded_create refuses a mixed pair)
    import torch
    import quiver_amd
    from quiver_amd.device_index import device_info, runtime_info
    ndev = torch.cuda.device_count()
    G, dim, k = a.gpus, a.dim, a.k
    if ndev < G and not a.peer_copy:
        raise SystemExit("--abi-sharded --gpus %d needs %d devices (have %d); --peer-copy co-locates shards for a dry run" % (G, G, ndev))
    devices = [g % max(ndev, 1) for g in range(G)]
    torch.cuda.set_device(devices[0])
    sh = quiver_amd.ShardedIndex(dim, a.metric, devices=devices, peer_copy=a.peer_copy)
    sh.reserve(a.rows)
    t_gen = time.perf_counter()
    sh.add_synthetic(CORPUS_SEED, 0, a.rows)
    sh.sync()
    t_gen = time.perf_counter() - t_gen
    nq_pool = 256
    qgen = quiver_amd.DeviceIndex(dim, a.metric, device=devices[0])
    qgen.add_synthetic(QUERY_SEED, 0, nq_pool)
    qs_host = np.stack([qgen.get_row(i) for i in range(nq_pool)])
    qgen.close()
    d_q = torch.from_numpy(qs_host).cuda()
    total = a.warmup + a.steps
    d_r = torch.empty((total, k), dtype=torch.int32, device="cuda")
    d_d = torch.empty((total, k), dtype=torch.float32, device="cuda")
    qsz = dim * 4

    def run(first, count):
        for i in range(first, first + count):
            sh.search_device(d_q.data_ptr() + (i % nq_pool) * qsz, 1, k, d_r.data_ptr() + i * k * 4, d_d.data_ptr() + i * k * 4)

    run(0, a.warmup)
    sh.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.warmup, a.steps)
    sh.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # verification: returned distances are the oracle's for the returned rows
    from tests import _oracle as O
    span = quiver_amd.lib().qv_sharded_span(G)
    bounds = [g * a.rows // G for g in range(G + 1)]
    rr, dd = d_r.cpu().numpy().view(np.uint32), d_d.cpu().numpy()
    verified = True
    mid = quiver_amd.metric_id(a.metric)
    for j in range(min(a.steps, 4)):
        step = a.warmup + j
        for t_ in range(k):
            g = int(rr[step, t_]) // span
            gen_row = bounds[g] + int(rr[step, t_]) % span
            want = O.distance(mid, qs_host[step % nq_pool], O.gen_rows(CORPUS_SEED, gen_row, 1, dim)[0])
            verified &= bool(np.float32(want).view(np.uint32) == dd[step, t_].view(np.uint32))
        verified &= all(dd[step, t_] <= dd[step, t_ + 1] for t_ in range(k - 1))
    n_chk = min(bounds[1], a.cpu_sample_rows)                          # membership over the first rows of shard 0 (global id == corpus row there)
    chk_rows = O.gen_rows(CORPUS_SEED, 0, n_chk, dim)
    for j in range(min(a.steps, 2)):
        step = a.warmup + j
        gen_rows_ = np.array([bounds[int(r) // span] + int(r) % span for r in rr[step]], dtype=np.int64)
        verified &= membership_check(O, mid, qs_host[step % nq_pool], chk_rows, 0, gen_rows_, dd[step])
    sh.profile(True)
    for j in range(min(a.steps, 50)):
        sh.search_device(d_q.data_ptr() + (j % nq_pool) * qsz, 1, k, d_r.data_ptr(), d_d.data_ptr())
    prof = sh.profile_read()
    # every shard's scan KERNEL (HIP events on the shard's own stream around k_flat_scan): the per-GPU roofline numerators
    per_gpu = []
    for g in range(G):
        ms_g, n_g = sh.profile_read_shard(g)
        rows_g = sh.shard_info(g)["rows"]
        kms = ms_g / max(n_g, 1)
        bytes_g = rows_g * dim * 4 + rows_g * 8
        per_gpu.append({"rank": g, "rows": rows_g, "scan_kernel_ms": kms, "hbm_frac": (bytes_g / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS) if kms > 0 else 0.0})
    sh.profile(False)
    n_local = sh.shard_info(0)["rows"]
    alg_bytes = n_local * dim * 4 + n_local * 8
    kern_ms0 = per_gpu[0]["scan_kernel_ms"]
    achieved = alg_bytes / (kern_ms0 * 1e-3) / 1e9 if kern_ms0 > 0 else 0.0
    info = device_info(devices[0])
    out = {
        "metric": "flat_cosine_qps_recall_1.0", "value": a.steps / dt, "unit": "queries/s", "n_gpus": G, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "flat %s scan %dx%d fp32, k=%d, single query per step, recall 1.0 (exact)" % (a.metric, a.rows, dim, k),
                   "rows_total": a.rows, "rows_per_gpu": [sh.shard_info(g)["rows"] for g in range(G)], "dim": dim, "k": k,
                   "sharding": "C ABI qv_sharded_*: ONE process, a shard per device, %s of the per-shard top-k inside libqv, merge on the first device"
                               % ("hipMemcpyPeerAsync (point-to-point)" if a.peer_copy else "ncclAllGather (RCCL)"),
                   "devices": devices, "device": info["name"], "cus": info["cus"], "corpus_gen_s": round(t_gen, 3), "runtime": short_runtime(runtime_info()),
                   # how many RCCL ranks took part in the exchange that was timed (0: point-to-point copies / one shard has nobody to talk to)
                   "rccl_ranks_seen": 0 if a.peer_copy else G},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "k_flat_scan", "kernel_ms": kern_ms0, "algorithmic_bytes_per_launch": alg_bytes, "per_gpu": per_gpu,
                     "scan_phase_ms": prof["scan_ms"], "allgather_plus_merge_us": (prof["exchange_ms"] + prof["merge_ms"]) * 1e3,
                     "exchange_us": prof["exchange_ms"] * 1e3, "merge_and_download_us": prof["merge_ms"] * 1e3, "launches_timed": prof["searches"]},
        "cpu_baseline": None, "verified_against_oracle": bool(verified),
    }
    emit(out)
    sh.close()


# ---------------------------------------------------------------------------------------------- one rank
def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1 and not a.abi_sharded:
        return self_launch(a)
    quiet_stdout()
    if a.abi_sharded:
        if a.preflight:
            return preflight_abi(a)
        return run_abi_sharded(a)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    a.gpus = world
    import torch
    import torch.distributed as dist
    ndev = max(torch.cuda.device_count(), 1)
    shared_devices = world > ndev                        # dry run: more ranks than GPUs
    device_id = local_rank % ndev
    torch.cuda.set_device(device_id)
    use_pg = world > 1 or a.force_exchange or a.preflight
    backend = "gloo" if shared_devices else "nccl"
    if world > 1 or a.preflight:                             # before anything is generated: a failed scaling run says why in seconds
        preflight_devices(torch, world, rank)
        if shared_devices and rank == 0:
            _say("%d ranks on %d device(s): ranks share devices, the exchange runs over gloo (RCCL needs one device per rank)" % (world, ndev))
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import datetime
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_id), timeout=datetime.timedelta(seconds=180))
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
        except Exception as ex:                                # noqa: BLE001
            raise SystemExit("[preflight] FAILED on rank %d: init_process_group(%s, world %d, MASTER %s:%s): %s" % (
                rank, backend, world, os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"), ex))
        if world > 1 or a.preflight:
            preflight_ranks(torch, dist, rank, world, device_id, backend, a.preflight)
        if a.preflight:
            dist.barrier()
            dist.destroy_process_group()
            if rank == 0:
                _say("process group destroyed: ok")
            return

    import quiver_amd
    from quiver_amd.device_index import device_info, runtime_info
    from quiver_amd.sharded import DeviceShard, ShardedFlatSearch, shard_bounds

    dim, k, G = a.dim, a.k, world
    base, n_local = shard_bounds(a.rows, G, rank)
    idx = quiver_amd.DeviceIndex(dim, a.metric, device=device_id)
    idx.reserve(n_local)
    t_gen = time.perf_counter()
    done = 0
    while done < n_local:                       # generate in slabs (one kernel launch each)
        m = min(n_local - done, 2_000_000)
        idx.add_synthetic(CORPUS_SEED, base + done, m)
        done += m
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen

    # queries: the product's own synthetic generator (seed QUERY_SEED), read back once
    nq_pool = 256
    qgen = quiver_amd.DeviceIndex(dim, a.metric, device=device_id)
    qgen.add_synthetic(QUERY_SEED, 0, nq_pool)
    qs_host = np.stack([qgen.get_row(i) for i in range(nq_pool)])
    qgen.close()
    d_q = torch.from_numpy(qs_host).cuda()
    sp = torch.cuda.current_stream().cuda_stream
    qsz = dim * 4
    total_steps = a.warmup + a.steps
    dev = torch.device("cuda", device_id)
    search = ShardedFlatSearch(DeviceShard(idx), base, k, dev, world=G, ring=total_steps + 2, force_exchange=a.force_exchange)

    # consecutive queries are independent: alternating between scan streams lets the ramp-up of scan i+1 fill the tail of scan i
    # Measured (profiles/r01_sweep.txt): 1.25M rows per GPU 0.5525 -> 0.5360 ms per query with two streams; from 2.5M rows up one
    # stream is faster (two long scans interleaving cost 0.7-3 %), so the second stream is used for short shards only.
    n_scan_streams = a.scan_streams if a.scan_streams > 0 else (2 if n_local <= 1_500_000 else 1)
    scan_streams = [torch.cuda.Stream(device=dev) for _ in range(n_scan_streams)]

    def run(first, count):
        """`count` steps; the exchange+merge of step i is issued after the scan of step i+1"""
        pending = None
        for i in range(first, first + count):
            with torch.cuda.stream(scan_streams[i % len(scan_streams)]):
                t = search.submit(d_q[i % nq_pool])
                if pending is not None:
                    search.finish(pending)
            pending = t
        if pending is not None:
            search.finish(pending)

    def barrier():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup, then EXACTLY K timed steps bracketed by barrier + synchronize ----
    run(0, a.warmup)
    barrier()
    t0 = time.perf_counter()
    run(a.warmup, a.steps)
    barrier()
    dt = time.perf_counter() - t0
    if use_pg:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    qps = a.steps / dt

    # results of the timed steps sit in the ring: slot j = step j
    def result_of(step):
        b = search._ring[step % len(search._ring)]
        r, d = (b["out_rows"], b["out_dist"]) if (G > 1 or a.force_exchange) else (b["rows"], b["dist"])
        return r.cpu().numpy().view(np.uint32), d.cpu().numpy()

    # ---- verification of what was timed: the CPU oracle as the CHECKER (never on the timed path) ----
    verified = True
    if rank == 0:
        from tests import _oracle as O
        verified &= bool(np.array_equal(qs_host[:4].view(np.uint32), O.gen_rows(QUERY_SEED, 0, 4, dim).view(np.uint32)))
        for j in range(min(a.steps, 8)):
            step = a.warmup + j
            rr, dd = result_of(step)
            q = qs_host[step % nq_pool]
            mid = quiver_amd.metric_id(a.metric)
            for t_ in range(k):
                want = O.distance(mid, q, O.gen_rows(CORPUS_SEED, int(rr[t_]), 1, dim)[0])
                verified &= bool(np.float32(want).view(np.uint32) == dd[t_].view(np.uint32))
            verified &= all(dd[t_] <= dd[t_ + 1] for t_ in range(k - 1))
        # ... and MEMBERSHIP: for two of the timed queries the oracle's distances over the first rows of the corpus — a scan that
        # skipped rows fails here, not only in pytest
        n_chk = min(a.rows, a.cpu_sample_rows)
        chk_rows = O.gen_rows(CORPUS_SEED, 0, n_chk, dim)
        for j in range(min(a.steps, 2)):
            step = a.warmup + j
            rr, dd = result_of(step)
            verified &= membership_check(O, quiver_amd.metric_id(a.metric), qs_host[step % nq_pool], chk_rows, 0, rr.astype(np.int64), dd)

    # ---- roofline of the dominant kernel: HIP events around k_flat_scan, separate pass ----
    d_r = torch.empty((k,), dtype=torch.int32, device="cuda")
    d_d = torch.empty((k,), dtype=torch.float32, device="cuda")
    idx.profile(True)
    for j in range(min(a.steps, 50)):
        idx.search_device(d_q.data_ptr() + (j % nq_pool) * qsz, 1, k, d_r.data_ptr(), d_d.data_ptr(), sp)
    torch.cuda.synchronize()
    scan_ms, launches = idx.profile_read()
    idx.profile(False)
    alg_bytes = n_local * dim * 4 + n_local * 8          # rows once + cached f64 row norms (cosine); SURVEY.md 8d
    kern_ms = scan_ms / max(launches, 1)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if launches else 0.0

    # ---- N > 1: every GPU's roofline fraction, and the exchange (all-gather + merge) latency on its own ----
    per_gpu, exchange_us = None, None
    if use_pg:
        mine = torch.tensor([achieved / HBM_PEAK_GBS, float(n_local), kern_ms], dtype=torch.float64, device="cuda")
        allv = torch.empty((G, 3), dtype=torch.float64, device="cuda")
        dist.all_gather_into_tensor(allv.view(-1), mine)
        per_gpu = [{"rank": g, "rows": int(allv[g, 1].item()), "scan_kernel_ms": float(allv[g, 2].item()), "hbm_frac": float(allv[g, 0].item())} for g in range(G)]
        b = search._ring[0]
        reps = 50
        barrier()
        t1 = time.perf_counter()
        for _ in range(reps):                            # the exchange step alone, queues empty: one all-gather of [2][k] words + one merge launch
            w = dist.all_gather_into_tensor(b["g_pack"].view(-1), b["pack"].view(-1), async_op=True)
            search.finish((b, (w,)))
            torch.cuda.synchronize()
        exchange_us = (time.perf_counter() - t1) / reps * 1e6
        t = torch.tensor([exchange_us], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exchange_us = float(t.item())

    also = None
    if G == 1 and rank == 0 and not a.no_also and a.metric == "cosine":
        also = also_entries(a, torch, quiver_amd, idx, d_q, qs_host, device_id)

    cpu = None
    if rank == 0 and G == 1 and not a.no_cpu_baseline:          # the CPU baseline is a single-GPU-run artefact (rank 0, N=1 only)
        cpu = cpu_baseline(dim, k, a.cpu_sample_rows, a.cpu_sample_queries, a.rows, a.cpu_threads)

    if rank == 0:
        info = device_info(device_id)
        traffic, traffic_src = pmc_traffic(n_local, dim)
        out = {
            "metric": "flat_cosine_qps_recall_1.0", "value": qps, "unit": "queries/s",
            "n_gpus": G, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "flat %s scan %dx%d fp32, k=%d, single query per step, recall 1.0 (exact, bit-identical to the CPU oracle)" % (a.metric, a.rows, dim, k),
                       "rows_total": a.rows, "rows_per_gpu": n_local, "dim": dim, "k": k,
                       "arithmetic": "float64 accumulation over float32 rows, one rounding to float32 (the reference's arithmetic)",
                       "scan_streams": n_scan_streams,
                       "sharding": ("contiguous row shards, one process per GPU, per-shard top-k + all-gather (k*8 B/rank) + deterministic merge; "
                                    "exchange of step i overlaps scan of step i+1") if use_pg else "single shard",
                       "exchange": None if not use_pg else ("RCCL (torch.distributed nccl backend)" if backend == "nccl" else "gloo (ranks share a device: RCCL needs one device per rank)"),
                       # HBM per row: float32 tiles + float64 norm + alive bit; QV_FLAG_ROWMAJOR (the device HNSW's gathers) adds a second
                       # float32 copy, QV_FLAG_BF16_ROWS (the default batched filter) a bfloat16 one: an index serving all three paths holds 2.5x
                       "device_bytes_per_row": {"this_index": dim * 4 + 8.125, "plus_rowmajor_for_hnsw": dim * 8 + 8.125, "plus_rowmajor_and_bf16_rows": dim * 10 + 8.125},
                       "device": info["name"], "cus": info["cus"], "corpus_gen_s": round(t_gen, 3),
                       "runtime": short_runtime(runtime_info()),
                       # how many RCCL ranks took part in the all-gather that was timed: the process group's size when its backend is nccl
                       # (= RCCL); 0 when the exchange ran over gloo or there was no exchange
                       "rccl_ranks_seen": (dist.get_world_size() if (use_pg and backend == "nccl") else 0)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_flat_scan", "kernel_ms": kern_ms, "launches_timed": launches,
                         "algorithmic_bytes_per_launch": alg_bytes, "traffic_source": traffic_src,
                         "per_gpu": per_gpu, "allgather_plus_merge_us": exchange_us},
            "cpu_baseline": cpu,
            "verified_against_oracle": bool(verified),
        }
        if also:
            out["also"] = also
        emit(out)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()
    idx.close()


if __name__ == "__main__":
    main()
