"""The `also` entries of bench.py's line (N = 1 only): BASELINE.json's other configurations and the serving shapes around them, each
measured on the corpus the headline run already holds (or a 1M x 768 one), every result compared with the exact scan / the oracle.

bench.py calls `also_entries(...)`; nothing here is part of the timed headline region.  One entry per key of the returned dict;
`bench.compact_also` picks one or two numbers of each for the contract line, the rest goes to the sidecar (profiles/ or gpurun_out/).
"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# roofline constants and seeds: bench.py's (one definition)
from bench import CORPUS_SEED, HBM_PEAK_GBS, MFMA_BF16_PEAK_TF, MFMA_F32_PEAK_TF, MFMA_F64_PEAK_TF, QUERY_SEED  # noqa: E402


def also_entries(a, torch, quiver_amd, idx, d_q, qs_host, local_rank):
    dim, k, nq_pool = a.dim, a.k, qs_host.shape[0]
    sp = torch.cuda.current_stream().cuda_stream
    qsz = dim * 4
    d_r = torch.empty((k,), dtype=torch.int32, device="cuda")
    d_d = torch.empty((k,), dtype=torch.float32, device="cuda")
    also = {}
    # host-pointer entry point on the same corpus: query up over PCIe, results down, one stream sync per query
    idx.search(qs_host[0], k)
    t1 = time.perf_counter()
    for j in range(20):
        idx.search(qs_host[j], k)
    also["pcie_inclusive_single_query"] = {"workload": "qv_index_search (host pointers) on the same %dx%d corpus" % (a.rows, dim),
                                           "qps": 20 / (time.perf_counter() - t1)}
    # configs[1]/[2] live on a 1M x 768 corpus
    idx1 = idx if a.rows == 1_000_000 else quiver_amd.DeviceIndex(dim, a.metric, device=local_rank)
    if idx1 is not idx:
        idx1.reserve(1_000_000)
        idx1.add_synthetic(CORPUS_SEED, 0, 1_000_000)
    for j in range(20):
        idx1.search_device(d_q.data_ptr() + (j % nq_pool) * qsz, 1, k, d_r.data_ptr(), d_d.data_ptr(), sp)
    torch.cuda.synchronize()
    steps1 = 500
    t1 = time.perf_counter()
    for j in range(steps1):
        idx1.search_device(d_q.data_ptr() + (j % nq_pool) * qsz, 1, k, d_r.data_ptr(), d_d.data_ptr(), sp)
    torch.cuda.synchronize()
    dt1 = time.perf_counter() - t1
    idx1.profile(True)
    for j in range(100):
        idx1.search_device(d_q.data_ptr() + (j % nq_pool) * qsz, 1, k, d_r.data_ptr(), d_d.data_ptr(), sp)
    torch.cuda.synchronize()
    ms1, n1 = idx1.profile_read()
    idx1.profile(False)
    b1 = 1_000_000 * dim * 4 + 1_000_000 * 8
    also["flat_1Mx768_single_query"] = {
        "workload": "flat cosine 1Mx768 fp32, k=10, single query (BASELINE configs[1])",
        "qps": steps1 / dt1, "ms_per_query": dt1 / steps1 * 1e3, "scan_kernel_ms": ms1 / max(n1, 1),
        "hbm_gbs": b1 / (ms1 / max(n1, 1) * 1e-3) / 1e9, "hbm_frac": b1 / (ms1 / max(n1, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS}
    # The traffic the reference's unchanged host produces on that corpus: Collection.Search holds a read lock and calls Index.Search(q, k)
    # once per request (collection.go:647); DB.BatchSearch reaches a batch entry only through a type assertion on the reference's own
    # wrapper (db.go:726-727) and otherwise fans out one goroutine per query (:805-828).  T native threads, qv_index_search with nq = 1,
    # closed loop (tools/native/qv_callers.cpp); libqv lets such callers share passes (qv_coalesce.h).  Every result is compared with the
    # first one seen for the same query, and those with one batch call.
    try:
        from tests import _callers
        ref_r, ref_d, _ = idx1.search(qs_host[:256], k, batched=True)
        by = []
        for t in (1, 8, 64, 256):
            cr = _callers.run("index", idx1.handle, qs_host[:256], k, threads=t, seconds=1.0)
            seen = cr["count"] != 0xFFFFFFFD
            same = bool(np.array_equal(cr["rows"][seen], ref_r[seen]) and np.array_equal(cr["dist"][seen].view(np.uint32), ref_d[seen].view(np.uint32)))
            by.append({"callers": t, "qps": cr["qps"], "p50_us": cr["p50_us"], "p99_us": cr["p99_us"], "errors": cr["errors"], "mismatches": cr["mismatches"],
                       "same_as_batch_call": same})
        st = _callers.coalesce_stats("index", idx1.handle)
        also["concurrent_single_query_callers"] = {
            "workload": "T threads x qv_index_search(nq = 1, k = %d) on 1Mx768 cosine, host pointers, closed loop" % k, "by_callers": by,
            "passes_shared": {"groups": st["groups"], "mean_queries_per_group": st["group_queries"] / max(st["groups"], 1), "solo_calls": st["solo"]}}
    except Exception as ex:                                # noqa: BLE001
        also["concurrent_single_query_callers"] = {"error": str(ex)}
    # configs[2]: 256 queries x 1M x 768
    nqb = 256
    d_rb = torch.empty((nqb, k), dtype=torch.int32, device="cuda")
    d_db = torch.empty((nqb, k), dtype=torch.float32, device="cuda")
    idx1.search_device(d_q.data_ptr(), nqb, k, d_rb.data_ptr(), d_db.data_ptr(), sp)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        idx1.search_device(d_q.data_ptr(), nqb, k, d_rb.data_ptr(), d_db.data_ptr(), sp)
    torch.cuda.synchronize()
    dtb = (time.perf_counter() - t1) / 5
    exact_rows = d_rb.cpu().numpy().view(np.uint32).copy()
    exact_dist = d_db.cpu().numpy().copy()
    flop = 2.0 * nqb * 1_000_000 * dim
    also["batched_256x1Mx768_exact_scan"] = {
        "workload": "256 queries x 1Mx768 cosine, k=10: exact multi-query scan on the f64 matrix cores (32 queries per corpus pass; v_mfma_f64 chains are bit-identical to the scalar f64 loop)",
        "batch_ms": dtb * 1e3, "qps": nqb / dtb, "f64_tflops_equiv": flop / dtb / 1e12,
        "frac_of_f64_matrix_peak": flop / dtb / 1e12 / MFMA_F64_PEAK_TF, "peak_tflops_measured": MFMA_F64_PEAK_TF}

    # which one-term kernel the library dispatches for this shape (qv_batched.hip qreg_filter_applies: 384 / 512 / 768 dimensions, whole
    # workgroups of 256 queries, QV_QREG != 2); QV_TRACE=1 prints the same decision from inside the library
    qreg_applies = dim in (384, 512, 768) and nqb % 256 == 0 and os.environ.get("QV_QREG", "") != "2"

    def filter_kernel_name(plane):
        if qreg_applies:
            return "k_qreg_filter (bfloat16 copy)" if plane else "k_qreg_filter (float32 rows)"
        return "k_bf16rows_filter (bfloat16 copy)" if plane and dim % 128 == 0 else "k_bf16x1_filter_w8x2 (float32 rows)"

    def bare_mfma_f32():
        """the chip's bare v_mfma_f32_32x32x2_f32 issue rate under load, one and two waves per SIMD, with the shader clock it held
        (tools/native/qv_ubench.hip; k_mfma_filter runs one wave per SIMD)"""
        try:
            import ctypes as C
            ub = C.CDLL(os.path.join(ROOT, "quiver_amd", "lib", "libqvubench.so"))
            ub.qvu_mfma_f32_rate.restype = C.c_int
            ub.qvu_mfma_f32_rate.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
            r = {}
            for w in (1, 2):
                tf, ghz = C.c_double(0.0), C.c_double(0.0)
                if ub.qvu_mfma_f32_rate(w, C.byref(tf), C.byref(ghz)) == 0:
                    r[w] = (tf.value, ghz.value)
            return r
        except Exception:                                    # noqa: BLE001
            return {}

    def mfma_entry(index, label, want_rows, want_dist, rows_n=1_000_000, kernel="bf16x3", plane=False):
        index.set_filter(kernel)                                # qv_index_set_filter: the index's own choice of filter kernel
        d_flags = torch.zeros((nqb,), dtype=torch.int32, device="cuda")
        index.search_batched_device(d_q.data_ptr(), nqb, k, d_rb.data_ptr(), d_db.data_ptr(), d_flags.data_ptr(), sp)
        torch.cuda.synchronize()
        index.profile(True)
        t2 = time.perf_counter()
        for _ in range(10):
            index.search_batched_device(d_q.data_ptr(), nqb, k, d_rb.data_ptr(), d_db.data_ptr(), d_flags.data_ptr(), sp)
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t2) / 10
        msm, nm = index.profile_read()
        index.profile(False)
        redo = int(d_flags.sum().item())
        rb, db = d_rb.cpu().numpy().view(np.uint32), d_db.cpu().numpy()
        done = d_flags.cpu().numpy() == 0                       # a flagged query (candidate buffer overflow) is the caller's to redo exactly
        same = bool(np.array_equal(rb[done], want_rows[done]) and np.array_equal(db.view(np.uint32)[done], want_dist.view(np.uint32)[done]))
        mf_ms = msm / max(nm, 1)
        flop = 2.0 * nqb * rows_n * dim
        index.set_filter("auto")
        bare = bare_mfma_f32() if kernel == "fp32" else {}
        entry = {
            "workload": "256 queries x %dx768 %s, k=10 (BASELINE configs[2]): %s filter + exact re-score, device-resident queries and "
                        "results (sample scan, prep, filter, re-score all inside the timed region)"
                        % (rows_n, label, "fp32-MFMA (v_mfma_f32_32x32x2_f32: the dense fp32 GEMM as written)" if kernel == "fp32" else
                           "bfloat16 x 3 MFMA (three exact-product v_mfma_f32_32x32x16_bf16 terms per operand pair: float32-class scores with a "
                           "proven margin, a quarter of the matrix cycles)" if kernel == "bf16x3" else
                           ("bfloat16 x 1 MFMA (one v_mfma_f32_32x32x16_bf16 term, |score error| <= 7.9e-3 |q||r| proven, a 4x larger sample to "
                            "bound the candidates: the library's default up to 1536 dimensions)" +
                            (" reading the index's bfloat16 copy of the rows (QV_FLAG_BF16_ROWS, +50 % memory)" if plane else ""))),
            "batch_ms": dtm * 1e3, "qps": nqb / dtm, "identical_to_exact_scan": same, "queries_sent_back_to_exact_scan": redo,
            "roofline": ({"bound": "mfma", "kernel": "k_mfma_filter", "kernel_ms": mf_ms, "achieved": flop / (mf_ms * 1e-3) / 1e12,
                          "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": flop / (mf_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF,
                          "end_to_end_frac": flop / dtm / 1e12 / MFMA_F32_PEAK_TF, "algorithmic_flop_per_launch": flop} if kernel == "fp32" else
                         {"bound": "hbm", "kernel": filter_kernel_name(plane), "kernel_ms": mf_ms,
                          "achieved": rows_n * (dim * (2 if plane else 4) + 8) / (mf_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": rows_n * (dim * (2 if plane else 4) + 8) / (mf_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "algorithmic_bytes_per_launch": rows_n * (dim * (2 if plane else 4) + 8),
                          "why_hbm": "one term is 0.39 PFLOP of bfloat16 matrix work per launch = 0.16 ms at 2.5 PFLOP/s; reading the float32 rows once is 0.38 ms at 8 TB/s, the bfloat16 copy 0.19 ms",
                          "matrix_pipe_note": "on this data the chip holds ~1.5 GHz under a bare chain of these matrix instructions: the kernel's own K loop with nothing but its 12M v_mfma_f32_32x32x16_bf16 per SIMD-set takes 0.255 ms (1.54 PFLOP/s, QV_QREG_DBG=15 build) — the practical floor of the bfloat16-copy form; the float32-row form's row stream alone takes 0.44 ms (6.95 TB/s)",
                          # (that constant was measured at 256 x 1M x 768 on the query-resident kernel: only there does the ratio mean anything)
                          "frac_of_bare_mfma_loop": (0.255 / mf_ms) if (qreg_applies and rows_n == 1_000_000 and dim == 768) else None,
                          "matrix_tflops": flop / (mf_ms * 1e-3) / 1e12, "fp32_equivalent_tflops_end_to_end": flop / dtm / 1e12,
                          "times_the_fp32_mfma_peak_end_to_end": flop / dtm / 1e12 / MFMA_F32_PEAK_TF} if kernel == "bf16x1" else
                         {"bound": "mfma", "kernel": "k_bf16x3_filter_shared", "kernel_ms": mf_ms, "achieved": 3.0 * flop / (mf_ms * 1e-3) / 1e12,
                          "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": 3.0 * flop / (mf_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF,
                          "matrix_flop_per_launch": 3.0 * flop, "fp32_equivalent_tflops_end_to_end": flop / dtm / 1e12,
                          "times_the_fp32_mfma_peak_end_to_end": flop / dtm / 1e12 / MFMA_F32_PEAK_TF})}
        if bare.get(1):
            # what the instruction itself reaches on this box, right now: the kernel holds ONE wave per SIMD (336 registers: a 64 x 128 tile
            # of accumulators + three steps of operands), and a lone wave leaves a few cycles between two matrix instructions
            rf = entry["roofline"]
            rf.update({"bare_mfma_loop_tflops_1_wave_per_simd": bare[1][0], "bare_mfma_loop_clock_ghz": bare[1][1],
                       "bare_mfma_loop_tflops_2_waves_per_simd": (bare.get(2) or (None, None))[0],
                       "frac_of_bare_mfma_loop": rf["achieved"] / bare[1][0],
                       "peak_at_the_held_clock_tflops": MFMA_F32_PEAK_TF * bare[1][1] / 2.4, "frac_at_the_held_clock": rf["achieved"] / (MFMA_F32_PEAK_TF * bare[1][1] / 2.4)})
        return entry
    # more than 64 results per query (round 4): the negative-example branches fetch max(2k, 30) (hybrid_index.go:516-522), BatchSearch
    # takes any k (:677-811).  k = 100 and 1000, one query (10M and 1M rows) and 256 queries x 1M rows; checked against the full ranking.
    try:
        ks_entry = {"workload": "k above the 64-key wave list: single query (wide wave lists to 128, one key per row + radix selection to 8192) and "
                                "256-query batches (filter + re-score to 4096: round 6 — the bound a guess from 65 536 sample rows checked after the "
                                "filter, the candidates narrowed by one kernel per query, the survivors' exact distances by a wave per 32 rows or, for "
                                "many survivors, one pass over the tiles; profiles/r06_largek_pmc.txt); ms per call, device-resident"}
        for label_k, index_k, rows_k in (("1x%dM" % (a.rows // 1_000_000), idx, a.rows), ("1x1M", idx1, 1_000_000)):
            full_r = torch.empty((1, rows_k), dtype=torch.int32, device="cuda"); full_d = torch.empty((1, rows_k), dtype=torch.float32, device="cuda")
            index_k.search_device(d_q.data_ptr(), 1, rows_k, full_r.data_ptr(), full_d.data_ptr(), sp)      # the full ranking (radix sort): the checker
            for kk_ in (10, 100, 1000):
                rr_ = torch.empty((1, kk_), dtype=torch.int32, device="cuda"); dd_ = torch.empty((1, kk_), dtype=torch.float32, device="cuda")
                index_k.search_device(d_q.data_ptr(), 1, kk_, rr_.data_ptr(), dd_.data_ptr(), sp)
                torch.cuda.synchronize()
                t4 = time.perf_counter()
                for _ in range(20):
                    index_k.search_device(d_q.data_ptr(), 1, kk_, rr_.data_ptr(), dd_.data_ptr(), sp)
                torch.cuda.synchronize()
                ks_entry["%s_k%d_ms" % (label_k, kk_)] = (time.perf_counter() - t4) / 20 * 1e3
                ks_entry["%s_k%d_same" % (label_k, kk_)] = bool(torch.equal(rr_, full_r[:, :kk_]) and torch.equal(dd_.view(torch.int32), full_d[:, :kk_].view(torch.int32)))
            del full_r, full_d
        for kk_ in (10, 64, 100, 1000, 4096):
            rb_ = torch.empty((nqb, kk_), dtype=torch.int32, device="cuda"); db_ = torch.empty((nqb, kk_), dtype=torch.float32, device="cuda")
            fl_ = torch.zeros((nqb,), dtype=torch.int32, device="cuda")
            for _ in range(3):                             # (first launches of a shape: workspace growth, clocks)
                idx1.search_batched_device(d_q.data_ptr(), nqb, kk_, rb_.data_ptr(), db_.data_ptr(), fl_.data_ptr(), sp)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            for _ in range(20):
                idx1.search_batched_device(d_q.data_ptr(), nqb, kk_, rb_.data_ptr(), db_.data_ptr(), fl_.data_ptr(), sp)
            torch.cuda.synchronize()
            ks_entry["256x1M_k%d_ms" % kk_] = (time.perf_counter() - t4) / 20 * 1e3
            # checked against the exact scans of the same index (filter off: multi-query scan / key per row + selection); a query the filter
            # handed back (flag set) is the caller's to redo and is left out
            xr_ = torch.empty((nqb, kk_), dtype=torch.int32, device="cuda"); xd_ = torch.empty((nqb, kk_), dtype=torch.float32, device="cuda")
            idx1.set_filter("off")
            idx1.search_device(d_q.data_ptr(), nqb, kk_, xr_.data_ptr(), xd_.data_ptr(), sp)
            torch.cuda.synchronize()
            idx1.set_filter("auto")
            ok_ = fl_ == 0
            ks_entry["256x1M_k%d_same" % kk_] = bool(torch.equal(rb_[ok_], xr_[ok_]) and torch.equal(db_[ok_].view(torch.int32), xd_[ok_].view(torch.int32)))
            ks_entry["256x1M_k%d_handed_back" % kk_] = int((~ok_).sum().item())
            del xr_, xd_
        also["k_above_64"] = ks_entry
    except Exception as ex:                                # noqa: BLE001
        also["k_above_64"] = {"error": str(ex)}
    also["batched_256x1Mx768_mfma"] = mfma_entry(idx1, "cosine", exact_rows, exact_dist, kernel="fp32")
    also["batched_256x1Mx768_bf16x3"] = mfma_entry(idx1, "cosine", exact_rows, exact_dist)
    also["batched_256x1Mx768_bf16x1"] = mfma_entry(idx1, "cosine", exact_rows, exact_dist, kernel="bf16x1")
    try:                                                   # the same with the optional bfloat16 copy of the rows
        ibf = quiver_amd.DeviceIndex(dim, a.metric, device=local_rank, bf16_rows=True)
        ibf.reserve(1_000_000)
        ibf.add_synthetic(CORPUS_SEED, 0, 1_000_000)
        also["batched_256x1Mx768_bf16x1_bf16rows"] = mfma_entry(ibf, "cosine", exact_rows, exact_dist, kernel="bf16x1", plane=True)
        # the usual BatchSearch sizes: one query block (k_bf16rows_filter_q64 with the plane, the per-wave three-term kernel without)
        small = {}
        for nqs in (16, 64):
            for label_s, index_s in (("float32_rows", idx1), ("bf16_rows", ibf)):
                fl_s = torch.zeros((nqs,), dtype=torch.int32, device="cuda")
                index_s.search_batched_device(d_q.data_ptr(), nqs, k, d_rb.data_ptr(), d_db.data_ptr(), fl_s.data_ptr(), sp)
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                for _ in range(10):
                    index_s.search_batched_device(d_q.data_ptr(), nqs, k, d_rb.data_ptr(), d_db.data_ptr(), fl_s.data_ptr(), sp)
                torch.cuda.synchronize()
                dts = (time.perf_counter() - t3) / 10
                ok_s = bool(np.array_equal(d_rb[:nqs].cpu().numpy().view(np.uint32), exact_rows[:nqs]) and
                            np.array_equal(d_db[:nqs].cpu().numpy().view(np.uint32), exact_dist[:nqs].view(np.uint32))) and int(fl_s.sum().item()) == 0
                small["%d_queries_%s" % (nqs, label_s)] = {"batch_ms": dts * 1e3, "qps": nqs / dts, "identical_to_exact_scan": ok_s}
        also["batched_small_1Mx768"] = small
        ibf.close()
    except Exception as ex:                                # noqa: BLE001
        also["batched_256x1Mx768_bf16x1_bf16rows"] = {"error": str(ex)}
    try:                                                   # configs[2] as written: dot-product
        idot = quiver_amd.DeviceIndex(dim, "dot_product", device=local_rank)
        idot.reserve(1_000_000)
        idot.add_synthetic(CORPUS_SEED, 0, 1_000_000)
        idot.search_device(d_q.data_ptr(), nqb, k, d_rb.data_ptr(), d_db.data_ptr(), sp)
        torch.cuda.synchronize()
        dot_rows, dot_dist = d_rb.cpu().numpy().view(np.uint32).copy(), d_db.cpu().numpy().copy()
        also["batched_256x1Mx768_mfma_dot"] = mfma_entry(idot, "dot-product (1 - dot, distances.go:77-90)", dot_rows, dot_dist, kernel="fp32")
        also["batched_256x1Mx768_bf16x3_dot"] = mfma_entry(idot, "dot-product (1 - dot, distances.go:77-90)", dot_rows, dot_dist)
        also["batched_256x1Mx768_bf16x1_dot"] = mfma_entry(idot, "dot-product (1 - dot, distances.go:77-90)", dot_rows, dot_dist, kernel="bf16x1")
        idot.close()
    except Exception as ex:                                # noqa: BLE001
        also["batched_256x1Mx768_mfma_dot"] = {"error": str(ex)}
    if idx1 is not idx:
        idx1.close()
    # the same two batched paths on the headline corpus (10M rows): the fixed costs of a batch amortise
    if a.rows > 1_000_000:
        try:
            idx.search_device(d_q.data_ptr(), nqb, k, d_rb.data_ptr(), d_db.data_ptr(), sp)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                idx.search_device(d_q.data_ptr(), nqb, k, d_rb.data_ptr(), d_db.data_ptr(), sp)
            torch.cuda.synchronize()
            dtb = (time.perf_counter() - t1) / 3
            big_rows, big_dist = d_rb.cpu().numpy().view(np.uint32).copy(), d_db.cpu().numpy().copy()
            flop_big = 2.0 * nqb * a.rows * dim
            also["batched_256x%dMx768_exact_scan" % (a.rows // 1_000_000)] = {
                "workload": "256 queries x %dx768 cosine, k=10: exact multi-query scan on the f64 matrix cores" % a.rows,
                "batch_ms": dtb * 1e3, "qps": nqb / dtb, "f64_tflops_equiv": flop_big / dtb / 1e12,
                "frac_of_f64_matrix_peak": flop_big / dtb / 1e12 / MFMA_F64_PEAK_TF}
            also["batched_256x%dMx768_mfma" % (a.rows // 1_000_000)] = mfma_entry(idx, "cosine", big_rows, big_dist, a.rows, kernel="fp32")
            also["batched_256x%dMx768_bf16x3" % (a.rows // 1_000_000)] = mfma_entry(idx, "cosine", big_rows, big_dist, a.rows)
            also["batched_256x%dMx768_bf16x1" % (a.rows // 1_000_000)] = mfma_entry(idx, "cosine", big_rows, big_dist, a.rows, kernel="bf16x1")
            ibf = quiver_amd.DeviceIndex(dim, a.metric, device=local_rank, bf16_rows=True)
            ibf.reserve(a.rows)
            ibf.add_synthetic(CORPUS_SEED, 0, a.rows)
            also["batched_256x%dMx768_bf16x1_bf16rows" % (a.rows // 1_000_000)] = mfma_entry(ibf, "cosine", big_rows, big_dist, a.rows, kernel="bf16x1", plane=True)
            ibf.close()
        except Exception as ex:                            # noqa: BLE001
            also["batched_256x%dMx768" % (a.rows // 1_000_000)] = {"error": str(ex)}
    # configs[0]: the reference's own CPU-runnable case, 10k x 128 cosine k=10, one query at a time through the host-pointer
    # C ABI (query up, results down, one sync per call) — latency, not bandwidth; the CPU port beside it on the same rows
    try:
        c0 = quiver_amd.DeviceIndex(128, a.metric, device=local_rank)
        c0.add_synthetic(CORPUS_SEED, 0, 10_000)
        q0g = quiver_amd.DeviceIndex(128, a.metric, device=local_rank)
        q0g.add_synthetic(QUERY_SEED, 0, 64)
        q0 = np.stack([q0g.get_row(i) for i in range(64)])
        q0g.close()
        for j in range(50):
            c0.search(q0[j % 64], k)
        t1 = time.perf_counter()
        for j in range(1000):
            r0, d0, _ = c0.search(q0[j % 64], k)
        dt0 = (time.perf_counter() - t1) / 1000
        entry = {"workload": "pkg/hybrid exact flat scan 10k x 128 fp32 cosine, k=10 (BASELINE configs[0]), one query per call, host pointers",
                 "latency_us": dt0 * 1e6, "qps_one_caller": 1.0 / dt0}
        if not a.no_cpu_baseline:
            from tests import _oracle as O
            rows0 = O.gen_rows(CORPUS_SEED, 0, 10_000, 128)
            f0 = O.Faithful(0, 128)
            for i in range(10_000):
                f0.insert("v%d" % i, rows0[i])
            f0.search(q0[0], k)
            t1 = time.perf_counter()
            for j in range(200):
                f0.search(q0[j % 64], k)
            dtc = (time.perf_counter() - t1) / 200
            entry["cpu_port_latency_us_1core"] = dtc * 1e6
            entry["identical_to_oracle"] = bool(np.array_equal(r0[0], O.exact_search(0, rows0, q0[999 % 64], k)[0]))
        try:
            # the same calls from plain C (what a cgo caller pays: no interpreter, no per-call allocations): tools/ubench/abi_latency.c,
            # compiled here and run as a child process; p50 / p99 of 5 000 calls and the aggregate rate of 8 concurrent callers
            import subprocess, tempfile
            root_ = os.path.dirname(os.path.abspath(__file__))
            exe_ = os.path.join(tempfile.gettempdir(), "qv_abi_latency_%d" % os.getpid())
            libdir_ = os.path.join(root_, "quiver_amd", "lib")
            subprocess.run(["gcc", "-O2", "-std=c11", "-I", os.path.join(root_, "include"), os.path.join(root_, "tools", "ubench", "abi_latency.c"),
                            "-L", libdir_, "-lqv", "-lm", "-lpthread", "-Wl,-rpath," + libdir_, "-o", exe_], check=True, capture_output=True, timeout=120)
            env_ = dict(os.environ); env_["HIP_VISIBLE_DEVICES"] = env_.get("HIP_VISIBLE_DEVICES", str(local_rank))
            out_ = subprocess.run([exe_, "10000", "128", str(k), "5000", "8"], check=True, capture_output=True, text=True, timeout=120, env=env_).stdout
            cj_ = json.loads(out_.strip().split("\n")[-1])
            entry["plain_c_caller"] = {"p50_us": cj_["p50_us"], "p99_us": cj_["p99_us"], "eight_callers_aggregate_qps": cj_["aggregate_qps"]}
            os.remove(exe_)
        except Exception as ex:                            # noqa: BLE001  (no compiler on the box, ...: the Python-side number stands)
            entry["plain_c_caller"] = {"skipped": str(ex)[:120]}
        also["config0_10kx128_single_query"] = entry
        c0.close()
    except Exception as ex:                                # noqa: BLE001
        also["config0_10kx128_single_query"] = {"error": str(ex)}
    # one query per call on SHORT collections of 768-d rows (the sizes the reference's own deployments have): host pointers, p50 of 300 calls
    # and the scan kernel's own time; since round 5 the tile-over-eight-waves form (k_flat_scan_split)
    try:
        ms_ = {}
        for n_ in (10_000, 30_000, 100_000):
            cs = quiver_amd.DeviceIndex(dim, a.metric, device=local_rank)
            cs.add_synthetic(CORPUS_SEED, 0, n_)
            for j in range(30):
                cs.search(qs_host[j % 32], k)
            lat_ = []
            for j in range(300):
                t1 = time.perf_counter(); rs_, ds_, _ = cs.search(qs_host[j % 32], k); lat_.append(time.perf_counter() - t1)
            lat_.sort()
            cs.profile(True)
            for j in range(50):
                cs.search(qs_host[j % 32], k)
            kms_, kl_ = cs.profile_read(); cs.profile(False)
            e_ = {"p50_us": lat_[150] * 1e6, "p99_us": lat_[297] * 1e6, "scan_kernel_us": kms_ / max(kl_, 1) * 1e3, "hbm_time_us": n_ * dim * 4 / (HBM_PEAK_GBS * 1e9) * 1e6}
            if not a.no_cpu_baseline and n_ <= 30_000:
                from tests import _oracle as O
                e_["identical_to_oracle"] = bool(np.array_equal(rs_[0], O.exact_search(0, O.gen_rows(CORPUS_SEED, 0, n_, dim), qs_host[299 % 32], k)[0]))
            ms_["%dkx%d" % (n_ // 1000, dim)] = e_
            cs.close()
        also["short_collections_single_query"] = ms_
    except Exception as ex:                                # noqa: BLE001
        also["short_collections_single_query"] = {"error": str(ex)}
    # configs[3]: HNSW M=16 (MaxM0=32) efConstruction=200 over 1M x 768, the graph INSERTION-BUILT on the device
    if not a.no_hnsw:
        from tests.bench.bench_hnsw_build import run as hnsw_run
        cpuq = 0 if a.no_cpu_baseline else 20
        # the structured corpus first: a graph search finds neighbours there, so its QPS is a QPS AT a recall; BASELINE's i.i.d. corpus and
        # the reference's default MaxLevel follow, each carrying its recall against the exact top-10 next to its QPS
        for key, max_level, efs, idim, note in (
                ("hnsw_1Mx768_maxlevel1_structured", 1, (16, 32, 64, 128, 256), 16,
                 "the same index shape over data WITH neighbourhood structure (unit vectors on a 16-dimensional subspace of R^768, the "
                 "regime embeddings live in): same bytes per row, same kernels — this is the 'QPS @ recall' curve of the graph search"),
                ("hnsw_1Mx768_maxlevel1", 1, (64, 128, 256, 512), 0,
                 "MaxLevel=1: every node on level 0, so the level quirk of the reference's default (next entry) is out of play and the level-0 graph is one connected "
                 "M=16/MaxM0=32 graph — the configuration in which 'QPS @ recall' describes a graph search.  BASELINE's corpus is "
                 "uniformly random 768-d unit vectors, which have no neighbourhood structure: recall stays low at any efSearch"),
                ("hnsw_1Mx768_reference_defaults", 16, (128,), 0,
                 "MaxLevel=16, the reference's default.  Its connectNode re-enters the lower levels from the new node itself "
                 "(hnsw.go:463-467), so every node of level >= 1 links only to itself on level 0 and the level-0 graph is a forest of "
                 "small islands around those nodes: graph traversals return few results and HNSW.Search completes most queries with its "
                 "brute-force top-up (hnsw.go:676-710).  Reproduced faithfully (the build equals the CPU restatement's); the numbers "
                 "below are what that structure gives")):
            try:
                e = hnsw_run(rows=a.hnsw_rows, dim=dim, metric=a.metric, m=16, efc=200, max_level=max_level, efs=efs, nq=8192, k=k,
                             cpu_queries=cpuq, device=local_rank, corpus_seed=CORPUS_SEED, query_seed=QUERY_SEED, intrinsic_dim=idim,
                             callers=(1, 8, 64, 256, 1024) if key == "hnsw_1Mx768_maxlevel1" else ())
                e["note"] = note
                also[key] = e
            except Exception as ex:                        # noqa: BLE001  (a measurement beside the headline; never fail the bench line over it)
                also[key] = {"error": str(ex)}
    return also
