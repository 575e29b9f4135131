// qv_graph_api.cpp — the HNSW part of the C ABI (include/qv.h): device-resident traversal of a graph the host built
// (qv_graph_create / qv_graph_search*) and device-resident construction (qv_graph_create_empty / qv_graph_insert /
// qv_graph_build / qv_graph_export).  Host-side responsibilities only: argument checks, device-memory ownership, the
// batch schedule of a build and the entry-point bookkeeping the reference does under its lock (hnsw.go:325-332).
// No CPU compute path: every distance and every selection runs in a HIP kernel (qv_hnsw.hip, qv_build.hip).
#include "qv_api_internal.h"

struct qv_graph {
    qv_index* idx = nullptr;
    qv::GraphView g{};
    // device arrays, sized by cap_nodes / cap_blocks
    int8_t* d_level = nullptr; uint32_t* d_l0deg = nullptr; uint32_t* d_l0links = nullptr; uint32_t* d_upoff = nullptr; uint32_t* d_uplinks = nullptr;
    float* d_l0dist = nullptr; float* d_updist = nullptr;      // per-link distances: only graphs built on the device carry them
    uint32_t cap_nodes = 0, cap_blocks = 0, n_blocks = 0;
    bool buildable = false;
    uint32_t efc = 0;
    std::vector<int8_t> h_level;                // host mirror of the node levels (entry-point bookkeeping, export)
    // visited sets (qv_hnsw.hip): a hash table per wave slot for the wave kernel, a bitmap per slot for the exact-heap kernel
    Buf vis_hash; uint32_t vis_hash_cap = 0;
    Buf vis_bits; uint32_t vis_bits_words = 0, vis_bits_slots = 0;
    uint32_t grid = 0;                          // wave slots of the wave-resident kernel
    uint64_t tie_reruns = 0;
    std::mutex mu;                              // one batch at a time (the visited sets are per wave slot)
    hipStream_t stream = nullptr;
    hipEvent_t ev_last = nullptr;               // end of the most recent traversal: the next one (on any stream) waits for it
    Buf d_q, d_qblk, d_rows, d_dist, d_cnt, d_ev;
    PinBuf h_stage[2];                          // pinned bounce buffers for the query upload (pageable callers)
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    hipStream_t stream2 = nullptr;              // exact-heap passes of qv_graph_search beside the next part's wave pass: only QV_HNSW_OVERLAP_REDO=1 creates it
    hipEvent_t ev_part[2] = {nullptr, nullptr}, ev_heap = nullptr;
    Buf s_redo, s_counters;                     // qv_graph_search's own redo list and counters (the construction has b_redo / b_counters)
    PinBuf h_counters;                          // counters read back: a pinned member, never the stack (an early return must not leave a copy in flight into a dead frame)
    // build workspace
    Buf b_self, b_keys_a, b_keys_b, b_hist, b_seg, b_redo, b_counters;
    double build_seconds = 0.0; uint64_t build_redo = 0, build_batches = 0;

    qv::BuildView bview() const {
        qv::BuildView b;
        b.level = d_level; b.l0_deg = d_l0deg; b.l0_links = d_l0links; b.l0_dist = d_l0dist; b.up_off = d_upoff; b.up_links = d_uplinks; b.up_dist = d_updist;
        b.cap_nodes = cap_nodes; b.max_m0 = g.max_m0; b.max_m = g.max_m;
        return b;
    }
    void refresh_view() {
        g.level = d_level; g.l0_deg = d_l0deg; g.l0_links = d_l0links; g.up_off = d_upoff; g.up_links = d_uplinks;
    }
};

namespace {

// (re)allocate one device array to `new_count` elements, keeping the first `keep` and zeroing the rest
template <typename T> hipError_t regrow(T** p, size_t keep, size_t new_count) {
    T* n = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&n), std::max<size_t>(new_count * sizeof(T), 16));
    if (e != hipSuccess) return e;
    if (keep) e = hipMemcpy(n, *p, keep * sizeof(T), hipMemcpyDeviceToDevice);
    if (e == hipSuccess && new_count > keep) e = hipMemset(n + keep, 0, (new_count - keep) * sizeof(T));
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);   // (the graph's streams are non-blocking: the fill must have run before they touch the array)
    if (e != hipSuccess) { (void)hipFree(n); return e; }
    (void)hipFree(*p);
    *p = n;
    return hipSuccess;
}

int graph_reserve(qv_graph* g, uint64_t nodes, uint64_t blocks) {
    if (nodes > 0xFFFFFFF0ull || nodes + blocks > 0xFFFFFFF0ull) return fail(QV_ERR_INVALID_ARG, "graph of %llu nodes exceeds the uint32 id space", (unsigned long long)nodes);
    if (g->ev_last) HIPCHK(hipEventSynchronize(g->ev_last));
    const uint32_t m0 = g->g.max_m0, m = g->g.max_m;
    if (nodes > g->cap_nodes) {
        const size_t nn = std::max<uint64_t>(nodes, (uint64_t)g->cap_nodes + g->cap_nodes / 2), keep = g->g.n_nodes;
        hipError_t e = regrow(&g->d_level, keep, nn);
        if (e == hipSuccess) e = regrow(&g->d_l0deg, keep, nn);
        if (e == hipSuccess) e = regrow(&g->d_l0links, keep * m0, nn * m0);
        if (e == hipSuccess) e = regrow(&g->d_upoff, keep, nn);
        if (e == hipSuccess && g->buildable) e = regrow(&g->d_l0dist, keep * m0, nn * m0);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? QV_ERR_OOM : QV_ERR_DEVICE, "graph storage for %zu nodes failed: %s", nn, hipGetErrorString(e));
        g->cap_nodes = (uint32_t)nn;
    }
    if (blocks > g->cap_blocks || !g->d_uplinks) {
        const size_t nb = std::max<uint64_t>(std::max<uint64_t>(blocks, 1), (uint64_t)g->cap_blocks + g->cap_blocks / 2), keep = g->n_blocks;
        hipError_t e = regrow(&g->d_uplinks, keep * (1 + m), nb * (1 + m));
        if (e == hipSuccess && g->buildable) e = regrow(&g->d_updist, keep * m, nb * m);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? QV_ERR_OOM : QV_ERR_DEVICE, "graph storage for %zu upper-level lists failed: %s", nb, hipGetErrorString(e));
        g->cap_blocks = (uint32_t)nb;
    }
    g->refresh_view();
    return QV_OK;
}

// visited-set storage for a traversal with list capacity `ef` over the current node capacity
int ensure_visited(qv_graph* g, uint32_t ef) {
    qv_index* idx = g->idx;
    if (!g->grid) g->grid = qv::hnsw_wave_grid(idx->cus, idx->metric, idx->dim4);
    const uint32_t cap = qv::hnsw_vis_hash_cap(ef);
    if (cap > g->vis_hash_cap) {
        if (g->ev_last) HIPCHK(hipEventSynchronize(g->ev_last));
        int rc = g->vis_hash.ensure((size_t)g->grid * cap * 4);
        if (rc != QV_OK) return rc;
        g->vis_hash_cap = cap;
    }
    const uint32_t words = (uint32_t)((((uint64_t)std::max(g->cap_nodes, g->g.n_nodes) + 31) / 32 + 63) / 64 * 64);
    if (words > g->vis_bits_words) {
        if (g->ev_last) HIPCHK(hipEventSynchronize(g->ev_last));
        // the exact-heap kernel has few slots (LDS heaps): at most 8 per CU, and never more than 2 GiB of bitmaps
        uint32_t slots = (uint32_t)idx->cus * 8;
        while (slots > 64 && (size_t)slots * words * 4 > ((size_t)2 << 30)) slots /= 2;
        int rc = g->vis_bits.ensure((size_t)slots * words * 4);
        if (rc != QV_OK) return rc;
        g->vis_bits_words = words; g->vis_bits_slots = slots;
    }
    return QV_OK;
}

qv::HnswOpts wave_opts(qv_graph* g) { qv::HnswOpts o; o.vis = static_cast<uint32_t*>(g->vis_hash.p); o.vis_cap = g->vis_hash_cap; return o; }
qv::HnswOpts heap_opts(qv_graph* g) { qv::HnswOpts o; o.vis = static_cast<uint32_t*>(g->vis_bits.p); o.vis_cap = g->vis_bits_words; return o; }

int graph_common_init(qv_graph* g) {
    HIPCHK(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&g->ev_last, hipEventDisableTiming));
    return QV_OK;
}

}  // namespace

extern "C" {

int qv_graph_create(qv_graph** out, qv_index* idx, uint32_t n_nodes, const int8_t* levels, uint32_t max_m0, uint32_t max_m,
                    const uint32_t* l0_deg, const uint32_t* l0_links, const uint32_t* up_off, const uint32_t* up_links,
                    uint32_t n_up_blocks, uint32_t entry, int cur_level) {
    if (!out) return fail(QV_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (!idx || !levels || !l0_deg || !l0_links || !up_off) return fail(QV_ERR_INVALID_ARG, "null argument");
    if (n_nodes == 0 || n_nodes > idx->n_rows) return fail(QV_ERR_INVALID_ARG, "graph has %u nodes but the index holds %u rows", n_nodes, idx->n_rows);
    if (max_m0 == 0 || max_m0 > 64 || max_m > 64) return fail(QV_ERR_UNSUPPORTED, "degree bounds above 64 are not supported (MaxM0=%u, M=%u)", max_m0, max_m);
    if (entry >= n_nodes || levels[entry] < 0) return fail(QV_ERR_INVALID_ARG, "entry point %u is not a live node", entry);
    HIPCHK(hipSetDevice(idx->device));
    qv_graph* g = new (std::nothrow) qv_graph();
    if (!g) return fail(QV_ERR_OOM, "out of host memory");
    g->idx = idx;
    g->g.max_m0 = max_m0; g->g.max_m = max_m ? max_m : 1;
    int rc = graph_common_init(g);
    if (rc == QV_OK) rc = graph_reserve(g, n_nodes, std::max(n_up_blocks, 1u));
    if (rc != QV_OK) { qv_graph_destroy(g); return rc; }
    hipError_t e = hipMemcpy(g->d_level, levels, (size_t)n_nodes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(g->d_l0deg, l0_deg, (size_t)n_nodes * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(g->d_l0links, l0_links, (size_t)n_nodes * max_m0 * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(g->d_upoff, up_off, (size_t)n_nodes * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && up_links && n_up_blocks) e = hipMemcpy(g->d_uplinks, up_links, (size_t)n_up_blocks * (1 + g->g.max_m) * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) { qv_graph_destroy(g); return fail(QV_ERR_DEVICE, "graph upload failed: %s", hipGetErrorString(e)); }
    g->h_level.assign(levels, levels + n_nodes);
    g->n_blocks = n_up_blocks;
    g->g.n_nodes = n_nodes; g->g.entry = entry; g->g.cur_level = cur_level;
    g->g.has_dead = false;
    for (uint32_t i = 0; i < n_nodes; i++) if (levels[i] < 0) { g->g.has_dead = true; break; }
    *out = g;
    return QV_OK;
}

void qv_graph_destroy(qv_graph* g) {
    if (!g) return;
    if (g->idx) (void)hipSetDevice(g->idx->device);
    if (g->ev_last) { (void)hipEventSynchronize(g->ev_last); (void)hipEventDestroy(g->ev_last); }
    for (int i = 0; i < 2; i++) { if (g->ev_stage[i]) (void)hipEventDestroy(g->ev_stage[i]); g->h_stage[i].release(); }
    if (g->stream2) { (void)hipStreamSynchronize(g->stream2); (void)hipStreamDestroy(g->stream2); }
    for (int i = 0; i < 2; i++) if (g->ev_part[i]) (void)hipEventDestroy(g->ev_part[i]);
    if (g->ev_heap) (void)hipEventDestroy(g->ev_heap);
    if (g->stream) { (void)hipStreamSynchronize(g->stream); (void)hipStreamDestroy(g->stream); }
    (void)hipFree(g->d_level); (void)hipFree(g->d_l0deg); (void)hipFree(g->d_l0links); (void)hipFree(g->d_upoff); (void)hipFree(g->d_uplinks);
    (void)hipFree(g->d_l0dist); (void)hipFree(g->d_updist);
    g->vis_hash.release(); g->vis_bits.release();
    g->d_q.release(); g->d_qblk.release(); g->d_rows.release(); g->d_dist.release(); g->d_cnt.release(); g->d_ev.release();
    g->s_redo.release(); g->s_counters.release(); g->h_counters.release();
    g->b_self.release(); g->b_keys_a.release(); g->b_keys_b.release(); g->b_hist.release(); g->b_seg.release(); g->b_redo.release(); g->b_counters.release();
    delete g;
}

int qv_graph_search(qv_graph* g, const float* queries, uint32_t nq, uint32_t k, uint32_t ef_search,
                    uint32_t* rows_out, float* dist_out, uint32_t* count_out, uint32_t* evals_out) {
    if (!g) return fail(QV_ERR_INVALID_ARG, "graph is null");
    if (nq == 0) return QV_OK;
    if (!queries || !rows_out || !dist_out || !count_out) return fail(QV_ERR_INVALID_ARG, "null argument");
    if (k == 0) return fail(QV_ERR_K_NOT_POSITIVE, "k must be positive");                      // hnsw.go:610-612
    if (k > 512 || ef_search > 512) return fail(QV_ERR_UNSUPPORTED, "k and efSearch above 512 are not supported on the device path");
    if (g->g.n_nodes == 0) return fail(QV_ERR_INVALID_ARG, "graph is empty");
    qv_index* idx = g->idx;
    HIPCHK(hipSetDevice(idx->device));
    std::lock_guard<std::mutex> lock(g->mu);
    HIPCHK(hipEventSynchronize(g->ev_last));                            // a device-form traversal still in flight uses the buffers (re)sized below
    const size_t qbytes = (size_t)nq * idx->dim * sizeof(float), obytes = (size_t)nq * k * 4, cbytes = (size_t)nq * 4;
    const uint32_t efx = std::max(ef_search, k);
    int rc;
    if ((rc = g->d_q.ensure(qbytes)) || (rc = g->d_rows.ensure(obytes)) || (rc = g->d_dist.ensure(obytes)) || (rc = g->d_cnt.ensure(cbytes)) ||
        (rc = g->d_ev.ensure(cbytes)) || (rc = g->d_qblk.ensure(qv::hnsw_qblk_bytes(nq, idx->dim4))) || (rc = ensure_visited(g, efx)))
        return rc;
    HIPCHK(hipStreamWaitEvent(g->stream, g->ev_last, 0));               // after any device-form traversal still running on another stream
    static const bool trace = getenv("QV_TRACE") && atoi(getenv("QV_TRACE")) > 0;
    const auto t_begin = std::chrono::steady_clock::now();
    {   // upload through two pinned bounce buffers: the CPU copy of slice i+1 overlaps the DMA of slice i
        // (a single hipMemcpyAsync from pageable memory ran at 2.5-5 GB/s: a third of a 16k-query batch's time)
        const size_t slice = (size_t)8 << 20;
        size_t off = 0; int slot = 0;
        while (off < qbytes) {
            const size_t n = std::min(slice, qbytes - off);
            if ((rc = g->h_stage[slot].ensure(slice))) return rc;
            if (!g->ev_stage[slot]) HIPCHK(hipEventCreateWithFlags(&g->ev_stage[slot], hipEventDisableTiming));
            else HIPCHK(hipEventSynchronize(g->ev_stage[slot]));        // the DMA that last read this buffer is done
            memcpy(g->h_stage[slot].p, reinterpret_cast<const unsigned char*>(queries) + off, n);
            HIPCHK(hipMemcpyAsync(static_cast<unsigned char*>(g->d_q.p) + off, g->h_stage[slot].p, n, hipMemcpyHostToDevice, g->stream));
            HIPCHK(hipEventRecord(g->ev_stage[slot], g->stream));
            off += n; slot ^= 1;
        }
    }
    if (trace) { (void)hipStreamSynchronize(g->stream); fprintf(stderr, "qv: graph search upload %.3f ms (%zu bytes)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), qbytes); }
    const auto t_p1 = std::chrono::steady_clock::now();
    // pass 1: wave-resident traversal (list in registers, rows streamed through LDS); queries that meet equal distances / NaN
    // or outgrow the visited table report 0xFFFFFFFE.  pass 2: the exact-heap kernel for those (heap pop order under ties depends on
    // the heap layout), from a work list compacted ON THE DEVICE (as the construction does): no host round trip between the passes.
    // QV_HNSW_OVERLAP_REDO=1 (a measurement, not the default): large batches in two halves, each pass 1 at eight wave slots per CU — the
    // traversal rate is flat from 8 to 16 (profiles/r03_hnsw_heap.txt) and that leaves the LDS an exact-heap workgroup needs — so that the
    // first half's pass 2 (a handful of queries, ~5 ms of pure latency) can run on a second stream beside the second half's pass 1.
    // Measured at 8192 queries, efSearch 128 / 512: 32.9 / 119.1 ms against 31.5 / 119.4 in one part: nothing gained.
    static const int overlap_env = getenv("QV_HNSW_OVERLAP_REDO") ? atoi(getenv("QV_HNSW_OVERLAP_REDO")) : 2;
    const uint32_t parts = nq >= 4096 && overlap_env == 1 ? 2u : 1u;
    const uint32_t part_n = (nq + parts - 1) / parts;
    const size_t part_qblk = (qv::hnsw_qblk_bytes(part_n, idx->dim4) + 255) / 256 * 256;
    if ((rc = g->d_qblk.ensure(part_qblk * parts)) || (rc = g->s_redo.ensure((size_t)nq * 4)) || (rc = g->s_counters.ensure(64)) || (rc = g->h_counters.ensure(64))) return rc;
    if (parts > 1) {                                                    // the two-part mode alone needs a second stream and its events
        if (!g->stream2) HIPCHK(hipStreamCreateWithFlags(&g->stream2, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) if (!g->ev_part[i]) HIPCHK(hipEventCreateWithFlags(&g->ev_part[i], hipEventDisableTiming));
        if (!g->ev_heap) HIPCHK(hipEventCreateWithFlags(&g->ev_heap, hipEventDisableTiming));
    }
    hipStream_t heap_stream = parts > 1 ? g->stream2 : g->stream;
    uint32_t* counters = static_cast<uint32_t*>(g->s_counters.p);
    HIPCHK(hipMemsetAsync(counters, 0, 64, g->stream));
    hipError_t e = hipSuccess;
    for (uint32_t pi = 0; pi < parts && e == hipSuccess; pi++) {
        const uint32_t q0 = pi * part_n, n_c = std::min(part_n, nq - q0);
        if (n_c == 0) break;
        const float* dq = static_cast<const float*>(g->d_q.p) + (size_t)q0 * idx->dim;
        void* qblk = static_cast<unsigned char*>(g->d_qblk.p) + part_qblk * pi;
        uint32_t* d_rows = static_cast<uint32_t*>(g->d_rows.p) + (size_t)q0 * k; float* d_dist = static_cast<float*>(g->d_dist.p) + (size_t)q0 * k;
        uint32_t* d_cnt = static_cast<uint32_t*>(g->d_cnt.p) + q0; uint32_t* d_ev = static_cast<uint32_t*>(g->d_ev.p) + q0;
        const uint32_t pgrid = std::min(parts > 1 ? std::min(g->grid, (uint32_t)idx->cus * 8u) : g->grid, n_c);
        e = qv::launch_hnsw_search_wave(idx->view(), g->g, dq, qblk, n_c, k, ef_search, wave_opts(g), pgrid, d_rows, d_dist, d_cnt, d_ev, g->stream);
        if (e == hipSuccess) e = qv::launch_build_compact_redo(d_cnt, n_c, static_cast<uint32_t*>(g->s_redo.p) + q0, counters + 4 * pi, g->stream);
        if (e != hipSuccess) break;
        if (parts > 1) {
            HIPCHK(hipEventRecord(g->ev_part[pi], g->stream));
            HIPCHK(hipStreamWaitEvent(g->stream2, g->ev_part[pi], 0));
        }
        qv::HnswOpts ho = heap_opts(g);
        ho.redo_idx = static_cast<const uint32_t*>(g->s_redo.p) + q0; ho.redo_n = counters + 4 * pi;
        const uint32_t hgrid = std::min(std::min(qv::hnsw_grid(idx->cus, efx, n_c), g->vis_bits_slots), n_c);
        e = qv::launch_hnsw_search(idx->view(), g->g, dq, qblk, n_c, k, ef_search, ho, hgrid, false, d_rows, d_dist, d_cnt, d_ev, heap_stream);
    }
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "hnsw search launch failed: %s", hipGetErrorString(e));
    if (parts > 1) {
        HIPCHK(hipEventRecord(g->ev_heap, g->stream2));
        HIPCHK(hipStreamWaitEvent(g->stream, g->ev_heap, 0));
    }
    uint32_t* hc = static_cast<uint32_t*>(g->h_counters.p);
    HIPCHK(hipMemcpyAsync(rows_out, g->d_rows.p, obytes, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipMemcpyAsync(dist_out, g->d_dist.p, obytes, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipMemcpyAsync(count_out, g->d_cnt.p, cbytes, hipMemcpyDeviceToHost, g->stream));
    if (evals_out) HIPCHK(hipMemcpyAsync(evals_out, g->d_ev.p, cbytes, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipMemcpyAsync(hc, counters, 32, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    HIPCHK(hipEventRecord(g->ev_last, g->stream));
    g->tie_reruns += (uint64_t)hc[0] + hc[4];
    if (trace) fprintf(stderr, "qv: graph search passes 1 + 2 + download %.3f ms (%u + %u flagged queries redone on the device, %u part%s)\n",
                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_p1).count(), hc[0], hc[4], parts, parts > 1 ? "s" : "");
    for (uint32_t q = 0; q < nq; q++)
        if (count_out[q] == 0xFFFFFFFEu) return fail(QV_ERR_DEVICE, "hnsw search: a flagged query was not redone");
    return QV_OK;
}

int qv_graph_search_device(qv_graph* g, const float* d_queries, uint32_t nq, uint32_t k, uint32_t ef_search,
                           uint32_t* d_rows_out, float* d_dist_out, uint32_t* d_count_out, uint32_t* d_evals_out, void* stream) {
    if (!g) return fail(QV_ERR_INVALID_ARG, "graph is null");
    if (nq == 0) return QV_OK;
    if (!d_queries || !d_rows_out || !d_dist_out || !d_count_out) return fail(QV_ERR_INVALID_ARG, "null device pointer");
    if (k == 0) return fail(QV_ERR_K_NOT_POSITIVE, "k must be positive");                      // hnsw.go:610-612
    if (k > 512 || ef_search > 512) return fail(QV_ERR_UNSUPPORTED, "k and efSearch above 512 are not supported on the device path");
    if (g->g.n_nodes == 0) return fail(QV_ERR_INVALID_ARG, "graph is empty");
    qv_index* idx = g->idx;
    HIPCHK(hipSetDevice(idx->device));
    std::lock_guard<std::mutex> lock(g->mu);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : g->stream;
    int rc;
    if ((rc = ensure_visited(g, std::max(ef_search, k)))) return rc;
    // the visited tables and the converted-query workspace belong to one traversal at a time: order this one after the last
    HIPCHK(hipStreamWaitEvent(s, g->ev_last, 0));
    if (qv::hnsw_qblk_bytes(nq, idx->dim4) > g->d_qblk.cap) HIPCHK(hipEventSynchronize(g->ev_last));
    if ((rc = g->d_qblk.ensure(qv::hnsw_qblk_bytes(nq, idx->dim4)))) return rc;
    const uint32_t grid = std::min(g->grid, nq);
    hipError_t e = qv::launch_hnsw_search_wave(idx->view(), g->g, d_queries, g->d_qblk.p, nq, k, ef_search, wave_opts(g), grid,
                                               d_rows_out, d_dist_out, d_count_out, d_evals_out, s);
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "hnsw search launch failed: %s", hipGetErrorString(e));
    HIPCHK(hipEventRecord(g->ev_last, s));
    return QV_OK;
}

// ---- construction ----------------------------------------------------------------------------------------------------

uint32_t qv_graph_batch_size(uint32_t n_done, uint32_t batch_max, uint32_t ramp_div) {
    if (n_done == 0 || batch_max <= 1) return 1;                       // the first node has nothing to search (hnsw.go:306-311)
    if (ramp_div == 0) return batch_max;
    return std::min(batch_max, std::max(1u, n_done / ramp_div));
}

int qv_graph_create_empty(qv_graph** out, qv_index* idx, uint32_t capacity_nodes, uint32_t m, uint32_t max_m0, uint32_t ef_construction) {
    if (!out) return fail(QV_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (!idx) return fail(QV_ERR_INVALID_ARG, "index is null");
    if (m == 0) m = 16;                                                // NewHNSW defaults, hnsw.go:223-231
    if (max_m0 == 0) max_m0 = 2 * m;
    if (ef_construction == 0) ef_construction = 200;
    if (max_m0 > 64 || m > 64) return fail(QV_ERR_UNSUPPORTED, "degree bounds above 64 are not supported (MaxM0=%u, M=%u)", max_m0, m);
    if (ef_construction > 512) return fail(QV_ERR_UNSUPPORTED, "efConstruction above 512 is not supported on the device path");
    if (!idx->d_rowmaj && idx->n_rows) return fail(QV_ERR_UNSUPPORTED, "device-side construction needs the row-major copy (create the index with QV_FLAG_ROWMAJOR)");
    if (!(idx->flags & QV_FLAG_ROWMAJOR)) return fail(QV_ERR_UNSUPPORTED, "device-side construction needs the row-major copy (create the index with QV_FLAG_ROWMAJOR)");
    HIPCHK(hipSetDevice(idx->device));
    qv_graph* g = new (std::nothrow) qv_graph();
    if (!g) return fail(QV_ERR_OOM, "out of host memory");
    g->idx = idx; g->buildable = true; g->efc = ef_construction;
    g->g.max_m0 = max_m0; g->g.max_m = m; g->g.n_nodes = 0; g->g.entry = 0; g->g.cur_level = -1; g->g.has_dead = false;
    int rc = graph_common_init(g);
    if (rc == QV_OK) rc = graph_reserve(g, std::max(capacity_nodes, 64u), std::max(capacity_nodes / 3 + capacity_nodes / 64, 64u));
    if (rc != QV_OK) { qv_graph_destroy(g); return rc; }
    *out = g;
    return QV_OK;
}

int qv_graph_make_buildable(qv_graph* g, uint32_t ef_construction) {
    if (!g) return fail(QV_ERR_INVALID_ARG, "graph is null");
    if (ef_construction == 0) ef_construction = 200;
    if (ef_construction > 512) return fail(QV_ERR_UNSUPPORTED, "efConstruction above 512 is not supported on the device path");
    qv_index* idx = g->idx;
    if (!idx->d_rowmaj) return fail(QV_ERR_UNSUPPORTED, "device-side construction needs the row-major copy (create the index with QV_FLAG_ROWMAJOR)");
    HIPCHK(hipSetDevice(idx->device));
    std::lock_guard<std::mutex> lock(g->mu);
    g->efc = ef_construction;
    if (g->buildable) return QV_OK;
    HIPCHK(hipEventSynchronize(g->ev_last));
    const uint32_t m0 = g->g.max_m0, m = g->g.max_m, n = g->g.n_nodes;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&g->d_l0dist), std::max<size_t>((size_t)g->cap_nodes * m0 * 4, 16));
    if (e == hipSuccess) e = hipMemset(g->d_l0dist, 0, std::max<size_t>((size_t)g->cap_nodes * m0 * 4, 16));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g->d_updist), std::max<size_t>((size_t)g->cap_blocks * m * 4, 16));
    if (e == hipSuccess) e = hipMemset(g->d_updist, 0, std::max<size_t>((size_t)g->cap_blocks * m * 4, 16));
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) { (void)hipFree(g->d_l0dist); (void)hipFree(g->d_updist); g->d_l0dist = g->d_updist = nullptr;
                           return fail(e == hipErrorOutOfMemory ? QV_ERR_OOM : QV_ERR_DEVICE, "link-distance storage failed: %s", hipGetErrorString(e)); }
    // score every existing link once, a chunk of nodes at a time (their vectors are the queries)
    const uint32_t chunk = 16384;
    int rc = g->d_qblk.ensure(qv::hnsw_qblk_bytes(std::min(chunk, std::max(n, 1u)), idx->dim4));
    if (rc != QV_OK) return rc;
    const uint32_t grid = qv::hnsw_wave_grid(idx->cus, idx->metric, idx->dim4);
    for (uint32_t n0 = 0; n0 < n; n0 += chunk) {
        e = qv::launch_graph_link_dists(idx->view(), g->g, g->d_qblk.p, n0, std::min(chunk, n - n0), g->d_l0dist, g->d_updist, grid, g->stream);
        if (e != hipSuccess) return fail(QV_ERR_DEVICE, "link-distance launch failed: %s", hipGetErrorString(e));
    }
    HIPCHK(hipStreamSynchronize(g->stream));
    g->buildable = true;
    return QV_OK;
}

int qv_graph_insert(qv_graph* g, uint32_t first_row, uint32_t n, const int8_t* levels, uint32_t batch_max, uint32_t ramp_div) {
    if (!g) return fail(QV_ERR_INVALID_ARG, "graph is null");
    if (n == 0) return QV_OK;
    if (!levels) return fail(QV_ERR_INVALID_ARG, "levels is null");
    if (!g->buildable) return fail(QV_ERR_UNSUPPORTED, "this graph was uploaded without link distances (qv_graph_create): call qv_graph_make_buildable first");
    qv_index* idx = g->idx;
    if (first_row != g->g.n_nodes) return fail(QV_ERR_INVALID_ARG, "nodes are appended: expected first row %u, got %u", g->g.n_nodes, first_row);
    if ((uint64_t)first_row + n > idx->n_rows) return fail(QV_ERR_OUT_OF_RANGE, "rows %u..%llu are not in the index (rows: %u)", first_row, (unsigned long long)first_row + n, idx->n_rows);
    if (!idx->d_rowmaj) return fail(QV_ERR_UNSUPPORTED, "device-side construction needs the row-major copy (QV_FLAG_ROWMAJOR)");
    uint64_t new_blocks = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (levels[i] < 0 || levels[i] > 63) return fail(QV_ERR_INVALID_ARG, "level %d of row %u out of range", (int)levels[i], first_row + i);
        new_blocks += (uint64_t)levels[i];
    }
    if (batch_max == 0) batch_max = 16384;                              // measured (profiles/r02_hnsw_build_sweeps.txt): 1M x 768 in 5.9 s vs 7.1 s at 4096, same recall
    batch_max = std::min(batch_max, 16384u);
    HIPCHK(hipSetDevice(idx->device));
    std::lock_guard<std::mutex> lock(g->mu);
    HIPCHK(hipEventSynchronize(g->ev_last));                            // no traversal in flight while the graph and the shared buffers change
    const auto t0 = std::chrono::steady_clock::now();
    int rc = graph_reserve(g, (uint64_t)first_row + n, (uint64_t)g->n_blocks + new_blocks);
    if (rc != QV_OK) return rc;
    const uint32_t m0 = g->g.max_m0;
    hipStream_t s = g->stream;
    // levels and upper-level block offsets of the new nodes
    {
        std::vector<uint32_t> upoff(n);
        uint32_t blk = g->n_blocks;
        for (uint32_t i = 0; i < n; i++) { upoff[i] = blk; blk += (uint32_t)levels[i]; }
        HIPCHK(hipMemcpy(g->d_level + first_row, levels, n, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(g->d_upoff + first_row, upoff.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        g->n_blocks = blk;
        g->h_level.insert(g->h_level.end(), levels, levels + n);
    }
    const uint32_t bmax = std::min(batch_max, n);
    const size_t nk = (size_t)bmax * m0;
    if ((rc = g->d_rows.ensure(nk * 4)) || (rc = g->d_dist.ensure(nk * 4)) || (rc = g->d_cnt.ensure((size_t)bmax * 4)) || (rc = g->d_ev.ensure((size_t)bmax * 4)) ||
        (rc = g->b_self.ensure((size_t)bmax * 4)) || (rc = g->b_keys_a.ensure(nk * 8)) || (rc = g->b_keys_b.ensure(nk * 8)) ||
        (rc = g->b_hist.ensure(qv::radix_hist_words((uint32_t)nk) * 4)) || (rc = g->b_seg.ensure(nk * 4)) || (rc = g->b_redo.ensure((size_t)bmax * 4)) ||
        (rc = g->b_counters.ensure(64)) || (rc = g->d_qblk.ensure(qv::hnsw_qblk_bytes(bmax, idx->dim4))) || (rc = ensure_visited(g, std::max(g->efc, m0))))
        return rc;
    uint32_t* counters = static_cast<uint32_t*>(g->b_counters.p);
    HIPCHK(hipStreamWaitEvent(s, g->ev_last, 0));
    HIPCHK(hipMemsetAsync(counters, 0, 64, s));
    const uint32_t hslots = std::min(qv::hnsw_grid(idx->cus, g->efc, 0xFFFFFFFFu), g->vis_bits_slots);
    uint32_t done = 0;
    while (done < n) {
        const uint32_t at = first_row + done;
        if (at == 0) {                                                  // first node: entry point, nothing to connect (hnsw.go:306-311)
            g->g.n_nodes = 1; g->g.entry = 0; g->g.cur_level = levels[0];
            done = 1; g->build_batches++;
            continue;
        }
        const uint32_t B = std::min(std::min(qv_graph_batch_size(at, batch_max, ramp_div), bmax), n - done);
        // search phase against the graph as it stands (n_nodes = at)
        g->g.n_nodes = at;
        qv::HnswOpts wo = wave_opts(g);
        wo.qlevel = g->d_level + at; wo.qnode0 = at; wo.self_dist = static_cast<float*>(g->b_self.p);
        const float* d_queries = idx->d_rowmaj + (size_t)at * idx->dim;
        HIPCHK(hipMemsetAsync(counters, 0, 8, s));
        hipError_t e = qv::launch_hnsw_search_wave(idx->view(), g->g, d_queries, g->d_qblk.p, B, m0, g->efc, wo, std::min(g->grid, B),
                                                   static_cast<uint32_t*>(g->d_rows.p), static_cast<float*>(g->d_dist.p), static_cast<uint32_t*>(g->d_cnt.p),
                                                   static_cast<uint32_t*>(g->d_ev.p), s);
        if (e == hipSuccess) e = qv::launch_build_compact_redo(static_cast<const uint32_t*>(g->d_cnt.p), B, static_cast<uint32_t*>(g->b_redo.p), counters, s);
        if (e == hipSuccess) {
            qv::HnswOpts ho = heap_opts(g);
            ho.qlevel = wo.qlevel; ho.qnode0 = at; ho.self_dist = wo.self_dist;
            ho.redo_idx = static_cast<const uint32_t*>(g->b_redo.p); ho.redo_n = counters;
            e = qv::launch_hnsw_search(idx->view(), g->g, d_queries, g->d_qblk.p, B, m0, g->efc, ho, std::min(hslots, B), false,
                                       static_cast<uint32_t*>(g->d_rows.p), static_cast<float*>(g->d_dist.p), static_cast<uint32_t*>(g->d_cnt.p),
                                       static_cast<uint32_t*>(g->d_ev.p), s);
        }
        // link phase
        if (e == hipSuccess)
            e = qv::launch_build_links(g->bview(), at, B, g->g.cur_level, static_cast<const uint32_t*>(g->d_rows.p), static_cast<const float*>(g->d_dist.p),
                                       static_cast<const uint32_t*>(g->d_cnt.p), static_cast<const float*>(g->b_self.p), static_cast<uint64_t*>(g->b_keys_a.p),
                                       static_cast<uint64_t*>(g->b_keys_b.p), static_cast<uint32_t*>(g->b_hist.p), static_cast<uint32_t*>(g->b_seg.p), counters,
                                       (uint32_t)idx->cus * 16, s);
        if (e != hipSuccess) return fail(QV_ERR_DEVICE, "graph build launch failed: %s", hipGetErrorString(e));
        // entry-point update, node by node (hnsw.go:325-332: level > oldCurrentLevel, then > CurrentLevel as it stands)
        const int snap = g->g.cur_level;
        for (uint32_t i = 0; i < B; i++) {
            const int lv = levels[done + i];
            if (lv > snap && lv > g->g.cur_level) { g->g.entry = at + i; g->g.cur_level = lv; }
        }
        done += B; g->build_batches++;
        g->g.n_nodes = at + B;
    }
    HIPCHK(hipEventRecord(g->ev_last, s));
    if ((rc = g->h_counters.ensure(64))) return rc;
    uint32_t* hc = static_cast<uint32_t*>(g->h_counters.p);
    HIPCHK(hipMemcpyAsync(hc, counters, 16, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    g->build_redo += hc[3];
    g->build_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (hc[2] & 1u) return fail(QV_ERR_UNSUPPORTED, "graph build: a construction search overflowed the device candidate heap; the graph is incomplete");
    if (hc[2] & 2u) return fail(QV_ERR_DEVICE, "graph build: a flagged construction search was not redone; the graph is incomplete");
    return QV_OK;
}

int qv_graph_build(qv_graph** out, qv_index* idx, uint32_t n_nodes, const int8_t* levels, uint32_t m, uint32_t max_m0,
                   uint32_t ef_construction, uint32_t batch_max, uint32_t ramp_div) {
    if (!out) return fail(QV_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (n_nodes == 0) return fail(QV_ERR_INVALID_ARG, "graph has no nodes");
    qv_graph* g = nullptr;
    int rc = qv_graph_create_empty(&g, idx, n_nodes, m, max_m0, ef_construction);
    if (rc != QV_OK) return rc;
    rc = qv_graph_insert(g, 0, n_nodes, levels, batch_max, ramp_div);
    if (rc != QV_OK) { qv_graph_destroy(g); return rc; }
    *out = g;
    return QV_OK;
}

int qv_graph_info(const qv_graph* g, uint32_t* n_nodes, uint32_t* n_up_blocks, uint32_t* max_m0, uint32_t* max_m, uint32_t* entry, int* cur_level) {
    if (!g) return fail(QV_ERR_INVALID_ARG, "graph is null");
    if (n_nodes) *n_nodes = g->g.n_nodes;
    if (n_up_blocks) *n_up_blocks = g->n_blocks;
    if (max_m0) *max_m0 = g->g.max_m0;
    if (max_m) *max_m = g->g.max_m;
    if (entry) *entry = g->g.entry;
    if (cur_level) *cur_level = g->g.cur_level;
    return QV_OK;
}

int qv_graph_stats(const qv_graph* g, double* build_seconds, uint64_t* build_batches, uint64_t* build_redo, uint64_t* search_redo) {
    if (!g) return fail(QV_ERR_INVALID_ARG, "graph is null");
    if (build_seconds) *build_seconds = g->build_seconds;
    if (build_batches) *build_batches = g->build_batches;
    if (build_redo) *build_redo = g->build_redo;
    if (search_redo) *search_redo = g->tie_reruns;
    return QV_OK;
}

int qv_graph_export(qv_graph* g, int8_t* levels, uint32_t* l0_deg, uint32_t* l0_links, uint32_t* up_off, uint32_t* up_links) {
    if (!g) return fail(QV_ERR_INVALID_ARG, "graph is null");
    HIPCHK(hipSetDevice(g->idx->device));
    std::lock_guard<std::mutex> lock(g->mu);
    HIPCHK(hipEventSynchronize(g->ev_last));
    const size_t n = g->g.n_nodes;
    if (levels) HIPCHK(hipMemcpy(levels, g->d_level, n, hipMemcpyDeviceToHost));
    if (l0_deg) HIPCHK(hipMemcpy(l0_deg, g->d_l0deg, n * 4, hipMemcpyDeviceToHost));
    if (l0_links) HIPCHK(hipMemcpy(l0_links, g->d_l0links, n * g->g.max_m0 * 4, hipMemcpyDeviceToHost));
    if (up_off) HIPCHK(hipMemcpy(up_off, g->d_upoff, n * 4, hipMemcpyDeviceToHost));
    if (up_links && g->n_blocks) HIPCHK(hipMemcpy(up_links, g->d_uplinks, (size_t)g->n_blocks * (1 + g->g.max_m) * 4, hipMemcpyDeviceToHost));
    return QV_OK;
}

}  // extern "C"
