// qv_graph_api.cpp — the HNSW part of the C ABI (include/qv.h): device-resident traversal of a graph the host built
// (qv_graph_create / qv_graph_search*) and device-resident construction (qv_graph_create_empty / qv_graph_insert /
// qv_graph_build / qv_graph_export).  Host-side responsibilities only: argument checks, device-memory ownership, the
// batch schedule of a build and the entry-point bookkeeping the reference does under its lock (hnsw.go:325-332).
// No CPU compute path: every distance and every selection runs in a HIP kernel (qv_hnsw.hip, qv_build.hip).
#include "qv_api_internal.h"

#include <atomic>
#include <functional>
#include <shared_mutex>

// Everything ONE host-pointer traversal call owns while it runs: stream, converted-query workspace, result buffers, its redo
// list and its visited sets.  The reference searches under a read lock (hnsw.go:602-606; a goroutine per query, adapter.go:253-279),
// so many qv_graph_search calls are in flight at once: each takes a context from the graph's pool and no two share a table.
// (Until round 5 the graph had ONE set of these and a mutex: concurrent single-query callers ran one after another, ~300 QPS.)
struct GraphCtx {
    hipStream_t stream = nullptr;
    Buf vis_hash; uint32_t vis_hash_cap = 0, vis_hash_slots = 0;      // wave kernel: a hash table per wave slot
    Buf vis_bits; uint32_t vis_bits_words = 0, vis_bits_slots = 0;    // exact-heap kernel: a bitmap per slot
    Buf d_q, d_qblk, d_rows, d_dist, d_cnt, d_ev, s_redo, s_counters, hub_table;
    PinBuf h_stage[2]; hipEvent_t ev_stage[2] = {nullptr, nullptr};
    PinBuf h_counters, h_out;
    hipEvent_t ev_block = nullptr;       // blocking-sync event: a thread that runs a shared batch sleeps until the device is done instead of spinning
    void release() {
        if (stream) (void)hipStreamSynchronize(stream);
        vis_hash.release(); vis_bits.release();
        d_q.release(); d_qblk.release(); d_rows.release(); d_dist.release(); d_cnt.release(); d_ev.release(); s_redo.release(); s_counters.release(); hub_table.release();
        for (int i = 0; i < 2; i++) { h_stage[i].release(); if (ev_stage[i]) (void)hipEventDestroy(ev_stage[i]); ev_stage[i] = nullptr; }
        h_counters.release(); h_out.release();
        if (ev_block) (void)hipEventDestroy(ev_block);
        ev_block = nullptr;
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
    }
};

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
    std::atomic<uint64_t> tie_reruns{0};
    std::mutex mu;                              // the graph's OWN buffers below (device-form traversals, construction, export): one user at a time
    // host-pointer searches: a context each (pool), and a front that lets concurrent small calls share a traversal batch
    // (qv_coalesce.h).  A traversal is a chain of hops, so a lone batch does not fill the chip and a second caller should not wait for the
    // first: two batches in flight, of at most 256 queries each — one query per CU, the batches the latency form of the wave kernel
    // serves (qv_hnsw.hip: 1.1 - 1.7 ms per batch against 4.5 - 6 for the wave-per-query form).  Measured on 1M x 768, efSearch 128,
    // QPS at 8 / 64 / 256 / 1024 callers (tools/dev_graph_lanes.sh): 1 lane x 4096 5.2 k / 36 k / 95 k / 87 k; 2 x 4096 5.6 k / 36 k /
    // 124 k / 92 k; 2 x 256 5.6 k / 34 k / 119 k / 158 k; 2 x 128 5.6 k / 35 k / 128 k / 145 k; 4 x 256 5.6 k / 34 k / 80 k / 152 k;
    // 4 x 128 5.7 k / 35 k / 96 k / 190 k (p99 at 256 callers 31 ms against 3 ms with two lanes); 8 x 128 4.3 k / 21 k / 64 k / 174 k.
    std::mutex ctx_mu;
    std::vector<GraphCtx*> free_ctx, all_ctx;
    static int lanes() { static const int n = getenv("QV_GRAPH_LANES") ? std::max(1, atoi(getenv("QV_GRAPH_LANES"))) : 2; return n; }   // (measurement switch, read once)
    static uint32_t max_group() { static const uint32_t n = getenv("QV_GRAPH_MAX_GROUP") ? (uint32_t)std::max(1, atoi(getenv("QV_GRAPH_MAX_GROUP"))) : 256u; return n; }
    qvco::Front front{lanes(), max_group()};
    // The hubs (qv_hnsw.hip "hubs"): the rows most traversals read, chosen by a sampling pass of the first large call, copied into
    // tiles of their own; a large call computes its queries' distances to all of them up front (k_hub_table) and its hops look them
    // up.  Valid while the graph has the nodes and the index the row contents they were taken from.
    struct Hubs {
        std::shared_mutex mu;                   // calls that use the hubs hold it shared while they enqueue; a (re)build holds it alone
        uint32_t H = 0;
        uint32_t nodes_at = 0xFFFFFFFFu; uint64_t writes_at = ~0ull;   // the state they were sampled in (H == 0: sampled, nothing worth it)
        Buf tiles, rnorm, rows, l0_hub, slot_of, hist, table;          // table: the device form's (host-form contexts have their own)
    } hubs;
    hipStream_t stream = nullptr;
    hipEvent_t ev_last = nullptr;               // end of the most recent traversal: the next one (on any stream) waits for it
    Buf d_qblk, d_rows, d_dist, d_cnt, d_ev;
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
    // wave slots of the traversal kernel: a constant of (device, metric, dimension), fixed here once — concurrent qv_graph_search
    // callers only ever read it
    g->grid = qv::hnsw_wave_grid(g->idx->cus, g->idx->metric, g->idx->dim4);
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
    for (GraphCtx* c : g->all_ctx) { c->release(); delete c; }
    if (g->stream) { (void)hipStreamSynchronize(g->stream); (void)hipStreamDestroy(g->stream); }
    (void)hipFree(g->d_level); (void)hipFree(g->d_l0deg); (void)hipFree(g->d_l0links); (void)hipFree(g->d_upoff); (void)hipFree(g->d_uplinks);
    (void)hipFree(g->d_l0dist); (void)hipFree(g->d_updist);
    g->vis_hash.release(); g->vis_bits.release();
    g->d_qblk.release(); g->d_rows.release(); g->d_dist.release(); g->d_cnt.release(); g->d_ev.release();
    g->h_counters.release();
    { auto& hb = g->hubs; hb.tiles.release(); hb.rnorm.release(); hb.rows.release(); hb.l0_hub.release(); hb.slot_of.release(); hb.hist.release(); hb.table.release(); }
    g->b_self.release(); g->b_keys_a.release(); g->b_keys_b.release(); g->b_hist.release(); g->b_seg.release(); g->b_redo.release(); g->b_counters.release();
    delete g;
}

}  // extern "C"

namespace {

int acquire_gctx(qv_graph* g, GraphCtx** out) {
    {
        std::lock_guard<std::mutex> l(g->ctx_mu);
        if (!g->free_ctx.empty()) { *out = g->free_ctx.back(); g->free_ctx.pop_back(); return QV_OK; }
    }
    GraphCtx* c = new (std::nothrow) GraphCtx();
    if (!c) return fail(QV_ERR_OOM, "out of host memory");
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(QV_ERR_DEVICE, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
    std::lock_guard<std::mutex> l(g->ctx_mu);
    g->all_ctx.push_back(c);
    *out = c;
    return QV_OK;
}
// ---- the wave traversal of one call (with the graph's hubs in the measurement build) ---------------------------------------------------
constexpr uint32_t kHubMinQueries = 2048;      // calls below this never sample, build or use hubs (the dense pass costs ~0.4 us per query and 64 hubs)
constexpr uint32_t kHubSample = 1024;          // queries of the sampling pass
constexpr uint32_t kHubMax = 16384;            // hubs at most (a 16-bit slot per link)
bool hubs_possible(const qv_graph* g, uint32_t nq) {
    // measured and NOT shipped (qv_hnsw.hip "hubs": the benchmark's corpora have no hubs to speak of): the measurement build with
    // QV_HNSW_HUBS=1 samples, builds and uses them; the product library never does
#ifdef QV_VARIANTS
    static const bool on = getenv("QV_HNSW_HUBS") && atoi(getenv("QV_HNSW_HUBS")) == 1;
#else
    constexpr bool on = false;
#endif
    if (!on) return false;
    const qv_index* idx = g->idx;
    return nq >= kHubMinQueries && idx->d_rowmaj && (idx->dim & 31u) == 0 && idx->dim >= 32 && !g->g.has_dead && g->g.max_m0 <= 32u && g->g.n_nodes >= 4096u;
}
// sample the call's first queries, choose the hubs, build their copies (synchronous: once per graph state).  Holds hubs.mu alone.
int hubs_build(qv_graph* g, const float* dq, void* qblk, uint32_t nq, uint32_t k, uint32_t ef, qv::HnswOpts wo, uint32_t grid,
               uint32_t* d_rows, float* d_dist, uint32_t* d_cnt, uint32_t* d_ev, hipStream_t s) {
    qv_index* idx = g->idx;
    auto& hb = g->hubs;
    HIPCHK(hipDeviceSynchronize());                                      // nothing in flight reads the arrays that are about to be replaced
    hb.H = 0; hb.nodes_at = g->g.n_nodes; hb.writes_at = idx->row_writes;
    const uint32_t n = g->g.n_nodes, ns = std::min(nq, kHubSample);
    int rc;
    if ((rc = hb.hist.ensure((size_t)n * 4))) return rc;
    HIPCHK(hipMemsetAsync(hb.hist.p, 0, (size_t)n * 4, s));
    wo.hist = static_cast<uint32_t*>(hb.hist.p);
    hipError_t e = qv::launch_hnsw_search_wave(idx->view(), g->g, dq, qblk, ns, k, ef, wo, std::min(grid, ns), d_rows, d_dist, d_cnt, d_ev, s);
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "hub sampling launch failed: %s", hipGetErrorString(e));
    std::vector<uint32_t> h(n);
    HIPCHK(hipMemcpyAsync(h.data(), hb.hist.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    // a hub is worth its column of the table when more than ~one query in twelve reads it: 2 D flops per (query, hub) at the dense
    // pass's ~30 TFLOP/s against 4 D bytes per read at the gather's ~5 TB/s
    const uint32_t thr = std::max(2u, ns / 12u);
    std::vector<std::pair<uint32_t, uint32_t>> cand;                     // (reads, row)
    for (uint32_t i = 0; i < n; i++) if (h[i] >= thr) cand.emplace_back(h[i], i);
    std::sort(cand.begin(), cand.end(), [](const auto& a, const auto& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
    uint32_t H = (uint32_t)std::min<size_t>(cand.size(), kHubMax) / 64u * 64u;
    static const bool trace0 = getenv("QV_TRACE") && atoi(getenv("QV_TRACE")) > 0;
    if (trace0) { std::vector<uint32_t> srt(h); std::sort(srt.begin(), srt.end(), std::greater<uint32_t>()); uint64_t tot = 0, cum = 0; for (uint32_t x : srt) tot += x;
                  for (uint32_t i = 0; i < n; i++) { cum += srt[i]; if (i + 1 == 64 || i + 1 == 1024 || i + 1 == 8192 || i + 1 == 65536 || i + 1 == 262144)
                      fprintf(stderr, "qv: graph hubs: the %u most-read rows take %.1f %% of the sample's reads (rank %u is read by %u of %u queries)\n", i + 1, 100.0 * (double)cum / (double)tot, i + 1, srt[i], ns); } }
    if (trace0 && H < 64) { uint64_t tot = 0; uint32_t mx = 0; for (uint32_t i = 0; i < n; i++) { tot += h[i]; mx = std::max(mx, h[i]); }
                            fprintf(stderr, "qv: graph hubs: none (%zu rows read by >= %u of %u sampled queries; %llu reads counted, the most-read row %u times)\n", cand.size(), thr, ns, (unsigned long long)tot, mx); }
    if (H < 64) return QV_OK;                                            // nothing worth it on this graph and these queries
    std::vector<uint32_t> rows(H);
    for (uint32_t i = 0; i < H; i++) rows[i] = cand[i].second;
    if ((rc = hb.rows.ensure((size_t)H * 4)) || (rc = hb.tiles.ensure((size_t)(H / 64) * idx->tile_bytes())) || (rc = hb.rnorm.ensure((size_t)H * 8)) ||
        (rc = hb.slot_of.ensure((size_t)n * 2)) || (rc = hb.l0_hub.ensure((size_t)n * g->g.max_m0 * 2)))
        return rc;
    HIPCHK(hipMemcpyAsync(hb.rows.p, rows.data(), (size_t)H * 4, hipMemcpyHostToDevice, s));
    e = qv::launch_hub_build(idx->view(), g->g, static_cast<const uint32_t*>(hb.rows.p), H, static_cast<float*>(hb.tiles.p), static_cast<double*>(hb.rnorm.p),
                             static_cast<uint16_t*>(hb.slot_of.p), static_cast<uint16_t*>(hb.l0_hub.p), s);
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "hub build launch failed: %s", hipGetErrorString(e));
    HIPCHK(hipStreamSynchronize(s));                                     // (`rows` leaves scope)
    hb.H = H;
    static const bool trace = getenv("QV_TRACE") && atoi(getenv("QV_TRACE")) > 0;
    if (trace) fprintf(stderr, "qv: graph hubs: %u rows read by >= %u of %u sampled queries (%zu candidates), %.1f MB of copies\n", H, thr, ns, cand.size(), (double)H * idx->dim * 4 / 1e6);
    return QV_OK;
}
// One call's wave traversal.  table: where this caller keeps its hub table (grown here).
int traverse_wave(qv_graph* g, const float* dq, void* qblk, Buf* table, uint32_t nq, uint32_t k, uint32_t ef, qv::HnswOpts wo, uint32_t grid,
                  uint32_t* d_rows, float* d_dist, uint32_t* d_cnt, uint32_t* d_ev, hipStream_t s) {
    qv_index* idx = g->idx;
    auto& hb = g->hubs;
    if (hubs_possible(g, nq)) {
        {
            std::shared_lock<std::shared_mutex> rl(hb.mu);
            const bool current = hb.nodes_at == g->g.n_nodes && hb.writes_at == idx->row_writes;
            if (!current) {
                rl.unlock();
                std::unique_lock<std::shared_mutex> wl(hb.mu);
                if (!(hb.nodes_at == g->g.n_nodes && hb.writes_at == idx->row_writes)) {
                    const int rc = hubs_build(g, dq, qblk, nq, k, ef, wo, grid, d_rows, d_dist, d_cnt, d_ev, s);
                    if (rc != QV_OK) return rc;
                }
            }
        }
        std::shared_lock<std::shared_mutex> rl(hb.mu);
        if (hb.H && hb.nodes_at == g->g.n_nodes && hb.writes_at == idx->row_writes) {
            // the table of the whole call where it stays under 2 GiB, else in pieces of that size (each piece a launch of its own)
            const uint32_t piece = (uint32_t)std::max<uint64_t>(1024, std::min<uint64_t>(nq, ((uint64_t)2 << 30) / ((uint64_t)hb.H * 4)));
            int rc = table->ensure((size_t)std::min(nq, piece) * hb.H * 4);
            if (rc != QV_OK) return rc;
            wo.l0_hub = static_cast<const uint16_t*>(hb.l0_hub.p); wo.hub_H = hb.H; wo.hub_tab = static_cast<const float*>(table->p);
            for (uint32_t q0 = 0; q0 < nq; q0 += piece) {
                const uint32_t m = std::min(piece, nq - q0);
                const float* dqm = dq + (size_t)q0 * idx->dim;
                hipError_t e = qv::launch_hub_table(idx->view(), static_cast<const float*>(hb.tiles.p), static_cast<const double*>(hb.rnorm.p), hb.H, dqm, qblk, m,
                                                    static_cast<float*>(table->p), idx->cus, s);
                if (e == hipSuccess) e = qv::launch_hnsw_search_wave(idx->view(), g->g, dqm, qblk, m, k, ef, wo, std::min(grid, m), d_rows + (size_t)q0 * k, d_dist + (size_t)q0 * k,
                                                                     d_cnt + q0, d_ev ? d_ev + q0 : nullptr, s);
                if (e != hipSuccess) return fail(QV_ERR_DEVICE, "hnsw search launch failed: %s", hipGetErrorString(e));
            }
            return QV_OK;
        }
    }
    hipError_t e = qv::launch_hnsw_search_wave(idx->view(), g->g, dq, qblk, nq, k, ef, wo, grid, d_rows, d_dist, d_cnt, d_ev, s);
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "hnsw search launch failed: %s", hipGetErrorString(e));
    return QV_OK;
}

struct GCtxGuard {
    qv_graph* g; GraphCtx* c;
    ~GCtxGuard() { if (c) { std::lock_guard<std::mutex> l(g->ctx_mu); g->free_ctx.push_back(c); } }
};

// visited-set storage of one context for a batch of nq queries with list capacity ef: as many slots as the batch can occupy
int ensure_visited_ctx(qv_graph* g, GraphCtx* c, uint32_t ef, uint32_t nq) {
    qv_index* idx = g->idx;
    const uint32_t cap = qv::hnsw_vis_hash_cap(ef);
    const uint32_t slots = std::min(g->grid, std::max(nq, 1u));
    if (cap > c->vis_hash_cap || slots > c->vis_hash_slots) {
        const uint32_t ncap = std::max(cap, c->vis_hash_cap), nslots = std::max(slots, c->vis_hash_slots);
        HIPCHK(hipStreamSynchronize(c->stream));
        int rc = c->vis_hash.ensure((size_t)nslots * ncap * 4);
        if (rc != QV_OK) return rc;
        c->vis_hash_cap = ncap; c->vis_hash_slots = nslots;
    }
    const uint32_t words = (uint32_t)((((uint64_t)std::max(g->cap_nodes, g->g.n_nodes) + 31) / 32 + 63) / 64 * 64);
    // the exact-heap kernel has few slots (LDS heaps): at most 8 per CU, and never more than 2 GiB of bitmaps
    uint32_t hslots = std::min((uint32_t)idx->cus * 8, std::max(nq, 1u));
    while (hslots > 64 && (size_t)hslots * words * 4 > ((size_t)2 << 30)) hslots /= 2;
    if (words > c->vis_bits_words || hslots > c->vis_bits_slots) {
        const uint32_t nwords = std::max(words, c->vis_bits_words), nslots = std::max(hslots, c->vis_bits_slots);
        HIPCHK(hipStreamSynchronize(c->stream));
        int rc = c->vis_bits.ensure((size_t)nslots * nwords * 4);
        if (rc != QV_OK) return rc;
        c->vis_bits_words = nwords; c->vis_bits_slots = nslots;
    }
    return QV_OK;
}

// One traversal call in a context of its own: upload, wave-resident pass, device-side compaction of the flagged queries, exact-heap
// pass for those, download.  Nothing of the graph's is written; any number of these run side by side.
int graph_search_ctx(qv_graph* g, GraphCtx* c, const float* queries, uint32_t nq, uint32_t k, uint32_t ef_search,
                     uint32_t* rows_out, float* dist_out, uint32_t* count_out, uint32_t* evals_out, const std::function<void()>& early) {
    qv_index* idx = g->idx;
    // A caller on its own waits for the device the way every other entry point does (the runtime spins: lowest latency).  A thread that
    // runs a batch for OTHER callers too does so while hundreds of them want the cores to come back with their next query: it sleeps on a
    // blocking event instead (measured with 1024 callers on 8 usable cores: the spinning leaders of four lanes — the setting then — took half of them).
    const bool blocking = (bool)early;
    if (blocking && !c->ev_block) HIPCHK(hipEventCreateWithFlags(&c->ev_block, hipEventBlockingSync | hipEventDisableTiming));
    auto wait_device = [&]() -> hipError_t {
        if (!blocking) return hipStreamSynchronize(c->stream);
        hipError_t e0 = hipEventRecord(c->ev_block, c->stream);
        return e0 == hipSuccess ? hipEventSynchronize(c->ev_block) : e0;
    };
    const size_t qbytes = (size_t)nq * idx->dim * sizeof(float), obytes = (size_t)nq * k * 4, cbytes = (size_t)nq * 4;
    const uint32_t efx = std::max(ef_search, k);
    int rc;
    if ((rc = c->d_q.ensure(qbytes)) || (rc = c->d_rows.ensure(obytes)) || (rc = c->d_dist.ensure(obytes)) || (rc = c->d_cnt.ensure(cbytes)) ||
        (rc = c->d_ev.ensure(cbytes)) || (rc = c->d_qblk.ensure(qv::hnsw_qblk_bytes(nq, idx->dim4))) || (rc = ensure_visited_ctx(g, c, efx, nq)) ||
        (rc = c->s_redo.ensure((size_t)nq * 4)) || (rc = c->s_counters.ensure(64)) || (rc = c->h_counters.ensure(64)))
        return rc;
    static const bool trace = getenv("QV_TRACE") && atoi(getenv("QV_TRACE")) > 0;
    const auto t_begin = std::chrono::steady_clock::now();
    {   // upload through two pinned bounce buffers: the CPU copy of slice i+1 overlaps the DMA of slice i
        // (a single hipMemcpyAsync from pageable memory ran at 2.5-5 GB/s: a third of a 16k-query batch's time)
        const size_t slice = (size_t)8 << 20;
        size_t off = 0; int slot = 0;
        while (off < qbytes) {
            const size_t n = std::min(slice, qbytes - off);
            if ((rc = c->h_stage[slot].ensure(std::min(slice, std::max(2 * qbytes, (size_t)65536))))) return rc;   // (twice the need: groups grow)
            if (!c->ev_stage[slot]) HIPCHK(hipEventCreateWithFlags(&c->ev_stage[slot], hipEventDisableTiming));
            else HIPCHK(hipEventSynchronize(c->ev_stage[slot]));        // the DMA that last read this buffer is done
            memcpy(c->h_stage[slot].p, reinterpret_cast<const unsigned char*>(queries) + off, n);
            HIPCHK(hipMemcpyAsync(static_cast<unsigned char*>(c->d_q.p) + off, c->h_stage[slot].p, n, hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipEventRecord(c->ev_stage[slot], c->stream));
            off += n; slot ^= 1;
        }
    }
    if (trace) { (void)hipStreamSynchronize(c->stream); fprintf(stderr, "qv: graph search upload %.3f ms (%zu bytes)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), qbytes); }
    const auto t_p1 = std::chrono::steady_clock::now();
    // pass 1: wave-resident traversal (list in registers, rows streamed through LDS); queries that meet equal distances / NaN
    // or outgrow the visited table report 0xFFFFFFFE.  pass 2: the exact-heap kernel for those (heap pop order under ties depends on
    // the heap layout), from a work list compacted ON THE DEVICE (as the construction does): no host round trip between the passes.
    // (Splitting large batches in two so that the first half's pass 2 runs beside the second half's pass 1 was measured in round 3 —
    // 32.9 / 119.1 ms against 31.5 / 119.4 at 8192 queries, efSearch 128 / 512 — and is gone.)
    uint32_t* counters = static_cast<uint32_t*>(c->s_counters.p);
    HIPCHK(hipMemsetAsync(counters, 0, 64, c->stream));
    const float* dq = static_cast<const float*>(c->d_q.p);
    uint32_t* d_rows = static_cast<uint32_t*>(c->d_rows.p); float* d_dist = static_cast<float*>(c->d_dist.p);
    uint32_t* d_cnt = static_cast<uint32_t*>(c->d_cnt.p); uint32_t* d_ev = static_cast<uint32_t*>(c->d_ev.p);
    qv::HnswOpts wo; wo.vis = static_cast<uint32_t*>(c->vis_hash.p); wo.vis_cap = c->vis_hash_cap;
    if ((rc = traverse_wave(g, dq, c->d_qblk.p, &c->hub_table, nq, k, ef_search, wo, std::min(c->vis_hash_slots, nq), d_rows, d_dist, d_cnt, d_ev, c->stream))) return rc;
    hipError_t e = qv::launch_build_compact_redo(d_cnt, nq, static_cast<uint32_t*>(c->s_redo.p), counters, c->stream);
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "hnsw search launch failed: %s", hipGetErrorString(e));
    uint32_t* hc = static_cast<uint32_t*>(c->h_counters.p);
    // small result sets come back through one pinned buffer (a copy into pageable memory is staged by the runtime, chunk by chunk)
    const size_t all = 2 * obytes + 2 * cbytes;
    const bool pinned_out = all <= ((size_t)4 << 20);
    if (pinned_out && (rc = c->h_out.ensure(all))) return rc;
    unsigned char* h = static_cast<unsigned char*>(c->h_out.p);
    auto download = [&]() -> int {
        if (pinned_out) {
            HIPCHK(hipMemcpyAsync(h, c->d_rows.p, obytes, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(h + obytes, c->d_dist.p, obytes, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(h + 2 * obytes, c->d_cnt.p, cbytes, hipMemcpyDeviceToHost, c->stream));
            if (evals_out) HIPCHK(hipMemcpyAsync(h + 2 * obytes + cbytes, c->d_ev.p, cbytes, hipMemcpyDeviceToHost, c->stream));
        } else {
            HIPCHK(hipMemcpyAsync(rows_out, c->d_rows.p, obytes, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(dist_out, c->d_dist.p, obytes, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(count_out, c->d_cnt.p, cbytes, hipMemcpyDeviceToHost, c->stream));
            if (evals_out) HIPCHK(hipMemcpyAsync(evals_out, c->d_ev.p, cbytes, hipMemcpyDeviceToHost, c->stream));
        }
        return QV_OK;
    };
    if ((rc = download())) return rc;
    HIPCHK(hipMemcpyAsync(hc, counters, 32, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(wait_device());
    if (pinned_out) {
        memcpy(rows_out, h, obytes); memcpy(dist_out, h + obytes, obytes); memcpy(count_out, h + 2 * obytes, cbytes);
        if (evals_out) memcpy(evals_out, h + 2 * obytes + cbytes, cbytes);
    }
    const uint32_t flagged = hc[0];
    if (trace) fprintf(stderr, "qv: graph search pass 1 + download %.3f ms (%u flagged)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_p1).count(), flagged);
    if (flagged == 0) return QV_OK;                                     // (most batches: the exact-heap kernel is not even launched)
    // pass 2: the flagged queries (0.06 - 2 % of them by efSearch: count 0xFFFFFFFE) through the exact-heap kernel, one wavefront per
    // query on a CU of its own — a latency-bound ~4.6 ms at efSearch 128 whether it redoes one query or a hundred.  Everything else is
    // final NOW: a caller that shares this batch with others (qv_coalesce.h) lets them go before it starts.
    if (early) early();
    g->tie_reruns.fetch_add(flagged, std::memory_order_relaxed);
    const auto t_p2 = std::chrono::steady_clock::now();
    qv::HnswOpts ho; ho.vis = static_cast<uint32_t*>(c->vis_bits.p); ho.vis_cap = c->vis_bits_words;
    ho.redo_idx = static_cast<const uint32_t*>(c->s_redo.p); ho.redo_n = counters;
    const uint32_t hgrid = std::min(std::min(std::min(qv::hnsw_grid(idx->cus, efx, nq), c->vis_bits_slots), nq), flagged);
    e = qv::launch_hnsw_search(idx->view(), g->g, dq, c->d_qblk.p, nq, k, ef_search, ho, hgrid, false, d_rows, d_dist, d_cnt, d_ev, c->stream);
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "hnsw search launch failed: %s", hipGetErrorString(e));
    if (pinned_out) {
        // only the flagged queries' entries change: the others' (already handed out, perhaps) are left alone
        if ((rc = download())) return rc;
        HIPCHK(wait_device());
        const uint32_t* hcnt = reinterpret_cast<const uint32_t*>(h + 2 * obytes);
        for (uint32_t q = 0; q < nq; q++) {
            if (count_out[q] != 0xFFFFFFFEu) continue;
            memcpy(rows_out + (size_t)q * k, h + (size_t)q * k * 4, (size_t)k * 4);
            memcpy(dist_out + (size_t)q * k, h + obytes + (size_t)q * k * 4, (size_t)k * 4);
            if (evals_out) evals_out[q] = reinterpret_cast<const uint32_t*>(h + 2 * obytes + cbytes)[q];
            count_out[q] = hcnt[q];
        }
    } else {
        if ((rc = download())) return rc;
        HIPCHK(wait_device());
    }
    if (trace) fprintf(stderr, "qv: graph search pass 2 (%u flagged queries redone by the exact-heap kernel) + download %.3f ms\n", flagged,
                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_p2).count());
    for (uint32_t q = 0; q < nq; q++)
        if (count_out[q] == 0xFFFFFFFEu) return fail(QV_ERR_DEVICE, "hnsw search: a flagged query was not redone");
    return QV_OK;
}

int graph_search_direct(qv_graph* g, const float* queries, uint32_t nq, uint32_t k, uint32_t ef_search,
                        uint32_t* rows_out, float* dist_out, uint32_t* count_out, uint32_t* evals_out, const std::function<void()>& early = nullptr) {
    GraphCtx* c = nullptr;
    int rc = acquire_gctx(g, &c);
    if (rc != QV_OK) return rc;
    GCtxGuard guard{g, c};
    return graph_search_ctx(g, c, queries, nq, k, ef_search, rows_out, dist_out, count_out, evals_out, early);
}

}  // namespace

extern "C" {

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
    // Small calls — the reference's host sends one query per call, concurrently (hnsw.go:602-606) — that find two batches already
    // in flight ride the next one together (same k and efSearch: those decide a traversal's result).  Larger batches fill the chip
    // by themselves and run as they are, each in its own context.
    static const bool off = getenv("QV_COALESCE") && atoi(getenv("QV_COALESCE")) == 0;        // measurement switch, read once per process
    if (nq > 64 || off) return graph_search_direct(g, queries, nq, k, ef_search, rows_out, dist_out, count_out, evals_out);
    char err[256]; err[0] = 0;
    const int rc = g->front.submit(
        (uint64_t)k | ((uint64_t)ef_search << 16), queries, nq, idx->dim, k, rows_out, dist_out, count_out, evals_out,
        [&] { return graph_search_direct(g, queries, nq, k, ef_search, rows_out, dist_out, count_out, evals_out); },
        [&](qvco::Group& grp, auto& early) {
            grp.size_outputs(true);
            (void)hipSetDevice(idx->device);
            return graph_search_direct(g, grp.queries(), grp.nq, grp.kmax, ef_search, grp.rows.data(), grp.dist.data(), grp.count.data(), grp.evals.data(),
                                       [&early] { early(); });
        },
        [] { return qv_last_error(); }, err, sizeof(err));
    if (rc != QV_OK && err[0]) return fail(rc, "%s", err);
    return rc;
}

int qv_graph_coalesce_stats(qv_graph* g, uint64_t out[8]) {
    if (!g || !out) return fail(QV_ERR_INVALID_ARG, "graph/out is null");
    g->front.stats.read(out);
    return QV_OK;
}
int qv_graph_coalesce_early_rounds(qv_graph* g, uint64_t* out) {
    if (!g || !out) return fail(QV_ERR_INVALID_ARG, "graph/out is null");
    *out = g->front.stats.early();
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
    if ((rc = traverse_wave(g, d_queries, g->d_qblk.p, &g->hubs.table, nq, k, ef_search, wave_opts(g), grid, d_rows_out, d_dist_out, d_count_out, d_evals_out, s))) return rc;
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
    if (search_redo) *search_redo = g->tie_reruns.load();
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
