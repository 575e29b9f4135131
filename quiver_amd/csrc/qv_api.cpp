// qv_api.cpp — the C ABI of include/qv.h over the gfx950 kernels (qv_*.hip).
//
// Host-side responsibilities only: argument checks in the reference's order and
// wording, device-memory ownership (tile storage sized for 288 GB HBM3E: one
// allocation per array, geometric growth, explicit reserve), a pool of per-call
// search contexts (stream + pinned staging + workspace) so concurrent searches do
// not serialise, and per-stream workspaces for the *_device entry points.
// There is NO CPU compute path here: every distance and every selection runs in
// a HIP kernel, and a missing/failed HIP runtime is a loud error, never a fallback.
#include "qv_api_internal.h"

#include <atomic>

namespace {
thread_local char g_err[512] = "";

// (GPU_MAX_HW_QUEUES — the HIP runtime's hardware queues per device, 4 by default — is the HOST's to set before its first HIP call:
// INTEGRATION.md "Environment"; qv_runtime_info reports what the process has.  The library does not touch the environment.)
}

int qv_fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace {

int acquire_ctx(qv_index* idx, SearchCtx** out) {
    {
        std::lock_guard<std::mutex> g(idx->ctx_mu);
        if (!idx->free_ctx.empty()) { *out = idx->free_ctx.back(); idx->free_ctx.pop_back(); return QV_OK; }
    }
    SearchCtx* c = new (std::nothrow) SearchCtx();
    if (!c) return fail(QV_ERR_OOM, "out of host memory");
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(QV_ERR_DEVICE, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
    std::lock_guard<std::mutex> g(idx->ctx_mu);
    idx->all_ctx.push_back(c);
    *out = c;
    return QV_OK;
}
void release_ctx(qv_index* idx, SearchCtx* c) {
    std::lock_guard<std::mutex> g(idx->ctx_mu);
    idx->free_ctx.push_back(c);
}
struct CtxGuard {
    qv_index* idx; SearchCtx* c;
    ~CtxGuard() { if (c) release_ctx(idx, c); }
};

// The workspace of a caller stream.  Returns with `hold` locked on that workspace: the caller keeps it until its launches
// are enqueued, so that another thread using the same stream cannot grow (free) the buffer between "fetch the pointer" and
// "launch" — once enqueued, stream order protects the kernels (a grow drains the stream first).
int stream_workspace(qv_index* idx, hipStream_t s, size_t bytes, void** out, std::unique_lock<std::mutex>* hold, uint32_t** tickets_out = nullptr) {
    Workspace* w;
    {
        std::lock_guard<std::mutex> g(idx->ws_mu);
        Workspace*& slot = idx->stream_ws[s];
        if (!slot) slot = new (std::nothrow) Workspace();
        if (!slot) return fail(QV_ERR_OOM, "out of host memory");
        w = slot;
    }
    *hold = std::unique_lock<std::mutex>(w->mu);
    if (bytes > w->ws.cap) {
        // a larger workspace replaces one the stream may still be using: drain first
        HIPCHK(hipStreamSynchronize(s));
        int rc = w->ws.ensure(bytes + bytes / 2);
        if (rc != QV_OK) return rc;
    }
    *out = w->ws.p;
    if (tickets_out) {                                                 // the single-launch small scan's tickets: zeroed once, left zero by the kernel
        if (!w->tickets.p) {
            int rc = w->tickets.ensure(256);
            if (rc != QV_OK) return rc;
            // on the caller's stream, ahead of the kernel that reads them: hipMemset on device memory returns before the fill has run, and the
            // null stream it runs on is not ordered against a non-blocking stream — a first search on a fresh stream could start on
            // tickets that were not zero yet, or be zeroed under (test_concurrent_searches_on_one_handle failed about one run in four)
            HIPCHK(hipMemsetAsync(w->tickets.p, 0, 256, s));
        }
        *tickets_out = static_cast<uint32_t*>(w->tickets.p);
    }
    return QV_OK;
}

// grow device storage to hold `rows` rows (whole tiles), preserving contents
// one device array grown in place of the old one: the first keep_bytes are carried over, the rest is zero
template <typename T> hipError_t regrow(T** p, size_t keep_bytes, size_t new_bytes) {
    T* n = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&n), new_bytes);
    if (e != hipSuccess) return e;
    if (new_bytes > keep_bytes) e = hipMemset(reinterpret_cast<char*>(n) + keep_bytes, 0, new_bytes - keep_bytes);
    if (e == hipSuccess && keep_bytes) e = hipMemcpy(n, *p, keep_bytes, hipMemcpyDeviceToDevice);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);   // the fill and the copy may return early, and the index's streams are non-blocking: nothing of theirs may overtake these
    if (e != hipSuccess) { (void)hipFree(n); return e; }
    (void)hipFree(*p);
    *p = n;
    return hipSuccess;
}

// grow device storage to hold `rows` rows (whole tiles), preserving contents.  The arrays are grown ONE AT A TIME (old + new
// copy of one array live together, never of all four), and a failure half way leaves every array valid at its old or its
// new size with the capacity unchanged: nothing leaks, nothing dangles.
int ensure_rows(qv_index* idx, uint64_t rows, bool exact) {
    uint64_t need_tiles = (rows + 63) / 64;
    if (need_tiles <= idx->cap_tiles) return QV_OK;
    if (rows > 0xFFFFFFF0ull) return fail(QV_ERR_INVALID_ARG, "row count %llu exceeds the uint32 row space", (unsigned long long)rows);
    uint64_t new_tiles = exact ? need_tiles : std::max<uint64_t>(need_tiles, idx->cap_tiles + idx->cap_tiles / 2 + 16);
    const size_t tb = idx->tile_bytes();
    const uint64_t used_tiles = (idx->n_rows + 63) / 64;
    // pad rows must be finite and alive bits must start clear: the new part of every array is zero
    hipError_t e = regrow(&idx->d_tiles, used_tiles * tb, new_tiles * tb);
    if (e == hipSuccess) e = regrow(&idx->d_rnorm, used_tiles * 64 * sizeof(double), new_tiles * 64 * sizeof(double));
    if (e == hipSuccess) e = regrow(&idx->d_alive, used_tiles * sizeof(uint64_t), new_tiles * sizeof(uint64_t));
    if (e == hipSuccess) e = regrow(&idx->d_rres, used_tiles * 64 * sizeof(float), new_tiles * 64 * sizeof(float));
    if (e == hipSuccess && (idx->flags & QV_FLAG_ROWMAJOR))
        e = regrow(&idx->d_rowmaj, (size_t)idx->n_rows * idx->dim * sizeof(float), new_tiles * 64 * (size_t)idx->dim * sizeof(float));
    if (e == hipSuccess && (idx->flags & QV_FLAG_BF16_ROWS))
        e = regrow(&idx->d_bf16, used_tiles * idx->bf16_tile_bytes(), new_tiles * idx->bf16_tile_bytes());
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? QV_ERR_OOM : QV_ERR_DEVICE, "device allocation for %llu rows failed: %s", (unsigned long long)(new_tiles * 64), hipGetErrorString(e));
    idx->cap_tiles = new_tiles;
    return QV_OK;
}

void mark_alive_host(qv_index* idx, uint32_t row0, uint32_t n) {
    idx->alive_host.resize(((size_t)row0 + n + 63) / 64, 0);
    for (uint32_t r = row0; r < row0 + n; r++) idx->alive_host[r >> 6] |= 1ull << (r & 63);
}

// undo a partially applied add: counters back, alive words of every tile from the
// first touched one re-uploaded from the host mirror (the scan masks by alive bits only)
void rollback_rows(qv_index* idx, uint32_t base_rows, uint32_t base_live) {
    idx->n_rows = base_rows; idx->n_live = base_live;
    const size_t t0 = base_rows / 64;
    std::vector<uint64_t> words(idx->cap_tiles - t0, 0);
    if (t0 < idx->alive_host.size()) {
        idx->alive_host.resize((base_rows + 63) / 64);
        if (base_rows & 63) idx->alive_host[t0] &= (1ull << (base_rows & 63)) - 1;
        if (t0 < idx->alive_host.size()) words[0] = idx->alive_host[t0];
    }
    if (!words.empty()) (void)hipMemcpy(idx->d_alive + t0, words.data(), words.size() * sizeof(uint64_t), hipMemcpyHostToDevice);
}

size_t lds_limit_dim() { return 16384; }   // f64 query staging: 16384 * 8 B = 128 KiB of the CU's 160 KiB

}  // namespace

extern "C" {

const char* qv_last_error(void) { return g_err; }
int qv_abi_version(void) { return QV_ABI_VERSION; }

int qv_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int qv_device_info(int device, char* name_out, size_t name_cap, int* cu_count, uint64_t* hbm_bytes) {
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, device));
    if (name_out && name_cap) { snprintf(name_out, name_cap, "%s (%s)", p.name, p.gcnArchName); }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (uint64_t)p.totalGlobalMem;
    return QV_OK;
}

int qv_index_create(qv_index** out, uint32_t dim, qv_metric metric, int device, uint64_t flags) {
    if (!out) return fail(QV_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (dim == 0) return fail(QV_ERR_INVALID_ARG, "dimension must be positive");
    if (dim > lds_limit_dim()) return fail(QV_ERR_UNSUPPORTED, "dimension %u exceeds the supported maximum %zu", dim, lds_limit_dim());
    if ((int)metric < 0 || (int)metric >= QV_METRIC_COUNT) return fail(QV_ERR_INVALID_ARG, "unknown metric %d", (int)metric);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(QV_ERR_NO_DEVICE, "no HIP device available (%s); libqv has no CPU path", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(QV_ERR_INVALID_ARG, "device %d out of range (have %d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, device));
    qv_index* idx = new (std::nothrow) qv_index();
    if (!idx) return fail(QV_ERR_OOM, "out of host memory");
    idx->device = device; idx->cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    idx->dim = dim; idx->dim4 = (dim + 3) / 4; idx->metric = (int)metric; idx->flags = flags;
    *out = idx;
    return QV_OK;
}

void qv_index_destroy(qv_index* idx) {
    if (!idx) return;
    (void)hipSetDevice(idx->device);
    (void)hipDeviceSynchronize();
    for (SearchCtx* c : idx->all_ctx) { c->release(); delete c; }
    for (auto& kv : idx->stream_ws) { kv.second->ws.release(); kv.second->tickets.release(); delete kv.second; }
    (void)hipFree(idx->d_tiles); (void)hipFree(idx->d_rnorm); (void)hipFree(idx->d_alive); (void)hipFree(idx->d_rres); (void)hipFree(idx->d_rowmaj); (void)hipFree(idx->d_bf16);
    idx->mut_stage.release();
    delete idx;
}

int qv_index_reserve(qv_index* idx, uint64_t rows) {
    if (!idx) return fail(QV_ERR_INVALID_ARG, "index is null");
    HIPCHK(hipSetDevice(idx->device));
    return ensure_rows(idx, rows, true);
}

uint32_t qv_index_rows(const qv_index* idx) { return idx ? idx->n_rows : 0; }
uint32_t qv_index_size(const qv_index* idx) { return idx ? idx->n_live : 0; }
uint32_t qv_index_dim(const qv_index* idx) { return idx ? idx->dim : 0; }
int qv_index_metric(const qv_index* idx) { return idx ? idx->metric : -1; }

int qv_index_add_device(qv_index* idx, const float* d_rows, uint32_t n, uint32_t* first_row_out, void* stream) {
    if (idx) idx->row_writes++;
    if (!idx) return fail(QV_ERR_INVALID_ARG, "index is null");
    if (first_row_out) *first_row_out = idx->n_rows;
    if (n == 0) return QV_OK;
    if (!d_rows) return fail(QV_ERR_INVALID_ARG, "rows is null");
    HIPCHK(hipSetDevice(idx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = ensure_rows(idx, (uint64_t)idx->n_rows + n, false);
    if (rc != QV_OK) return rc;
    const uint32_t row0 = idx->n_rows;
    idx->n_rows += n;                                   // view() must cover the new rows
    hipError_t e = qv::launch_ingest(idx->view(), d_rows, row0, n, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { idx->n_rows = row0; return fail(QV_ERR_DEVICE, "ingest failed: %s", hipGetErrorString(e)); }
    idx->n_live += n;
    mark_alive_host(idx, row0, n);
    return QV_OK;
}

int qv_index_add(qv_index* idx, const float* rows, uint32_t n, uint32_t* first_row_out) {
    if (idx) idx->row_writes++;
    if (!idx) return fail(QV_ERR_INVALID_ARG, "index is null");
    if (first_row_out) *first_row_out = idx->n_rows;
    if (n == 0) return QV_OK;
    if (!rows) return fail(QV_ERR_INVALID_ARG, "rows is null");
    HIPCHK(hipSetDevice(idx->device));
    int rc = ensure_rows(idx, (uint64_t)idx->n_rows + n, false);
    if (rc != QV_OK) return rc;
    // stage in chunks of <= 256 MiB so a 30 GB corpus never needs a second full-size buffer
    const size_t row_bytes = (size_t)idx->dim * sizeof(float);
    const uint32_t chunk_rows = (uint32_t)std::max<size_t>(1, std::min<size_t>(n, ((size_t)256 << 20) / row_bytes));
    if ((rc = idx->mut_stage.ensure((size_t)chunk_rows * row_bytes))) return rc;
    float* d_stage = static_cast<float*>(idx->mut_stage.p);
    uint32_t done = 0;
    const uint32_t base_rows = idx->n_rows, base_live = idx->n_live;
    while (done < n) {
        uint32_t m = std::min(chunk_rows, n - done);
        hipError_t e = hipMemcpy(d_stage, rows + (size_t)done * idx->dim, (size_t)m * row_bytes, hipMemcpyHostToDevice);
        rc = e == hipSuccess ? QV_OK : fail(QV_ERR_DEVICE, "hipMemcpy H2D failed: %s", hipGetErrorString(e));
        uint32_t fr = 0;
        if (rc == QV_OK) rc = qv_index_add_device(idx, d_stage, m, &fr, nullptr);
        if (rc != QV_OK) {                      // all-or-nothing (InsertBatch rolls back, hybrid_index.go:175-216)
            rollback_rows(idx, base_rows, base_live);
            return rc;
        }
        done += m;
    }
    if (idx->mut_stage.cap > ((size_t)64 << 20)) idx->mut_stage.release();     // do not keep a bulk-load buffer around
    if (first_row_out) *first_row_out = base_rows;
    return QV_OK;
}

int qv_index_add_synthetic(qv_index* idx, uint64_t seed, uint64_t gen_row0, uint32_t n, uint32_t* first_row_out) {
    if (idx) idx->row_writes++;
    if (!idx) return fail(QV_ERR_INVALID_ARG, "index is null");
    if (first_row_out) *first_row_out = idx->n_rows;
    if (n == 0) return QV_OK;
    HIPCHK(hipSetDevice(idx->device));
    int rc = ensure_rows(idx, (uint64_t)idx->n_rows + n, false);
    if (rc != QV_OK) return rc;
    const uint32_t row0 = idx->n_rows;
    idx->n_rows += n;
    hipError_t e = qv::launch_generate(idx->view(), seed, gen_row0, row0, n, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) { idx->n_rows = row0; return fail(QV_ERR_DEVICE, "generate failed: %s", hipGetErrorString(e)); }
    idx->n_live += n;
    mark_alive_host(idx, row0, n);
    return QV_OK;
}

int qv_index_remove(qv_index* idx, const uint32_t* rows, uint32_t n) {
    if (!idx) return fail(QV_ERR_INVALID_ARG, "index is null");
    if (n == 0) return QV_OK;
    if (!rows) return fail(QV_ERR_INVALID_ARG, "rows is null");
    for (uint32_t i = 0; i < n; i++)
        if (rows[i] >= idx->n_rows) return fail(QV_ERR_OUT_OF_RANGE, "row %u out of range (rows: %u)", rows[i], idx->n_rows);
    HIPCHK(hipSetDevice(idx->device));
    { const int rc0 = idx->mut_stage.ensure((size_t)n * sizeof(uint32_t)); if (rc0 != QV_OK) return rc0; }
    uint32_t* d = static_cast<uint32_t*>(idx->mut_stage.p);
    hipError_t e = hipMemcpy(d, rows, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = qv::launch_set_alive(idx->view(), d, n, 0, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "remove failed: %s", hipGetErrorString(e));
    for (uint32_t i = 0; i < n; i++) {
        uint64_t bit = 1ull << (rows[i] & 63);
        if (idx->alive_host[rows[i] >> 6] & bit) { idx->alive_host[rows[i] >> 6] &= ~bit; idx->n_live--; }
    }
    return QV_OK;
}

int qv_index_update(qv_index* idx, uint32_t row, const float* vec) {
    if (idx) idx->row_writes++;
    if (!idx) return fail(QV_ERR_INVALID_ARG, "index is null");
    if (!vec) return fail(QV_ERR_INVALID_ARG, "vector is null");
    if (row >= idx->n_rows) return fail(QV_ERR_OUT_OF_RANGE, "row %u out of range (rows: %u)", row, idx->n_rows);
    HIPCHK(hipSetDevice(idx->device));
    { const int rc0 = idx->mut_stage.ensure((size_t)idx->dim * sizeof(float)); if (rc0 != QV_OK) return rc0; }
    float* d = static_cast<float*>(idx->mut_stage.p);
    hipError_t e = hipMemcpy(d, vec, (size_t)idx->dim * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = qv::launch_ingest(idx->view(), d, row, 1, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "update failed: %s", hipGetErrorString(e));
    uint64_t bit = 1ull << (row & 63);
    if (!(idx->alive_host[row >> 6] & bit)) { idx->alive_host[row >> 6] |= bit; idx->n_live++; }
    return QV_OK;
}

int qv_index_get_row(qv_index* idx, uint32_t row, float* vec_out) {
    if (!idx) return fail(QV_ERR_INVALID_ARG, "index is null");
    if (!vec_out) return fail(QV_ERR_INVALID_ARG, "vec_out is null");
    if (row >= idx->n_rows) return fail(QV_ERR_OUT_OF_RANGE, "row %u out of range (rows: %u)", row, idx->n_rows);
    HIPCHK(hipSetDevice(idx->device));
    SearchCtx* c = nullptr;                                            // a pooled context: its stream and staging buffers, no allocation per call
    int rc = acquire_ctx(idx, &c);
    if (rc != QV_OK) return rc;
    CtxGuard guard{idx, c};
    const size_t bytes = (size_t)idx->dim * sizeof(float);
    if ((rc = c->d_q.ensure(bytes)) || (rc = c->h_q.ensure(bytes))) return rc;
    hipError_t e = qv::launch_fetch_row(idx->view(), row, static_cast<float*>(c->d_q.p), c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c->h_q.p, c->d_q.p, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "get_row failed: %s", hipGetErrorString(e));
    memcpy(vec_out, c->h_q.p, bytes);
    return QV_OK;
}

int qv_index_get_rows(qv_index* idx, const uint32_t* rows, uint32_t n, float* out) {
    if (!idx) return fail(QV_ERR_INVALID_ARG, "index is null");
    if (n == 0) return QV_OK;
    if (!rows || !out) return fail(QV_ERR_INVALID_ARG, "rows/out is null");
    for (uint32_t i = 0; i < n; i++)
        if (rows[i] >= idx->n_rows) return fail(QV_ERR_OUT_OF_RANGE, "row %u out of range (rows: %u)", rows[i], idx->n_rows);
    HIPCHK(hipSetDevice(idx->device));
    SearchCtx* c = nullptr;
    int rc = acquire_ctx(idx, &c);
    if (rc != QV_OK) return rc;
    CtxGuard guard{idx, c};
    const size_t row_bytes = (size_t)idx->dim * sizeof(float);
    const uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(n, ((size_t)64 << 20) / row_bytes));   // <= 64 MiB per pass
    if ((rc = c->d_ids.ensure((size_t)chunk * 4)) || (rc = c->h_ids.ensure((size_t)chunk * 4)) || (rc = c->d_q.ensure((size_t)chunk * row_bytes))) return rc;
    for (uint32_t done = 0; done < n; done += chunk) {
        const uint32_t m = std::min(chunk, n - done);
        memcpy(c->h_ids.p, rows + done, (size_t)m * 4);
        HIPCHK(hipMemcpyAsync(c->d_ids.p, c->h_ids.p, (size_t)m * 4, hipMemcpyHostToDevice, c->stream));
        hipError_t e = qv::launch_fetch_rows(idx->view(), static_cast<const uint32_t*>(c->d_ids.p), m, static_cast<float*>(c->d_q.p), c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(out + (size_t)done * idx->dim, c->d_q.p, (size_t)m * row_bytes, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return fail(QV_ERR_DEVICE, "get_rows failed: %s", hipGetErrorString(e));
    }
    return QV_OK;
}

// shared by the host and device entry points: enqueue nq searches of list length kk
// (kk = min(k, live)), results written with row stride k_stride
static int enqueue_search(qv_index* idx, const float* d_queries, uint32_t nq, uint32_t kk, uint32_t k_stride,
                          void* ws, size_t ws_bytes, uint32_t* d_rows_out, float* d_dist_out, hipStream_t s,
                          const uint64_t* d_candidates = nullptr /* row bitmap replacing the tombstone bitmap (filtered search) */,
                          uint32_t* d_tickets = nullptr /* the stream's tickets: allows the single-launch small scan */,
                          uint32_t* done_flag = nullptr, uint32_t done_seq = 0, bool* flag_used = nullptr) {
    qv::IndexView v = idx->view();
    if (d_candidates) v.alive = const_cast<uint64_t*>(d_candidates);   // read-only in every scan kernel
    const qv::ScanPlan plan = qv::plan_scan(v.n_tiles, idx->cus);
    (void)ws_bytes;
    if (kk <= (uint32_t)qv::kMaxFusedK && kk == k_stride) {
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (idx->profiling) {
            if (hipEventCreate(&ev0) == hipSuccess && hipEventCreate(&ev1) == hipSuccess) {
                std::lock_guard<std::mutex> g(idx->prof_mu);
                idx->prof_events.emplace_back(ev0, ev1);
            } else { ev0 = ev1 = nullptr; }
        }
        if (d_tickets && qv::flat_small_applies(v, nq, kk) && !qv::flat_split_applies(v, nq, kk)) {   // small collection: scan + merge in one launch
            hipError_t e = qv::launch_flat_small(v, d_queries, nq, kk, ws, d_tickets, d_rows_out, d_dist_out, nq == 1 ? done_flag : nullptr, done_seq, s, ev0, ev1);
            if (e != hipSuccess) return fail(QV_ERR_DEVICE, "small scan launch failed: %s", hipGetErrorString(e));
            if (flag_used) *flag_used = nq == 1 && done_flag != nullptr;
            return QV_OK;
        }
        hipError_t e = qv::launch_flat_topk(v, plan, d_queries, nq, kk, ws, d_rows_out, d_dist_out, s, ev0, ev1, d_tickets, nq == 1 ? done_flag : nullptr, done_seq, flag_used);
        if (e != hipSuccess) return fail(QV_ERR_DEVICE, "flat scan launch failed: %s", hipGetErrorString(e));
        return QV_OK;
    }
    // 64 < k <= 8192 (the negative-example branches fetch max(2k, 30), hybrid_index.go:516-522; BatchSearch takes any k,
    // :677-811), and any shorter list whose output stride differs from its length: one key per row, then a radix SELECT
    if (kk > (uint32_t)qv::kMaxFusedK && kk <= (uint32_t)qv::kMaxWideK && nq == 1) {   // (several queries: shared corpus passes through the key-per-row path below)
        // up to 128: the scan's own stream with 2 keys per lane in the wave's list, then a selection over the waves' lists
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (idx->profiling && hipEventCreate(&ev0) == hipSuccess && hipEventCreate(&ev1) == hipSuccess) {
            std::lock_guard<std::mutex> g(idx->prof_mu);
            idx->prof_events.emplace_back(ev0, ev1);
        }
        hipError_t e = qv::launch_flat_wide(v, plan, d_queries, nq, kk, k_stride, ws, d_rows_out, d_dist_out, s, ev0, ev1);
        if (e != hipSuccess) return fail(QV_ERR_DEVICE, "wide scan launch failed: %s", hipGetErrorString(e));
        return QV_OK;
    }
    if (kk <= (uint32_t)qv::kMaxSelectK) {
        hipError_t e = qv::launch_flat_select(v, plan, d_queries, nq, kk, k_stride, ws, d_rows_out, d_dist_out, s);
        if (e != hipSuccess) return fail(QV_ERR_DEVICE, "scan + select launch failed: %s", hipGetErrorString(e));
        return QV_OK;
    }
    // full-ranking path (e.g. a filtered search asking for k = N, collection.go:679-682)
    for (uint32_t q = 0; q < nq; q++) {
        hipError_t e = qv::launch_flat_fullsort(v, plan, d_queries + (size_t)q * idx->dim, k_stride, ws,
                                                d_rows_out + (size_t)q * k_stride, d_dist_out + (size_t)q * k_stride, s);
        if (e != hipSuccess) return fail(QV_ERR_DEVICE, "full-sort launch failed: %s", hipGetErrorString(e));
    }
    return QV_OK;
}

static size_t search_ws_bytes(const qv_index* idx, uint32_t nq, uint32_t kk, uint32_t k_stride) {
    const uint32_t n_tiles = (idx->n_rows + 63) / 64;
    const qv::ScanPlan plan = qv::plan_scan(n_tiles, idx->cus);
    if (kk <= (uint32_t)qv::kMaxFusedK && kk == k_stride)   // partial lists + the multi-query kernels' query blocks (the small scan's lists fit in them)
        return std::max(qv::flat_small_workspace_bytes(std::min(nq, 4u), kk),
                        qv::scan_workspace_bytes(plan, nq, kk) + std::max((size_t)(nq + 16) * idx->dim4 * 4 * sizeof(double), qv::mq64_workspace_bytes(nq, idx->dim4)));
    if (kk > (uint32_t)qv::kMaxFusedK && kk <= (uint32_t)qv::kMaxWideK && nq == 1) return qv::flat_wide_workspace_bytes(plan, nq, kk);
    if (kk <= (uint32_t)qv::kMaxSelectK) return qv::flat_select_workspace_bytes(n_tiles, nq, kk, idx->dim4);
    return qv::full_sort_workspace_bytes(n_tiles);
}

constexpr uint32_t kHostBatch = 8192;   // queries per device pass of the host-pointer entry points

// the exact scans (single-query / multi-query / full ranking), host pointers
static int exact_search_host(qv_index* idx, const float* queries, uint32_t nq, uint32_t k,
                             uint32_t* rows_out, float* dist_out, uint32_t* count_out) {
    if (!idx) return fail(QV_ERR_INVALID_ARG, "index is null");
    if (nq == 0) return QV_OK;
    if (!queries || !count_out) return fail(QV_ERR_INVALID_ARG, "queries/count_out is null");
    if (idx->n_live == 0) {                                           // exact.go:96-98: empty index -> empty result, nil error
        for (uint32_t q = 0; q < nq; q++) count_out[q] = 0;
        return QV_OK;
    }
    if (k == 0) return fail(QV_ERR_K_NOT_POSITIVE, "k must be positive");        // exact.go:104-106
    if (!rows_out || !dist_out) return fail(QV_ERR_INVALID_ARG, "rows_out/dist_out is null");
    if (nq > kHostBatch) {                                                        // bound the workspace: very large batches go in slices
        for (uint32_t q0 = 0; q0 < nq; q0 += kHostBatch) {
            const int rc0 = exact_search_host(idx, queries + (size_t)q0 * idx->dim, std::min(kHostBatch, nq - q0), k,
                                              rows_out + (size_t)q0 * k, dist_out + (size_t)q0 * k, count_out + q0);
            if (rc0 != QV_OK) return rc0;
        }
        return QV_OK;
    }
    const uint32_t kk = std::min(k, idx->n_live);                                // exact.go:109-111
    HIPCHK(hipSetDevice(idx->device));
    SearchCtx* c = nullptr;
    int rc = acquire_ctx(idx, &c);
    if (rc != QV_OK) return rc;
    CtxGuard guard{idx, c};
    const size_t qbytes = (size_t)nq * idx->dim * sizeof(float);
    const size_t obytes = (size_t)nq * kk * sizeof(uint32_t);
    if ((rc = c->d_q.ensure(qbytes)) || (rc = c->h_q.ensure(qbytes)) || (rc = c->d_rows.ensure(obytes)) || (rc = c->d_dist.ensure(obytes)) ||
        (rc = c->h_rows.ensure(obytes)) || (rc = c->h_dist.ensure(obytes)) || (rc = c->ws.ensure(search_ws_bytes(idx, nq, kk, kk))))
        return rc;
    memcpy(c->h_q.p, queries, qbytes);
    // small result sets are written by the last kernel straight into the (device-visible) pinned buffers: two copy commands less
    // on the latency path of a single query; large ones go through device buffers and DMA.  On a small corpus (few workgroups,
    // each staging the query once) the query is read from the pinned buffer too: no copy command at all.
    const bool direct = (size_t)nq * kk <= 1024;
    const bool q_direct = direct && nq <= 4 && idx->n_rows <= 262144;
    if (!q_direct) HIPCHK(hipMemcpyAsync(c->d_q.p, c->h_q.p, qbytes, hipMemcpyHostToDevice, c->stream));
    // small collections: one launch for scan + merge, and the host polls a sequence number the kernel writes behind its results
    // instead of waiting for the stream
    uint32_t* tickets = nullptr; uint32_t* flag = nullptr; bool flag_used = false;
    if (!c->tickets.p) { if ((rc = c->tickets.ensure(256))) return rc; HIPCHK(hipMemsetAsync(c->tickets.p, 0, 256, c->stream)); }   // (on the stream that reads them: see stream_workspace)
    tickets = static_cast<uint32_t*>(c->tickets.p);                    // single-launch scans: the small collection's, and one query over a large one
    // (any single query whose scan is short enough to poll for: up to ~4 GB of rows, about half a millisecond)
    if (direct && (q_direct || (nq == 1 && (uint64_t)idx->n_rows * idx->dim4 * 16 <= (4ull << 30)))) {
        if (!c->h_flag.p) { if ((rc = c->h_flag.ensure(64))) return rc; *static_cast<volatile uint32_t*>(c->h_flag.p) = 0; }
        flag = static_cast<uint32_t*>(c->h_flag.p);
        c->flag_seq++;
    }
    rc = enqueue_search(idx, static_cast<const float*>(q_direct ? c->h_q.p : c->d_q.p), nq, kk, kk, c->ws.p, c->ws.cap,
                        static_cast<uint32_t*>(direct ? c->h_rows.p : c->d_rows.p), static_cast<float*>(direct ? c->h_dist.p : c->d_dist.p), c->stream,
                        nullptr, tickets, flag, c->flag_seq, &flag_used);
    if (rc != QV_OK) return rc;
    if (!direct) {
        HIPCHK(hipMemcpyAsync(c->h_rows.p, c->d_rows.p, obytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(c->h_dist.p, c->d_dist.p, obytes, hipMemcpyDeviceToHost, c->stream));
    }
    bool seen = false;
    if (flag_used) {
        const volatile uint32_t* f = static_cast<const volatile uint32_t*>(c->h_flag.p);
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spin = 0; !(seen = *f == c->flag_seq); spin++) {
            __builtin_ia32_pause();
            if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;   // something else holds the GPU: wait properly
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (seen && (c->flag_seq & 63u) == 0) HIPCHK(hipStreamSynchronize(c->stream));   // (lets the runtime retire its finished commands now and then)
    }
    if (!seen) HIPCHK(hipStreamSynchronize(c->stream));
    const uint32_t* hr = static_cast<const uint32_t*>(c->h_rows.p);
    const float* hd = static_cast<const float*>(c->h_dist.p);
    for (uint32_t q = 0; q < nq; q++) {
        memcpy(rows_out + (size_t)q * k, hr + (size_t)q * kk, (size_t)kk * sizeof(uint32_t));
        memcpy(dist_out + (size_t)q * k, hd + (size_t)q * kk, (size_t)kk * sizeof(float));
        for (uint32_t i = kk; i < k; i++) { rows_out[(size_t)q * k + i] = 0xFFFFFFFFu; dist_out[(size_t)q * k + i] = __builtin_inff(); }
        count_out[q] = kk;
    }
    return QV_OK;
}

// one call's worth of queries, in a context of the caller's own (what qv_index_search was before callers could share passes)
static int search_direct(qv_index* idx, const float* queries, uint32_t nq, uint32_t k,
                         uint32_t* rows_out, float* dist_out, uint32_t* count_out) {
    // many queries over a large corpus: the matrix-core filter + exact re-score returns the identical result faster (from 9 queries
    // with the bfloat16 filters, 32 with the fp32 one: batched_supported holds the measured crossovers)
    if (idx && k > 0 && idx->n_live >= 4 * (uint64_t)k && qv::batched_supported(idx->view(), nq, std::min(k, idx->n_live)))
        return qv_index_search_batched(idx, queries, nq, k, rows_out, dist_out, count_out);
    return exact_search_host(idx, queries, nq, k, rows_out, dist_out, count_out);
}

// Callers that may share a pass: small requests (the reference's host sends one query per call) with fused-list k over a corpus
// large enough that a pass costs more than a launch and a wake-up.  Everything else runs on its own as before — so does a
// request that fails a check, which must report the reference's error in the reference's order (exact.go:96-106).
constexpr uint32_t kCoalesceMaxNq = 8;
static bool coalesce_applies(const qv_index* idx, const float* queries, uint32_t nq, uint32_t k,
                             const uint32_t* rows_out, const float* dist_out, const uint32_t* count_out) {
    static const bool off = getenv("QV_COALESCE") && atoi(getenv("QV_COALESCE")) == 0;   // measurement switch, read once per process
    return !off && idx && queries && rows_out && dist_out && count_out && nq >= 1 && nq <= kCoalesceMaxNq && k >= 1 && k <= (uint32_t)qv::kMaxFusedK &&
           idx->n_live > 0;
}
// Passes in flight by what a pass costs: a scan of hundreds of megabytes is HBM-bound and a second one beside it only halves both (one
// lane); a small collection's pass is launch and wait latency, which concurrent passes on their own streams hide — until there are so
// many callers that the launches themselves queue up, and then the callers share passes there too.  Measured with 8 / 64 callers
// (profiles/r05_notes.md): 12k x 768 with ONE lane 42 k / 155 k QPS against 102 k / 44 k for every call on its own.
static int coalesce_lanes(const qv_index* idx) {
    static const int forced = getenv("QV_FLAT_LANES") ? atoi(getenv("QV_FLAT_LANES")) : 0;              // measurement switch, read once
    if (forced > 0) return forced;
    // a scan of 256 MiB or more fills the memory system by itself: one pass at a time, and its leader holds the group open for the
    // callers the previous pass released.  Below that a single-query pass leaves most of the device idle and a multi-query pass costs
    // more than it: four passes side by side, nobody held (profiles/r05_notes.md, the lane sweep at 5 MiB - 300 MiB).
    const uint64_t bytes = (uint64_t)idx->n_rows * idx->dim4 * 16;
    return bytes >= ((uint64_t)256 << 20) ? 1 : 4;
}

int qv_index_search(qv_index* idx, const float* queries, uint32_t nq, uint32_t k,
                    uint32_t* rows_out, float* dist_out, uint32_t* count_out) {
    if (!coalesce_applies(idx, queries, nq, k, rows_out, dist_out, count_out)) return search_direct(idx, queries, nq, k, rows_out, dist_out, count_out);
    {   // Lanes by size (coalesce_lanes); and on the smallest collections short shared passes: up to 32 queries a pass is the partial-chain
        // form of the scan (k_flat_scan_split_mq: 16 queries over 10 k x 768 in 87 us), beyond it the multi-query kernels (64 queries: 215 us), so
        // 16 per pass and four passes side by side carry more — callers on 10 k x 768 at 64 / 256 / 1024: 497 k / 473 k / 479 k QPS against 355 k /
        // 280 k / 107 k with passes of up to 256; 10 k x 128: 517 k / 253 k / 256 k against 508 k / 181 k / 101 k.  QV_FLAT_SMALL_GROUP overrides (measurements).
        const int lanes = coalesce_lanes(idx);
        idx->front.set_lanes(lanes, lanes == 1);
        static const int small_group = getenv("QV_FLAT_SMALL_GROUP") ? atoi(getenv("QV_FLAT_SMALL_GROUP")) : 0;
        const uint64_t bytes = (uint64_t)idx->n_rows * idx->dim4 * 16;
        // (64 - 256 MiB, still four lanes: 30 k x 768 at 256 / 1024 callers 445 k / 483 k QPS with passes of up to 64 against 425 k / 210 k with 256 and
        // 283 k / 261 k with 32; 60 k x 768 459 k / 459 k with 32 against 279 k / 99 k with 256 and 320 k / 156 k with 64; 80 k x 768 408 k / 392 k against 303 k / 105 k)
        const uint32_t by_size = bytes < ((uint64_t)64 << 20) ? 16u : (bytes < ((uint64_t)128 << 20) ? 64u : 32u);
        // (one lane, 256 MiB and more: 100 k x 768 at 256 / 1024 / 2048 callers 525 k / 463 k / 508 k QPS with passes of up to 128 against 350 k / 256 k / 189 k
        // with 256; 300 k x 768 374 k / 375 k / 355 k against 302 k / 289 k / 405 k; 1M x 768 200 k / 191 k / 194 k against 207 k / 289 k / 304 k: 128 below 1 GiB)
        static const int big_group = getenv("QV_FLAT_BIG_GROUP") ? atoi(getenv("QV_FLAT_BIG_GROUP")) : 0;
        const uint32_t one_lane = big_group > 0 ? (uint32_t)std::max(big_group, 8) : (bytes < ((uint64_t)1 << 30) ? 128u : 256u);
        idx->front.set_max_group(lanes > 1 ? (small_group > 0 ? (uint32_t)std::max(small_group, 8) : by_size) : one_lane);
    }
    char err[256]; err[0] = 0;
    const int rc = idx->front.submit(
        0, queries, nq, idx->dim, k, rows_out, dist_out, count_out, nullptr,
        [&] { return search_direct(idx, queries, nq, k, rows_out, dist_out, count_out); },
        [&](qvco::Group& g, auto&) {
            g.size_outputs(false);
            return search_direct(idx, g.queries(), g.nq, g.kmax, g.rows.data(), g.dist.data(), g.count.data());
        },
        [] { return qv_last_error(); }, err, sizeof(err));
    if (rc != QV_OK && err[0]) return fail(rc, "%s", err);                // (a rider's message comes from the thread that ran its group)
    return rc;
}

int qv_index_coalesce_stats(qv_index* idx, uint64_t out[8]) {
    if (!idx || !out) return fail(QV_ERR_INVALID_ARG, "index/out is null");
    idx->front.stats.read(out);
    return QV_OK;
}
int qv_index_coalesce_early_rounds(qv_index* idx, uint64_t* out) {
    if (!idx || !out) return fail(QV_ERR_INVALID_ARG, "index/out is null");
    *out = idx->front.stats.early();
    return QV_OK;
}

// Search with a negative example, device side of hybrid_index.go:517-570 / hnsw/adapter.go:345-437: the reference fetches
// retrieveK = max(2k, 30) nearest results, then calls the distance function once more per result against the negative
// example and re-ranks by d - w * d_neg.  Which rows are candidates depends on d alone, so the device part is: the flat scan
// for k_fetch results, then — on the same stream, row ids never leaving the device — the neighbour-batch kernel for those
// rows against the negative vector.  ONE call, one synchronisation; the host combines two floats per row and sorts <= 30
// records by (score, string id).
int qv_index_search_negative(qv_index* idx, const float* query, const float* negative, uint32_t k_fetch,
                             uint32_t* rows_out, float* dist_out, float* neg_dist_out, uint32_t* count_out) {
    if (!idx) return fail(QV_ERR_INVALID_ARG, "index is null");
    if (!query || !negative || !count_out) return fail(QV_ERR_INVALID_ARG, "query/negative/count_out is null");
    if (idx->n_live == 0) { *count_out = 0; return QV_OK; }                     // exact.go:96-98
    if (k_fetch == 0) return fail(QV_ERR_K_NOT_POSITIVE, "k must be positive"); // exact.go:104-106
    if (!rows_out || !dist_out || !neg_dist_out) return fail(QV_ERR_INVALID_ARG, "rows_out/dist_out/neg_dist_out is null");
    const uint32_t kk = std::min(k_fetch, idx->n_live);
    HIPCHK(hipSetDevice(idx->device));
    SearchCtx* c = nullptr;
    int rc = acquire_ctx(idx, &c);
    if (rc != QV_OK) return rc;
    CtxGuard guard{idx, c};
    const size_t vbytes = (size_t)idx->dim * sizeof(float), obytes = (size_t)kk * 4;
    if ((rc = c->d_q.ensure(2 * vbytes)) || (rc = c->h_q.ensure(2 * vbytes)) || (rc = c->d_rows.ensure(obytes)) || (rc = c->d_dist.ensure(2 * obytes)) ||
        (rc = c->h_rows.ensure(obytes)) || (rc = c->h_dist.ensure(2 * obytes)) || (rc = c->ws.ensure(search_ws_bytes(idx, 1, kk, kk))))
        return rc;
    memcpy(c->h_q.p, query, vbytes);
    memcpy(static_cast<char*>(c->h_q.p) + vbytes, negative, vbytes);
    HIPCHK(hipMemcpyAsync(c->d_q.p, c->h_q.p, 2 * vbytes, hipMemcpyHostToDevice, c->stream));
    float* d_dist = static_cast<float*>(c->d_dist.p);
    rc = enqueue_search(idx, static_cast<const float*>(c->d_q.p), 1, kk, kk, c->ws.p, c->ws.cap, static_cast<uint32_t*>(c->d_rows.p), d_dist, c->stream);
    if (rc != QV_OK) return rc;
    hipError_t e = qv::launch_distance_rows(idx->view(), static_cast<const float*>(c->d_q.p) + idx->dim, static_cast<const uint32_t*>(c->d_rows.p), kk, d_dist + kk, c->stream);
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "distance_rows launch failed: %s", hipGetErrorString(e));
    HIPCHK(hipMemcpyAsync(c->h_rows.p, c->d_rows.p, obytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->h_dist.p, d_dist, 2 * obytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(rows_out, c->h_rows.p, obytes);
    memcpy(dist_out, c->h_dist.p, obytes);
    memcpy(neg_dist_out, static_cast<const char*>(c->h_dist.p) + obytes, obytes);
    for (uint32_t i = kk; i < k_fetch; i++) { rows_out[i] = 0xFFFFFFFFu; dist_out[i] = __builtin_inff(); neg_dist_out[i] = __builtin_inff(); }
    *count_out = kk;
    return QV_OK;
}

int qv_index_search_masked(qv_index* idx, const float* queries, uint32_t nq, uint32_t k, const uint64_t* mask,
                           uint32_t* rows_out, float* dist_out, uint32_t* count_out) {
    if (!idx) return fail(QV_ERR_INVALID_ARG, "index is null");
    if (nq == 0) return QV_OK;
    if (!queries || !count_out || !mask) return fail(QV_ERR_INVALID_ARG, "queries/mask/count_out is null");
    if (idx->n_live == 0) { for (uint32_t q = 0; q < nq; q++) count_out[q] = 0; return QV_OK; }      // exact.go:96-98
    if (k == 0) return fail(QV_ERR_K_NOT_POSITIVE, "k must be positive");                             // exact.go:104-106
    if (!rows_out || !dist_out) return fail(QV_ERR_INVALID_ARG, "rows_out/dist_out is null");
    if (nq > kHostBatch) {
        for (uint32_t q0 = 0; q0 < nq; q0 += kHostBatch) {
            const int rc0 = qv_index_search_masked(idx, queries + (size_t)q0 * idx->dim, std::min(kHostBatch, nq - q0), k, mask,
                                                   rows_out + (size_t)q0 * k, dist_out + (size_t)q0 * k, count_out + q0);
            if (rc0 != QV_OK) return rc0;
        }
        return QV_OK;
    }
    HIPCHK(hipSetDevice(idx->device));
    SearchCtx* c = nullptr;
    int rc = acquire_ctx(idx, &c);
    if (rc != QV_OK) return rc;
    CtxGuard guard{idx, c};
    // candidates = live AND selected; the tile loop reads whole 64-row words, so the bitmap covers every tile
    const size_t words = ((size_t)idx->n_rows + 63) / 64;
    if ((rc = c->h_mask.ensure(words * 8)) || (rc = c->d_mask.ensure(words * 8))) return rc;
    uint64_t* hm = static_cast<uint64_t*>(c->h_mask.p);
    uint64_t matching = 0;
    for (size_t w = 0; w < words; w++) {
        const uint64_t a = w < idx->alive_host.size() ? idx->alive_host[w] : 0;
        hm[w] = a & mask[w];
        matching += (uint64_t)__builtin_popcountll(hm[w]);
    }
    const uint32_t kk = (uint32_t)std::min<uint64_t>(k, matching);                // the first k matches of the full ranking
    for (uint32_t q = 0; q < nq; q++) count_out[q] = kk;
    if (kk == 0) return QV_OK;
    const size_t qbytes = (size_t)nq * idx->dim * sizeof(float);
    const size_t obytes = (size_t)nq * kk * sizeof(uint32_t);
    if ((rc = c->d_q.ensure(qbytes)) || (rc = c->h_q.ensure(qbytes)) || (rc = c->d_rows.ensure(obytes)) || (rc = c->d_dist.ensure(obytes)) ||
        (rc = c->h_rows.ensure(obytes)) || (rc = c->h_dist.ensure(obytes)) || (rc = c->ws.ensure(search_ws_bytes(idx, nq, kk, kk))))
        return rc;
    memcpy(c->h_q.p, queries, qbytes);
    HIPCHK(hipMemcpyAsync(c->d_mask.p, c->h_mask.p, words * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_q.p, c->h_q.p, qbytes, hipMemcpyHostToDevice, c->stream));
    rc = enqueue_search(idx, static_cast<const float*>(c->d_q.p), nq, kk, kk, c->ws.p, c->ws.cap,
                        static_cast<uint32_t*>(c->d_rows.p), static_cast<float*>(c->d_dist.p), c->stream, static_cast<const uint64_t*>(c->d_mask.p));
    if (rc != QV_OK) return rc;
    HIPCHK(hipMemcpyAsync(c->h_rows.p, c->d_rows.p, obytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->h_dist.p, c->d_dist.p, obytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const uint32_t* hr = static_cast<const uint32_t*>(c->h_rows.p);
    const float* hd = static_cast<const float*>(c->h_dist.p);
    for (uint32_t q = 0; q < nq; q++) {
        memcpy(rows_out + (size_t)q * k, hr + (size_t)q * kk, (size_t)kk * sizeof(uint32_t));
        memcpy(dist_out + (size_t)q * k, hd + (size_t)q * kk, (size_t)kk * sizeof(float));
        for (uint32_t i = kk; i < k; i++) { rows_out[(size_t)q * k + i] = 0xFFFFFFFFu; dist_out[(size_t)q * k + i] = __builtin_inff(); }
    }
    return QV_OK;
}

int qv_index_search_device(qv_index* idx, const float* d_queries, uint32_t nq, uint32_t k,
                           uint32_t* d_rows_out, float* d_dist_out, void* stream) {
    if (!idx) return fail(QV_ERR_INVALID_ARG, "index is null");
    if (nq == 0) return QV_OK;
    if (!d_queries || !d_rows_out || !d_dist_out) return fail(QV_ERR_INVALID_ARG, "null device pointer");
    if (k == 0) return fail(QV_ERR_K_NOT_POSITIVE, "k must be positive");
    HIPCHK(hipSetDevice(idx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (idx->n_live == 0) {                                           // pad everything: no results
        HIPCHK(hipMemsetAsync(d_rows_out, 0xFF, (size_t)nq * k * sizeof(uint32_t), s));
        std::vector<float> inf((size_t)nq * k, __builtin_inff());
        HIPCHK(hipMemcpyAsync(d_dist_out, inf.data(), inf.size() * sizeof(float), hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
        return QV_OK;
    }
    const uint32_t kk = std::min(k, idx->n_live);
    void* ws = nullptr;
    std::unique_lock<std::mutex> ws_hold;
    uint32_t* tickets = nullptr;
    int rc = stream_workspace(idx, s, search_ws_bytes(idx, nq, kk, k), &ws, &ws_hold, &tickets);
    if (rc != QV_OK) return rc;
    return enqueue_search(idx, d_queries, nq, kk, k, ws, 0, d_rows_out, d_dist_out, s, nullptr, tickets);
}

}  // extern "C"

// Internal (qv_api_internal.h), for the sharded handle: the exact scan over the rows selected by a DEVICE bitmap d_candidates
// (already ANDed with the live rows by the caller; `matching` = its population count), enqueued on `stream`, no
// synchronisation.  Writes [nq][k_stride] lists: min(k_stride, matching) results, the rest padded (0xFFFFFFFF, +inf).
int qv_internal_search_candidates_device(qv_index* idx, const float* d_queries, uint32_t nq, uint32_t k_stride, const uint64_t* d_candidates,
                                         uint64_t matching, uint32_t* d_rows_out, float* d_dist_out, void* stream) {
    if (!idx || !d_queries || !d_candidates || !d_rows_out || !d_dist_out || k_stride == 0) return fail(QV_ERR_INVALID_ARG, "null argument");
    if (nq == 0) return QV_OK;
    HIPCHK(hipSetDevice(idx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t kk = (uint32_t)std::min<uint64_t>(k_stride, matching);
    void* ws = nullptr;
    std::unique_lock<std::mutex> ws_hold;
    int rc = stream_workspace(idx, s, search_ws_bytes(idx, nq, kk, k_stride), &ws, &ws_hold);
    if (rc != QV_OK) return rc;
    return enqueue_search(idx, d_queries, nq, kk, k_stride, ws, 0, d_rows_out, d_dist_out, s, d_candidates);
}

// Internal, for the sharded handle: the exact scan for the queries of a batch whose d_flags word is set (handed back by
// qv_index_search_batched_device), enqueued on `stream` behind the batch — listed and scanned on the device, nothing read back.
int qv_internal_redo_flagged_device(qv_index* idx, const float* d_queries, uint32_t nq, uint32_t k, const uint32_t* d_flags,
                                    uint32_t* d_rows_out, float* d_dist_out, void* stream) {
    if (!idx || !d_queries || !d_flags || !d_rows_out || !d_dist_out || k == 0 || k > (uint32_t)qv::kMaxSelectK) return fail(QV_ERR_INVALID_ARG, "bad argument");
    if (nq == 0) return QV_OK;
    HIPCHK(hipSetDevice(idx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const qv::IndexView v = idx->view();
    const qv::ScanPlan plan = qv::plan_scan(v.n_tiles, idx->cus);
    void* ws = nullptr;
    std::unique_lock<std::mutex> ws_hold;
    if (k > (uint32_t)qv::kMaxFusedK) {                               // the selection path's hand-backs: QV_ERR_UNSUPPORTED = the caller reads the flags itself
        int rc0 = stream_workspace(idx, s, qv::flat_select_redo_workspace_bytes(v.n_tiles, nq, k, v.dim, v.dim4), &ws, &ws_hold);
        if (rc0 != QV_OK) return rc0;
        hipError_t e0 = qv::launch_flat_select_redo(v, plan, d_queries, nq, k, k, d_flags, ws, d_rows_out, d_dist_out, s);
        if (e0 == hipErrorNotSupported) return QV_ERR_UNSUPPORTED;
        if (e0 != hipSuccess) return fail(QV_ERR_DEVICE, "redo launch failed: %s", hipGetErrorString(e0));
        return QV_OK;
    }
    int rc = stream_workspace(idx, s, qv::redo_workspace_bytes(plan, nq, k), &ws, &ws_hold);
    if (rc != QV_OK) return rc;
    hipError_t e = qv::launch_flat_redo_flagged(v, plan, d_queries, nq, k, k, d_flags, ws, d_rows_out, d_dist_out, s);
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "redo launch failed: %s", hipGetErrorString(e));
    return QV_OK;
}

void qv_internal_drop_stream_workspace(qv_index* idx, void* stream) {
    if (!idx) return;
    Workspace* w = nullptr;
    {
        std::lock_guard<std::mutex> g(idx->ws_mu);
        auto it = idx->stream_ws.find(static_cast<hipStream_t>(stream));
        if (it == idx->stream_ws.end()) return;
        w = it->second;
        idx->stream_ws.erase(it);
    }
    { std::lock_guard<std::mutex> hold(w->mu); w->ws.release(); w->tickets.release(); }   // (hipFree waits for the device: nothing of the stream's is still running on it)
    delete w;
}

extern "C" {

// The filter walks the corpus once per 256 queries, so 257-320 queries cost two walks (2.6 ms against 1.4 ms at 1M x 768) while
// a batch of 64 or fewer costs half a walk: a small tail goes in a call of its own (0 = no split).
static uint32_t batched_tail(const qv::IndexView& v, uint32_t nq, uint32_t k) {
    const uint32_t rem = nq & 255u;
    return nq > 256 && rem >= 1 && rem <= 64 && (rem <= 8 || qv::batched_supported(v, rem, k)) ? rem : 0;
}

int qv_index_search_batched_device(qv_index* idx, const float* d_queries, uint32_t nq, uint32_t k,
                                   uint32_t* d_rows_out, float* d_dist_out, uint32_t* d_redo_flags_out, void* stream) {
    if (!idx) return fail(QV_ERR_INVALID_ARG, "index is null");
    if (nq == 0) return QV_OK;
    if (!d_queries || !d_rows_out || !d_dist_out || !d_redo_flags_out) return fail(QV_ERR_INVALID_ARG, "null device pointer");
    if (k == 0) return fail(QV_ERR_K_NOT_POSITIVE, "k must be positive");
    const qv::IndexView v = idx->view();
    if (k > idx->n_live || idx->n_live < 4 * (uint64_t)k || !qv::batched_supported(v, nq, k))
        return fail(QV_ERR_UNSUPPORTED, "the MFMA batched path does not apply to this index/query shape; use qv_index_search_device");
    HIPCHK(hipSetDevice(idx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (nq > kHostBatch) {                                                        // bound the workspace (sample scores + candidate slots: ~160 KB per query)
        for (uint32_t q0 = 0; q0 < nq; q0 += kHostBatch) {
            const uint32_t m = std::min(kHostBatch, nq - q0);
            int rc0;
            if (qv::batched_supported(v, m, k))
                rc0 = qv_index_search_batched_device(idx, d_queries + (size_t)q0 * idx->dim, m, k, d_rows_out + (size_t)q0 * k, d_dist_out + (size_t)q0 * k, d_redo_flags_out + q0, stream);
            else {
                HIPCHK(hipMemsetAsync(d_redo_flags_out + q0, 0, (size_t)m * sizeof(uint32_t), s));
                rc0 = qv_index_search_device(idx, d_queries + (size_t)q0 * idx->dim, m, k, d_rows_out + (size_t)q0 * k, d_dist_out + (size_t)q0 * k, stream);
            }
            if (rc0 != QV_OK) return rc0;
        }
        return QV_OK;
    }
    if (const uint32_t tail = batched_tail(v, nq, k)) {
        const uint32_t head = nq - tail;
        int rc0 = qv_index_search_batched_device(idx, d_queries, head, k, d_rows_out, d_dist_out, d_redo_flags_out, stream);
        if (rc0 != QV_OK) return rc0;
        const float* tq = d_queries + (size_t)head * idx->dim;
        if (qv::batched_supported(v, tail, k))
            return qv_index_search_batched_device(idx, tq, tail, k, d_rows_out + (size_t)head * k, d_dist_out + (size_t)head * k, d_redo_flags_out + head, stream);
        HIPCHK(hipMemsetAsync(d_redo_flags_out + head, 0, (size_t)tail * sizeof(uint32_t), s));
        return qv_index_search_device(idx, tq, tail, k, d_rows_out + (size_t)head * k, d_dist_out + (size_t)head * k, stream);
    }
    const qv::ScanPlan plan = qv::plan_scan(v.n_tiles, idx->cus);
    void* ws = nullptr;
    std::unique_lock<std::mutex> ws_hold;
    int rc = stream_workspace(idx, s, qv::batched_workspace_bytes(v, plan, nq, k), &ws, &ws_hold);
    if (rc != QV_OK) return rc;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (idx->profiling && hipEventCreate(&ev0) == hipSuccess && hipEventCreate(&ev1) == hipSuccess) {
        std::lock_guard<std::mutex> g(idx->prof_mu);
        idx->prof_events.emplace_back(ev0, ev1);
    }
    uint32_t* d_ovf = nullptr;
    hipError_t e = qv::launch_batched(v, plan, d_queries, nq, k, ws, d_rows_out, d_dist_out, &d_ovf, idx->cus, s, ev0, ev1);
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "batched launch failed: %s", hipGetErrorString(e));
    HIPCHK(hipMemcpyAsync(d_redo_flags_out, d_ovf, (size_t)nq * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    return QV_OK;
}

int qv_merge_topk_device(const float* d_dist_lists, const uint32_t* d_row_lists, uint32_t n_lists, uint32_t k,
                         uint32_t* d_rows_out, float* d_dist_out, void* stream) {
    if (!d_dist_lists || !d_row_lists || !d_rows_out || !d_dist_out) return fail(QV_ERR_INVALID_ARG, "null device pointer");
    if (k == 0) return fail(QV_ERR_K_NOT_POSITIVE, "k must be positive");
    if (k > (uint32_t)qv::kMaxFusedK || n_lists == 0 || (uint64_t)n_lists * k > 65536) return fail(QV_ERR_UNSUPPORTED, "merge supports k <= %d and n_lists*k <= 65536", qv::kMaxFusedK);
    hipError_t e = qv::launch_merge_pairs(d_dist_lists, d_row_lists, n_lists, k, d_rows_out, d_dist_out, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "merge launch failed: %s", hipGetErrorString(e));
    return QV_OK;
}

int qv_merge_topk_shards_device(const uint32_t* d_packed_lists, const uint32_t* d_bases, uint32_t n_lists, uint32_t nq, uint32_t k,
                                uint32_t* d_rows_out, float* d_dist_out, void* stream) {
    if (!d_packed_lists || !d_bases || !d_rows_out || !d_dist_out) return fail(QV_ERR_INVALID_ARG, "null device pointer");
    if (nq == 0) return QV_OK;
    if (k == 0) return fail(QV_ERR_K_NOT_POSITIVE, "k must be positive");
    if (k > (uint32_t)qv::kMaxFusedK || n_lists == 0 || (uint64_t)n_lists * k > 65536) return fail(QV_ERR_UNSUPPORTED, "merge supports k <= %d and n_lists*k <= 65536", qv::kMaxFusedK);
    hipError_t e = qv::launch_merge_shards(d_packed_lists, d_bases, n_lists, nq, k, d_rows_out, d_dist_out, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "merge launch failed: %s", hipGetErrorString(e));
    return QV_OK;
}

int qv_index_set_filter(qv_index* idx, int filter) {
    if (!idx) return fail(QV_ERR_INVALID_ARG, "index is null");
    if (filter < 0 || filter > QV_FILTER_OFF) return fail(QV_ERR_INVALID_ARG, "filter must be 0 (automatic), 1 (fp32 MFMA), 2 (bfloat16 x 3), 3 (bfloat16 x 1) or 4 (off); got %d", filter);
    idx->filter = filter;
    return QV_OK;
}

int qv_index_profile(qv_index* idx, int enable) {
    if (!idx) return fail(QV_ERR_INVALID_ARG, "index is null");
    idx->profiling = enable != 0;
    return QV_OK;
}

int qv_index_profile_read(qv_index* idx, double* scan_ms_sum_out, uint64_t* launches_out) {
    if (!idx) return fail(QV_ERR_INVALID_ARG, "index is null");
    HIPCHK(hipSetDevice(idx->device));
    std::lock_guard<std::mutex> g(idx->prof_mu);
    double sum = 0.0; uint64_t n = 0;
    for (auto& pr : idx->prof_events) {
        float ms = 0.f;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { sum += ms; n++; }
        (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
    }
    idx->prof_events.clear();
    if (scan_ms_sum_out) *scan_ms_sum_out = sum;
    if (launches_out) *launches_out = n;
    return QV_OK;
}

int qv_index_search_batched(qv_index* idx, const float* queries, uint32_t nq, uint32_t k,
                            uint32_t* rows_out, float* dist_out, uint32_t* count_out) {
    if (!idx) return fail(QV_ERR_INVALID_ARG, "index is null");
    if (nq == 0) return QV_OK;
    if (!queries || !count_out) return fail(QV_ERR_INVALID_ARG, "queries/count_out is null");
    if (idx->n_live == 0) { for (uint32_t q = 0; q < nq; q++) count_out[q] = 0; return QV_OK; }
    if (k == 0) return fail(QV_ERR_K_NOT_POSITIVE, "k must be positive");
    if (!rows_out || !dist_out) return fail(QV_ERR_INVALID_ARG, "rows_out/dist_out is null");
    const uint32_t kk = std::min(k, idx->n_live);
    const qv::IndexView v = idx->view();
    // the MFMA filter pays off for many queries over a large cosine/dot corpus; everything else
    // (and any k > 64) takes the exact multi-query scan, which returns the same result
    if (!qv::batched_supported(v, nq, kk) || kk != k || idx->n_live < 4 * kk)
        return exact_search_host(idx, queries, nq, k, rows_out, dist_out, count_out);
    if (nq > kHostBatch) {
        for (uint32_t q0 = 0; q0 < nq; q0 += kHostBatch) {
            const int rc0 = qv_index_search_batched(idx, queries + (size_t)q0 * idx->dim, std::min(kHostBatch, nq - q0), k,
                                                    rows_out + (size_t)q0 * k, dist_out + (size_t)q0 * k, count_out + q0);
            if (rc0 != QV_OK) return rc0;
        }
        return QV_OK;
    }
    if (const uint32_t tail = batched_tail(v, nq, kk)) {              // qv_index_search_batched falls back to the exact scan by itself
        const uint32_t head = nq - tail;
        const int rc0 = qv_index_search_batched(idx, queries, head, k, rows_out, dist_out, count_out);
        if (rc0 != QV_OK) return rc0;
        return qv_index_search_batched(idx, queries + (size_t)head * idx->dim, tail, k, rows_out + (size_t)head * k, dist_out + (size_t)head * k, count_out + head);
    }
    HIPCHK(hipSetDevice(idx->device));
    SearchCtx* c = nullptr;
    int rc = acquire_ctx(idx, &c);
    if (rc != QV_OK) return rc;
    CtxGuard guard{idx, c};
    const qv::ScanPlan plan = qv::plan_scan(v.n_tiles, idx->cus);
    const size_t qbytes = (size_t)nq * idx->dim * sizeof(float);
    const size_t obytes = (size_t)nq * kk * sizeof(uint32_t);
    const size_t fbytes = (size_t)nq * sizeof(uint32_t);
    if ((rc = c->d_q.ensure(qbytes)) || (rc = c->h_q.ensure(qbytes)) || (rc = c->d_rows.ensure(obytes)) || (rc = c->d_dist.ensure(obytes)) ||
        (rc = c->h_rows.ensure(obytes)) || (rc = c->h_dist.ensure(obytes)) || (rc = c->h_ids.ensure(fbytes)) ||
        (rc = c->ws.ensure(qv::batched_workspace_bytes(v, plan, nq, kk))))
        return rc;
    memcpy(c->h_q.p, queries, qbytes);
    HIPCHK(hipMemcpyAsync(c->d_q.p, c->h_q.p, qbytes, hipMemcpyHostToDevice, c->stream));
    uint32_t* d_ovf = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (idx->profiling && hipEventCreate(&ev0) == hipSuccess && hipEventCreate(&ev1) == hipSuccess) {
        std::lock_guard<std::mutex> g(idx->prof_mu);
        idx->prof_events.emplace_back(ev0, ev1);
    }
    hipError_t e = qv::launch_batched(v, plan, static_cast<const float*>(c->d_q.p), nq, kk, c->ws.p, static_cast<uint32_t*>(c->d_rows.p),
                                      static_cast<float*>(c->d_dist.p), &d_ovf, idx->cus, c->stream, ev0, ev1);
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "batched launch failed: %s", hipGetErrorString(e));
    HIPCHK(hipMemcpyAsync(c->h_rows.p, c->d_rows.p, obytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->h_dist.p, c->d_dist.p, obytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->h_ids.p, d_ovf, fbytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(rows_out, c->h_rows.p, obytes);
    memcpy(dist_out, c->h_dist.p, obytes);
    for (uint32_t q = 0; q < nq; q++) count_out[q] = kk;
    // queries whose candidate buffer overflowed (loose sample bound, e.g. a corpus sorted by cluster):
    // redo them with the exact scan
    const uint32_t* ovf = static_cast<const uint32_t*>(c->h_ids.p);
    std::vector<uint32_t> redo;
    for (uint32_t q = 0; q < nq; q++) if (ovf[q]) redo.push_back(q);
    idx->batched_redo += redo.size();
    guard.c = nullptr; release_ctx(idx, c);                          // the exact scan takes its own context
    if (!redo.empty()) {
        std::vector<float> rq((size_t)redo.size() * idx->dim);
        for (size_t i = 0; i < redo.size(); i++) memcpy(&rq[i * idx->dim], queries + (size_t)redo[i] * idx->dim, idx->dim * sizeof(float));
        std::vector<uint32_t> rr((size_t)redo.size() * kk), rc2(redo.size());
        std::vector<float> rd((size_t)redo.size() * kk);
        rc = exact_search_host(idx, rq.data(), (uint32_t)redo.size(), kk, rr.data(), rd.data(), rc2.data());
        if (rc != QV_OK) return rc;
        for (size_t i = 0; i < redo.size(); i++) {
            memcpy(rows_out + (size_t)redo[i] * kk, &rr[i * kk], kk * sizeof(uint32_t));
            memcpy(dist_out + (size_t)redo[i] * kk, &rd[i * kk], kk * sizeof(float));
        }
    }
    return QV_OK;
}

int qv_distance_rows_device(qv_index* idx, const float* d_query, const uint32_t* d_rows, uint32_t n,
                            float* d_dist_out, void* stream) {
    if (!idx) return fail(QV_ERR_INVALID_ARG, "index is null");
    if (n == 0) return QV_OK;
    if (!d_query || !d_rows || !d_dist_out) return fail(QV_ERR_INVALID_ARG, "null device pointer");
    HIPCHK(hipSetDevice(idx->device));
    hipError_t e = qv::launch_distance_rows(idx->view(), d_query, d_rows, n, d_dist_out, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "distance_rows launch failed: %s", hipGetErrorString(e));
    return QV_OK;
}

int qv_distance_rows(qv_index* idx, const float* query, const uint32_t* rows, uint32_t n, float* dist_out) {
    if (!idx) return fail(QV_ERR_INVALID_ARG, "index is null");
    if (n == 0) return QV_OK;
    if (!query || !rows || !dist_out) return fail(QV_ERR_INVALID_ARG, "query/rows/dist_out is null");
    for (uint32_t i = 0; i < n; i++)
        if (rows[i] >= idx->n_rows) return fail(QV_ERR_OUT_OF_RANGE, "row %u out of range (rows: %u)", rows[i], idx->n_rows);
    HIPCHK(hipSetDevice(idx->device));
    SearchCtx* c = nullptr;
    int rc = acquire_ctx(idx, &c);
    if (rc != QV_OK) return rc;
    CtxGuard guard{idx, c};
    const size_t qbytes = (size_t)idx->dim * sizeof(float), ibytes = (size_t)n * sizeof(uint32_t), obytes = (size_t)n * sizeof(float);
    if ((rc = c->d_q.ensure(qbytes)) || (rc = c->h_q.ensure(qbytes)) || (rc = c->d_ids.ensure(ibytes)) || (rc = c->h_ids.ensure(ibytes)) ||
        (rc = c->d_dist.ensure(obytes)) || (rc = c->h_dist.ensure(obytes)))
        return rc;
    memcpy(c->h_q.p, query, qbytes);
    memcpy(c->h_ids.p, rows, ibytes);
    // This call is a latency path (one searchLayer hop: a 3 KB query, <= 64 row ids, <= 64 floats back).  The pinned staging
    // buffers are device-visible, so the kernel reads the query and the ids from them and writes the distances into one
    // directly: one launch and one stream sync, no copy commands in between.
    hipError_t e = qv::launch_distance_rows(idx->view(), static_cast<const float*>(c->h_q.p), static_cast<const uint32_t*>(c->h_ids.p), n,
                                            static_cast<float*>(c->h_dist.p), c->stream);
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "distance_rows launch failed: %s", hipGetErrorString(e));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(dist_out, c->h_dist.p, obytes);
    return QV_OK;
}

// The DistanceFunc contract for ONE pair (surface.go:8; SURVEY.md 8b lists this entry point as host code): the reference calls
// it per pair at ~78 ns (final_bench.txt:47) from code that is not part of the scan — Surface.Distance, tests, the Go HNSW that
// reloaded collections use — and no GPU round trip (tens of microseconds) can serve that.  It is NOT a fallback of anything:
// every search / scan / batch entry point of this library runs on the device or fails, and none of them calls this.  The body
// is the kernels' own pair_distance<M> (qv_kernels.h) compiled for the host, so device and host state the arithmetic once.
int qv_distance_pair(qv_metric metric, const float* a, const float* b, uint32_t dim, float* out) {
    if ((int)metric < 0 || (int)metric >= QV_METRIC_COUNT) return fail(QV_ERR_INVALID_ARG, "unknown metric %d", (int)metric);
    if (!out || (dim && (!a || !b))) return fail(QV_ERR_INVALID_ARG, "a/b/out is null");
    *out = qv::host_pair_distance((int)metric, a, b, dim);
    return QV_OK;
}

int qv_distance_pairs(qv_metric metric, const float* a, const float* b, uint32_t n, uint32_t dim, float* dist_out, int device) {
    if ((int)metric < 0 || (int)metric >= QV_METRIC_COUNT) return fail(QV_ERR_INVALID_ARG, "unknown metric %d", (int)metric);
    if (n == 0) return QV_OK;
    if (!a || !b || !dist_out) return fail(QV_ERR_INVALID_ARG, "a/b/dist_out is null");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(QV_ERR_NO_DEVICE, "no HIP device available; libqv has no CPU path");
    if (device < 0 || device >= ndev) return fail(QV_ERR_INVALID_ARG, "device %d out of range (have %d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    // grow-only device buffers per device, shared by all callers (one call at a time per device): no hipMalloc per call
    struct PairBufs { std::mutex mu; Buf a, b, out; };
    static std::mutex table_mu;
    static std::map<int, PairBufs*> table;
    PairBufs* pb;
    { std::lock_guard<std::mutex> g(table_mu); PairBufs*& slot = table[device]; if (!slot) slot = new PairBufs(); pb = slot; }
    std::lock_guard<std::mutex> g(pb->mu);
    const size_t bytes = (size_t)n * dim * sizeof(float);
    int rc;
    if ((rc = pb->a.ensure(std::max<size_t>(bytes, 16))) || (rc = pb->b.ensure(std::max<size_t>(bytes, 16))) || (rc = pb->out.ensure((size_t)n * sizeof(float)))) return rc;
    if (bytes) e = hipMemcpy(pb->a.p, a, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && bytes) e = hipMemcpy(pb->b.p, b, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = qv::launch_distance_pairs((int)metric, static_cast<const float*>(pb->a.p), static_cast<const float*>(pb->b.p), n, dim, static_cast<float*>(pb->out.p), nullptr);
    if (e == hipSuccess) e = hipMemcpy(dist_out, pb->out.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "distance_pairs failed: %s", hipGetErrorString(e));
    return QV_OK;
}

}  // extern "C"
