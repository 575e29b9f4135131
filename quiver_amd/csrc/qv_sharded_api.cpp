// qv_sharded_api.cpp — the multi-GPU part of the C ABI (include/qv.h qv_sharded_*): one host process, one exact index
// (row shard) per GPU, one stream per GPU, ONE RCCL all-gather of the per-shard top-k per search, deterministic merge.
//
// This is SURVEY.md 8e behind the boundary: a Go host links libqv through cgo and gets the 8 GPUs of a node from a single
// handle, the same way it gets one GPU from a qv_index (the reference itself is single-process and has no counterpart).
//
//   shard g         a qv_index on devices[g]; its rows carry global row ids  base_g + local row,  base_g = g * span
//                   (span = 2^32 / n_shards rounded down to a tile multiple), so "global row = shard base + local row"
//                   holds without knowing the corpus size up front and ids stay stable as shards grow
//   search          query block -> every device (H2D on its stream); every shard runs its flat scan (the same kernels as
//                   qv_index_search_device) writing local rows + distances straight into its half-planes of a packed
//                   buffer [2][nq][k]; ncclAllGather (RCCL, communicator from ncclCommInitAll over the devices; xGMI
//                   between GPUs of a node) of that buffer — nq*k*8 bytes per shard: a latency collective;
//                   k_merge_shards on the first device under the same (distance, global row) order a single index uses
//   exchange modes  QV_SHARDED_RCCL (default) as above;  QV_SHARDED_PEER_COPY: every shard copies its packed buffer into
//                   the first device's gather buffer with hipMemcpyPeerAsync (point-to-point, what xGMI is) — also what
//                   lets several shards share one device, which RCCL refuses (used by the tests on a 1-GPU box)
#include "qv_api_internal.h"

#include <rccl/rccl.h>

namespace {

struct Shard {
    int device = 0;
    qv_index* idx = nullptr;
    uint32_t base = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_done = nullptr;       // this shard's part of the current search is on its way to the gather buffer
    Buf d_q, d_pack, d_gath;            // queries; [2][nq][k] local results; [G][2][nq][k] (RCCL: every shard; peer copy: first only)
    Buf d_flags;                        // [nq] queries the batched filter hands back (candidate overflow): redone with the exact scan
    ncclComm_t comm = nullptr;
};

}  // namespace

struct qv_sharded {
    uint32_t dim = 0; int metric = 0; uint64_t flags = 0;
    uint32_t span = 0;
    std::vector<Shard> sh;
    bool rccl = true;
    std::mutex mu;                       // one search / mutation at a time per handle
    PinBuf h_q, h_rows, h_dist;
    Buf d_bases, d_out_rows, d_out_dist;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;   // profiling: scans done / exchange done / merge done (first device)
    hipEvent_t ev_merged = nullptr;      // the previous search's merge has read the gather buffer
    bool profiling = false;
    double prof_scan_ms = 0, prof_exchange_ms = 0, prof_merge_ms = 0; uint64_t prof_n = 0;
};

namespace {

#define NCCLCHK(call)                                                                                    \
    do {                                                                                                 \
        ncclResult_t r_ = (call);                                                                        \
        if (r_ != ncclSuccess) return fail(QV_ERR_DEVICE, "%s failed: %s", #call, ncclGetErrorString(r_)); \
    } while (0)

// how a batch of n rows is cut over the shards: fill the emptiest up to the common level first, the rest to the last ones
void plan_add(const uint64_t* have, uint32_t G, uint64_t n, uint64_t* give) {
    uint64_t total = n, left = n;
    for (uint32_t g = 0; g < G; g++) { total += have[g]; give[g] = 0; }
    for (uint32_t g = 0; g < G && left; g++) {
        const uint64_t level = total / G + (g < total % G ? 1 : 0);
        uint64_t want = have[g] < level ? level - have[g] : 0;
        want = std::min(want, left);
        give[g] = want; left -= want;
    }
    for (uint32_t g = 0; g < G && left; g++) { give[g] += left; left = 0; }     // only when a shard already exceeds the level
}

uint64_t total_live(const qv_sharded* s) { uint64_t t = 0; for (auto& x : s->sh) t += qv_index_size(x.idx); return t; }

int search_locked(qv_sharded* s, const float* queries_host, const float* d_queries_dev0, uint32_t nq, uint32_t k,
                  uint32_t* rows_out, float* dist_out, uint32_t* count_out, uint32_t* d_rows_out, float* d_dist_out) {
    const uint32_t G = (uint32_t)s->sh.size();
    const size_t qbytes = (size_t)nq * s->dim * sizeof(float);
    const size_t words = (size_t)2 * nq * k;                               // per shard, 32-bit words
    int rc;
    if (queries_host) {
        if ((rc = s->h_q.ensure(qbytes))) return rc;
        memcpy(s->h_q.p, queries_host, qbytes);
    }
    Shard& s0 = s->sh[0];
    // scans
    for (uint32_t g = 0; g < G; g++) {
        Shard& x = s->sh[g];
        HIPCHK(hipSetDevice(x.device));
        HIPCHK(hipStreamWaitEvent(x.stream, s->ev_merged, 0));            // the gather buffer of the previous search has been consumed
        if ((rc = x.d_q.ensure(qbytes)) || (rc = x.d_pack.ensure(words * 4))) return rc;
        if (s->rccl || g == 0) { if ((rc = x.d_gath.ensure(words * 4 * G))) return rc; }
        if (queries_host) HIPCHK(hipMemcpyAsync(x.d_q.p, s->h_q.p, qbytes, hipMemcpyHostToDevice, x.stream));
        else if (g == 0) HIPCHK(hipMemcpyAsync(x.d_q.p, d_queries_dev0, qbytes, hipMemcpyDeviceToDevice, x.stream));
        else {                                                             // device-resident queries live on the first device
            HIPCHK(hipStreamWaitEvent(x.stream, s0.ev_done, 0));
            HIPCHK(hipMemcpyPeerAsync(x.d_q.p, x.device, s0.d_q.p, s0.device, qbytes, x.stream));
        }
        if (g == 0 && !queries_host) HIPCHK(hipEventRecord(s0.ev_done, s0.stream));   // the query block is on the first device
    }
    if (s->profiling) { HIPCHK(hipSetDevice(s0.device)); HIPCHK(hipEventRecord(s->ev0, s0.stream)); }
    std::vector<char> filtered(G, 0);
    std::vector<uint32_t> flags_host;
    for (uint32_t g = 0; g < G; g++) {
        Shard& x = s->sh[g];
        uint32_t* pack = static_cast<uint32_t*>(x.d_pack.p);
        if (qv_index_size(x.idx) == 0) {                                   // empty shard: no results (0xFFFFFFFF rows are skipped by the merge)
            HIPCHK(hipSetDevice(x.device));
            HIPCHK(hipMemsetAsync(pack, 0xFF, words * 4, x.stream));
            continue;
        }
        // batches go through the matrix-core filter + exact re-score where it applies (same results, qv_index_search's own rule);
        // everything else, and whatever the filter declines, through the exact scan
        int rcb = QV_ERR_UNSUPPORTED;
        if (nq >= 9) {
            if ((rc = x.d_flags.ensure((size_t)nq * 4))) return rc;
            rcb = qv_index_search_batched_device(x.idx, static_cast<const float*>(x.d_q.p), nq, k, pack, reinterpret_cast<float*>(pack + (size_t)nq * k),
                                                 static_cast<uint32_t*>(x.d_flags.p), x.stream);
        }
        if (rcb == QV_OK) filtered[g] = 1;
        else if (rcb != QV_ERR_UNSUPPORTED) return rcb;
        else if ((rc = qv_index_search_device(x.idx, static_cast<const float*>(x.d_q.p), nq, k, pack, reinterpret_cast<float*>(pack + (size_t)nq * k), x.stream)))
            return rc;
    }
    for (uint32_t g = 0; g < G; g++) {                                      // queries whose candidate buffer overflowed: the exact scan, one by one (rare)
        if (!filtered[g]) continue;
        Shard& x = s->sh[g];
        HIPCHK(hipSetDevice(x.device));
        flags_host.resize(nq);
        HIPCHK(hipMemcpyAsync(flags_host.data(), x.d_flags.p, (size_t)nq * 4, hipMemcpyDeviceToHost, x.stream));
        HIPCHK(hipStreamSynchronize(x.stream));
        uint32_t* pack = static_cast<uint32_t*>(x.d_pack.p);
        for (uint32_t q = 0; q < nq; q++) {
            if (!flags_host[q]) continue;
            if ((rc = qv_index_search_device(x.idx, static_cast<const float*>(x.d_q.p) + (size_t)q * s->dim, 1, k, pack + (size_t)q * k,
                                             reinterpret_cast<float*>(pack + (size_t)nq * k) + (size_t)q * k, x.stream)))
                return rc;
        }
    }
    // exchange
    if (s->profiling) { HIPCHK(hipSetDevice(s0.device)); HIPCHK(hipEventRecord(s->ev1, s0.stream)); }
    if (s->rccl) {
        NCCLCHK(ncclGroupStart());
        for (uint32_t g = 0; g < G; g++) {
            Shard& x = s->sh[g];
            ncclResult_t r = ncclAllGather(x.d_pack.p, x.d_gath.p, words, ncclUint32, x.comm, x.stream);
            if (r != ncclSuccess) { (void)ncclGroupEnd(); return fail(QV_ERR_DEVICE, "ncclAllGather failed: %s", ncclGetErrorString(r)); }
        }
        NCCLCHK(ncclGroupEnd());
    } else {
        for (uint32_t g = 0; g < G; g++) {
            Shard& x = s->sh[g];
            HIPCHK(hipSetDevice(x.device));
            unsigned char* dst = static_cast<unsigned char*>(s0.d_gath.p) + (size_t)g * words * 4;
            if (x.device == s0.device) HIPCHK(hipMemcpyAsync(dst, x.d_pack.p, words * 4, hipMemcpyDeviceToDevice, x.stream));
            else HIPCHK(hipMemcpyPeerAsync(dst, s0.device, x.d_pack.p, x.device, words * 4, x.stream));
            if (g != 0) HIPCHK(hipEventRecord(x.ev_done, x.stream));
        }
        HIPCHK(hipSetDevice(s0.device));
        for (uint32_t g = 1; g < G; g++) HIPCHK(hipStreamWaitEvent(s0.stream, s->sh[g].ev_done, 0));
    }
    // merge on the first device
    HIPCHK(hipSetDevice(s0.device));
    if (s->profiling) HIPCHK(hipEventRecord(s->ev2, s0.stream));
    uint32_t* o_rows = d_rows_out; float* o_dist = d_dist_out;
    if (!o_rows) {
        if ((rc = s->d_out_rows.ensure((size_t)nq * k * 4)) || (rc = s->d_out_dist.ensure((size_t)nq * k * 4))) return rc;
        o_rows = static_cast<uint32_t*>(s->d_out_rows.p); o_dist = static_cast<float*>(s->d_out_dist.p);
    }
    hipError_t e = qv::launch_merge_shards(static_cast<const uint32_t*>(s0.d_gath.p), static_cast<const uint32_t*>(s->d_bases.p), G, nq, k, o_rows, o_dist, s0.stream, true);
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "merge launch failed: %s", hipGetErrorString(e));
    HIPCHK(hipEventRecord(s->ev_merged, s0.stream));
    if (rows_out) {
        if ((rc = s->h_rows.ensure((size_t)nq * k * 4)) || (rc = s->h_dist.ensure((size_t)nq * k * 4))) return rc;
        HIPCHK(hipMemcpyAsync(s->h_rows.p, o_rows, (size_t)nq * k * 4, hipMemcpyDeviceToHost, s0.stream));
        HIPCHK(hipMemcpyAsync(s->h_dist.p, o_dist, (size_t)nq * k * 4, hipMemcpyDeviceToHost, s0.stream));
    }
    hipEvent_t ev3 = nullptr;
    if (s->profiling) { HIPCHK(hipEventCreate(&ev3)); HIPCHK(hipEventRecord(ev3, s0.stream)); }
    // the first device's stream is behind every shard's contribution (the collective / the copy events), so one sync
    // covers the whole search; the other streams stay ordered for the next call by themselves
    if (rows_out || s->profiling) HIPCHK(hipStreamSynchronize(s0.stream));
    if (s->profiling) {
        float a = 0, b = 0, c = 0;
        (void)hipEventElapsedTime(&a, s->ev0, s->ev1); (void)hipEventElapsedTime(&b, s->ev1, s->ev2); (void)hipEventElapsedTime(&c, s->ev2, ev3);
        (void)hipEventDestroy(ev3);
        s->prof_scan_ms += a; s->prof_exchange_ms += b; s->prof_merge_ms += c; s->prof_n++;
    }
    if (rows_out) {
        memcpy(rows_out, s->h_rows.p, (size_t)nq * k * 4);
        memcpy(dist_out, s->h_dist.p, (size_t)nq * k * 4);
        const uint64_t live = total_live(s);
        for (uint32_t q = 0; q < nq; q++) count_out[q] = (uint32_t)std::min<uint64_t>(k, live);
    }
    return QV_OK;
}

}  // namespace

extern "C" {

int qv_sharded_create(qv_sharded** out, uint32_t dim, qv_metric metric, const int* devices, int n_devices, uint64_t flags) {
    if (!out) return fail(QV_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (!devices || n_devices <= 0 || n_devices > 64) return fail(QV_ERR_INVALID_ARG, "device list must hold 1..64 devices (got %d)", n_devices);
    qv_sharded* s = new (std::nothrow) qv_sharded();
    if (!s) return fail(QV_ERR_OOM, "out of host memory");
    s->dim = dim; s->metric = (int)metric; s->flags = flags;
    s->rccl = !(flags & QV_SHARDED_PEER_COPY);
    s->span = qv_sharded_span(n_devices);
    s->sh.resize((size_t)n_devices);
    int rc = QV_OK;
    for (int g = 0; g < n_devices && rc == QV_OK; g++) {
        Shard& x = s->sh[(size_t)g];
        x.device = devices[g]; x.base = (uint32_t)g * s->span;
        rc = qv_index_create(&x.idx, dim, metric, x.device, flags & (QV_FLAG_ROWMAJOR | QV_FLAG_BF16_ROWS));
        if (rc != QV_OK) break;
        hipError_t e = hipSetDevice(x.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&x.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&x.ev_done, hipEventDisableTiming);
        if (e != hipSuccess) rc = fail(QV_ERR_DEVICE, "stream setup on device %d failed: %s", x.device, hipGetErrorString(e));
    }
    if (rc == QV_OK && !s->rccl) {                                         // peer copies between distinct devices need peer access
        for (int g = 1; g < n_devices; g++) {
            if (devices[g] == devices[0]) continue;
            (void)hipSetDevice(devices[g]); hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { rc = fail(QV_ERR_DEVICE, "peer access %d -> %d failed: %s", devices[g], devices[0], hipGetErrorString(e)); break; }
            (void)hipGetLastError();
        }
    }
    if (rc == QV_OK && s->rccl) {
        for (int a = 0; a < n_devices && rc == QV_OK; a++)
            for (int b = a + 1; b < n_devices; b++)
                if (devices[a] == devices[b]) { rc = fail(QV_ERR_INVALID_ARG, "device %d listed twice: RCCL needs one shard per device (use QV_SHARDED_PEER_COPY to co-locate shards)", devices[a]); break; }
        if (rc == QV_OK) {
            std::vector<ncclComm_t> comms((size_t)n_devices);
            ncclResult_t r = ncclCommInitAll(comms.data(), n_devices, devices);
            if (r != ncclSuccess) rc = fail(QV_ERR_DEVICE, "ncclCommInitAll over %d devices failed: %s", n_devices, ncclGetErrorString(r));
            else for (int g = 0; g < n_devices; g++) s->sh[(size_t)g].comm = comms[(size_t)g];
        }
    }
    if (rc == QV_OK) {
        hipError_t e = hipSetDevice(devices[0]);
        std::vector<uint32_t> bases((size_t)n_devices);
        for (int g = 0; g < n_devices; g++) bases[(size_t)g] = s->sh[(size_t)g].base;
        if (e == hipSuccess) { rc = s->d_bases.ensure(bases.size() * 4); if (rc == QV_OK) e = hipMemcpy(s->d_bases.p, bases.data(), bases.size() * 4, hipMemcpyHostToDevice); }
        if (e == hipSuccess) e = hipEventCreate(&s->ev0);
        if (e == hipSuccess) e = hipEventCreate(&s->ev1);
        if (e == hipSuccess) e = hipEventCreate(&s->ev2);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_merged, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(s->ev_merged, s->sh[0].stream);
        if (rc == QV_OK && e != hipSuccess) rc = fail(QV_ERR_DEVICE, "setup on device %d failed: %s", devices[0], hipGetErrorString(e));
    }
    if (rc != QV_OK) { char keep[512]; snprintf(keep, sizeof(keep), "%s", qv_last_error()); qv_sharded_destroy(s); return fail(rc, "%s", keep); }
    *out = s;
    return QV_OK;
}

void qv_sharded_destroy(qv_sharded* s) {
    if (!s) return;
    for (auto& x : s->sh) {
        (void)hipSetDevice(x.device);
        if (x.stream) (void)hipStreamSynchronize(x.stream);
        if (x.comm) (void)ncclCommDestroy(x.comm);
        x.d_q.release(); x.d_pack.release(); x.d_gath.release(); x.d_flags.release();
        if (x.ev_done) (void)hipEventDestroy(x.ev_done);
        if (x.stream) (void)hipStreamDestroy(x.stream);
        if (x.idx) qv_index_destroy(x.idx);
    }
    if (!s->sh.empty()) (void)hipSetDevice(s->sh[0].device);
    s->d_bases.release(); s->d_out_rows.release(); s->d_out_dist.release();
    s->h_q.release(); s->h_rows.release(); s->h_dist.release();
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->ev2) (void)hipEventDestroy(s->ev2);
    if (s->ev_merged) (void)hipEventDestroy(s->ev_merged);
    delete s;
}

uint32_t qv_sharded_span(int n_shards) {
    if (n_shards <= 0) return 0;
    const uint64_t span = (((uint64_t)1 << 32) - 64) / (uint64_t)n_shards;      // ids stay below the 0xFFFFFFFF 'no result' marker
    return (uint32_t)span & ~63u;
}

int qv_sharded_plan_add(const uint64_t* rows_per_shard, int n_shards, uint64_t n, uint64_t* give_out) {
    if (!rows_per_shard || !give_out || n_shards <= 0) return fail(QV_ERR_INVALID_ARG, "null argument");
    plan_add(rows_per_shard, (uint32_t)n_shards, n, give_out);
    return QV_OK;
}

int qv_sharded_shards(const qv_sharded* s) { return s ? (int)s->sh.size() : 0; }
uint64_t qv_sharded_size(const qv_sharded* s) { return s ? total_live(s) : 0; }
uint32_t qv_sharded_dim(const qv_sharded* s) { return s ? s->dim : 0; }

int qv_sharded_shard_info(const qv_sharded* s, int shard, int* device, uint32_t* base_row, uint32_t* rows, uint32_t* live) {
    if (!s || shard < 0 || shard >= (int)s->sh.size()) return fail(QV_ERR_INVALID_ARG, "shard %d out of range", shard);
    const Shard& x = s->sh[(size_t)shard];
    if (device) *device = x.device;
    if (base_row) *base_row = x.base;
    if (rows) *rows = qv_index_rows(x.idx);
    if (live) *live = qv_index_size(x.idx);
    return QV_OK;
}

int qv_sharded_reserve(qv_sharded* s, uint64_t rows_total) {
    if (!s) return fail(QV_ERR_INVALID_ARG, "handle is null");
    std::lock_guard<std::mutex> l(s->mu);
    const uint64_t G = s->sh.size();
    for (auto& x : s->sh) { int rc = qv_index_reserve(x.idx, (rows_total + G - 1) / G); if (rc != QV_OK) return rc; }
    return QV_OK;
}

// Appends n rows; global_rows_out[i] = the global row id of rows[i].  A batch is cut into contiguous pieces, one per shard,
// sized to even out the shards' fill (a batch of one goes to the emptiest shard): each piece is ONE device copy.
int qv_sharded_add(qv_sharded* s, const float* rows, uint32_t n, uint32_t* global_rows_out) {
    if (!s) return fail(QV_ERR_INVALID_ARG, "handle is null");
    if (n == 0) return QV_OK;
    if (!rows) return fail(QV_ERR_INVALID_ARG, "rows is null");
    std::lock_guard<std::mutex> l(s->mu);
    const uint32_t G = (uint32_t)s->sh.size();
    // target fill after the add: everyone at the same level where possible
    std::vector<uint64_t> have(G), give(G, 0);
    for (uint32_t g = 0; g < G; g++) have[g] = qv_index_rows(s->sh[g].idx);
    plan_add(have.data(), G, n, give.data());
    uint64_t off = 0;
    for (uint32_t g = 0; g < G; g++) {
        if (!give[g]) continue;
        Shard& x = s->sh[g];
        if (have[g] + give[g] > s->span) return fail(QV_ERR_OUT_OF_RANGE, "shard %u is full (%u rows per shard)", g, s->span);
        uint32_t first = 0;
        int rc = qv_index_add(x.idx, rows + off * s->dim, (uint32_t)give[g], &first);
        if (rc != QV_OK) return rc;                                      // earlier pieces stay (their ids were not handed out: the caller sees the error)
        if (global_rows_out) for (uint64_t i = 0; i < give[g]; i++) global_rows_out[off + i] = x.base + first + (uint32_t)i;
        off += give[g];
    }
    return QV_OK;
}

// Benchmark / test helper: n synthetic rows (the generator of qv_index_add_synthetic), shard g taking the contiguous block
// [g*n/G, (g+1)*n/G) of generator rows gen_row0.. — SURVEY.md 8e's contiguous row blocks.
int qv_sharded_add_synthetic(qv_sharded* s, uint64_t seed, uint64_t gen_row0, uint64_t n) {
    if (!s) return fail(QV_ERR_INVALID_ARG, "handle is null");
    std::lock_guard<std::mutex> l(s->mu);
    const uint64_t G = s->sh.size();
    for (uint64_t g = 0; g < G; g++) {
        const uint64_t b = g * n / G, e = (g + 1) * n / G;
        uint64_t done = b;
        while (done < e) {
            const uint32_t m = (uint32_t)std::min<uint64_t>(e - done, 2000000);
            uint32_t first = 0;
            int rc = qv_index_add_synthetic(s->sh[g].idx, seed, gen_row0 + done, m, &first);
            if (rc != QV_OK) return rc;
            done += m;
        }
    }
    return QV_OK;
}

int qv_sharded_remove(qv_sharded* s, const uint32_t* global_rows, uint32_t n) {
    if (!s) return fail(QV_ERR_INVALID_ARG, "handle is null");
    if (n == 0) return QV_OK;
    if (!global_rows) return fail(QV_ERR_INVALID_ARG, "rows is null");
    std::lock_guard<std::mutex> l(s->mu);
    const uint32_t G = (uint32_t)s->sh.size();
    std::vector<std::vector<uint32_t>> per(G);
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t g = std::min(global_rows[i] / s->span, G - 1);
        const uint32_t local = global_rows[i] - s->sh[g].base;
        if (local >= qv_index_rows(s->sh[g].idx)) return fail(QV_ERR_OUT_OF_RANGE, "global row %u is not in shard %u", global_rows[i], g);
        per[g].push_back(local);
    }
    for (uint32_t g = 0; g < G; g++)
        if (!per[g].empty()) { int rc = qv_index_remove(s->sh[g].idx, per[g].data(), (uint32_t)per[g].size()); if (rc != QV_OK) return rc; }
    return QV_OK;
}

int qv_sharded_search(qv_sharded* s, const float* queries, uint32_t nq, uint32_t k, uint32_t* rows_out, float* dist_out, uint32_t* count_out) {
    if (!s) return fail(QV_ERR_INVALID_ARG, "handle is null");
    if (nq == 0) return QV_OK;
    if (!queries || !rows_out || !dist_out || !count_out) return fail(QV_ERR_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> l(s->mu);
    const uint64_t live = total_live(s);
    if (live == 0) { for (uint32_t q = 0; q < nq; q++) count_out[q] = 0; return QV_OK; }          // exact.go:96-98
    if (k == 0) return fail(QV_ERR_K_NOT_POSITIVE, "k must be positive");                          // exact.go:104-106
    if (k > qv::kMaxFusedK) return fail(QV_ERR_UNSUPPORTED, "sharded search returns at most %d results per query (asked for %u)", qv::kMaxFusedK, k);
    return search_locked(s, queries, nullptr, nq, k, rows_out, dist_out, count_out, nullptr, nullptr);   // entries past count: row 0xFFFFFFFF, +inf
}

// Queries and results resident on the FIRST device of the handle; enqueues everything (no host synchronisation unless
// profiling is on) — the form bench.py times with HIP events.  `stream` (may be null) is made to wait for the results.
int qv_sharded_search_device(qv_sharded* s, const float* d_queries, uint32_t nq, uint32_t k, uint32_t* d_rows_out, float* d_dist_out, void* stream) {
    if (!s) return fail(QV_ERR_INVALID_ARG, "handle is null");
    if (nq == 0) return QV_OK;
    if (!d_queries || !d_rows_out || !d_dist_out) return fail(QV_ERR_INVALID_ARG, "null device pointer");
    if (k == 0) return fail(QV_ERR_K_NOT_POSITIVE, "k must be positive");
    if (k > qv::kMaxFusedK) return fail(QV_ERR_UNSUPPORTED, "sharded search returns at most %d results per query (asked for %u)", qv::kMaxFusedK, k);
    std::lock_guard<std::mutex> l(s->mu);
    Shard& s0 = s->sh[0];
    HIPCHK(hipSetDevice(s0.device));
    if (stream) {                                                          // order after the caller's earlier work on its stream
        hipEvent_t ev; HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(ev, static_cast<hipStream_t>(stream))); HIPCHK(hipStreamWaitEvent(s0.stream, ev, 0)); (void)hipEventDestroy(ev);
    }
    int rc = search_locked(s, nullptr, d_queries, nq, k, nullptr, nullptr, nullptr, d_rows_out, d_dist_out);
    if (rc != QV_OK) return rc;
    if (stream) {
        hipEvent_t ev; HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(ev, s0.stream)); HIPCHK(hipStreamWaitEvent(static_cast<hipStream_t>(stream), ev, 0)); (void)hipEventDestroy(ev);
    }
    return QV_OK;
}

int qv_sharded_sync(qv_sharded* s) {
    if (!s) return fail(QV_ERR_INVALID_ARG, "handle is null");
    for (auto& x : s->sh) { HIPCHK(hipSetDevice(x.device)); HIPCHK(hipStreamSynchronize(x.stream)); }
    return QV_OK;
}

int qv_sharded_profile(qv_sharded* s, int enable) {
    if (!s) return fail(QV_ERR_INVALID_ARG, "handle is null");
    std::lock_guard<std::mutex> l(s->mu);
    s->profiling = enable != 0;
    s->prof_scan_ms = s->prof_exchange_ms = s->prof_merge_ms = 0; s->prof_n = 0;
    return QV_OK;
}

int qv_sharded_profile_read(qv_sharded* s, double* scan_ms, double* exchange_ms, double* merge_ms, uint64_t* searches) {
    if (!s) return fail(QV_ERR_INVALID_ARG, "handle is null");
    std::lock_guard<std::mutex> l(s->mu);
    if (scan_ms) *scan_ms = s->prof_scan_ms;
    if (exchange_ms) *exchange_ms = s->prof_exchange_ms;
    if (merge_ms) *merge_ms = s->prof_merge_ms;
    if (searches) *searches = s->prof_n;
    s->prof_scan_ms = s->prof_exchange_ms = s->prof_merge_ms = 0; s->prof_n = 0;
    return QV_OK;
}

}  // extern "C"
