// qv_sharded_api.cpp — the multi-GPU part of the C ABI (include/qv.h qv_sharded_*): one host process, one exact index
// (row shard) per GPU, ONE exchange of the per-shard result lists per search, deterministic merge on the first device.
//
// This is SURVEY.md 8e behind the boundary: a Go host links libqv through cgo and gets the 8 GPUs of a node from a single
// handle with the whole core.Index surface (add / remove / update / get / search with any k / filtered search / search with
// a negative example / listed-row distances), the same way it gets one GPU from a qv_index (the reference itself is
// single-process and has no counterpart).
//
//   shard g         a qv_index on devices[g]; its rows carry global row ids  base_g + local row,  base_g = g * span
//                   (span = 2^32 / n_shards rounded down to a tile multiple), so "global row = shard base + local row"
//                   holds without knowing the corpus size up front and ids stay stable as shards grow
//   search          query block -> every device; every shard scans (the same kernels as qv_index_search_device) straight
//                   into its planes of a packed buffer [planes][nq][kcap] (local rows, distances, optional payload);
//                   ncclAllGather (RCCL; xGMI between the GPUs of a node) of that buffer — nq*k*8 bytes per shard for a
//                   top-k: a latency collective; merge on the first device under the (distance, global row) order a
//                   single index uses: k <= 64 in one wavefront-list kernel (k_merge_shards), larger k by one stable radix
//                   sort of the gathered keys (launch_merge_ranked) — a filtered Collection.Search asks for k = N
//   call contexts   every search runs in a context of its own (a stream per shard, staging and exchange buffers) taken from a
//                   pool, so searches on one handle run CONCURRENTLY like the reference's under its read lock
//                   (collection.go:647); only the enqueue of the collective is serialised (one communicator per GPU).
//                   Mutations take the handle exclusively (the reference holds c.Lock there)
//   exchange modes  RCCL (default) as above;  QV_SHARDED_PEER_COPY: every shard copies its packed buffer into the first
//                   device's gather buffer with hipMemcpyPeerAsync (point-to-point, what xGMI is) — also what lets several
//                   shards share one device, which RCCL refuses (used by the tests on a 1-GPU box)
#include "qv_api_internal.h"

#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <shared_mutex>
#include <thread>

#include <dlfcn.h>
#include <limits.h>
#include <stdlib.h>
#include <string>

#include <rccl/rccl.h>

namespace {

// Which HIP runtime and which RCCL this process actually bound (both resolve by soname to whatever was loaded first: in a
// PyTorch process torch's bundled pair, otherwise the /opt/rocm pair libqv was linked against).
std::string lib_of(const void* sym) {
    Dl_info info;
    if (!dladdr(sym, &info) || !info.dli_fname) return "?";
    char real[PATH_MAX];
    return realpath(info.dli_fname, real) ? std::string(real) : std::string(info.dli_fname);
}
std::string dir_of(const std::string& path) { const size_t p = path.rfind('/'); return p == std::string::npos ? std::string() : path.substr(0, p); }

struct Shard {
    int device = 0;
    qv_index* idx = nullptr;
    uint32_t base = 0;
    ncclComm_t comm = nullptr;
};

struct ShardBufs {                       // one shard's part of a call context (allocated on that shard's device)
    hipStream_t stream = nullptr;
    hipEvent_t ev_done = nullptr;        // this shard's part of the current search is on its way to the gather buffer
    Buf d_q, d_pack, d_gath;             // queries; [planes][nq][kcap] local results; [G][planes][nq][kcap] (RCCL: every shard; peer copy: first only)
    Buf d_flags, d_mask, d_ids, d_out;   // batched-filter redo flags; candidate bitmap; listed rows; listed-row distances
    PinBuf h_mask, h_ids, h_out, h_flags;
    void release() {
        d_q.release(); d_pack.release(); d_gath.release(); d_flags.release(); d_mask.release(); d_ids.release(); d_out.release();
        h_mask.release(); h_ids.release(); h_out.release(); h_flags.release();
        if (ev_done) (void)hipEventDestroy(ev_done);
        if (stream) (void)hipStreamDestroy(stream);
        ev_done = nullptr; stream = nullptr;
    }
};

struct CallCtx {
    std::vector<ShardBufs> sh;
    PinBuf h_q, h_rows, h_dist, h_aux;
    Buf d_out_rows, d_out_dist, d_aux, d_sort;   // first device: merged results, payload, radix-sort workspace
    hipEvent_t ev_merged = nullptr;      // this context's previous search has read the gather buffer
    hipEvent_t ev_in = nullptr, ev_out = nullptr;   // qv_sharded_search_device: the caller's stream before / after the search (pooled: two
                                         // hipEventCreate + hipEventDestroy per call were on the path whose whole budget at 8 GPUs is a 0.48 ms scan)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;   // profiling: scans enqueued / exchange enqueued / merge enqueued / done
};

// Per-shard host work of one search, optionally issued in parallel.  One thread walking the shards makes ~10 HIP calls per shard
// (wait, copy, two launches, copy, record, ...): 172-226 us per search at 8 shards (tools/dev_sharded_hostcost.py,
// profiles/r04_sharded_hostcost.md) against a 480 us scan per shard at 10M x 768 over 8 GPUs — hidden while searches are
// pipelined, not for a lone one.  With QV_SHARDED_WORKERS=1 a handle of more than two shards keeps G - 1 workers; the caller's
// thread takes shard 0 and the parts that must stay on one thread (the grouped collective, the merge).  Workers spin briefly after
// a job before they block, so a stream of searches finds them awake.
struct ShardJob {
    std::function<int(uint32_t)> fn;
    std::atomic<uint32_t> remaining{0};
    std::vector<int> rc;
    std::vector<std::string> msg;
};
struct ShardPool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::pair<ShardJob*, uint32_t>> q;
    bool stop = false;
    void start(uint32_t n) {
        for (uint32_t i = 0; i < n; i++) th.emplace_back([this] { loop(); });
    }
    void loop() {
        for (;;) {
            std::pair<ShardJob*, uint32_t> item{nullptr, 0};
            for (int spin = 0; spin < 4000 && !item.first; spin++) {          // ~100 us of polling before sleeping
                if (m.try_lock()) {
                    if (!q.empty()) { item = q.front(); q.pop_front(); }
                    const bool st = stop;
                    m.unlock();
                    if (st && !item.first) return;
                }
                if (!item.first) { for (int p = 0; p < 16; p++) __builtin_ia32_pause(); }
            }
            if (!item.first) {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [this] { return stop || !q.empty(); });
                if (q.empty()) return;                                           // stop
                item = q.front(); q.pop_front();
            }
            ShardJob* j = item.first;
            int rc;
            try {                                                              // (nothing may leave a pool thread: an escaping exception ends the process)
                rc = j->fn(item.second);
                if (rc != QV_OK) j->msg[item.second] = qv_last_error();        // the message is thread-local: carry it to the caller
            } catch (const std::bad_alloc&) { rc = QV_ERR_OOM; }
            catch (...) { rc = QV_ERR_DEVICE; }
            j->rc[item.second] = rc;
            j->remaining.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
    void shutdown() {
        { std::lock_guard<std::mutex> l(m); stop = true; }
        cv.notify_all();
        for (auto& t : th) t.join();
        th.clear();
    }
};

}  // namespace

struct qv_sharded {
    ShardPool pool;
    uint32_t dim = 0; int metric = 0; uint64_t flags = 0;
    uint32_t span = 0;
    std::vector<Shard> sh;
    bool rccl = true;
    std::shared_mutex mu;                // searches shared, mutations exclusive (collection.go:647 RLock / :139 Lock)
    std::mutex ctx_mu;
    std::vector<CallCtx*> free_ctx, all_ctx, retired;   // retired: surplus contexts waiting to be freed (release_ctx)
    std::mutex exch_mu;                  // RCCL: one group of collectives is enqueued at a time on the communicators
    Buf d_bases;
    std::atomic<bool> profiling{false};
    std::mutex prof_mu;
    double prof_scan_ms = 0, prof_exchange_ms = 0, prof_merge_ms = 0; uint64_t prof_n = 0;
    qvco::Front front{1, 256, 4};           // concurrent single-query callers share passes, as on one index (qv_coalesce.h): every shard's scan is HBM-bound
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

// fn(g) for every shard: shard 0 on this thread, the others on the pool's workers (or all here when the handle has no pool).
// Returns the first failure with its message.
int for_shards(qv_sharded* s, const std::function<int(uint32_t)>& fn) {
    const uint32_t G = (uint32_t)s->sh.size();
    if (s->pool.th.empty()) {
        for (uint32_t g = 0; g < G; g++) { const int rc = fn(g); if (rc != QV_OK) return rc; }
        return QV_OK;
    }
    ShardJob job;
    job.fn = fn; job.rc.assign(G, QV_OK); job.msg.resize(G);
    job.remaining.store(G - 1, std::memory_order_relaxed);
    {
        std::lock_guard<std::mutex> l(s->pool.m);
        for (uint32_t g = 1; g < G; g++) s->pool.q.emplace_back(&job, g);
    }
    s->pool.cv.notify_all();
    job.rc[0] = fn(0);
    if (job.rc[0] != QV_OK) job.msg[0] = qv_last_error();
    // the workers' shares are a handful of enqueues each: spin briefly, then give the core away between looks (a wedged device call in a
    // worker must not pin this thread at 100 %)
    for (uint32_t spin = 0; job.remaining.load(std::memory_order_acquire) != 0; spin++) {
        if (spin < 4096) __builtin_ia32_pause();
        else std::this_thread::sleep_for(std::chrono::microseconds(spin < 100000 ? 20 : 1000));
    }
    for (uint32_t g = 0; g < G; g++)
        if (job.rc[g] != QV_OK) return fail(job.rc[g], "%s", job.msg[g].empty() ? (job.rc[g] == QV_ERR_OOM ? "out of host memory" : "shard task failed") : job.msg[g].c_str());
    return QV_OK;
}

uint64_t total_live(const qv_sharded* s) { uint64_t t = 0; for (auto& x : s->sh) t += qv_index_size(x.idx); return t; }

int acquire_ctx(qv_sharded* s, CallCtx** out) {
    {
        std::lock_guard<std::mutex> g(s->ctx_mu);
        if (!s->free_ctx.empty()) { *out = s->free_ctx.back(); s->free_ctx.pop_back(); return QV_OK; }
        // a parked surplus context is as good as a new one: a read-only workload with repeated bursts of more than kKeepCtx callers
        // therefore never holds more contexts than its largest burst had callers (nothing reaps `retired` until the next mutation)
        if (!s->retired.empty()) { *out = s->retired.back(); s->retired.pop_back(); return QV_OK; }
    }
    CallCtx* c = new (std::nothrow) CallCtx();
    if (!c) return fail(QV_ERR_OOM, "out of host memory");
    c->sh.resize(s->sh.size());
    hipError_t e = hipSuccess;
    for (size_t g = 0; g < s->sh.size() && e == hipSuccess; g++) {
        e = hipSetDevice(s->sh[g].device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->sh[g].stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->sh[g].ev_done, hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipSetDevice(s->sh[0].device);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = hipEventCreate(&c->ev2);
    if (e == hipSuccess) e = hipEventCreate(&c->ev3);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_merged, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(c->ev_merged, c->sh[0].stream);
    std::lock_guard<std::mutex> g(s->ctx_mu);
    s->all_ctx.push_back(c);                                              // destroyed with the handle whatever happened above
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "call-context setup failed: %s", hipGetErrorString(e));
    *out = c;
    return QV_OK;
}
void destroy_ctx(qv_sharded* s, CallCtx* c);
// A burst of concurrent searches leaves as many contexts as it had callers, each with streams, staging buffers and — on every
// shard's index — a workspace per stream: beyond kKeepCtx idle contexts the extra ones are destroyed instead of pooled.
constexpr size_t kKeepCtx = 8;
void release_ctx(qv_sharded* s, CallCtx* c) {
    // Surplus contexts are PARKED, not destroyed here: destroying one synchronises its streams and frees device and pinned memory
    // (an implicit device-wide synchronisation on every shard's device) — on the thread of a search, under the handle's shared
    // lock, at the tail of a burst, that stalled every other search in flight.  reap_retired frees them from the next exclusive
    // operation (add / remove / update / reserve hold the handle alone) or with the handle.
    std::lock_guard<std::mutex> g(s->ctx_mu);
    if (s->free_ctx.size() < kKeepCtx) s->free_ctx.push_back(c);
    else s->retired.push_back(c);
}
// (the caller holds the handle exclusively: no search is running)
void reap_retired(qv_sharded* s) {
    std::vector<CallCtx*> dead;
    {
        std::lock_guard<std::mutex> g(s->ctx_mu);
        dead.swap(s->retired);
        for (CallCtx* c : dead) s->all_ctx.erase(std::find(s->all_ctx.begin(), s->all_ctx.end(), c));
    }
    for (CallCtx* c : dead) destroy_ctx(s, c);
}
struct CtxGuard { qv_sharded* s; CallCtx* c; ~CtxGuard() { if (c) release_ctx(s, c); } };

void destroy_ctx(qv_sharded* s, CallCtx* c) {
    for (size_t g = 0; g < c->sh.size(); g++) {
        (void)hipSetDevice(s->sh[g].device);
        if (c->sh[g].stream) { (void)hipStreamSynchronize(c->sh[g].stream); qv_internal_drop_stream_workspace(s->sh[g].idx, c->sh[g].stream); }
        c->sh[g].release();
    }
    (void)hipSetDevice(s->sh[0].device);
    c->h_q.release(); c->h_rows.release(); c->h_dist.release(); c->h_aux.release();
    c->d_out_rows.release(); c->d_out_dist.release(); c->d_aux.release(); c->d_sort.release();
    for (hipEvent_t ev : {c->ev_merged, c->ev_in, c->ev_out, c->ev0, c->ev1, c->ev2, c->ev3}) if (ev) (void)hipEventDestroy(ev);
    delete c;
}

// which shard owns a global row id (ids past the last base belong to the last shard)
inline uint32_t shard_of(const qv_sharded* s, uint32_t global_row) { return std::min(global_row / s->span, (uint32_t)s->sh.size() - 1); }

// What one search does besides the plain top-k.
struct SearchMode {
    const uint32_t* selected = nullptr;  // filtered search: the candidate rows (global ids), n_selected of them
    uint32_t n_selected = 0;
    bool masked = false;
    const float* negative = nullptr;     // search with a negative example (host vector): payload plane = distance(row, negative)
    float* neg_out = nullptr;            // [k] host
};

// One search: nq queries (host block, or device block on the first device), lists of length k, any k.
//   host outputs (rows_out != null): synchronous; count_out[q] = results per query
//   device outputs (d_rows_out != null): enqueued only, unless the batched filter hands queries back (nq >= 9) or profiling is on
int search_ctx(qv_sharded* s, CallCtx* c, const float* queries_host, const float* d_queries_dev0, uint32_t nq, uint32_t k, const SearchMode& mode,
               uint32_t* rows_out, float* dist_out, uint32_t* count_out, uint32_t* d_rows_out, float* d_dist_out) {
    const uint32_t G = (uint32_t)s->sh.size();
    const size_t qbytes = (size_t)nq * s->dim * sizeof(float);
    const bool prof = s->profiling.load();
    int rc;
    Shard& s0 = s->sh[0];
    ShardBufs& c0 = c->sh[0];

    // ---- candidates (filtered search): one bitmap per shard = selection AND live rows
    std::vector<uint64_t> matching(G, 0);
    uint64_t live_total = 0, cand_total = 0;
    for (uint32_t g = 0; g < G; g++) live_total += qv_index_size(s->sh[g].idx);
    if (mode.masked) {
        std::vector<size_t> words(G);
        for (uint32_t g = 0; g < G; g++) {
            words[g] = ((size_t)qv_index_rows(s->sh[g].idx) + 63) / 64;
            HIPCHK(hipSetDevice(s->sh[g].device));
            if ((rc = c->sh[g].h_mask.ensure(std::max<size_t>(words[g] * 8, 8))) || (rc = c->sh[g].d_mask.ensure(std::max<size_t>(words[g] * 8, 8)))) return rc;
            memset(c->sh[g].h_mask.p, 0, words[g] * 8);
        }
        for (uint32_t i = 0; i < mode.n_selected; i++) {
            const uint32_t r = mode.selected[i], g = shard_of(s, r), local = r - s->sh[g].base;
            if (local >= qv_index_rows(s->sh[g].idx)) return fail(QV_ERR_OUT_OF_RANGE, "global row %u is not in shard %u", r, g);
            static_cast<uint64_t*>(c->sh[g].h_mask.p)[local >> 6] |= 1ull << (local & 63);
        }
        for (uint32_t g = 0; g < G; g++) {
            uint64_t* hm = static_cast<uint64_t*>(c->sh[g].h_mask.p);
            const std::vector<uint64_t>& alive = s->sh[g].idx->alive_host;
            for (size_t w = 0; w < words[g]; w++) { hm[w] &= w < alive.size() ? alive[w] : 0; matching[g] += (uint64_t)__builtin_popcountll(hm[w]); }
            cand_total += matching[g];
        }
    } else {
        for (uint32_t g = 0; g < G; g++) matching[g] = qv_index_size(s->sh[g].idx);
        cand_total = live_total;
    }
    const uint64_t kk_total = std::min<uint64_t>(k, cand_total);               // exact.go:109-111 over the whole corpus
    if (count_out) for (uint32_t q = 0; q < nq; q++) count_out[q] = (uint32_t)kk_total;
    if (kk_total == 0) {                                                       // nothing matches: all padding
        if (rows_out) for (size_t i = 0; i < (size_t)nq * k; i++) { rows_out[i] = 0xFFFFFFFFu; dist_out[i] = __builtin_inff(); }
        if (mode.neg_out) for (uint32_t i = 0; i < k; i++) mode.neg_out[i] = __builtin_inff();
        if (d_rows_out) {
            HIPCHK(hipSetDevice(s0.device));
            HIPCHK(hipMemsetAsync(d_rows_out, 0xFF, (size_t)nq * k * 4, c0.stream));
            std::vector<float> inf((size_t)nq * k, __builtin_inff());
            HIPCHK(hipMemcpyAsync(d_dist_out, inf.data(), inf.size() * 4, hipMemcpyHostToDevice, c0.stream));
            HIPCHK(hipStreamSynchronize(c0.stream));
        }
        return QV_OK;
    }
    // list length every shard contributes: k for the wavefront-list merge; for a ranking (k > 64) no more than the fullest shard holds
    const bool ranked = k > (uint32_t)qv::kMaxFusedK;
    uint64_t most = 0;
    for (uint32_t g = 0; g < G; g++) most = std::max(most, matching[g]);
    const uint32_t kcap = ranked ? (uint32_t)std::min<uint64_t>(k, most) : k;
    const uint32_t planes = mode.negative ? 3 : 2;
    const size_t words = (size_t)planes * nq * kcap;                           // per shard, 32-bit words
    if (ranked && (uint64_t)G * kcap > 0xFFFFFF00ull) return fail(QV_ERR_UNSUPPORTED, "ranking of %llu candidates exceeds the merge's key space", (unsigned long long)G * kcap);

    if (queries_host) {
        HIPCHK(hipSetDevice(s0.device));
        const size_t hb = qbytes + (mode.negative ? (size_t)s->dim * sizeof(float) : 0);
        if ((rc = c->h_q.ensure(hb))) return rc;
        memcpy(c->h_q.p, queries_host, qbytes);
        if (mode.negative) memcpy(static_cast<char*>(c->h_q.p) + qbytes, mode.negative, (size_t)s->dim * sizeof(float));
    }
    const size_t up_bytes = qbytes + (mode.negative ? (size_t)s->dim * sizeof(float) : 0);
    // ---- the query block on the first device (device-resident queries): placed by this thread BEFORE the other shards' work is
    // issued, because they wait on the event recorded here (a wait captures the event's state at the call)
    if (!queries_host) {
        HIPCHK(hipSetDevice(s0.device));
        HIPCHK(hipStreamWaitEvent(c0.stream, c->ev_merged, 0));           // this context's previous gather buffer has been consumed
        if ((rc = c0.d_q.ensure(up_bytes))) return rc;
        HIPCHK(hipMemcpyAsync(c0.d_q.p, d_queries_dev0, qbytes, hipMemcpyDeviceToDevice, c0.stream));
        HIPCHK(hipEventRecord(c0.ev_done, c0.stream));
    }
    HIPCHK(hipSetDevice(s0.device));
    if ((rc = c0.d_gath.ensure(words * 4 * G))) return rc;                // (before the shards' tasks: with the point-to-point exchange they all write into it)
    if (prof) HIPCHK(hipEventRecord(c->ev0, c0.stream));
    // ---- per shard, in parallel (for_shards): queries in, scan, hand-backs of the batched filter redone, and — point-to-point
    // exchange — the shard's packed lists on their way to the first device's gather buffer
    rc = for_shards(s, [&](uint32_t g) -> int {
        Shard& x = s->sh[g]; ShardBufs& b = c->sh[g];
        int rc2;
        HIPCHK(hipSetDevice(x.device));
        if (queries_host || g != 0) HIPCHK(hipStreamWaitEvent(b.stream, c->ev_merged, 0));
        if ((rc2 = b.d_q.ensure(up_bytes)) || (rc2 = b.d_pack.ensure(words * 4))) return rc2;
        if (s->rccl && g != 0) { if ((rc2 = b.d_gath.ensure(words * 4 * G))) return rc2; }
        if (queries_host) HIPCHK(hipMemcpyAsync(b.d_q.p, c->h_q.p, up_bytes, hipMemcpyHostToDevice, b.stream));
        else if (g != 0) {                                                 // device-resident queries live on the first device
            HIPCHK(hipStreamWaitEvent(b.stream, c0.ev_done, 0));
            if (x.device == s0.device) HIPCHK(hipMemcpyAsync(b.d_q.p, c0.d_q.p, qbytes, hipMemcpyDeviceToDevice, b.stream));
            else HIPCHK(hipMemcpyPeerAsync(b.d_q.p, x.device, c0.d_q.p, s0.device, qbytes, b.stream));
        }
        if (mode.masked && matching[g]) {
            const size_t mb = (((size_t)qv_index_rows(x.idx) + 63) / 64) * 8;
            HIPCHK(hipMemcpyAsync(b.d_mask.p, b.h_mask.p, mb, hipMemcpyHostToDevice, b.stream));
        }
        // -- scan
        uint32_t* pack = static_cast<uint32_t*>(b.d_pack.p);
        float* pack_dist = reinterpret_cast<float*>(pack + (size_t)nq * kcap);
        const float* dq = static_cast<const float*>(b.d_q.p);
        bool filtered = false;
        if (matching[g] == 0) {                                            // no candidates here: no results (0xFFFFFFFF rows are skipped by the merge)
            HIPCHK(hipMemsetAsync(pack, 0xFF, words * 4, b.stream));
        } else if (mode.masked) {
            if ((rc2 = qv_internal_search_candidates_device(x.idx, dq, nq, kcap, static_cast<const uint64_t*>(b.d_mask.p), matching[g], pack, pack_dist, b.stream))) return rc2;
        } else {
            // batches go through the matrix-core filter + exact re-score where it applies (same results, qv_index_search's own rule);
            // everything else, and whatever the filter declines, through the exact scan
            int rcb = QV_ERR_UNSUPPORTED;
            if (nq >= 9) {                                                  // (declines — QV_ERR_UNSUPPORTED — above kMaxBatchedK results per query)
                if ((rc2 = b.d_flags.ensure((size_t)nq * 4)) || (rc2 = b.h_flags.ensure((size_t)nq * 4))) return rc2;
                rcb = qv_index_search_batched_device(x.idx, dq, nq, kcap, pack, pack_dist, static_cast<uint32_t*>(b.d_flags.p), b.stream);
            }
            if (rcb == QV_OK) filtered = true;
            else if (rcb != QV_ERR_UNSUPPORTED) return rcb;
            else if ((rc2 = qv_index_search_device(x.idx, dq, nq, kcap, pack, pack_dist, b.stream))) return rc2;
        }
        bool redone = !filtered;
        if (filtered) {
            // queries the filter handed back (a candidate buffer that overflowed, a guessed bound that did not hold: rare): the exact scan for those,
            // listed and scanned ON THE DEVICE behind the batch — until round 5 their flags were read on the host here, one synchronisation per shard
            // and batch, and until round 6 still so above 64 results per query
            rc2 = qv_internal_redo_flagged_device(x.idx, dq, nq, kcap, static_cast<const uint32_t*>(b.d_flags.p), pack, pack_dist, b.stream);
            if (rc2 == QV_OK) redone = true;
            else if (rc2 != QV_ERR_UNSUPPORTED) return rc2;
        }
        if (!redone) {                                                     // (a corpus so large that a query's keys leave room for one query at a time: from the host)
            HIPCHK(hipMemcpyAsync(b.h_flags.p, b.d_flags.p, (size_t)nq * 4, hipMemcpyDeviceToHost, b.stream));
            HIPCHK(hipStreamSynchronize(b.stream));
            const uint32_t* fl = static_cast<const uint32_t*>(b.h_flags.p);
            for (uint32_t q = 0; q < nq; q++) {
                if (!fl[q]) continue;
                if ((rc2 = qv_index_search_device(x.idx, dq + (size_t)q * s->dim, 1, kcap, pack + (size_t)q * kcap, pack_dist + (size_t)q * kcap, b.stream))) return rc2;
            }
        }
        if (mode.negative && matching[g]) {                                // hybrid_index.go:536-546: distFunc(vector, negative) for this shard's candidates
            const uint32_t valid = (uint32_t)std::min<uint64_t>(kcap, matching[g]);
            hipError_t e = qv::launch_distance_rows(x.idx->view(), dq + (size_t)nq * s->dim, pack, valid, reinterpret_cast<float*>(pack + (size_t)2 * nq * kcap), b.stream);
            if (e != hipSuccess) return fail(QV_ERR_DEVICE, "distance_rows launch failed: %s", hipGetErrorString(e));
        }
        // -- point-to-point exchange: this shard's lists into the first device's gather buffer
        if (!s->rccl) {
            HIPCHK(hipSetDevice(x.device));
            unsigned char* dst = static_cast<unsigned char*>(c0.d_gath.p) + (size_t)g * words * 4;
            if (x.device == s0.device) HIPCHK(hipMemcpyAsync(dst, b.d_pack.p, words * 4, hipMemcpyDeviceToDevice, b.stream));
            else HIPCHK(hipMemcpyPeerAsync(dst, s0.device, b.d_pack.p, x.device, words * 4, b.stream));
            if (g != 0) HIPCHK(hipEventRecord(b.ev_done, b.stream));
        }
        return QV_OK;
    });
    if (rc != QV_OK) return rc;
    // ---- exchange
    if (prof) { HIPCHK(hipSetDevice(s0.device)); HIPCHK(hipEventRecord(c->ev1, c0.stream)); }
    if (s->rccl) {
        std::lock_guard<std::mutex> l(s->exch_mu);
        NCCLCHK(ncclGroupStart());
        for (uint32_t g = 0; g < G; g++) {
            ncclResult_t r = ncclAllGather(c->sh[g].d_pack.p, c->sh[g].d_gath.p, words, ncclUint32, s->sh[g].comm, c->sh[g].stream);
            if (r != ncclSuccess) { (void)ncclGroupEnd(); return fail(QV_ERR_DEVICE, "ncclAllGather failed: %s", ncclGetErrorString(r)); }
        }
        NCCLCHK(ncclGroupEnd());
    } else {
        HIPCHK(hipSetDevice(s0.device));
        for (uint32_t g = 1; g < G; g++) HIPCHK(hipStreamWaitEvent(c0.stream, c->sh[g].ev_done, 0));
    }
    // ---- merge on the first device
    HIPCHK(hipSetDevice(s0.device));
    if (prof) HIPCHK(hipEventRecord(c->ev2, c0.stream));
    uint32_t* o_rows = d_rows_out; float* o_dist = d_dist_out;
    if (!o_rows) {
        if ((rc = c->d_out_rows.ensure((size_t)nq * k * 4)) || (rc = c->d_out_dist.ensure((size_t)nq * k * 4))) return rc;
        o_rows = static_cast<uint32_t*>(c->d_out_rows.p); o_dist = static_cast<float*>(c->d_out_dist.p);
    }
    const uint32_t* gath = static_cast<const uint32_t*>(c0.d_gath.p);
    const uint32_t* bases = static_cast<const uint32_t*>(s->d_bases.p);
    hipError_t e = hipSuccess;
    if (!ranked) e = qv::launch_merge_shards(gath, bases, G, nq, k, o_rows, o_dist, c0.stream, planes);
    else if (k <= (uint32_t)qv::kMaxSelectK) {
        // 64 < k <= 8192 (max(2k, 30) of the negative-example branches, any-k BatchSearch): every shard SELECTED its kcap best
        // (qv_index_search_device), and the first device selects the k best of the G * kcap gathered keys, all queries at once
        if ((rc = c->d_sort.ensure(qv::merge_select_workspace_bytes(G, nq, kcap, k)))) return rc;
        e = qv::launch_merge_select(gath, bases, G, nq, kcap, planes, k, (uint32_t)kk_total, c->d_sort.p, o_rows, o_dist, c0.stream);
    } else {
        if ((rc = c->d_sort.ensure(qv::merge_ranked_workspace_bytes((uint64_t)G * kcap)))) return rc;
        for (uint32_t q = 0; q < nq && e == hipSuccess; q++)
            e = qv::launch_merge_ranked(gath, bases, G, nq, q, kcap, planes, k, c->d_sort.p, o_rows + (size_t)q * k, o_dist + (size_t)q * k, c0.stream);
    }
    if (e != hipSuccess) return fail(QV_ERR_DEVICE, "merge launch failed: %s", hipGetErrorString(e));
    if (mode.negative) {
        if ((rc = c->d_aux.ensure((size_t)k * 4)) || (rc = c->h_aux.ensure((size_t)k * 4))) return rc;
        e = qv::launch_lookup_payload(gath, bases, G, nq, 0, kcap, planes, 2, o_rows, k, static_cast<float*>(c->d_aux.p), c0.stream);
        if (e != hipSuccess) return fail(QV_ERR_DEVICE, "payload lookup launch failed: %s", hipGetErrorString(e));
        HIPCHK(hipMemcpyAsync(c->h_aux.p, c->d_aux.p, (size_t)k * 4, hipMemcpyDeviceToHost, c0.stream));
    }
    HIPCHK(hipEventRecord(c->ev_merged, c0.stream));
    if (rows_out) {
        if ((rc = c->h_rows.ensure((size_t)nq * k * 4)) || (rc = c->h_dist.ensure((size_t)nq * k * 4))) return rc;
        HIPCHK(hipMemcpyAsync(c->h_rows.p, o_rows, (size_t)nq * k * 4, hipMemcpyDeviceToHost, c0.stream));
        HIPCHK(hipMemcpyAsync(c->h_dist.p, o_dist, (size_t)nq * k * 4, hipMemcpyDeviceToHost, c0.stream));
    }
    if (prof) HIPCHK(hipEventRecord(c->ev3, c0.stream));
    // the first device's stream is behind every shard's contribution (the collective / the copy events), so one sync
    // covers the whole search; the other streams stay ordered for the context's next call by themselves
    if (rows_out || prof) HIPCHK(hipStreamSynchronize(c0.stream));
    if (prof) {
        float a = 0, b = 0, d = 0;
        (void)hipEventElapsedTime(&a, c->ev0, c->ev1); (void)hipEventElapsedTime(&b, c->ev1, c->ev2); (void)hipEventElapsedTime(&d, c->ev2, c->ev3);
        std::lock_guard<std::mutex> l(s->prof_mu);
        s->prof_scan_ms += a; s->prof_exchange_ms += b; s->prof_merge_ms += d; s->prof_n++;
    }
    if (rows_out) {
        memcpy(rows_out, c->h_rows.p, (size_t)nq * k * 4);
        memcpy(dist_out, c->h_dist.p, (size_t)nq * k * 4);
        if (mode.neg_out) memcpy(mode.neg_out, c->h_aux.p, (size_t)k * 4);
    }
    return QV_OK;
}

// argument checks of a host-pointer search in the reference's order (exact.go:96-106), then one context, one search
int search_host(qv_sharded* s, const float* queries, uint32_t nq, uint32_t k, const SearchMode& mode, uint32_t* rows_out, float* dist_out, uint32_t* count_out) {
    if (!s) return fail(QV_ERR_INVALID_ARG, "handle is null");
    if (nq == 0) return QV_OK;
    if (!queries || !count_out) return fail(QV_ERR_INVALID_ARG, "queries/count_out is null");
    std::shared_lock<std::shared_mutex> l(s->mu);
    if (total_live(s) == 0) { for (uint32_t q = 0; q < nq; q++) count_out[q] = 0; return QV_OK; }   // exact.go:96-98
    if (k == 0) return fail(QV_ERR_K_NOT_POSITIVE, "k must be positive");                            // exact.go:104-106
    if (!rows_out || !dist_out) return fail(QV_ERR_INVALID_ARG, "rows_out/dist_out is null");
    CallCtx* c = nullptr;
    int rc = acquire_ctx(s, &c);
    if (rc != QV_OK) return rc;
    CtxGuard guard{s, c};
    return search_ctx(s, c, queries, nullptr, nq, k, mode, rows_out, dist_out, count_out, nullptr, nullptr);
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
        // RCCL must be the build that belongs to the HIP runtime this process runs on: both come from one directory (torch/lib in a
        // PyTorch process, /opt/rocm/lib otherwise).  A pair from two installations is refused before the first collective.
        const std::string hip_lib = lib_of(reinterpret_cast<const void*>(&hipGetDeviceCount)), rccl_lib = lib_of(reinterpret_cast<const void*>(&ncclAllGather));
        const char* allow = getenv("QV_ALLOW_RUNTIME_SKEW");
        if (dir_of(hip_lib) != dir_of(rccl_lib) && !(allow && atoi(allow) == 1))
            rc = fail(QV_ERR_UNSUPPORTED, "HIP runtime %s and RCCL %s come from different installations; load libqv before or after PyTorch consistently "
                      "(QV_ALLOW_RUNTIME_SKEW=1 overrides, QV_SHARDED_PEER_COPY needs no RCCL)", hip_lib.c_str(), rccl_lib.c_str());
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
        if (rc == QV_OK && e != hipSuccess) rc = fail(QV_ERR_DEVICE, "setup on device %d failed: %s", devices[0], hipGetErrorString(e));
    }
    if (rc != QV_OK) { char keep[512]; snprintf(keep, sizeof(keep), "%s", qv_last_error()); qv_sharded_destroy(s); return fail(rc, "%s", keep); }
    // Workers for the per-shard host work of a search: OPT-IN (QV_SHARDED_WORKERS=1, read at create).  Measured with 8 shards
    // co-located on ONE device they do not help (207 against 189 us per search: HIP calls on one device serialise on that device's
    // lock — four caller threads only reach 1.9 x); whether they pay off on 8 distinct devices has to be measured on such a node.
    const char* wk = getenv("QV_SHARDED_WORKERS");
    if (n_devices > 2 && wk && atoi(wk) == 1) s->pool.start((uint32_t)n_devices - 1);
    *out = s;
    return QV_OK;
}

void qv_sharded_destroy(qv_sharded* s) {
    if (!s) return;
    s->pool.shutdown();
    for (CallCtx* c : s->all_ctx) destroy_ctx(s, c);
    for (auto& x : s->sh) {
        (void)hipSetDevice(x.device);
        if (x.comm) (void)ncclCommDestroy(x.comm);
        if (x.idx) qv_index_destroy(x.idx);
    }
    if (!s->sh.empty()) (void)hipSetDevice(s->sh[0].device);
    s->d_bases.release();
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
uint64_t qv_sharded_rows(const qv_sharded* s) { uint64_t t = 0; if (s) for (auto& x : s->sh) t += qv_index_rows(x.idx); return t; }
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
    std::unique_lock<std::shared_mutex> l(s->mu);
    reap_retired(s);
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
    std::unique_lock<std::shared_mutex> l(s->mu);
    reap_retired(s);
    const uint32_t G = (uint32_t)s->sh.size();
    // target fill after the add: everyone at the same level where possible
    std::vector<uint64_t> have(G), give(G, 0);
    for (uint32_t g = 0; g < G; g++) have[g] = qv_index_rows(s->sh[g].idx);
    plan_add(have.data(), G, n, give.data());
    for (uint32_t g = 0; g < G; g++)
        if (have[g] + give[g] > s->span) return fail(QV_ERR_OUT_OF_RANGE, "shard %u is full (%u rows per shard)", g, s->span);
    uint64_t off = 0;
    for (uint32_t g = 0; g < G; g++) {
        if (!give[g]) continue;
        Shard& x = s->sh[g];
        uint32_t first = 0;
        int rc = qv_index_add(x.idx, rows + off * s->dim, (uint32_t)give[g], &first);
        if (rc != QV_OK) {                                               // all-or-nothing (InsertBatch rolls back, hybrid_index.go:175-216):
            char keep[512]; snprintf(keep, sizeof(keep), "%s", qv_last_error());   // the pieces already placed are tombstoned, their ids never handed out
            for (uint32_t h = 0; h < g; h++) {
                if (!give[h]) continue;
                std::vector<uint32_t> undo((size_t)give[h]);
                for (uint64_t i = 0; i < give[h]; i++) undo[(size_t)i] = (uint32_t)(have[h] + i);
                (void)qv_index_remove(s->sh[h].idx, undo.data(), (uint32_t)undo.size());
            }
            return fail(rc, "%s", keep);
        }
        if (global_rows_out) for (uint64_t i = 0; i < give[g]; i++) global_rows_out[off + i] = x.base + first + (uint32_t)i;
        off += give[g];
    }
    return QV_OK;
}

// Benchmark / test helper: n synthetic rows (the generator of qv_index_add_synthetic), shard g taking the contiguous block
// [g*n/G, (g+1)*n/G) of generator rows gen_row0.. — SURVEY.md 8e's contiguous row blocks.
int qv_sharded_add_synthetic(qv_sharded* s, uint64_t seed, uint64_t gen_row0, uint64_t n) {
    if (!s) return fail(QV_ERR_INVALID_ARG, "handle is null");
    std::unique_lock<std::shared_mutex> l(s->mu);
    reap_retired(s);
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

// global rows -> (shard, local row), validated; per[g] lists the local rows, where[i] = (shard, position in per[shard])
static int split_rows(const qv_sharded* s, const uint32_t* global_rows, uint32_t n, std::vector<std::vector<uint32_t>>* per,
                      std::vector<std::pair<uint32_t, uint32_t>>* where) {
    const uint32_t G = (uint32_t)s->sh.size();
    per->assign(G, {});
    if (where) where->resize(n);
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t g = shard_of(s, global_rows[i]);
        const uint32_t local = global_rows[i] - s->sh[g].base;
        if (local >= qv_index_rows(s->sh[g].idx)) return fail(QV_ERR_OUT_OF_RANGE, "global row %u is not in shard %u", global_rows[i], g);
        if (where) (*where)[i] = {g, (uint32_t)(*per)[g].size()};
        (*per)[g].push_back(local);
    }
    return QV_OK;
}

int qv_sharded_remove(qv_sharded* s, const uint32_t* global_rows, uint32_t n) {
    if (!s) return fail(QV_ERR_INVALID_ARG, "handle is null");
    if (n == 0) return QV_OK;
    if (!global_rows) return fail(QV_ERR_INVALID_ARG, "rows is null");
    std::unique_lock<std::shared_mutex> l(s->mu);
    reap_retired(s);
    std::vector<std::vector<uint32_t>> per;
    int rc = split_rows(s, global_rows, n, &per, nullptr);
    if (rc != QV_OK) return rc;
    for (size_t g = 0; g < per.size(); g++)
        if (!per[g].empty() && (rc = qv_index_remove(s->sh[g].idx, per[g].data(), (uint32_t)per[g].size())) != QV_OK) return rc;
    return QV_OK;
}

int qv_sharded_update(qv_sharded* s, uint32_t global_row, const float* vec) {
    if (!s) return fail(QV_ERR_INVALID_ARG, "handle is null");
    if (!vec) return fail(QV_ERR_INVALID_ARG, "vector is null");
    std::unique_lock<std::shared_mutex> l(s->mu);
    reap_retired(s);
    const uint32_t g = shard_of(s, global_row), local = global_row - s->sh[g].base;
    if (local >= qv_index_rows(s->sh[g].idx)) return fail(QV_ERR_OUT_OF_RANGE, "global row %u is not in shard %u", global_row, g);
    return qv_index_update(s->sh[g].idx, local, vec);
}

int qv_sharded_get_rows(qv_sharded* s, const uint32_t* global_rows, uint32_t n, float* out) {
    if (!s) return fail(QV_ERR_INVALID_ARG, "handle is null");
    if (n == 0) return QV_OK;
    if (!global_rows || !out) return fail(QV_ERR_INVALID_ARG, "rows/out is null");
    std::shared_lock<std::shared_mutex> l(s->mu);
    std::vector<std::vector<uint32_t>> per;
    std::vector<std::pair<uint32_t, uint32_t>> where;
    int rc = split_rows(s, global_rows, n, &per, &where);
    if (rc != QV_OK) return rc;
    std::vector<std::vector<float>> got(per.size());
    for (size_t g = 0; g < per.size(); g++) {
        if (per[g].empty()) continue;
        got[g].resize(per[g].size() * (size_t)s->dim);
        if ((rc = qv_index_get_rows(s->sh[g].idx, per[g].data(), (uint32_t)per[g].size(), got[g].data())) != QV_OK) return rc;
    }
    for (uint32_t i = 0; i < n; i++)
        memcpy(out + (size_t)i * s->dim, got[where[i].first].data() + (size_t)where[i].second * s->dim, (size_t)s->dim * sizeof(float));
    return QV_OK;
}

int qv_sharded_get_row(qv_sharded* s, uint32_t global_row, float* vec_out) { return qv_sharded_get_rows(s, &global_row, 1, vec_out); }

int qv_sharded_search(qv_sharded* s, const float* queries, uint32_t nq, uint32_t k, uint32_t* rows_out, float* dist_out, uint32_t* count_out) {
    // small calls ride the next pass together (qv_index_search's rule, qv_coalesce.h; one pass at a time here, the shards' contexts
    // being the handle's); everything else, and anything that must fail a check in the reference's order, runs as it is
    const bool share = s && queries && rows_out && dist_out && count_out && nq >= 1 && nq <= 8 && k >= 1 && k <= (uint32_t)qv::kMaxFusedK &&
                       !s->profiling.load() && qv_sharded_rows(s) > 0;
    if (!share) return search_host(s, queries, nq, k, SearchMode{}, rows_out, dist_out, count_out);   // entries past count: row 0xFFFFFFFF, +inf
    char err[256]; err[0] = 0;
    const int rc = s->front.submit(
        0, queries, nq, s->dim, k, rows_out, dist_out, count_out, nullptr,
        [&] { return search_host(s, queries, nq, k, SearchMode{}, rows_out, dist_out, count_out); },
        [&](qvco::Group& g, auto&) {
            g.size_outputs(false);
            return search_host(s, g.queries(), g.nq, g.kmax, SearchMode{}, g.rows.data(), g.dist.data(), g.count.data());
        },
        [] { return qv_last_error(); }, err, sizeof(err));
    if (rc != QV_OK && err[0]) return fail(rc, "%s", err);
    return rc;
}

int qv_sharded_search_masked(qv_sharded* s, const float* queries, uint32_t nq, uint32_t k, const uint32_t* selected_global_rows, uint32_t n_selected,
                             uint32_t* rows_out, float* dist_out, uint32_t* count_out) {
    if (n_selected && !selected_global_rows) return fail(QV_ERR_INVALID_ARG, "selected rows is null");
    SearchMode m; m.masked = true; m.selected = selected_global_rows; m.n_selected = n_selected;
    return search_host(s, queries, nq, k, m, rows_out, dist_out, count_out);
}

int qv_sharded_search_negative(qv_sharded* s, const float* query, const float* negative, uint32_t k_fetch,
                               uint32_t* rows_out, float* dist_out, float* neg_dist_out, uint32_t* count_out) {
    if (!negative || (k_fetch && !neg_dist_out)) return fail(QV_ERR_INVALID_ARG, "negative/neg_dist_out is null");
    SearchMode m; m.negative = negative; m.neg_out = neg_dist_out;
    return search_host(s, query, 1, k_fetch, m, rows_out, dist_out, count_out);
}

// distance of one query to n listed rows (global ids): every shard evaluates its own rows on its own stream, all in flight together
int qv_sharded_distance_rows(qv_sharded* s, const float* query, const uint32_t* global_rows, uint32_t n, float* dist_out) {
    if (!s) return fail(QV_ERR_INVALID_ARG, "handle is null");
    if (n == 0) return QV_OK;
    if (!query || !global_rows || !dist_out) return fail(QV_ERR_INVALID_ARG, "query/rows/dist_out is null");
    std::shared_lock<std::shared_mutex> l(s->mu);
    std::vector<std::vector<uint32_t>> per;
    std::vector<std::pair<uint32_t, uint32_t>> where;
    int rc = split_rows(s, global_rows, n, &per, &where);
    if (rc != QV_OK) return rc;
    CallCtx* c = nullptr;
    if ((rc = acquire_ctx(s, &c)) != QV_OK) return rc;
    CtxGuard guard{s, c};
    const size_t qbytes = (size_t)s->dim * sizeof(float);
    HIPCHK(hipSetDevice(s->sh[0].device));
    if ((rc = c->h_q.ensure(qbytes))) return rc;
    memcpy(c->h_q.p, query, qbytes);
    for (size_t g = 0; g < per.size(); g++) {
        if (per[g].empty()) continue;
        Shard& x = s->sh[g]; ShardBufs& b = c->sh[g];
        const size_t ib = per[g].size() * 4;
        HIPCHK(hipSetDevice(x.device));
        if ((rc = b.d_q.ensure(qbytes)) || (rc = b.d_ids.ensure(ib)) || (rc = b.h_ids.ensure(ib)) || (rc = b.d_out.ensure(ib)) || (rc = b.h_out.ensure(ib))) return rc;
        memcpy(b.h_ids.p, per[g].data(), ib);
        HIPCHK(hipMemcpyAsync(b.d_q.p, c->h_q.p, qbytes, hipMemcpyHostToDevice, b.stream));
        HIPCHK(hipMemcpyAsync(b.d_ids.p, b.h_ids.p, ib, hipMemcpyHostToDevice, b.stream));
        hipError_t e = qv::launch_distance_rows(x.idx->view(), static_cast<const float*>(b.d_q.p), static_cast<const uint32_t*>(b.d_ids.p), (uint32_t)per[g].size(),
                                                static_cast<float*>(b.d_out.p), b.stream);
        if (e != hipSuccess) return fail(QV_ERR_DEVICE, "distance_rows launch failed: %s", hipGetErrorString(e));
        HIPCHK(hipMemcpyAsync(b.h_out.p, b.d_out.p, ib, hipMemcpyDeviceToHost, b.stream));
    }
    for (size_t g = 0; g < per.size(); g++)
        if (!per[g].empty()) { HIPCHK(hipSetDevice(s->sh[g].device)); HIPCHK(hipStreamSynchronize(c->sh[g].stream)); }
    for (uint32_t i = 0; i < n; i++) dist_out[i] = static_cast<const float*>(c->sh[where[i].first].h_out.p)[where[i].second];
    return QV_OK;
}

// Queries and results resident on the FIRST device of the handle; enqueues everything — the form bench.py times with HIP
// events.  The caller's `stream` (null = the null stream, as in qv_index_search_device) is ordered before and after the search.
int qv_sharded_search_device(qv_sharded* s, const float* d_queries, uint32_t nq, uint32_t k, uint32_t* d_rows_out, float* d_dist_out, void* stream) {
    if (!s) return fail(QV_ERR_INVALID_ARG, "handle is null");
    if (nq == 0) return QV_OK;
    if (!d_queries || !d_rows_out || !d_dist_out) return fail(QV_ERR_INVALID_ARG, "null device pointer");
    if (k == 0) return fail(QV_ERR_K_NOT_POSITIVE, "k must be positive");
    std::shared_lock<std::shared_mutex> l(s->mu);
    CallCtx* c = nullptr;
    int rc = acquire_ctx(s, &c);
    if (rc != QV_OK) return rc;
    CtxGuard guard{s, c};                                                  // released once everything is enqueued: the context's streams keep its next use in order
    HIPCHK(hipSetDevice(s->sh[0].device));
    hipStream_t cs = static_cast<hipStream_t>(stream);
    hipStream_t s0 = c->sh[0].stream;
    HIPCHK(hipEventRecord(c->ev_in, cs)); HIPCHK(hipStreamWaitEvent(s0, c->ev_in, 0));   // order after the caller's earlier work on its stream
    rc = search_ctx(s, c, nullptr, d_queries, nq, k, SearchMode{}, nullptr, nullptr, nullptr, d_rows_out, d_dist_out);
    if (rc != QV_OK) return rc;
    HIPCHK(hipSetDevice(s->sh[0].device));
    HIPCHK(hipEventRecord(c->ev_out, s0)); HIPCHK(hipStreamWaitEvent(cs, c->ev_out, 0));
    return QV_OK;
}

int qv_sharded_sync(qv_sharded* s) {
    if (!s) return fail(QV_ERR_INVALID_ARG, "handle is null");
    std::vector<CallCtx*> all;
    { std::lock_guard<std::mutex> g(s->ctx_mu); all = s->all_ctx; }
    for (CallCtx* c : all)
        for (size_t g = 0; g < c->sh.size(); g++)
            if (c->sh[g].stream) { HIPCHK(hipSetDevice(s->sh[g].device)); HIPCHK(hipStreamSynchronize(c->sh[g].stream)); }
    return QV_OK;
}

int qv_runtime_info(char* out, size_t cap) {
    if (!out || cap == 0) return fail(QV_ERR_INVALID_ARG, "out is null");
    int hip_v = 0, nccl_v = 0;
    (void)hipRuntimeGetVersion(&hip_v);
    (void)ncclGetVersion(&nccl_v);
    const char* hwq = getenv("GPU_MAX_HW_QUEUES");
    snprintf(out, cap, "hip_runtime=%d.%d.%d lib=%s; rccl=%d.%d.%d lib=%s; GPU_MAX_HW_QUEUES=%s", hip_v / 10000000, (hip_v / 100000) % 100, hip_v % 100000,
             lib_of(reinterpret_cast<const void*>(&hipGetDeviceCount)).c_str(), nccl_v / 10000, (nccl_v / 100) % 100, nccl_v % 100,
             lib_of(reinterpret_cast<const void*>(&ncclAllGather)).c_str(), hwq ? hwq : "unset (runtime default: 4)");
    return QV_OK;
}

int qv_sharded_set_filter(qv_sharded* s, int filter) {
    if (!s) return fail(QV_ERR_INVALID_ARG, "handle is null");
    std::unique_lock<std::shared_mutex> l(s->mu);
    reap_retired(s);
    for (auto& x : s->sh) { const int rc = qv_index_set_filter(x.idx, filter); if (rc != QV_OK) return rc; }
    return QV_OK;
}

int qv_sharded_profile(qv_sharded* s, int enable) {
    if (!s) return fail(QV_ERR_INVALID_ARG, "handle is null");
    std::unique_lock<std::shared_mutex> l(s->mu);
    reap_retired(s);
    s->profiling.store(enable != 0);
    for (auto& x : s->sh) { (void)qv_index_profile(x.idx, enable); double ms; uint64_t n; (void)qv_index_profile_read(x.idx, &ms, &n); }
    std::lock_guard<std::mutex> g(s->prof_mu);
    s->prof_scan_ms = s->prof_exchange_ms = s->prof_merge_ms = 0; s->prof_n = 0;
    return QV_OK;
}

int qv_sharded_profile_read(qv_sharded* s, double* scan_ms, double* exchange_ms, double* merge_ms, uint64_t* searches) {
    if (!s) return fail(QV_ERR_INVALID_ARG, "handle is null");
    std::lock_guard<std::mutex> l(s->prof_mu);
    if (scan_ms) *scan_ms = s->prof_scan_ms;
    if (exchange_ms) *exchange_ms = s->prof_exchange_ms;
    if (merge_ms) *merge_ms = s->prof_merge_ms;
    if (searches) *searches = s->prof_n;
    s->prof_scan_ms = s->prof_exchange_ms = s->prof_merge_ms = 0; s->prof_n = 0;
    return QV_OK;
}

int qv_sharded_profile_read_shard(qv_sharded* s, int shard, double* scan_kernel_ms_sum, uint64_t* launches) {
    if (!s || shard < 0 || shard >= (int)s->sh.size()) return fail(QV_ERR_INVALID_ARG, "shard %d out of range", shard);
    return qv_index_profile_read(s->sh[(size_t)shard].idx, scan_kernel_ms_sum, launches);
}

}  // extern "C"
