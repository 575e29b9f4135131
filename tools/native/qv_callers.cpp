// qv_callers.cpp — measurement / test driver (NOT part of the product library): T native threads, each calling the C ABI's
// host-pointer search with ONE query per call in a closed loop — the traffic the reference's Go host produces
// (Collection.Search under RLock, pkg/core/collection.go:647; DB.BatchSearch's "parallel individual searches",
// pkg/core/db.go:805-828; HNSW.Search under RLock, pkg/hnsw/hnsw.go:602-606).  Python threads cannot generate it: at a few
// hundred thousand calls per second the interpreter lock is the bottleneck, not libqv.
//
// Every call's result is compared with the first result seen for the same query (a query's answer must not depend on who
// shared its pass); the first result per query is handed back so that the caller can check it against the oracle.
#include "../../include/qv.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace {
struct Shared {
    int kind; void* handle; const float* queries; uint32_t n_queries, dim, k, ef;
    uint32_t* rows_out; float* dist_out; uint32_t* count_out;
    std::vector<std::atomic<uint32_t>> seen;       // 0 = no result yet, 1 = being written, 2 = recorded
    std::atomic<uint64_t> calls{0}, mismatches{0}, errors{0};
    std::atomic<int> ready{0}; std::atomic<bool> go{false};
    explicit Shared(uint32_t n) : seen(n) { for (auto& s : seen) s.store(0); }
};

int one_call(Shared& S, const float* q, uint32_t* rows, float* dist, uint32_t* count) {
    switch (S.kind) {
        case 0: return qv_index_search(static_cast<qv_index*>(S.handle), q, 1, S.k, rows, dist, count);
        case 1: return qv_sharded_search(static_cast<qv_sharded*>(S.handle), q, 1, S.k, rows, dist, count);
        default: return qv_graph_search(static_cast<qv_graph*>(S.handle), q, 1, S.k, S.ef, rows, dist, count, nullptr);
    }
}
}  // namespace

extern "C" {

// kind: 0 = qv_index*, 1 = qv_sharded*, 2 = qv_graph* (ef = efSearch).  Thread t starts at query t mod n_queries and walks the
// pool with stride n_threads.  Stops after `seconds` or after max_calls_per_thread calls per thread (0 = no limit), whichever
// comes first.  rows_out / dist_out [n_queries][k], count_out [n_queries]: the first result recorded per query (count 0xFFFFFFFD =
// never asked).  Returns 0, or the first failing call's status.
int qvc_run(int kind, void* handle, const float* queries, uint32_t n_queries, uint32_t dim, uint32_t k, uint32_t ef,
            uint32_t n_threads, double seconds, uint64_t max_calls_per_thread,
            uint32_t* rows_out, float* dist_out, uint32_t* count_out,
            uint64_t* calls_out, uint64_t* mismatches_out, uint64_t* errors_out, double* elapsed_s_out,
            double* lat_p50_us, double* lat_p99_us, double* lat_max_us, char* first_error, size_t first_error_cap) {
    if (!handle || !queries || !n_queries || !n_threads || !k) return QV_ERR_INVALID_ARG;
    Shared S(n_queries);
    S.kind = kind; S.handle = handle; S.queries = queries; S.n_queries = n_queries; S.dim = dim; S.k = k; S.ef = ef;
    S.rows_out = rows_out; S.dist_out = dist_out; S.count_out = count_out;
    for (uint32_t q = 0; q < n_queries; q++) count_out[q] = 0xFFFFFFFDu;
    std::vector<std::vector<float>> lat(n_threads);
    std::atomic<int> first_rc{0};
    std::mutex err_mu;
    if (first_error && first_error_cap) first_error[0] = 0;
    std::vector<std::thread> th;
    std::chrono::steady_clock::time_point t_start;
    for (uint32_t t = 0; t < n_threads; t++)
        th.emplace_back([&, t] {
            std::vector<uint32_t> rows(k); std::vector<float> dist(k); uint32_t count = 0;
            lat[t].reserve(4096);
            S.ready.fetch_add(1);
            while (!S.go.load(std::memory_order_acquire)) std::this_thread::yield();
            const auto deadline = t_start + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(seconds));
            uint64_t n = 0;
            for (uint32_t qi = t % n_queries;; qi = (qi + n_threads) % n_queries) {
                if (max_calls_per_thread && n >= max_calls_per_thread) break;
                const auto t0 = std::chrono::steady_clock::now();
                if (t0 >= deadline) break;
                const int rc = one_call(S, queries + (size_t)qi * dim, rows.data(), dist.data(), &count);
                const auto t1 = std::chrono::steady_clock::now();
                n++;
                if (rc != QV_OK) {
                    S.errors.fetch_add(1);
                    int z = 0;
                    if (first_rc.compare_exchange_strong(z, rc) && first_error) { std::lock_guard<std::mutex> l(err_mu); snprintf(first_error, first_error_cap, "%s", qv_last_error()); }
                    continue;
                }
                lat[t].push_back(std::chrono::duration<float, std::micro>(t1 - t0).count());
                uint32_t z = 0;
                if (S.seen[qi].compare_exchange_strong(z, 1)) {
                    memcpy(rows_out + (size_t)qi * k, rows.data(), (size_t)k * 4);
                    memcpy(dist_out + (size_t)qi * k, dist.data(), (size_t)k * 4);
                    count_out[qi] = count;
                    S.seen[qi].store(2, std::memory_order_release);
                } else if (S.seen[qi].load(std::memory_order_acquire) == 2) {
                    if (count != count_out[qi] || memcmp(rows_out + (size_t)qi * k, rows.data(), (size_t)k * 4) || memcmp(dist_out + (size_t)qi * k, dist.data(), (size_t)k * 4))
                        S.mismatches.fetch_add(1);
                }
            }
            S.calls.fetch_add(n);
        });
    while (S.ready.load() < (int)n_threads) std::this_thread::yield();
    t_start = std::chrono::steady_clock::now();
    S.go.store(true, std::memory_order_release);
    for (auto& x : th) x.join();
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    std::vector<float> all;
    for (auto& v : lat) all.insert(all.end(), v.begin(), v.end());
    std::sort(all.begin(), all.end());
    if (calls_out) *calls_out = S.calls.load();
    if (mismatches_out) *mismatches_out = S.mismatches.load();
    if (errors_out) *errors_out = S.errors.load();
    if (elapsed_s_out) *elapsed_s_out = el;
    if (lat_p50_us) *lat_p50_us = all.empty() ? 0.0 : all[all.size() / 2];
    if (lat_p99_us) *lat_p99_us = all.empty() ? 0.0 : all[std::min(all.size() - 1, all.size() * 99 / 100)];
    if (lat_max_us) *lat_max_us = all.empty() ? 0.0 : all.back();
    return first_rc.load();
}

}  // extern "C"
