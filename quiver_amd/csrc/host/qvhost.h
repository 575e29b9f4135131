// qvhost.h — host-side mirror (C++) of the reference's Go callers of the hot path,
// written ABOVE the C ABI of include/qv.h.  The reference is Go and no Go toolchain
// exists in this image, so the callers are restated in C++ with the same names,
// argument meaning, check order and error wording as the Go packages they stand in for:
//
//   quiver::ExactIndex    pkg/hybrid/exact.go:14-160
//   quiver::HNSW          pkg/hnsw/hnsw.go:58-842 (graph walk on the host; every distance
//                         is a libqv call: one qv_distance_rows batch per searchLayer hop)
//   quiver::HNSWAdapter   pkg/hnsw/adapter.go:15-95, 345-437 + pkg/hybrid/hnsw_adapter.go
//   quiver::HybridIndex   pkg/hybrid/hybrid_index.go:15-811, adaptive.go:41-72, 226-231
//
// No arithmetic on vector data happens here: distances and selections are libqv's
// (HIP) job; this layer owns string ids, check order, graph bookkeeping, re-rank order.
//
// Locking mirrors the reference: every index embeds a sync.RWMutex (exact.go:25, hnsw.go:58,
// hybrid_index.go:25) — searches take it shared and may run concurrently, mutations take it
// exclusively.  Here that is a std::shared_mutex per object; what a search touches besides the
// index (visited stamps, counters, the strategy RNG) is per thread, atomic, or under its own small mutex.
// A flat extern "C" surface (qvh_*) at the bottom lets pytest drive it through ctypes.
#pragma once
#include <atomic>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/qv.h"

namespace quiver {

struct BasicSearchResult {   // pkg/types/search.go:9-14
    std::string id;
    float distance;
};

// error = empty string means nil
using Error = std::string;

// Where an exact index keeps its rows: one GPU (a qv_index) or the GPUs of a node (a qv_sharded handle: one row shard per
// listed device).  The reference has no counterpart (it is one process on CPU cores); the host code above is
// the same either way.  peer_copy: point-to-point exchange instead of the RCCL all-gather (lets shards share a device).
// bf16_rows: also keep the bfloat16 copy of the rows that BatchSearch's filter reads (QV_FLAG_BF16_ROWS, +50 % device memory).
struct Placement {
    std::vector<int> devices{0};
    bool peer_copy = false;
    bool bf16_rows = false;
    Placement() = default;
    Placement(int device) : devices{device} {}                                   // NOLINT: one device is the common case
    Placement(std::vector<int> devs, bool peer = false, bool bf16 = false) : devices(std::move(devs)), peer_copy(peer), bf16_rows(bf16) {}
    bool sharded() const { return devices.size() > 1 || peer_copy; }
    int first() const { return devices.empty() ? 0 : devices[0]; }
};

// The rows behind an ExactIndex: qv_index_* on one device, qv_sharded_* over a device list — the same calls, row ids opaque.
class RowStore {
public:
    ~RowStore() { destroy(); }
    bool live() const { return one_ || many_; }
    int create(uint32_t dim, qv_metric metric, const Placement& where);
    void destroy();
    int add(const float* rows, uint32_t n, uint32_t* rows_out /* [n] */);
    int update(uint32_t row, const float* v);
    int remove(const uint32_t* rows, uint32_t n);
    int search(const float* qs, uint32_t nq, uint32_t k, uint32_t* rows, float* dist, uint32_t* count);
    int search_negative(const float* q, const float* neg, uint32_t k_fetch, uint32_t* rows, float* dist, float* neg_dist, uint32_t* count);
    int distance_rows(const float* q, const uint32_t* rows, uint32_t n, float* out);
    uint64_t rows() const;
private:
    qv_index* one_ = nullptr;
    qv_sharded* many_ = nullptr;
};

class ExactIndex {
public:
    ExactIndex(qv_metric metric, const Placement& where);
    ~ExactIndex();
    Error Insert(const std::string& id, const float* v, uint32_t len);                 // exact.go:38-58
    // n Inserts under one lock and (when no tombstoned row is waiting for reuse) one device copy; all-or-nothing on a
    // duplicate id or a wrong dimension (what HybridIndex.InsertBatch's rollback loop restores, hybrid_index.go:175-192)
    Error InsertMany(const std::vector<std::string>& ids, const float* packed, uint32_t len, std::string* failed_id = nullptr);
    Error Delete(const std::string& id);                                               // exact.go:61-70
    Error Search(const float* q, uint32_t len, int k, std::vector<BasicSearchResult>* out);   // exact.go:92-133
    // nq searches in one device call (what BatchSearch's goroutine fan-out becomes)
    Error SearchMany(const float* qs, uint32_t len, uint32_t nq, int k, std::vector<std::vector<BasicSearchResult>>* out);
    int Size() const { std::shared_lock<std::shared_mutex> l(mu_); return (int)row_of_.size(); }   // exact.go:136-141
    // the retrieveK nearest results of q together with distFunc(vector, negative) for each of them (hybrid_index.go:524-546),
    // in ONE device call (qv_index_search_negative)
    Error SearchWithNegativeDistances(const float* q, const float* neg, uint32_t len, int retrieveK,
                                      std::vector<BasicSearchResult>* out, std::vector<float>* neg_out);
    // distance(vector of `id`, other) for the re-rank loop (hybrid_index.go:543)
    Error DistancesTo(const float* other, uint32_t len, const std::vector<std::string>& ids, std::vector<float>* out);
    bool Has(const std::string& id) const { std::shared_lock<std::shared_mutex> l(mu_); return row_of_.count(id) != 0; }
    int dim() const { std::shared_lock<std::shared_mutex> l(mu_); return dim_; }
    uint32_t DeviceRows() const;                   // rows the device index holds, tombstones included (bounded under churn: rows are reused)

private:
    Error insertLocked(const std::string& id, const float* v, uint32_t len);
    Error deleteLocked(const std::string& id);
    mutable std::shared_mutex mu_;                 // exact.go:25
    qv_metric metric_; Placement where_;
    RowStore h_;
    int dim_ = 0;                                  // 0 until the first insert (exact.go:43-47)
    std::unordered_map<std::string, uint32_t> row_of_;
    std::unordered_map<uint32_t, std::string> id_of_;   // row -> id (row ids are sparse over shards: one id range per shard)
    std::vector<uint32_t> free_rows_;              // tombstoned rows, reused by the next Insert (the reference's map frees the entry, exact.go:65)
};

struct HNSWConfig {          // pkg/hnsw/hnsw.go:27-41; defaults hnsw.go:223-237
    int M = 0, MaxM0 = 0, EfConstruction = 0, EfSearch = 0, MaxLevel = 0;
    uint64_t seed = 1;       // the reference seeds from the wall clock (hnsw.go:248)
};

struct HNSWResult { std::string id; float distance; uint32_t index; };   // hnsw.go:87-95

class HNSW {
public:
    HNSW(qv_metric metric, int device, const HNSWConfig& cfg);
    ~HNSW();
    Error Insert(const std::string& id, const float* v, uint32_t len);                 // hnsw.go:266-334
    // n Inserts connected on the device (qv_graph_insert): the reference connects concurrently (hnsw.go:313-315); the
    // batch is the deterministic form of that — searches against the graph before the batch, links applied in id order.
    // packed = [n][len] row-major.  batch_max = 1 gives the sequential graph.  Available while every node so far came
    // in through InsertBatch (the device graph carries the link distances); otherwise it is a loop of Insert.
    Error InsertBatch(const std::vector<std::string>& ids, const float* packed, uint32_t len, uint32_t batch_max = 16384, uint32_t ramp_div = 16);
    Error Delete(const std::string& id);                                               // hnsw.go:741-842
    Error Search(const float* q, uint32_t len, int k, std::vector<HNSWResult>* out);   // hnsw.go:602-713
    // nq searches walked on the device, one wavefront per query (qv_graph_search); a query whose
    // graph search under-fills (hnsw.go:676) or overflows the device heap is redone by Search()
    Error SearchBatch(const float* qs, uint32_t len, uint32_t nq, int k, std::vector<std::vector<HNSWResult>>* out,
                      std::vector<uint32_t>* evals_out = nullptr);
    // the device call alone (no id strings, no top-up): rows/dist [nq][k], count [nq]; seconds spent in qv_graph_search
    Error SearchBatchRaw(const float* qs, uint32_t len, uint32_t nq, int k, uint32_t* rows, float* dist, uint32_t* count, uint32_t* evals, double* seconds);
    uint32_t DeviceFallbacks() const { return device_fallbacks_; }   // heap overflows redone on the host
    uint32_t TopUps() const { return topups_; }                       // under-filled graph searches completed by an exact scan
    uint32_t Size() const { return size_.load(); }                    // hnsw.go:253-255 atomic.LoadUint32
    // introspection for graph-equality tests
    uint32_t Nodes() const { std::shared_lock<std::shared_mutex> l(mu_); return (uint32_t)nodes_.size(); }
    bool BuiltOnDevice() const { return bg_synced_ && bg_ != nullptr; }
    int NodeLevel(uint32_t n) const { return n < nodes_.size() && nodes_[n].alive ? nodes_[n].level : -1; }
    const std::vector<uint32_t>* Links(uint32_t n, int level) const;
    void EntryPoint(uint32_t* ep, int* lvl) const { *ep = entry_; *lvl = cur_level_; }
    int RandomLevel();                                                                 // hnsw.go:716-738
    uint64_t DistanceCalls() const { return n_calls_; }
    uint64_t DistanceEvals() const { return n_evals_; }
    void SetEfSearch(int ef) { std::unique_lock<std::shared_mutex> l(mu_); if (ef > 0) efS_ = ef; }
    bool IndexOf(const std::string& id, uint32_t* out) const;
    const std::vector<float>& Vector(uint32_t n) const { return nodes_[n].vec; }
    bool Alive(uint32_t n) const { return n < nodes_.size() && nodes_[n].alive; }
    const std::string& IdOf(uint32_t n) const { return nodes_[n].id; }
    // one device call: distance(query, node) for each listed node (takes the read lock)
    Error Distances(const float* query, const std::vector<uint32_t>& nodes, std::vector<float>* out);
    // NodeLevel / Links / Vector / Alive / IdOf / IndexOf / EntryPoint read without locking: introspection for tests and, inside
    // the adapter (a friend), sections that hold the read lock themselves
    friend class HNSWAdapter;

private:
    struct Node { std::string id; std::vector<float> vec; int level = 0; std::vector<std::vector<uint32_t>> conn; bool alive = false; };
    struct Res { float dist; uint32_t idx; };
    // table (optional): distance of the query to EVERY row, computed in one device call — the host-driven walk of a search with k or
    // efSearch above the device traversal's 512 (a filtered Collection.Search asks k = Size()) then makes no device call per hop
    Error searchLayer(const float* q, uint32_t entry, int ef, int level, std::vector<Res>* out, const std::vector<float>* table = nullptr);   // hnsw.go:471-580
    Error connectNode(uint32_t nodeIdx, const float* v, int level, int graphLevel);                // hnsw.go:337-468
    static int selectNeighbors(std::vector<Res>& c, int k);                                        // hnsw.go:583-599
    bool ok(uint32_t i) const { return i < nodes_.size() && nodes_[i].alive; }
    Error insertLocked(const std::string& id, const float* v, uint32_t len);
    Error searchLocked(const float* q, uint32_t len, int k, std::vector<HNSWResult>* out);
    Error ensureIndex(uint32_t len);
    Error distancesLocked(const float* query, const std::vector<uint32_t>& nodes, std::vector<float>* out);
    Error pullGraphFromDevice();                   // qv_graph_export -> nodes_[].conn, entry point

    mutable std::shared_mutex mu_;                 // hnsw.go:58 (the embedded sync.RWMutex)
    std::mutex dg_mu_;                             // the lazily uploaded device copy of the graph
    qv_metric metric_; int device_;
    qv_index* h_ = nullptr; int dim_ = 0;
    int M_, maxM0_, efC_, efS_, maxLevel_;
    std::vector<Node> nodes_;
    std::unordered_map<std::string, uint32_t> by_id_;
    uint32_t entry_ = 0; int cur_level_ = -1; std::atomic<uint32_t> size_{0};
    uint64_t rng_;
    std::atomic<uint64_t> n_calls_{0}, n_evals_{0};
    qv_graph* dg_ = nullptr; std::atomic<bool> dg_dirty_{true}; std::atomic<uint32_t> device_fallbacks_{0}, topups_{0};
    qv_graph* bg_ = nullptr; bool bg_synced_ = true;   // graph under device-side construction (== dg_ while in sync with nodes_)
    Error syncDeviceGraph();
};

class HNSWAdapter {          // pkg/hnsw/adapter.go + pkg/hybrid/hnsw_adapter.go
public:
    HNSWAdapter(qv_metric metric, int device, const HNSWConfig& cfg) : hnsw_(metric, device, cfg) {}
    Error Insert(const std::string& id, const float* v, uint32_t len);                 // hnsw_adapter.go:47-54
    Error InsertBatch(const std::vector<std::string>& ids, const float* packed, uint32_t len);   // n Inserts, connected on the device
    Error Delete(const std::string& id);                                               // hnsw_adapter.go:57-63
    Error Search(const float* q, uint32_t len, int k, std::vector<BasicSearchResult>* out);        // hnsw_adapter.go:66-71 -> adapter.go:41-95
    Error SearchWithNegative(const float* q, uint32_t len, const float* neg, uint32_t neg_len, float w, int k,
                             std::vector<BasicSearchResult>* out);                     // hnsw_adapter.go:75-85 -> adapter.go:345-437
    // nq adapter searches: the graph walks in ONE device call (HNSW::SearchBatch), then the adapter's conversion and fill pass
    Error SearchMany(const float* qs, uint32_t len, uint32_t nq, int k, std::vector<std::vector<BasicSearchResult>>* out);
    int Size() const { return (int)hnsw_.Size(); }
    HNSW& graph() { return hnsw_; }

private:
    Error adapterSearch(const float* q, uint32_t len, int k, std::vector<BasicSearchResult>* out); // adapter.go:41-95
    Error fillPass(const float* q, int k, const std::vector<HNSWResult>& hr, std::vector<BasicSearchResult>* out);   // adapter.go:57-92
    HNSW hnsw_;
    std::atomic<int> dim_{0};
};

struct HybridConfig {        // pkg/hybrid/types.go:27-45
    qv_metric metric = QV_COSINE;
    HNSWConfig hnsw;
    int exact_threshold = 1000;
    double exploration_factor = 0.1;   // adaptive.go:46; tests set 0 (adaptive_test.go:44-104)
    uint64_t seed = 1;
    Placement placement;               // the exact index shards over placement.devices; the HNSW graph lives on the first of them
                                       // (graph traversal is sequentially dependent: "replicas only")
};

class HybridIndex {
public:
    explicit HybridIndex(const HybridConfig& cfg);
    Error Insert(const std::string& id, const float* v, uint32_t len);                 // hybrid_index.go:86-129
    Error InsertBatch(const std::vector<std::string>& ids, const std::vector<const float*>& vecs, const std::vector<uint32_t>& lens);  // :132-242
    Error Delete(const std::string& id);                                               // :245-289
    Error DeleteBatch(const std::vector<std::string>& ids);                            // :292-372
    Error Search(const float* q, uint32_t len, int k, std::vector<BasicSearchResult>* out) { return searchWithStrategy(q, len, k, "", nullptr, 0, 0.5f, false, out, nullptr); }
    // strategy: "", "exact", "hnsw".  neg may be null.  used_out receives the strategy taken.
    Error searchWithStrategy(const float* q, uint32_t len, int k, const std::string& strategy, const float* neg, uint32_t neg_len,
                             float neg_weight, bool has_weight, std::vector<BasicSearchResult>* out, std::string* used_out);   // :473-585
    Error SearchWithRequest(const float* q, uint32_t len, int k, const std::string& force, const float* neg, uint32_t neg_len,
                            float neg_weight, std::vector<BasicSearchResult>* out, std::string* used_out);                    // :383-470
    Error BatchSearch(const float* qs, uint32_t len, uint32_t nq, int k, const std::string& force,
                      std::vector<std::vector<BasicSearchResult>>* out, std::vector<std::string>* used_out);                  // :677-811
    int Size() const { std::shared_lock<std::shared_mutex> l(mu_); return (int)vectors_.size(); }
    std::string SelectStrategy(int vectorCount, int dimension, int k);                 // adaptive.go:41-72
    ExactIndex& exact() { return exact_; }
    HNSWAdapter& hnsw() { return hnsw_; }

private:
    Error searchImpl(const float* q, uint32_t len, int k, const std::string& strategy, const float* neg, uint32_t neg_len,
                     float neg_weight, bool has_weight, std::vector<BasicSearchResult>* out, std::string* used_out);
    void updateThresholds() { exact_threshold_ = vector_count_; dim_threshold_ = avg_dim_; }      // adaptive.go:226-231 (the overwrite quirk)
    mutable std::shared_mutex mu_;                 // hybrid_index.go:25
    std::mutex rng_mu_;                            // the selector draws under the READ lock (adaptive.go:46 uses the locked global source)
    HybridConfig cfg_;
    ExactIndex exact_;
    HNSWAdapter hnsw_;
    std::map<std::string, std::vector<float>> vectors_;   // hybrid_index.go:33 (ordered: deterministic batch order)
    int vector_dim_ = 0;
    int vector_count_ = 0, avg_dim_ = 0;
    std::vector<int> dimensions_;
    int exact_threshold_, dim_threshold_ = 100;
    uint64_t rng_;
};

}  // namespace quiver
