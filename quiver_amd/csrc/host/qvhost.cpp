// qvhost.cpp — see qvhost.h.  Host-side restatement of the reference's Go callers of the
// hot path over libqv's C ABI.  Citations are file:line in the reference tree.
#include <cstdlib>
#include "qvhost.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <chrono>

namespace quiver {

namespace {

std::string fmt(const char* f, ...) {
    char buf[512];
    va_list ap; va_start(ap, f);
    vsnprintf(buf, sizeof(buf), f, ap);
    va_end(ap);
    return std::string(buf);
}
Error qv_err() { return std::string(qv_last_error()); }

inline uint64_t sm64_next(uint64_t* s) {
    uint64_t x = (*s += 0x9E3779B97F4A7C15ull);
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
inline double rng_float64(uint64_t* s) { return (double)(sm64_next(s) >> 11) * (1.0 / 9007199254740992.0); }

}  // namespace

// --------------------------------------------------------------- RowStore ------------

int RowStore::create(uint32_t dim, qv_metric metric, const Placement& w) {
    const uint64_t flags = w.bf16_rows ? QV_FLAG_BF16_ROWS : QV_FLAG_NONE;
    if (!w.sharded()) return qv_index_create(&one_, dim, metric, w.first(), flags);
    return qv_sharded_create(&many_, dim, metric, w.devices.data(), (int)w.devices.size(), flags | (w.peer_copy ? QV_SHARDED_PEER_COPY : 0));
}
void RowStore::destroy() {
    if (one_) qv_index_destroy(one_);
    if (many_) qv_sharded_destroy(many_);
    one_ = nullptr; many_ = nullptr;
}
int RowStore::add(const float* rows, uint32_t n, uint32_t* rows_out) {
    if (many_) return qv_sharded_add(many_, rows, n, rows_out);
    uint32_t first = 0;
    const int rc = qv_index_add(one_, rows, n, &first);
    if (rc == QV_OK) for (uint32_t i = 0; i < n; i++) rows_out[i] = first + i;
    return rc;
}
int RowStore::update(uint32_t row, const float* v) { return many_ ? qv_sharded_update(many_, row, v) : qv_index_update(one_, row, v); }
int RowStore::remove(const uint32_t* rows, uint32_t n) { return many_ ? qv_sharded_remove(many_, rows, n) : qv_index_remove(one_, rows, n); }
int RowStore::search(const float* qs, uint32_t nq, uint32_t k, uint32_t* rows, float* dist, uint32_t* count) {
    if (many_) return qv_sharded_search(many_, qs, nq, k, rows, dist, count);     // batches take the matrix-core filter per shard by themselves
    // nq > 1: the batched entry point (MFMA filter + exact re-score when it pays, exact multi-query scan otherwise)
    return nq > 1 ? qv_index_search_batched(one_, qs, nq, k, rows, dist, count) : qv_index_search(one_, qs, nq, k, rows, dist, count);
}
int RowStore::search_negative(const float* q, const float* neg, uint32_t k_fetch, uint32_t* rows, float* dist, float* neg_dist, uint32_t* count) {
    return many_ ? qv_sharded_search_negative(many_, q, neg, k_fetch, rows, dist, neg_dist, count)
                 : qv_index_search_negative(one_, q, neg, k_fetch, rows, dist, neg_dist, count);
}
int RowStore::distance_rows(const float* q, const uint32_t* rows, uint32_t n, float* out) {
    return many_ ? qv_sharded_distance_rows(many_, q, rows, n, out) : qv_distance_rows(one_, q, rows, n, out);
}
uint64_t RowStore::rows() const { return many_ ? qv_sharded_rows(many_) : (one_ ? qv_index_rows(one_) : 0); }

// =============================================================== ExactIndex ==========

ExactIndex::ExactIndex(qv_metric metric, const Placement& where) : metric_(metric), where_(where) {}
ExactIndex::~ExactIndex() {}

Error ExactIndex::insertLocked(const std::string& id, const float* v, uint32_t len) {
    if (dim_ == 0) {                                                   // exact.go:43-44 dimension lock-in
        if (len == 0) return "vector dimension mismatch: expected >0, got 0";
        if (h_.create(len, metric_, where_) != QV_OK) return qv_err();
        dim_ = (int)len;
    } else if ((int)len != dim_) {
        return fmt("vector dimension mismatch: expected %d, got %u", dim_, len);      // exact.go:45-47
    }
    if (row_of_.count(id)) return fmt("vector with ID %s already exists", id.c_str());   // exact.go:48-50
    uint32_t row = 0;
    if (!free_rows_.empty()) {                                         // a deleted entry's slot is reused (the reference's map frees it, exact.go:65):
        row = free_rows_.back();                                       // storage, HBM and scan time stay bounded under insert/delete churn
        if (h_.update(row, v) != QV_OK) return qv_err();               // overwrites the row in place and marks it live
        free_rows_.pop_back();
    } else if (h_.add(v, 1, &row) != QV_OK) return qv_err();           // copies (exact.go:53-56)
    row_of_[id] = row;
    id_of_[row] = id;
    return "";
}

Error ExactIndex::Insert(const std::string& id, const float* v, uint32_t len) {
    std::unique_lock<std::shared_mutex> l(mu_);                        // exact.go:39-40
    return insertLocked(id, v, len);
}

Error ExactIndex::InsertMany(const std::vector<std::string>& ids, const float* packed, uint32_t len, std::string* failed_id) {
    std::unique_lock<std::shared_mutex> l(mu_);
    if (ids.empty()) return "";
    // every check of every Insert first: nothing is inserted unless all of them pass
    auto failed = [&](const std::string& id, Error e) { if (failed_id) *failed_id = id; return e; };
    if (dim_ == 0 ? len == 0 : (int)len != dim_)
        return failed(ids[0], dim_ == 0 ? Error("vector dimension mismatch: expected >0, got 0") : fmt("vector dimension mismatch: expected %d, got %u", dim_, len));
    {
        std::unordered_map<std::string, int> seen;
        for (auto& id : ids) if (row_of_.count(id) || seen[id]++) return failed(id, fmt("vector with ID %s already exists", id.c_str()));
    }
    size_t i = 0;
    for (; i < ids.size() && (!free_rows_.empty() || dim_ == 0); i++) {    // reuse tombstoned rows first (and create the index on the first insert)
        Error e = insertLocked(ids[i], packed + i * (size_t)len, len);
        if (!e.empty()) { for (size_t j = 0; j < i; j++) (void)deleteLocked(ids[j]); return failed(ids[i], e); }
    }
    if (i < ids.size()) {                                              // the rest: one device copy
        const uint32_t n = (uint32_t)(ids.size() - i);
        std::vector<uint32_t> rows(n);
        if (h_.add(packed + i * (size_t)len, n, rows.data()) != QV_OK) {
            Error e = qv_err();
            for (size_t j = 0; j < i; j++) (void)deleteLocked(ids[j]);
            return failed(ids[i], e);
        }
        for (uint32_t j = 0; j < n; j++) { row_of_[ids[i + j]] = rows[j]; id_of_[rows[j]] = ids[i + j]; }
    }
    return "";
}

Error ExactIndex::deleteLocked(const std::string& id) {
    auto it = row_of_.find(id);
    if (it != row_of_.end()) {                                         // exact.go:65 delete(map, id): absent id is not an error
        uint32_t row = it->second;
        if (h_.remove(&row, 1) != QV_OK) return qv_err();
        id_of_.erase(row);
        row_of_.erase(it);
        free_rows_.push_back(row);
    }
    if (row_of_.empty() && h_.live()) {                                // exact.go:66-68 reset the dimension when empty
        h_.destroy(); dim_ = 0; id_of_.clear(); free_rows_.clear();
    }
    return "";
}

Error ExactIndex::Delete(const std::string& id) {
    std::unique_lock<std::shared_mutex> l(mu_);                        // exact.go:62-63
    return deleteLocked(id);
}

uint32_t ExactIndex::DeviceRows() const { std::shared_lock<std::shared_mutex> l(mu_); return (uint32_t)h_.rows(); }

Error ExactIndex::SearchMany(const float* qs, uint32_t len, uint32_t nq, int k, std::vector<std::vector<BasicSearchResult>>* out) {
    out->assign(nq, {});
    std::shared_lock<std::shared_mutex> l(mu_);                        // exact.go:93-94
    if (row_of_.empty()) return "";                                    // exact.go:96-98
    if (dim_ > 0 && (int)len != dim_) return fmt("query dimension mismatch: expected %d, got %u", dim_, len);   // :100-102
    if (k <= 0) return "k must be positive";                           // :104-106
    uint32_t kk = (uint32_t)std::min<size_t>((size_t)k, row_of_.size());   // :109-111
    std::vector<uint32_t> rows((size_t)nq * kk), count(nq);
    std::vector<float> dist((size_t)nq * kk);
    if (h_.search(qs, nq, kk, rows.data(), dist.data(), count.data()) != QV_OK) return qv_err();
    for (uint32_t q = 0; q < nq; q++) {
        auto& o = (*out)[q];
        o.reserve(count[q]);
        for (uint32_t i = 0; i < count[q]; i++) o.push_back({id_of_.at(rows[(size_t)q * kk + i]), dist[(size_t)q * kk + i]});
    }
    return "";
}

Error ExactIndex::Search(const float* q, uint32_t len, int k, std::vector<BasicSearchResult>* out) {
    std::vector<std::vector<BasicSearchResult>> many;
    Error e = SearchMany(q, len, 1, k, &many);
    if (!e.empty()) return e;
    *out = std::move(many[0]);
    return "";
}

Error ExactIndex::SearchWithNegativeDistances(const float* q, const float* neg, uint32_t len, int retrieveK,
                                              std::vector<BasicSearchResult>* out, std::vector<float>* neg_out) {
    out->clear(); neg_out->clear();
    std::shared_lock<std::shared_mutex> l(mu_);
    if (row_of_.empty()) return "";                                    // exact.go:96-98
    if (dim_ > 0 && (int)len != dim_) return fmt("query dimension mismatch: expected %d, got %u", dim_, len);   // :100-102
    if (retrieveK <= 0) return "k must be positive";                   // :104-106
    const uint32_t kk = (uint32_t)std::min<size_t>((size_t)retrieveK, row_of_.size());
    std::vector<uint32_t> rows(kk); std::vector<float> d(kk), nd(kk); uint32_t cnt = 0;
    if (h_.search_negative(q, neg, kk, rows.data(), d.data(), nd.data(), &cnt) != QV_OK) return qv_err();
    for (uint32_t i = 0; i < cnt; i++) { out->push_back({id_of_.at(rows[i]), d[i]}); neg_out->push_back(nd[i]); }
    return "";
}

Error ExactIndex::DistancesTo(const float* other, uint32_t len, const std::vector<std::string>& ids, std::vector<float>* out) {
    out->assign(ids.size(), 0.f);
    if (ids.empty()) return "";
    std::shared_lock<std::shared_mutex> l(mu_);
    if ((int)len != dim_) return fmt("negative example dimension mismatch: expected %d, got %u", dim_, len);
    std::vector<uint32_t> rows(ids.size());
    for (size_t i = 0; i < ids.size(); i++) {
        auto it = row_of_.find(ids[i]);
        if (it == row_of_.end()) return fmt("vector with ID %s not found", ids[i].c_str());
        rows[i] = it->second;
    }
    // distFunc(vector, negExample) (hybrid_index.go:543): every metric here is symmetric in its
    // value AND in its bits (products and |differences| commute; sqrt(ma)*sqrt(mb) commutes)
    if (h_.distance_rows(other, rows.data(), (uint32_t)rows.size(), out->data()) != QV_OK) return qv_err();
    return "";
}

// =============================================================== HNSW =================

HNSW::HNSW(qv_metric metric, int device, const HNSWConfig& c) : metric_(metric), device_(device) {
    M_ = c.M > 0 ? c.M : 16;                                           // hnsw.go:223-225
    maxM0_ = c.MaxM0 > 0 ? c.MaxM0 : M_ * 2;                           // :226-228
    efC_ = c.EfConstruction > 0 ? c.EfConstruction : 200;              // :229-231
    efS_ = c.EfSearch > 0 ? c.EfSearch : 100;                          // :232-234
    maxLevel_ = c.MaxLevel > 0 ? c.MaxLevel : 16;                      // :235-237
    rng_ = c.seed;
}
HNSW::~HNSW() { if (dg_ && dg_ != bg_) qv_graph_destroy(dg_); if (bg_) qv_graph_destroy(bg_); if (h_) qv_index_destroy(h_); }

const std::vector<uint32_t>* HNSW::Links(uint32_t n, int level) const {
    if (!ok(n) || level < 0 || level > nodes_[n].level) return nullptr;
    return &nodes_[n].conn[level];
}
bool HNSW::IndexOf(const std::string& id, uint32_t* out) const {
    auto it = by_id_.find(id);
    if (it == by_id_.end()) return false;
    *out = it->second; return true;
}

int HNSW::RandomLevel() {                                              // hnsw.go:716-738
    int level = 0;
    int maxAttempts = std::min(maxLevel_, 10);
    for (int i = 0; i < maxAttempts; i++) { if (rng_float64(&rng_) < 0.25) level++; else break; }
    if (level >= maxLevel_) level = maxLevel_ - 1;
    return level;
}

Error HNSW::Distances(const float* query, const std::vector<uint32_t>& nodes, std::vector<float>* out) {
    std::shared_lock<std::shared_mutex> l(mu_);
    return distancesLocked(query, nodes, out);
}
Error HNSW::distancesLocked(const float* query, const std::vector<uint32_t>& nodes, std::vector<float>* out) {
    out->resize(nodes.size());
    if (nodes.empty()) return "";
    n_calls_++; n_evals_ += nodes.size();
    if (qv_distance_rows(h_, query, nodes.data(), (uint32_t)nodes.size(), out->data()) != QV_OK) return qv_err();
    return "";
}

// heaps of hnsw.go:101-196, restated with the reference's own sift loops so that
// equal-distance pops come out in the same order as in Go
namespace {
struct R { float dist; uint32_t idx; };
void min_up(std::vector<R>& rs, int j) { for (;;) { int i = (j - 1) / 2; if (i == j || rs[j].dist >= rs[i].dist) break; std::swap(rs[i], rs[j]); j = i; } }
void min_down(std::vector<R>& rs, int i0, int n) {
    int i = i0;
    for (;;) {
        int j1 = 2 * i + 1;
        if (j1 >= n || j1 < 0) break;
        int j = j1, j2 = j1 + 1;
        if (j2 < n && rs[j2].dist < rs[j1].dist) j = j2;
        if (rs[i].dist <= rs[j].dist) break;
        std::swap(rs[i], rs[j]);
        i = j;
    }
}
void min_push(std::vector<R>& h, R x) { h.push_back(x); min_up(h, (int)h.size() - 1); }
R min_pop(std::vector<R>& h) { int n = (int)h.size() - 1; std::swap(h[0], h[n]); min_down(h, 0, n); R r = h[n]; h.pop_back(); return r; }
void max_up(std::vector<R>& rs, int j) { for (;;) { int i = (j - 1) / 2; if (i == j || rs[j].dist <= rs[i].dist) break; std::swap(rs[i], rs[j]); j = i; } }
void max_down(std::vector<R>& rs, int i0, int n) {
    int i = i0;
    for (;;) {
        int j1 = 2 * i + 1;
        if (j1 >= n || j1 < 0) break;
        int j = j1, j2 = j1 + 1;
        if (j2 < n && rs[j2].dist > rs[j1].dist) j = j2;
        if (rs[i].dist >= rs[j].dist) break;
        std::swap(rs[i], rs[j]);
        i = j;
    }
}
void max_push(std::vector<R>& h, R x) { h.push_back(x); max_up(h, (int)h.size() - 1); }
R max_pop(std::vector<R>& h) { int n = (int)h.size() - 1; std::swap(h[0], h[n]); max_down(h, 0, n); R r = h[n]; h.pop_back(); return r; }
}  // namespace

// searchLayer, hnsw.go:471-580.  The neighbour loop (:536-563) evaluates every unvisited
// neighbour's distance unconditionally and only the heap admission depends on earlier
// neighbours of the hop, so one batched device call per hop followed by the sequential
// admission in adjacency order is exactly the reference's behaviour.
Error HNSW::searchLayer(const float* q, uint32_t entry, int ef, int level, std::vector<Res>* out, const std::vector<float>* table) {
    out->clear();
    if (nodes_.empty()) return "";                                     // :473-475
    if (!ok(entry)) return fmt("invalid entry point ID: %u", entry);   // :478-480
    // the reference takes a cleared map from a sync.Pool per call (:483-488): searches run concurrently under the read
    // lock, so the stamps are per thread (re-zeroed when another index used this thread's array last)
    struct Visited { const void* owner = nullptr; std::vector<uint32_t> stamp; uint32_t epoch = 0; };
    static thread_local Visited tv;
    if (tv.owner != this) { tv.owner = this; std::fill(tv.stamp.begin(), tv.stamp.end(), 0); tv.epoch = 0; }
    if (tv.stamp.size() < nodes_.size()) tv.stamp.resize(nodes_.size() * 2 + 16, 0);
    if (++tv.epoch == 0) { std::fill(tv.stamp.begin(), tv.stamp.end(), 0); tv.epoch = 1; }
    std::vector<uint32_t>& visited_ = tv.stamp; const uint32_t epoch_ = tv.epoch;
    visited_[entry] = epoch_;
    std::vector<uint32_t> batch; std::vector<float> bd;
    batch.push_back(entry);
    auto distances = [&]() -> Error {                                  // the hop's batch: from the table when the caller made one, else one device call
        if (!table) return distancesLocked(q, batch, &bd);
        bd.resize(batch.size());
        for (size_t i = 0; i < batch.size(); i++) bd[i] = (*table)[batch[i]];
        return "";
    };
    Error e = distances();                                             // :492
    if (!e.empty()) return e;
    std::vector<R> cand, res;
    min_push(cand, {bd[0], entry}); max_push(res, {bd[0], entry});     // :498-506
    while (!cand.empty()) {                                            // :509
        R cur = min_pop(cand);                                         // :511
        if ((int)res.size() >= ef && cur.dist > res[0].dist) break;    // :514-516
        if (!ok(cur.idx)) continue;                                    // :519-521
        const Node& nd = nodes_[cur.idx];
        if (level >= (int)nd.conn.size()) continue;                    // :528-531
        batch.clear();
        for (uint32_t c : nd.conn[level]) {                            // :537
            if (!ok(c)) continue;                                      // :539-541
            if (visited_[c] != epoch_) { visited_[c] = epoch_; batch.push_back(c); }   // :543-544
        }
        if (batch.empty()) continue;
        e = distances();                                               // :548, batched
        if (!e.empty()) return e;
        for (size_t i = 0; i < batch.size(); i++) {
            float cd = bd[i];
            if ((int)res.size() < ef || cd < res[0].dist) {            // :553
                min_push(cand, {cd, batch[i]}); max_push(res, {cd, batch[i]});   // :554-555
                if ((int)res.size() > ef) (void)max_pop(res);          // :558-560
            }
        }
    }
    out->resize(res.size());                                           // :566-577
    for (int i = (int)res.size() - 1; i >= 0; i--) { R r = max_pop(res); (*out)[i] = {r.dist, r.idx}; }
    return "";
}

int HNSW::selectNeighbors(std::vector<Res>& c, int k) {                // hnsw.go:583-599
    if (k <= 0 || c.empty()) return 0;
    std::sort(c.begin(), c.end(), [](const Res& a, const Res& b) {
        if (a.dist == b.dist) return a.idx < b.idx;
        if (a.dist < b.dist) return true;
        if (a.dist > b.dist) return false;
        return a.idx < b.idx;                                          // NaN: index order
    });
    return (int)c.size() > k ? k : (int)c.size();
}

Error HNSW::connectNode(uint32_t nodeIdx, const float* v, int level, int graphLevel) {   // hnsw.go:337-468
    if (level >= maxLevel_) level = maxLevel_ - 1;                     // :342-344
    if (nodes_.size() == 1) { entry_ = nodeIdx; cur_level_ = level; return ""; }   // :347-351
    uint32_t entry = entry_;
    if (!ok(entry)) {                                                  // :356-364
        for (uint32_t i = 0; i < nodes_.size(); i++) if (i != nodeIdx && nodes_[i].alive) { entry = i; break; }
    }
    std::vector<Res> buf;
    for (int lc = graphLevel; lc > level; lc--) {                      // :367-380
        if (!ok(entry)) break;
        if (lc >= (int)nodes_[entry].conn.size()) continue;            // :369-371
        Error e = searchLayer(v, entry, 1, lc, &buf);
        if (!e.empty()) return e;
        if (!buf.empty()) entry = buf[0].idx;
    }
    for (int lc = std::min(level, graphLevel); lc >= 0; lc--) {        // :383
        if (!ok(entry)) break;
        Error e = searchLayer(v, entry, efC_, lc, &buf);               // :385
        if (!e.empty()) return e;
        if (buf.empty()) continue;                                     // :391-393
        int maxConn = lc == 0 ? maxM0_ : M_;                           // :395-398
        int nsel = selectNeighbors(buf, std::min(maxConn, (int)buf.size()));   // :401
        for (int i = 0; i < nsel; i++) nodes_[nodeIdx].conn[lc].push_back(buf[i].idx);   // :407-409
        for (int i = 0; i < nsel; i++) {                               // :413 back-links
            uint32_t nb = buf[i].idx;
            if (!ok(nb)) continue;                                     // :415-417
            Node& nbn = nodes_[nb];
            if (lc > nbn.level || lc >= (int)nbn.conn.size()) continue;   // :420-422
            nbn.conn[lc].push_back(nodeIdx);                           // :426
            if ((int)nbn.conn[lc].size() > maxConn) {                  // :429 prune: re-score the whole list from nb
                std::vector<uint32_t> ids;
                for (uint32_t ci : nbn.conn[lc]) if (ok(ci)) ids.push_back(ci);   // :432-436
                std::vector<float> d;
                Error e2 = distancesLocked(nbn.vec.data(), ids, &d);         // :438 computeDistance(neighborNode.Vector, conn.Vector)
                if (!e2.empty()) return e2;
                std::vector<Res> nd(ids.size());
                for (size_t j = 0; j < ids.size(); j++) nd[j] = {d[j], ids[j]};
                int keep = selectNeighbors(nd, maxConn);               // :451
                nbn.conn[lc].clear();                                  // :454-457
                for (int j = 0; j < keep; j++) nbn.conn[lc].push_back(nd[j].idx);
            }
        }
        if (nsel > 0) entry = nodeIdx;                                 // :463-465 (re-enter from the new node itself)
    }
    return "";
}

// the device index behind the nodes (row == node index); created at the first insert, rebuilt when every node is a
// tombstone and a vector of another dimension arrives
Error HNSW::ensureIndex(uint32_t len) {
    if (!h_) {
        if (len == 0) return "vector dimensions do not match";
        if (qv_index_create(&h_, len, metric_, device_, QV_FLAG_ROWMAJOR) != QV_OK) return qv_err();   // row gathers are this index's hot path
        dim_ = (int)len;
    } else if ((int)len != dim_) {
        if (size_ != 0) return "vector dimensions do not match";       // adapter.go:168 ErrDimensionMismatch from the distance func
        // every node is a tombstone: Go slices carry no dimension, so a new one is fine.
        // Rebuild the device index at the new dimension, keeping row == node index.
        if (len == 0) return "vector dimensions do not match";
        if (dg_ && dg_ != bg_) qv_graph_destroy(dg_);
        if (bg_) qv_graph_destroy(bg_);
        dg_ = nullptr; bg_ = nullptr; dg_dirty_ = true; bg_synced_ = false;
        qv_index_destroy(h_); h_ = nullptr;
        if (qv_index_create(&h_, len, metric_, device_, QV_FLAG_ROWMAJOR) != QV_OK) return qv_err();
        dim_ = (int)len;
        if (!nodes_.empty()) {
            std::vector<float> zeros((size_t)nodes_.size() * len, 0.f);
            std::vector<uint32_t> dead(nodes_.size());
            for (uint32_t i = 0; i < nodes_.size(); i++) dead[i] = i;
            uint32_t first = 0;
            if (qv_index_add(h_, zeros.data(), (uint32_t)nodes_.size(), &first) != QV_OK) return qv_err();
            if (qv_index_remove(h_, dead.data(), (uint32_t)dead.size()) != QV_OK) return qv_err();
        }
    }
    return "";
}

Error HNSW::Insert(const std::string& id, const float* v, uint32_t len) {   // hnsw.go:266-334
    std::unique_lock<std::shared_mutex> l(mu_);                        // :267 (the reference drops it before connectNode; one writer here)
    return insertLocked(id, v, len);
}

Error HNSW::insertLocked(const std::string& id, const float* v, uint32_t len) {
    if (by_id_.count(id)) return fmt("vector with ID %s already exists", id.c_str());   // :269-272
    Error ei = ensureIndex(len);
    if (!ei.empty()) return ei;
    dg_dirty_ = true; bg_synced_ = false;                              // the host graph moves on without the device graph's link distances
    int level = RandomLevel();                                         // :275
    int oldLevel = cur_level_;                                         // :276
    uint32_t row = 0;
    if (qv_index_add(h_, v, 1, &row) != QV_OK) return qv_err();        // :281-282 copy; device row == node index
    uint32_t idx = (uint32_t)nodes_.size();
    if (row != idx) return "internal: device row and node index diverged";
    nodes_.emplace_back();
    Node& nd = nodes_.back();
    nd.id = id; nd.vec.assign(v, v + len); nd.level = level; nd.alive = true;
    nd.conn.assign((size_t)level + 1, {});                             // :287-298
    by_id_[id] = idx; size_++;                                         // :301-303
    if (nodes_.size() == 1) { entry_ = 0; cur_level_ = level; return ""; }   // :306-311
    Error e = connectNode(idx, v, level, oldLevel);                    // :315
    if (!e.empty()) {                                                  // :316-322 rollback keeps the slot
        nodes_[idx].alive = false; by_id_.erase(id); size_--;
        (void)qv_index_remove(h_, &idx, 1);
        return e;
    }
    if (level > oldLevel && level > cur_level_) { entry_ = idx; cur_level_ = level; }   // :325-332
    return "";
}

// n Inserts connected on the device.  Same checks and bookkeeping as Insert; the connectNode work of all n nodes is
// qv_graph_insert (searches: the traversal kernels in build mode; links: qv_build.hip), and the adjacency comes back
// with qv_graph_export so that Links / the host-driven Search / Delete keep working on nodes_.
Error HNSW::InsertBatch(const std::vector<std::string>& ids, const float* packed, uint32_t len, uint32_t batch_max, uint32_t ramp_div) {
    std::unique_lock<std::shared_mutex> l(mu_);
    if (ids.empty()) return "";
    bool device_ok = M_ <= 64 && maxM0_ <= 64 && efC_ <= 512 && maxLevel_ <= 64 && (h_ == nullptr || (int)len == dim_);
    if (device_ok && !(bg_synced_ && (bg_ != nullptr || nodes_.empty()))) {
        // the host graph has moved on (Insert / Delete) without the device graph's link distances: upload it as it stands and
        // have the device score its links once (qv_graph_make_buildable); from then on it is extended on the device again
        device_ok = false;
        if (size_ > 0 && syncDeviceGraph().empty() && qv_graph_make_buildable(dg_, (uint32_t)efC_) == QV_OK) {
            if (bg_ && bg_ != dg_) qv_graph_destroy(bg_);
            bg_ = dg_; bg_synced_ = true; device_ok = true;
        }
    }
    if (!device_ok) {                                                  // configurations the device build does not take: plain Inserts
        for (size_t i = 0; i < ids.size(); i++) {
            Error e = insertLocked(ids[i], packed + i * (size_t)len, len);
            if (!e.empty()) return e;
        }
        return "";
    }
    {
        std::unordered_map<std::string, int> seen;
        for (auto& id : ids) if (by_id_.count(id) || seen[id]++) return fmt("vector with ID %s already exists", id.c_str());   // :269-272
    }
    Error e = ensureIndex(len);
    if (!e.empty()) return e;
    const uint32_t n = (uint32_t)ids.size(), first = (uint32_t)nodes_.size();
    // the graph under construction exists BEFORE any row is added: a failed allocation here leaves nothing to roll back
    // (device rows and nodes_ stay in step, so later Inserts keep working)
    if (!bg_ && qv_graph_create_empty(&bg_, h_, std::max<uint32_t>(first + n, 1024u), (uint32_t)M_, (uint32_t)maxM0_, (uint32_t)efC_) != QV_OK) {
        bg_ = nullptr;
        return qv_err();
    }
    uint32_t row0 = 0;
    if (qv_index_add(h_, packed, n, &row0) != QV_OK) return qv_err();  // :281-282 copies; device row == node index
    if (row0 != first) return "internal: device row and node index diverged";
    std::vector<int8_t> levels(n);
    for (uint32_t i = 0; i < n; i++) levels[i] = (int8_t)RandomLevel();    // :275, in node order
    // The device graph always starts from a live node.  When EntryPoint names a deleted node the reference keeps it (Delete finds
    // no replacement, hnsw.go:803-827) and every connectNode / Search walks from the first live node instead (:355-363, :621-629)
    // — which is the entry the uploaded graph carries.  Unless the build itself moves the entry point (a node above CurrentLevel,
    // :325-332), EntryPoint therefore stays what it was, deleted or not.
    uint32_t eff_entry = entry_;
    if (!ok(eff_entry)) for (uint32_t i = 0; i < first; i++) if (nodes_[i].alive) { eff_entry = i; break; }
    const uint32_t entry_before = entry_; const int level_before = cur_level_;
    const bool had_nodes = first > 0;
    if (qv_graph_insert(bg_, first, n, levels.data(), batch_max, ramp_div) != QV_OK) {
        Error ge = qv_err();                                           // :316-322 rollback keeps the slots (tombstones)
        std::vector<uint32_t> rows(n); for (uint32_t i = 0; i < n; i++) rows[i] = first + i;
        (void)qv_index_remove(h_, rows.data(), n);
        for (uint32_t i = 0; i < n; i++) { nodes_.emplace_back(); nodes_.back().alive = false; }
        if (dg_ == bg_) dg_ = nullptr;
        qv_graph_destroy(bg_); bg_ = nullptr; bg_synced_ = false; dg_dirty_ = true;
        return ge;
    }
    for (uint32_t i = 0; i < n; i++) {                                 // :287-303
        nodes_.emplace_back();
        Node& nd = nodes_.back();
        nd.id = ids[i]; nd.vec.assign(packed + i * (size_t)len, packed + (i + 1) * (size_t)len); nd.level = levels[i]; nd.alive = true;
        by_id_[ids[i]] = first + i;
    }
    size_ += n;
    e = pullGraphFromDevice();
    if (!e.empty()) return e;
    if (had_nodes && entry_ == eff_entry && cur_level_ == level_before) entry_ = entry_before;   // the build did not move it
    if (dg_ && dg_ != bg_) qv_graph_destroy(dg_);
    dg_ = bg_; dg_dirty_ = false;                                      // the graph just built is the one SearchBatch walks
    return "";
}

Error HNSW::pullGraphFromDevice() {
    uint32_t n = 0, nb = 0, m0 = 0, m = 0, ep = 0; int lvl = -1;
    if (qv_graph_info(bg_, &n, &nb, &m0, &m, &ep, &lvl) != QV_OK) return qv_err();
    if (n != nodes_.size()) return "internal: device graph and node list diverged";
    std::vector<uint32_t> l0deg(n), l0links((size_t)n * m0), upoff(n), uplinks((size_t)std::max(nb, 1u) * (1 + m));
    if (qv_graph_export(bg_, nullptr, l0deg.data(), l0links.data(), upoff.data(), nb ? uplinks.data() : nullptr) != QV_OK) return qv_err();
    for (uint32_t i = 0; i < n; i++) {
        Node& nd = nodes_[i];
        if (!nd.alive) continue;
        nd.conn.assign((size_t)nd.level + 1, {});
        nd.conn[0].assign(l0links.begin() + (size_t)i * m0, l0links.begin() + (size_t)i * m0 + std::min(l0deg[i], m0));
        for (int l = 1; l <= nd.level; l++) {
            const uint32_t* blk = &uplinks[(size_t)(upoff[i] + (uint32_t)(l - 1)) * (1 + m)];
            nd.conn[l].assign(blk + 1, blk + 1 + std::min(blk[0], m));
        }
    }
    entry_ = ep; cur_level_ = lvl;                                     // :325-332, replayed per batch by qv_graph_insert
    return "";
}

Error HNSW::Delete(const std::string& id) {                            // hnsw.go:741-842
    std::unique_lock<std::shared_mutex> l(mu_);                        // :742-743
    auto it = by_id_.find(id);
    if (it == by_id_.end()) return fmt("vector with ID %s not found", id.c_str());   // :745-749
    uint32_t idx = it->second;
    if (!ok(idx)) return fmt("vector index %u is invalid", idx);       // :752-755
    dg_dirty_ = true; bg_synced_ = false;
    Node& nd = nodes_[idx];
    for (int level = 0; level <= nd.level; level++) {                  // :762
        if (level >= (int)nd.conn.size()) continue;
        std::vector<uint32_t> snap = nd.conn[level];                   // Go ranges over the slice as it was
        for (uint32_t ci : snap) {
            if (!ok(ci)) continue;                                     // :770-772
            Node& cn = nodes_[ci];
            if (level < (int)cn.conn.size()) {                         // :778-787
                auto& l = cn.conn[level];
                l.erase(std::remove(l.begin(), l.end(), idx), l.end());
            }
        }
    }
    if (entry_ == idx) {                                               // :796
        if (nodes_.size() == 1) { entry_ = 0; cur_level_ = -1; }       // :797-800
        else {
            bool found = false;
            for (int level = nd.level; level >= 0 && !found; level--) {   // :805-816
                if (level < (int)nd.conn.size() && !nd.conn[level].empty()) {
                    uint32_t c = nd.conn[level][0];
                    if (ok(c)) { entry_ = c; cur_level_ = level; found = true; }
                }
            }
            if (!found)                                                // :819-827
                for (uint32_t i = 0; i < nodes_.size(); i++) if (i != idx && nodes_[i].alive) { entry_ = i; cur_level_ = nodes_[i].level; break; }
        }
    }
    nd.alive = false;                                                  // :832 tombstone (Nodes[idx] = nil)
    by_id_.erase(it);
    size_--;
    (void)qv_index_remove(h_, &idx, 1);
    return "";
}

Error HNSW::Search(const float* q, uint32_t len, int k, std::vector<HNSWResult>* out) {   // hnsw.go:602-713
    // A graph whose device copy is current (built or extended on the device, or uploaded by an earlier batch) is walked ON the device:
    // the search is a batch of one, and concurrent callers — the reference searches under a read lock, one goroutine per query
    // (hnsw.go:602-606, adapter.go:253-279) — share traversal batches inside libqv (qv_graph_search).  While the copy is stale (the host
    // graph has moved on since) the host drives the walk hop by hop, as before: re-uploading the graph per search would cost more.
    bool device_current;
    {
        std::shared_lock<std::shared_mutex> l(mu_);                    // :603-604
        bool uploaded;
        { std::lock_guard<std::mutex> lk(dg_mu_); uploaded = dg_ != nullptr && !dg_dirty_.load(); }   // dg_ is written under dg_mu_ by whichever SEARCH uploads the copy
        device_current = uploaded && !nodes_.empty() && size_ > 0 && k > 0 && k <= 512 && efS_ <= 512 && M_ <= 64 && maxM0_ <= 64 &&
                         (int)len == dim_;
        if (!device_current) return searchLocked(q, len, k, out);
    }
    std::vector<std::vector<HNSWResult>> one;
    Error e = SearchBatch(q, len, 1, k, &one, nullptr);                // (takes the lock itself; a mutation in between only means a fresh upload)
    out->clear();
    if (e.empty() && !one.empty()) *out = std::move(one[0]);
    return e;
}

Error HNSW::searchLocked(const float* q, uint32_t len, int k, std::vector<HNSWResult>* out) {
    out->clear();
    if (nodes_.empty()) return "";                                     // :606-608
    if (k <= 0) return "k must be positive";                           // :610-612
    if ((int)len != dim_) return "vector dimensions do not match";     // the distance func's ErrDimensionMismatch (adapter.go:106-108)
    if (k > (int)nodes_.size()) k = (int)nodes_.size();                // :615-617
    uint32_t entry = entry_;
    if (!ok(entry)) {                                                  // :621-629
        uint32_t i; for (i = 0; i < nodes_.size(); i++) if (nodes_[i].alive) { entry = i; break; }
        if (i == nodes_.size()) return "";                             // :632-634
    }
    std::vector<Res> buf;
    int ef = std::max(efS_, k);                                        // :660-663
    // A walk this wide (the device traversal stops at 512; a filtered Collection.Search asks k = Size(), collection.go:679-682,
    // adapter.go:41-52) takes thousands of hops: ONE device call for the distance of the query to every row (the same kernel, the
    // same bits as a hop's batch) instead of one ~20 us call per hop — 1M nodes at k = Size(): minutes -> under a second.
    std::vector<float> table;
    const std::vector<float>* tp = nullptr;
    static const bool table_off = getenv("QV_HOST_WALK_TABLE") && atoi(getenv("QV_HOST_WALK_TABLE")) == 0;   // (measurement: one device call per hop, as before round 4)
    if (ef > 512 && nodes_.size() >= 64 && !table_off) {
        std::vector<uint32_t> all(nodes_.size());
        for (uint32_t i = 0; i < all.size(); i++) all[i] = i;
        Error te = distancesLocked(q, all, &table);
        if (!te.empty()) return te;
        tp = &table;
    }
    for (int level = cur_level_; level > 0; level--) {                 // :649-657 (errors swallowed)
        Error e = searchLayer(q, entry, 1, level, &buf, tp);
        if (!e.empty() || buf.empty()) continue;
        entry = buf[0].idx;
    }
    Error e = searchLayer(q, entry, ef, 0, &buf, tp);                  // :664
    if (!e.empty()) return e;
    if ((int)buf.size() > k) buf.resize(k);                            // :670-672
    if ((int)buf.size() < k) {                                         // :676 under-filled: exact top-up
        std::vector<char> have(nodes_.size(), 0);
        for (auto& r : buf) have[r.idx] = 1;
        std::vector<uint32_t> rest;
        for (uint32_t i = 0; i < nodes_.size(); i++) if (nodes_[i].alive && !have[i]) rest.push_back(i);   // :682-688
        std::vector<float> d;
        if (tp) { d.resize(rest.size()); for (size_t i = 0; i < rest.size(); i++) d[i] = table[rest[i]]; }
        else e = distancesLocked(q, rest, &d);                               // :690
        if (!e.empty()) return e;
        for (size_t i = 0; i < rest.size(); i++) buf.push_back({d[i], rest[i]});
        std::sort(buf.begin(), buf.end(), [&](const Res& a, const Res& b) {   // :699-704 (Distance, VectorID)
            if (a.dist == b.dist) return nodes_[a.idx].id < nodes_[b.idx].id;
            if (a.dist < b.dist) return true;
            if (a.dist > b.dist) return false;
            return a.idx < b.idx;
        });
        if ((int)buf.size() > k) buf.resize(k);
    }
    for (auto& r : buf) out->push_back({nodes_[r.idx].id, r.dist, r.idx});
    return "";
}

// flatten the host graph into the arrays qv_graph_create takes and upload it
Error HNSW::syncDeviceGraph() {
    std::lock_guard<std::mutex> lk(dg_mu_);                            // searches hold mu_ shared: one of them uploads, the others wait
    if (dg_ && !dg_dirty_) return "";
    if (M_ > 64 || maxM0_ > 64) return "device traversal unsupported: degree bounds above 64";
    if (dg_) { if (dg_ == bg_) bg_ = nullptr; qv_graph_destroy(dg_); dg_ = nullptr; }
    const uint32_t n = (uint32_t)nodes_.size();
    uint32_t entry = entry_;
    if (!ok(entry)) {                                                  // hnsw.go:621-629
        uint32_t i; for (i = 0; i < n; i++) if (nodes_[i].alive) { entry = i; break; }
        if (i == n) return "graph has no live node";
    }
    std::vector<int8_t> levels(n);
    std::vector<uint32_t> l0deg(n), l0links((size_t)n * maxM0_, 0), upoff(n, 0), uplinks;
    uint32_t blocks = 0;
    for (uint32_t i = 0; i < n; i++) {
        const Node& nd = nodes_[i];
        levels[i] = nd.alive ? (int8_t)nd.level : (int8_t)-1;
        if (!nd.alive) continue;
        if ((int)nd.conn[0].size() > maxM0_) return "level-0 degree exceeds MaxM0";
        l0deg[i] = (uint32_t)nd.conn[0].size();
        std::copy(nd.conn[0].begin(), nd.conn[0].end(), l0links.begin() + (size_t)i * maxM0_);
        if (nd.level >= 1) {
            upoff[i] = blocks;
            for (int l = 1; l <= nd.level; l++) {
                if ((int)nd.conn[l].size() > M_) return "upper-level degree exceeds M";
                uplinks.push_back((uint32_t)nd.conn[l].size());
                for (int j = 0; j < M_; j++) uplinks.push_back(j < (int)nd.conn[l].size() ? nd.conn[l][j] : 0u);
                blocks++;
            }
        }
    }
    if (uplinks.empty()) uplinks.assign((size_t)1 + M_, 0);
    if (qv_graph_create(&dg_, h_, n, levels.data(), (uint32_t)maxM0_, (uint32_t)M_, l0deg.data(), l0links.data(), upoff.data(), uplinks.data(),
                        std::max(blocks, 1u), entry, cur_level_) != QV_OK) return qv_err();
    dg_dirty_ = false;
    return "";
}

Error HNSW::SearchBatchRaw(const float* qs, uint32_t len, uint32_t nq, int k, uint32_t* rows, float* dist, uint32_t* count, uint32_t* evals, double* seconds) {
    if (seconds) *seconds = 0.0;
    std::shared_lock<std::shared_mutex> l(mu_);
    if (nodes_.empty() || nq == 0 || size_ == 0) return "graph is empty";
    if (k <= 0) return "k must be positive";
    if ((int)len != dim_) return "vector dimensions do not match";
    Error e = syncDeviceGraph();
    if (!e.empty()) return e;
    auto t0 = std::chrono::steady_clock::now();
    if (qv_graph_search(dg_, qs, nq, (uint32_t)k, (uint32_t)efS_, rows, dist, count, evals) != QV_OK) return qv_err();
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return "";
}

Error HNSW::SearchBatch(const float* qs, uint32_t len, uint32_t nq, int k, std::vector<std::vector<HNSWResult>>* out, std::vector<uint32_t>* evals_out) {
    out->assign(nq, {});
    if (evals_out) evals_out->assign(nq, 0);
    std::shared_lock<std::shared_mutex> l(mu_);                        // :603-604, once for the batch
    if (nodes_.empty() || nq == 0) return "";                          // hnsw.go:606-608
    if (k <= 0) return "k must be positive";                           // :610-612
    if ((int)len != dim_) return "vector dimensions do not match";
    if (k > (int)nodes_.size()) k = (int)nodes_.size();                // :615-617
    if (size_ == 0) return "";                                         // :632-634
    // configurations the device traversal does not take (k or efSearch above 512, degree bounds above 64) go through the
    // host-driven Search query by query, like a device heap overflow
    const bool on_device = k <= 512 && efS_ <= 512 && M_ <= 64 && maxM0_ <= 64;
    Error e = on_device ? syncDeviceGraph() : Error("");
    if (!e.empty()) return e;
    std::vector<uint32_t> rows((size_t)nq * k), cnt(nq), ev(nq);
    std::vector<float> dist((size_t)nq * k);
    if (!on_device) { std::fill(cnt.begin(), cnt.end(), 0xFFFFFFFFu); }
    else if (qv_graph_search(dg_, qs, nq, (uint32_t)k, (uint32_t)efS_, rows.data(), dist.data(), cnt.data(), ev.data()) != QV_OK) return qv_err();
    std::vector<uint32_t> underfilled;
    for (uint32_t q = 0; q < nq; q++) {
        if (cnt[q] == (uint32_t)k) {
            auto& o = (*out)[q];
            for (int i = 0; i < k; i++) { uint32_t r = rows[(size_t)q * k + i]; o.push_back({nodes_[r].id, dist[(size_t)q * k + i], r}); }
            if (evals_out) (*evals_out)[q] = ev[q];
            n_evals_ += ev[q];
        } else if (cnt[q] != 0xFFFFFFFFu) {
            underfilled.push_back(q);                                  // graph search under-filled: exact top-up (hnsw.go:676-710)
            if (evals_out) (*evals_out)[q] = ev[q];
        } else {                                                       // device heap overflow: host traversal
            device_fallbacks_++;
            e = searchLocked(qs + (size_t)q * len, len, k, &(*out)[q]);
            if (!e.empty()) return e;
        }
    }
    // Top-up = existing results + every other live node, sorted by (Distance, VectorID), first k
    // (hnsw.go:676-710) — i.e. the exact top-k of all live nodes under that order.  One flat-scan
    // call for all under-filled queries; ties at equal distance are re-ordered by id string, and
    // the fetch is widened until the k-th distance's tie group is complete.
    topups_ += (uint32_t)underfilled.size();
    if (!underfilled.empty()) {
        const uint32_t m = (uint32_t)underfilled.size();
        std::vector<float> packed((size_t)m * len);
        for (uint32_t j = 0; j < m; j++) memcpy(&packed[(size_t)j * len], qs + (size_t)underfilled[j] * len, len * sizeof(float));
        uint32_t kk = std::min<uint32_t>(size_, (uint32_t)k + 8);
        for (;;) {
            std::vector<uint32_t> r2((size_t)m * kk), c2(m); std::vector<float> d2((size_t)m * kk);
            if (qv_index_search_batched(h_, packed.data(), m, kk, r2.data(), d2.data(), c2.data()) != QV_OK) return qv_err();
            bool widen = false;
            for (uint32_t j = 0; j < m && !widen; j++)
                if ((uint32_t)k < c2[j] && c2[j] == kk && kk < size_ && d2[(size_t)j * kk + k - 1] == d2[(size_t)j * kk + kk - 1]) widen = true;
            if (widen) { kk = std::min<uint32_t>(size_, kk * 2); continue; }
            for (uint32_t j = 0; j < m; j++) {
                std::vector<HNSWResult> all;
                for (uint32_t i = 0; i < c2[j]; i++) { uint32_t r = r2[(size_t)j * kk + i]; all.push_back({nodes_[r].id, d2[(size_t)j * kk + i], r}); }
                std::stable_sort(all.begin(), all.end(), [](const HNSWResult& a, const HNSWResult& b) {          // :699-704
                    if (a.distance == b.distance) return a.id < b.id;
                    return a.distance < b.distance;
                });
                if ((int)all.size() > k) all.resize(k);
                (*out)[underfilled[j]] = std::move(all);
            }
            break;
        }
    }
    return "";
}

// =============================================================== HNSWAdapter ==========

Error HNSWAdapter::Insert(const std::string& id, const float* v, uint32_t len) {   // hybrid/hnsw_adapter.go:47-54
    int want = 0;
    if (!dim_.compare_exchange_strong(want, (int)len) && (int)len != want) return fmt("vector dimension mismatch: expected %d, got %u", want, len);
    return hnsw_.Insert(id, v, len);
}
Error HNSWAdapter::InsertBatch(const std::vector<std::string>& ids, const float* packed, uint32_t len) {
    if (ids.empty()) return "";
    int want = 0;
    if (!dim_.compare_exchange_strong(want, (int)len) && (int)len != want) return fmt("vector dimension mismatch: expected %d, got %u", want, len);
    return hnsw_.InsertBatch(ids, packed, len);
}
Error HNSWAdapter::Delete(const std::string& id) {                     // hnsw_adapter.go:57-63
    Error e = hnsw_.Delete(id);
    if (e.empty() && Size() == 0) dim_ = 0;
    return e;
}

// adapter.go:57-92: convert the graph results; if they are fewer than k, add every other live node and sort by distance
Error HNSWAdapter::fillPass(const float* q, int k, const std::vector<HNSWResult>& hr, std::vector<BasicSearchResult>* out) {
    out->clear();
    for (auto& r : hr) out->push_back({r.id, r.distance});             // :57-63
    if ((int)out->size() < k) {                                        // :66 second fill pass
        std::shared_lock<std::shared_mutex> l(hnsw_.mu_);
        const uint32_t n = (uint32_t)hnsw_.nodes_.size();
        std::vector<char> have(n, 0);
        for (auto& r : hr) if (r.index < n) have[r.index] = 1;
        std::vector<uint32_t> rest;
        for (uint32_t i = 0; i < n; i++) if (hnsw_.Alive(i) && !have[i]) rest.push_back(i);   // :72-79
        std::vector<float> d;
        Error e = hnsw_.distancesLocked(q, rest, &d);
        if (!e.empty()) return e;
        for (size_t i = 0; i < rest.size(); i++) out->push_back({hnsw_.IdOf(rest[i]), d[i]});
        std::stable_sort(out->begin(), out->end(), [](const BasicSearchResult& a, const BasicSearchResult& b) { return a.distance < b.distance; });   // :88
        if ((int)out->size() > k) out->resize(k);
    }
    return "";
}

Error HNSWAdapter::adapterSearch(const float* q, uint32_t len, int k, std::vector<BasicSearchResult>* out) {   // hnsw/adapter.go:41-95
    out->clear();
    if (k <= 0) return "k must be positive";                           // :42-44
    int searchK = std::min(k, Size());                                 // :47-50
    std::vector<HNSWResult> hr;
    Error e = hnsw_.Search(q, len, searchK, &hr);                      // :52 (searchK == 0 on an empty graph returns empty first, hnsw.go:606)
    if (!e.empty()) return e;
    return fillPass(q, k, hr, out);
}

Error HNSWAdapter::SearchMany(const float* qs, uint32_t len, uint32_t nq, int k, std::vector<std::vector<BasicSearchResult>>* out) {
    out->assign(nq, {});
    const int dim = dim_.load();
    if (dim > 0 && (int)len != dim) return fmt("query dimension mismatch: expected %d, got %u", dim, len);   // hnsw_adapter.go:67-69
    if (k <= 0) return "k must be positive";                           // adapter.go:42-44
    int searchK = std::min(k, Size());                                 // :47-50
    std::vector<std::vector<HNSWResult>> hr;
    if (searchK > 0) {
        Error e = hnsw_.SearchBatch(qs, len, nq, searchK, &hr);        // the graph walks of all nq queries in one device call
        if (!e.empty()) return e;
    } else hr.assign(nq, {});
    for (uint32_t i = 0; i < nq; i++) {
        Error e = fillPass(qs + (size_t)i * len, k, hr[i], &(*out)[i]);
        if (!e.empty()) return e;
    }
    return "";
}

Error HNSWAdapter::Search(const float* q, uint32_t len, int k, std::vector<BasicSearchResult>* out) {   // hnsw_adapter.go:66-71
    const int dim = dim_.load();
    if (dim > 0 && (int)len != dim) return fmt("query dimension mismatch: expected %d, got %u", dim, len);
    return adapterSearch(q, len, k, out);
}

Error HNSWAdapter::SearchWithNegative(const float* q, uint32_t len, const float* neg, uint32_t neg_len, float w, int k,
                                      std::vector<BasicSearchResult>* out) {
    const int dim = dim_.load();
    if (dim > 0) {                                                     // hnsw_adapter.go:76-83
        if ((int)len != dim) return fmt("query dimension mismatch: expected %d, got %u", dim, len);
        if ((int)neg_len != dim) return fmt("negative example dimension mismatch: expected %d, got %u", dim, neg_len);
    }
    out->clear();
    if (k <= 0) return "k must be positive";                           // adapter.go:347-349
    int retrieveK = std::max(2 * k, 30);                               // :353
    if (retrieveK > Size()) retrieveK = Size();                        // :354-356
    std::vector<BasicSearchResult> initial;
    if (retrieveK > 0) {
        Error e = adapterSearch(q, len, retrieveK, &initial);          // :359
        if (!e.empty()) return "initial search failed: " + e;
    }
    if (neg_len == 0 || w <= 0 || (int)initial.size() <= k) {          // :366-372
        if ((int)initial.size() > k) initial.resize(k);
        *out = initial; return "";
    }
    if (w > 1.0f) w = 1.0f;                                            // :375-377
    std::vector<uint32_t> rows; std::vector<size_t> pos;
    std::vector<float> nd;
    {
        std::shared_lock<std::shared_mutex> l(hnsw_.mu_);              // adapter.go:380 RLock over the re-rank loop
        for (size_t i = 0; i < initial.size(); i++) {                  // :387-405
            uint32_t n;
            if (!hnsw_.IndexOf(initial[i].id, &n) || !hnsw_.Alive(n)) continue;
            rows.push_back(n); pos.push_back(i);
        }
        Error e = hnsw_.distancesLocked(neg, rows, &nd);               // :407 DistanceFunc(node.Vector, negativeExample), batched
        if (!e.empty()) return e;
    }
    std::vector<BasicSearchResult> ext;
    for (size_t i = 0; i < rows.size(); i++) {
        float prod = w * nd[i];                                        // :419 float32 arithmetic
        ext.push_back({initial[pos[i]].id, initial[pos[i]].distance - prod});
    }
    std::stable_sort(ext.begin(), ext.end(), [](const BasicSearchResult& a, const BasicSearchResult& b) {   // :422-427
        if (a.distance == b.distance) return a.id < b.id;
        return a.distance < b.distance;
    });
    if ((int)ext.size() > k) ext.resize(k);                            // :430-433
    *out = ext;
    return "";
}

// =============================================================== HybridIndex ==========

static int avg_dim(const std::vector<int>& d) {                        // hybrid_index.go:617-628
    if (d.empty()) return 0;
    long s = 0; for (int x : d) s += x;
    return (int)(s / (long)d.size());
}

HybridIndex::HybridIndex(const HybridConfig& c)
    : cfg_(c), exact_(c.metric, c.placement), hnsw_(c.metric, c.placement.first(), c.hnsw), exact_threshold_(c.exact_threshold), rng_(c.seed) {}

std::string HybridIndex::SelectStrategy(int vectorCount, int dimension, int k) {   // adaptive.go:41-72
    {
        std::lock_guard<std::mutex> g(rng_mu_);                        // searches draw concurrently under the read lock
        if (rng_float64(&rng_) < cfg_.exploration_factor) {            // :46-51 exploration
            if (rng_float64(&rng_) < 0.5) return "exact";
            return "hnsw";
        }
    }
    if (vectorCount < exact_threshold_) return "exact";                // :56-58
    if (dimension > dim_threshold_) {                                  // :61-68
        if (k < 50) return "hnsw";
        return "exact";
    }
    return "hnsw";                                                     // :71
}

Error HybridIndex::Insert(const std::string& id, const float* v, uint32_t len) {   // hybrid_index.go:86-129
    std::unique_lock<std::shared_mutex> l(mu_);                        // :87
    if (vector_dim_ != 0 && (int)len != vector_dim_) return fmt("vector dimension mismatch: expected %d, got %u", vector_dim_, len);   // :88-91
    if (vectors_.count(id)) return fmt("vector with ID %s already exists", id.c_str());   // :92-95
    Error e = exact_.Insert(id, v, len);                               // :103-105
    if (!e.empty()) return e;
    e = hnsw_.Insert(id, v, len);                                      // :107-114
    if (!e.empty()) {
        Error d = exact_.Delete(id);
        if (!d.empty()) return "insert failed (" + e + ") and rollback also failed: " + d;
        return e;
    }
    if (vector_dim_ == 0) vector_dim_ = (int)len;                      // :118-120
    vectors_[id].assign(v, v + len);                                   // :121 (the copy)
    dimensions_.push_back((int)len);
    vector_count_++;
    avg_dim_ = avg_dim(dimensions_);
    updateThresholds();                                                // :125
    return "";
}

Error HybridIndex::InsertBatch(const std::vector<std::string>& ids, const std::vector<const float*>& vecs, const std::vector<uint32_t>& lens) {
    std::unique_lock<std::shared_mutex> l(mu_);                        // :137
    if (ids.empty()) return "";                                        // :133-135
    int verifyDim = vector_dim_;                                       // :138-144
    if (verifyDim == 0) verifyDim = (int)lens[0];
    for (size_t i = 0; i < ids.size(); i++)                            // :145-150
        if ((int)lens[i] != verifyDim) return fmt("vector dimension mismatch: expected %d, got %u", verifyDim, lens[i]);
    for (size_t i = 0; i < ids.size(); i++)                            // :151-156
        if (vectors_.count(ids[i])) return fmt("vector with ID %s already exists", ids[i].c_str());
    // one packed copy of the batch: both indexes take it in ONE device call each
    const uint32_t len = (uint32_t)verifyDim;
    std::vector<float> packed(ids.size() * (size_t)len);
    for (size_t i = 0; i < ids.size(); i++) memcpy(&packed[i * (size_t)len], vecs[i], (size_t)len * sizeof(float));   // :161-172 (the copies)
    // exact inserts, all-or-nothing (:175-192: insert one by one, delete the inserted ones on failure)
    std::string failed;
    Error e = exact_.InsertMany(ids, packed.data(), len, &failed);
    if (!e.empty()) return "batch insert failed at ID " + failed + ": " + e;
    // HNSW inserts, connected on the device; rollback everything on failure (:195-216)
    e = hnsw_.InsertBatch(ids, packed.data(), len);
    if (!e.empty()) {
        for (size_t j = 0; j < ids.size(); j++) { (void)exact_.Delete(ids[j]); (void)hnsw_.Delete(ids[j]); }
        return "batch insert failed: " + e;
    }
    if (vector_dim_ == 0) vector_dim_ = (int)lens[0];                  // :220-225
    for (size_t i = 0; i < ids.size(); i++) { vectors_[ids[i]].assign(vecs[i], vecs[i] + lens[i]); dimensions_.push_back((int)lens[i]); }
    vector_count_ += (int)ids.size();                                  // :233
    avg_dim_ = avg_dim(dimensions_);
    updateThresholds();                                                // :238
    return "";
}

Error HybridIndex::Delete(const std::string& id) {                     // hybrid_index.go:245-289
    std::unique_lock<std::shared_mutex> l(mu_);                        // :246
    if (!vectors_.count(id)) return fmt("vector with ID %s not found", id.c_str());   // :247-250
    Error e = exact_.Delete(id); if (!e.empty()) return e;             // :254-256
    e = hnsw_.Delete(id); if (!e.empty()) return e;                    // :258-260
    vectors_.erase(id);
    if (vector_count_ > 0) vector_count_--;                            // :267-269
    if (!dimensions_.empty()) { dimensions_.pop_back(); avg_dim_ = avg_dim(dimensions_); }   // :272-280
    if (vectors_.empty()) vector_dim_ = 0;                             // :282-284
    updateThresholds();                                                // :286
    return "";
}

Error HybridIndex::DeleteBatch(const std::vector<std::string>& ids) {  // hybrid_index.go:292-372
    if (ids.empty()) return "";
    std::unique_lock<std::shared_mutex> l(mu_);                        // :297
    std::string missing;
    for (auto& id : ids) if (!vectors_.count(id)) { if (!missing.empty()) missing += " "; missing += id; }   // :299-305
    if (!missing.empty()) return "some vectors not found: [" + missing + "]";   // :308-310
    std::string errs;
    for (auto& id : ids) { Error e = exact_.Delete(id); if (!e.empty()) errs += "failed to delete " + id + " from exact index: " + e + " "; }
    for (auto& id : ids) { Error e = hnsw_.Delete(id); if (!e.empty()) errs += "failed to delete " + id + " from HNSW index: " + e + " "; }
    for (auto& id : ids) vectors_.erase(id);
    vector_count_ = std::max(0, vector_count_ - (int)ids.size());      // :339-343
    if (dimensions_.size() > ids.size()) dimensions_.resize(dimensions_.size() - ids.size()); else dimensions_.clear();   // :346-350
    avg_dim_ = avg_dim(dimensions_);
    if (!errs.empty()) return "errors during batch delete: [" + errs + "]";   // :359-362
    if (vectors_.empty()) vector_dim_ = 0;
    updateThresholds();
    return "";
}

Error HybridIndex::searchWithStrategy(const float* q, uint32_t len, int k, const std::string& strategy_in, const float* neg, uint32_t neg_len,
                                      float neg_weight, bool has_weight, std::vector<BasicSearchResult>* out, std::string* used_out) {
    std::shared_lock<std::shared_mutex> l(mu_);                        // :477-478
    return searchImpl(q, len, k, strategy_in, neg, neg_len, neg_weight, has_weight, out, used_out);
}

Error HybridIndex::searchImpl(const float* q, uint32_t len, int k, const std::string& strategy_in, const float* neg, uint32_t neg_len,
                              float neg_weight, bool has_weight, std::vector<BasicSearchResult>* out, std::string* used_out) {
    out->clear();
    if (vector_dim_ > 0 && (int)len != vector_dim_) return fmt("query dimension mismatch: expected %d, got %u", vector_dim_, len);   // :480-482
    std::string strategy = strategy_in;
    if (strategy.empty()) strategy = SelectStrategy(vector_count_, avg_dim_, k);   // :485-488
    if (used_out) *used_out = strategy;
    bool hasNegative = neg != nullptr && neg_len > 0;                  // :495-497
    float negWeight = has_weight ? neg_weight : 0.5f;                  // :492, :500-502
    if (hasNegative && vector_dim_ > 0 && (int)neg_len != vector_dim_) // :503-505
        return fmt("negative example dimension mismatch: expected %d, got %u", vector_dim_, neg_len);

    if (strategy == "exact") {                                         // :515
        int retrieveK = k;
        if (hasNegative) { retrieveK = std::max(2 * k, 30); if (retrieveK > (int)vectors_.size()) retrieveK = (int)vectors_.size(); }   // :517-522
        if (!hasNegative) return exact_.Search(q, len, retrieveK, out);   // :524
        {                                                              // :524-570: the fetch and the negative distances in ONE device call
            std::vector<float> nd_all;
            Error e = exact_.SearchWithNegativeDistances(q, neg, len, retrieveK, out, &nd_all);   // :524 + :543 distFunc(vector, negExample)
            if (!e.empty()) return e;
            std::vector<BasicSearchResult> rr;
            for (size_t i = 0; i < out->size(); i++) {
                if (!vectors_.count((*out)[i].id)) continue;           // :537-540
                float prod = negWeight * nd_all[i];                    // :549 float32
                rr.push_back({(*out)[i].id, (*out)[i].distance - prod});
            }
            if (!rr.empty()) {
                std::stable_sort(rr.begin(), rr.end(), [](const BasicSearchResult& a, const BasicSearchResult& b) {   // :552-557
                    if (a.distance == b.distance) return a.id < b.id;
                    return a.distance < b.distance;
                });
                *out = rr;                                             // :559-562
            }
            if ((int)out->size() > k) out->resize(k);                  // :564-566
        }
        return "";
    }
    if (strategy == "hnsw") {                                          // :572-579
        if (hasNegative) return hnsw_.SearchWithNegative(q, len, neg, neg_len, negWeight, k, out);
        return hnsw_.Search(q, len, k, out);
    }
    return "invalid search strategy: " + strategy;                     // :581
}

Error HybridIndex::SearchWithRequest(const float* q, uint32_t len, int k, const std::string& force, const float* neg, uint32_t neg_len,
                                     float neg_weight, std::vector<BasicSearchResult>* out, std::string* used_out) {   // :383-470
    out->clear();
    std::shared_lock<std::shared_mutex> l(mu_);                        // :388-389
    if (vector_dim_ > 0 && (int)len != vector_dim_) return fmt("query dimension mismatch: expected %d, got %u", vector_dim_, len);   // :392-394
    if (neg_len > 0 && vector_dim_ > 0 && (int)neg_len != vector_dim_)  // :395-397
        return fmt("negative example dimension mismatch: expected %d, got %u", vector_dim_, neg_len);
    if (k <= 0) return "k must be positive";                           // :400-402
    std::string strategy = force.empty() ? SelectStrategy(vector_count_, avg_dim_, k) : force;   // :405-411
    if (neg_len > 0 && neg_weight > 0) return searchImpl(q, len, k, strategy, neg, neg_len, neg_weight, true, out, used_out);   // :417-420
    return searchImpl(q, len, k, strategy, nullptr, 0, 0.5f, false, out, used_out);   // :422
}

Error HybridIndex::BatchSearch(const float* qs, uint32_t len, uint32_t nq, int k, const std::string& force,
                               std::vector<std::vector<BasicSearchResult>>* out, std::vector<std::string>* used_out) {   // :677-811
    out->clear();
    if (nq == 0) return "no queries provided";                         // :678-680
    std::shared_lock<std::shared_mutex> l(mu_);                        // every goroutine's searchWithStrategy takes it shared (:477)
    if (vector_dim_ > 0 && (int)len != vector_dim_) return fmt("query 0 dimension mismatch: expected %d, got %u", vector_dim_, len);   // :707-713
    out->assign(nq, {});
    if (used_out) used_out->assign(nq, "");
    // the reference fans out one goroutine per query (:703-705), each an independent searchWithStrategy; here the
    // queries routed to one strategy go down together: ONE exact-scan call, ONE graph-traversal call
    std::vector<uint32_t> exact_q, hnsw_q;
    for (uint32_t i = 0; i < nq; i++) {
        std::string s = force.empty() ? SelectStrategy(vector_count_, avg_dim_, k) : force;   // :730-738
        if (used_out) (*used_out)[i] = s;
        if (s == "exact") exact_q.push_back(i);
        else if (s == "hnsw") hnsw_q.push_back(i);
        else return fmt("search %u failed: invalid search strategy: ", i) + s;
    }
    auto run = [&](const std::vector<uint32_t>& which, bool exact) -> Error {
        if (which.empty()) return "";
        std::vector<float> packed((size_t)which.size() * len);
        for (size_t j = 0; j < which.size(); j++) memcpy(&packed[j * len], qs + (size_t)which[j] * len, len * sizeof(float));
        std::vector<std::vector<BasicSearchResult>> res;
        Error e = exact ? exact_.SearchMany(packed.data(), len, (uint32_t)which.size(), k, &res)
                        : hnsw_.SearchMany(packed.data(), len, (uint32_t)which.size(), k, &res);
        if (!e.empty()) return fmt("search %u failed: ", which[0]) + e;   // :752-759
        for (size_t j = 0; j < which.size(); j++) (*out)[which[j]] = std::move(res[j]);
        return "";
    };
    Error e = run(hnsw_q, false);
    if (!e.empty()) return e;
    return run(exact_q, true);
}

}  // namespace quiver

// ===================================================================== flat C surface ==
// for ctypes-driven tests; not the drop-in boundary (that is include/qv.h)
using namespace quiver;

namespace {
thread_local std::string g_herr;
int ret(const Error& e) { g_herr = e; return e.empty() ? 0 : -1; }
struct Results { std::vector<BasicSearchResult> r; std::vector<std::vector<BasicSearchResult>> many; std::vector<std::string> used; std::string used1; };
}  // namespace

extern "C" {

const char* qvh_last_error(void) { return g_herr.c_str(); }

void* qvh_results_new(void) { return new Results(); }
void qvh_results_free(void* r) { delete static_cast<Results*>(r); }
int qvh_results_count(void* r) { return (int)static_cast<Results*>(r)->r.size(); }
const char* qvh_results_id(void* r, int i) { return static_cast<Results*>(r)->r[i].id.c_str(); }
float qvh_results_distance(void* r, int i) { return static_cast<Results*>(r)->r[i].distance; }
const char* qvh_results_strategy(void* r) { return static_cast<Results*>(r)->used1.c_str(); }
int qvh_results_many_count(void* r) { return (int)static_cast<Results*>(r)->many.size(); }
int qvh_results_many_len(void* r, int q) { return (int)static_cast<Results*>(r)->many[q].size(); }
const char* qvh_results_many_id(void* r, int q, int i) { return static_cast<Results*>(r)->many[q][i].id.c_str(); }
float qvh_results_many_distance(void* r, int q, int i) { return static_cast<Results*>(r)->many[q][i].distance; }
const char* qvh_results_many_strategy(void* r, int q) { return static_cast<Results*>(r)->used[q].c_str(); }

// ---- ExactIndex
void* qvh_exact_new(int metric, int device) { return new ExactIndex((qv_metric)metric, Placement(device)); }
// the same index over a device list (one row shard per entry); peer_copy: point-to-point exchange (shards may share a device)
void* qvh_exact_new_placed(int metric, const int* devices, int n_devices, int peer_copy, int bf16_rows) {
    return new ExactIndex((qv_metric)metric, Placement(std::vector<int>(devices, devices + n_devices), peer_copy != 0, bf16_rows != 0));
}
void qvh_exact_free(void* p) { delete static_cast<ExactIndex*>(p); }
int qvh_exact_insert(void* p, const char* id, const float* v, uint32_t len) { return ret(static_cast<ExactIndex*>(p)->Insert(id, v, len)); }
int qvh_exact_delete(void* p, const char* id) { return ret(static_cast<ExactIndex*>(p)->Delete(id)); }
int qvh_exact_search(void* p, const float* q, uint32_t len, int k, void* res) { return ret(static_cast<ExactIndex*>(p)->Search(q, len, k, &static_cast<Results*>(res)->r)); }
int qvh_exact_size(void* p) { return static_cast<ExactIndex*>(p)->Size(); }
uint32_t qvh_exact_device_rows(void* p) { return static_cast<ExactIndex*>(p)->DeviceRows(); }

// ---- HNSW
void* qvh_hnsw_new(int metric, int device, int M, int maxM0, int efC, int efS, int maxLevel, uint64_t seed) {
    HNSWConfig c; c.M = M; c.MaxM0 = maxM0; c.EfConstruction = efC; c.EfSearch = efS; c.MaxLevel = maxLevel; c.seed = seed;
    return new HNSW((qv_metric)metric, device, c);
}
void qvh_hnsw_free(void* p) { delete static_cast<HNSW*>(p); }
int qvh_hnsw_insert(void* p, const char* id, const float* v, uint32_t len) { return ret(static_cast<HNSW*>(p)->Insert(id, v, len)); }
int qvh_hnsw_insert_batch(void* p, const char** ids, const float* packed, uint32_t len, uint32_t n, uint32_t batch_max, uint32_t ramp_div) {
    std::vector<std::string> i; for (uint32_t j = 0; j < n; j++) i.push_back(ids[j]);
    return ret(static_cast<HNSW*>(p)->InsertBatch(i, packed, len, batch_max, ramp_div));
}
int qvh_hnsw_built_on_device(void* p) { return static_cast<HNSW*>(p)->BuiltOnDevice() ? 1 : 0; }
int qvh_hnsw_delete(void* p, const char* id) { return ret(static_cast<HNSW*>(p)->Delete(id)); }
int qvh_hnsw_search(void* p, const float* q, uint32_t len, int k, void* res, uint32_t* idx_out) {
    std::vector<HNSWResult> hr;
    Error e = static_cast<HNSW*>(p)->Search(q, len, k, &hr);
    auto& r = static_cast<Results*>(res)->r; r.clear();
    for (size_t i = 0; i < hr.size(); i++) { r.push_back({hr[i].id, hr[i].distance}); if (idx_out) idx_out[i] = hr[i].index; }
    return ret(e);
}
int qvh_hnsw_search_batch(void* p, const float* qs, uint32_t len, uint32_t nq, int k, void* res, uint32_t* idx_out, uint32_t* evals_out) {
    std::vector<std::vector<HNSWResult>> hr; std::vector<uint32_t> ev;
    Error e = static_cast<HNSW*>(p)->SearchBatch(qs, len, nq, k, &hr, &ev);
    auto& many = static_cast<Results*>(res)->many; many.assign(hr.size(), {});
    auto& used = static_cast<Results*>(res)->used; used.assign(hr.size(), "hnsw");
    for (size_t q = 0; q < hr.size(); q++)
        for (size_t i = 0; i < hr[q].size(); i++) {
            many[q].push_back({hr[q][i].id, hr[q][i].distance});
            if (idx_out) idx_out[q * (size_t)k + i] = hr[q][i].index;
        }
    if (evals_out) for (size_t q = 0; q < ev.size(); q++) evals_out[q] = ev[q];
    return ret(e);
}
int qvh_hnsw_search_batch_raw(void* p, const float* qs, uint32_t len, uint32_t nq, int k, uint32_t* rows, float* dist, uint32_t* count, uint32_t* evals, double* seconds) {
    return ret(static_cast<HNSW*>(p)->SearchBatchRaw(qs, len, nq, k, rows, dist, count, evals, seconds));
}
uint32_t qvh_hybrid_exact_device_rows(void* p) { return static_cast<HybridIndex*>(p)->exact().DeviceRows(); }
int qvh_hybrid_hnsw_built_on_device(void* p) { return static_cast<HybridIndex*>(p)->hnsw().graph().BuiltOnDevice() ? 1 : 0; }
uint32_t qvh_hnsw_device_fallbacks(void* p) { return static_cast<HNSW*>(p)->DeviceFallbacks(); }
uint32_t qvh_hnsw_topups(void* p) { return static_cast<HNSW*>(p)->TopUps(); }
uint32_t qvh_hnsw_size(void* p) { return static_cast<HNSW*>(p)->Size(); }
uint32_t qvh_hnsw_nodes(void* p) { return static_cast<HNSW*>(p)->Nodes(); }
int qvh_hnsw_node_level(void* p, uint32_t n) { return static_cast<HNSW*>(p)->NodeLevel(n); }
int qvh_hnsw_links(void* p, uint32_t n, int level, uint32_t* out, uint32_t cap) {
    const std::vector<uint32_t>* l = static_cast<HNSW*>(p)->Links(n, level);
    if (!l) return -1;
    uint32_t c = std::min<uint32_t>((uint32_t)l->size(), cap);
    memcpy(out, l->data(), c * sizeof(uint32_t));
    return (int)c;
}
void qvh_hnsw_entry_point(void* p, uint32_t* ep, int* lvl) { static_cast<HNSW*>(p)->EntryPoint(ep, lvl); }
void qvh_hnsw_set_ef_search(void* p, int ef) { static_cast<HNSW*>(p)->SetEfSearch(ef); }
uint64_t qvh_hnsw_distance_calls(void* p) { return static_cast<HNSW*>(p)->DistanceCalls(); }
uint64_t qvh_hnsw_distance_evals(void* p) { return static_cast<HNSW*>(p)->DistanceEvals(); }

// ---- HNSWAdapter (hybrid's view)
void* qvh_adapter_new(int metric, int device, int M, int maxM0, int efC, int efS, uint64_t seed) {
    HNSWConfig c; c.M = M; c.MaxM0 = maxM0; c.EfConstruction = efC; c.EfSearch = efS; c.seed = seed;
    return new HNSWAdapter((qv_metric)metric, device, c);
}
void qvh_adapter_free(void* p) { delete static_cast<HNSWAdapter*>(p); }
int qvh_adapter_insert(void* p, const char* id, const float* v, uint32_t len) { return ret(static_cast<HNSWAdapter*>(p)->Insert(id, v, len)); }
int qvh_adapter_delete(void* p, const char* id) { return ret(static_cast<HNSWAdapter*>(p)->Delete(id)); }
int qvh_adapter_search(void* p, const float* q, uint32_t len, int k, void* res) { return ret(static_cast<HNSWAdapter*>(p)->Search(q, len, k, &static_cast<Results*>(res)->r)); }
int qvh_adapter_search_negative(void* p, const float* q, uint32_t len, const float* neg, uint32_t neg_len, float w, int k, void* res) {
    return ret(static_cast<HNSWAdapter*>(p)->SearchWithNegative(q, len, neg, neg_len, w, k, &static_cast<Results*>(res)->r));
}
int qvh_adapter_size(void* p) { return static_cast<HNSWAdapter*>(p)->Size(); }

// ---- HybridIndex
void* qvh_hybrid_new_placed(int metric, const int* devices, int n_devices, int peer_copy, int bf16_rows, int M, int maxM0, int efC, int efS,
                            int exact_threshold, double exploration, uint64_t seed) {
    HybridConfig c; c.metric = (qv_metric)metric; c.placement = Placement(std::vector<int>(devices, devices + n_devices), peer_copy != 0, bf16_rows != 0);
    c.hnsw.M = M; c.hnsw.MaxM0 = maxM0; c.hnsw.EfConstruction = efC; c.hnsw.EfSearch = efS;
    c.hnsw.seed = seed; c.exact_threshold = exact_threshold; c.exploration_factor = exploration; c.seed = seed ^ 0xA5A5A5A5ull;
    return new HybridIndex(c);
}
void* qvh_hybrid_new(int metric, int device, int M, int maxM0, int efC, int efS, int exact_threshold, double exploration, uint64_t seed) {
    HybridConfig c; c.metric = (qv_metric)metric; c.placement = Placement(device); c.hnsw.M = M; c.hnsw.MaxM0 = maxM0; c.hnsw.EfConstruction = efC; c.hnsw.EfSearch = efS;
    c.hnsw.seed = seed; c.exact_threshold = exact_threshold; c.exploration_factor = exploration; c.seed = seed ^ 0xA5A5A5A5ull;
    return new HybridIndex(c);
}
void qvh_hybrid_free(void* p) { delete static_cast<HybridIndex*>(p); }
int qvh_hybrid_insert(void* p, const char* id, const float* v, uint32_t len) { return ret(static_cast<HybridIndex*>(p)->Insert(id, v, len)); }
int qvh_hybrid_insert_batch(void* p, const char** ids, const float* packed, const uint32_t* lens, uint32_t n) {
    std::vector<std::string> i; std::vector<const float*> v; std::vector<uint32_t> l;
    size_t off = 0;
    for (uint32_t j = 0; j < n; j++) { i.push_back(ids[j]); v.push_back(packed + off); l.push_back(lens[j]); off += lens[j]; }
    return ret(static_cast<HybridIndex*>(p)->InsertBatch(i, v, l));
}
int qvh_hybrid_delete(void* p, const char* id) { return ret(static_cast<HybridIndex*>(p)->Delete(id)); }
int qvh_hybrid_delete_batch(void* p, const char** ids, uint32_t n) {
    std::vector<std::string> i; for (uint32_t j = 0; j < n; j++) i.push_back(ids[j]);
    return ret(static_cast<HybridIndex*>(p)->DeleteBatch(i));
}
int qvh_hybrid_search(void* p, const float* q, uint32_t len, int k, void* res) {
    Results* r = static_cast<Results*>(res);
    return ret(static_cast<HybridIndex*>(p)->searchWithStrategy(q, len, k, "", nullptr, 0, 0.5f, false, &r->r, &r->used1));
}
int qvh_hybrid_search_request(void* p, const float* q, uint32_t len, int k, const char* force, const float* neg, uint32_t neg_len, float w, void* res) {
    Results* r = static_cast<Results*>(res);
    return ret(static_cast<HybridIndex*>(p)->SearchWithRequest(q, len, k, force ? force : "", neg, neg_len, w, &r->r, &r->used1));
}
int qvh_hybrid_batch_search(void* p, const float* qs, uint32_t len, uint32_t nq, int k, const char* force, void* res) {
    Results* r = static_cast<Results*>(res);
    return ret(static_cast<HybridIndex*>(p)->BatchSearch(qs, len, nq, k, force ? force : "", &r->many, &r->used));
}
int qvh_hybrid_size(void* p) { return static_cast<HybridIndex*>(p)->Size(); }
const char* qvh_hybrid_select_strategy(void* p, int count, int dim, int k) {
    static thread_local std::string s;
    s = static_cast<HybridIndex*>(p)->SelectStrategy(count, dim, k);
    return s.c_str();
}

}  // extern "C"
