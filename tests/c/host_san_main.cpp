// host_san_main.cpp — quiver_amd/csrc/host/qvhost.cpp (the C++ mirror of the reference's Go callers) under the sanitizers, against
// tests/c/qv_stub.cpp (a CPU stand-in for libqv answered by the oracle).  TEST INFRASTRUCTURE (tests/c/Makefile: host_tsan, host_asan).
// The reference's concurrency contract is what is exercised: searches under a read lock from many threads (collection.go:647,
// hnsw.go:602-606, hybrid_index.go:473) while mutations take the write lock (exact.go:38-70, hnsw.go:266-334, hybrid_index.go:86-372).
// Exit status 0: every result was well-formed (ascending, <= k, ids that exist or existed); the sanitizers report the rest.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <random>
#include <thread>

#include "../../quiver_amd/csrc/host/qvhost.h"

using namespace quiver;
static std::atomic<int> bad{0};
#define CHECK(c, ...) do { if (!(c)) { bad++; fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); } } while (0)

static std::vector<float> unit(std::mt19937& g, uint32_t d) {
    std::normal_distribution<float> n(0.f, 1.f);
    std::vector<float> v(d); double s = 0; for (auto& x : v) { x = n(g); s += (double)x * x; }
    for (auto& x : v) x = (float)(x / std::sqrt(s));
    return v;
}
static void ascending(const std::vector<BasicSearchResult>& r, size_t k) {
    CHECK(r.size() <= k, "%zu results for k = %zu", r.size(), k);
    for (size_t i = 1; i < r.size(); i++) CHECK(!(r[i].distance < r[i - 1].distance), "results not ascending at %zu", i);
}

static void exact_index(int n_search) {
    const uint32_t D = 24;
    ExactIndex ex(QV_COSINE, Placement(0));
    std::mt19937 g0(1);
    for (int i = 0; i < 200; i++) { auto v = unit(g0, D); CHECK(ex.Insert("base" + std::to_string(i), v.data(), D).empty(), "insert"); }
    std::atomic<bool> stop{false};
    std::thread mut([&] {
        std::mt19937 g(2);
        for (int r = 0; !stop.load(); r++) {
            auto v = unit(g, D);
            const std::string id = "churn" + std::to_string(r % 40);
            Error e = ex.Insert(id, v.data(), D);
            if (!e.empty()) { CHECK(e.find("already exists") != std::string::npos, "%s", e.c_str()); CHECK(ex.Delete(id).empty(), "delete"); }
            if (r % 7 == 0) { std::vector<std::string> ids; std::vector<float> pk; for (int j = 0; j < 5; j++) { ids.push_back("many" + std::to_string(r) + "_" + std::to_string(j)); auto w = unit(g, D); pk.insert(pk.end(), w.begin(), w.end()); }
                              CHECK(ex.InsertMany(ids, pk.data(), D).empty(), "insert many"); for (auto& s : ids) CHECK(ex.Delete(s).empty(), "delete many"); }
        }
    });
    std::vector<std::thread> th;
    for (int t = 0; t < n_search; t++)
        th.emplace_back([&, t] {
            std::mt19937 g(100 + t);
            for (int i = 0; i < 150; i++) {
                auto q = unit(g, D);
                std::vector<BasicSearchResult> out;
                const int k = 1 + (i % 12);
                CHECK(ex.Search(q.data(), D, k, &out).empty(), "search");
                ascending(out, (size_t)k);
                if (i % 10 == 0) { std::vector<std::vector<BasicSearchResult>> many; auto q2 = unit(g, D); q.insert(q.end(), q2.begin(), q2.end());
                                   CHECK(ex.SearchMany(q.data(), D, 2, k, &many).empty(), "search many"); for (auto& r : many) ascending(r, (size_t)k); }
                if (i % 15 == 0) { auto neg = unit(g, D); std::vector<float> nd; CHECK(ex.SearchWithNegativeDistances(q.data(), neg.data(), D, 30, &out, &nd).empty(), "negative"); CHECK(nd.size() == out.size(), "negative sizes"); }
                (void)ex.Size(); (void)ex.Has("base3");
            }
        });
    for (auto& x : th) x.join();
    stop = true; mut.join();
    CHECK(ex.Size() >= 200, "size %d", ex.Size());
}

static void hnsw_index(int n_search) {
    const uint32_t D = 16;
    HNSWConfig cfg; cfg.M = 8; cfg.MaxM0 = 16; cfg.EfConstruction = 40; cfg.EfSearch = 32; cfg.MaxLevel = 6; cfg.seed = 5;
    HNSW h(QV_COSINE, 0, cfg);
    std::mt19937 g0(7);
    {   // the first half through the device-side construction call, the second node by node (host-driven walk, one distance batch per hop)
        std::vector<std::string> ids; std::vector<float> pk;
        for (int i = 0; i < 120; i++) { ids.push_back("b" + std::to_string(i)); auto v = unit(g0, D); pk.insert(pk.end(), v.begin(), v.end()); }
        Error e = h.InsertBatch(ids, pk.data(), D, 64, 4);
        CHECK(e.empty(), "insert batch: %s", e.c_str());
    }
    for (int i = 0; i < 60; i++) { auto v = unit(g0, D); Error e = h.Insert("n" + std::to_string(i), v.data(), D); CHECK(e.empty(), "insert: %s", e.c_str()); }
    std::atomic<bool> stop{false};
    std::thread mut([&] {
        std::mt19937 g(8);
        for (int r = 0; !stop.load() && r < 60; r++) {
            auto v = unit(g, D);
            const std::string id = "c" + std::to_string(r % 10);
            Error e = h.Insert(id, v.data(), D);
            if (!e.empty()) (void)h.Delete(id);
        }
    });
    std::vector<std::thread> th;
    for (int t = 0; t < n_search; t++)
        th.emplace_back([&, t] {
            std::mt19937 g(200 + t);
            for (int i = 0; i < 60; i++) {
                auto q = unit(g, D);
                std::vector<HNSWResult> out;
                const int k = 1 + (i % 8);
                Error e = h.Search(q.data(), D, k, &out);
                CHECK(e.empty(), "hnsw search: %s", e.c_str());
                CHECK(out.size() <= (size_t)k, "hnsw k");
                for (size_t j = 1; j < out.size(); j++) CHECK(!(out[j].distance < out[j - 1].distance), "hnsw order");
                if (i % 6 == 0) {
                    std::vector<std::vector<HNSWResult>> many; auto q2 = unit(g, D); q.insert(q.end(), q2.begin(), q2.end());
                    e = h.SearchBatch(q.data(), D, 2, k, &many);
                    CHECK(e.empty(), "hnsw batch: %s", e.c_str());
                }
                (void)h.Size();
            }
        });
    for (auto& x : th) x.join();
    stop = true; mut.join();
}

static void hybrid_index(int n_search) {
    const uint32_t D = 16;
    HybridConfig cfg; cfg.metric = QV_COSINE; cfg.hnsw.M = 8; cfg.hnsw.MaxM0 = 16; cfg.hnsw.EfConstruction = 40; cfg.hnsw.EfSearch = 32; cfg.hnsw.MaxLevel = 6;
    cfg.exact_threshold = 50; cfg.seed = 3;
    HybridIndex hy(cfg);
    std::mt19937 g0(11);
    {
        std::vector<std::string> ids; std::vector<std::vector<float>> vs; std::vector<const float*> ps; std::vector<uint32_t> lens;
        for (int i = 0; i < 100; i++) { ids.push_back("h" + std::to_string(i)); vs.push_back(unit(g0, D)); }
        for (auto& v : vs) { ps.push_back(v.data()); lens.push_back(D); }
        Error e = hy.InsertBatch(ids, ps, lens);
        CHECK(e.empty(), "hybrid insert batch: %s", e.c_str());
    }
    std::atomic<bool> stop{false};
    std::thread mut([&] {
        std::mt19937 g(12);
        for (int r = 0; !stop.load() && r < 80; r++) {
            auto v = unit(g, D);
            const std::string id = "x" + std::to_string(r % 12);
            Error e = hy.Insert(id, v.data(), D);
            if (!e.empty()) (void)hy.Delete(id);
            if (r % 9 == 0) (void)hy.DeleteBatch({"x1", "x2"});
        }
    });
    std::vector<std::thread> th;
    for (int t = 0; t < n_search; t++)
        th.emplace_back([&, t] {
            std::mt19937 g(300 + t);
            const char* force[] = {"", "exact", "hnsw"};
            for (int i = 0; i < 60; i++) {
                auto q = unit(g, D);
                std::vector<BasicSearchResult> out; std::string used;
                const int k = 1 + (i % 9);
                Error e = hy.SearchWithRequest(q.data(), D, k, force[i % 3], nullptr, 0, 0.5f, &out, &used);
                CHECK(e.empty(), "hybrid search: %s", e.c_str());
                ascending(out, (size_t)k);
                if (i % 5 == 0) { auto neg = unit(g, D); e = hy.SearchWithRequest(q.data(), D, k, force[(i / 5) % 3], neg.data(), D, 0.3f, &out, &used); CHECK(e.empty(), "hybrid negative: %s", e.c_str()); CHECK(out.size() <= (size_t)k, "negative k"); }
                if (i % 8 == 0) { std::vector<std::vector<BasicSearchResult>> many; std::vector<std::string> um; auto q2 = unit(g, D); q.insert(q.end(), q2.begin(), q2.end());
                                  e = hy.BatchSearch(q.data(), D, 2, k, force[i % 3], &many, &um); CHECK(e.empty(), "hybrid batch: %s", e.c_str()); }
                (void)hy.Size();
            }
        });
    for (auto& x : th) x.join();
    stop = true; mut.join();
}

int main(int argc, char** argv) {
    const int n_search = argc > 1 ? atoi(argv[1]) : 6;
    exact_index(n_search); printf("exact index: %d searchers beside a mutator, failures so far %d\n", n_search, bad.load());
    hnsw_index(n_search);  printf("hnsw: %d searchers beside a mutator, failures so far %d\n", n_search, bad.load());
    hybrid_index(n_search); printf("hybrid index: %d searchers beside a mutator, failures so far %d\n", n_search, bad.load());
    return bad.load() ? 1 : 0;
}
