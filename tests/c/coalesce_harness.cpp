// coalesce_harness.cpp — CPU test harness for quiver_amd/csrc/qv_coalesce.h (the front that lets concurrent single-query callers
// share device passes).  The "device pass" here is a sleep; the "result" of a query is a function of its first element, so a
// caller that received somebody else's share, or a share cut at the wrong k, is caught.  No GPU, no libqv.
#include "../../quiver_amd/csrc/qv_coalesce.h"

#include <chrono>
#include <thread>

extern "C" int coalesce_harness(int lanes, unsigned max_group, unsigned n_threads, unsigned calls_per_thread, unsigned pass_us, unsigned think_us, int two_keys, unsigned slow_every,
                                unsigned long long* out /* solo, led, rode, groups, group_queries, lingers, wrong, max_concurrent_passes */) {
    qvco::Front front(lanes, max_group);
    std::atomic<unsigned long long> wrong{0};
    std::atomic<int> running{0}, max_running{0};
    const unsigned dim = 4;
    auto pass = [&](unsigned nq) {
        const int r = running.fetch_add(1) + 1;
        int m = max_running.load();
        while (r > m && !max_running.compare_exchange_weak(m, r)) {}
        std::this_thread::sleep_for(std::chrono::microseconds(pass_us + nq));
        running.fetch_sub(1);
    };
    auto answer = [](float q0, unsigned i) { return (uint32_t)(q0 * 1000.f) + i; };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < n_threads; t++)
        th.emplace_back([&, t] {
            for (unsigned c = 0; c < calls_per_thread; c++) {
                const unsigned nq = 1 + (t + c) % 3, k = 1 + (t * 7 + c * 3) % 9;
                std::vector<float> q((size_t)nq * dim);
                for (unsigned i = 0; i < nq; i++) q[(size_t)i * dim] = (float)(t * 100 + c * 3 + i);
                std::vector<uint32_t> rows((size_t)nq * k, 7u), count(nq, 99u); std::vector<float> dist((size_t)nq * k, -1.f);
                char err[256]; err[0] = 0;
                const int rc = front.submit(
                    (two_keys && t % 5 == 4) ? 1 : 0, q.data(), nq, dim, k, rows.data(), dist.data(), count.data(), nullptr,
                    [&] {                                                   // solo: straight into the caller's buffers
                        pass(nq);
                        for (unsigned i = 0; i < nq; i++) { count[i] = k; for (unsigned j = 0; j < k; j++) { rows[(size_t)i * k + j] = answer(q[(size_t)i * dim], j); dist[(size_t)i * k + j] = (float)j; } }
                        return 0;
                    },
                    [&](qvco::Group& g, auto& early) {
                        g.size_outputs(false);
                        pass(g.nq);
                        // a "slow" query now and then (qv_graph_search's exact-heap redo): everything else is final after the first
                        // pass and goes out early; the slow ones follow after a second pass
                        bool slow = false;
                        for (unsigned i = 0; i < g.nq; i++) {
                            const bool s = slow_every && ((unsigned)g.queries()[(size_t)i * dim]) % slow_every == 0;
                            slow |= s;
                            g.count[i] = s ? 0xFFFFFFFEu : g.kmax;
                            if (!s) for (unsigned j = 0; j < g.kmax; j++) { g.rows[(size_t)i * g.kmax + j] = answer(g.queries()[(size_t)i * dim], j); g.dist[(size_t)i * g.kmax + j] = (float)j; }
                        }
                        if (slow) {
                            early();
                            std::this_thread::sleep_for(std::chrono::microseconds(pass_us));   // (second passes run beside the next groups' first passes: not counted against the lanes)
                            for (unsigned i = 0; i < g.nq; i++)
                                if (g.count[i] == 0xFFFFFFFEu) { g.count[i] = g.kmax; for (unsigned j = 0; j < g.kmax; j++) { g.rows[(size_t)i * g.kmax + j] = answer(g.queries()[(size_t)i * dim], j); g.dist[(size_t)i * g.kmax + j] = (float)j; } }
                        }
                        return 0;
                    },
                    [] { return ""; }, err, sizeof(err));
                if (rc != 0) wrong.fetch_add(1000000);
                for (unsigned i = 0; i < nq; i++) {
                    if (count[i] != k) wrong.fetch_add(1);
                    for (unsigned j = 0; j < k; j++)
                        if (rows[(size_t)i * k + j] != answer(q[(size_t)i * dim], j) || dist[(size_t)i * k + j] != (float)j) wrong.fetch_add(1);
                }
                if (think_us) std::this_thread::sleep_for(std::chrono::microseconds(think_us));
            }
        });
    for (auto& x : th) x.join();
    out[0] = front.stats.solo.load(); out[1] = front.stats.led.load(); out[2] = front.stats.rode.load(); out[3] = front.stats.groups.load();
    out[4] = front.stats.group_queries.load(); out[5] = front.stats.lingers.load(); out[6] = wrong.load(); out[7] = (unsigned long long)max_running.load();
    return 0;
}
