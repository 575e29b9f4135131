// qv_coalesce.h — "ride the next pass": concurrent small host-pointer searches on one handle share device passes.
//
// Why it exists.  The reference searches under a READ lock, one query per call, from as many goroutines as there are requests
// (pkg/core/collection.go:647; pkg/hnsw/hnsw.go:602-606; the only batch entry, DB.BatchSearch, type-asserts the reference's own
// *HybridIndexWrapper (pkg/core/db.go:726-727), so behind any other core.Index every batch becomes "parallel individual
// searches", db.go:805-828).  A drop-in index therefore sees N concurrent Index.Search(q, k) calls and nothing else.  One
// query over a large corpus is a whole HBM pass (0.45 ms at 1M x 768) whether it carries 1 query or 8, and 256 queries cost
// 0.6 ms through the matrix-core filter: N callers that each stream the corpus get N passes' worth of time for N answers.
//
// What it does.  A handle has `lanes` passes in flight at most (1 for a bandwidth-bound flat scan: a second concurrent pass
// only halves the first one's rate; a few for graph traversals, where a pass is one wavefront per query).  A caller that finds
// a lane free and nobody waiting runs at once, in its own buffers, exactly as before (a lone caller pays two uncontended mutex
// operations and nothing else — there is NO timer and no waiting for company).  A caller that finds every lane busy joins the
// open GROUP of its key (or opens one); when a lane frees up, the thread that finished hands the lane to the oldest group's
// first member, which runs the whole group as ONE multi-query call and distributes the results.  Groups therefore hold exactly
// the queries that arrived while the previous pass was running.
//
// Waiting is on two futex words per group (`go` for its leader, `done` for everybody else: one system call wakes the lot).
#pragma once
#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <vector>

#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace qvco {

inline void futex_wait(std::atomic<uint32_t>* w, uint32_t expect) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAIT_PRIVATE, expect, nullptr, nullptr, 0);
}
inline void futex_wake(std::atomic<uint32_t>* w, int n) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAKE_PRIVATE, n, nullptr, nullptr, 0);
}
inline void wait_set(std::atomic<uint32_t>* w) {
    for (int spin = 0; spin < 64; spin++) { if (w->load(std::memory_order_acquire)) return; __builtin_ia32_pause(); }
    while (!w->load(std::memory_order_acquire)) futex_wait(w, 0);
}

struct Member {                       // one caller's request inside a group
    uint32_t q0, nq, k;               // its queries are rows q0 .. q0+nq-1 of the group's block
    uint32_t* rows_out; float* dist_out; uint32_t* count_out; uint32_t* evals_out;
};

struct Group {
    uint64_t key = 0;                 // callers whose requests may share a call have equal keys
    uint32_t dim = 0, nq = 0, kmax = 0;
    std::vector<float> queries;       // [nq][dim]
    std::vector<Member> members;      // members[0] leads
    // results of the group's call: lists of length kmax, padded past count like every host-pointer search
    std::vector<uint32_t> rows, count, evals;
    std::vector<float> dist;
    std::atomic<uint32_t> go{0}, done{0};
    int rc = 0;
    char err[256] = "";

    void size_outputs(bool with_evals) {
        rows.assign((size_t)nq * kmax, 0xFFFFFFFFu); dist.assign((size_t)nq * kmax, __builtin_inff()); count.assign(nq, 0);
        if (with_evals) evals.assign(nq, 0);
    }
    // every member's share: the first k of each of its queries' kmax results (top-k is a prefix of top-kmax under one total
    // order); a count of 0xFFFFFFFF ("redo on the host", qv_graph_search) passes through
    void scatter() const {
        for (const Member& m : members)
            for (uint32_t i = 0; i < m.nq; i++) {
                const size_t src = (size_t)(m.q0 + i) * kmax, dst = (size_t)i * m.k;
                memcpy(m.rows_out + dst, rows.data() + src, (size_t)m.k * 4);
                memcpy(m.dist_out + dst, dist.data() + src, (size_t)m.k * 4);
                const uint32_t c = count[m.q0 + i];
                m.count_out[i] = c == 0xFFFFFFFFu ? c : (c < m.k ? c : m.k);
                if (m.evals_out && !evals.empty()) m.evals_out[i] = evals[m.q0 + i];
            }
    }
};

class Front {
  public:
    Front(int lanes, uint32_t max_group_queries) : lanes_(lanes), max_q_(max_group_queries) {}

    // What happened to a request (for tests and reports).
    struct Stats { std::atomic<uint64_t> solo{0}, led{0}, rode{0}, groups{0}, group_queries{0}; };
    Stats stats;

    // solo():       run the caller's own request in its own buffers (what the entry point did before there was a front)
    // run(Group&):  run g.queries (g.nq of them, lists of g.kmax) into g.rows / g.dist / g.count (/ g.evals); returns a status
    // last_error(): the thread-local message of a failed run, copied for the members
    template <class Solo, class Run, class LastErr>
    int submit(uint64_t key, const float* queries, uint32_t nq, uint32_t dim, uint32_t k,
               uint32_t* rows_out, float* dist_out, uint32_t* count_out, uint32_t* evals_out,
               Solo&& solo, Run&& run, LastErr&& last_error, char* err_out, size_t err_cap) {
        std::shared_ptr<Group> grp;
        bool leader = false;
        {
            std::lock_guard<std::mutex> l(mu_);
            if (inflight_ < lanes_ && pending_.empty()) inflight_++;          // a free lane and nobody waiting: go now
            else {
                for (auto it = pending_.rbegin(); it != pending_.rend(); ++it)
                    if ((*it)->key == key && (*it)->dim == dim && (*it)->nq + nq <= max_q_) { grp = *it; break; }
                size_t members_before = grp ? grp->members.size() : 0, floats_before = grp ? grp->queries.size() : 0;
                try {
                    if (!grp) {
                        grp = std::make_shared<Group>();
                        grp->key = key; grp->dim = dim;
                        grp->queries.reserve((size_t)std::min<uint32_t>(max_q_, 64) * dim);
                        leader = true;
                    }
                    grp->members.push_back(Member{grp->nq, nq, k, rows_out, dist_out, count_out, evals_out});
                    grp->queries.insert(grp->queries.end(), queries, queries + (size_t)nq * dim);
                    if (leader) pending_.push_back(grp);                       // (last: a group is visible only once it is whole)
                } catch (...) {                                                // out of host memory while queueing: leave the group as it was
                    if (!leader && grp) { grp->members.resize(members_before); grp->queries.resize(floats_before); }
                    snprintf(err_out, err_cap, "out of host memory");
                    return -7;
                }
                grp->nq += nq;
                if (k > grp->kmax) grp->kmax = k;
            }
        }
        if (!grp) {                                                            // solo
            const int rc = solo();
            stats.solo.fetch_add(1, std::memory_order_relaxed);
            finish_lane();
            return rc;
        }
        if (!leader) {                                                         // ride: the leader writes this caller's outputs
            wait_set(&grp->done);
            stats.rode.fetch_add(1, std::memory_order_relaxed);
            if (grp->rc != 0) snprintf(err_out, err_cap, "%s", grp->err);
            return grp->rc;
        }
        wait_set(&grp->go);                                                    // a lane was handed to this group: nobody can join any more
        int rc;
        const bool alone = grp->members.size() == 1;
        try {
            rc = alone ? solo() : run(*grp);
        } catch (...) { rc = -7; }
        if (rc != 0) snprintf(grp->err, sizeof(grp->err), "%s", rc == -7 && !*last_error() ? "out of host memory" : last_error());
        finish_lane();                                                         // the next group starts before this one's results are handed out
        if (!alone && rc == 0) grp->scatter();
        grp->rc = rc;
        stats.led.fetch_add(1, std::memory_order_relaxed);
        stats.groups.fetch_add(1, std::memory_order_relaxed);
        stats.group_queries.fetch_add(grp->nq, std::memory_order_relaxed);
        grp->done.store(1, std::memory_order_release);
        if (!alone) futex_wake(&grp->done, INT_MAX);
        if (rc != 0) snprintf(err_out, err_cap, "%s", grp->err);
        return rc;
    }

  private:
    void finish_lane() {
        std::shared_ptr<Group> next;
        {
            std::lock_guard<std::mutex> l(mu_);
            if (!pending_.empty()) { next = pending_.front(); pending_.pop_front(); }
            else inflight_--;
        }
        if (next) { next->go.store(1, std::memory_order_release); futex_wake(&next->go, 1); }
    }

    std::mutex mu_;
    std::deque<std::shared_ptr<Group>> pending_;
    int inflight_ = 0;
    const int lanes_;
    const uint32_t max_q_;
};

}  // namespace qvco
