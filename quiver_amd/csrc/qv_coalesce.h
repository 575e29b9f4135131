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
// a lane free and nobody waiting runs at once, in its own buffers, exactly as before (a lone caller pays two uncontended lock
// operations and nothing else — no waiting for company that is not known to be on its way).  A caller that finds every lane busy
// joins the open GROUP of its key (or opens one); when a lane frees up, the thread that finished hands the lane to the oldest
// group's first member, which runs the whole group as ONE multi-query call and distributes the results.  Groups therefore hold
// the queries that arrived while the previous pass was running — plus, see Front::linger_then_close, the callers that pass
// itself released, for whom the leader waits a bounded moment (a quarter of a pass at most) because it knows they are coming back.
//
// Waiting is on futex words of the group (`go` for its leader, `phase` for everybody else: one system call wakes the lot).
// What a caller does under the front's lock is a slot claim (a few integer operations); its query is copied outside it.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include <linux/futex.h>
#include <sched.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

namespace qvco {

inline void futex_wait(std::atomic<uint32_t>* w, uint32_t expect) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAIT_PRIVATE, expect, nullptr, nullptr, 0);
}
inline void futex_wait_for(std::atomic<uint32_t>* w, uint32_t expect, int64_t ns) {
    struct timespec ts; ts.tv_sec = ns / 1000000000; ts.tv_nsec = ns % 1000000000;
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAIT_PRIVATE, expect, &ts, nullptr, 0);
}
inline void futex_wake(std::atomic<uint32_t>* w, int n) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAKE_PRIVATE, n, nullptr, nullptr, 0);
}
inline void wait_set(std::atomic<uint32_t>* w) {
    for (int spin = 0; spin < 64; spin++) { if (w->load(std::memory_order_acquire)) return; __builtin_ia32_pause(); }
    while (!w->load(std::memory_order_acquire)) futex_wait(w, 0);
}

// The front's own lock.  Its critical sections are tens of nanoseconds (a slot claim) and a thousand callers come through it
// within a millisecond of every pass's end: a sleeping mutex turns that burst into a convoy (every hand-over a futex wake and a
// context switch, ~7 us each: measured, 1024 callers needed longer to queue up again than the pass they were queueing for
// took).  Test-and-test-and-set; a waiter backs off longer each time it finds the lock taken (hundreds of hardware threads
// hammering one cache line slow the holder down) and gives the core away after a while.
class SpinLock {
  public:
    void lock() {
        int backoff = 4;
        for (int tries = 0;; tries++) {
            if (!f_.load(std::memory_order_relaxed) && !f_.exchange(1, std::memory_order_acquire)) return;
            for (int i = 0; i < backoff; i++) __builtin_ia32_pause();
            if (backoff < 256) backoff *= 2;
            else if ((tries & 7) == 7) sched_yield();
        }
    }
    void unlock() { f_.store(0, std::memory_order_release); }
  private:
    std::atomic<uint32_t> f_{0};
};

struct Member {                       // one caller's request inside a group
    uint32_t q0, nq, k;               // its queries are rows q0 .. q0+nq-1 of the group's block
    uint32_t* rows_out; float* dist_out; uint32_t* count_out; uint32_t* evals_out;
};

struct Group {
    uint64_t key = 0;                 // callers whose requests may share a call have equal keys
    uint32_t dim = 0, cap_q = 0;      // room for cap_q queries / members
    // claimed under the front's lock; final once the leader has closed the group
    uint32_t nq = 0, n_mem = 0, kmax = 0;
    bool has_lane = false;
    uint32_t want = 0; int64_t linger_ns = 0;   // hold the group open until it has `want` members, linger_ns at most
    std::unique_ptr<float[]> qbuf;    // [cap_q][dim]: every member writes its own rows, outside the lock
    std::unique_ptr<Member[]> mbuf;   // [cap_q]
    std::atomic<uint32_t> copied{0};  // members whose record and queries are in place
    std::atomic<uint32_t> go{0}, phase{0}, n_members{0}, full{0};   // phase: 0 running, 1 the members marked in delivered[] have their results, 2 all have
    // results of the group's call: lists of length kmax, padded past count like every host-pointer search
    std::vector<uint32_t> rows, count, evals;
    std::vector<float> dist;
    std::vector<uint8_t> delivered;   // per member (written before phase becomes 1)
    int rc = 0;
    char err[256] = "";

    const float* queries() const { return qbuf.get(); }
    void size_outputs(bool with_evals) {
        rows.assign((size_t)nq * kmax, 0xFFFFFFFFu); dist.assign((size_t)nq * kmax, __builtin_inff()); count.assign(nq, 0);
        if (with_evals) evals.assign(nq, 0);
    }
    // Every member's share: the first k of each of its queries' kmax results (top-k is a prefix of top-kmax under one total
    // order); a count of 0xFFFFFFFF ("redo on the host", qv_graph_search) passes through.
    void give(const Member& m) {
        for (uint32_t i = 0; i < m.nq; i++) {
            const size_t src = (size_t)(m.q0 + i) * kmax, dst = (size_t)i * m.k;
            memcpy(m.rows_out + dst, rows.data() + src, (size_t)m.k * 4);
            memcpy(m.dist_out + dst, dist.data() + src, (size_t)m.k * 4);
            const uint32_t c = count[m.q0 + i];
            m.count_out[i] = c == 0xFFFFFFFFu ? c : (c < m.k ? c : m.k);
            if (m.evals_out && !evals.empty()) m.evals_out[i] = evals[m.q0 + i];
        }
    }
    // The early round: the members whose queries are all final (count != not_final) are marked in delivered[]; returns how many.
    // delivered[] is written HERE ONLY, before the release store of phase = 1 — a rider reads its flag after loading phase == 1,
    // possibly much later (descheduled between the two loads), so the final round must never store to it: a rider that saw the
    // final round's mark would return before its results were written and the leader would then write into buffers the caller
    // owns again (the thread sanitizer reports exactly that pair on tests/c/coalesce_harness.cpp).
    uint32_t mark_early(uint32_t not_final = 0xFFFFFFFEu) {
        delivered.assign(n_mem, 0);
        uint32_t n = 0;
        for (uint32_t mi = 0; mi < n_mem; mi++) {
            const Member& m = mbuf[mi];
            bool fin = true;
            for (uint32_t i = 0; i < m.nq; i++) if (count[m.q0 + i] == not_final) { fin = false; break; }
            if (fin) { delivered[mi] = 1; n++; }
        }
        return n;
    }
    void scatter_early() { for (uint32_t mi = 0; mi < n_mem; mi++) if (delivered[mi]) give(mbuf[mi]); }
    // The final round: everybody the early round (if there was one) did not serve.  Reads delivered[], never writes it.
    void scatter_rest() { for (uint32_t mi = 0; mi < n_mem; mi++) if (delivered.empty() || !delivered[mi]) give(mbuf[mi]); }
};

class Front {
  public:
    // linger_div: a leader holds its group open for returning callers for 1 / linger_div of a pass at most
    Front(int lanes, uint32_t max_group_queries, int linger_div = 8) : lanes_(lanes), max_q_(max_group_queries), linger_div_(env_linger_div() ? env_linger_div() : linger_div) {}

    // What happened to a request (for tests and reports).
    struct Stats {
        std::atomic<uint64_t> solo{0}, led{0}, rode{0}, groups{0}, group_queries{0}, lingers{0}, linger_ns{0}, group_pass_ns{0}, early_rounds{0};
        void read(uint64_t out[8]) const {
            out[0] = solo.load(); out[1] = led.load(); out[2] = rode.load(); out[3] = groups.load(); out[4] = group_queries.load();
            out[5] = lingers.load(); out[6] = linger_ns.load(); out[7] = group_pass_ns.load();
        }
        uint64_t early() const { return early_rounds.load(); }        // groups whose first pass released part of their members early
    };
    Stats stats;

    // passes in flight at most, from now on (a handle whose pass cost changes with its size adjusts it; passes already running finish)
    // hold_open: whether a leader holds its group open for returning callers at all — worth it where a multi-query pass costs what a
    // single-query pass costs (a large scan, a traversal batch), not where grouping has a price and free lanes are the better answer
    void set_max_group(uint32_t q) { max_q_.store(q < 1 ? 1 : q, std::memory_order_relaxed); }
    void set_lanes(int n, bool hold_open = true) { lanes_.store(n < 1 ? 1 : n, std::memory_order_relaxed); hold_open_.store(hold_open, std::memory_order_relaxed); }

    // solo():       run the caller's own request in its own buffers (what the entry point did before there was a front)
    // run(Group&, early): run g.queries() (g.nq of them, lists of g.kmax) into g.rows / g.dist / g.count (/ g.evals); returns a
    //               status; may call early() once (see there)
    // last_error(): the thread-local message of a failed run, copied for the members
    template <class Solo, class Run, class LastErr>
    int submit(uint64_t key, const float* queries, uint32_t nq, uint32_t dim, uint32_t k,
               uint32_t* rows_out, float* dist_out, uint32_t* count_out, uint32_t* evals_out,
               Solo&& solo, Run&& run, LastErr&& last_error, char* err_out, size_t err_cap) {
        std::shared_ptr<Group> grp, fresh;
        bool leader = false, has_lane = false, wake_leader = false, counted = false;
        uint32_t my = 0, q0 = 0;                                               // this caller's place in its group
        for (;;) {
            std::shared_ptr<Group> hand_lane_to;                               // a group to start (outside the lock)
            bool done;
            uint32_t cap = 64;
            {
                std::lock_guard<SpinLock> l(mu_);
                if (!counted) {                                                // this caller is (as good as) one of those the last pass released
                    counted = true;
                    if (released_ > 0) { if (now_ns() - released_at_ < kReturnWindowNs) released_--; else released_ = 0; }
                }
                // an open group of this key — one that waits for a lane, or one its leader holds open for returning callers: join it
                for (auto it = pending_.rbegin(); it != pending_.rend(); ++it)
                    if ((*it)->key == key && (*it)->dim == dim && (*it)->nq + nq <= (*it)->cap_q) { grp = *it; break; }
                if (!grp && !has_lane && inflight_ < lanes_.load(std::memory_order_relaxed) && !waiting_for_lane()) {   // a free lane and nobody queueing for one
                    inflight_++; has_lane = true;
                }
                uint32_t expect = 0; int64_t linger = 0;
                if (has_lane && !grp) { expect = released_; linger = linger_ns(); }
                // No lane: open a group that waits for one.  A caller that HAS a lane but knows that the pass which just ended released
                // other callers opens a group too and holds it open for them (see linger_then_close).
                const bool open_group = !grp && (!has_lane || (expect > 0 && linger > 0));
                if (open_group && fresh) {
                    grp = std::move(fresh);
                    pending_.push_back(grp);
                    leader = true;
                    if (has_lane) { grp->has_lane = true; grp->want = 1 + expect; grp->linger_ns = linger; grp->go.store(1, std::memory_order_relaxed); }
                }
                if (grp) {                                                     // claim a place
                    my = grp->n_mem++; q0 = grp->nq;
                    grp->nq += nq;
                    if (k > grp->kmax) grp->kmax = k;
                    grp->n_members.store(grp->n_mem, std::memory_order_release);
                    if (grp->nq + 1 > grp->cap_q) grp->full.store(1, std::memory_order_relaxed);
                    if (has_lane && !leader) {
                        // (somebody opened this group while this caller, lane in hand, was allocating its own: the lane goes to that
                        // group if it has none, else to the oldest group that waits for one)
                        has_lane = false;
                        if (!grp->has_lane) { grp->has_lane = true; grp->want = grp->n_mem + released_; grp->linger_ns = linger_ns(); hand_lane_to = grp; }
                        else hand_lane_to = give_lane_locked();
                    }
                    wake_leader = !leader && grp->has_lane && (grp->n_mem >= grp->want || grp->full.load(std::memory_order_relaxed));
                }
                done = grp || (has_lane && !open_group);                       // a place in a group, or a lane and nobody to wait for
                // capacity of the group to be: twice what the last one held (the block is written once, never zero-filled)
                if (!done) cap = std::min<uint32_t>(max_q_.load(std::memory_order_relaxed), std::max<uint32_t>(std::max<uint32_t>(64, nq), 2 * last_group_q_));
            }
            if (hand_lane_to) { hand_lane_to->go.store(1, std::memory_order_release); futex_wake(&hand_lane_to->go, 1); }
            if (done) break;
            // out of the lock: the buffers of a group this caller is about to open
            fresh = std::shared_ptr<Group>(new (std::nothrow) Group());
            if (fresh) {
                fresh->key = key; fresh->dim = dim; fresh->cap_q = std::max(cap, nq);
                fresh->qbuf.reset(new (std::nothrow) float[(size_t)fresh->cap_q * dim]);
                fresh->mbuf.reset(new (std::nothrow) Member[fresh->cap_q]);
            }
            if (!fresh || !fresh->qbuf || !fresh->mbuf) {
                if (has_lane) finish_lane(0, 0);
                snprintf(err_out, err_cap, "out of host memory");
                return -7;
            }
        }
        if (grp) {                                                             // this caller's record and queries, in its own place
            grp->mbuf[my] = Member{q0, nq, k, rows_out, dist_out, count_out, evals_out};
            memcpy(grp->qbuf.get() + (size_t)q0 * dim, queries, (size_t)nq * dim * sizeof(float));
            grp->copied.fetch_add(1, std::memory_order_release);
        }
        if (wake_leader) futex_wake(&grp->n_members, 1);                       // the leader holds the group open for exactly this
        if (!grp) {                                                            // solo: nobody to wait for
            const int64_t t0 = now_ns();
            const int rc = solo();
            stats.solo.fetch_add(1, std::memory_order_relaxed);
            finish_lane(1, now_ns() - t0);
            return rc;
        }
        if (!leader) {                                                         // ride: the leader writes this caller's outputs
            uint32_t ph;
            for (int spin = 0; (ph = grp->phase.load(std::memory_order_acquire)) == 0; spin++) {
                if (spin < 64) __builtin_ia32_pause(); else futex_wait(&grp->phase, 0);
            }
            stats.rode.fetch_add(1, std::memory_order_relaxed);
            if (ph == 1) {                                                     // an early round: this caller's share may be in it
                if (grp->delivered[my]) return 0;
                while (grp->phase.load(std::memory_order_acquire) != 2) futex_wait(&grp->phase, 1);
            }
            if (grp->rc != 0) snprintf(err_out, err_cap, "%s", grp->err);
            return grp->rc;
        }
        wait_set(&grp->go);                                                    // a lane was handed to this group
        linger_then_close(grp);                                                // from here on nobody can join
        int rc;
        const bool alone = grp->n_mem == 1;
        const int64_t t0 = now_ns();
        bool lane_released = false;
        // run() may call this once, when g.count says which queries are final (everything but the entries equal to 0xFFFFFFFE) and
        // the rest needs a second, slow pass (qv_graph_search's exact-heap redo): the lane goes to the next group and the members
        // whose queries are all final get their results now instead of waiting for the slowest query of the group.
        uint32_t early_served = 0;
        auto early = [&] {
            if (alone || lane_released) return;
            early_served = grp->mark_early();                                  // only they return now: the rest stays blocked in the second pass
            lane_released = true;                                              // (mark_early may throw: nothing has changed before this line)
            finish_lane(early_served, now_ns() - t0);
            grp->scatter_early();
            grp->phase.store(1, std::memory_order_release);
            futex_wake(&grp->phase, INT_MAX);
            stats.early_rounds.fetch_add(1, std::memory_order_relaxed);
        };
        int64_t pass_ns = 0;
        try {
            rc = alone ? solo() : run(*grp, early);
            pass_ns = now_ns() - t0;
            if (!lane_released) finish_lane(grp->n_mem, pass_ns);              // the next group starts before this one's results are handed out
            else credit_released(grp->n_mem - early_served);                   // the members the second pass kept are released now
            if (!alone && rc == 0) grp->scatter_rest();
        } catch (...) {
            rc = -7;
            if (!pass_ns) { pass_ns = now_ns() - t0; if (!lane_released) finish_lane(grp->n_mem, pass_ns); else credit_released(grp->n_mem - early_served); }
        }
        if (rc != 0) snprintf(grp->err, sizeof(grp->err), "%s", rc == -7 && !*last_error() ? "out of host memory" : last_error());
        grp->rc = rc;
        if (alone) stats.solo.fetch_add(1, std::memory_order_relaxed);
        else {
            stats.led.fetch_add(1, std::memory_order_relaxed);
            stats.groups.fetch_add(1, std::memory_order_relaxed);
            stats.group_queries.fetch_add(grp->nq, std::memory_order_relaxed);
            stats.group_pass_ns.fetch_add((uint64_t)pass_ns, std::memory_order_relaxed);
        }
        grp->phase.store(2, std::memory_order_release);
        if (!alone) futex_wake(&grp->phase, INT_MAX);
        if (rc != 0) snprintf(err_out, err_cap, "%s", grp->err);
        return rc;
    }

  private:
    static int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    static constexpr int64_t kReturnWindowNs = 2000000;    // a released caller that has not come back after this long is not coming
    static constexpr int64_t kLingerMaxNs = 1000000;
    bool waiting_for_lane() const { for (auto& g : pending_) if (!g->has_lane) return true; return false; }   // (under mu_)
    // how long a group may be held open for the callers the last pass released: 1 / linger_div of a pass, 1 ms at most (under mu_)
#ifdef QV_VARIANTS
    static int env_linger_div() { static const int d = getenv("QV_COALESCE_LINGER_DIV") && atoi(getenv("QV_COALESCE_LINGER_DIV")) > 0 ? atoi(getenv("QV_COALESCE_LINGER_DIV")) : 0; return d; }   // (measurement switch, read once)
#else
    static int env_linger_div() { return 0; }
#endif
    int64_t linger_ns() const { return hold_open_.load(std::memory_order_relaxed) ? std::min<int64_t>(pass_ns_ / linger_div_, kLingerMaxNs) : 0; }

    // Closed-loop callers come back TOGETHER, a few microseconds after the pass that served them ends — just after the next group
    // has started without them, so that N callers alternate in two groups of N/2 and each waits two passes per answer (measured:
    // 8 callers on 1M x 768, 7.9 k QPS in groups of 4; 13.9 k in groups of 8 with this).  The leader therefore holds its group open
    // until the callers which that last pass released are back — it knows how many — or for a fraction of a pass (a quarter on the flat
    // indexes, an eighth on a graph — measured, profiles/r05_notes.md; 1 ms at most), whichever comes first.  No caller waits for company that is not known to be on its way: a lone caller never does.
    void linger_then_close(const std::shared_ptr<Group>& grp) {
        if (grp->linger_ns > 0 && grp->n_members.load(std::memory_order_acquire) < grp->want) {
            const int64_t t0 = now_ns(), deadline = t0 + grp->linger_ns;
            for (;;) {
                const uint32_t have = grp->n_members.load(std::memory_order_acquire);
                if (have >= grp->want || grp->full.load(std::memory_order_relaxed)) break;
                const int64_t now = now_ns();
                if (now >= deadline) break;
                if (now - t0 < 20000) { for (int i = 0; i < 32; i++) __builtin_ia32_pause(); }   // the first ones are back within microseconds
                else futex_wait_for(&grp->n_members, have, deadline - now);                        // (the join that completes the group wakes)
            }
            stats.lingers.fetch_add(1, std::memory_order_relaxed);
            stats.linger_ns.fetch_add((uint64_t)(now_ns() - t0), std::memory_order_relaxed);
        }
        uint32_t n;
        {
            std::lock_guard<SpinLock> l(mu_);
            for (auto it = pending_.begin(); it != pending_.end(); ++it)
                if (*it == grp) { pending_.erase(it); break; }
            n = grp->n_mem;
            last_group_q_ = grp->nq;
        }
        // every member that claimed a place finishes writing its record and its queries (a 3 KB copy: it is on its way)
        for (int spin = 0; grp->copied.load(std::memory_order_acquire) < n; spin++) { if (spin < 256) __builtin_ia32_pause(); else sched_yield(); }
    }

    // the lane of the caller goes to the oldest group that waits for one (returned: start it outside the lock), or becomes free (under mu_)
    std::shared_ptr<Group> give_lane_locked() {
        for (auto& g : pending_)
            if (!g->has_lane) { g->has_lane = true; g->want = g->n_mem + released_; g->linger_ns = linger_ns(); return g; }
        inflight_--;
        return nullptr;
    }

    // a pass with `callers` members has ended after pass_ns: hand the lane on
    void finish_lane(uint32_t callers, int64_t pass_ns) {
        std::shared_ptr<Group> next;
        {
            std::lock_guard<SpinLock> l(mu_);
            const int64_t now = now_ns();
            released_ = (now - released_at_ < kReturnWindowNs ? released_ : 0) + callers;
            released_at_ = now;
            if (pass_ns > 0) pass_ns_ = pass_ns_ ? (3 * pass_ns_ + pass_ns) / 4 : pass_ns;
            next = give_lane_locked();
        }
        if (next) { next->go.store(1, std::memory_order_release); futex_wake(&next->go, 1); }
    }

    // callers that a pass released without a lane changing hands (the members an early round left behind, when their group completes)
    void credit_released(uint32_t callers) {
        if (!callers) return;
        std::lock_guard<SpinLock> l(mu_);
        const int64_t now = now_ns();
        released_ = (now - released_at_ < kReturnWindowNs ? released_ : 0) + callers;
        released_at_ = now;
    }

    SpinLock mu_;
    std::deque<std::shared_ptr<Group>> pending_;   // groups that can still be joined: waiting for a lane, or held open by their leader
    int inflight_ = 0;
    uint32_t released_ = 0;                        // callers that recent passes released and that have not called again yet
    int64_t released_at_ = 0, pass_ns_ = 0;
    uint32_t last_group_q_ = 0;
    std::atomic<int> lanes_;
    std::atomic<bool> hold_open_{true};
    std::atomic<uint32_t> max_q_;
    const int linger_div_;
};

}  // namespace qvco
