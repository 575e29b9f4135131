// qv_select.h — pieces of the radix selection (qv_select.hip) that the kernels producing the keys share: the per-query state,
// the histogram update of one window and the end-of-window step.  k_flat_keys (qv_rank.hip) counts the first window while it
// writes its keys, so the selection does not read them once more for it.
#pragma once
#include "qv_kernels.h"

namespace qv {

constexpr int kSelBlock = 256;
constexpr int kSelBins = 4096;
constexpr int kSelSortBlock = 1024;

struct SelState {
    unsigned long long prefix;   // leading `bits` bits of the k-th key (the rest zero)
    uint32_t bits;               // decided bits: 0, 12, 24, 32
    uint32_t k_rem;              // rank (1-based) of the k-th key inside the current bucket
    uint32_t below;              // keys strictly below the bucket: all of them are in the answer
    uint32_t bucket;             // keys in the bucket
    uint32_t done;               // below + bucket fit the sort: the remaining windows return at once
    uint32_t ticket;             // workgroups that have finished the current window
    uint32_t n_cand;             // keys the compaction kept
    uint32_t pad[7];
};
static_assert(sizeof(SelState) == 64, "SelState is one 64-byte record per query");

template <int W> struct SelWindow {
    static constexpr int shift = W == 0 ? 52 : (W == 1 ? 40 : 32);
    static constexpr int wbits = W == 2 ? 8 : 12;
    static constexpr uint32_t nb = 1u << wbits;
};

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x, uint32_t lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off); if ((int)lane >= off) x += y; }
    return x;
}

// A thread's keys into the workgroup's LDS histogram h, run by run.  The keys of a window crowd into a few bins — the distances of
// unit vectors share their exponent, and a later window only sees the keys of one bucket — so one LDS atomic per key queued 64
// lanes on two or three addresses (107 us for the first window over 10M keys), and peeling equal digits off with ballots cost
// ~25 instructions per key (150 us per window over 67M sample bounds).  A thread instead keeps (digit, count) of its current run
// of equal digits in registers and touches LDS when the digit changes: long runs where the bins are crowded, and where digits
// scatter there is no crowd to queue behind.  No cross-lane operation: callers need not be converged.
struct SelRun { uint32_t d, n; };
template <int W>
__device__ __forceinline__ void sel_count(uint32_t* h, SelRun& run, uint64_t key, bool match) {
    const uint32_t d = (uint32_t)(key >> SelWindow<W>::shift) & (SelWindow<W>::nb - 1);
    if (!match) return;
    if (d == run.d) { run.n++; return; }
    if (run.n) atomicAdd(&h[run.d], run.n);
    run.d = d; run.n = 1;
}
__device__ __forceinline__ void sel_flush(uint32_t* h, SelRun& run) {
    if (run.n) atomicAdd(&h[run.d], run.n);
    run.n = 0;
}

// End of a window for one workgroup: its bins go to the query's global histogram; the LAST workgroup of the query to arrive
// (a ticket) scans it, extends the k-th key's prefix in the query's state and leaves the global bins zero for the next window.
// n_groups = workgroups that call this for the query.  h: the workgroup's LDS histogram (kSelBins words; reused by the scan).
template <int W>
__device__ __forceinline__ void sel_finish_window(uint32_t* h, uint32_t* __restrict__ gh, SelState* __restrict__ s, uint32_t n_groups, uint32_t k, uint32_t cap) {
    constexpr int shift = SelWindow<W>::shift;
    constexpr uint32_t nb = SelWindow<W>::nb;
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t s_last;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6, nthreads = blockDim.x;
    __syncthreads();
    // No __threadfence() here.  On this multi-XCD part an agent-scope fence writes the XCD's L2 back: with 489 workgroups at the
    // end of a window over 8 MB of keys it was 27 of the kernel's 43 us (measured by compiling the pieces out).  Nothing but
    // device-scope atomics crosses workgroups in this step, and those execute at the memory side: the bins are added with
    // RETURNING atomics (a returned value means the add has been performed), the barrier collects the workgroup, the ticket follows;
    // the last workgroup takes the bins with atomic exchanges (which also leave them zero for the next window).
    uint32_t sink = 0;
    for (uint32_t b = threadIdx.x; b < nb; b += nthreads) if (h[b]) sink |= atomicAdd(&gh[b], h[b]);
    if (sink == 0xFFFFFFFFu) s->pad[0] = sink;                       // (keeps the returns — and the wait for them — alive; a bin never reaches 2^32 - 1)
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&s->ticket, 1u) == n_groups - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    {   // all of a thread's bins requested together
        uint32_t tmp[16];
#pragma unroll
        for (uint32_t u = 0; u < 16; u++) { const uint32_t b = threadIdx.x + u * nthreads; tmp[u] = b < nb ? atomicExch(&gh[b], 0u) : 0u; }
#pragma unroll
        for (uint32_t u = 0; u < 16; u++) { const uint32_t b = threadIdx.x + u * nthreads; if (b < nb) h[b] = tmp[u]; }
    }
    __syncthreads();
    const unsigned long long prefix = W > 0 ? s->prefix : 0ull;
    const uint32_t k_rem = W > 0 ? s->k_rem : k;
    const uint32_t per_t = (nb + nthreads - 1) / nthreads;            // bins per thread
    uint32_t mine = 0;
    for (uint32_t u = 0; u < per_t; u++) { const uint32_t b = threadIdx.x * per_t + u; mine += b < nb ? h[b] : 0u; }
    const uint32_t inc = wave_incl_scan(mine, lane);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; w++) before += wsum[w];
    const uint32_t excl = before + inc - mine;
    if (excl < k_rem && k_rem <= excl + mine) {                       // exactly one thread: the k-th key's digit is among its bins
        uint32_t cum = excl, d = threadIdx.x * per_t;
        for (uint32_t u = 0; u < per_t; u++, d++) { if (k_rem <= cum + h[d]) break; cum += h[d]; }
        const uint32_t below = (W > 0 ? s->below : 0u) + cum;
        s->prefix = prefix | ((unsigned long long)d << shift);
        s->bits = 64 - shift;
        s->k_rem = k_rem - cum;
        s->below = below;
        s->bucket = h[d];
        s->done = below + h[d] <= cap ? 1u : 0u;
        s->ticket = 0;
    }
}

}  // namespace qv
