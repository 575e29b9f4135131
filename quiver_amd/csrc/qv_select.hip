// qv_select.hip — the k smallest of n 64-bit (distance, row) keys for k above the 64-key wave list, by radix SELECT:
// histogram passes find the k-th key's leading bits, one compaction keeps the keys up to it, one workgroup sorts the
// few that are left.  (shared helpers and the key format: qv_kernels.h)
//
// Who asks: the reference's negative-example branches fetch max(2k, 30) results (pkg/hybrid/hybrid_index.go:516-522,
// pkg/hnsw/adapter.go:353-359 — k = 50 is a 100-key search), HybridIndex.BatchSearch takes any k (hybrid_index.go:677-811),
// and every k above 64 used to be a full stable radix SORT of all n keys per query (qv_rank.hip): 4 scatter passes moving
// 16 n bytes each, launched once per query.  Selection reads the keys (8 n bytes) once per decided window and writes ~k.
//
// The keys are (ord(distance) << 32) | row, distinct, so "the k smallest by (distance, row)" is exactly the order every other
// path of the library produces (qv.h: distance ascending, ties by row ascending), and the result is the first k of the full
// ranking, bit for bit (tests/test_gpu_select.py compares both and the oracle).
//
//   window 0   bits 63..52  sign + exponent + 3 mantissa bits of the distance      4096 bins
//   window 1   bits 51..40  the next 12 mantissa bits                              4096 bins
//   window 2   bits 39..32  the last 8 bits of the distance                         256 bins
// After a window the LAST workgroup to finish (a ticket) scans the histogram and extends the k-th key's prefix; as soon as
// (keys below the prefix's bucket) + (keys in it) fit the sort's capacity, the later windows return at once.  Unstructured
// 768-d data is done after two windows (the bucket of the 100th of a million distances holds about one key then).  Distances
// that tie beyond the capacity after all 32 bits (thousands of identical rows, zero vectors under cosine) are settled in row
// order by the sort kernel itself: a key's place in the keys array IS its row, so the first k_rem ties in array order are the ones.
#include "qv_select.h"

namespace qv {

// One window of the selection.  grid = (workgroups, nq); keys of query q at keys + q * stride, n of them (n even).
template <int W>
__global__ void __launch_bounds__(kSelBlock)
k_select_hist(const uint64_t* __restrict__ keys, size_t stride, uint32_t n, uint32_t k, uint32_t cap, SelState* __restrict__ st, uint32_t* __restrict__ hist,
              const uint32_t* __restrict__ active = nullptr /* a device word: queries from *active on are left alone */) {
    constexpr int shift = SelWindow<W>::shift;
    constexpr int wbits = SelWindow<W>::wbits;
    constexpr uint32_t nb = SelWindow<W>::nb;
    __shared__ uint32_t h[kSelBins];
    const uint32_t q = blockIdx.y;
    if (active != nullptr && q >= *active) return;
    SelState* s = st + q;
    if (s->done || s->bits != 64 - shift - wbits) return;              // decided already (uniform over the query's workgroups), or this window was
                                                                      // taken by the kernel that made the keys (k_flat_keys counts window 0 itself)
    const unsigned long long prefix = W > 0 ? s->prefix : 0ull;
    for (uint32_t b = threadIdx.x; b < nb; b += kSelBlock) h[b] = 0;
    __syncthreads();
    const uint64_t* src = keys + (size_t)q * stride;
    // a contiguous slice per workgroup, two keys (16 bytes) per lane and request
    const uint32_t per = ((n + gridDim.x - 1) / gridDim.x + 2 * kSelBlock - 1) / (2 * kSelBlock) * (2 * kSelBlock);
    const uint32_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    SelRun run{0u, 0u};
    auto count = [&](uint64_t key, bool valid) {
        sel_count<W>(h, run, key, valid && (W == 0 || (key >> (shift + wbits)) == (prefix >> (shift + wbits))));
    };
    for (uint32_t base = lo; base < hi; base += 2 * 4 * kSelBlock) {
        ulonglong2 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t j = base + (uint32_t)u * 2 * kSelBlock + 2 * threadIdx.x;
            v[u] = j < hi ? *reinterpret_cast<const ulonglong2*>(src + j) : make_ulonglong2(~0ull, ~0ull);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (base + (uint32_t)u * 2 * kSelBlock >= hi) break;        // this request of the whole workgroup is past the slice
            const uint32_t j = base + (uint32_t)u * 2 * kSelBlock + 2 * threadIdx.x;
            count(v[u].x, j < hi); count(v[u].y, j + 1 < hi);
        }
    }
    sel_flush(h, run);
    sel_finish_window<W>(h, hist + (size_t)q * kSelBins, s, gridDim.x, k, cap);
}

// keys up to the decided prefix (inclusive) -> cand[q][...]; when even 32 decided bits leave more ties than the sort holds
// (done == 0), only the keys strictly below the tied distance: the sort kernel adds the first k_rem ties in row order
__global__ void __launch_bounds__(kSelBlock)
k_select_compact(const uint64_t* __restrict__ keys, size_t stride, uint32_t n, uint32_t cap, SelState* __restrict__ st, uint64_t* __restrict__ cand,
                 const uint32_t* __restrict__ active = nullptr) {
    const uint32_t q = blockIdx.y;
    if (active != nullptr && q >= *active) return;
    SelState* s = st + q;
    const uint32_t sh = 64 - s->bits;
    const unsigned long long top = s->prefix >> sh;
    const bool incl = s->done != 0;
    const uint32_t lane = lane_id();
    const uint64_t* src = keys + (size_t)q * stride;
    uint64_t* dst = cand + (size_t)q * cap;
    const uint32_t per = ((n + gridDim.x - 1) / gridDim.x + 2 * kSelBlock - 1) / (2 * kSelBlock) * (2 * kSelBlock);
    const uint32_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (uint32_t base = lo; base < hi; base += 2 * 4 * kSelBlock) {
        ulonglong2 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t j = base + (uint32_t)u * 2 * kSelBlock + 2 * threadIdx.x;
            v[u] = j < hi ? *reinterpret_cast<const ulonglong2*>(src + j) : make_ulonglong2(~0ull, ~0ull);
        }
        // the round's eight keys per lane are placed with ONE returning atomic per wave (a wave whose lanes keep nothing — nearly all
        // of them over a scan's keys — issues none): a returning atomic per key pair and wave was a chain of up to eight round trips
        uint64_t key[8]; bool take[8]; uint32_t c = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t j = base + (uint32_t)u * 2 * kSelBlock + 2 * threadIdx.x;
            key[2 * u] = v[u].x; key[2 * u + 1] = v[u].y;
            const unsigned long long t0 = v[u].x >> sh, t1 = v[u].y >> sh;
            take[2 * u] = j < hi && (incl ? t0 <= top : t0 < top);
            take[2 * u + 1] = j + 1 < hi && (incl ? t1 <= top : t1 < top);
            c += (take[2 * u] ? 1u : 0u) + (take[2 * u + 1] ? 1u : 0u);
        }
        if (__ballot(c != 0) == 0) continue;
        const uint32_t inc = wave_incl_scan(c, lane);
        const uint32_t total = __builtin_amdgcn_readlane(inc, 63);
        uint32_t slot0 = 0;
        if (lane == 0) slot0 = atomicAdd(&s->n_cand, total);
        uint32_t slot = __builtin_amdgcn_readfirstlane(slot0) + inc - c;
#pragma unroll
        for (int u = 0; u < 8; u++) if (take[u]) { if (slot < cap) dst[slot] = key[u]; slot++; }
    }
}

// One workgroup per query: the kept keys (plus, in the tie case, the first k_rem tied keys in array order) are sorted in LDS
// (bitonic, full 64-bit keys) and the first kk written out; the rest of the k_stride slots are padded (0xFFFFFFFF, +inf).
// payload != null: the keys' low words are places in `payload_rows` (rows_out gets payload_rows[q][low word]) — used when
// the keys were built over a candidate list rather than over the corpus.
__global__ void __launch_bounds__(kSelSortBlock)
k_select_sort(const uint64_t* __restrict__ keys, size_t stride, uint32_t n, uint32_t kk, uint32_t k_stride, uint32_t cap,
              SelState* __restrict__ st, const uint64_t* __restrict__ cand, uint32_t* __restrict__ rows_out, float* __restrict__ dist_out, int ordered,
              const uint32_t* __restrict__ counts = nullptr /* direct form: live keys of query q are keys[q][0 .. counts[q]) */,
              const uint32_t* __restrict__ active = nullptr) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t* a = reinterpret_cast<uint64_t*>(smem);                   // [cap]
    __shared__ uint32_t wcnt[kSelSortBlock / 64];
    __shared__ uint32_t s_run;
    __shared__ uint32_t hh[kSelBins];                                  // histogram of the in-LDS radix passes
    __shared__ uint32_t r_d, r_cum, r_bucket, r_cnt;
    const uint32_t q = blockIdx.x;
    if (active != nullptr && q >= *active) return;
    SelState* s = st + q;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
#ifdef QV_SEL_PROF
    uint64_t tk[7]; int tn = 0;
#define SELTK() tk[tn++] = wall_clock64()
#else
#define SELTK()
#endif
    SELTK();
    // st == nullptr: no global windows ran — the query's n keys (n <= cap) come straight from the array and the selection happens
    // here (launch_select_topk: a few thousand keys per query, where six launches cost more than the work)
    const bool direct = st == nullptr;
    uint32_t nc = direct ? (counts != nullptr && counts[q] < n ? counts[q] : n) : (s->n_cand < cap ? s->n_cand : cap);
    const uint64_t* src = direct ? keys + (size_t)q * stride : cand + (size_t)q * cap;
    for (uint32_t i = threadIdx.x; i < nc; i += kSelSortBlock) a[i] = src[i];
    const uint32_t lane_ = lane, wave_ = wave;
    // hh[0 .. nb) holds a histogram (nb <= 4096, already synchronised): the digit d whose bin holds the krem-th key in bin order,
    // cum = keys in the bins below d, bucket = hh[d].  Every thread gets the result.
    auto find_digit = [&](uint32_t nb, uint32_t krem, uint32_t& d_out, uint32_t& cum_out, uint32_t& bucket_out) {
        uint32_t sum = 0;
        for (uint32_t u = 0; u < 4; u++) { const uint32_t b = threadIdx.x * 4 + u; sum += b < nb ? hh[b] : 0u; }
        const uint32_t inc = wave_incl_scan(sum, lane_);
        if (lane_ == 63) wcnt[wave_] = inc;
        // (fewer than krem keys in the histogram — a query with fewer live candidates than results asked for — leaves no thread to
        // write the answer: the last bin then, everything below it counted, an empty bucket)
        if (threadIdx.x == 0) { r_d = nb - 1; r_cum = 0; r_bucket = 0; }
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t x = 0; x < wave_; x++) before += wcnt[x];
        const uint32_t excl = before + inc - sum;
        if (excl < krem && krem <= excl + sum) {                       // exactly one thread (or none: see above)
            uint32_t cum = excl, d = threadIdx.x * 4;
            for (uint32_t u = 0; u < 4; u++, d++) { if (krem <= cum + hh[d]) break; cum += hh[d]; }
            r_d = d; r_cum = cum; r_bucket = hh[d];
        }
        __syncthreads();
        d_out = r_d; cum_out = r_cum; bucket_out = r_bucket;
        __syncthreads();
    };
    if (!direct && !s->done) {
        // more equal distances than the sort holds: the answer takes the k_rem of them with the smallest rows
        const unsigned long long tie = s->prefix >> 32;
        const uint32_t want = s->k_rem;
        const uint64_t* all = keys + (size_t)q * stride;
        if (threadIdx.x == 0) s_run = 0;
        // the slots the ties go to start out dead: when fewer than k_rem live ties exist (the k-th key fell among dead keys: a
        // handed-back query, a query with fewer live candidates than k) the rest is padding, not whatever LDS held
        for (uint32_t i = threadIdx.x; i < want; i += kSelSortBlock) a[nc + i] = kDeadKey;
        __syncthreads();
        if (ordered) {
        // the keys array is in row order — walk it front to back, a block of 4096 keys per round, and keep the first k_rem ties
        for (uint32_t base = 0; base < n; base += 4 * kSelSortBlock) {
            const uint32_t run = s_run;
            if (run >= want) break;
            uint64_t key[4]; uint32_t c = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {                             // thread t takes 4 consecutive keys: array order is kept by (thread, u)
                const uint32_t i = base + threadIdx.x * 4 + (uint32_t)u;
                key[u] = i < n ? all[i] : ~0ull;
                c += (key[u] >> 32) == tie && key[u] != kDeadKey ? 1u : 0u;
            }
            const uint32_t inc = wave_incl_scan(c, lane);
            if (lane == 63) wcnt[wave] = inc;
            __syncthreads();
            uint32_t before = run;
            for (uint32_t w = 0; w < wave; w++) before += wcnt[w];
            uint32_t pos = before + inc - c;
#pragma unroll
            for (int u = 0; u < 4; u++)
                if ((key[u] >> 32) == tie && key[u] != kDeadKey) { if (pos < want) a[nc + pos] = key[u]; pos++; }
            __syncthreads();
            if (threadIdx.x == kSelSortBlock - 1) s_run = pos;         // the last thread's end position = ties seen so far
            __syncthreads();
        }
        } else {
            // keys in no particular order (the waves' lists of k_flat_scan_wide, gathered shard lists): the k_rem-th smallest ROW
            // among the tied keys by three radix windows on the low word (12 + 12 + 8 bits), then every tie up to that row
            uint32_t rprefix = 0, rbits = 0, rrem = want;
            while (rbits < 32) {
                const uint32_t w = rbits < 24 ? 12u : 8u, shift = 32 - rbits - w, nb = 1u << w;
                for (uint32_t b = threadIdx.x; b < nb; b += kSelSortBlock) hh[b] = 0;
                __syncthreads();
                for (uint32_t i = threadIdx.x; i < n; i += kSelSortBlock) {
                    const uint64_t key = all[i];
                    const uint32_t row = (uint32_t)key;
                    if ((key >> 32) == tie && key != kDeadKey && (rbits == 0 || (row >> (32 - rbits)) == (rprefix >> (32 - rbits)))) atomicAdd(&hh[(row >> shift) & (nb - 1)], 1u);
                }
                __syncthreads();
                uint32_t d, cum, bucket;
                find_digit(nb, rrem, d, cum, bucket);
                rprefix |= d << shift; rrem -= cum; rbits += w;
            }
            if (threadIdx.x == 0) r_cnt = 0;
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < n; i += kSelSortBlock) {
                const uint64_t key = all[i];
                if ((key >> 32) == tie && key != kDeadKey && (uint32_t)key <= rprefix) { const uint32_t pos = atomicAdd(&r_cnt, 1u); if (pos < want) a[nc + pos] = key; }
            }
            __syncthreads();
        }
        nc += want;                                                    // below + k_rem = k <= cap
    }
    __syncthreads();
    SELTK();
    // Far more keys than wanted (the global windows stop as soon as the kept keys FIT the sort: 2 519 keys for the 100 best of
    // 10M rows): the same radix selection once more, inside LDS, 12 bits of the full 64-bit key per pass, until what is kept
    // fits the smallest power of two that holds kk — then only that many keys are sorted (128 instead of 4096).
    uint32_t m_need = 64;
    while (m_need < kk) m_need <<= 1;
    if (nc > m_need) {
        uint64_t mine[16];                                             // this thread's keys (cap <= 16384)
#pragma unroll
        for (int u = 0; u < 16; u++) { const uint32_t i = threadIdx.x + (uint32_t)u * kSelSortBlock; mine[u] = i < nc ? a[i] : kDeadKey; }
        unsigned long long prefix = 0; uint32_t bits = 0, krem = kk, below = 0;
        while (bits < 64) {
            const uint32_t w = 64 - bits < 12 ? 64 - bits : 12, shift = 64 - bits - w, nb = 1u << w;
            for (uint32_t b = threadIdx.x; b < nb; b += kSelSortBlock) hh[b] = 0;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const uint32_t i = threadIdx.x + (uint32_t)u * kSelSortBlock;
                if (i < nc && (bits == 0 || (mine[u] >> (64 - bits)) == (prefix >> (64 - bits)))) atomicAdd(&hh[(uint32_t)(mine[u] >> shift) & (nb - 1)], 1u);
            }
            __syncthreads();
            uint32_t d, cum, bucket;
            find_digit(nb, krem, d, cum, bucket);
            prefix |= (unsigned long long)d << shift; bits += w; krem -= cum; below += cum;
            if (below + bucket <= m_need) break;                       // (at 64 decided bits the bucket is the k-th key alone: below + 1 = kk)
        }
        if (threadIdx.x == 0) r_cnt = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 16; u++) {                                 // in place: every key is in a register by now
            const uint32_t i = threadIdx.x + (uint32_t)u * kSelSortBlock;
            if (i < nc && (mine[u] >> (64 - bits)) <= (prefix >> (64 - bits))) a[atomicAdd(&r_cnt, 1u)] = mine[u];
        }
        __syncthreads();
        nc = r_cnt;
    }
    uint32_t m = 64;
    while (m < nc) m <<= 1;
    for (uint32_t i = nc + threadIdx.x; i < m; i += kSelSortBlock) a[i] = kDeadKey;
    __syncthreads();
    SELTK();
    // Bitonic sort of m keys.  Every exchange at a distance below 64 stays inside a wave's registers (chunks of 64 consecutive
    // keys, shuffles, no barrier; four chunks per wave at a time so that the shuffles' latencies overlap): the chunks are sorted
    // outright first, and each merge phase goes through LDS only for its distances of 64 and more.
    const uint32_t nchunks = m >> 6, nwaves = kSelSortBlock / 64;
    auto in_wave = [&](uint64_t key, uint32_t j, bool up) {
        const uint32_t lo32 = __shfl_xor((uint32_t)key, (int)j), hi32 = __shfl_xor((uint32_t)(key >> 32), (int)j);
        const uint64_t other = ((uint64_t)hi32 << 32) | lo32;
        const bool lower = (lane & j) == 0;
        const uint64_t mn = key < other ? key : other, mx = key < other ? other : key;
        return (lower == up) ? mn : mx;
    };
    constexpr uint32_t R = 4;
    for (uint32_t c0 = wave * R; c0 < nchunks; c0 += nwaves * R) {
        uint64_t key[R];
#pragma unroll
        for (uint32_t r = 0; r < R; r++) key[r] = c0 + r < nchunks ? a[(c0 + r) * 64 + lane] : kDeadKey;
#pragma unroll
        for (uint32_t k2 = 2; k2 <= 64; k2 <<= 1)
#pragma unroll
            for (uint32_t j = k2 >> 1; j > 0; j >>= 1)
#pragma unroll
                for (uint32_t r = 0; r < R; r++) key[r] = in_wave(key[r], j, (((c0 + r) * 64 + lane) & k2) == 0);
#pragma unroll
        for (uint32_t r = 0; r < R; r++) if (c0 + r < nchunks) a[(c0 + r) * 64 + lane] = key[r];
    }
    __syncthreads();
    SELTK();
    for (uint32_t k2 = 128; k2 <= m; k2 <<= 1) {
        for (uint32_t j = k2 >> 1; j >= 64; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < m / 2; t += kSelSortBlock) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
                const uint64_t x = a[i], y = a[p];
                const bool up = (i & k2) == 0;
                if ((x > y) == up) { a[i] = y; a[p] = x; }
            }
            __syncthreads();
        }
        for (uint32_t c0 = wave * R; c0 < nchunks; c0 += nwaves * R) {
            uint64_t key[R];
#pragma unroll
            for (uint32_t r = 0; r < R; r++) key[r] = c0 + r < nchunks ? a[(c0 + r) * 64 + lane] : kDeadKey;
#pragma unroll
            for (uint32_t j = 32; j > 0; j >>= 1)
#pragma unroll
                for (uint32_t r = 0; r < R; r++) key[r] = in_wave(key[r], j, (((c0 + r) * 64) & k2) == 0);
#pragma unroll
            for (uint32_t r = 0; r < R; r++) if (c0 + r < nchunks) a[(c0 + r) * 64 + lane] = key[r];
        }
        __syncthreads();
    }
    SELTK();
    for (uint32_t i = threadIdx.x; i < k_stride; i += kSelSortBlock) {
        const uint64_t key = i < kk ? a[i] : kDeadKey;
        const bool dead = key == kDeadKey;
        rows_out[(size_t)q * k_stride + i] = dead ? 0xFFFFFFFFu : (uint32_t)key;
        dist_out[(size_t)q * k_stride + i] = dead ? __uint_as_float(0x7F800000u) : unord_f32((uint32_t)(key >> 32));
    }
#ifdef QV_SEL_PROF
    SELTK();
    if (threadIdx.x == 0 && q == 0) printf("select_sort: nc %u m %u | load %llu, refine %llu, chunk sort %llu, merges %llu, output %llu (x10 ns)\n", nc, m,
                                           tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2], tk[4] - tk[3], tk[5] - tk[4]);
#endif
#undef SELTK
}

// The direct form over lists whose live keys are a counted prefix (a batch's exact distances: survivors first, dead slots behind): only the
// prefix is read — 2 100 of 16 384 slots at k = 1000.  n <= 16384.
hipError_t launch_select_topk_counted(const uint64_t* d_keys, size_t stride, uint32_t n, const uint32_t* d_counts, uint32_t nq, uint32_t kk, uint32_t k_stride,
                                      uint32_t* d_rows_out, float* d_dist_out, hipStream_t s) {
    if (nq == 0 || kk == 0 || kk > (uint32_t)kMaxSelectK || kk > k_stride || n == 0 || n > 16384u || kk > n || d_counts == nullptr) return hipErrorInvalidValue;
    uint32_t m_sort = 64;
    while (m_sort < kk) m_sort <<= 1;
    const size_t lds_d = (size_t)std::max<uint32_t>(n, m_sort) * sizeof(uint64_t);
    hipError_t e = set_lds(k_select_sort, lds_d);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_select_sort, dim3(nq), dim3(kSelSortBlock), lds_d, s, d_keys, stride, n, kk, k_stride, n, (SelState*)nullptr, (const uint64_t*)nullptr, d_rows_out, d_dist_out, 0, d_counts);
    return hipGetLastError();
}

uint32_t select_cap(uint32_t kk) { return kk <= 4096 ? 8192u : 16384u; }   // sort capacity: twice the largest k it serves (64 / 128 KB of LDS)

size_t select_workspace_bytes(uint32_t nq, uint32_t kk) {
    return (size_t)nq * (sizeof(SelState) + (size_t)kSelBins * sizeof(uint32_t) + (size_t)select_cap(kk) * sizeof(uint64_t)) + 512;
}

// the workspace's layout: [nq] states, [nq][kSelBins] histogram words, [nq][cap] kept keys.  select_prepare zeroes the first two
// (a caller whose keys kernel counts window 0 itself does this BEFORE that kernel and passes window0_counted = true below)
__global__ void __launch_bounds__(256)
k_select_zero(uint4* __restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
hipError_t select_prepare(void* d_ws, uint32_t nq, uint32_t kk, SelState** st_out, uint32_t** hist_out, hipStream_t s) {
    char* w = static_cast<char*>(d_ws);
    const size_t st_bytes = ((size_t)nq * sizeof(SelState) + 255) / 256 * 256;
    *st_out = reinterpret_cast<SelState*>(w);
    *hist_out = reinterpret_cast<uint32_t*>(w + st_bytes);
    (void)kk;
    // (a kernel of our own: hipMemsetAsync took 43 us for the 4 MB of a 256-query batch)
    const size_t n16 = (st_bytes + (size_t)nq * kSelBins * sizeof(uint32_t)) / 16;
    hipLaunchKernelGGL(k_select_zero, dim3((unsigned)std::min<size_t>(1024, (n16 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<uint4*>(w), n16);
    return hipGetLastError();
}

hipError_t launch_select_topk(const uint64_t* d_keys, size_t stride, uint32_t n, uint32_t nq, uint32_t kk, uint32_t k_stride, void* d_ws,
                              uint32_t* d_rows_out, float* d_dist_out, hipStream_t s, bool window0_counted, bool ordered, const uint32_t* d_active) {
    if (nq == 0 || kk == 0 || kk > (uint32_t)kMaxSelectK || kk > k_stride || (n & 1u) || (stride & 1u) || n == 0 || kk > n) return hipErrorInvalidValue;
    const uint32_t cap = select_cap(kk);
    char* w = static_cast<char*>(d_ws);
    SelState* st = reinterpret_cast<SelState*>(w);
    uint32_t* hist = reinterpret_cast<uint32_t*>(w + (((size_t)nq * sizeof(SelState) + 255) / 256 * 256));
    uint64_t* cand = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(hist) + (size_t)nq * kSelBins * sizeof(uint32_t));
    hipError_t e = hipSuccess;
    // 8 KiB of keys per workgroup and round; enough workgroups to fill the chip, no more than the keys can feed
    const uint32_t grid = std::max(1u, std::min(1024u, (n + 4 * 2 * kSelBlock - 1) / (4 * 2 * kSelBlock)));
    if (!window0_counted && n <= 16384u) {
        // few keys per query (the candidate lists of a batch: 4096-16384 slots): one workgroup per query takes them all into LDS
        uint32_t m_sort = 64;                                             // the sort pads what it keeps to a power of two that holds kk
        while (m_sort < kk) m_sort <<= 1;
        const size_t lds_d = (size_t)std::max<uint32_t>(n, m_sort) * sizeof(uint64_t);
        e = set_lds(k_select_sort, lds_d);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_select_sort, dim3(nq), dim3(kSelSortBlock), lds_d, s, d_keys, stride, n, kk, k_stride, n, (SelState*)nullptr, (const uint64_t*)nullptr, d_rows_out, d_dist_out, 0, (const uint32_t*)nullptr, d_active);
        return hipGetLastError();
    }
    if (!window0_counted) {
        e = select_prepare(d_ws, nq, kk, &st, &hist, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_select_hist<0>, dim3(grid, nq), dim3(kSelBlock), 0, s, d_keys, stride, n, kk, cap, st, hist, d_active);
    }
    hipLaunchKernelGGL(k_select_hist<1>, dim3(grid, nq), dim3(kSelBlock), 0, s, d_keys, stride, n, kk, cap, st, hist, d_active);
    hipLaunchKernelGGL(k_select_hist<2>, dim3(grid, nq), dim3(kSelBlock), 0, s, d_keys, stride, n, kk, cap, st, hist, d_active);
    hipLaunchKernelGGL(k_select_compact, dim3(grid, nq), dim3(kSelBlock), 0, s, d_keys, stride, n, cap, st, cand, d_active);
    const size_t lds = (size_t)cap * sizeof(uint64_t);
    e = set_lds(k_select_sort, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_select_sort, dim3(nq), dim3(kSelSortBlock), lds, s, d_keys, stride, n, kk, k_stride, cap, st, cand, d_rows_out, d_dist_out, ordered ? 1 : 0, (const uint32_t*)nullptr, d_active);
    return hipGetLastError();
}

}  // namespace qv
