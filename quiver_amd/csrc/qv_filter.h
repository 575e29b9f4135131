// qv_filter.h — what the batched filter kernels share: the error model of the filters (filter_gamma), operand helpers, the
// per-wave candidate queue and the filter epilogue (level 1 / dump / dense pass).  Included by qv_batched.hip and qv_qreg.hip.
#pragma once
#include "qv_select.h"

namespace qv {

typedef float f16v __attribute__((ext_vector_type(16)));
// next representable float towards +inf / -inf (directed rounding of the filter constants)
__device__ __forceinline__ float f32_up(float x) {
    if (!(x == x) || x == __uint_as_float(0x7F800000u)) return x;
    if (x == 0.0f) return __uint_as_float(1u);
    uint32_t u = __float_as_uint(x);
    return __uint_as_float(x > 0.0f ? u + 1 : u - 1);
}
__device__ __forceinline__ float f32_down(float x) { return -f32_up(-x); }
constexpr int kMfmaCandCap = 4096;        // candidate slots per query

// |S~ - S| <= filter_gamma * |q||r| for the two filter kernels.
//   fp32 MFMA (k_mfma_filter): a chain of K rounded fp32 multiply-adds: gamma_K = (K+2)u / (1 - (K+2)u), u = 2^-24.
//   bf16 x 3 (k_bf16x3_filter): every operand is split q = qh + ql + rq, x = xh + xl + rx with qh = bf16(q), ql = bf16(q - qh)
//     (round to nearest: |q - qh| <= 2^-8 |q|, q - qh exact in float32, |rq| <= 2^-16 |q|, likewise x) and the kernel sums the
//     three products qh*xh + qh*xl + ql*xh, each EXACT in float32 (8 x 8 significand bits), in float32 accumulators.
//     Dropped: ql*xl + rq*x + q*rx, at most 3.03 * 2^-16 |q_i||x_i| per element, hence (Cauchy-Schwarz) 4.63e-5 |q||r|.
//     Accumulation: 3K + 2 float32 additions in whatever order the matrix core takes; allowing a full ulp per addition
//     (u' = 2^-23, i.e. even a truncating adder) over terms of total magnitude <= 1.012 |q||r|.
//   bf16 x 1 (k_bf16x3_filter_shared<.., 1>, mode 2 here): only qh*xh.  Dropped: (q - qh)*x + qh*(x - xh), at most (2 * 2^-8 + 2^-16)
//     |q_i||x_i| per element, hence 7.83e-3 |q||r|; K + 2 additions.  A third of the matrix work for a filter that passes ~8 rows
//     per query instead of ~1 on unstructured 768-d data (the exact re-score decides either way).
//     Round 3: that 7.83e-3 is a worst case over operands (every element at the far end of its rounding interval, q and r
//     parallel in absolute value).  The dropped part is bounded just as rigorously by what the operands REALLY lose,
//       |(q - qh).r + qh.(r - rh)| <= |q - qh| |r| + |qh| |r - rh|        (Cauchy-Schwarz on each term),
//     with |q - qh|, |qh| computed per query (k_mfma_prep) and |r - rh| kept per row (IndexView::rres, k_row_residual): round to
//     nearest leaves ~0.38 * 2^-8 of a vector's norm on ordinary data, so the window is 2.6 x narrower — a third of the rows pass
//     the filter at the same sample bound and a quarter survive the interval test into the exact pass.  filter_gamma(., 2) is
//     then only the accumulation part (filter_gamma_acc); the 7.83e-3 stays for callers without per-row data.
__host__ __device__ static inline double filter_gamma_acc(uint32_t dim) {     // K + 2 float32 additions of the one-term kernel
    const double g = (double)(dim + 2) * 1.1920928955078125e-7;
    return 1.008 * g / (1.0 - g);
}
__host__ __device__ static inline double filter_gamma(uint32_t dim, int mode /* 0 fp32, 1 bf16 x 3, 2 bf16 x 1 */) {
    if (!mode) { const double g = (double)(dim + 2) * 5.9604644775390625e-8; return g / (1.0 - g); }
    if (mode == 2) { const double g = (double)(dim + 2) * 1.1920928955078125e-7; return 1.008 * g / (1.0 - g) + 7.83e-3; }
    const double g = (double)(3 * dim + 2) * 1.1920928955078125e-7;
    return 1.012 * g / (1.0 - g) + 4.63e-5;
}
// Norms below which a query or a row bypasses the filter (the row goes to the exact pass whatever its score; the query gets the
// "everything is a candidate" threshold).  The error model above has no underflow in it: a matrix core may flush operands, products
// or partial sums below 2^-126, each such product is lost whole (<= 1.2e-38), up to `dim` of them per score.  That stays inside the
// slack the thresholds keep in reserve (5e-7 |q||r|) as long as |q||r| >= dim * 2.4e-32, i.e. when both norms are at least
// sqrt(dim * 2.4e-32) (1e-14 up to 4096 dimensions).  Data of that scale is not a workload; the guard makes the claim unconditional.
__host__ __device__ static inline float filter_tiny_norm(uint32_t dim) {
    const float t = __builtin_sqrtf((float)dim * 2.4e-32f);
    return t > 1.0e-14f ? t : 1.0e-14f;
}
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
// two floats -> (hi pair, lo pair) of bfloat16, packed: hi = bf16(x) (RNE), lo = bf16(x - hi) (x - hi is exact)
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    const bf2 h = {(__bf16)x0, (__bf16)x1};
    hi = __builtin_bit_cast(uint32_t, h);
    const float h0 = __uint_as_float(hi << 16), h1 = __uint_as_float(hi & 0xFFFF0000u);
    const bf2 l = {(__bf16)(x0 - h0), (__bf16)(x1 - h1)};
    lo = __builtin_bit_cast(uint32_t, l);
}

// sum over the wave's 64 lanes (every lane gets it); used for norms that only feed error bounds with 1e-6 of slack, where the
// order of the additions does not matter
__device__ __forceinline__ double wave_sum_f64(double x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
    return x;
}
// a wave's first batch into its empty list: one bitonic sort instead of up to 64 serial inserts
__device__ __forceinline__ void list_seed(uint64_t& list, uint64_t& thr, uint64_t key, uint32_t kth_lane, uint32_t lane) {
    list = wave_sort64(key, lane);
    thr = readlane64(list, kth_lane);
}
__device__ __forceinline__ uint32_t pack_bf16(float x0, float x1) { const bf2 h = {(__bf16)x0, (__bf16)x1}; return __builtin_bit_cast(uint32_t, h); }

// Which query block and which row walk a persistent filter workgroup takes.  `wgs` workgroups (one per block of 256 queries) walk the
// same rows; with more than one, those go to the same XCD — workgroup b runs on XCD b mod 8 — so that a row group comes out of HBM
// once and out of that XCD's L2 for the others (1024 x 1M x 768 on k_qreg_filter: 2.29 -> 2.16 ms).
__device__ __forceinline__ void filter_block_role(uint32_t wgs, uint32_t& qblock, uint32_t& first_walk) {
    qblock = blockIdx.x % wgs; first_walk = blockIdx.x / wgs;
    if (wgs > 1 && gridDim.x % (8 * wgs) == 0) {
        const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
        qblock = slot % wgs;
        first_walk = (slot / wgs) * 8 + xcd;
    }
}

// The per-wave kernels (k_mfma_filter, k_bf16x3_filter): workgroup L's four waves take query blocks 4 L .. 4 L + 3 (mod nqb64), so `share` =
// nqb64 / 4 consecutive workgroups walk the same row groups.  The same placement rule: those go to one XCD.  Returns the workgroup's
// logical number L (blockIdx.x itself when the grid does not divide).
__device__ __forceinline__ uint32_t filter_logical_block(uint32_t share) {
    if (share <= 1 || gridDim.x % (8 * share) != 0) return blockIdx.x;
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    return ((slot / share) * 8 + xcd) * share + slot % share;
}

// Candidate queue of a wave.  A row that passes the filter test used to be appended to its query's list with a RETURNING global
// atomic (the slot), inside an epilogue that also spilled around itself.  Vector-memory operations complete in issue order per
// wave (`s_waitcnt vmcnt` is one counter), so waiting for that slot — or for any vector-memory result: a spill reload is one too
// — also waits for every row request the wave has in flight, i.e. the prefetch ring of the NEXT row group, issued during the last
// steps of this one.  Hits now go to a queue in LDS that belongs to the wave (its fill count is a wave-uniform register, the
// slot is count + the lane's rank in the ballot: no atomic at all, nothing in the append path touches vector memory), and the
// wave appends its queue to the per-query lists itself when it fills up (rare) and after its last row group.
constexpr uint32_t kCandQueueWords = 3;                               // {query, row, score bits}
struct CandQueue { uint32_t* rec; uint32_t cap; };                    // this WAVE's records in LDS (cap >= 64)

// cap: slots per query.  kMfmaCandCap for k <= 64; a batch asking for more per query (k up to kMaxBatchedK) gets batched_cand_cap(k).
// The filter kernels read it from the word in front of the counters (cand_cnt[-1], written by k_mfma_prep), so that none of their
// signatures changes with it.
__device__ __forceinline__ void cand_append_global(uint32_t q, uint32_t row, float score, uint32_t* __restrict__ cand_rows, float* __restrict__ cand_score,
                                                   uint32_t* __restrict__ cand_cnt, uint32_t cap) {
    const uint32_t slot = atomicAdd(&cand_cnt[q], 1u);
    if (slot < cap) {
        cand_rows[(size_t)q * cap + slot] = row;
        cand_score[(size_t)q * cap + slot] = score;
    }
}
struct CandOut { uint32_t* rows; float* score; uint32_t* cnt; uint32_t cap; };      // the per-query candidate lists in global memory
// every lane of the wave calls this: the wave's n queued records go to the per-query lists
__device__ __forceinline__ void cand_flush(const CandQueue& cqu, uint32_t n, const CandOut& out) {
    __threadfence_block();
    for (uint32_t i = lane_id(); i < n; i += 64)
        cand_append_global(cqu.rec[kCandQueueWords * i], cqu.rec[kCandQueueWords * i + 1], __uint_as_float(cqu.rec[kCandQueueWords * i + 2]), out.rows, out.score, out.cnt, out.cap);
    __threadfence_block();
}
// called by ALL lanes of a wave (converged) with the wave's mask m (non-zero) of the lanes that append their record; n = the queue's fill
__device__ __forceinline__ void cand_push(const CandQueue& cqu, uint32_t& n, uint64_t m, uint32_t q, uint32_t row, float score, const CandOut& out) {
    const uint32_t lane = lane_id();
    const uint32_t cnt = (uint32_t)__builtin_popcountll(m);
    if (__builtin_expect(n + cnt > cqu.cap, 0)) { cand_flush(cqu, n, out); n = 0; }
    if ((m >> lane) & 1ull) {
        uint32_t* d = cqu.rec + kCandQueueWords * (n + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull)));
        d[0] = q; d[1] = row; d[2] = __float_as_uint(score);
    }
    n += cnt;
}
#define QV_EPI_DUMP(name, WAVES, CAP)                                         \
    __shared__ __align__(16) uint32_t name##_area[(WAVES) * (CAP) * kEpiEntryWords]; \
    const EpiDump name{name##_area + (threadIdx.x >> 6) * (CAP) * kEpiEntryWords, (CAP)}
#define QV_CAND_QUEUE(name, WAVES, CAP)                                       \
    __shared__ uint32_t name##_rec[(WAVES) * (CAP) * kCandQueueWords];        \
    const CandQueue name{name##_rec + (threadIdx.x >> 6) * (CAP) * kCandQueueWords, (CAP)}; \
    uint32_t name##_n = 0;                                                    \
    const CandOut name##_out{cand_rows, cand_score, cand_cnt, cand_cnt[-1]}

// The filter test on a wave's 64 x 128 scores: acc[i][j][r] = S~[query 64*qb64 + 32*i + (r&3)+8*(r>>2)+4*half][row 64*(t0|t1) + 32*(j&1) + l31];
// a row that may be in some query's top-k goes to that query's candidate list with its score.
//
// Cost matters here: 128 scores per lane and row group, of which ~1 in 1400 passes, on a lone wave per SIMD that pays ~13 cycles
// per instruction (nobody to cover its dependencies).  Testing every score against its own query's threshold, and entering an
// unrolled 16-way append sequence in every block in which any lane hit, was 1 300 instructions per wave and row group: 30 % of the
// kernel (profiles/r03_batched_epilogue.txt: 855 us with, 594 us without any epilogue).  Three steps now:
//   level 1, per lane and 16-score block (the part that must touch every score): ONE compare of the block's maximum (8 v_max3)
//     against a lower bound of the block's 16 thresholds — the smallest c and largest m among the 16 queries of that block and
//     lane half (EpiConsts, once per wave).  ~1-3 % of the lanes pass.
//   dump: a lane that passes writes its 16 scores and its row constants to the wave's dump area in LDS (80 bytes).
//   dense pass, once per row group: the dumped (entry, score) pairs are spread over the 64 lanes — 4 entries per pass — and each
//     lane does the exact per-query test for its pair and appends.  The sparse work of a few lanes becomes a dense wave's work.
// No score is NaN or infinite when both norms are below 1e18 (|partial sum| <= |q||r| < 1e36), so overflow is guarded per ROW
// here (a norm that is NaN, infinite or >= 1e18 passes both tests outright: the exact pass decides) and per QUERY in k_mfma_prep
// (such a query gets the "everything is a candidate" threshold), not per score.
struct EpiConsts { float cmin[2], mmax[2], bmax[2]; };
constexpr uint32_t kEpiQ = 256;                                      // query slots of a workgroup's constants: sm[0..256) = m_q, sm[256..512) = b_q
// sc / sm: the workgroup's filter constants in LDS (one float per query); qbase: the wave's first query in them; NI 32-query blocks
template <int METRIC, int NI = 2>
__device__ __forceinline__ EpiConsts epi_consts(const float* sc, const float* sm, uint32_t qbase, uint32_t half) {
    EpiConsts e;
    e.cmin[1] = e.mmax[1] = e.bmax[1] = 0.f;
#pragma unroll
    for (int i = 0; i < NI; i++) {
        float cm = __uint_as_float(0x7F800000u), mm = 0.f, bm = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t ql = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
            cm = fminf(cm, sc[qbase + ql]); mm = fmaxf(mm, sm[qbase + ql]); bm = fmaxf(bm, sm[kEpiQ + qbase + ql]);
        }
        e.cmin[i] = cm; e.mmax[i] = mm; e.bmax[i] = bm;
    }
    return e;
}
template <int METRIC>
__device__ __forceinline__ float filter_threshold(float c, float m, float b, float rn, float rn2c, float rho) {
    // rho = |r - bf16(r)| and b = the query's |qh| (one-term filter; b = 0 for the others): see filter_gamma
    return METRIC == QV_COSINE ? c * rn - b * rho - 1e-30f : (METRIC == QV_DOT ? c - m * rn - b * rho : 0.5f * (c + rn2c - m * rn - b * rho));
}
// (1-2e-6)|r|^2 rounded down, from rn = the row norm rounded to float32 and one step up (L2 family)
__device__ __forceinline__ float rn2c_of(float rn) {
    const float rlo = f32_down(f32_down(rn));
    return f32_down(f32_down(rlo * rlo) * 0.999998f);
}
constexpr uint32_t kEpiEntryWords = 20;                               // 16 scores, row, |r| up, |r - bf16(r)| up, flags (i | half << 1 | unsure << 2)
struct EpiDump { uint32_t* area; uint32_t cap; };                     // this WAVE's dump area in LDS, cap entries of kEpiEntryWords words (16-byte aligned)

// the dense pass: entries [0, n) of the wave's dump area, 16 (entry, score) pairs per entry, 64 pairs per round
template <int METRIC>
__device__ __forceinline__ void epi_dense_pass(const EpiDump& du, uint32_t n, const float* sc, const float* sm, uint32_t qbase, uint32_t qglobal,
                                               const CandQueue& cqu, uint32_t& cqn, const CandOut& out) {
    __threadfence_block();                                            // the dump's LDS writes, before other lanes of the wave read them
    const uint32_t lane = lane_id();
    for (uint32_t p = lane; p < 16 * n + 63; p += 64) {              // the loop count is wave-uniform (cand_push is a wave operation)
        bool take = false; uint32_t q = 0, row = 0; float score = 0.f;
        if (p < 16 * n) {
            const uint32_t* d = du.area + (p >> 4) * kEpiEntryWords;
            const uint32_t r = p & 15, fl = d[19];
            const uint32_t ql = 32 * (fl & 1) + (r & 3) + 8 * (r >> 2) + 4 * ((fl >> 1) & 1);
            score = __uint_as_float(d[r]); row = d[16];
            const float c = sc[qbase + ql];
            const float rn = __uint_as_float(d[17]);
            const float thr = filter_threshold<METRIC>(c, sm[qbase + ql], sm[kEpiQ + qbase + ql], rn, rn2c_of(rn), __uint_as_float(d[18]));
            take = (!(score < thr) || (fl >> 2)) && c < 3.0e38f;      // (padded query slots carry +inf)
            if (fl >> 2) score = __builtin_nanf("");                  // a row the scores say nothing about: the exact pass must not read an interval out of this one either
            q = qglobal + ql;
        }
        const uint64_t m = __ballot(take);
        if (m) cand_push(cqu, cqn, m, q, row, score, out);
        if (p - lane + 64 >= 16 * n) break;
    }
}

// acc[i][j][r] = S~[query qglobal + 32*i + (r&3)+8*(r>>2)+4*half][row 64*(t0|t1) + 32*(j&1) + l31]; the wave's queries start at sc[qbase] / sm[qbase]
// The epilogue of one row group in pieces — row block J (level 1 + dump), then the dense pass — so that a kernel can run them one
// at a time between the steps of the NEXT group's K loop (k_bf16x1_filter_w8), where they cost memory-wait time instead of their own.
// n: entries in the wave's dump area (wave-uniform), carried from block to block.
template <int METRIC, int NI, int NJ, int J>
__device__ __forceinline__ void filter_epilogue_block(const f16v (&acc)[NI][NJ], uint32_t t0, uint32_t t1, const float* sc, const float* sm,
                                                      uint32_t qbase, uint32_t half, uint32_t l31, uint32_t qglobal, float tiny_rn, const EpiConsts& ec,
                                                      const double (&rnd)[NJ], const float (&rho)[NJ], const uint64_t (&alv)[NJ / 2], const CandQueue& cqu, uint32_t& cqn,
                                                      const CandOut& out, const EpiDump& du, uint32_t& n) {
    constexpr int j = J;
    const uint32_t t = j < 2 ? t0 : t1;
    if (j >= 2 && t1 == t0) return;
    const uint32_t row = t * 64 + 32 * (j & 1) + l31;
    const bool live = (alv[j >> 1] >> (32 * (j & 1) + l31)) & 1ull;
    const float rn = f32_up((float)rnd[j]);
    const float rn2c = rn2c_of(rn);
    // rows the scores say nothing about: bf16 operands of a vanishing row would flush; a norm that is NaN, infinite or huge
    // may have overflowed the float32 sums (in either direction, possibly only on the way)
    const bool unsure = rn < tiny_rn || !(rn < 1.0e18f);
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const f16v& a = acc[i][j];
        const float m0 = __builtin_fmaxf(__builtin_fmaxf(a[0], a[1]), a[2]), m1 = __builtin_fmaxf(__builtin_fmaxf(a[3], a[4]), a[5]);
        const float m2 = __builtin_fmaxf(__builtin_fmaxf(a[6], a[7]), a[8]), m3 = __builtin_fmaxf(__builtin_fmaxf(a[9], a[10]), a[11]);
        const float m4 = __builtin_fmaxf(__builtin_fmaxf(a[12], a[13]), a[14]);
        const float mx = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(m0, m1), m2), __builtin_fmaxf(__builtin_fmaxf(m3, m4), a[15]));
        // monotone in c (up), m and b (down) operation by operation, so this is a lower bound of every one of the 16 thresholds
        const float thr_lo = filter_threshold<METRIC>(ec.cmin[i], ec.mmax[i], ec.bmax[i], rn, rn2c, rho[j]);
        const bool pre = (!(mx < thr_lo) || unsure) && live;
        const uint64_t pm = __ballot(pre);
#if defined(QV_DBG_EPI) && QV_DBG_EPI == 2                               // measurement build: level 1 only
        if (pm == 0x123456789abcull) out.cnt[0] = 1;
        continue;
#endif
        if (__builtin_expect(pm != 0, 0)) {                 // some row of the block may be in some query's top-k
            const uint32_t cnt = (uint32_t)__builtin_popcountll(pm), rank = (uint32_t)__builtin_popcountll(pm & ((1ull << lane_id()) - 1ull));
            for (uint32_t done = 0; done < cnt;) {          // one round, unless the dump area fills up (wave-uniform loop)
                if (n == du.cap) { epi_dense_pass<METRIC>(du, n, sc, sm, qbase, qglobal, cqu, cqn, out); n = 0; __threadfence_block(); }
                const uint32_t now = (du.cap - n) < (cnt - done) ? (du.cap - n) : (cnt - done);
                if (pre && rank >= done && rank < done + now) {
                    uint32_t* d = du.area + (n + rank - done) * kEpiEntryWords;
                    f4* d4 = reinterpret_cast<f4*>(d);
                    d4[0] = f4{a[0], a[1], a[2], a[3]}; d4[1] = f4{a[4], a[5], a[6], a[7]};
                    d4[2] = f4{a[8], a[9], a[10], a[11]}; d4[3] = f4{a[12], a[13], a[14], a[15]};
                    d[16] = row; d[17] = __float_as_uint(rn); d[18] = __float_as_uint(rho[j]); d[19] = (uint32_t)i | (half << 1) | ((uint32_t)unsure << 2);
                }
                n += now; done += now;
            }
        }
    }
}
// level 1 of one row block, branch-free: the mask of lanes whose block maximum reaches the lower bound of the block's 16 thresholds
template <int METRIC>
__device__ __forceinline__ uint64_t epi_level1(const f16v& a, float cmin, float mmax, float bmax, float rn, float rn2c, float rho, bool unsure, bool live) {
    const float m0 = __builtin_fmaxf(__builtin_fmaxf(a[0], a[1]), a[2]), m1 = __builtin_fmaxf(__builtin_fmaxf(a[3], a[4]), a[5]);
    const float m2 = __builtin_fmaxf(__builtin_fmaxf(a[6], a[7]), a[8]), m3 = __builtin_fmaxf(__builtin_fmaxf(a[9], a[10]), a[11]);
    const float m4 = __builtin_fmaxf(__builtin_fmaxf(a[12], a[13]), a[14]);
    const float mx = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(m0, m1), m2), __builtin_fmaxf(__builtin_fmaxf(m3, m4), a[15]));
    const float thr_lo = filter_threshold<METRIC>(cmin, mmax, bmax, rn, rn2c, rho);
    return __ballot((!(mx < thr_lo) || unsure) && live);
}
// the dump of one row block whose level-1 mask pm is not empty (wave-uniform call)
template <int METRIC>
__device__ __forceinline__ void epi_dump(const f16v& a, uint64_t pm, uint32_t row, float rn, float rho, uint32_t i, uint32_t half, bool unsure,
                                         const float* sc, const float* sm, uint32_t qbase, uint32_t qglobal, const CandQueue& cqu, uint32_t& cqn,
                                         const CandOut& out, const EpiDump& du, uint32_t& n) {
    const bool pre = (pm >> lane_id()) & 1ull;
    const uint32_t cnt = (uint32_t)__builtin_popcountll(pm), rank = (uint32_t)__builtin_popcountll(pm & ((1ull << lane_id()) - 1ull));
    for (uint32_t done = 0; done < cnt;) {                  // one round, unless the dump area fills up (wave-uniform loop)
        if (n == du.cap) { epi_dense_pass<METRIC>(du, n, sc, sm, qbase, qglobal, cqu, cqn, out); n = 0; __threadfence_block(); }
        const uint32_t now = (du.cap - n) < (cnt - done) ? (du.cap - n) : (cnt - done);
        if (pre && rank >= done && rank < done + now) {
            uint32_t* d = du.area + (n + rank - done) * kEpiEntryWords;
            f4* d4 = reinterpret_cast<f4*>(d);
            d4[0] = f4{a[0], a[1], a[2], a[3]}; d4[1] = f4{a[4], a[5], a[6], a[7]};
            d4[2] = f4{a[8], a[9], a[10], a[11]}; d4[3] = f4{a[12], a[13], a[14], a[15]};
            d[16] = row; d[17] = __float_as_uint(rn); d[18] = __float_as_uint(rho); d[19] = i | (half << 1) | ((uint32_t)unsure << 2);
        }
        n += now; done += now;
    }
}
template <int METRIC>
__device__ __forceinline__ void filter_epilogue_finish(const float* sc, const float* sm, uint32_t qbase, uint32_t qglobal, const CandQueue& cqu, uint32_t& cqn,
                                                       const CandOut& out, const EpiDump& du, uint32_t& n) {
#if defined(QV_DBG_EPI) && QV_DBG_EPI == 3                               // measurement build: level 1 + dump, no dense pass
    if (n == 0xFFFFFFFFu) out.cnt[0] = 1;
    n = 0;
    return;
#endif
    if (n) epi_dense_pass<METRIC>(du, n, sc, sm, qbase, qglobal, cqu, cqn, out);
    n = 0;
}
template <int METRIC, int NI, int NJ>
__device__ __forceinline__ void filter_epilogue(const IndexView& v, const f16v (&acc)[NI][NJ], uint32_t t0, uint32_t t1, const float* sc, const float* sm,
                                                uint32_t qbase, uint32_t half, uint32_t l31, uint32_t qglobal, float tiny_rn, const EpiConsts& ec,
                                                const double (&rnd)[NJ], const float (&rho)[NJ], const uint64_t (&alv)[NJ / 2], const CandQueue& cqu, uint32_t& cqn, const CandOut& out,
                                                const EpiDump& du) {
        uint32_t* const cand_cnt = out.cnt; (void)cand_cnt; (void)v;
#if defined(QV_DBG_EPI) && QV_DBG_EPI == 1                               // measurement build: no epilogue (the accumulators stay live through one compare)
        {
            float sdbg = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; j++)
#pragma unroll
                for (int i = 0; i < NI; i++)
#pragma unroll
                    for (int r = 0; r < 16; r++) sdbg += acc[i][j][r];
            if (sdbg == 1.2345678f) cand_cnt[0] = 1;
            return;
        }
#endif
        if constexpr (NI == 1) {
            // one 32-query block per wave (the eight-wave kernel): block after block measures 1 % faster than the two phases below (608 against 615 us)
            uint32_t n1 = 0;
#define QV_EPI_BLK(JJ) if constexpr (JJ < NJ) filter_epilogue_block<METRIC, NI, NJ, (JJ < NJ ? JJ : 0)>(acc, t0, t1, sc, sm, qbase, half, l31, qglobal, tiny_rn, ec, rnd, rho, alv, cqu, cqn, out, du, n1)
            QV_EPI_BLK(0); QV_EPI_BLK(1); QV_EPI_BLK(2); QV_EPI_BLK(3);
#undef QV_EPI_BLK
            filter_epilogue_finish<METRIC>(sc, sm, qbase, qglobal, cqu, cqn, out, du, n1);
            return;
        }
        // Two phases (two 32-query blocks per wave: eight row blocks; k_bf16rows_filter 437 -> 423 us, k_mfma_filter 3.06 -> 3.03 ms).
        // Level 1 of ALL row blocks first, branch-free: the blocks' max trees and threshold products are independent chains the
        // compiler can interleave (a wave-uniform branch after each block's ballot kept them one behind the other: ~500 cycles a block on
        // waves that all sit in their epilogue at the same time).  Then the dumps of the blocks whose mask is not empty.
        uint32_t n = 0;
        uint64_t pm[NJ][NI];
        float rnv[NJ]; bool unsv[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const bool live = (alv[j >> 1] >> (32 * (j & 1) + l31)) & 1ull;
            const float rn = f32_up((float)rnd[j]);
            // rows the scores say nothing about: bf16 operands of a vanishing row would flush; a norm that is NaN, infinite or huge
            // may have overflowed the float32 sums (in either direction, possibly only on the way)
            const bool unsure = rn < tiny_rn || !(rn < 1.0e18f);
            rnv[j] = rn; unsv[j] = unsure;
            const float rn2c = rn2c_of(rn);
#pragma unroll
            for (int i = 0; i < NI; i++)
                pm[j][i] = (j >= 2 && t1 == t0) ? 0ull : epi_level1<METRIC>(acc[i][j], ec.cmin[i], ec.mmax[i], ec.bmax[i], rn, rn2c, rho[j], unsure, live);
        }
#if !(defined(QV_DBG_EPI) && QV_DBG_EPI == 2)                             // (measurement build 2: level 1 only)
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const uint32_t row = (j < 2 ? t0 : t1) * 64 + 32 * (j & 1) + l31;
#pragma unroll
            for (int i = 0; i < NI; i++)
                if (__builtin_expect(pm[j][i] != 0, 0))     // some row of the block may be in some query's top-k
                    epi_dump<METRIC>(acc[i][j], pm[j][i], row, rnv[j], rho[j], (uint32_t)i, half, unsv[j], sc, sm, qbase, qglobal, cqu, cqn, out, du, n);
        }
#else
        if (pm[0][0] == 0x123456789abcull) cand_cnt[0] = 1;
#endif
        filter_epilogue_finish<METRIC>(sc, sm, qbase, qglobal, cqu, cqn, out, du, n);
}

// the group's row norms and alive words, requested before the K loop so that their latency is not the epilogue's
template <int NJ>
__device__ __forceinline__ void filter_row_consts(const IndexView& v, uint32_t t0, uint32_t t1, uint32_t l31, double (&rnd)[NJ], float (&rho)[NJ], uint64_t (&alv)[NJ / 2]) {
#pragma unroll
    for (int j = 0; j < NJ; j++) { const size_t row = (size_t)(j < 2 ? t0 : t1) * 64 + 32 * (j & 1) + l31; rnd[j] = v.rnorm[row]; rho[j] = v.rres[row]; }
    alv[0] = v.alive[t0];
    if constexpr (NJ == 4) alv[1] = v.alive[t1];
}

}  // namespace qv
