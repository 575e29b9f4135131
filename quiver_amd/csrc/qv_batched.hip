// qv_batched.hip — batched filter on the matrix cores (fp32 MFMA chain, or three exact-product bfloat16 terms) + exact re-scoring
// (shared helpers, the arithmetic contract and the build flags: qv_kernels.h)
#include <numeric>
#include "qv_filter.h"

namespace qv {

// ---------------------------------------------------------------- MFMA batched path --
// Batched queries x corpus is a dense fp32 GEMM (AI = Q/2 flop/B): scores S[q][r] = <q, r>
// by v_mfma_f32_32x32x2_f32.  The fp32 result is only a FILTER: a row becomes a candidate
// of query q when its approximate score cannot rule it out of q's top-k given a rigorous
// error margin; candidates are then re-scored by the exact kernels, so the final result is
// bit-identical to qv_index_search.
//
//   exact:   row r is in q's top-k  =>  d(q,r) <= U_q, U_q = exact k-th distance over a
//            SAMPLE of the corpus (any subset's k-th best bounds the full k-th best)
//   cosine:  d = 1 - S/(|q||r|) <= U  <=>  S >= (1-U)|q| * |r|
//   dot:     d = 1 - S         <= U  <=>  S >= 1-U
//   fp32 MFMA chain error: |S~ - S| <= gamma_K |q||r|, gamma_K = (K+2)u/(1-(K+2)u), u = 2^-24
//   filter:  keep r when  S~ >= c_q*s_r - m_q*|r|,  c_q = threshold above, m_q = (gamma_K + 1e-6)|q|
//
// Operands come straight from HBM/L2 into registers (no LDS): B = corpus in its tile layout
// (lanes 0-31 take rows of chunk c, lanes 32-63 the same rows of chunk c+1: any fixed
// permutation of k is a valid GEMM as long as A uses the same one), A = queries re-laid-out
// the same way by k_mfma_prep.  One wave = 64 queries x 128 rows (8 accumulator tiles).

// Qt[qb32][chunk][32 queries][4 dims] (zero padded), per-query filter constants, counters reset
__global__ void k_mfma_prep(const float* __restrict__ queries, uint32_t nq, uint32_t nq_pad, uint32_t dim, uint32_t dim4,
                            const float* __restrict__ sample_dist /*[nq][k], or [nq][parts][k] ascending partial lists*/, uint32_t parts, uint32_t k, int metric,
                            float* __restrict__ Qt, float* __restrict__ cq, float* __restrict__ mq /*[2][nq_pad]: m_q, b_q*/, float* __restrict__ eq /*[nq_pad][2]*/,
                            uint32_t* __restrict__ cand_cnt, uint32_t* __restrict__ overflow, int bf16x3, int what, uint32_t cand_cap,
                            uint32_t steps_pad /* bfloat16 layout: steps of 16 dimensions per query block, zero padded (0 = ceil(dim / 16)) */,
                            const float* __restrict__ group_min = nullptr, uint32_t n_groups = 0 /* the sample's per-group minima [nq_pad][n_groups]: the bound is selected here */) {
    // what: 1 = operand layout only, 2 = filter constants only (needs sample_dist), 3 = both
    const uint32_t q = blockIdx.x;                     // one block per (padded) query
    if (q == 0 && threadIdx.x == 0) cand_cnt[-1] = cand_cap;           // candidate slots per query, for the filter kernels (CandOut)
    const uint32_t dim4p = (dim4 + 1) & ~1u;           // chunk count padded to even: the MFMA step eats two chunks
    if (!(what & 1)) {
    } else if (bf16x3) {
        // Qbf[(qb32 * steps + s) * 2 + {hi, lo}][64 lanes] x 16 bytes: lane 32h + j holds dims 16s + 8h .. +7 of query 32 qb32 + j
        const uint32_t steps = steps_pad ? steps_pad : (dim4 + 3) / 4;
        uint4* Qbf = reinterpret_cast<uint4*>(Qt);
        if (steps_pad && q == 0) Qbf[(size_t)(nq_pad >> 5) * steps * 128 + threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);   // (64 threads) the padded kernel's KiB of zeros
        for (uint32_t i = threadIdx.x; i < steps * 2; i += blockDim.x) {
            const uint32_t st = i >> 1, h = i & 1;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; e++) { const uint32_t d = 16 * st + 8 * h + e; x[e] = (q < nq && d < dim) ? queries[(size_t)q * dim + d] : 0.f; }
            uint4 hi, lo;
            split2(x[0], x[1], hi.x, lo.x); split2(x[2], x[3], hi.y, lo.y); split2(x[4], x[5], hi.z, lo.z); split2(x[6], x[7], hi.w, lo.w);
            const size_t base = ((size_t)(q >> 5) * steps + st) * 2 * 64 + 32 * h + (q & 31);
            Qbf[base] = hi; Qbf[base + 64] = lo;
        }
    } else
    for (uint32_t c = threadIdx.x; c < dim4p; c += blockDim.x) {
        f4 x = {0.f, 0.f, 0.f, 0.f};
        if (q < nq && c < dim4) {
            const float* src = queries + (size_t)q * dim;
            uint32_t j = 4 * c;
            x.x = j < dim ? src[j] : 0.f; x.y = j + 1 < dim ? src[j + 1] : 0.f; x.z = j + 2 < dim ? src[j + 2] : 0.f; x.w = j + 3 < dim ? src[j + 3] : 0.f;
        }
        reinterpret_cast<f4*>(Qt)[((size_t)(q >> 5) * dim4p + c) * 32 + (q & 31)] = x;
    }
    double n2 = 0.0, d2 = 0.0, h2 = 0.0;                            // |q|^2, |q - qh|^2, |qh|^2 (qh = bf16(q)): the block is one wave
    if (q < nq) {
        for (uint32_t i = threadIdx.x; i < dim; i += 64) {
            const float af = queries[(size_t)q * dim + i];
            float hf = (float)(__bf16)af;
            if (__builtin_fabsf(hf) < 1.17549435e-38f) hf = 0.f;            // an operand the matrix core may flush
            const double a = af, h = hf;
            n2 = __builtin_fma(a, a, n2); d2 = __builtin_fma(a - h, a - h, d2); h2 = __builtin_fma(h, h, h2);
        }
        n2 = wave_sum_f64(n2); d2 = wave_sum_f64(d2); h2 = wave_sum_f64(h2);
    }
    if (threadIdx.x == 0 && what == 1) {
        // the sample pass reads |q| here (rounded up) and, when it runs on the one-term kernel, that kernel's error constants in eq
        const double qn1 = __builtin_sqrt(n2);
        cq[q] = q < nq ? f32_up((float)qn1) : 0.f;
        eq[2 * q] = q < nq ? f32_up((float)(filter_gamma_acc(dim) * qn1 + __builtin_sqrt(d2) * (1.0 + 1e-9))) : 0.f;
        eq[2 * q + 1] = q < nq ? f32_up((float)(__builtin_sqrt(h2) * (1.0 + 1e-9))) : 0.f;
    }
    __shared__ float s_U;
    __shared__ uint32_t s_bins[256];
    constexpr uint32_t kPrepKeys = 4096;                             // 16 KiB of group minima (the measurement mode only: keeps the block's LDS small — 8192 padded queries are 8192 blocks)
    __shared__ uint32_t s_keys[kPrepKeys];
    if ((what & 2) && group_min && q < nq) {
        // The bound from the sample, here instead of in kernels of its own (k_sample_bound / k_sample_hist: launches on the batch's
        // critical path).  One wave; the selection is a radix selection on the ordered bits of the values, 8 bits a pass, 256 bins
        // in LDS, starting at the first bit in which the values differ at all (distances of one query share their exponent and leading
        // mantissa bits: from bit 31 down the first passes threw every value at ONE bin — 4096 serialized LDS atomics, 58 us).
        // (The minima-only bound: QV_MFMA_SAMPLE_GROUP_MIN=3, a measurement — see k_bf16x1_filter_w8's sample mode.  The default selects
        // among every row's bound: k_sample_select.)
        const uint32_t lane = threadIdx.x;
        auto kth_smallest = [&](uint32_t n, uint32_t kk) -> uint32_t {      // over s_keys[0 .. n); kk >= 1; n >= 1
            uint32_t lo = 0xFFFFFFFFu, hi = 0u;
            for (uint32_t i = lane; i < n; i += 64) { const uint32_t x = s_keys[i]; lo = x < lo ? x : lo; hi = x > hi ? x : hi; }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { const uint32_t a = __shfl_xor(lo, m), b = __shfl_xor(hi, m); lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
            if (kk > n) return 0xFFFFFFFFu;
            if (lo == hi) return lo;
            const int top = 31 - __builtin_clz(lo ^ hi);                  // highest differing bit
            uint32_t prefix = top >= 31 ? 0u : (lo >> (top + 1)) << (top + 1);
            uint32_t pmask = top >= 31 ? 0u : ~((1u << (top + 1)) - 1u); // bits decided so far
            uint32_t krem = kk;
            for (int shift = top - 7; ; shift -= 8) {
                const int sh = shift < 0 ? 0 : shift;
                const uint32_t dmask = shift < 0 ? ((1u << (shift + 8)) - 1u) : 255u;
#pragma unroll
                for (int u = 0; u < 4; u++) s_bins[4 * lane + u] = 0;
                __syncthreads();
                for (uint32_t i = lane; i < n; i += 64) {
                    const uint32_t x = s_keys[i];
                    if ((x & pmask) == prefix) atomicAdd(&s_bins[(x >> sh) & dmask], 1u);
                }
                __syncthreads();
                uint32_t b4[4], sum = 0;
#pragma unroll
                for (int u = 0; u < 4; u++) { b4[u] = s_bins[4 * lane + u]; sum += b4[u]; }
                uint32_t inc = sum;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(inc, off); if ((int)lane >= off) inc += y; }
                const uint32_t excl = inc - sum;
                const bool mine = excl < krem && krem <= inc;           // exactly one lane
                uint32_t d = 4 * lane, cum = excl;
#pragma unroll
                for (int u = 0; u < 4; u++) { if (krem > cum + b4[u] && u < 3) { cum += b4[u]; d++; } else break; }
                const uint64_t who = __ballot(mine);
                const uint32_t src = who ? (uint32_t)__builtin_ctzll(who) : 0u;
                d = __builtin_amdgcn_readlane(d, (int)src); cum = __builtin_amdgcn_readlane(cum, (int)src);
                prefix |= d << sh; pmask |= dmask << sh; krem -= cum;
                __syncthreads();
                if (shift <= 0) break;
            }
            return prefix;
        };
        const float* gm = group_min + (size_t)q * n_groups;
        const uint32_t ng = n_groups < kPrepKeys ? n_groups : kPrepKeys;   // (the launcher keeps n_groups within kPrepKeys)
        for (uint32_t base = 0; base < ng; base += 64 * 16) {              // sixteen requests in flight per lane
            float x[16];
#pragma unroll
            for (int u = 0; u < 16; u++) { const uint32_t i = base + 64 * u + lane; x[u] = i < ng ? gm[i] : 0.f; }
#pragma unroll
            for (int u = 0; u < 16; u++) { const uint32_t i = base + 64 * u + lane; if (i < ng) s_keys[i] = ord_f32(x[u]); }
        }
        __syncthreads();
        const uint32_t ukey = kth_smallest(ng, k);
        if (lane == 0) {
            float x = ukey == 0xFFFFFFFFu ? __builtin_inff() : unord_f32(ukey);   // the k-th smallest bound itself (+inf: fewer than k live sample rows)
            const float gref = (float)filter_gamma(dim, 0) * 1.000001f;
            if (x == x && x < __builtin_inff()) x = metric == QV_L2 ? __builtin_sqrtf(x) * 1.000002f : (metric == QV_L2SQ ? x * (1.0f + gref + 4e-6f) : x);   // sample_bound_finish
            else x = __builtin_inff();
            s_U = x;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && (what & 2)) {
        float c_ = __uint_as_float(0x7F800000u), m_ = 0.f, b_ = 0.f;       // padded queries: +inf threshold, nothing passes
        float ea_ = 0.f, eb_ = 0.f;
        if (q < nq) {
            const double qn = __builtin_sqrt(n2);
            // |S~ - S| <= ea |r| + eb |r - rh|: per query what the filter's scores can be off by (one-term filter: the query's own
            // rounding loss and |qh|; the others: gamma |q|, nothing per row)
            const double ea = bf16x3 == 2 ? filter_gamma_acc(dim) * qn + __builtin_sqrt(d2) * (1.0 + 1e-9) : filter_gamma(dim, bf16x3) * qn;
            const double eb = bf16x3 == 2 ? __builtin_sqrt(h2) * (1.0 + 1e-9) : 0.0;
            ea_ = f32_up((float)ea); eb_ = f32_up((float)eb);
            double U;                                                          // +inf if the sample held < k live rows
            if (group_min) U = (double)s_U;                                    // selected by the whole wave below
            else if (parts <= 1) U = (double)sample_dist[(size_t)q * k + (k - 1)];
            else {                                                             // k-th smallest over the parts' ascending lists (k_sample_bound)
                const float* sp = sample_dist + (size_t)q * parts * k;
                uint32_t at[4] = {0, 0, 0, 0};
                float best = 0.f;
                for (uint32_t t = 0; t < k; t++) {
                    best = __builtin_inff(); uint32_t bp = 0;
                    for (uint32_t pp = 0; pp < parts; pp++) {
                        const float val = at[pp] < k ? sp[pp * k + at[pp]] : __builtin_inff();
                        if (val < best) { best = val; bp = pp; }
                    }
                    at[bp]++;
                }
                U = (double)best;
            }
            const double gref = filter_gamma(dim, 0);                          // the REFERENCE's float32 accumulation error (QV_L2SQ), not the filter's
            double c, m;
            if (metric == QV_L2 || metric == QV_L2SQ) {
                // squared domain: real d^2 = |q|^2 + |r|^2 - 2S.  The reference value D relates to the real d by
                // D = d(1+eta), |eta| <= 1.3e-7 (QV_L2: float32 differences, float64 sum, sqrt, one rounding) or
                // D = d^2(1+eta), |eta| <= (K+2)u (QV_L2SQ: float32 accumulation), so D <= U implies d^2 <= T:
                const double T = metric == QV_L2 ? U * U * (1.0 + 4e-7) : U * (1.0 + gref + 2e-6);
                c = n2 * (1.0 - 2e-6) - T;                           // A_q; test: 2S~ >= A_q + (1-2e-6)|r|^2 - B_q|r| - 2 eb |r - rh|
                m = 2.0 * (ea + 1e-6 * qn);                          // B_q
                b_ = f32_up((float)(2.0 * eb));
                if (!(U == U) || U > 1.0e18 || qn < (double)filter_tiny_norm(dim) || !(qn < 1.0e18)) { c = -3.0e38; m = 0.0; b_ = 0.f; }   // (a query norm >= 1e18 or NaN: its float32 scores may overflow)
            } else {
                c = metric == QV_COSINE ? (1.0 - U - 4e-7) * qn : (1.0 - U - 4e-7 * (1.0 + __builtin_fabs(U)));
                m = ea + 1e-6 * qn;
                b_ = f32_up((float)eb);
                if (!(U == U) || U > 3.0e38 || qn < (double)filter_tiny_norm(dim) || !(qn < 1.0e18)) { c = -3.0e38; m = 0.0; b_ = 0.f; }   // no bound: everything is a candidate (overflow -> exact path); bf16 operands below 2^-126 flush
            }
            c_ = f32_down((float)c);                                         // round towards "keep more"
            m_ = f32_up((float)m);
        }
        cq[q] = c_; mq[q] = m_; mq[nq_pad + q] = b_;
        eq[2 * q] = ea_; eq[2 * q + 1] = eb_;
        if (q < nq) { cand_cnt[q] = 0; overflow[q] = 0; }
    }
}


// grid: persistent waves; wave g -> query 64-block (g % nqb64), row groups (g / nqb64) + i*stride; a row group = NJ / 2 tiles.
// NJ = 4: 128 rows per group, 8 accumulator tiles, ONE wave per SIMD (rounds 1-5).  NJ = 2: 64 rows, 4 tiles, 128 accumulator
// registers less — TWO waves per SIMD fit (two workgroups per CU), and one wave's epilogue (6 % of the kernel: 3.10 against 2.91 ms
// with the epilogue compiled out, -DQV_MFMA_F32_NOEPI) runs under the other's matrix instructions; the price is the query operand
// read once per 64 rows instead of once per 128 by twice the waves: 16 bytes per cycle and CU from L2, ~10 TB/s over the chip —
// measured 4.25 against 3.06 ms, same results.  Round 6 built it, measured it and ships NJ = 4 alone (NJ = 2: measurement build).
template <int METRIC, int NJ>
__global__ void __launch_bounds__(256, NJ == 2 ? 2 : 1)
k_mfma_filter(IndexView v, const float* __restrict__ Qt, const float* __restrict__ cq, const float* __restrict__ mq, uint32_t nq_pad,
              uint32_t* __restrict__ cand_rows, float* __restrict__ cand_score, uint32_t* __restrict__ cand_cnt) {
    __shared__ __align__(16) float s_c[4][64], s_m[8][64];                        // this wave's 64 queries' filter constants
    QV_CAND_QUEUE(cqu, 4, 512);                                       // 24 KiB
    QV_EPI_DUMP(du, 4, 64);                                           // 20 KiB
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nqb64 = nq_pad >> 6;
    const uint32_t gw = filter_logical_block((nqb64 & 3u) == 0 ? nqb64 >> 2 : 1u) * 4 + wave, tw = gridDim.x * 4;
    const uint32_t qb64 = gw % nqb64;
    const uint32_t n_groups = NJ == 4 ? (v.n_tiles + 1) / 2 : v.n_tiles;
    const uint32_t stride = tw / nqb64;
    {   // cosine: one constant t_q = c_q - m_q (test S~ >= t_q |r|); dot: c_q and m_q (test S~ >= c_q - m_q |r|)
        const float c = cq[64 * qb64 + lane], m = mq[64 * qb64 + lane];
        s_c[wave][lane] = METRIC == QV_COSINE ? c - m : c;     // L2 family: A_q (c) and B_q (m)
        s_m[wave][lane] = m;
        s_m[4 + wave][lane] = mq[nq_pad + 64 * qb64 + lane];
    }
    __syncthreads();
    if (stride == 0) return;
    const uint32_t half = lane >> 5, l31 = lane & 31;
    const EpiConsts ec = epi_consts<METRIC>(&s_c[0][0], &s_m[0][0], 64 * wave, half);
    const f4* tiles = reinterpret_cast<const f4*>(v.tiles);
    const f4* qt = reinterpret_cast<const f4*>(Qt);
    const uint32_t steps = (v.dim4 + 1) / 2;                       // 8 dims per step
    const uint32_t dim4p = 2 * steps;                              // Qt is zero-padded to an even chunk count
    const f4* a_base0 = qt + ((size_t)(2 * qb64) * dim4p) * 32 + l31;
    const f4* a_base1 = qt + ((size_t)(2 * qb64 + 1) * dim4p) * 32 + l31;

    for (uint32_t g = gw / nqb64; g < n_groups; g += stride) {
        const uint32_t t0 = NJ == 4 ? 2 * g : g, t1 = NJ == 4 ? ((2 * g + 1 < v.n_tiles) ? 2 * g + 1 : t0) : t0;     // odd tail: tile duplicated, masked below
        const f4* b0 = tiles + (size_t)t0 * v.dim4 * 64 + l31;
        const f4* b1 = tiles + (size_t)t1 * v.dim4 * 64 + l31;
        f16v acc[2][NJ];
        double rnd[NJ]; float rho[NJ]; uint64_t alv[NJ / 2];
        filter_row_consts<NJ>(v, t0, t1, l31, rnd, rho, alv);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

        // branch-free operand fetch: a step index past the end re-reads the last step (never used).
        // For an odd dim4 the upper lane half of the last step reads Qt's zero padding on the A side
        // and re-reads the last real chunk on the B side (0 * finite = 0).
        auto load = [&](uint32_t st, f4 (&A)[2], f4 (&B)[NJ]) {
            const uint32_t sc = st < steps ? st : steps - 1;
            const uint32_t ca = 2 * sc + half;
            const uint32_t cb = ca < v.dim4 ? ca : v.dim4 - 1;
            A[0] = a_base0[(size_t)ca * 32];
            A[1] = a_base1[(size_t)ca * 32];
            B[0] = __builtin_nontemporal_load(&b0[(size_t)cb * 64]);
            B[1] = __builtin_nontemporal_load(&b0[(size_t)cb * 64 + 32]);
            if constexpr (NJ == 4) {
                B[2] = __builtin_nontemporal_load(&b1[(size_t)cb * 64]);
                B[3] = __builtin_nontemporal_load(&b1[(size_t)cb * 64 + 32]);
            }
        };
        auto mma = [&](const f4 (&A)[2], const f4 (&B)[NJ]) {
#pragma unroll
            for (int d = 0; d < 4; d++)
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < NJ; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i][d], B[j][d], acc[i][j], 0, 0, 0);
        };
        // 3-deep software pipeline over the K steps (operands for steps s+1, s+2 in flight while s computes)
        f4 A0[2], B0[NJ], A1[2], B1[NJ], A2[2], B2[NJ];
        load(0, A0, B0);
        load(1, A1, B1);
        uint32_t st = 0;
        for (; st + 3 <= steps; st += 3) {                          // sched_barrier: keep the issue order as written
            load(st + 2, A2, B2); __builtin_amdgcn_sched_barrier(0);   // (hipcc otherwise sinks the loads next to their
            mma(A0, B0);          __builtin_amdgcn_sched_barrier(0);   //  first use and waits vmcnt(0) mid-loop)
            load(st + 3, A0, B0); __builtin_amdgcn_sched_barrier(0);
            mma(A1, B1);          __builtin_amdgcn_sched_barrier(0);
            load(st + 4, A1, B1); __builtin_amdgcn_sched_barrier(0);
            mma(A2, B2);          __builtin_amdgcn_sched_barrier(0);
        }
        if (st < steps) { mma(A0, B0); st++; }
        if (st < steps) { mma(A1, B1); st++; }

#ifdef QV_MFMA_F32_NOEPI
        // TIMING ONLY (tools/build_variant.sh): what the epilogue costs — one value of every accumulator tile kept alive, no test, no candidates
        { float keep = 0.f;
#pragma unroll
          for (int i = 0; i < 2; i++)
#pragma unroll
              for (int j = 0; j < NJ; j++) keep += acc[i][j][0];
          if (keep == 1.2345e-30f) cand_cnt[0] = 1; }
#else
        filter_epilogue<METRIC>(v, acc, t0, t1, &s_c[0][0], &s_m[0][0], 64 * wave, half, l31, 64 * qb64, filter_tiny_norm(v.dim), ec, rnd, rho, alv, cqu, cqu_n, cqu_out, du);
#endif
    }
    cand_flush(cqu, cqu_n, cqu_out);
}

// An UPPER bound of the reference distance from a score, in float32 (the sample pass: 32 768 rows per query): score_interval's
// formulas with every float32 rounding covered by an explicit allowance — any upper bound of the k-th smallest distance serves as
// U_q, a looser one only lets a few more rows through.  128 of these per lane and row group, so everything that depends on the
// query alone or the row alone is prepared once (sample_query_consts / sample_row_consts) and the per-score part is 3-6 instructions
// without a division.  g = the filter's gamma (|S~ - S| <= g |q||r|); qn, rn rounded UP from float64.  A query or row the scores
// say nothing about (norm below filter_tiny_norm: flushed operands or products; 1e18 or above: float32 sums may overflow; a row that is gone) carries
// NaN constants: its bounds come out NaN and are stored as +inf.  With both norms inside that range no score overflows.
// The L2 family yields the bound on d^2 (one kernel serves QV_L2 and QV_L2SQ); sample_bound_finish takes the SELECTED value into
// the metric's units — both steps are monotone, so the k-th smallest commutes with them.
template <int M>
__device__ __forceinline__ void sample_query_consts(float qn, float g, float tiny, float& qa, float& qb) {
    const float nan = __builtin_nanf("");
    const bool none = (qn != 0.f && qn < tiny) || !(qn < 1.0e18f);
    if constexpr (M == QV_COSINE) { qa = none ? nan : (qn == 0.f ? 0.f : 1.0f / qn); qb = 0.f; }        // zero norm: the bound becomes 1 + g (the distance is 1)
    else if constexpr (M == QV_DOT) { qa = none ? nan : g * qn * 1.000001f; qb = 0.f; }
    else { qa = none ? nan : qn * qn * 1.000001f; qb = 2.0f * g * qn * 1.000002f; }
}
template <int M>
__device__ __forceinline__ void sample_row_consts(float rn, bool gone, float tiny, float& ra, float& rb) {
    const float nan = __builtin_nanf("");
    const bool none = gone || (rn != 0.f && rn < tiny) || !(rn < 1.0e18f);
    if constexpr (M == QV_COSINE) { ra = none ? nan : (rn == 0.f ? 0.f : 1.0f / rn); rb = 0.f; }
    else if constexpr (M == QV_DOT) { ra = none ? nan : rn; rb = 0.f; }
    else { ra = none ? nan : rn * rn * 1.000001f; rb = rn; }
}
template <int M>
__device__ __forceinline__ float sample_upper(float S, float g, float qa, float qb, float ra, float rb) {
    float hi;
    if constexpr (M == QV_COSINE) hi = (1.0f - S * (qa * ra)) + (g + 5e-6f);          // |S|/(|q||r|) <= 1 + g: two reciprocals, two products, a difference, a sum: < 1e-6
    else if constexpr (M == QV_DOT) { const float d = 1.0f - S; hi = d + (qa * ra + 4e-6f * (1.0f + __builtin_fabsf(d))); }
    else {
        const float sum = qa + ra;                                   // >= |q|^2 + |r|^2
        const float h2 = (sum - 2.0f * S) + (qb * rb + 5e-6f * sum);  // 2|S| <= sum: every rounding is of magnitude <= 2 sum
        hi = h2 > 0.f ? h2 : (h2 == h2 ? 0.f : h2);
    }
    return hi == hi ? hi : __builtin_inff();
}
// The same for the ONE-TERM filter's scores (round 3: the sample pass of the one-term path runs on the eight-wave one-term kernel,
// a third of the three-term kernel's time; its bounds are looser by the difference of the two error bounds, ~0.07 sigma of the
// scores of 768-d data, i.e. ~1.3 x the candidates): |S~ - S| <= ea |r| + eb |r - rh| with ea, eb per query (k_mfma_prep), so a
// query and a row contribute three constants each.
template <int M>
__device__ __forceinline__ void sample1_query_consts(float qn, float ea, float eb, float tiny, float& qa, float& qb, float& qc) {
    const float nan = __builtin_nanf("");
    const bool none = (qn != 0.f && qn < tiny) || !(qn < 1.0e18f);
    if constexpr (M == QV_COSINE) {
        const float inv = qn == 0.f ? 0.f : 1.0f / qn;
        qa = none ? nan : inv; qb = ea * inv * 1.000001f; qc = eb * inv * 1.000001f;
    } else if constexpr (M == QV_DOT) { qa = none ? nan : ea * 1.000001f; qb = eb * 1.000001f; qc = 0.f; }
    else { qa = none ? nan : qn * qn * 1.000001f; qb = 2.0f * ea * 1.000002f; qc = 2.0f * eb * 1.000002f; }
}
template <int M>
__device__ __forceinline__ void sample1_row_consts(float rn, float rho, bool gone, float tiny, float& ra, float& rb, float& rc) {
    const float nan = __builtin_nanf("");
    const bool none = gone || (rn != 0.f && rn < tiny) || !(rn < 1.0e18f);
    if constexpr (M == QV_COSINE) {
        const float inv = rn == 0.f ? 0.f : 1.0f / rn;
        ra = none ? nan : inv; rb = rho * inv * 1.000001f; rc = 0.f;
    } else if constexpr (M == QV_DOT) { ra = none ? nan : rn; rb = rho; rc = 0.f; }
    else { ra = none ? nan : rn * rn * 1.000001f; rb = rn; rc = rho; }
}
template <int M>
__device__ __forceinline__ float sample1_upper(float S, float qa, float qb, float qc, float ra, float rb, float rc) {
    float hi;
    if constexpr (M == QV_COSINE) hi = (1.0f - S * (qa * ra)) + ((qb + qc * rb) + 5e-6f);      // E / (|q||r|) = ea/|q| + (eb/|q|)(|r - rh|/|r|)
    else if constexpr (M == QV_DOT) { const float d = 1.0f - S; hi = d + ((qa * ra + qb * rb) + 4e-6f * (1.0f + __builtin_fabsf(d))); }
    else {
        const float sum = qa + ra;
        const float h2 = (sum - 2.0f * S) + ((qb * rb + qc * rc) + 5e-6f * sum);
        hi = h2 > 0.f ? h2 : (h2 == h2 ? 0.f : h2);
    }
    return hi == hi ? hi : __builtin_inff();
}
template <int M>
__device__ __forceinline__ float sample_bound_finish(float x, float gref) {
    if constexpr (M == QV_L2) return __builtin_sqrtf(x) * 1.000002f;
    else if constexpr (M == QV_L2SQ) return x * (1.0f + gref + 4e-6f);
    else return x;
}

// The same filter on the bfloat16 matrix instruction, three products per pair of operands (see filter_gamma): the scores keep
// float32-class accuracy (margin 3.3e-4 |q||r| at 768 dims against 4.7e-5 for the fp32 chain), the matrix work drops from four
// 64-cycle v_mfma_f32_32x32x2_f32 per dimension to 1.5 32-cycle v_mfma_f32_32x32x16_bf16, and the kernel becomes a reader of
// the corpus: 128 rows x 3 KiB per 37 k matrix cycles per workgroup.  Rows are split on the fly (v_cvt_pk_bf16_f32, a shift,
// a subtract, a second convert per pair of floats); queries are split once by k_mfma_prep.
// Operand mapping: lane 32h + j supplies dims 16s + 8h .. +7 of query / row j of its 32-block for BOTH operands, so whatever
// order the instruction walks k in, A and B agree on it.
template <int METRIC>
__global__ void __launch_bounds__(256, 1)
k_bf16x3_filter(IndexView v, const uint4* __restrict__ Qbf, const float* __restrict__ cq, const float* __restrict__ mq, uint32_t nq_pad,
                uint32_t* __restrict__ cand_rows, float* __restrict__ cand_score, uint32_t* __restrict__ cand_cnt,
                float* __restrict__ score_out, uint32_t score_stride, uint32_t gstep) {
    // score_out != null: no filter — the SAMPLE pass.  Row groups 0, gstep, 2 gstep, ... (score_stride / 128 of them, spread over the
    // corpus so that a corpus stored cluster by cluster still yields a representative bound); what is written to
    // score_out[query * score_stride + 128 * (group's place in the sample) + row of the group] is the UPPER BOUND of the row's
    // reference distance that its score implies (score_upper_f32; +inf for rows that are gone or whose score says nothing), so that
    // k_sample_bound only selects: the row norms and alive words are read once here instead of once per query there.
    // cq = the queries' norms, rounded up (k_mfma_prep, what = 1).
    __shared__ __align__(16) float s_c[4][64], s_m[8][64];
    QV_CAND_QUEUE(cqu, 4, 512);                                       // 24 KiB
    QV_EPI_DUMP(du, 4, 64);                                           // 20 KiB
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nqb64 = nq_pad >> 6;
    const uint32_t gw = filter_logical_block((nqb64 & 3u) == 0 ? nqb64 >> 2 : 1u) * 4 + wave, tw = gridDim.x * 4;
    const uint32_t qb64 = gw % nqb64;
    const uint32_t n_groups = score_out ? (score_stride + 127) / 128 : (v.n_tiles + 1) / 2;
    const uint32_t stride = tw / nqb64;
    if (!score_out) {
        const float c = cq[64 * qb64 + lane], m = mq[64 * qb64 + lane];
        s_c[wave][lane] = METRIC == QV_COSINE ? c - m : c;
        s_m[wave][lane] = m;
        s_m[4 + wave][lane] = mq[nq_pad + 64 * qb64 + lane];
    } else sample_query_consts<METRIC>(cq[64 * qb64 + lane], f32_up((float)filter_gamma(v.dim, 1)), filter_tiny_norm(v.dim), s_c[wave][lane], s_m[wave][lane]);
    __syncthreads();
    if (stride == 0) return;
    const uint32_t half = lane >> 5, l31 = lane & 31;
    const EpiConsts ec = epi_consts<METRIC>(&s_c[0][0], &s_m[0][0], 64 * wave, half);
    const f4* tiles = reinterpret_cast<const f4*>(v.tiles);
    const uint32_t steps = (v.dim4 + 3) / 4;                        // 16 dims per step
    const uint4* a0 = Qbf + ((size_t)(2 * qb64) * steps) * 2 * 64 + lane;
    const uint4* a1 = Qbf + ((size_t)(2 * qb64 + 1) * steps) * 2 * 64 + lane;

    for (uint32_t g = gw / nqb64; g < n_groups; g += stride) {
        const uint32_t ga = score_out ? g * gstep : g;             // the group's place in the corpus
        const uint32_t t0 = 2 * ga, t1 = (2 * ga + 1 < v.n_tiles) ? 2 * ga + 1 : t0;
        const f4* b0 = tiles + (size_t)t0 * v.dim4 * 64 + l31;
        const f4* b1 = tiles + (size_t)t1 * v.dim4 * 64 + l31;
        f16v acc[2][4];
        double rnd[4]; float rho[4]; uint64_t alv[2];
        filter_row_consts(v, t0, t1, l31, rnd, rho, alv);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

        struct Ops { uint4 ah[2], al[2]; f4 b[4][2]; };           // a step's operands as loaded: A split by k_mfma_prep, B raw float32
        struct Hl { uint4 ah[2], al[2], bh[4], bl[4]; };          // the same step ready for the matrix core
        // branch-free fetch: a step past the end re-reads the last one (never used); chunks past dim4 re-read the last real chunk
        // on the B side against zeros on the A side
        auto load = [&](uint32_t st, Ops& o) {
            const uint32_t sc = st < steps ? st : steps - 1;
            o.ah[0] = a0[(size_t)sc * 128]; o.al[0] = a0[(size_t)sc * 128 + 64];
            o.ah[1] = a1[(size_t)sc * 128]; o.al[1] = a1[(size_t)sc * 128 + 64];
            const uint32_t c0 = 4 * sc + 2 * half;
            const uint32_t ca = c0 < v.dim4 ? c0 : v.dim4 - 1, cb = c0 + 1 < v.dim4 ? c0 + 1 : v.dim4 - 1;
            o.b[0][0] = __builtin_nontemporal_load(&b0[(size_t)ca * 64]);      o.b[0][1] = __builtin_nontemporal_load(&b0[(size_t)cb * 64]);
            o.b[1][0] = __builtin_nontemporal_load(&b0[(size_t)ca * 64 + 32]); o.b[1][1] = __builtin_nontemporal_load(&b0[(size_t)cb * 64 + 32]);
            o.b[2][0] = __builtin_nontemporal_load(&b1[(size_t)ca * 64]);      o.b[2][1] = __builtin_nontemporal_load(&b1[(size_t)cb * 64]);
            o.b[3][0] = __builtin_nontemporal_load(&b1[(size_t)ca * 64 + 32]); o.b[3][1] = __builtin_nontemporal_load(&b1[(size_t)cb * 64 + 32]);
        };
        auto split = [&](const Ops& o, Hl& h) {                     // ~110 VALU instructions
            h.ah[0] = o.ah[0]; h.al[0] = o.al[0]; h.ah[1] = o.ah[1]; h.al[1] = o.al[1];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                split2(o.b[j][0].x, o.b[j][0].y, h.bh[j].x, h.bl[j].x); split2(o.b[j][0].z, o.b[j][0].w, h.bh[j].y, h.bl[j].y);
                split2(o.b[j][1].x, o.b[j][1].y, h.bh[j].z, h.bl[j].z); split2(o.b[j][1].z, o.b[j][1].w, h.bh[j].w, h.bl[j].w);
            }
        };
        auto mfma = [&](const Hl& h) {                              // 24 matrix instructions, 32 cycles each
            const bf8 ah0 = __builtin_bit_cast(bf8, h.ah[0]), al0 = __builtin_bit_cast(bf8, h.al[0]);
            const bf8 ah1 = __builtin_bit_cast(bf8, h.ah[1]), al1 = __builtin_bit_cast(bf8, h.al[1]);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bf8 bh = __builtin_bit_cast(bf8, h.bh[j]), bl = __builtin_bit_cast(bf8, h.bl[j]);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl, acc[0][j], 0, 0, 0);      // small terms first
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl, acc[1][j], 0, 0, 0);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh, acc[1][j], 0, 0, 0);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh, acc[1][j], 0, 0, 0);
            }
        };
        // One step of the pipeline: the loads of step s+2 are requested, then the matrix instructions of step s are issued
        // INTERLEAVED with the vector instructions that split step s+1 (one MFMA, five VALU, ...): a single wave per SIMD has
        // nobody else to fill the 32 cycles a matrix instruction occupies the pipe.  (Split first, then 24 MFMAs back to back:
        // matrix pipe 30 % busy, SQ_VALU_MFMA_COEXEC_CYCLES 6 % — profiles/r02_bf16x3_filter.txt.)
        auto step = [&](uint32_t s_load, Ops& o_load, const Ops& o_split, Hl& h_split, const Hl& h_mfma) {
            load(s_load, o_load);
            __builtin_amdgcn_sched_barrier(0);
            mfma(h_mfma);
            split(o_split, h_split);
#pragma unroll
            for (int n = 0; n < 24; n++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   // five VALU
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        Ops o0, o1, o2;
        Hl h0, h1;
        uint32_t st = 0;
        if (steps >= 6) {
            load(0, o0); load(1, o1);
            split(o0, h0);
            __builtin_amdgcn_sched_barrier(0);
            for (; st + 6 <= steps; st += 6) {      // loads run two steps ahead (three raw buffers), the split one step ahead (two)
                step(st + 2, o2, o1, h1, h0);
                step(st + 3, o0, o2, h0, h1);
                step(st + 4, o1, o0, h1, h0);
                step(st + 5, o2, o1, h0, h1);
                step(st + 6, o0, o2, h1, h0);
                step(st + 7, o1, o0, h0, h1);
            }
        }
        for (; st < steps; st++) {                   // the steps that do not fill a round of six: one at a time
            load(st, o0); split(o0, h0); mfma(h0);
        }
        if (score_out) {
            const float gsm = f32_up((float)filter_gamma(v.dim, 1));
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (j >= 2 && t1 == t0) continue;
                const uint32_t row = g * 128 + (j < 2 ? 0u : 64u) + 32 * (j & 1) + l31;      // place in the sample
                if (row >= score_stride) continue;
                float ra, rb;
                sample_row_consts<METRIC>(f32_up((float)rnd[j]), !((alv[j >> 1] >> (32 * (j & 1) + l31)) & 1ull), filter_tiny_norm(v.dim), ra, rb);
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const uint32_t ql = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                        score_out[(size_t)(64 * qb64 + ql) * score_stride + row] = sample_upper<METRIC>(acc[i][j][r], gsm, s_c[wave][ql], s_m[wave][ql], ra, rb);
                    }
            }
            continue;
        }
        filter_epilogue<METRIC>(v, acc, t0, t1, &s_c[0][0], &s_m[0][0], 64 * wave, half, l31, 64 * qb64, filter_tiny_norm(v.dim), ec, rnd, rho, alv, cqu, cqu_n, cqu_out, du);
    }
    cand_flush(cqu, cqu_n, cqu_out);
}

// The same with the row operand shared by the four waves of a workgroup (nq_pad a multiple of 256: the waves of a workgroup then
// hold four different query blocks and walk the SAME row groups).  k_bf16x3_filter asks the CU's vector L1 for 2.4 MB per row
// group — 64 B per clock at the matrix rate, all the L1 can deliver — because every wave fetches and splits all 128 rows itself.
// Here wave w fetches and splits only rows 32w .. 32w+31 of the group and publishes the two bfloat16 planes in LDS (8 KiB per
// step, three stages, one barrier per step); all four read their B operands from there: half the L1 traffic, a quarter of the
// vector instructions and of the row loads per wave.
template <int METRIC, int TERMS, int RING>
__global__ void __launch_bounds__(256, 1)
k_bf16x3_filter_shared(IndexView v, const uint4* __restrict__ Qbf, const float* __restrict__ cq, const float* __restrict__ mq, uint32_t nq_pad,
                       uint32_t* __restrict__ cand_rows, float* __restrict__ cand_score, uint32_t* __restrict__ cand_cnt) {
    __shared__ __align__(16) float s_c[4][64], s_m[8][64];
    QV_CAND_QUEUE(cqu, 4, 512);                                       // 24 KiB
    QV_EPI_DUMP(du, 4, 64);                                           // 20 KiB
    __shared__ uint4 s_b[4][4][2][64];                              // [stage][32-row block][hi, lo][lane]: 32 KiB
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nqb64 = nq_pad >> 6;                             // a multiple of 4
    const uint32_t wgs_per_group = nqb64 >> 2;                      // workgroups that share one row group (different query blocks)
    uint32_t qblk_, walk0_; filter_block_role(wgs_per_group, qblk_, walk0_);
    const uint32_t qb64 = qblk_ * 4 + wave;
    const uint32_t n_groups = (v.n_tiles + 1) / 2;
    const uint32_t stride = gridDim.x / wgs_per_group;
    {
        const float c = cq[64 * qb64 + lane], m = mq[64 * qb64 + lane];
        s_c[wave][lane] = METRIC == QV_COSINE ? c - m : c;
        s_m[wave][lane] = m;
        s_m[4 + wave][lane] = mq[nq_pad + 64 * qb64 + lane];
    }
    __syncthreads();
    if (stride == 0) return;
    const uint32_t half = lane >> 5, l31 = lane & 31;
    const EpiConsts ec = epi_consts<METRIC>(&s_c[0][0], &s_m[0][0], 64 * wave, half);
    const f4* tiles = reinterpret_cast<const f4*>(v.tiles);
    const uint32_t steps = (v.dim4 + 3) / 4;                        // 16 dims per step
    const uint4* a0 = Qbf + ((size_t)(2 * qb64) * steps) * 2 * 64 + lane;
    const uint4* a1 = Qbf + ((size_t)(2 * qb64 + 1) * steps) * 2 * 64 + lane;

    struct Raw { f4 b[2]; };
    struct Aop { uint4 ah[2], al[2]; };
    struct Bset { uint4 h[4], l[4]; };
    // the rings of the eight-step form live across row groups: the requests that run past the end of a group are the first
    // ones of the workgroup's next group, so a group starts with its operands on the way or already published
    Raw r[RING];                                                    // rows requested RING + 2 steps ahead of their use
    constexpr int AR = 4;                                           // query operands are requested AR - 1 steps ahead (8 and a 16-deep row ring
    Aop q[AR];                                                      // change nothing: 1.10 / 1.15 ms against 1.07 ms for the one-term filter, 256 x 1M x 768)
    Bset b0, b1;
    bool primed = false;
    const f4* lp = nullptr;                                         // running pointers of the eight-step form: the next row chunk pair
    const uint4* ap0 = a0; const uint4* ap1 = a1;                   // and the next query operands to request
    const bool deep = steps % RING == 0 && steps >= 2 * RING;
    auto rows_of = [&](uint32_t g_) {                                // this wave's quarter of group g_: rows 32*(wave&1) .. +31 of one of its tiles
        const uint32_t ta = 2 * g_, tb = (2 * g_ + 1 < v.n_tiles) ? 2 * g_ + 1 : ta;
        return tiles + (size_t)(wave < 2 ? ta : tb) * v.dim4 * 64 + 32 * (wave & 1) + l31;
    };

    for (uint32_t g = walk0_; g < n_groups; g += stride) {
        const uint32_t t0 = 2 * g, t1 = (2 * g + 1 < v.n_tiles) ? 2 * g + 1 : t0;
        const f4* bw = rows_of(g);
        const f4* bwn = rows_of(g + stride < n_groups ? g + stride : g);
        f16v acc[2][4];
        double rnd[4]; float rho[4]; uint64_t alv[2];
        filter_row_consts(v, t0, t1, l31, rnd, rho, alv);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

        auto load_b = [&](uint32_t st, Raw& o) {
            // deep form: a step past the end of this group is a step of the next one; plain form: re-reads the last step (never used)
            const bool nx = deep && st >= steps;
            const f4* base = nx ? bwn : bw;
            const uint32_t sc = nx ? st - steps : (st < steps ? st : steps - 1);
            const uint32_t c0 = 4 * sc + 2 * half;
            const uint32_t ca = c0 < v.dim4 ? c0 : v.dim4 - 1, cb = c0 + 1 < v.dim4 ? c0 + 1 : v.dim4 - 1;
            o.b[0] = __builtin_nontemporal_load(&base[(size_t)ca * 64]); o.b[1] = __builtin_nontemporal_load(&base[(size_t)cb * 64]);
        };
        auto load_a = [&](uint32_t st, Aop& o) {
            const uint32_t sc = deep && st >= steps ? st - steps : (st < steps ? st : steps - 1);
            o.ah[0] = a0[(size_t)sc * 128]; o.ah[1] = a1[(size_t)sc * 128];
            if constexpr (TERMS == 3) { o.al[0] = a0[(size_t)sc * 128 + 64]; o.al[1] = a1[(size_t)sc * 128 + 64]; }
        };
        auto publish = [&](const Raw& o, uint32_t stage) {          // split this wave's rows and put the two planes in LDS
            uint4 h, l;
            if constexpr (TERMS == 3) {
                split2(o.b[0].x, o.b[0].y, h.x, l.x); split2(o.b[0].z, o.b[0].w, h.y, l.y);
                split2(o.b[1].x, o.b[1].y, h.z, l.z); split2(o.b[1].z, o.b[1].w, h.w, l.w);
                s_b[stage][wave][0][lane] = h; s_b[stage][wave][1][lane] = l;
            } else {
                h.x = pack_bf16(o.b[0].x, o.b[0].y); h.y = pack_bf16(o.b[0].z, o.b[0].w);
                h.z = pack_bf16(o.b[1].x, o.b[1].y); h.w = pack_bf16(o.b[1].z, o.b[1].w);
                s_b[stage][wave][0][lane] = h;
            }
        };
        auto mfma = [&](const Aop& a, uint32_t stage) {
            const bf8 ah0 = __builtin_bit_cast(bf8, a.ah[0]), ah1 = __builtin_bit_cast(bf8, a.ah[1]);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bf8 bh = __builtin_bit_cast(bf8, s_b[stage][j][0][lane]);
                if constexpr (TERMS == 3) {
                    const bf8 al0 = __builtin_bit_cast(bf8, a.al[0]), al1 = __builtin_bit_cast(bf8, a.al[1]);
                    const bf8 bl = __builtin_bit_cast(bf8, s_b[stage][j][1][lane]);
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl, acc[0][j], 0, 0, 0);      // small terms first
                    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl, acc[1][j], 0, 0, 0);
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh, acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh, acc[1][j], 0, 0, 0);
                }
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh, acc[1][j], 0, 0, 0);
            }
        };
        if (deep) {
            // Dimensions that are a multiple of 128: a round of eight steps with everything but the matrix instructions running
            // underneath them.  Step s: barrier (step s+1 is published), read the B operands of step s+1 from LDS into the
            // second register set, request the rows of step s+8 and the queries of step s+3, then the 24 matrix instructions
            // of step s with the splitting and publishing of step s+2 scheduled between them.  A lone wave per SIMD has nobody
            // to cover an exposed LDS read, barrier skew or split (the unpipelined loop below: matrix pipe 35 % busy).
            auto read_b = [&](uint32_t stage, Bset& b) {
#pragma unroll
                for (int j = 0; j < 4; j++) { b.h[j] = s_b[stage][j][0][lane]; if constexpr (TERMS == 3) b.l[j] = s_b[stage][j][1][lane]; }
            };
            auto mfma_r = [&](const Aop& a, const Bset& b) {
                const bf8 ah0 = __builtin_bit_cast(bf8, a.ah[0]), ah1 = __builtin_bit_cast(bf8, a.ah[1]);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const bf8 bh = __builtin_bit_cast(bf8, b.h[j]);
                    if constexpr (TERMS == 3) {
                        const bf8 al0 = __builtin_bit_cast(bf8, a.al[0]), al1 = __builtin_bit_cast(bf8, a.al[1]);
                        const bf8 bl = __builtin_bit_cast(bf8, b.l[j]);
                        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl, acc[0][j], 0, 0, 0);
                        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl, acc[1][j], 0, 0, 0);
                        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh, acc[0][j], 0, 0, 0);
                        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh, acc[1][j], 0, 0, 0);
                    }
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh, acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh, acc[1][j], 0, 0, 0);
                }
            };
            // requests walk running pointers (no per-step clamps or selects: dimensions here are whole steps); the row pointer jumps
            // to the workgroup's next group, the query pointers back to step 0, at fixed places of a group's last round
            auto load_b_run = [&](Raw& o) {
                o.b[0] = __builtin_nontemporal_load(lp); o.b[1] = __builtin_nontemporal_load(lp + 64);
                lp += 256;
            };
            auto load_a_run = [&](Aop& o) {
                o.ah[0] = ap0[0]; o.ah[1] = ap1[0];
                if constexpr (TERMS == 3) { o.al[0] = ap0[64]; o.al[1] = ap1[64]; }
                ap0 += 128; ap1 += 128;
            };
            if (!primed) {                                          // the workgroup's first group: fill the rings
                primed = true;
                lp = bw + 2 * half * 64;
#pragma unroll
                for (int i = 0; i < RING; i++) load_b_run(r[i]);
#pragma unroll
                for (int i = 0; i < AR - 1; i++) load_a_run(q[i]);
                publish(r[0], 0); publish(r[1], 1);
                load_b_run(r[0]); load_b_run(r[1]);
                __syncthreads();
                read_b(0, b0);
            }
            auto pstep = [&](uint32_t s_, int k8, const Bset& b_use, Bset& b_next) {
                __syncthreads();                                    // step s+1 is in LDS (published during step s-1)
                read_b((uint32_t)(k8 + 1) & 3, b_next);
                if (s_ + (AR - 1) == steps) { ap0 = a0; ap1 = a1; }
                load_a_run(q[(k8 + AR - 1) & (AR - 1)]);
                __builtin_amdgcn_sched_barrier(0);
                mfma_r(q[k8 & (AR - 1)], b_use);
                publish(r[(k8 + 2) & (RING - 1)], (uint32_t)(k8 + 2) & 3);   // rows of step s+2, requested RING steps ago
                if (s_ + RING + 2 == steps) lp = bwn + 2 * half * 64;
                load_b_run(r[(k8 + 2) & (RING - 1)]);
#pragma unroll
                for (int n = 0; n < 8 * TERMS; n++) {               // one matrix instruction, two others, ...
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002 | 0x100 | 0x200 | 0x020, 2, 0);   // VALU / DS read / DS write / VMEM read
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            for (uint32_t st = 0; st < steps; st += RING) {
#pragma unroll
                for (int k8 = 0; k8 < RING; k8 += 2) { pstep(st + k8, k8, b0, b1); pstep(st + k8 + 1, k8 + 1, b1, b0); }
            }
        } else {
        // step s: request rows of step s+2 and queries of step s+1, publish step s+1, barrier, consume step s.  A stage is
        // rewritten three steps after it was read, with a barrier in between.
        Raw r0, r1, r2;
        Aop q0, q1;
        load_b(0, r0); load_b(1, r1); load_a(0, q0);
        __syncthreads();                                            // the previous group's last stage has been read by everyone
        publish(r0, 0);
        uint32_t st = 0;
        auto step = [&](uint32_t s_, Raw& r_load, const Raw& r_pub, Aop& a_load, const Aop& a_use) {
            load_b(s_ + 2, r_load); load_a(s_ + 1, a_load);
            __builtin_amdgcn_sched_barrier(0);
            publish(r_pub, (s_ + 1) % 3);
            __syncthreads();
            mfma(a_use, s_ % 3);
            __builtin_amdgcn_sched_barrier(0);
        };
        for (; st + 6 <= steps; st += 6) {
            step(st,     r2, r1, q1, q0);
            step(st + 1, r0, r2, q0, q1);
            step(st + 2, r1, r0, q1, q0);
            step(st + 3, r2, r1, q0, q1);
            step(st + 4, r0, r2, q1, q0);
            step(st + 5, r1, r0, q0, q1);
        }
        for (; st < steps; st++) {                                  // the steps that do not fill a round of six: unpipelined
            __syncthreads();
            load_b(st, r0); load_a(st, q0);
            publish(r0, 0);
            __syncthreads();
            mfma(q0, 0);
        }
        }
        filter_epilogue<METRIC>(v, acc, t0, t1, &s_c[0][0], &s_m[0][0], 64 * wave, half, l31, 64 * qb64, filter_tiny_norm(v.dim), ec, rnd, rho, alv, cqu, cqu_n, cqu_out, du);
    }
    cand_flush(cqu, cqu_n, cqu_out);
}

// The one-term filter over float32 rows with EIGHT waves per workgroup (round 3; the default for 256 queries and more).
// k_bf16x3_filter_shared gives each of four waves a 64-query x 128-row tile: 128 accumulators, ~450 registers, so ONE wave per SIMD —
// and a lone wave pays every dependency itself (~13 cycles per instruction; 880 cycles per 16-dimension step where the matrix work
// is 256 and the HBM share ~680).  Here wave w owns a 64-query x 64-row tile: queries 64*(w&3) .. +63 of the workgroup's 256 x the
// rows of tile w>>2 of the 128-row group: 64 accumulators, ~190 registers, TWO waves per SIMD, and per step and wave only one
// row request (wave w fetches chunk w&3 of tile w>>2: its lane's 4 floats become 4 bfloat16 = 8 bytes of the B image in LDS),
// two query-operand requests (2 KiB; the wave's partner w^4 asks for the same lines: an L1 hit), two 16-byte LDS reads and four
// matrix instructions.  HBM and L2 traffic are what they were (rows once, query operands once per 128-row group) and so are
// the LDS reads (16 KiB per step per CU) — a 32-query x 128-row tile per wave reads every B block from LDS eight times
// (32 KiB per step) and the eight waves' LDS instructions, all issued right after the barrier, held the matrix instructions
// behind them up (measured: the same time as the four-wave kernel).  Dimensions that are a multiple of 128, nq_pad of 256.
// BF: the rows come from the index's bfloat16 copy (QV_FLAG_BF16_ROWS): a wave's 1-KiB request IS one 32-row B operand of one step,
// no conversion; a round is then two steps (eight pieces, one per wave: piece w = step w>>2 of the round, block w&3).
template <int METRIC, int RING, int AR, int SPB, bool BF, bool DEFER, bool SAMPLE = false>
__global__ void __launch_bounds__(512, 1)
k_bf16x1_filter_w8(IndexView v, const uint4* __restrict__ Qbf, const float* __restrict__ cq, const float* __restrict__ mq, uint32_t nq_pad,
                   uint32_t* __restrict__ cand_rows, float* __restrict__ cand_score, uint32_t* __restrict__ cand_cnt,
                   float* __restrict__ score_out = nullptr, uint32_t score_stride = 0, uint32_t gstep = 1, uint32_t group_min = 0) {
    // SAMPLE: no filter — the sample pass of the one-term path (see k_bf16x3_filter's score_out): row groups 0, gstep, 2 gstep, ... ; cq = the
    // queries' norms (rounded up), mq = their error constants ea, eb ([nq_pad][2], k_mfma_prep with what = 1); what is written is the upper
    // bound of the row's reference distance that its one-term score implies (sample1_upper)
    // RING, AR: row chunks / query operands in flight, counted in ROUNDS; a round = SPB steps of 16 dimensions between two barriers
    __shared__ __align__(16) float s_c[256], s_m[512];
    __shared__ __align__(16) unsigned char s_b[4][SPB][4][1024];        // [stage][step of the round][32-row block][lane * 16 bytes]: 16 KiB per step of a round
    QV_CAND_QUEUE(cqu, 8, 256);                                      // 24 KiB
    QV_EPI_DUMP(du, 8, 32);                                          // 20 KiB
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t wgs_per_group = nq_pad >> 8;                      // workgroups that share one row group (different query blocks)
    uint32_t qblk_, walk0_; filter_block_role(wgs_per_group, qblk_, walk0_);
    const uint32_t qb256 = qblk_;
    const uint32_t n_groups = SAMPLE ? (score_stride + 127) / 128 : (v.n_tiles + 1) / 2;
    const uint32_t stride = gridDim.x / wgs_per_group;
    if (threadIdx.x < 256) {
        if constexpr (SAMPLE) {
            const uint32_t qq = 256 * qb256 + threadIdx.x;
            sample1_query_consts<METRIC>(cq[qq], mq[2 * qq], mq[2 * qq + 1], filter_tiny_norm(v.dim), s_c[threadIdx.x], s_m[threadIdx.x], s_m[256 + threadIdx.x]);
        } else {
            const float c = cq[256 * qb256 + threadIdx.x], m = mq[256 * qb256 + threadIdx.x];
            s_c[threadIdx.x] = METRIC == QV_COSINE ? c - m : c;
            s_m[threadIdx.x] = m;
            s_m[256 + threadIdx.x] = mq[nq_pad + 256 * qb256 + threadIdx.x];
        }
    }
    __syncthreads();
    if (stride == 0) return;
    const uint32_t half = lane >> 5, l31 = lane & 31;
    const EpiConsts ec = epi_consts<METRIC, 1>(s_c, s_m, 32 * wave, half);
    const f4* tiles = reinterpret_cast<const f4*>(v.tiles);
    const uint32_t rounds = v.dim4 / (4 * SPB);                      // a multiple of RING here
    // the running request pointers are wave-uniform (scalar registers, advanced by scalar adds); the lane's 16 bytes are an offset
    // of the request itself — 15 vector instructions per wave and step were pointer arithmetic before
    const uint4* a0 = Qbf + ((size_t)(8 * qb256 + wave) * rounds * SPB) * 2 * 64;          // the hi plane of 32-query block 8*qb256 + wave
    static_assert(!BF || SPB == 2, "the bfloat16 plane is read in rounds of two steps");
    struct Raw { f4 c[BF ? 1 : SPB]; };                              // this wave's row chunk of every step of a round (BF: its one piece of the round)
    struct Aop { uint4 h[SPB]; };
    struct Bset { uint4 h[SPB][4]; };
    Raw r[RING];                                                     // requested RING + 2 rounds ahead of their use
    Aop qa[AR];
    Bset b0, b1;
    // where this wave's converted chunk goes in a step's image: row lane of tile (wave >> 2) is row lane & 31 of block 2*(wave>>2) + (lane>>5);
    // chunk c = wave & 3 holds dims 4c .. 4c+3 of the step = bytes 8*(c&1) .. +7 of the lane slot (row, dims 8*(c>>1) .. +7)
    const uint32_t pub_off = BF ? ((wave >> 2) * 4 + (wave & 3)) * 1024 + lane * 16
                                : (2 * (wave >> 2) + (lane >> 5)) * 1024 + (l31 + 32 * ((wave & 3) >> 1)) * 16 + ((wave & 3) & 1) * 8;
    auto rows_of = [&](uint32_t g_) {
#if defined(QV_DBG_ROWS) && QV_DBG_ROWS == 1                              // measurement build: every group reads the same 16 groups (cache-resident rows)
        g_ &= 15u;
#endif
        const uint32_t ta = 2 * g_, tb = (2 * g_ + 1 < v.n_tiles) ? 2 * g_ + 1 : ta;
        if constexpr (BF)      // plane: [tile][step][row block][half][32 rows] x 16 bytes; this wave's piece: step (wave >> 2) of the round, block wave & 3
            return reinterpret_cast<const f4*>(v.bf16) + ((((size_t)((wave & 2) ? tb : ta) * (v.dim4 / 4)) + (wave >> 2)) * 2 + (wave & 1)) * 64;
        else
            return tiles + ((size_t)(wave < 4 ? ta : tb) * v.dim4 + (wave & 3)) * 64;         // wave-uniform: the lane is added in the request
    };
    const f4* lp = nullptr;
    const uint4* ap = a0;
    auto load_b_run = [&](Raw& o) {
        if constexpr (BF) { o.c[0] = __builtin_nontemporal_load(lp + lane); lp += 256; }           // next round: two steps of 128 x 16 bytes on
        else {
#pragma unroll
            for (int u = 0; u < SPB; u++) { o.c[u] = __builtin_nontemporal_load(lp + lane); lp += 256; }      // next step: four chunks on
        }
    };
    auto load_a_run = [&](Aop& o) {
#pragma unroll
        for (int u = 0; u < SPB; u++) { o.h[u] = ap[lane]; ap += 128; }
    };
    auto publish = [&](const Raw& o, uint32_t stage) {
        if constexpr (BF) { *reinterpret_cast<f4*>(&s_b[stage][0][0][0] + pub_off) = o.c[0]; return; }
#pragma unroll
        for (int u = 0; u < SPB; u++) {
            uint2 h; h.x = pack_bf16(o.c[u].x, o.c[u].y); h.y = pack_bf16(o.c[u].z, o.c[u].w);
            *reinterpret_cast<uint2*>(&s_b[stage][u][0][0] + pub_off) = h;
        }
    };
    auto read_b = [&](uint32_t stage, Bset& b) {
#pragma unroll
        for (int u = 0; u < SPB; u++)
#pragma unroll
            for (int j = 0; j < 4; j++) b.h[u][j] = *reinterpret_cast<const uint4*>(&s_b[stage][u][j][lane * 16]);
    };
    bool primed = false;
    // DEFER (a measurement, not the default: 610.7 us against 607.3): the dense pass of a row group's epilogue (the exact per-query tests and the appends, which work from the wave's dump
    // area in LDS alone) runs after the first eight steps of the NEXT group's K loop.  All eight waves reach the epilogue together
    // (the barriers keep them in step), so after the loop it is time in which the CU neither computes nor requests rows; between
    // steps its LDS round trips fill cycles the wave would wait for memory in.  (Carrying the 64 accumulators over as well, so that
    // level 1 could be deferred too, does not fit: 256 registers at two waves per SIMD, 16 spilled.)
    constexpr int UNR = RING > AR ? RING : AR;                      // both rings are indexed by the unrolled step number: the loop body covers the longer one
    uint32_t epn = 0;                                               // entries waiting in the dump area (wave-uniform)
    for (uint32_t g = walk0_; g < n_groups; g += stride) {
        const uint32_t ga = SAMPLE ? g * gstep : g;                 // the group's place in the corpus
        const uint32_t t0 = 2 * ga, t1 = (2 * ga + 1 < v.n_tiles) ? 2 * ga + 1 : t0;
        const f4* bwn = rows_of((g + stride < n_groups ? g + stride : g) * (SAMPLE ? gstep : 1u));
        f16v acc[1][4];
        double rnd[4]; float rho[4]; uint64_t alv[2];
        filter_row_consts(v, t0, t1, l31, rnd, rho, alv);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[0][j][e] = 0.f;
        if (!primed) {                                              // the workgroup's first group: fill the rings
            primed = true;
            lp = rows_of(ga);
#pragma unroll
            for (int i = 0; i < RING; i++) load_b_run(r[i]);
#pragma unroll
            for (int i = 0; i < AR - 1; i++) load_a_run(qa[i]);
            publish(r[0], 0); publish(r[1], 1);
            load_b_run(r[0]); load_b_run(r[1]);
            __syncthreads();
            read_b(0, b0);
        }
        // round s: barrier (round s+1 is published); publish round s+2 (requested RING rounds ago) and re-request its ring slot (rows
        // of round s+2+RING); read the B operands of round s+1; request the query operands of round s+AR-1; then the matrix
        // instructions of round s.  Requests that run past the end of the group are the first rounds of this workgroup's next group.
        // The LDS write goes FIRST after the barrier (its stage was last read three rounds ago) so that it and the reads drain under
        // the matrix instructions: with the write after them every round ended in "wait for my LDS write, then the barrier".
        // The scheduling barriers matter: without them the compiler hoists the conversions of LATER ring entries into this round's
        // matrix gaps, which moves their s_waitcnt up to vmcnt(1..3) — the wave then waits for rows it requested a step or two ago.
#if defined(QV_DBG_STAMP)
#define QV_STAMP(i) if (stamping && k8 == 2) { stamp[i] = __builtin_amdgcn_s_memtime(); }
        const bool stamping = blockIdx.x == 7 && g == walk0_ + 3 * stride;
        uint64_t stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#else
#define QV_STAMP(i)
#endif
        auto pstep = [&](uint32_t s_, int k8, const Bset& b_use, Bset& b_next) {
            __builtin_amdgcn_sched_barrier(0);
            QV_STAMP(0)
            __syncthreads();
#if defined(QV_DBG_STAMP)
            if (stamping && k8 == 3) { stamp[6] = __builtin_amdgcn_s_memtime(); }
#endif
            QV_STAMP(1)
            publish(r[(k8 + 2) & (RING - 1)], (uint32_t)(k8 + 2) & 3);
            if (s_ + RING + 2 == rounds) lp = bwn;
            load_b_run(r[(k8 + 2) & (RING - 1)]);
            read_b((uint32_t)(k8 + 1) & 3, b_next);
            if (s_ + (AR - 1) == rounds) ap = a0;
            load_a_run(qa[(k8 + AR - 1) & (AR - 1)]);
#if defined(QV_DBG_STAMP)
            __builtin_amdgcn_sched_barrier(0);
            QV_STAMP(2)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            QV_STAMP(3)
#endif
#pragma unroll
            for (int u = 0; u < SPB; u++) {
                const bf8 ah = __builtin_bit_cast(bf8, qa[k8 & (AR - 1)].h[u]);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf8, b_use.h[u][j]), acc[0][j], 0, 0, 0);
            }
#if !defined(QV_DBG_STAMP)
            // The eight waves leave the barrier together and the CU's vector memory path takes one 1-KiB request per 16 cycles: with
            // all requests in front of the matrix instructions a wave queued ~300 cycles before its first one (in-kernel stamps,
            // profiles/r03_batched_w8.txt).  One memory / LDS instruction per matrix-instruction gap instead: while a wave waits for
            // the memory path to take its request, its SIMD partner's matrix instructions run.
            if constexpr (BF) {
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // the LDS write of round s+2
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // row request
#pragma unroll
                for (int n = 0; n < 2; n++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // query operand request
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            } else
#pragma unroll
            for (int u = 0; u < SPB; u++) {
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // the conversion (2 VALU)
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // ... and the LDS write of round s+2
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // matrix
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // row request
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // two B operands of round s+1
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // query operand request
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // the other two B operands
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
#if defined(QV_DBG_STAMP)
            QV_STAMP(4)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            QV_STAMP(5)
#endif
        };
        for (uint32_t st = 0; st < rounds; st += UNR) {
#pragma unroll
            for (int k8 = 0; k8 < UNR; k8 += 2) { pstep(st + k8, k8, b0, b1); pstep(st + k8 + 1, k8 + 1, b1, b0); }
            if constexpr (DEFER) {
                if (st == 0 && epn >= 16) filter_epilogue_finish<METRIC>(s_c, s_m, 32 * wave, 256 * qb256 + 32 * wave, cqu, cqu_n, cqu_out, du, epn);   // the dump area holds 32 entries; a group adds ~2
            }
        }
#if defined(QV_DBG_STAMP)
        if (stamping && lane == 0)
            printf("stamp wave %u: barrier-in 0, barrier-out %llu, issued %llu, A ready %llu, mfma issued %llu, lds drained %llu, next step start %llu\n", wave,
                   stamp[1] - stamp[0], stamp[2] - stamp[0], stamp[3] - stamp[0], stamp[4] - stamp[0], stamp[5] - stamp[0], stamp[6] - stamp[0]);
#endif
        if constexpr (SAMPLE) {
            // group_min (round 4, k <= 64): per query only the SMALLEST upper bound of the group's 128 rows leaves the kernel —
            // score_out[query][group].  The k-th smallest of those minima is still an upper bound of the k-th smallest distance (they
            // belong to k different rows), nearly as tight as the k-th smallest of all bounds while the groups outnumber k several
            // times (the k best rows rarely share a group), and k_sample_bound then selects among S / 128 values per query instead
            // of S: 29 -> 5 us at 256 x 32768, and 33 MB of scores are neither written nor read.
            if (group_min) {
                // group_min (QV_MFMA_SAMPLE_GROUP_MIN=3, a measurement): only each 128-row group's MINIMUM leaves the kernel
                // (score_out[query][group]) and k_mfma_prep selects the k-th smallest of those.  Fast (no pass over every row's bound), and
                // fragile: on a corpus stored cluster by cluster a group's rows are all near or all far, the k best sample rows of a query
                // sit in few groups, and the k-th smallest minimum hands back 256 of 256 queries on 30 clusters of 10 000 rows where the
                // k-th smallest ROW bound hands back 20 (tools/dev_clustered_bound.py).  Not the default.
                float ra[4], rb[4], rc[4];                                 // the four row blocks' constants first (+inf bounds for rows that are not there),
#pragma unroll
                for (int j = 0; j < 4; j++) {                              // then one query at a time: min over the blocks, min over the half-wave's 32 rows
                    const uint32_t row = g * 128 + (j < 2 ? 0u : 64u) + 32 * (j & 1) + l31;
                    const bool gone = (j >= 2 && t1 == t0) || row >= score_stride || !((alv[j >> 1] >> (32 * (j & 1) + l31)) & 1ull);
                    sample1_row_consts<METRIC>(f32_up((float)rnd[j]), rho[j], gone, filter_tiny_norm(v.dim), ra[j], rb[j], rc[j]);
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const uint32_t ql = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float qa_ = s_c[ql], qb_ = s_m[ql], qc_ = s_m[256 + ql];
                    float m = __builtin_inff();
#pragma unroll
                    for (int j = 0; j < 4; j++) m = fminf(m, sample1_upper<METRIC>(acc[0][j][r], qa_, qb_, qc_, ra[j], rb[j], rc[j]));
#pragma unroll
                    for (int off = 1; off < 32; off <<= 1) m = fminf(m, __shfl_xor(m, off));
                    if (l31 == 0) score_out[(size_t)(256 * qb256 + ql) * n_groups + g] = m;
                }
            } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (j >= 2 && t1 == t0) continue;
                const uint32_t row = g * 128 + (j < 2 ? 0u : 64u) + 32 * (j & 1) + l31;      // place in the sample
                if (row >= score_stride) continue;
                float ra, rb, rc;
                sample1_row_consts<METRIC>(f32_up((float)rnd[j]), rho[j], !((alv[j >> 1] >> (32 * (j & 1) + l31)) & 1ull), filter_tiny_norm(v.dim), ra, rb, rc);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const uint32_t ql = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half;
                    score_out[(size_t)(256 * qb256 + ql) * score_stride + row] = sample1_upper<METRIC>(acc[0][j][r], s_c[ql], s_m[ql], s_m[256 + ql], ra, rb, rc);
                }
            }
            }
        } else if constexpr (DEFER) {
#define QV_W8_BLK(JJ) filter_epilogue_block<METRIC, 1, 4, JJ>(acc, t0, t1, s_c, s_m, 32 * wave, half, l31, 256 * qb256 + 32 * wave, filter_tiny_norm(v.dim), ec, rnd, rho, alv, cqu, cqu_n, cqu_out, du, epn)
            QV_W8_BLK(0); QV_W8_BLK(1); QV_W8_BLK(2); QV_W8_BLK(3);
#undef QV_W8_BLK
        } else filter_epilogue<METRIC>(v, acc, t0, t1, s_c, s_m, 32 * wave, half, l31, 256 * qb256 + 32 * wave, filter_tiny_norm(v.dim), ec, rnd, rho, alv, cqu, cqu_n, cqu_out, du);
    }
    if constexpr (DEFER) filter_epilogue_finish<METRIC>(s_c, s_m, 32 * wave, 256 * qb256 + 32 * wave, cqu, cqu_n, cqu_out, du, epn);
    cand_flush(cqu, cqu_n, cqu_out);
}

// k_bf16x1_filter_w8 with 256 rows per round of the K loop (round 3, late).  In the eight-wave kernel a step is paced by things that do
// not overlap because the barriers keep the waves in phase — all eight request (the CU's vector-memory path takes one request per 16
// cycles), then all eight multiply (4 matrix instructions per wave: 130 of a step's ~680 cycles), then LDS drains, then the barrier.
// Twice the rows per barrier doubles the matrix work a step's fixed costs are spread over (8 instructions per wave, 512 cycles per SIMD)
// and halves the query-operand requests per row (one 1-KiB A operand per wave and step now serves 256 rows: L2 -> L1 4.7 GB per launch
// instead of 6.2).  A wave holds 32 queries x 256 rows (128 accumulator registers); wave w converts dims 8(w&1) .. +7 of the step for the
// 64 rows of tile (w >> 1) of the group — two 16-byte chunks in, ONE 16-byte LDS write out, which is a whole lane slot of a B operand —
// and the B operands are read from LDS right before their matrix instructions (a second set of 32 registers would not fit).
__host__ __device__ static inline uint32_t w8x2_rounds(uint32_t dim4) { const uint32_t r = ((dim4 + 3) / 4 + 3) / 4 * 4; return r < 8 ? 8 : r; }
// PAD (round 4): any dimension.  The K loop runs over `rounds` = the steps of 16 dimensions rounded up to a multiple of four (eight at
// least); a chunk past the row's last one is requested from the row's first chunk instead (a valid address) and replaced by zeros,
// and k_mfma_prep pads the query operands with zeros to the same count: +0 terms, the scores and their error bounds are unchanged.
// The predicates are wave-uniform (scalar compares) and exist only in this instantiation.
template <int METRIC, int RING, int AR, bool PAD = false>
__global__ void __launch_bounds__(512, 1)
k_bf16x1_filter_w8x2(IndexView v, const uint4* __restrict__ Qbf, const float* __restrict__ cq, const float* __restrict__ mq, uint32_t nq_pad,
                     uint32_t* __restrict__ cand_rows, float* __restrict__ cand_score, uint32_t* __restrict__ cand_cnt) {
    __shared__ __align__(16) float s_c[256], s_m[512];
    __shared__ __align__(16) unsigned char s_b[4][8][1024];           // [stage][32-row block of the 256][lane * 16 bytes]: 8 KiB per step
    QV_CAND_QUEUE(cqu, 8, 256);                                      // 24 KiB
    QV_EPI_DUMP(du, 8, 32);                                          // 20 KiB
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t wgs_per_group = nq_pad >> 8;
    uint32_t qblk_, walk0_; filter_block_role(wgs_per_group, qblk_, walk0_);
    const uint32_t qb256 = qblk_;
    const uint32_t n_groups = (v.n_tiles + 3) / 4;                   // groups of four tiles = 256 rows
    const uint32_t stride = gridDim.x / wgs_per_group;
    if (threadIdx.x < 256) {
        const float c = cq[256 * qb256 + threadIdx.x], m = mq[256 * qb256 + threadIdx.x];
        s_c[threadIdx.x] = METRIC == QV_COSINE ? c - m : c;
        s_m[threadIdx.x] = m;
        s_m[256 + threadIdx.x] = mq[nq_pad + 256 * qb256 + threadIdx.x];
    }
    __syncthreads();
    if (stride == 0) return;
    const uint32_t half = lane >> 5, l31 = lane & 31;
    const EpiConsts ec = epi_consts<METRIC, 1>(s_c, s_m, 32 * wave, half);
    const f4* tiles = reinterpret_cast<const f4*>(v.tiles);
    const uint32_t rounds = PAD ? w8x2_rounds(v.dim4) : v.dim4 / 4;   // steps of 16 dimensions: a multiple of RING and AR here
    const uint4* a0 = Qbf + ((size_t)(8 * qb256 + wave) * rounds) * 2 * 64;                // the hi plane of 32-query block 8*qb256 + wave
    struct Raw { f4 c[2]; };                                         // this wave's two chunks of a step (dims 8(w&1) .. +7 of its tile's 64 rows)
    Raw r[RING];
    uint4 qa[AR];
    const uint32_t tw = wave >> 1, hw = wave & 1;
    const uint32_t pub_off = (2 * tw + (lane >> 5)) * 1024 + (l31 + 32 * hw) * 16;          // block 2*tw + lane/32, lane slot (row l31, dims 8*hw .. +7)
    auto tile_of = [&](uint32_t g_, uint32_t t_) { const uint32_t t = 4 * g_ + t_; return t < v.n_tiles ? t : v.n_tiles - 1; };   // past the end: the last tile again (masked below)
    auto rows_of = [&](uint32_t g_) { return tiles + ((size_t)tile_of(g_, tw) * v.dim4 + 2 * hw) * 64; };   // wave-uniform: the lane is added in the request
    const f4* lp = nullptr;
    const uint4* ap = a0;
    uint32_t lstep = 0;                                              // PAD: the step (of its group) the next row request belongs to
    const f4* zeros = reinterpret_cast<const f4*>(Qbf + (size_t)(nq_pad >> 5) * rounds * 128);   // PAD: 64 x 16 bytes of zeros behind the operand planes
    auto load_b_run = [&](Raw& o) {                                  // next step: four chunks on
        if constexpr (PAD) {
            // a chunk past the row's end comes from a KiB of zeros behind the query operands (k_mfma_prep): only the wave-uniform
            // address differs — two scalar selects per request, no vector instruction (selecting the VALUES measured 862 us against
            // 550 for the unpadded kernel at 960 dimensions, where nothing is padded at all: the loop's hand-placed schedule
            // counts its vector instructions)
            const uint32_t c0 = 4 * lstep + 2 * hw;
            o.c[0] = __builtin_nontemporal_load((c0 < v.dim4 ? lp : zeros) + lane);
            o.c[1] = __builtin_nontemporal_load((c0 + 1 < v.dim4 ? lp + 64 : zeros) + lane);
            lstep++;
        } else { o.c[0] = __builtin_nontemporal_load(lp + lane); o.c[1] = __builtin_nontemporal_load(lp + 64 + lane); }
        lp += 256;
    };
    auto load_a_run = [&](uint4& o) { o = ap[lane]; ap += 128; };
    auto publish = [&](const Raw& o, uint32_t stage) {
        uint4 h;
        h.x = pack_bf16(o.c[0].x, o.c[0].y); h.y = pack_bf16(o.c[0].z, o.c[0].w); h.z = pack_bf16(o.c[1].x, o.c[1].y); h.w = pack_bf16(o.c[1].z, o.c[1].w);
        *reinterpret_cast<uint4*>(&s_b[stage][0][0] + pub_off) = h;
    };
    bool primed = false;
    constexpr int UNR = RING > AR ? RING : AR;
    for (uint32_t g = walk0_; g < n_groups; g += stride) {
        const f4* bwn = rows_of(g + stride < n_groups ? g + stride : g);
        f16v accA[1][4], accB[1][4];                                // tiles 4g, 4g+1 and 4g+2, 4g+3
        const uint32_t tA0 = 4 * g, tA1 = 4 * g + 1 < v.n_tiles ? 4 * g + 1 : tA0;
        const bool hasB = 4 * g + 2 < v.n_tiles;
        const uint32_t tB0 = hasB ? 4 * g + 2 : tA0, tB1 = 4 * g + 3 < v.n_tiles ? 4 * g + 3 : tB0;
        double rndA[4], rndB[4]; float rhoA[4], rhoB[4]; uint64_t alvA[2], alvB[2];
        filter_row_consts(v, tA0, tA1, l31, rndA, rhoA, alvA);
        filter_row_consts(v, tB0, tB1, l31, rndB, rhoB, alvB);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) { accA[0][j][e] = 0.f; accB[0][j][e] = 0.f; }
        if (!primed) {                                              // the workgroup's first group: fill the rings
            primed = true;
            lp = rows_of(g); lstep = 0;
#pragma unroll
            for (int i = 0; i < RING; i++) load_b_run(r[i]);
#pragma unroll
            for (int i = 0; i < AR - 1; i++) load_a_run(qa[i]);
            publish(r[0], 0); publish(r[1], 1);
            load_b_run(r[0]); load_b_run(r[1]);
        }
        // step s: barrier (steps s and s+1 are published); publish step s+2 (requested RING steps ago) and re-request its ring slot (rows of
        // step s+2+RING); request the query operand of step s+AR-1; read the B operands of step s and multiply.  One memory or LDS
        // instruction per matrix-instruction gap, the LDS write first (see k_bf16x1_filter_w8).
#if defined(QV_DBG_STAMP)
        const bool stamping = blockIdx.x == 7 && g == walk0_ + 3 * stride;
        uint64_t stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define QV_XSTAMP(i) if (stamping && k8 == 2) { __builtin_amdgcn_sched_barrier(0); stamp[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define QV_XSTAMP(i)
#endif
        auto pstep = [&](uint32_t s_, int k8) {
            __builtin_amdgcn_sched_barrier(0);
            QV_XSTAMP(0)
            __syncthreads();
#if defined(QV_DBG_STAMP)
            if (stamping && k8 == 3) { stamp[6] = __builtin_amdgcn_s_memtime(); }
#endif
            QV_XSTAMP(1)
            publish(r[(k8 + 2) & (RING - 1)], (uint32_t)(k8 + 2) & 3);
            if (s_ + (AR - 1) == rounds) ap = a0;
            load_a_run(qa[(k8 + AR - 1) & (AR - 1)]);                // before the row requests: waiting for a step's query operand then waits for no row younger than the ones this step publishes
            if (s_ + RING + 2 == rounds) { lp = bwn; lstep = 0; }
            load_b_run(r[(k8 + 2) & (RING - 1)]);
            uint4 b[8];
#pragma unroll
            for (int j = 0; j < 8; j++) b[j] = *reinterpret_cast<const uint4*>(&s_b[(uint32_t)k8 & 3][j][lane * 16]);
            const bf8 ah = __builtin_bit_cast(bf8, qa[k8 & (AR - 1)]);
#if defined(QV_DBG_STAMP)
            QV_XSTAMP(2)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            QV_XSTAMP(3)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            QV_XSTAMP(4)
#endif
#pragma unroll
            for (int j = 0; j < 4; j++) {
                accA[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf8, b[j]), accA[0][j], 0, 0, 0);
                accB[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf8, b[4 + j]), accB[0][j], 0, 0, 0);
            }
#if defined(QV_DBG_STAMP)
            QV_XSTAMP(5)
#endif
#if !defined(QV_DBG_STAMP)
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);       // the first two B operands
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);       // the conversion (4 VALU)
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);       // ... and the LDS write of step s+2
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // query operand request
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // row request
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // row request
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
        };
        for (uint32_t st = 0; st < rounds; st += UNR) {
#pragma unroll
            for (int k8 = 0; k8 < UNR; k8++) pstep(st + k8, k8);
        }
#if defined(QV_DBG_STAMP)
        if (stamping && lane == 0)
            printf("x2 stamp wave %u: barrier-in 0, barrier-out %llu, requests + LDS ops issued %llu, A operand + rows to publish here %llu, B operands here %llu, 8 mfma issued %llu, next step's barrier-in %llu\n", wave,
                   stamp[1] - stamp[0], stamp[2] - stamp[0], stamp[3] - stamp[0], stamp[4] - stamp[0], stamp[5] - stamp[0], stamp[6] - stamp[0]);
#endif
#if defined(QV_DBG_EPI) && QV_DBG_EPI == 1                               // measurement build: no epilogue (the accumulators stay live through one compare)
        {
            float sdbg = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) sdbg += accA[0][j][e] + accB[0][j][e];
            if (sdbg == 1.2345678f) cand_cnt[0] = 1;
            continue;
        }
#endif
        {   // the eight row blocks one after the other (see filter_epilogue, NI == 1), ONE dense pass for the group
            uint32_t n = 0;
            const float tiny = filter_tiny_norm(v.dim);
#define QV_X2_BLK(AA, T0, T1, RR, HH, LL, JJ) filter_epilogue_block<METRIC, 1, 4, JJ>(AA, T0, T1, s_c, s_m, 32 * wave, half, l31, 256 * qb256 + 32 * wave, tiny, ec, RR, HH, LL, cqu, cqu_n, cqu_out, du, n)
            QV_X2_BLK(accA, tA0, tA1, rndA, rhoA, alvA, 0); QV_X2_BLK(accA, tA0, tA1, rndA, rhoA, alvA, 1);
            QV_X2_BLK(accA, tA0, tA1, rndA, rhoA, alvA, 2); QV_X2_BLK(accA, tA0, tA1, rndA, rhoA, alvA, 3);
            if (hasB) {
                QV_X2_BLK(accB, tB0, tB1, rndB, rhoB, alvB, 0); QV_X2_BLK(accB, tB0, tB1, rndB, rhoB, alvB, 1);
                QV_X2_BLK(accB, tB0, tB1, rndB, rhoB, alvB, 2); QV_X2_BLK(accB, tB0, tB1, rndB, rhoB, alvB, 3);
            }
#undef QV_X2_BLK
            filter_epilogue_finish<METRIC>(s_c, s_m, 32 * wave, 256 * qb256 + 32 * wave, cqu, cqu_n, cqu_out, du, n);
        }
    }
    cand_flush(cqu, cqu_n, cqu_out);
}

// The one-term filter reading the index's bfloat16 copy of the rows (QV_FLAG_BF16_ROWS) instead of converting float32 rows on the fly:
// half the row bytes from HBM and through the vector L1s, no conversion work.  Same workgroup shape as k_bf16x3_filter_shared (four
// query blocks share a 128-row group through LDS); wave w fetches rows 32w .. 32w+31, one 16-byte request per lane and step — which
// IS the lane's B operand.  The pipelined form only: dimensions that are a multiple of 128 (16-dim steps in rounds of eight).
// Two workgroups per CU (256 registers per lane: the lean rings leave room; the float32-row kernel at that occupancy spills into its
// loop, 1.47 ms against 1.0): a second wave per SIMD covers the other's barriers — 0.86 -> 0.79 ms, 256 x 10M x 768 5.73 -> 5.09 ms.
template <int METRIC>
__global__ void __launch_bounds__(256, 2)
k_bf16rows_filter(IndexView v, const uint4* __restrict__ Qbf, const float* __restrict__ cq, const float* __restrict__ mq, uint32_t nq_pad,
                  uint32_t* __restrict__ cand_rows, float* __restrict__ cand_score, uint32_t* __restrict__ cand_cnt) {
    __shared__ __align__(16) float s_c[4][64], s_m[8][64];
    QV_CAND_QUEUE(cqu, 4, 512);                                       // 24 KiB
    QV_EPI_DUMP(du, 4, 64);                                           // 20 KiB
    __shared__ uint4 s_b[4][4][64];                                 // [stage][32-row block][lane]: 16 KiB
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nqb64 = nq_pad >> 6;                             // a multiple of 4
    const uint32_t wgs_per_group = nqb64 >> 2;
    uint32_t qblk_, walk0_; filter_block_role(wgs_per_group, qblk_, walk0_);
    const uint32_t qb64 = qblk_ * 4 + wave;
    const uint32_t n_groups = (v.n_tiles + 1) / 2;
    const uint32_t stride = gridDim.x / wgs_per_group;
    {
        const float c = cq[64 * qb64 + lane], m = mq[64 * qb64 + lane];
        s_c[wave][lane] = METRIC == QV_COSINE ? c - m : c;
        s_m[wave][lane] = m;
        s_m[4 + wave][lane] = mq[nq_pad + 64 * qb64 + lane];
    }
    __syncthreads();
    if (stride == 0) return;
    const uint32_t half = lane >> 5, l31 = lane & 31;
    const EpiConsts ec = epi_consts<METRIC>(&s_c[0][0], &s_m[0][0], 64 * wave, half);
    const uint4* plane = reinterpret_cast<const uint4*>(v.bf16);
    const uint32_t steps = (v.dim4 + 3) / 4;                        // 16 dims per step; a multiple of 8 here
    const uint32_t dim8 = (v.dim4 + 1) / 2;
    // request pointers are wave-uniform (scalar registers, scalar adds); the lane's 16 bytes are an offset of the request itself
    const uint4* a0 = Qbf + ((size_t)(2 * qb64) * steps) * 2 * 64;
    const uint4* a1 = Qbf + ((size_t)(2 * qb64 + 1) * steps) * 2 * 64;
    constexpr int RING = 8;
    uint4 r[RING];
    uint4 qa[4][2];
    struct Bset { uint4 h[4]; };
    Bset b0, b1;
    bool primed = false;
    const uint4* lp = nullptr;
    const uint4* ap0 = a0; const uint4* ap1 = a1;
    auto rows_of = [&](uint32_t g_) {                                // 8-dim group `half` of row 32*(wave&1) + l31 of this wave's tile of group g_
        const uint32_t ta = 2 * g_, tb = (2 * g_ + 1 < v.n_tiles) ? 2 * g_ + 1 : ta;
        return plane + (size_t)(wave < 2 ? ta : tb) * dim8 * 64 + 64 * (wave & 1);    // + lane (= 32 * half + l31) in the request; dim8 is even here: dim8 * 64 = steps * 128
    };
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    auto load_b_run = [&](uint4& o) { o = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(lp + lane))); lp += 128; };
    auto load_a_run = [&](uint4 (&o)[2]) { o[0] = ap0[lane]; o[1] = ap1[lane]; ap0 += 128; ap1 += 128; };
    auto read_b = [&](uint32_t stage, Bset& b) {
#pragma unroll
        for (int j = 0; j < 4; j++) b.h[j] = s_b[stage][j][lane];
    };
    for (uint32_t g = walk0_; g < n_groups; g += stride) {
        const uint32_t t0 = 2 * g, t1 = (2 * g + 1 < v.n_tiles) ? 2 * g + 1 : t0;
        const uint4* bwn = rows_of(g + stride < n_groups ? g + stride : g);
        f16v acc[2][4];
        double rnd[4]; float rho[4]; uint64_t alv[2];
        filter_row_consts(v, t0, t1, l31, rnd, rho, alv);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
        if (!primed) {                                              // the workgroup's first group: fill the rings
            primed = true;
            lp = rows_of(g);
#pragma unroll
            for (int i = 0; i < RING; i++) load_b_run(r[i]);
#pragma unroll
            for (int i = 0; i < 3; i++) load_a_run(qa[i]);
            s_b[0][wave][lane] = r[0]; s_b[1][wave][lane] = r[1];
            load_b_run(r[0]); load_b_run(r[1]);
            __syncthreads();
            read_b(0, b0);
        }
        auto pstep = [&](uint32_t s_, int k8, const Bset& b_use, Bset& b_next) {
            __syncthreads();                                        // step s+1 is in LDS (published during step s-1)
            read_b((uint32_t)(k8 + 1) & 3, b_next);
            if (s_ + 3 == steps) { ap0 = a0; ap1 = a1; }
            load_a_run(qa[(k8 + 3) & 3]);
            __builtin_amdgcn_sched_barrier(0);
            const bf8 ah0 = __builtin_bit_cast(bf8, qa[k8 & 3][0]), ah1 = __builtin_bit_cast(bf8, qa[k8 & 3][1]);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bf8 bh = __builtin_bit_cast(bf8, b_use.h[j]);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh, acc[1][j], 0, 0, 0);
            }
            s_b[(uint32_t)(k8 + 2) & 3][wave][lane] = r[(k8 + 2) & (RING - 1)];   // rows of step s+2, requested RING steps ago
            if (s_ + RING + 2 == steps) lp = bwn;
            load_b_run(r[(k8 + 2) & (RING - 1)]);
#pragma unroll
            for (int n = 0; n < 8; n++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002 | 0x100 | 0x200 | 0x020, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        for (uint32_t st = 0; st < steps; st += RING) {
#pragma unroll
            for (int k8 = 0; k8 < RING; k8 += 2) { pstep(st + k8, k8, b0, b1); pstep(st + k8 + 1, k8 + 1, b1, b0); }
        }
        filter_epilogue<METRIC>(v, acc, t0, t1, &s_c[0][0], &s_m[0][0], 64 * wave, half, l31, 64 * qb64, filter_tiny_norm(v.dim), ec, rnd, rho, alv, cqu, cqu_n, cqu_out, du);
    }
    cand_flush(cqu, cqu_n, cqu_out);
}

// One query block (9-64 queries, the usual BatchSearch sizes) over an index with the bfloat16 row copy: all eight waves of the
// workgroup work for the SAME 64 queries, so their one-term operands (steps x 2 KiB: 96 KiB at 768 dimensions) sit in LDS for the
// whole kernel, and every wave streams its own row groups from the bfloat16 plane straight into B operands — no barrier, no
// conversion, no query traffic in the loop: one pass over half the bytes.  Dimensions that are a multiple of 64, up to 1024.
// F32 (round 4): the same kernel on the float32 tiles of an index WITHOUT the copy — a lane's two 16-byte chunk requests are its B operand's
// eight dimensions, converted right before use (4 v_cvt_pk_bf16_f32); no LDS for rows, no barrier: what 9-64 queries cost on the default
// index was the per-wave three-term kernel (0.61-0.64 ms at 1M x 768).
template <int METRIC, int NB, int R, bool F32 = false>
__global__ void __launch_bounds__(512, 1)
k_bf16rows_filter_q64(IndexView v, const uint4* __restrict__ Qbf, const float* __restrict__ cq, const float* __restrict__ mq, uint32_t nq_pad,
                      uint32_t* __restrict__ cand_rows, float* __restrict__ cand_score, uint32_t* __restrict__ cand_cnt) {
    extern __shared__ __align__(16) unsigned char smem_q64[];
    uint4* s_a = reinterpret_cast<uint4*>(smem_q64);                // [step][query half][lane]
    __shared__ __align__(16) float s_c[4][64], s_m[8][64];                        // row 0 is used (filter_epilogue's shape)
    QV_CAND_QUEUE(cqu, 8, 128);                                       // 12 KiB: the query operands take up to 128 KiB of the 160
    QV_EPI_DUMP(du, 8, 16);                                           // 10 KiB
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t steps = (v.dim4 + 3) / 4;                        // 16 dims per step; a multiple of 4 here
    const uint32_t dim8 = (v.dim4 + 1) / 2;
    for (uint32_t i = threadIdx.x; i < steps * 128; i += 512) {
        const uint32_t st = i >> 7, hq = (i >> 6) & 1, l = i & 63;
        s_a[i] = Qbf[(((size_t)hq * steps + st) * 2) * 64 + l];     // the hi plane of query half hq, step st
    }
    if (threadIdx.x < 64) {
        const float c = cq[threadIdx.x], m = mq[threadIdx.x];
        s_c[0][threadIdx.x] = METRIC == QV_COSINE ? c - m : c;
        s_m[0][threadIdx.x] = m;
        s_m[4][threadIdx.x] = mq[nq_pad + threadIdx.x];
    }
    __syncthreads();
    const uint32_t half = lane >> 5, l31 = lane & 31;
    const EpiConsts ec = epi_consts<METRIC>(&s_c[0][0], &s_m[0][0], 0u, half);
    const uint4* plane = reinterpret_cast<const uint4*>(F32 ? reinterpret_cast<const void*>(v.tiles) : reinterpret_cast<const void*>(v.bf16));   // (both are walked in 16-byte units)
    // NB = 4: a row group is two tiles (128 rows), R steps of them in flight; NB = 2: one tile per group and 64 accumulators less, which
    // buys twice the steps in flight (each wave's requests are what feeds the HBM stream: 4.9 TB/s with 16 KB per wave)
    const uint32_t n_groups = NB == 4 ? (v.n_tiles + 1) / 2 : v.n_tiles;
    const uint32_t gw = blockIdx.x * 8 + wave, tw = gridDim.x * 8;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    struct Rb { uint4 x[F32 ? 2 : 1]; };
    Rb rb[R][NB];
    const uint4* p[NB];
    auto bases = [&](uint32_t g_, const uint4* (&o)[NB]) {           // block j of group g_: rows 32*(j&1) .. +31 of the group's tile j >> 1
        const uint32_t ta = NB == 4 ? 2 * g_ : g_, tb = (NB == 4 && 2 * g_ + 1 < v.n_tiles) ? 2 * g_ + 1 : ta;
#pragma unroll
        for (int j = 0; j < NB; j++) o[j] = F32 ? plane + ((size_t)(j < 2 ? ta : tb) * v.dim4 + 2 * half) * 64 + 32 * (j & 1) + l31     // chunk 2 half (+1) of row 32 (j & 1) + l31
                                                 : plane + (size_t)(j < 2 ? ta : tb) * dim8 * 64 + 64 * (j & 1) + 32 * half + l31;
    };
    auto load_step = [&](Rb (&o)[NB]) {
#pragma unroll
        for (int j = 0; j < NB; j++) {
            o[j].x[0] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p[j])));
            if constexpr (F32) { o[j].x[1] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p[j] + 64))); p[j] += 256; }   // a step is four chunks
            else p[j] += 128;
        }
    };
    bool primed = false;
    for (uint32_t g = gw; g < n_groups; g += tw) {
        const uint32_t t0 = NB == 4 ? 2 * g : g, t1 = (NB == 4 && 2 * g + 1 < v.n_tiles) ? 2 * g + 1 : t0;
        const uint4* nb[NB];
        bases(g + tw < n_groups ? g + tw : g, nb);
        f16v acc[2][NB];
        double rnd[NB]; float rho[NB]; uint64_t alv[NB / 2];
        filter_row_consts(v, t0, t1, l31, rnd, rho, alv);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < NB; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
        if (!primed) {
            primed = true;
            bases(g, p);
#pragma unroll
            for (int i = 0; i < R; i++) load_step(rb[i]);
        }
        for (uint32_t st = 0; st < steps; st += R) {
#pragma unroll
            for (int k = 0; k < R; k++) {
                const uint4 a0 = s_a[(st + k) * 128 + lane], a1 = s_a[(st + k) * 128 + 64 + lane];
                const bf8 ah0 = __builtin_bit_cast(bf8, a0), ah1 = __builtin_bit_cast(bf8, a1);
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    uint4 bu = rb[k][j].x[0];
                    if constexpr (F32) {
                        const f4 x0 = __builtin_bit_cast(f4, rb[k][j].x[0]), x1 = __builtin_bit_cast(f4, rb[k][j].x[F32 ? 1 : 0]);
                        bu.x = pack_bf16(x0.x, x0.y); bu.y = pack_bf16(x0.z, x0.w); bu.z = pack_bf16(x1.x, x1.y); bu.w = pack_bf16(x1.z, x1.w);
                    }
                    const bf8 bh = __builtin_bit_cast(bf8, bu);
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh, acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh, acc[1][j], 0, 0, 0);
                }
                if (st + k + R == steps) {                          // the requests from here on are the first steps of this wave's next group
#pragma unroll
                    for (int j = 0; j < NB; j++) p[j] = nb[j];
                }
                load_step(rb[k]);
            }
        }
        filter_epilogue<METRIC>(v, acc, t0, t1, &s_c[0][0], &s_m[0][0], 0u, half, l31, 0u, filter_tiny_norm(v.dim), ec, rnd, rho, alv, cqu, cqu_n, cqu_out, du);
    }
    cand_flush(cqu, cqu_n, cqu_out);
}

// [lo, hi] containing the reference distance d(q, r) given an approximate score S~ with |S~ - S| <= E (the filter's bound for this
// query and row: filter_gamma |q||r|, or ea |r| + eb |r - rh| for the one-term filter); gref = the reference's own float32
// accumulation error (QV_L2SQ); qn_cos = the cosine metric's own query norm, qn_l2 = |q|, rn = the stored row norm
template <int M>
__device__ __forceinline__ void score_interval(double S, double qn_cos, double qn_l2, double rn, double E, double gref, float& lo, float& hi) {
    if (!(__builtin_fabs(S) < 3.0e38)) { lo = -__builtin_inff(); hi = __builtin_inff(); return; }   // overflowed float32 sum (or NaN): no information
    double d, e;
    if constexpr (M == QV_COSINE) {
        if (qn_cos == 0.0 || rn == 0.0) { d = 1.0; e = 0.0; }
        else { d = 1.0 - S / (qn_cos * rn); e = E / (qn_cos * rn) + 2e-6; }
    } else if constexpr (M == QV_DOT) {
        d = 1.0 - S; e = E + 2e-6 * (1.0 + __builtin_fabs(d));
    } else {                                                        // QV_L2 / QV_L2SQ: interval on d^2, then into the metric's units
        const double q2 = qn_l2 * qn_l2, r2 = rn * rn;
        const double d2 = q2 + r2 - 2.0 * S, e2 = 2.0 * E + 2e-6 * (q2 + r2);
        double l2 = d2 - e2 > 0.0 ? d2 - e2 : 0.0, h2 = d2 + e2 > 0.0 ? d2 + e2 : 0.0;
        if constexpr (M == QV_L2) { l2 = __builtin_sqrt(l2) * (1.0 - 4e-7); h2 = __builtin_sqrt(h2) * (1.0 + 4e-7); }
        else { l2 = l2 * (1.0 - gref - 2e-6); h2 = h2 * (1.0 + gref + 2e-6); }
        lo = f32_down((float)l2); hi = f32_up((float)h2);
        if (!(d2 == d2)) { lo = -__builtin_inff(); hi = __builtin_inff(); }
        return;
    }
    lo = f32_down((float)(d - e)); hi = f32_up((float)(d + e));
    if (!(d == d)) { lo = -__builtin_inff(); hi = __builtin_inff(); }   // NaN score: keep, the exact pass decides
}

// The sample bound without an exact scan: the three-term filter kernel's scores of the sample give every sampled row an UPPER
// bound of its reference distance (written by that kernel: score_upper_f32), and the k-th smallest of those bounds is at least the
// k-th smallest true distance over the sample, hence over the corpus.  One workgroup per query (and part); a pure selection.
// The k-th smallest (kk >= 1) of the ordered keys[0 .. n) in LDS, by the whole workgroup (any multiple of 64 threads): a radix selection, 8 bits a
// pass from the first bit in which the keys differ at all (distances of one query share their exponent and leading mantissa bits: from bit
// 31 down the first passes throw every key at ONE bin — serialized LDS atomics), 256 bins.  Every thread returns the key; 0xFFFFFFFF when
// there are fewer than kk keys.  s_bins: 256 words, s_scal: 4 words of LDS.  Contains barriers: call it from uniform control flow.
__device__ __forceinline__ uint32_t block_kth_smallest(const uint32_t* keys, uint32_t n, uint32_t kk, uint32_t* s_bins, uint32_t* s_scal) {
    const uint32_t t = threadIdx.x, nt = blockDim.x, lane = lane_id(), wave = threadIdx.x >> 6;
    if (t == 0) { s_scal[0] = 0xFFFFFFFFu; s_scal[1] = 0u; }
    __syncthreads();
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (uint32_t i = t; i < n; i += nt) { const uint32_t x = keys[i]; lo = x < lo ? x : lo; hi = x > hi ? x : hi; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const uint32_t a = __shfl_xor(lo, m), b = __shfl_xor(hi, m); lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
    if (lane == 0) { atomicMin(&s_scal[0], lo); atomicMax(&s_scal[1], hi); }
    __syncthreads();
    lo = s_scal[0]; hi = s_scal[1];
    if (kk > n) return 0xFFFFFFFFu;
    if (lo == hi) return lo;
    const int top = 31 - __builtin_clz(lo ^ hi);
    uint32_t prefix = top >= 31 ? 0u : (lo >> (top + 1)) << (top + 1);
    uint32_t pmask = top >= 31 ? 0u : ~((1u << (top + 1)) - 1u);
    uint32_t krem = kk;
    for (int shift = top - 7; ; shift -= 8) {
        const int sh = shift < 0 ? 0 : shift;
        const uint32_t dmask = shift < 0 ? ((1u << (shift + 8)) - 1u) : 255u;
        for (uint32_t i = t; i < 256; i += nt) s_bins[i] = 0;
        __syncthreads();
        for (uint32_t i = t; i < n; i += nt) {
            const uint32_t x = keys[i];
            if ((x & pmask) == prefix) atomicAdd(&s_bins[(x >> sh) & dmask], 1u);
        }
        __syncthreads();
        if (wave == 0) {                                              // 256 bins over 64 lanes: which bin holds the krem-th
            uint32_t b4[4], sum = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) { b4[u] = s_bins[4 * lane + u]; sum += b4[u]; }
            uint32_t inc = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(inc, off); if ((int)lane >= off) inc += y; }
            const uint32_t excl = inc - sum;
            if (excl < krem && krem <= inc) {
                uint32_t d = 4 * lane, cum = excl;
#pragma unroll
                for (int u = 0; u < 4; u++) { if (krem > cum + b4[u] && u < 3) { cum += b4[u]; d++; } else break; }
                s_scal[2] = d; s_scal[3] = cum;
            }
        }
        __syncthreads();
        prefix |= s_scal[2] << sh; pmask |= dmask << sh; krem -= s_scal[3];
        if (shift <= 0) break;
    }
    return prefix;
}

// The same selection in two stages, all of it in LDS (round 4, late): k_sample_bound's sixteen waves keep sorted lists and insert one
// key at a time — a chain of dependent steps per row that passes (29 us for 32 768 bounds per query, 85 us at k = 64).  Here thread t
// reads its chunk of the bounds (rows t, t + 1024, ...) and keeps its minimum; the k-th smallest of the 1024 minima, T1, is at or above the k-th
// smallest bound U (k different rows are at or below it), and every row at or below U lies in a chunk whose minimum is at or below
// T1 — k chunks, more only on exact ties.  Their rows are read again (from L2) into LDS, and U is the k-th smallest of them: a
// radix selection on the ordered bits, 8 bits a pass from the first bit in which the values differ, 256 LDS bins, whole workgroup.
// Writes what k_mfma_prep reads of k_sample_bound's output: sample_dist[q][k - 1].  Needs k chunks' rows to fit kSelKeys.
constexpr uint32_t kSelKeys = 24576;                                 // 96 KiB of ordered keys (k = 128 over a sixth of a million rows)
__host__ __device__ static inline uint32_t sample_select_chunk(uint32_t srows) { return ((srows + 1023) / 1024 + 3) / 4 * 4; }
static size_t sample_select_lds_bytes(uint32_t srows, uint32_t k) { return (size_t)std::min<uint64_t>(kSelKeys, (uint64_t)(k + 8) * sample_select_chunk(srows)) * 4; }   // (a few chunks of slack for exact ties)
static bool sample_select_applies(uint32_t srows, uint32_t k) { return srows >= 4096 && (uint64_t)k * sample_select_chunk(srows) <= kSelKeys && k <= 1024; }
template <int M>
__global__ void __launch_bounds__(1024)
k_sample_select(const float* __restrict__ bounds, uint32_t srows, uint32_t k, float gref, float* __restrict__ sample_dist) {
    extern __shared__ uint32_t s_keys[];                              // k chunks' rows (sample_select_lds_bytes): small k leaves room for a second block per CU
    __shared__ uint32_t s_min[1024];
    __shared__ uint32_t s_bins[256];
    __shared__ uint32_t s_sel[1024];
    __shared__ uint32_t s_n, s_scal[4];
    const uint32_t t = threadIdx.x;
    const uint32_t qi = blockIdx.x;
    const float* sc = bounds + (size_t)qi * srows;
    const uint32_t chunk = sample_select_chunk(srows);
    // stage 1: this thread's chunk = rows t, t + 1024, t + 2048, ... (consecutive threads read consecutive bounds): its minimum
    {
        float m = __builtin_inff();
        for (uint32_t i0 = 0; i0 < chunk; i0 += 8) {
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint32_t r = t + 1024u * (i0 + u); x[u] = i0 + u < chunk && r < srows ? sc[r] : __builtin_inff(); }
#pragma unroll
            for (int u = 0; u < 8; u++) m = fminf(m, x[u]);
        }
        s_min[t] = ord_f32(m);
    }
    if (t == 0) s_n = 0;
    __syncthreads();
    const uint32_t t1 = block_kth_smallest(s_min, 1024, k, s_bins, s_scal);
    uint32_t ukey = t1;
    if (t1 != 0xFFFFFFFFu && t1 < ord_f32(__builtin_inff())) {
        // stage 2: the chunks at or below T1 (beyond kSelKeys / chunk of them — exact ties only — the bound comes from a subset: still k rows)
        const uint32_t cap_keys = (k + 8) * chunk < kSelKeys ? (k + 8) * chunk : kSelKeys;   // = sample_select_lds_bytes / 4
        const uint32_t cap = cap_keys / chunk;
        if (s_min[t] <= t1) { const uint32_t slot = atomicAdd(&s_n, 1u); if (slot < 1024) s_sel[slot] = t; }
        __syncthreads();
        const uint32_t n_sel = s_n < cap ? s_n : cap;
        const uint32_t total = n_sel * chunk;
        for (uint32_t i = t; i < total; i += 1024) {
            const uint32_t e = i / chunk, r = s_sel[e] + 1024u * (i - e * chunk);
            s_keys[i] = ord_f32(r < srows ? sc[r] : __builtin_inff());
        }
        __syncthreads();
        ukey = block_kth_smallest(s_keys, total, k, s_bins, s_scal);
    }
    if (t == 0) {
        const float x = ukey == 0xFFFFFFFFu ? __builtin_inff() : unord_f32(ukey);
        sample_dist[(size_t)qi * k + (k - 1)] = x == x && x < __builtin_inff() ? sample_bound_finish<M>(x, gref) : __builtin_inff();
    }
}

// (round 3's selection by sorted wave lists: since round 6 every shape that reached it takes k_sample_select or the selection path;
// kept in the measurement build behind QV_MFMA_SAMPLE_SELECT=2)
#ifdef QV_VARIANTS
template <int M>
__global__ void __launch_bounds__(1024)
k_sample_bound(const float* __restrict__ bounds, uint32_t srows, uint32_t k, float gref, float* __restrict__ sample_dist, uint32_t parts) {
    __shared__ uint64_t wl[16 * 64];
    __shared__ uint64_t s_sorted[64];
    __shared__ unsigned long long s_thr;                            // the smallest k-th key any wave of the workgroup holds: no key above it is in the answer
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t qi = blockIdx.x;
    if (threadIdx.x < 64) s_sorted[threadIdx.x] = kDeadKey;
    if (threadIdx.x == 0) s_thr = kDeadKey;
    __syncthreads();
    const uint32_t kth = k - 1;
    const float* sc = bounds + (size_t)qi * srows;
    // grid (nq, parts): this workgroup takes the rows [r_lo, r_hi) of the sample and writes the k smallest upper bounds it saw, ascending;
    // k_mfma_prep takes the k-th smallest over the parts (a few queries alone would leave most CUs idle: 38 us for 64 queries)
    const uint32_t part = blockIdx.y;
    const uint32_t per_part = ((srows + parts - 1) / parts + 63) / 64 * 64;
    const uint32_t r_lo = part * per_part, r_hi = r_lo + per_part < srows ? r_lo + per_part : srows;
    uint64_t list = kDeadKey, thr = kDeadKey;
    // eight batches of 64 rows per round, requested together
    for (uint32_t base0 = r_lo + wave * 64; base0 < r_hi; base0 += 8 * 16 * 64) {
        float scv[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t row = base0 + (uint32_t)j * (16 * 64) + lane;                        // place in the sample
            scv[j] = row < r_hi ? __builtin_nontemporal_load(&sc[row]) : __builtin_inff();
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t row = base0 + (uint32_t)j * (16 * 64) + lane;
            const uint64_t key = row < r_hi && scv[j] < __builtin_inff() ? make_key(scv[j], row) : kDeadKey;
            if (base0 + (uint32_t)j * (16 * 64) >= r_hi) continue;
            if (base0 == r_lo + wave * 64 && j == 0) list_seed(list, thr, key, kth, lane); else list_insert(list, thr, key, kth, lane);
            // the sixteen waves tighten ONE threshold (each alone would take ~k ln(rows/k) serial inserts: 1 000 per workgroup against ~100)
            if (lane == 0 && thr < s_thr) atomicMin(&s_thr, (unsigned long long)thr);
            const uint64_t shared_thr = s_thr;
            thr = shared_thr < thr ? shared_thr : thr;
        }
    }
    wl[wave * 64 + lane] = list;
    __syncthreads();
    {   // merge of the sixteen lists by rank: a wave counts, for each of its k keys, the keys of the workgroup below it (keys are
        // distinct: the row is part of them) and puts those that rank under k in place — no serial inserts
        const uint64_t mine = lane < k ? list : kDeadKey;
        uint32_t rank = 0;
        for (uint32_t w = 0; w < 16; w++)
            for (uint32_t i = 0; i < k; i++) rank += wl[w * 64 + i] < mine ? 1u : 0u;
        if (mine != kDeadKey && rank < k) s_sorted[rank] = mine;
    }
    __syncthreads();
    if (wave == 0) {
        const uint64_t mine = s_sorted[lane];
        const float val = mine == kDeadKey ? __builtin_inff() : sample_bound_finish<M>(unord_f32((uint32_t)(mine >> 32)), gref);
        if (parts <= 1) { if (lane == kth) sample_dist[(size_t)qi * k + kth] = val; }
        else if (lane < k) sample_dist[((size_t)qi * parts + part) * k + lane] = val;
    }
}
#endif

// Exact re-scoring of one query's candidates + top-k; one workgroup per query.
// Stage 1 narrows the candidates with their fp32 scores: with d~ the approximate distance and
// e_r its error bound, d_r is in [d~ - e_r, d~ + e_r]; let H be the k-th smallest upper bound
// over the candidates (which contain the true top-k).  Then the true k-th distance is <= H, so
// only candidates with lower bound <= H can be in the answer — typically k..k+2 of hundreds.
// Stage 2 re-scores those exactly (same arithmetic as k_flat_scan) and sorts them.
template <int M, int U>
__global__ void __launch_bounds__(256)
k_rescore_select(IndexView v, const float* __restrict__ queries, const uint32_t* __restrict__ cand_rows, const float* __restrict__ cand_score,
                 const uint32_t* __restrict__ cand_cnt, uint32_t k, uint32_t* __restrict__ rows_out, float* __restrict__ dist_out,
                 uint32_t* __restrict__ overflow, const float* __restrict__ eq /*[nq][2]: |S~ - S| <= eq[0] |r| + eq[1] |r - rh| (k_mfma_prep)*/) {
    using Q = typename MT<M>::Q;
    extern __shared__ __align__(16) unsigned char smem[];
    Q* q_lds = reinterpret_cast<Q*>(smem);
    uint64_t* wl = reinterpret_cast<uint64_t*>(smem + (((size_t)v.dim4 * 4 * sizeof(Q)) + 15) / 16 * 16);   // [4][64]
    uint32_t* surv = reinterpret_cast<uint32_t*>(wl + 4 * 64);                                               // [kMfmaCandCap]
    __shared__ uint32_t s_ns;
    __shared__ float s_H;
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t qi = blockIdx.x;
    const uint32_t cnt = cand_cnt[qi];
    if (cnt > (uint32_t)kMfmaCandCap) { if (threadIdx.x == 0) overflow[qi] = 1; return; }
#ifdef QV_RS_PROF
    uint64_t tk[6]; int tn = 0; tk[tn++] = wall_clock64();
#define RSTK() tk[tn++] = wall_clock64()
#else
#define RSTK()
#endif
    __shared__ double s_part[4], s_qn_exact;
    stage_query<M>(q_lds, queries + (size_t)qi * v.dim, v.dim, v.dim4);
    if (threadIdx.x == 0) s_ns = 0;
    __syncthreads();
    // |q| for the error bounds: it only enters intervals that carry 2e-6 of slack, so the order of the additions is free — a sum over
    // the workgroup's 256 threads.  The cosine metric's OWN query norm (a chain of dim rounded fmas in element order, distances.go:20:
    // 4 us on one lane) is only needed by stage 2's finalize: wave 3 walks it while the other waves are already in stage 1
    // (every thread used to walk it up front, and the non-cosine metrics walked a second chain for |q|).
    {
        double n2 = 0.0;
        for (uint32_t i = threadIdx.x; i < v.dim; i += blockDim.x) { const double a = (double)q_lds[i]; n2 = __builtin_fma(a, a, n2); }
        n2 = wave_sum_f64(n2);
        if (lane == 0) s_part[wave] = n2;
    }
    __syncthreads();
    const double qn_l2 = __builtin_sqrt((s_part[0] + s_part[1]) + (s_part[2] + s_part[3]));
    QConst qc; qc.qn = qn_l2; qc.qn32 = 0.f;                        // stage 1: the approximate norm; stage 2 replaces it for cosine
    if constexpr (M == QV_COSINE) {
        if (wave == 3) { const QConst t = query_const<M>(q_lds, v.dim); if (lane == 0) s_qn_exact = t.qn; }
    }
    const uint32_t kth = k - 1;
    const uint32_t* cr = cand_rows + (size_t)qi * kMfmaCandCap;
    const float* cs = cand_score + (size_t)qi * kMfmaCandCap;
    // gamma: |S~ - S| <= gamma |q||r| for the filter kernel that produced the scores (filter_gamma)
    RSTK();

    // ---- stage 1: H = k-th smallest upper bound.  Every candidate's interval, four candidates per thread at a time: their scores and rows
    // requested together, then their norms and residuals together (two round trips per 1024 candidates instead of two per 256); lower
    // bounds and the upper bounds' ordered keys are parked in LDS, and H is a radix selection over the keys by the whole workgroup
    // (the four waves' sorted lists and their merge were ~30 serial inserts at k = 10, ~190 at k = 64: 12 / 27 us of this kernel).
    const double ea = (double)eq[2 * qi], eb = (double)eq[2 * qi + 1], gref = filter_gamma(v.dim, 0);
    uint64_t list = kDeadKey, thr = kDeadKey;
    float* lo_l = reinterpret_cast<float*>(surv + kMfmaCandCap);                                              // [kMfmaCandCap]
    uint32_t* hi_k = reinterpret_cast<uint32_t*>(lo_l + kMfmaCandCap);                                        // [kMfmaCandCap]
    __shared__ uint32_t s_bins[256], s_scal[4];
    for (uint32_t base0 = wave * 64; base0 < cnt; base0 += 4 * 4 * 64) {
        uint32_t rowv[4]; float scv[4]; double rnv[4]; float rhv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { const uint32_t i = base0 + (uint32_t)j * 256 + lane; rowv[j] = i < cnt ? cr[i] : 0u; scv[j] = i < cnt ? cs[i] : 0.f; }
#pragma unroll
        for (int j = 0; j < 4; j++) { const uint32_t i = base0 + (uint32_t)j * 256 + lane; rnv[j] = i < cnt ? v.rnorm[rowv[j]] : 0.0; rhv[j] = i < cnt ? v.rres[rowv[j]] : 0.f; }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t i = base0 + (uint32_t)j * 256 + lane;
            if (i < cnt) {
                float lo, hi;
                score_interval<M>((double)scv[j], qc.qn, qn_l2, rnv[j], ea * rnv[j] + eb * (double)rhv[j], gref, lo, hi);
                lo_l[i] = lo;
                hi_k[i] = ord_f32(hi == hi ? hi : __builtin_inff());
            }
        }
    }
    __syncthreads();
    {
        const uint32_t hk = block_kth_smallest(hi_k, cnt, k, s_bins, s_scal);
        if (threadIdx.x == 0) s_H = hk == 0xFFFFFFFFu ? __builtin_inff() : unord_f32(hk);          // < k candidates: keep all
    }
    __syncthreads();
    RSTK();
    const float H = s_H;
    for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x)
        if (lo_l[i] <= H) surv[atomicAdd(&s_ns, 1u)] = cr[i];
    __syncthreads();
    const uint32_t ns = s_ns;
    if constexpr (M == QV_COSINE) qc.qn = s_qn_exact;              // from here on the metric's own norm (wave 3 wrote it before the barriers above)
    RSTK();

    // ---- stage 2: exact distances of the survivors, top-k by (distance, row)
    list = kDeadKey; thr = kDeadKey;
    for (uint32_t base = wave * 64; base < ns; base += 4 * 64) {
        const uint32_t i = base + lane;
        uint64_t key = kDeadKey;
        if (i < ns) {
            const uint32_t row = surv[i];
            // a survivor's row out of the tile layout is dim4 separate 16-byte pieces, one memory line each (~80 survivors per query
            // with the one-term filter: the pass is bound by those lines); the row-major copy, when the index keeps one, holds the
            // same floats contiguously — same values, same order, same result
            typename MT<M>::A acc;
            if (v.rowmaj != nullptr && (v.dim & 3) == 0) acc = row_accumulate<M, U, false, true>(reinterpret_cast<const f4*>(v.rowmaj + (size_t)row * v.dim), 1, q_lds, v.dim4);
            else acc = row_accumulate<M, U, false, true>(reinterpret_cast<const f4*>(v.tiles) + (size_t)(row >> 6) * v.dim4 * 64 + (row & 63), 64, q_lds, v.dim4);
            double rn = 0.0;
            if constexpr (MT<M>::needs_rnorm) rn = v.rnorm[row];
            key = make_key(finalize<M>(acc, qc, rn), row);
        }
        if (base == wave * 64) list_seed(list, thr, key, kth, lane); else list_insert(list, thr, key, kth, lane);
    }
    __syncthreads();
    wl[wave * 64 + lane] = list;
    __syncthreads();
    if (wave == 0) {
        for (uint32_t w = 1; w < 4; w++) {
            uint64_t key = lane < k ? wl[w * 64 + lane] : kDeadKey;
            list_insert(list, thr, key, kth, lane);
        }
        if (lane < k) {
            bool dead = list == kDeadKey;
            rows_out[(size_t)qi * k + lane] = dead ? 0xFFFFFFFFu : (uint32_t)list;
            dist_out[(size_t)qi * k + lane] = dead ? __uint_as_float(0x7F800000u) : unord_f32((uint32_t)(list >> 32));
        }
#ifdef QV_RS_PROF
        RSTK();
        if (lane == 0 && (qi == 3 || qi == 200)) printf("rescore q%u: cnt %u survivors %u | stage+const %llu, H %llu, survivors %llu, exact+merge %llu (x10 ns)\n", qi, cnt, ns,
                                                        tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2], tk[4] - tk[3]);
#endif
    }
#undef RSTK
}

// ---------------------------------------------------------------- the selection path: 16 <= k <= kMaxBatchedK --
// HybridIndex.BatchSearch takes any k (hybrid_index.go:677-811), and the negative-example branches ask for max(2k, 30)
// (:516-522).  The wave lists above hold 64 keys; beyond — and, since round 6, from 16 results per query wherever the bound can be a guess
// (batched_large_k) — the same three steps, the sample's bound, the candidates' narrowing, the exact top-k, work on arrays of keys:
//   the bound                     a guess from a small sample, rank k S / N + 4 sigma (batched_guess; k_sample_select), or — fewer than
//                                 131 072 rows — the k-th smallest bound of half the corpus: k_sample_select, or k_sample_hist<0>, <1> (to 24
//                                 bits, rounded UP to its bucket's edge: a valid bound, 2^-15 looser at most) where k chunks outgrow its LDS
//   k_cand_narrow                 per query: every candidate's interval [lo, hi] from its score, H = the k-th smallest hi (the true k-th
//                                 distance is at most H), the guess's check H <= U, the candidates with lo <= H compacted (~2.1 k of them);
//                                 above 2 048 results as k_cand_qnorms, k_cand_bounds, a selection over (hi, slot) keys, k_cand_survive
//   k_cand_exact_wave / k_tp_*    their exact distances, the scan's own arithmetic: a wave per 32 survivors, or one pass over the tiles
//   k_select_sort                 the k best of those, in (distance, row) order, a workgroup per query over the counted prefix
template <int W>
__global__ void __launch_bounds__(kSelBlock)
k_sample_hist(const float* __restrict__ bounds, uint32_t srows, uint32_t k, SelState* __restrict__ st, uint32_t* __restrict__ hist) {
    constexpr int shift = SelWindow<W>::shift;
    constexpr int wbits = SelWindow<W>::wbits;
    __shared__ uint32_t h[kSelBins];
    const uint32_t q = blockIdx.y;
    SelState* s = st + q;
    const unsigned long long prefix = W > 0 ? s->prefix : 0ull;
    SelRun run{0u, 0u};
    for (uint32_t b = threadIdx.x; b < (uint32_t)kSelBins; b += kSelBlock) h[b] = 0;
    __syncthreads();
    const float* src = bounds + (size_t)q * srows;
    const uint32_t per = ((srows + gridDim.x - 1) / gridDim.x + kSelBlock - 1) / kSelBlock * kSelBlock;
    const uint32_t lo = blockIdx.x * per, hi = lo + per < srows ? lo + per : srows;
    for (uint32_t base = lo; base < hi; base += 16 * kSelBlock) {      // sixteen requests per thread and round, lanes contiguous
        float e[16];
#pragma unroll
        for (int u = 0; u < 16; u++) { const uint32_t j = base + (uint32_t)u * kSelBlock + threadIdx.x; e[u] = j < hi ? __builtin_nontemporal_load(&src[j]) : __builtin_inff(); }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (base + (uint32_t)u * kSelBlock >= hi) break;
            const uint32_t j = base + (uint32_t)u * kSelBlock + threadIdx.x;
            const uint64_t key = (uint64_t)ord_f32(e[u]) << 32;
            sel_count<W>(h, run, key, j < hi && (W == 0 || (key >> (shift + wbits)) == (prefix >> (shift + wbits))));
        }
    }
    sel_flush(h, run);
    sel_finish_window<W>(h, hist + (size_t)q * kSelBins, s, gridDim.x, k, 0u);
}
// U_q = the upper edge of the 24-bit bucket that holds the k-th smallest bound (+inf when the sample has fewer than k finite ones)
template <int M>
__global__ void k_sample_bound_from_state(const SelState* __restrict__ st, uint32_t nq, uint32_t k, float gref, float* __restrict__ sample_dist) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const uint32_t top = (uint32_t)(st[q].prefix >> 32) | 0xFFu;
    const float edge = unord_f32(top);
    sample_dist[(size_t)q * k + (k - 1)] = edge == edge && edge < __builtin_inff() ? sample_bound_finish<M>(edge, gref) : __builtin_inff();
}

// per query: |q| for the intervals (any order of additions: it only enters bounds with 2e-6 of slack) and, for cosine, the
// metric's own norm — a chain of dim rounded fmas in element order (distances.go:20) walked by one lane
template <int M>
__global__ void __launch_bounds__(64)
k_cand_qnorms(const float* __restrict__ queries, uint32_t dim, double* __restrict__ qnorms /*[nq][2]: |q| approx, the metric's norm*/) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* ql = reinterpret_cast<float*>(smem);                        // the query, staged with the wave's 64 lanes (one lane walking it from global memory: 30 us)
    const uint32_t q = blockIdx.x, lane = lane_id();
    const float* src = queries + (size_t)q * dim;
    double n2 = 0.0;
    for (uint32_t i = lane; i < dim; i += 64) { const float x = src[i]; ql[i] = x; const double a = (double)x; n2 = __builtin_fma(a, a, n2); }
    n2 = wave_sum_f64(n2);
    __syncthreads();
    double exact = 0.0;
    if constexpr (M == QV_COSINE) { if (lane == 0) { double ma = 0.0; for (uint32_t i = 0; i < dim; i++) { const double a = (double)ql[i]; ma = __builtin_fma(a, a, ma); } exact = __builtin_sqrt(ma); } }
    if (lane == 0) { qnorms[2 * q] = __builtin_sqrt(n2); qnorms[2 * q + 1] = exact; }
}

// grid (cap / 256, nq)
template <int M>
__global__ void __launch_bounds__(256)
k_cand_bounds(IndexView v, const uint32_t* __restrict__ cand_rows, const float* __restrict__ cand_score, const uint32_t* __restrict__ cand_cnt, uint32_t cap,
              const float* __restrict__ eq, const double* __restrict__ qnorms, uint64_t* __restrict__ keys_hi, float* __restrict__ lo_out,
              uint32_t* __restrict__ overflow, uint32_t* __restrict__ n_surv) {
    const uint32_t q = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t cnt = cand_cnt[q];
    if (cnt > cap) { if (i == 0) overflow[q] = 1; cnt = 0; }           // the caller redoes this query with the exact scan
    if (i == 0) n_surv[q] = 0;
#ifdef QV_LK_PROF
    if (i == 0 && (q == 0 || q == 100)) printf("large-k re-score q%u: candidates %u of cap %u\n", q, cand_cnt[q], cap);
#endif
    if (i >= cap) return;
    uint64_t key = kDeadKey;
    if (i < cnt) {
        const uint32_t row = cand_rows[(size_t)q * cap + i];
        const double rn = v.rnorm[row], qn = qnorms[2 * q];
        float lo, hi;
        score_interval<M>((double)cand_score[(size_t)q * cap + i], qn, qn, rn, (double)eq[2 * q] * rn + (double)eq[2 * q + 1] * (double)v.rres[row], filter_gamma(v.dim, 0), lo, hi);
        lo_out[(size_t)q * cap + i] = lo;
        key = make_key(hi, i);
    }
    keys_hi[(size_t)q * cap + i] = key;
}

// k_cand_qnorms, k_cand_bounds, the selection of H and k_cand_survive in ONE kernel, a workgroup per query, while a query's candidate slots
// (as ordered 32-bit keys) and the query fit LDS — up to 32 768 slots = 2 048 results.  Five launches on the batch's critical path
// (63 us of 880 at k = 100, 122 of 1530 at k = 1000, 256 x 1M x 768) for work that is a few thousand candidates per query.
// Phases: the query into LDS and |q| (any order); lane 0 of wave 0 walks the cosine norm's chain (distances.go:20) while the other
// waves turn scores into intervals (lower bounds to lo_out, upper bounds' keys to LDS); H = the k-th smallest upper bound
// (block_kth_smallest); the guess's check; the candidates with lo <= H compacted into surv.
template <int M>
__global__ void __launch_bounds__(1024)
k_cand_narrow(IndexView v, const float* __restrict__ queries, const uint32_t* __restrict__ cand_rows, const float* __restrict__ cand_score,
              const uint32_t* __restrict__ cand_cnt, uint32_t cap, const float* __restrict__ eq, uint32_t k, const float* __restrict__ guess, uint32_t ks,
              double* __restrict__ qnorms, float* __restrict__ lo_out, uint32_t* __restrict__ surv, uint32_t* __restrict__ n_surv, uint32_t* __restrict__ overflow,
              uint32_t* __restrict__ zero_words, uint32_t n_zero /* the tile pass's counters, cleared here for k_tp_count (a memset of its own: three fill kernels, 25 us) */) {
    extern __shared__ __align__(16) unsigned char smem[];
    for (uint32_t i = blockIdx.x * 1024u + threadIdx.x; i < n_zero; i += gridDim.x * 1024u) zero_words[i] = 0u;
    uint32_t* s_keys = reinterpret_cast<uint32_t*>(smem);                  // [cap]
    float* ql = reinterpret_cast<float*>(smem + (size_t)cap * 4);          // [dim]
    __shared__ uint32_t s_bins[256];
    __shared__ uint32_t s_scal[4];
    __shared__ double s_part[16];
    __shared__ uint32_t s_n;
    const uint32_t q = blockIdx.x, t = threadIdx.x, lane = lane_id(), wave = t >> 6;
    const uint32_t cnt_raw = cand_cnt[q];
    const uint32_t cnt = cnt_raw <= cap ? cnt_raw : 0u;
    if (t == 0) { s_n = 0; if (cnt_raw > cap) overflow[q] = 1; }           // (more candidates than slots: the caller redoes this query with the exact scan)
    const float* src = queries + (size_t)q * v.dim;
    double n2 = 0.0;
    for (uint32_t i = t; i < v.dim; i += 1024) { const float x = src[i]; ql[i] = x; const double a = (double)x; n2 = __builtin_fma(a, a, n2); }
    n2 = wave_sum_f64(n2);
    if (lane == 0) s_part[wave] = n2;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < 16; w++) tot += s_part[w];
    const double qn = __builtin_sqrt(tot);
    if (t == 0) {
        double exact = 0.0;
        if constexpr (M == QV_COSINE) { double ma = 0.0; for (uint32_t i = 0; i < v.dim; i++) { const double a = (double)ql[i]; ma = __builtin_fma(a, a, ma); } exact = __builtin_sqrt(ma); }
        qnorms[2 * q] = qn; qnorms[2 * q + 1] = exact;
    }
    const double ea = (double)eq[2 * q], eb = (double)eq[2 * q + 1], gref = filter_gamma(v.dim, 0);
    if (wave != 0 || M != QV_COSINE) {                                     // (cosine: wave 0 is busy with the chain; its share goes to the others)
        const uint32_t first = M == QV_COSINE ? t - 64 : t, step = M == QV_COSINE ? 960u : 1024u;
        for (uint32_t i = first; i < cnt; i += step) {
            const uint32_t row = cand_rows[(size_t)q * cap + i];
            const double rn = v.rnorm[row];
            float lo, hi;
            score_interval<M>((double)cand_score[(size_t)q * cap + i], qn, qn, rn, ea * rn + eb * (double)v.rres[row], gref, lo, hi);
            lo_out[(size_t)q * cap + i] = lo;
            s_keys[i] = ord_f32(hi);
        }
    }
    __syncthreads();
    const uint32_t hkey = block_kth_smallest(s_keys, cnt, k, s_bins, s_scal);
    const float H = hkey == 0xFFFFFFFFu ? __builtin_inff() : unord_f32(hkey);
    // a guessed bound U holds when k candidates are at most U away (batched_guess); if not, the query is handed back
    if (guess != nullptr && !(H <= guess[(size_t)q * ks + (ks - 1)])) { if (t == 0) { overflow[q] = 1; n_surv[q] = 0; } return; }
    for (uint32_t i0 = 0; i0 < cnt; i0 += 1024) {
        const uint32_t i = i0 + t;
        const bool take = i < cnt && lo_out[(size_t)q * cap + i] <= H;
        const uint64_t m = __ballot(take);
        if (m == 0) continue;
        uint32_t base = 0;
        if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&s_n, (uint32_t)__builtin_popcountll(m));
        base = __builtin_amdgcn_readlane(base, (int)__builtin_ctzll(m));
        if (take) surv[(size_t)q * cap + base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = cand_rows[(size_t)q * cap + i];
    }
    __syncthreads();
    if (t == 0) n_surv[q] = s_n;
}

// survivors: candidates whose lower bound does not exceed H_q = the k-th smallest upper bound (sel_dist[q][k - 1]; +inf = keep all)
__global__ void __launch_bounds__(256)
k_cand_survive(const uint32_t* __restrict__ cand_rows, const uint32_t* __restrict__ cand_cnt, uint32_t cap, const float* __restrict__ lo_in,
               const float* __restrict__ sel_dist, uint32_t k, uint32_t* __restrict__ surv, uint32_t* __restrict__ n_surv,
               const float* __restrict__ guess /* [nq][ks] or null */, uint32_t ks, uint32_t* __restrict__ overflow) {
    const uint32_t q = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x, lane = lane_id();
    const uint32_t cnt = cand_cnt[q] <= cap ? cand_cnt[q] : 0u;
    const float H = sel_dist[(size_t)q * k + (k - 1)];
    // a guessed bound U holds when k candidates are at most U away (batched_guess); if not, the query is handed back
    if (guess != nullptr && !(H <= guess[(size_t)q * ks + (ks - 1)])) { if (i == 0) overflow[q] = 1; return; }
    const bool take = i < cnt && lo_in[(size_t)q * cap + i] <= H;
    const uint64_t m = __ballot(take);
    if (m == 0) return;
    uint32_t base = 0;
    if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&n_surv[q], (uint32_t)__builtin_popcountll(m));
    base = __builtin_amdgcn_readlane(base, (int)__builtin_ctzll(m));
    if (take) surv[(size_t)q * cap + base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = cand_rows[(size_t)q * cap + i];
}

#ifdef QV_VARIANTS
// exact distances of the survivors: the scan's arithmetic, a lane per row (its chunks are a gather: requests pinned per block);
// keys (distance, row), dead beyond the survivors.
template <int M, int U>
__global__ void __launch_bounds__(256)
k_cand_exact(IndexView v, const float* __restrict__ queries, const uint32_t* __restrict__ surv, const uint32_t* __restrict__ n_surv, uint32_t cap,
             const double* __restrict__ qnorms, uint64_t* __restrict__ keys_ex) {
    using Q = typename MT<M>::Q;
    extern __shared__ __align__(16) unsigned char smem[];
    Q* q_lds = reinterpret_cast<Q*>(smem);
    // grid (nq, cap / 256): the query in x.  Only the first slices of a query have work, and workgroups go to the XCDs round-robin
    // by linear id: with the slice in x the working workgroups (ids 16 q) all landed on ONE XCD — 1.5 ms for 56 k rows
    const uint32_t q = blockIdx.x, i = blockIdx.y * blockDim.x + threadIdx.x;
    const uint32_t ns = n_surv[q];
#ifdef QV_LK_PROF
    if (i == 0 && (q == 0 || q == 100)) printf("large-k re-score q%u: survivors %u of cap %u\n", q, ns, cap);
#endif
    if (blockIdx.y * blockDim.x >= ns) { if (i < cap) keys_ex[(size_t)q * cap + i] = kDeadKey; return; }
    stage_query<M>(q_lds, queries + (size_t)q * v.dim, v.dim, v.dim4);
    __syncthreads();
    uint64_t key = kDeadKey;
    if (i < ns) {
        const uint32_t row = surv[(size_t)q * cap + i];
        QConst qc; qc.qn = 0.0; qc.qn32 = 0.0f;
        if constexpr (M == QV_COSINE) qc.qn = qnorms[2 * q + 1];
        typename MT<M>::A acc;
        if (v.rowmaj != nullptr && (v.dim & 3) == 0) acc = row_accumulate<M, U, false, true>(reinterpret_cast<const f4*>(v.rowmaj + (size_t)row * v.dim), 1, q_lds, v.dim4);
        else acc = row_accumulate<M, U, false, true>(reinterpret_cast<const f4*>(v.tiles) + (size_t)(row >> 6) * v.dim4 * 64 + (row & 63), 64, q_lds, v.dim4);
        double rn = 0.0;
        if constexpr (MT<M>::needs_rnorm) rn = v.rnorm[row];
        key = make_key(finalize<M>(acc, qc, rn), row);
    }
    if (i < cap) keys_ex[(size_t)q * cap + i] = key;
}

#endif
// The same, a WAVE per 32 survivors of one query — the traversal's row machinery (qv_hnsw.hip: hnsw_eval_hop_front) on a list of
// rows that is known up front.  A lane per row walking its 192 chunks asks for one line per step and waits for it (k_cand_exact:
// 2.0 TB/s of useful bytes from the row-major copy, 0.73 from the tiles at k = 1000); here the wave's 64 lanes request 128 bytes
// of each of the 32 rows per slab straight into LDS (global_load_lds, 16 bytes per lane, two 4 KiB buffers: slab s + 1 is in
// flight while lanes 0-31 walk slab s of their rows in element order), and the query stays in LDS as the caller's float32 words,
// a word per lane and slab, handed to the steps by v_readlane.  Rows come from the row-major copy when the index keeps one
// (a row's slab piece = 128 contiguous bytes = two 64-byte requests, all used) and from the tiles otherwise (a row's chunk c is 16
// bytes at tile + c KiB: one 64-byte request per 16 bytes, 4 x the bytes — the layout's price, paid here at the rate of a
// stream of independent requests instead of a chain of round trips).  Arithmetic: acc1 / finalize, element order, as everywhere.
typedef __attribute__((address_space(3))) unsigned char rs_lds_u8;
typedef __attribute__((address_space(3))) uint32_t rs_lds_u32;
constexpr uint32_t kRsSlabBytes = 32 * 8 * 16;
__device__ __forceinline__ void rs_glds16(const float* g, rs_lds_u8* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}
template <int M>
__global__ void __launch_bounds__(64)
k_cand_exact_wave(IndexView v, const float* __restrict__ queries, const uint32_t* __restrict__ surv, const uint32_t* __restrict__ n_surv, uint32_t cap,
                  const double* __restrict__ qnorms, uint64_t* __restrict__ keys_ex) {
    using Q = typename MT<M>::Q;
    typedef const __attribute__((address_space(3))) f4* lds_f4p;
    extern __shared__ __align__(16) unsigned char smem[];
    rs_lds_u8* slabs = (rs_lds_u8*)smem;                               // 2 x 4 KiB
    rs_lds_u32* qres = (rs_lds_u32*)(slabs + 2 * kRsSlabBytes);        // the query, float32 words, zero beyond dim: nslab x 32
    // grid (nq, cap / 32): the query in x, so that the few working groups of every query spread over the XCDs (see k_cand_exact)
    const uint32_t q = blockIdx.x, base = blockIdx.y * 32u, lane = threadIdx.x;
    const uint32_t ns = n_surv[q];
    if (base >= ns) { if (lane < 32 && base + lane < cap) keys_ex[(size_t)q * cap + base + lane] = kDeadKey; return; }
    const uint32_t n = ns - base < 32u ? ns - base : 32u;
    const uint32_t nslab = (v.dim4 + 7) >> 3;
    const bool me = lane < n;
    // lanes beyond the count take the group's first row: valid memory, LDS slots nobody reads
    const uint32_t myrow = surv[(size_t)q * cap + base + (me ? lane : 0u)];
    double rn = 0.0;
    if constexpr (MT<M>::needs_rnorm) rn = v.rnorm[myrow];
    for (uint32_t i = lane; i < nslab * 32u; i += 64u) qres[i] = i < v.dim ? __float_as_uint(queries[(size_t)q * v.dim + i]) : 0u;
    // this lane's requests: row (8 g + lane / 8) of group g, slot lane % 8; the slot's chunk is swizzled by the row so that the
    // walkers' 16-byte reads (32 lanes, rows 128 bytes apart) spread over the banks
    const uint32_t drow = lane >> 3, dslot = lane & 7u, ng = (n + 7u) >> 3;
    const bool rm = v.rowmaj != nullptr && (v.dim & 3u) == 0;
    const size_t cstride = rm ? 4 : 256;                                // floats between two chunks of a row
    const float* src[4]; uint32_t sw[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const uint32_t r = (uint32_t)__shfl((int)myrow, (int)((uint32_t)g * 8u + drow));
        sw[g] = dslot ^ drow ^ ((uint32_t)g & 1u);
        src[g] = (rm ? v.rowmaj + (size_t)r * v.dim : v.tiles + ((size_t)(r >> 6) * v.dim4 * 64 + (r & 63u)) * 4) + (size_t)sw[g] * cstride;
    }
    auto issue = [&](uint32_t sl) {
        rs_lds_u8* buf = slabs + (sl & 1u) * kRsSlabBytes;
        const uint32_t c0 = sl * 8u;
        if (c0 + 8u <= v.dim4) {
#pragma unroll
            for (int g = 0; g < 4; g++) if ((uint32_t)g < ng) rs_glds16(src[g] + (size_t)c0 * cstride, buf + g * 1024);
        } else {
#pragma unroll
            for (int g = 0; g < 4; g++) if ((uint32_t)g < ng && c0 + sw[g] < v.dim4) rs_glds16(src[g] + (size_t)c0 * cstride, buf + g * 1024);
        }
    };
    issue(0);
    typename MT<M>::A acc = 0;
    const uint32_t mg = lane >> 3, mr = lane & 7u, msw = mr ^ (mg & 1u);   // (as a walker: lane = row of the group, lanes 0..31)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // the query words are in LDS (a single wave: no barrier)
    uint32_t qw = qres[lane & 31u];
    for (uint32_t sl = 0; sl < nslab; sl++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // slab sl has landed
        uint32_t qn = 0;
        if (sl + 1 < nslab) { issue(sl + 1); qn = qres[(sl + 1) * 32u + (lane & 31u)]; }
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t nc = v.dim4 - sl * 8u < 8u ? v.dim4 - sl * 8u : 8u;
        if (me) {
            const rs_lds_u8* mine = slabs + (sl & 1u) * kRsSlabBytes + (mg & 3u) * 1024 + mr * 128;
            if (nc == 8u) {
                f4 x[2];
                x[0] = *(lds_f4p)(mine + ((0u ^ msw) << 4));
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const int cur = c & 1, nxt = cur ^ 1;
                    if (c + 1 < 8) x[nxt] = *(lds_f4p)(mine + (((uint32_t)(c + 1) ^ msw) << 4));
                    const Q q0 = (Q)__uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)qw, 4 * c)), q1 = (Q)__uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)qw, 4 * c + 1));
                    const Q q2 = (Q)__uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)qw, 4 * c + 2)), q3 = (Q)__uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)qw, 4 * c + 3));
                    acc1<M>(acc, q0, x[cur].x); acc1<M>(acc, q1, x[cur].y); acc1<M>(acc, q2, x[cur].z); acc1<M>(acc, q3, x[cur].w);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                for (uint32_t c = 0; c < nc; c++) {                       // the last slab of a row whose chunk count is no multiple of 8
                    const f4 x = *(lds_f4p)(mine + ((c ^ msw) << 4));
                    const int cl = __builtin_amdgcn_readfirstlane((int)(4 * c));      // (v_readlane ignores the exec mask: lanes beyond the count still hold the query)
                    const Q q0 = (Q)__uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)qw, cl)), q1 = (Q)__uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)qw, cl + 1));
                    const Q q2 = (Q)__uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)qw, cl + 2)), q3 = (Q)__uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)qw, cl + 3));
                    acc1<M>(acc, q0, x.x); acc1<M>(acc, q1, x.y); acc1<M>(acc, q2, x.z); acc1<M>(acc, q3, x.w);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // this slab's buffer is read before it is refilled
        qw = qn;
    }
    QConst qc; qc.qn = 0.0; qc.qn32 = 0.0f;
    if constexpr (M == QV_COSINE) qc.qn = qnorms[2 * q + 1];
    if (lane < 32 && base + lane < cap) keys_ex[(size_t)q * cap + base + lane] = me ? make_key(finalize<M>(acc, qc, rn), myrow) : kDeadKey;
}

// Many survivors and no row-major copy: a pass over the TILES instead of a gather.  From the tiles a survivor's row costs a 64-byte
// request per 16-byte chunk — 12 KiB for a 3 KiB row — and at k = 1000 the 540 000 survivors of 256 queries over a million rows
// (35 per 64-row tile) ask for more than twice the corpus: 1.7 ms.  Here the (query, row) pairs are sorted by tile (count, scan,
// scatter: three small kernels), and a wave per tile streams the tile through LDS slab by slab — contiguous KiBs, every byte used,
// tiles without a survivor not touched — while each lane walks ITS pair: the row's chunk from LDS, the query's from L1/L2 (a
// lane's own 16 bytes per chunk; consecutive chunks share a line).  A tile with more than 64 pairs is streamed once more per 64.
// Same chain per pair as everywhere (acc1 / finalize in element order), keys into the pairs' own slots.
__global__ void __launch_bounds__(256)
k_tp_count(const uint32_t* __restrict__ surv, const uint32_t* __restrict__ n_surv, uint32_t cap, uint32_t* __restrict__ tile_cnt, uint64_t* __restrict__ keys_ex) {
    const uint32_t q = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    if (i < n_surv[q]) atomicAdd(&tile_cnt[surv[(size_t)q * cap + i] >> 6], 1u);
    else keys_ex[(size_t)q * cap + i] = kDeadKey;
}
// exclusive scan of the tiles' counts (one workgroup: 15 625 tiles per million rows); the counts become zero: the scatter's cursors
__global__ void __launch_bounds__(1024)
k_tp_scan(uint32_t* __restrict__ tile_cnt, uint32_t n_tiles, uint32_t* __restrict__ tile_off) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_tiles; base += 4096) {
        const uint32_t t0 = base + threadIdx.x * 4;
        uint32_t c[4], mine = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) { c[u] = t0 + u < n_tiles ? tile_cnt[t0 + u] : 0u; mine += c[u]; }
        const uint32_t inc = wave_incl_scan(mine, lane);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t before = carry_s;
        for (uint32_t x = 0; x < wave; x++) before += wsum[x];
        uint32_t run = before + inc - mine;
#pragma unroll
        for (int u = 0; u < 4; u++) if (t0 + u < n_tiles) { tile_off[t0 + u] = run; tile_cnt[t0 + u] = 0u; run += c[u]; }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = run;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_off[n_tiles] = carry_s;
}
__global__ void __launch_bounds__(256)
k_tp_scatter(const uint32_t* __restrict__ surv, const uint32_t* __restrict__ n_surv, uint32_t cap, const uint32_t* __restrict__ tile_off,
             uint32_t* __restrict__ tile_cnt, uint2* __restrict__ pairs) {
    const uint32_t q = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap || i >= n_surv[q]) return;
    const uint32_t row = surv[(size_t)q * cap + i], t = row >> 6;
    pairs[tile_off[t] + atomicAdd(&tile_cnt[t], 1u)] = make_uint2(q * cap + i, row);      // (nq x cap < 2^32: the launcher's condition)
}
// SL chunks per slab: the tile's (SL KiB: 64 rows x SL x 16 bytes, a KiB per request instruction) and the 64 pairs' queries' (SL KiB:
// 64 x SL x 16 bytes, SL lanes per query and instruction = 16 SL bytes contiguous per query), both by global_load_lds, both double-
// buffered.  (A first form read the queries with a 16-byte load per lane and chunk: 64 different lines per instruction, 300 of its
// 876 us at k = 1000.)  The queries' pieces are swizzled by the pair so that the walkers' 16-byte reads spread over the banks.
template <int M, int SL>
__global__ void __launch_bounds__(64)
k_tp_exact(IndexView v, const float* __restrict__ queries, const uint32_t* __restrict__ tile_off, const uint2* __restrict__ pairs, uint32_t cap,
           const double* __restrict__ qnorms, uint64_t* __restrict__ keys_ex) {
    using Q = typename MT<M>::Q;
    typedef const __attribute__((address_space(3))) f4* lds_f4p;
    constexpr uint32_t kSlab = SL * 1024, kPer = 64 / SL;              // bytes of a slab; pairs per query-request instruction
    extern __shared__ __align__(16) unsigned char smem[];
    rs_lds_u8* tbuf = (rs_lds_u8*)smem;                                // 2 x SL KiB: the tile
    rs_lds_u8* qbuf = tbuf + 2 * kSlab;                                // 2 x SL KiB: the pairs' queries
    const uint32_t t = blockIdx.x, lane = threadIdx.x;
    const uint32_t p_lo = tile_off[t], p_hi = tile_off[t + 1];
    if (p_lo == p_hi) return;
    const float* tile = v.tiles + (size_t)t * v.dim4 * 256 + lane * 4;  // this lane's 16 bytes of every chunk's KiB
    const uint32_t nslab = (v.dim4 + SL - 1) / SL;
    const uint32_t dpair = lane / SL, dslot = lane % SL;               // as a requester of query pieces: pair (j kPer + dpair) in instruction j
    auto swz = [](uint32_t p) { return SL == 8 ? (p >> 1) & 7u : (p >> 2) & 3u; };
    for (uint32_t p0 = p_lo; p0 < p_hi; p0 += 64) {
        const bool me = p0 + lane < p_hi;
        const uint2 pr = pairs[me ? p0 + lane : p_lo];                 // lanes beyond the count: the tile's first pair (valid memory, results dropped)
        const uint32_t q = pr.x / cap, r = pr.y & 63u;
        double rn = 0.0;
        if constexpr (MT<M>::needs_rnorm) rn = v.rnorm[pr.y];
        QConst qc; qc.qn = 0.0; qc.qn32 = 0.0f;
        if constexpr (M == QV_COSINE) qc.qn = qnorms[2 * q + 1];
        uint32_t qoff[SL], qsw[SL];                                    // float offsets of this lane's piece per instruction (nq x dim < 2^32)
#pragma unroll
        for (uint32_t j = 0; j < (uint32_t)SL; j++) {
            const uint32_t pj = j * kPer + dpair;
            qsw[j] = dslot ^ swz(pj);
            qoff[j] = (uint32_t)__shfl((int)q, (int)pj) * v.dim + qsw[j] * 4u;
        }
        auto issue = [&](uint32_t sl) {
            rs_lds_u8* tb = tbuf + (sl & 1u) * kSlab;
            rs_lds_u8* qb = qbuf + (sl & 1u) * kSlab;
            const uint32_t c0 = sl * SL;
            if (c0 + SL <= v.dim4) {
#pragma unroll
                for (uint32_t c = 0; c < (uint32_t)SL; c++) rs_glds16(tile + (size_t)(c0 + c) * 256, tb + c * 1024);
#pragma unroll
                for (uint32_t j = 0; j < (uint32_t)SL; j++) rs_glds16(queries + qoff[j] + (size_t)c0 * 4, qb + j * 1024);
            } else {
#pragma unroll
                for (uint32_t c = 0; c < (uint32_t)SL; c++) if (c0 + c < v.dim4) rs_glds16(tile + (size_t)(c0 + c) * 256, tb + c * 1024);
#pragma unroll
                for (uint32_t j = 0; j < (uint32_t)SL; j++) if (c0 + qsw[j] < v.dim4) rs_glds16(queries + qoff[j] + (size_t)c0 * 4, qb + j * 1024);
            }
        };
        typename MT<M>::A acc = 0;
        const uint32_t msw = swz(lane);
        issue(0);
        for (uint32_t sl = 0; sl < nslab; sl++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // slab sl has landed
            if (sl + 1 < nslab) issue(sl + 1);
            __builtin_amdgcn_sched_barrier(0);
            const rs_lds_u8* tm = tbuf + (sl & 1u) * kSlab + r * 16;
            const rs_lds_u8* qm = qbuf + (sl & 1u) * kSlab + lane * (SL * 16);
            const uint32_t nc = v.dim4 - sl * SL < (uint32_t)SL ? v.dim4 - sl * SL : (uint32_t)SL;
#pragma unroll
            for (uint32_t c = 0; c < (uint32_t)SL; c++) {
                if (c < nc) {
                    const f4 x = *(lds_f4p)(tm + c * 1024);
                    const f4 y = *(lds_f4p)(qm + ((c ^ msw) << 4));
                    acc1<M>(acc, (Q)y.x, x.x); acc1<M>(acc, (Q)y.y, x.y); acc1<M>(acc, (Q)y.z, x.z); acc1<M>(acc, (Q)y.w, x.w);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // both buffers of this slab are read before they are refilled
        }
        if (me) keys_ex[pr.x] = make_key(finalize<M>(acc, qc, rn), pr.y);
    }
}

hipError_t launch_select_topk_counted(const uint64_t* d_keys, size_t stride, uint32_t n, const uint32_t* d_counts, uint32_t nq, uint32_t kk, uint32_t k_stride,
                                      uint32_t* d_rows_out, float* d_dist_out, hipStream_t s);   // qv_select.hip
// 1 = fp32 MFMA chain (BASELINE configs[2] as written), 2 = bfloat16 x 3, 3 = bfloat16 x 1; QV_MFMA_FILTER unset: the one-term
// filter up to 1536 dimensions, three terms above.  Its window (2 x 7.9e-3 |q||r|) is measured in units of the scores' spread,
// which shrinks like 1/sqrt(dim) on unstructured data: 256 queries x 3 GB of rows take 1.93 / 1.14 / 1.08 / 1.39 / 1.34 / 2.55 ms at
// 128 / 384 / 768 / 1024 / 1536 / 3072 dimensions against 2.65 / 1.69 / 1.63 / 2.00 / 1.56 / 1.73 ms with three terms.
// The choice is the index's (qv_index_set_filter, carried in IndexView::filter); QV_MFMA_FILTER, read ONCE per process, only
// sets the default of indexes that never chose.
int filter_mode(const IndexView& v) {
    static const int env_default = env_int("QV_MFMA_FILTER", 0);
    const int m = v.filter >= 1 && v.filter <= 3 ? v.filter : env_default;
    return m == 1 ? 1 : (m == 2 ? 2 : (m == 3 ? 3 : (v.dim <= 1536 ? 3 : 2)));
}
// From how many results per query the batch takes the selection path (a guessed bound, k_cand_narrow, a wave per 32 survivors or the tile
// pass, k_select_sort) instead of the 64-key wave lists (the sample's exact k-th bound, k_rescore_select).  Above 64 there is no choice.
// Below, round 6's selection path wins from k = 16 wherever its bound can be a guess (256 x 1M x 768, same box: k = 16 0.670 against
// 0.698 ms, 32: 0.728 / 0.769, 64: 0.806 / 0.916; k = 10: 0.677 / 0.670): the wave lists' work grows with k — a sample of k N / 384
// rows, k dependent inserts per list — and the selections' does not.  QV_BATCHED_SELECT_FROM (measurement build) moves the switch.
constexpr uint32_t kGuessSampleRows = 32768;
static bool guess_eligible(const IndexView& v) { return v.n_rows >= 4 * kGuessSampleRows; }
static bool batched_large_k(const IndexView& v, uint32_t k) {
    static const int from = dev_env_int("QV_BATCHED_SELECT_FROM", 16);
    if (k > (uint32_t)kMaxFusedK) return true;
    if (!guess_eligible(v)) return false;
    if ((int)k >= from) return true;
    // a corpus so large that the wave lists' sample (k N / 384 rows: 45 M rows up at k = 15) outgrows k_sample_select's LDS: the guess's small sample
    const uint64_t per = filter_mode(v) == 3 ? 384 : 1536;
    const uint64_t want = ((uint64_t)v.n_rows * std::max(k, 1u) / per + 8191) / 8192 * 8192;
    return !sample_select_applies((uint32_t)std::min<uint64_t>(v.n_rows, want), k);
}
// ---- MFMA batched path --------------------------------------------------------------
// Rows of the exact sample scan that bounds each query's k-th distance.  A sample of S of N rows lets about k*N/S rows through
// the filter per query; the candidate buffer holds kMfmaCandCap (4096), so S grows with N and k to keep that near 1536
// (10M rows or k = 64 with the former fixed 8192 overflowed nearly every query into the exact redo: 256 x 10M x 768 took 108 ms).
// More than 64 results per query over a large corpus: the bound is a GUESS, verified afterwards.  The k-th smallest bound of a sample is
// a valid bound whatever the sample, but a tight one needs a sample of about N / 2 at k = 1000 — half the filter's work again and two
// histogram passes over 128 M bounds (0.86 of 2.26 ms at 256 x 1M x 768).  Instead: 32 768 rows or more (batched_sample_rows) spread over the corpus, and the bound
// at the rank the k-th row is expected at in it, r = k S / N, plus four standard deviations of that count (+ 2).  Nothing guarantees
// that k rows of the corpus lie under it — so the batch checks: H, the k-th smallest UPPER bound among the rows the filter let through,
// must not exceed the guess U (k_cand_survive).  Then k rows are at most H away, every row the filter dropped is farther than U >= H,
// and the answer is among the candidates; otherwise the query is handed back (redo flag) as an overflowing one is.
static bool batched_guess(const IndexView& v, uint32_t k) {
    static const int on = dev_env_int("QV_LK_GUESS", 1);
    return on == 1 && batched_large_k(v, k) && guess_eligible(v);
}
static uint32_t guess_rank(uint32_t n_rows, uint32_t srows, uint32_t k) {
    const double r = (double)k * srows / (double)n_rows;
    return std::min<uint32_t>((uint32_t)std::ceil(r + 4.0 * std::sqrt(r) + 2.0), srows);
}
uint32_t batched_sample_rows(const IndexView& v, uint32_t k);
uint32_t batched_sample_rank(const IndexView& v, uint32_t k) { return batched_guess(v, k) ? guess_rank(v.n_rows, batched_sample_rows(v, k), k) : k; }
uint32_t batched_cand_cap(uint32_t k);
uint32_t batched_sample_rows(const IndexView& v, uint32_t k) {
    const uint32_t n_rows = v.n_rows;
    static const int forced = dev_env_int("QV_MFMA_SAMPLE_ROWS", 0);
    if (forced > 0) return std::min<uint32_t>(n_rows, (uint32_t)forced);
    // ~1536 expected candidates (3 sigma of the k-th order statistic at k = 10 stays under the 4096 slots), in whole multiples
    // of 8192 rows = 128 tiles: with 16 query groups that is one full round of the 2048 scan waves per multiple
    // the one-term bfloat16 filter's window is 2 x 7.9e-3 |q||r| wide — 0.44 sigma of the scores of unstructured 768-d data, which
    // lets ~4.6 x as many rows through at the same bound: four times the sample keeps the candidate count where it was
    const bool one = filter_mode(v) == 3;
    // beyond 64 results per query the candidate slots grow with k (batched_cand_cap): half the sample, twice the candidates
    const uint64_t per = (one ? 384 : 1536) * (batched_large_k(v, k) ? 2 : 1);
    const uint64_t want = ((uint64_t)n_rows * std::max(k, 1u) / per + 8191) / 8192 * 8192;
    uint64_t cap_rows = n_rows;
    if (batched_large_k(v, k)) cap_rows = std::max<uint64_t>(32768, ((uint64_t)n_rows / 2 + 8191) / 8192 * 8192);   // the sample costs a pass over its rows: half the corpus at most
    const uint32_t full = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(n_rows, cap_rows), std::max<uint64_t>(one ? 32768 : 8192, want));
    if (!batched_guess(v, k)) return full;
    // a guessed bound: the smallest sample whose rank leaves the expected candidates (rows under the bound x the filter's window: ~4.6 x
    // with one term) within the slots with four standard deviations to spare — 32 768 rows at 1M x k = 16 .. 64 and 1000 .. 4096, 65 536 at k = 100, 524 288 at 10M x k = 100 — and never more than the rule above
    const double f = one ? 4.6 : 1.2;
    uint32_t srows = kGuessSampleRows;
    auto fits = [&](uint32_t sr) {                                      // rows under the bound ~ Gamma(rank): mean rank N / S, four of its standard deviations above
        const double rank = guess_rank(n_rows, sr, k);
        return rank * ((double)n_rows / sr) * f * (1.0 + 4.0 / std::sqrt(rank)) <= 0.9 * batched_cand_cap(k);
    };
    while (srows < full && !fits(srows)) srows *= 2;
    return std::min(srows, full);
}
bool batched_supported(const IndexView& v, uint32_t nq, uint32_t k) {
    // Measured crossover against the exact multi-query scans (256 queries x 768 dims, host pointers for the filter): 12k-16k
    // rows 0.49-0.50 vs 0.36-0.40 ms, 32k 0.51 vs 0.67, 64k 0.64 vs 1.34, 128k 0.85 vs 2.44, 200k 1.19 vs 3.16 — the filter's
    // fixed cost (sample scan, prep, re-score: ~0.4 ms) pays off from about 8M query-rows.
    // with the bfloat16 filter a batch of 9..31 queries over 1M x 768 takes 0.70-0.76 ms against 0.96-1.2 ms for the exact f64-matrix scan
    // (8 queries or fewer share one HBM-bound pass of k_flat_scan_mq: 0.45 ms); the fp32 filter pays off from 32 queries
    static const int min_rows = env_int("QV_MFMA_MIN_ROWS", 32768);
    if (v.filter == QV_FILTER_OFF) return false;                      // the index's choice (qv_index_set_filter): exact scans only
    const int min_q = filter_mode(v) >= 2 ? 9 : 32;
    // millions of query-rows.  Round 2's one-term filter moved the crossover down: 16 / 64 queries x 100k x 768 take 0.17 / 0.18 ms here against
    // 0.30 / 0.57 ms on the exact multi-query scan, x 400k rows 0.36 / 0.37 against 0.59 / 1.65 (8 M query-rows was the three-term crossover)
    static const int min_work_m = dev_env_int("QV_MFMA_MIN_MROWS", 1);
    return (v.metric == QV_COSINE || v.metric == QV_DOT || v.metric == QV_L2 || v.metric == QV_L2SQ) && k <= (uint32_t)kMaxBatchedK && nq >= (uint32_t)min_q &&
           v.n_rows >= (uint32_t)min_rows && (uint64_t)nq * v.n_rows >= (uint64_t)min_work_m * 1000000ull;
}
// candidate slots per query: kMfmaCandCap up to 256 results; beyond, 16 k rounded up to a power of two (the expected candidates grow
// with k: about (1 + 4 / sqrt(rank)) f k under a guessed bound, 2 f k under half the corpus's, f ~ 4.6 for the one-term filter)
uint32_t batched_cand_cap(uint32_t k) {
    uint32_t c = (uint32_t)kMfmaCandCap;
    while (c < 16u * k) c <<= 1;
    return c;
}
// queries are padded (zero vector, +inf threshold: nothing passes) to 1, 2 or a multiple of 4 blocks of 64: the four waves of a
// workgroup then work on the same row group (its tiles are fetched once and hit L1/L2 for the other three) and the wave count
// divides evenly over the blocks with one workgroup per CU.  Measured, 1M x 768: 160-192 queries as 3 blocks took 4.8 ms
// (258 workgroups on 256 CUs: a second round), as 4 blocks 3.4 ms.
// With the one-term filter 65-128 queries as two blocks took 1.24-1.33 ms on the per-wave kernel against 1.0 ms for 256 on the shared one:
// there everything above one block is padded to whole workgroups of four.
static uint32_t batched_nq_pad(uint32_t nq, const IndexView& v) { return nq <= 64 ? 64u : (nq <= 128 && filter_mode(v) != 3 ? 128u : (nq + 255) / 256 * 256); }

size_t batched_workspace_bytes(const IndexView& v, const ScanPlan& p, uint32_t nq, uint32_t k) {
    const uint32_t nq_pad = batched_nq_pad(nq, v);
    size_t b = scan_workspace_bytes(p, nq, k) + (size_t)(nq + 16) * v.dim4 * 4 * sizeof(double);   // sample scan (partials + query blocks)
    b = (b + 255) / 256 * 256;
    b += (size_t)nq_pad * (v.dim4 + 36) * 16 + 1024;         // Qt (chunk count padded to even) / the bf16 hi + lo planes (padded to whole steps; up to eight steps at least for the padded eight-wave kernel)
    const size_t ccap = batched_cand_cap(k);
    b += (size_t)nq_pad * 20;                                // cq, mq (m_q and b_q), eq
    b += (size_t)nq * ccap * 8;                              // candidates: rows + fp32 scores
    b += (size_t)nq * 8 + 256;                               // capacity word + counters, overflow flags
    if (batched_large_k(v, k))                               // keys of the upper bounds and of the exact distances, lower bounds, survivors, counters, norms, selection
        b += (size_t)nq * ccap * (8 + 8 + 4 + 4) + (size_t)nq * (4 + 16) + 1024 + select_workspace_bytes(nq, k) + (size_t)(v.n_tiles + 2) * 8 + 512;
    b += (size_t)nq * k * 8;                                 // sample rows/dist
    b += (size_t)nq * k * 16 + 256;                          // k_sample_bound's partial lists (up to four parts per query)
    b += (size_t)nq_pad * batched_sample_rows(v, k) * 4 + (size_t)nq_pad * (batched_sample_rows(v, k) / 128 + 2) * 4 + 256;   // the sample's scores (bfloat16 filter: bound without an exact scan) and per-group minima
    return b + 1024;
}

// the largest multiple of `unit` that fits `want` workgroups (one workgroup per CU is resident: rounding UP past the CU count
// would leave a second, nearly empty round — 3 831 queries = 60 query blocks took 36 ms against 21 ms for 4 096)
static uint32_t grid_multiple(uint32_t want, uint32_t unit) { const uint32_t g = want / unit * unit; return g ? g : unit; }

hipError_t launch_batched(const IndexView& v, const ScanPlan& p, const float* d_queries, uint32_t nq, uint32_t k, void* d_ws,
                          uint32_t* d_rows_out, float* d_dist_out, uint32_t** d_overflow_out, int cus, hipStream_t s,
                          hipEvent_t ev0, hipEvent_t ev1) {
    const uint32_t nq_pad = batched_nq_pad(nq, v);
    char* w = static_cast<char*>(d_ws);
    size_t off = scan_workspace_bytes(p, nq, k) + (size_t)(nq + 16) * v.dim4 * 4 * sizeof(double);
    off = (off + 255) / 256 * 256;
    float* Qt = reinterpret_cast<float*>(w + off); off += (size_t)nq_pad * (v.dim4 + 36) * 16 + 1024;
    float* cq = reinterpret_cast<float*>(w + off); off += (size_t)nq_pad * 4;
    float* mq = reinterpret_cast<float*>(w + off); off += (size_t)nq_pad * 8;      // m_q, b_q
    float* eq = reinterpret_cast<float*>(w + off); off += (size_t)nq_pad * 8;      // the scores' error bound per query (k_mfma_prep -> k_rescore_select)
    const uint32_t ccap = batched_cand_cap(k);
    const bool large_k = batched_large_k(v, k);
    uint32_t* cand = reinterpret_cast<uint32_t*>(w + off); off += (size_t)nq * ccap * 4;
    float* cscore = reinterpret_cast<float*>(w + off); off += (size_t)nq * ccap * 4;
    off = (off + 255) / 256 * 256;
    uint32_t* cnt = reinterpret_cast<uint32_t*>(w + off) + 1; off += (size_t)nq * 4 + 4;   // cnt[-1]: the capacity, for the filter kernels
    uint32_t* ovf = reinterpret_cast<uint32_t*>(w + off); off += (size_t)nq * 4;
    // large k: the selections' arrays
    uint64_t* keys_hi = nullptr; uint64_t* keys_ex = nullptr; float* lo_b = nullptr; uint32_t* surv = nullptr; uint32_t* nsurv = nullptr; double* qnorms = nullptr; void* sel_ws = nullptr; uint32_t* tp_cnt = nullptr; uint32_t* tp_off = nullptr;
    if (large_k) {
        off = (off + 255) / 256 * 256;
        keys_hi = reinterpret_cast<uint64_t*>(w + off); off += (size_t)nq * ccap * 8;
        keys_ex = reinterpret_cast<uint64_t*>(w + off); off += (size_t)nq * ccap * 8;
        lo_b = reinterpret_cast<float*>(w + off); off += (size_t)nq * ccap * 4;
        surv = reinterpret_cast<uint32_t*>(w + off); off += (size_t)nq * ccap * 4;
        qnorms = reinterpret_cast<double*>(w + off); off += (size_t)nq * 16;
        nsurv = reinterpret_cast<uint32_t*>(w + off); off += (size_t)nq * 4;
        off = (off + 255) / 256 * 256;
        sel_ws = w + off; off += select_workspace_bytes(nq, k);
        tp_cnt = reinterpret_cast<uint32_t*>(w + off); off += (size_t)(v.n_tiles + 2) * 4;
        tp_off = reinterpret_cast<uint32_t*>(w + off); off += (size_t)(v.n_tiles + 2) * 4;
        off = (off + 255) / 256 * 256;
    }
    uint32_t* srows = reinterpret_cast<uint32_t*>(w + off); off += (size_t)nq * k * 4;
    float* sdist = reinterpret_cast<float*>(w + off); off += (size_t)nq * k * 4;
    off = (off + 255) / 256 * 256;
    float* sparts = reinterpret_cast<float*>(w + off); off += (size_t)nq * k * 16;
    const uint32_t bparts = nq <= 64 ? 4u : (nq <= 128 ? 2u : 1u);       // (four parts at 256 queries: 44 -> 74 us, every workgroup stages its query again)
    // a guessed bound (batched_guess): rank ks of the small sample, kept in ubuf for the check after the filter (sdist is overwritten by the first selection)
    const bool guess = batched_guess(v, k);
    const uint32_t ks = batched_sample_rank(v, k);
    float* ubuf = guess ? sparts : sdist;
    // 1. per-query upper bound U_q of the k-th distance from a sample (the first rows)
    IndexView vs = v;
    vs.n_rows = batched_sample_rows(v, k);
    vs.n_tiles = (vs.n_rows + 63) / 64;
    // QV_MFMA_FILTER: 1 = fp32 MFMA (BASELINE configs[2] as written), 2 = bf16 x 3 (default: same candidates up to the margin,
    // a quarter of the matrix cycles); the index's choice (qv_index_set_filter), so one process can compare them
    const int fmode = filter_mode(v);
    const int bf = fmode >= 2 ? 1 : 0;                                 // bfloat16 operand layout
    static const int sample_gemm = dev_env_int("QV_MFMA_SAMPLE_GEMM", 1);
    static const int share_env = dev_env_int("QV_MFMA_SHARE_ROWS", 1);
    const bool shared = bf && (nq_pad / 64) % 4 == 0 && share_env == 1;
    static const int q64_env = dev_env_int("QV_MFMA_Q64", 1);                                     // 2 = off
    const uint32_t fsteps0 = (v.dim4 + 3) / 4;
    const bool q64_shape = bf && fmode == 3 && nq_pad == 64 && q64_env == 1 && (v.dim4 & 3u) == 0 && fsteps0 % 4 == 0 && fsteps0 >= 8 && fsteps0 <= 64;
    const bool q64f = q64_shape && v.bf16 == nullptr;                     // the one-block kernel on float32 rows (round 4)
    const bool q64 = q64_shape;
    const int gmode = !bf ? 0 : (fmode == 3 && (shared || q64) ? 2 : 1);   // which filter_gamma the main pass obeys (one term: the shared kernels and the one-block kernel)
    // one-term filter on float32 rows at a dimension the eight-wave kernel's loop does not divide: its zero-padded form (round 4);
    // the query operands are then laid out once more, padded, after the sample pass has read them in its own layout
    static const int w8_env0 = dev_env_int("QV_MFMA_W8", 1);
    const bool w8_exact = (v.dim4 & 3u) == 0 && fsteps0 % 8 == 0 && fsteps0 >= 16;
    const bool pad_main = gmode == 2 && shared && !q64 && w8_env0 == 1 && !w8_exact;
    hipError_t e = hipSuccess;
    if (sample_gemm == 1 || large_k) {
        // the sample's scores by the three-term bfloat16 kernel, their upper bounds' k-th smallest as U_q (k_sample_bound): 0.1 ms
        // against 0.24 for an exact scan of the sample.  The fp32-MFMA filter takes its bound the same way (round 3): the query
        // operands are laid out as bfloat16 for the sample and then, in the same buffer, as float32 for the main pass
        off = (off + 255) / 256 * 256;
        float* sscore = reinterpret_cast<float*>(w + off); off += (size_t)nq_pad * vs.n_rows * 4 + (size_t)nq_pad * ((vs.n_rows + 127) / 128) * 4;   // every sample row's bound, the group minima behind them
        hipLaunchKernelGGL(k_mfma_prep, dim3(nq_pad), dim3(64), 0, s, d_queries, nq, nq_pad, v.dim, v.dim4, sdist, 1u, k, v.metric, Qt, cq, mq, eq, cnt, ovf, 1, 1, ccap, 0u);
        const uint32_t nqb64s = nq_pad / 64;
        const uint32_t gs = grid_multiple(std::min<uint32_t>((uint32_t)cus, ((vs.n_tiles + 1) / 2 * nqb64s + 3) / 4), nqb64s / std::gcd(nqb64s, 4u));
        const uint4* Qbf = reinterpret_cast<const uint4*>(Qt);
        // the sample's row groups are spread evenly over the corpus (a corpus stored cluster by cluster: the first S rows would bound nothing)
        const uint32_t sgroups = (vs.n_rows + 127) / 128, all_groups = v.n_tiles / 2;
        const uint32_t gstep = sgroups && all_groups > sgroups ? all_groups / sgroups : 1u;
        // one-term path: the sample on the eight-wave one-term kernel (QV_MFMA_SAMPLE1=2: on the three-term kernel, as the other filters' samples)
        static const int sample1_env = dev_env_int("QV_MFMA_SAMPLE1", 1);
        static const int w8s_env = dev_env_int("QV_MFMA_W8", 1);
        const bool sample1 = sample1_env == 1 && w8s_env == 1 && gmode == 2 && shared && (v.dim4 & 3u) == 0 && fsteps0 % 8 == 0 && fsteps0 >= 16;
        const uint32_t gs1 = grid_multiple(std::min<uint32_t>((uint32_t)cus, sgroups * (nq_pad >> 8)), std::max<uint32_t>(nq_pad >> 8, 1u));
        // large k: the k-th smallest bound by two histogram windows over the bounds (no wave list holds k keys), per-query state in the selection's workspace
        SelState* sst = nullptr; uint32_t* shist = nullptr;

        const uint32_t hgrid = std::max(1u, std::min(256u, (vs.n_rows + 16 * kSelBlock - 1) / (16 * kSelBlock)));
        // the eight-wave sample kernel hands out one value per query and 128-row group when the groups outnumber k at least four times
        const uint32_t sample_groups = (vs.n_rows + 127) / 128;
        static const int gmin_env = dev_env_int("QV_MFMA_SAMPLE_GROUP_MIN", 1);                      // 2 = every row's bound (round 3)
        // How the bound is taken from the sample's per-row upper bounds: k_sample_select (per-thread chunk minima, then the rows of the k
        // best chunks: the exact k-th smallest row bound, in LDS) while k chunks fit its LDS — k <= 128 at a million rows; beyond, the
        // histogram kernels (k_sample_hist: two radix windows over every row's bound) or, up to 64, k_sample_bound's wave lists.
        // QV_MFMA_SAMPLE_GROUP_MIN=3 (a measurement): one MINIMUM per query and 128-row group, selected inside k_mfma_prep — 11 us
        // faster at k = 10, 85 at k = 64, and a cliff on a corpus stored cluster by cluster (tools/dev_clustered_bound.py).
        const uint32_t gmin_mode = gmin_env == 3 && sample1 ? 1u : 0u;
        const bool group_min = gmin_mode != 0 && sample_groups >= 4 * k && sample_groups <= 4096;
        const uint32_t gmin_vals = sample_groups;
        static const int sel2_env = dev_env_int("QV_MFMA_SAMPLE_SELECT", 1);                        // 2 = k_sample_bound (wave lists) as in round 3
        const bool sel2 = sel2_env == 1 && !group_min && sample_select_applies(vs.n_rows, ks);
        const size_t sel2_lds = sel2 ? sample_select_lds_bytes(vs.n_rows, ks) : 0;
        if (large_k && !sel2 && !group_min) { e = select_prepare(sel_ws, nq, ks, &sst, &shist, s); if (e != hipSuccess) return e; }   // (the histogram kernels' state)
#ifdef QV_VARIANTS
#define QV_SB_LISTS(MMM) hipLaunchKernelGGL(k_sample_bound<MMM>, dim3(nq, bparts), dim3(1024), 0, s, sscore, vs.n_rows, k, (float)filter_gamma(v.dim, 0) * 1.000001f, bparts > 1 ? sparts : sdist, bparts);
#else
#define QV_SB_LISTS(MMM) return hipErrorNotSupported;     /* unreachable: k_sample_select applies to every sample of at least 4096 rows with k <= 64 */
#endif
#define QV_SB(MMM) { if (sample1) hipLaunchKernelGGL((k_bf16x1_filter_w8<MMM == QV_L2SQ ? QV_L2 : MMM, 8, 4, 1, false, false, true>), dim3(gs1), dim3(512), 0, s, v, Qbf, cq, eq, nq_pad, cand, cscore, cnt, sscore, vs.n_rows, gstep, group_min ? 1u : 0u); \
                     else hipLaunchKernelGGL(k_bf16x3_filter<MMM == QV_L2SQ ? QV_L2 : MMM>, dim3(gs), dim3(256), 0, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt, sscore, vs.n_rows, gstep); \
                     if (group_min) { /* the bound is selected inside k_mfma_prep from the group minima */ } \
                     else if (sel2) { e = set_lds(k_sample_select<MMM>, sel2_lds); if (e != hipSuccess) return e; \
                                      hipLaunchKernelGGL(k_sample_select<MMM>, dim3(nq), dim3(1024), sel2_lds, s, sscore, vs.n_rows, ks, (float)filter_gamma(v.dim, 0) * 1.000001f, ubuf); } \
                     else if (large_k) { hipLaunchKernelGGL(k_sample_hist<0>, dim3(hgrid, nq), dim3(kSelBlock), 0, s, sscore, vs.n_rows, ks, sst, shist); \
                                    hipLaunchKernelGGL(k_sample_hist<1>, dim3(hgrid, nq), dim3(kSelBlock), 0, s, sscore, vs.n_rows, ks, sst, shist); \
                                    hipLaunchKernelGGL(k_sample_bound_from_state<MMM>, dim3((nq + 255) / 256), dim3(256), 0, s, sst, nq, ks, (float)filter_gamma(v.dim, 0) * 1.000001f, ubuf); } \
                     else QV_SB_LISTS(MMM) }
        if (v.metric == QV_COSINE) QV_SB(QV_COSINE) else if (v.metric == QV_DOT) QV_SB(QV_DOT) else if (v.metric == QV_L2) QV_SB(QV_L2) else QV_SB(QV_L2SQ)
#undef QV_SB
#undef QV_SB_LISTS
        hipLaunchKernelGGL(k_mfma_prep, dim3(nq_pad), dim3(64), 0, s, d_queries, nq, nq_pad, v.dim, v.dim4, bparts > 1 && !large_k && !group_min && !sel2 ? sparts : ubuf, large_k || group_min || sel2 ? 1u : bparts, ks, v.metric, Qt, cq, mq, eq, cnt, ovf, gmode, pad_main ? 3 : (bf ? 2 : 3), ccap, pad_main ? w8x2_rounds(v.dim4) : 0u, group_min ? sscore : nullptr, gmin_vals);
    } else {
        ScanPlan ps = plan_scan(vs.n_tiles, cus);
        e = launch_flat_topk(vs, ps, d_queries, nq, k, d_ws, srows, sdist, s);
        if (e != hipSuccess) return e;
        // 2. query re-layout + filter constants
        hipLaunchKernelGGL(k_mfma_prep, dim3(nq_pad), dim3(64), 0, s, d_queries, nq, nq_pad, v.dim, v.dim4, sdist, 1u, k, v.metric, Qt, cq, mq, eq, cnt, ovf, gmode, 3, ccap, 0u);
    }
    // 3. MFMA filter
    const uint32_t nqb64 = nq_pad / 64;
    // one 4-wave workgroup per CU (512-register waves); every query block gets the same number of waves
    static const int f32_nj = dev_env_int("QV_MFMA_F32_NJ", 4);             // 2 (measurement build): 64 rows per group, two waves per SIMD — 4.25 against 3.06 ms
    const uint32_t grid = grid_multiple((uint32_t)cus * (!bf && !q64 && f32_nj == 2 ? 2 : 1), nqb64 / std::gcd(nqb64, 4u));
    if (ev0) (void)hipEventRecord(ev0, s);
    if (q64) {
        const uint4* Qbf = reinterpret_cast<const uint4*>(Qt);
        const size_t lds_a = (size_t)fsteps0 * 128 * sizeof(uint4);
#define QV_FQ1(MMM, NBB, RR) { e = set_lds(k_bf16rows_filter_q64<MMM, NBB, RR>, lds_a); if (e != hipSuccess) return e; \
                     hipLaunchKernelGGL((k_bf16rows_filter_q64<MMM, NBB, RR>), dim3((uint32_t)cus), dim3(512), lds_a, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt); }
        // (one tile per group with sixteen steps in flight, <MMM, 2, 16>, and a plane layout with contiguous KiB per request measure the same
        // 304-315 us at 64 x 1M x 768 as <MMM, 4, 4>: the kernel sits at 4.9-5.0 TB/s whatever each wave keeps in flight)
#define QV_FQ1F(MMM) { e = set_lds(k_bf16rows_filter_q64<MMM, 2, 4, true>, lds_a); if (e != hipSuccess) return e; \
                     hipLaunchKernelGGL((k_bf16rows_filter_q64<MMM, 2, 4, true>), dim3((uint32_t)cus), dim3(512), lds_a, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt); }
#define QV_FQ(MMM) { if (q64f) QV_FQ1F(MMM) else QV_FQ1(MMM, 4, 4) }
        if (v.metric == QV_COSINE) QV_FQ(QV_COSINE) else if (v.metric == QV_DOT) QV_FQ(QV_DOT) else QV_FQ(QV_L2)
#undef QV_FQ
#undef QV_FQ1
#undef QV_FQ1F
    } else if (shared) {
        const uint4* Qbf = reinterpret_cast<const uint4*>(Qt);
        const uint32_t gs = grid_multiple((uint32_t)cus, nqb64 / 4);      // every row group is walked by nqb64/4 workgroups
        const uint32_t fsteps = (v.dim4 + 3) / 4;
        static const int bfrows_env = dev_env_int("QV_MFMA_BF16_ROWS", 1);                        // 2 = ignore the index's bfloat16 plane
        const bool bfrows = gmode == 2 && v.bf16 != nullptr && bfrows_env == 1 && (v.dim4 & 3u) == 0 && fsteps % 8 == 0 && fsteps >= 16;
        static const int w8_env = dev_env_int("QV_MFMA_W8", 1);                                   // 2 = the four-wave kernel (k_bf16x3_filter_shared<., 1>)
        const bool w8 = w8_env == 1 && (v.dim4 & 3u) == 0 && fsteps % 8 == 0 && fsteps >= 16;       // rounds of two steps, four in flight: a multiple of 4 rounds, at least 6
#ifdef QV_VARIANTS
        // Measurement build only (make VARIANTS=1 -> libqv_dev.so; profiles/r03_batched_epilogue.txt has what each measured): the
        // shapes of the eight-wave kernel that lost to the shipped one.  The product library instantiates none of them.
        static const int w8_bf = env_int("QV_MFMA_W8_BF", 2);                                  // 1 = the eight-wave kernel on the bfloat16 plane too (485 against 474 us for k_bf16rows_filter)
        static const int w8x2 = env_int("QV_MFMA_W8X2", 1);                                 // 2 = 128 rows per round (k_bf16x1_filter_w8: 607 against 552 us); 3 = 256 with rows 8 steps ahead (spills)
        static const int w8_shape = env_int("QV_MFMA_W8_SHAPE", 1);                           // 7 = the dense pass deferred into the next group's K loop (610.7 against 607.3 us), 5 / 6 = rows 16 rounds ahead (614 / 613), 3 / 4 = query operands 15 steps ahead (644.6 / 648.6)
#define QV_FS_VARIANTS(MMM)                                                                                                                                                        \
                     if (bfrows && w8 && w8_bf == 1) hipLaunchKernelGGL((k_bf16x1_filter_w8<MMM, 4, 8, 2, true, false>), dim3(gs), dim3(512), 0, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt); \
                     else if (!bfrows && gmode == 2 && w8 && w8_shape == 3 && fsteps % 16 == 0 && fsteps >= 32) hipLaunchKernelGGL((k_bf16x1_filter_w8<MMM, 16, 16, 1, false, false>), dim3(gs), dim3(512), 0, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt); \
                     else if (!bfrows && gmode == 2 && w8 && w8_shape == 4 && fsteps % 16 == 0) hipLaunchKernelGGL((k_bf16x1_filter_w8<MMM, 8, 16, 1, false, false>), dim3(gs), dim3(512), 0, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt); \
                     else if (!bfrows && gmode == 2 && w8 && w8_shape == 5 && fsteps % 16 == 0 && fsteps >= 32) hipLaunchKernelGGL((k_bf16x1_filter_w8<MMM, 16, 4, 1, false, false>), dim3(gs), dim3(512), 0, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt); \
                     else if (!bfrows && gmode == 2 && w8 && w8_shape == 6 && fsteps % 16 == 0 && fsteps >= 32) hipLaunchKernelGGL((k_bf16x1_filter_w8<MMM, 16, 8, 1, false, false>), dim3(gs), dim3(512), 0, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt); \
                     else if (!bfrows && gmode == 2 && w8 && w8x2 == 3) hipLaunchKernelGGL((k_bf16x1_filter_w8x2<MMM, 8, 4>), dim3(gs), dim3(512), 0, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt); \
                     else if (!bfrows && gmode == 2 && w8 && w8x2 != 1 && w8_shape == 7) hipLaunchKernelGGL((k_bf16x1_filter_w8<MMM, 8, 4, 1, false, true>), dim3(gs), dim3(512), 0, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt); \
                     else if (!bfrows && gmode == 2 && w8 && w8x2 != 1) hipLaunchKernelGGL((k_bf16x1_filter_w8<MMM, 8, 4, 1, false, false>), dim3(gs), dim3(512), 0, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt); \
                     else
#else
#define QV_FS_VARIANTS(MMM)
#endif
        const bool qreg = gmode == 2 && (bfrows || w8) && qreg_filter_applies(v, nq_pad, bfrows);   // query operands in registers (qv_qreg.hip)
        // shipped: the bfloat16 copy's own kernel when the index keeps one; 256 rows per round on float32 rows (k_bf16x1_filter_w8x2)
        // where the dimension allows the eight-wave shape; the four-wave shared-row kernels otherwise
// (the one-term filter as four waves, k_bf16x3_filter_shared<., 1>: only QV_MFMA_W8=2 of the measurement build reaches it — every shape the
// eight-wave kernel's loop does not divide takes its zero-padded form, pad_main)
#ifdef QV_VARIANTS
#define QV_FS_ONE_TERM_4W(MMM) else if (gmode == 2) hipLaunchKernelGGL((k_bf16x3_filter_shared<MMM, 1, 8>), dim3(gs), dim3(256), 0, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt);
#else
#define QV_FS_ONE_TERM_4W(MMM)
#endif
#define QV_FS(MMM) { QV_FS_VARIANTS(MMM)                                                                                                                                           \
                     if (qreg) { e = launch_qreg_filter(v, Qbf, cq, mq, nq_pad, cand, cscore, cnt, bfrows, cus, s); if (e != hipSuccess) return e; } \
                     else if (bfrows) hipLaunchKernelGGL((k_bf16rows_filter<MMM>), dim3(grid_multiple(2 * (uint32_t)cus, nqb64 / 4)), dim3(256), 0, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt); \
                     else if (gmode == 2 && w8) hipLaunchKernelGGL((k_bf16x1_filter_w8x2<MMM, 4, 4>), dim3(gs), dim3(512), 0, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt); \
                     else if (pad_main) hipLaunchKernelGGL((k_bf16x1_filter_w8x2<MMM, 4, 4, true>), dim3(gs), dim3(512), 0, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt); \
                     QV_FS_ONE_TERM_4W(MMM)                                                                                                                                           \
                     else hipLaunchKernelGGL((k_bf16x3_filter_shared<MMM, 3, 8>), dim3(gs), dim3(256), 0, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt); }
        if (v.metric == QV_COSINE) QV_FS(QV_COSINE) else if (v.metric == QV_DOT) QV_FS(QV_DOT) else QV_FS(QV_L2)
#undef QV_FS
#undef QV_FS_VARIANTS
#undef QV_FS_ONE_TERM_4W
    } else if (bf) {
        const uint4* Qbf = reinterpret_cast<const uint4*>(Qt);
        if (v.metric == QV_COSINE) hipLaunchKernelGGL(k_bf16x3_filter<QV_COSINE>, dim3(grid), dim3(256), 0, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt, (float*)nullptr, 0u, 1u);
        else if (v.metric == QV_DOT) hipLaunchKernelGGL(k_bf16x3_filter<QV_DOT>, dim3(grid), dim3(256), 0, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt, (float*)nullptr, 0u, 1u);
        else hipLaunchKernelGGL(k_bf16x3_filter<QV_L2>, dim3(grid), dim3(256), 0, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt, (float*)nullptr, 0u, 1u);
#ifdef QV_VARIANTS
    } else if (f32_nj == 2) {
        if (v.metric == QV_COSINE) hipLaunchKernelGGL((k_mfma_filter<QV_COSINE, 2>), dim3(grid), dim3(256), 0, s, v, Qt, cq, mq, nq_pad, cand, cscore, cnt);
        else if (v.metric == QV_DOT) hipLaunchKernelGGL((k_mfma_filter<QV_DOT, 2>), dim3(grid), dim3(256), 0, s, v, Qt, cq, mq, nq_pad, cand, cscore, cnt);
        else hipLaunchKernelGGL((k_mfma_filter<QV_L2, 2>), dim3(grid), dim3(256), 0, s, v, Qt, cq, mq, nq_pad, cand, cscore, cnt);
#endif
    } else if (v.metric == QV_COSINE) hipLaunchKernelGGL((k_mfma_filter<QV_COSINE, 4>), dim3(grid), dim3(256), 0, s, v, Qt, cq, mq, nq_pad, cand, cscore, cnt);
    else if (v.metric == QV_DOT) hipLaunchKernelGGL((k_mfma_filter<QV_DOT, 4>), dim3(grid), dim3(256), 0, s, v, Qt, cq, mq, nq_pad, cand, cscore, cnt);
    else hipLaunchKernelGGL((k_mfma_filter<QV_L2, 4>), dim3(grid), dim3(256), 0, s, v, Qt, cq, mq, nq_pad, cand, cscore, cnt);   // L2 and L2SQ share the filter
    if (ev1) (void)hipEventRecord(ev1, s);
    // 4. exact re-scoring + selection
    const size_t lds = query_lds_bytes(v.metric, v.dim4) + 4 * 64 * sizeof(uint64_t) + (size_t)kMfmaCandCap * (sizeof(uint32_t) + sizeof(float) + sizeof(uint32_t));   // query, wave lists, survivors, lower bounds, upper bounds' keys
    // chunk requests per round of the exact pass, all issued before the round's arithmetic (row_accumulate's BAR): a survivor's row is
    // a gather of dim4 separate lines and the pass is a chain of dim4 / U dependent round trips — 8 / 16 / 32 / 48 / 64 per round:
    // 64.4 / 58.0 / 54.2 / 55.3 / 55.6 us per launch at 256 x 1M x 768 (left to the compiler's own schedule, as the streaming scans
    // are for cosine, deeper rounds were slower: 90 us at 16)
    static const int rs_u = dev_env_int("QV_MFMA_RESCORE_U", 32);
#define QV_RS1(MMM, UU) { e = set_lds(k_rescore_select<MMM, UU>, lds); if (e != hipSuccess) return e;                                   \
        hipLaunchKernelGGL((k_rescore_select<MMM, UU>), dim3(nq), dim3(256), lds, s, v, d_queries, cand, cscore, cnt, k, d_rows_out, d_dist_out, ovf, eq); }
#ifdef QV_VARIANTS
#define QV_RS(MMM) { if (rs_u == 8) QV_RS1(MMM, 8) else QV_RS1(MMM, 32) }
#else
#define QV_RS(MMM) { (void)rs_u; QV_RS1(MMM, 32) }
#endif
    if (large_k) {
        // intervals -> H (k-th smallest upper bound) -> survivors -> exact distances -> the k best; srows / sdist take the first
        // selection's output (the sample's bound in sdist has been consumed by k_mfma_prep)
        const dim3 cgrid((ccap + 255) / 256, nq);
        const size_t lds_x = query_lds_bytes(v.metric, v.dim4);
        const size_t lds_n = (size_t)ccap * 4 + (size_t)v.dim * 4;
        static const int narrow_mode = dev_env_int("QV_LK_NARROW", 1);
        const bool narrow = narrow_mode == 1 && lds_n <= 140 * 1024;       // (beyond: k_cand_qnorms, k_cand_bounds, the selection's kernels, k_cand_survive)
        const size_t lds_w = 2 * (size_t)kRsSlabBytes + (size_t)((v.dim4 + 7) / 8) * 128;
        static const int wave_mode = dev_env_int("QV_LK_EXACT_WAVE", 1);
        const bool wave_exact = wave_mode == 1;                          // (2: the lane-per-row kernel of rounds 4-5, measurement build only)
        // gather or tile pass: ~2.1 k survivors per query of unstructured rows, 12 KiB of requests each at ~3.5 TB/s, against the
        // corpus once at ~5.5 TB/s and 60 us for the three sorting kernels (256 queries x 1M x 768: from k ~ 300)
        static const int tp_mode = dev_env_int("QV_LK_TILE_PASS", 1);   // 2 = never, 3 = always (measurements)
        const double t_gather = 2.1 * k * nq * (double)v.dim4 * 64.0 / 3.5e12, t_pass = (double)v.n_rows * v.dim4 * 16.0 / 5.5e12 + 60e-6;
        const bool tile_pass = v.rowmaj == nullptr && (v.dim & 3u) == 0 && (reinterpret_cast<uintptr_t>(d_queries) & 15u) == 0 && (uint64_t)nq * ccap < (1ull << 32) && (uint64_t)nq * v.dim < (1ull << 32) && tp_mode != 2 && (tp_mode == 3 || t_gather > t_pass);
#ifdef QV_VARIANTS
#define QV_LK_LANE(MMM) { e = set_lds(k_cand_exact<MMM, 32>, lds_x); if (e != hipSuccess) return e;                                                         \
        hipLaunchKernelGGL((k_cand_exact<MMM, 32>), dim3(nq, (ccap + 255) / 256), dim3(256), lds_x, s, v, d_queries, surv, nsurv, ccap, qnorms, keys_ex); }
#else
#define QV_LK_LANE(MMM) { (void)lds_x; return hipErrorNotSupported; }
#endif
#define QV_LK(MMM) { e = set_lds(k_cand_qnorms<MMM>, (size_t)v.dim * 4); if (e != hipSuccess) return e;                                                 \
        if (narrow) { e = set_lds(k_cand_narrow<MMM>, lds_n); if (e != hipSuccess) return e;                                                                \
        hipLaunchKernelGGL(k_cand_narrow<MMM>, dim3(nq), dim3(1024), lds_n, s, v, d_queries, cand, cscore, cnt, ccap, eq, k, guess ? ubuf : (const float*)nullptr, ks, qnorms, lo_b, surv, nsurv, ovf, tp_cnt, tile_pass ? v.n_tiles + 1 : 0u); \
        } else {                                                                                                                                            \
        hipLaunchKernelGGL(k_cand_qnorms<MMM>, dim3(nq), dim3(64), (size_t)v.dim * 4, s, d_queries, v.dim, qnorms);                                         \
        hipLaunchKernelGGL(k_cand_bounds<MMM>, cgrid, dim3(256), 0, s, v, cand, cscore, cnt, ccap, eq, qnorms, keys_hi, lo_b, ovf, nsurv);                 \
        e = launch_select_topk(keys_hi, ccap, ccap, nq, k, k, sel_ws, srows, sdist, s, false, false); if (e != hipSuccess) return e;                       \
        hipLaunchKernelGGL(k_cand_survive, cgrid, dim3(256), 0, s, cand, cnt, ccap, lo_b, sdist, k, surv, nsurv, guess ? ubuf : (const float*)nullptr, ks, ovf); }                                          \
        if (tile_pass) {                                                                                                                                    \
            if (!narrow) (void)hipMemsetAsync(tp_cnt, 0, (size_t)(v.n_tiles + 1) * 4, s);                                                                                 \
            hipLaunchKernelGGL(k_tp_count, cgrid, dim3(256), 0, s, surv, nsurv, ccap, tp_cnt, keys_ex);                                                     \
            hipLaunchKernelGGL(k_tp_scan, dim3(1), dim3(1024), 0, s, tp_cnt, v.n_tiles, tp_off);                                                            \
            hipLaunchKernelGGL(k_tp_scatter, cgrid, dim3(256), 0, s, surv, nsurv, ccap, tp_off, tp_cnt, reinterpret_cast<uint2*>(keys_hi));                \
            hipLaunchKernelGGL((k_tp_exact<MMM, 8>), dim3(v.n_tiles), dim3(64), 4 * 8192, s, v, d_queries, tp_off, reinterpret_cast<const uint2*>(keys_hi), ccap, qnorms, keys_ex); \
        } else if (wave_exact) { e = set_lds(k_cand_exact_wave<MMM>, lds_w); if (e != hipSuccess) return e;                                                  \
            hipLaunchKernelGGL(k_cand_exact_wave<MMM>, dim3(nq, ccap / 32), dim3(64), lds_w, s, v, d_queries, surv, nsurv, ccap, qnorms, keys_ex); }        \
        else QV_LK_LANE(MMM)                                                                                                                                \
        e = ccap <= 16384u ? launch_select_topk_counted(keys_ex, ccap, ccap, nsurv, nq, k, k, d_rows_out, d_dist_out, s)                                   \
                          : launch_select_topk(keys_ex, ccap, ccap, nq, k, k, sel_ws, d_rows_out, d_dist_out, s, false, false); if (e != hipSuccess) return e; }
        if (v.metric == QV_COSINE) QV_LK(QV_COSINE) else if (v.metric == QV_DOT) QV_LK(QV_DOT) else if (v.metric == QV_L2) QV_LK(QV_L2) else QV_LK(QV_L2SQ)
#undef QV_LK
#undef QV_LK_LANE
        *d_overflow_out = ovf;
        return hipGetLastError();
    }
    if (v.metric == QV_COSINE) QV_RS(QV_COSINE) else if (v.metric == QV_DOT) QV_RS(QV_DOT) else if (v.metric == QV_L2) QV_RS(QV_L2) else QV_RS(QV_L2SQ)
#undef QV_RS
#undef QV_RS1
    *d_overflow_out = ovf;
    return hipGetLastError();
}


}  // namespace qv
