// qv_batched.hip — fp32-MFMA batched filter + exact re-scoring
// (shared helpers, the arithmetic contract and the build flags: qv_kernels.h)
#include "qv_kernels.h"

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

// Qt[qb32][chunk][32 queries][4 dims] (zero padded), per-query filter constants, counters reset
__global__ void k_mfma_prep(const float* __restrict__ queries, uint32_t nq, uint32_t nq_pad, uint32_t dim, uint32_t dim4,
                            const float* __restrict__ sample_dist /*[nq][k]*/, uint32_t k, int metric,
                            float* __restrict__ Qt, float* __restrict__ cq, float* __restrict__ mq,
                            uint32_t* __restrict__ cand_cnt, uint32_t* __restrict__ overflow) {
    const uint32_t q = blockIdx.x;                     // one block per (padded) query
    const uint32_t dim4p = (dim4 + 1) & ~1u;           // chunk count padded to even: the MFMA step eats two chunks
    for (uint32_t c = threadIdx.x; c < dim4p; c += blockDim.x) {
        f4 x = {0.f, 0.f, 0.f, 0.f};
        if (q < nq && c < dim4) {
            const float* src = queries + (size_t)q * dim;
            uint32_t j = 4 * c;
            x.x = j < dim ? src[j] : 0.f; x.y = j + 1 < dim ? src[j + 1] : 0.f; x.z = j + 2 < dim ? src[j + 2] : 0.f; x.w = j + 3 < dim ? src[j + 3] : 0.f;
        }
        reinterpret_cast<f4*>(Qt)[((size_t)(q >> 5) * dim4p + c) * 32 + (q & 31)] = x;
    }
    if (threadIdx.x == 0) {
        float c_ = __uint_as_float(0x7F800000u), m_ = 0.f;       // padded queries: +inf threshold, nothing passes
        if (q < nq) {
            double n2 = 0.0;
            for (uint32_t i = 0; i < dim; i++) { double a = queries[(size_t)q * dim + i]; n2 = __builtin_fma(a, a, n2); }
            const double qn = __builtin_sqrt(n2);
            const double U = (double)sample_dist[(size_t)q * k + (k - 1)];     // +inf if the sample held < k live rows
            const double gamma = (double)(dim + 2) * 5.9604644775390625e-8 / (1.0 - (double)(dim + 2) * 5.9604644775390625e-8);
            double c, m;
            if (metric == QV_L2 || metric == QV_L2SQ) {
                // squared domain: real d^2 = |q|^2 + |r|^2 - 2S.  The reference value D relates to the real d by
                // D = d(1+eta), |eta| <= 1.3e-7 (QV_L2: float32 differences, float64 sum, sqrt, one rounding) or
                // D = d^2(1+eta), |eta| <= (K+2)u (QV_L2SQ: float32 accumulation), so D <= U implies d^2 <= T:
                const double T = metric == QV_L2 ? U * U * (1.0 + 4e-7) : U * (1.0 + gamma + 2e-6);
                c = n2 * (1.0 - 2e-6) - T;                           // A_q; test: 2S~ >= A_q + (1-2e-6)|r|^2 - B_q|r|
                m = 2.0 * (gamma + 1e-6) * qn;                       // B_q
                if (!(U == U) || U > 1.0e18) { c = -3.0e38; m = 0.0; }
            } else {
                c = metric == QV_COSINE ? (1.0 - U - 4e-7) * qn : (1.0 - U - 4e-7 * (1.0 + __builtin_fabs(U)));
                m = (gamma + 1e-6) * qn;
                if (!(U == U) || U > 3.0e38) { c = -3.0e38; m = 0.0; }            // no bound: everything is a candidate (overflow -> exact path)
            }
            c_ = f32_down((float)c);                                         // round towards "keep more"
            m_ = f32_up((float)m);
        }
        cq[q] = c_; mq[q] = m_;
        if (q < nq) { cand_cnt[q] = 0; overflow[q] = 0; }
    }
}

// grid: persistent waves; wave g -> query 64-block (g % nqb64), row groups (g / nqb64) + i*stride; a row group = 2 tiles = 128 rows
template <int METRIC>
__global__ void __launch_bounds__(256, 1)
k_mfma_filter(IndexView v, const float* __restrict__ Qt, const float* __restrict__ cq, const float* __restrict__ mq, uint32_t nq_pad,
              uint32_t* __restrict__ cand_rows, float* __restrict__ cand_score, uint32_t* __restrict__ cand_cnt) {
    __shared__ float s_c[4][64], s_m[4][64];                        // this wave's 64 queries' filter constants
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gw = blockIdx.x * 4 + wave, tw = gridDim.x * 4;
    const uint32_t nqb64 = nq_pad >> 6;
    const uint32_t qb64 = gw % nqb64;
    const uint32_t n_groups = (v.n_tiles + 1) / 2;
    const uint32_t stride = tw / nqb64;
    {   // cosine: one constant t_q = c_q - m_q (test S~ >= t_q |r|); dot: c_q and m_q (test S~ >= c_q - m_q |r|)
        const float c = cq[64 * qb64 + lane], m = mq[64 * qb64 + lane];
        s_c[wave][lane] = METRIC == QV_COSINE ? c - m : c;     // L2 family: A_q (c) and B_q (m)
        s_m[wave][lane] = m;
    }
    __syncthreads();
    if (stride == 0) return;
    const uint32_t half = lane >> 5, l31 = lane & 31;
    const f4* tiles = reinterpret_cast<const f4*>(v.tiles);
    const f4* qt = reinterpret_cast<const f4*>(Qt);
    const uint32_t steps = (v.dim4 + 1) / 2;                       // 8 dims per step
    const uint32_t dim4p = 2 * steps;                              // Qt is zero-padded to an even chunk count
    const f4* a_base0 = qt + ((size_t)(2 * qb64) * dim4p) * 32 + l31;
    const f4* a_base1 = qt + ((size_t)(2 * qb64 + 1) * dim4p) * 32 + l31;

    for (uint32_t g = gw / nqb64; g < n_groups; g += stride) {
        const uint32_t t0 = 2 * g, t1 = (2 * g + 1 < v.n_tiles) ? 2 * g + 1 : t0;     // odd tail: tile duplicated, masked below
        const f4* b0 = tiles + (size_t)t0 * v.dim4 * 64 + l31;
        const f4* b1 = tiles + (size_t)t1 * v.dim4 * 64 + l31;
        f16v acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

        // branch-free operand fetch: a step index past the end re-reads the last step (never used).
        // For an odd dim4 the upper lane half of the last step reads Qt's zero padding on the A side
        // and re-reads the last real chunk on the B side (0 * finite = 0).
        auto load = [&](uint32_t st, f4 (&A)[2], f4 (&B)[4]) {
            const uint32_t sc = st < steps ? st : steps - 1;
            const uint32_t ca = 2 * sc + half;
            const uint32_t cb = ca < v.dim4 ? ca : v.dim4 - 1;
            A[0] = a_base0[(size_t)ca * 32];
            A[1] = a_base1[(size_t)ca * 32];
            B[0] = __builtin_nontemporal_load(&b0[(size_t)cb * 64]);
            B[1] = __builtin_nontemporal_load(&b0[(size_t)cb * 64 + 32]);
            B[2] = __builtin_nontemporal_load(&b1[(size_t)cb * 64]);
            B[3] = __builtin_nontemporal_load(&b1[(size_t)cb * 64 + 32]);
        };
        auto mma = [&](const f4 (&A)[2], const f4 (&B)[4]) {
#pragma unroll
            for (int d = 0; d < 4; d++)
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i][d], B[j][d], acc[i][j], 0, 0, 0);
        };
        // 3-deep software pipeline over the K steps (operands for steps s+1, s+2 in flight while s computes)
        f4 A0[2], B0[4], A1[2], B1[4], A2[2], B2[4];
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

        // epilogue: acc[i][j][r] = S~[query 64*qb64 + 32*i + (r&3)+8*(r>>2)+4*half][row 64*(t0|t1) + 32*(j&1) + l31]
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t t = j < 2 ? t0 : t1;
            if (j >= 2 && t1 == t0) continue;
            const uint32_t row = t * 64 + 32 * (j & 1) + l31;
            const bool live = (v.alive[t] >> (32 * (j & 1) + l31)) & 1ull;
            const float rn = f32_up((float)v.rnorm[row]);
            const float rlo = f32_down((float)v.rnorm[row]);
            const float rn2c = f32_down(f32_down(rlo * rlo) * 0.999998f);   // (1-2e-6)|r|^2, rounded down (L2 family)
            (void)rn2c;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                bool hit = false;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const uint32_t ql = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float thr = METRIC == QV_COSINE ? s_c[wave][ql] * rn - 1e-30f : (METRIC == QV_DOT ? s_c[wave][ql] - s_m[wave][ql] * rn : 0.5f * (s_c[wave][ql] + rn2c - s_m[wave][ql] * rn));
                    hit |= acc[i][j][r] >= thr;
                }
                if (hit && live) {                                  // rare: a row that may be in some query's top-k
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const uint32_t ql = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                        const float thr = METRIC == QV_COSINE ? s_c[wave][ql] * rn - 1e-30f : (METRIC == QV_DOT ? s_c[wave][ql] - s_m[wave][ql] * rn : 0.5f * (s_c[wave][ql] + rn2c - s_m[wave][ql] * rn));
                        if (acc[i][j][r] >= thr) {
                            const uint32_t q = 64 * qb64 + ql;
                            uint32_t slot = atomicAdd(&cand_cnt[q], 1u);
                            if (slot < (uint32_t)kMfmaCandCap) {
                                cand_rows[(size_t)q * kMfmaCandCap + slot] = row;
                                cand_score[(size_t)q * kMfmaCandCap + slot] = acc[i][j][r];
                            }
                        }
                    }
                }
            }
        }
    }
}

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
                 uint32_t* __restrict__ overflow) {
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
    stage_query<M>(q_lds, queries + (size_t)qi * v.dim, v.dim, v.dim4);
    if (threadIdx.x == 0) s_ns = 0;
    __syncthreads();
    const QConst qc = query_const<M>(q_lds, v.dim);                 // the metric's own query constant
    double qn_l2 = qc.qn;                                           // |q| for the error bounds (every lane, same value)
    if constexpr (M != QV_COSINE) {
        double n2 = 0.0;
        for (uint32_t i = 0; i < v.dim; i++) { double a = (double)q_lds[i]; n2 = __builtin_fma(a, a, n2); }
        qn_l2 = __builtin_sqrt(n2);
    }
    const uint32_t kth = k - 1;
    const uint32_t* cr = cand_rows + (size_t)qi * kMfmaCandCap;
    const float* cs = cand_score + (size_t)qi * kMfmaCandCap;
    const double gamma = (double)(v.dim + 2) * 5.9604644775390625e-8 / (1.0 - (double)(v.dim + 2) * 5.9604644775390625e-8);

    // ---- stage 1: H = k-th smallest upper bound
    auto bounds = [&](uint32_t i, float& lo, float& hi) {
        const uint32_t row = cr[i];
        const double rn = v.rnorm[row], S = (double)cs[i];
        double d, e;
        if constexpr (M == QV_COSINE) {
            if (qc.qn == 0.0 || rn == 0.0) { d = 1.0; e = 0.0; }
            else { d = 1.0 - S / (qc.qn * rn); e = gamma + 2e-6; }  // |S~ - S| <= gamma |q||r|
        } else if constexpr (M == QV_DOT) {
            d = 1.0 - S; e = gamma * qn_l2 * rn + 2e-6 * (1.0 + __builtin_fabs(d));
        } else {                                                    // QV_L2 / QV_L2SQ: interval on d^2, then into the metric's units
            const double q2 = qn_l2 * qn_l2, r2 = rn * rn;
            const double d2 = q2 + r2 - 2.0 * S, e2 = 2.0 * gamma * qn_l2 * rn + 2e-6 * (q2 + r2);
            double l2 = d2 - e2 > 0.0 ? d2 - e2 : 0.0, h2 = d2 + e2 > 0.0 ? d2 + e2 : 0.0;
            if constexpr (M == QV_L2) { l2 = __builtin_sqrt(l2) * (1.0 - 4e-7); h2 = __builtin_sqrt(h2) * (1.0 + 4e-7); }
            else { l2 = l2 * (1.0 - gamma - 2e-6); h2 = h2 * (1.0 + gamma + 2e-6); }
            lo = f32_down((float)l2); hi = f32_up((float)h2);
            if (!(d2 == d2)) { lo = -__builtin_inff(); hi = __builtin_inff(); }
            return;
        }
        lo = f32_down((float)(d - e)); hi = f32_up((float)(d + e));
        if (!(d == d)) { lo = -__builtin_inff(); hi = __builtin_inff(); }   // NaN score: keep, the exact pass decides
    };
    uint64_t list = kDeadKey, thr = kDeadKey;
    for (uint32_t base = wave * 64; base < cnt; base += 4 * 64) {
        const uint32_t i = base + lane;
        uint64_t key = kDeadKey;
        if (i < cnt) { float lo, hi; bounds(i, lo, hi); key = make_key(hi, i); }
        list_insert(list, thr, key, kth, lane);
    }
    wl[wave * 64 + lane] = list;
    __syncthreads();
    if (wave == 0) {
        for (uint32_t w = 1; w < 4; w++) {
            uint64_t key = lane < k ? wl[w * 64 + lane] : kDeadKey;
            list_insert(list, thr, key, kth, lane);
        }
        const uint64_t kk = readlane64(list, kth);
        if (lane == 0) s_H = kk == kDeadKey ? __builtin_inff() : unord_f32((uint32_t)(kk >> 32));   // < k candidates: keep all
    }
    __syncthreads();
    const float H = s_H;
    for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
        float lo, hi; bounds(i, lo, hi);
        if (lo <= H) surv[atomicAdd(&s_ns, 1u)] = cr[i];
    }
    __syncthreads();
    const uint32_t ns = s_ns;

    // ---- stage 2: exact distances of the survivors, top-k by (distance, row)
    list = kDeadKey; thr = kDeadKey;
    for (uint32_t base = wave * 64; base < ns; base += 4 * 64) {
        const uint32_t i = base + lane;
        uint64_t key = kDeadKey;
        if (i < ns) {
            const uint32_t row = surv[i];
            const f4* p = reinterpret_cast<const f4*>(v.tiles) + (size_t)(row >> 6) * v.dim4 * 64 + (row & 63);
            typename MT<M>::A acc = row_accumulate<M, U, false>(p, 64, q_lds, v.dim4);
            double rn = 0.0;
            if constexpr (MT<M>::needs_rnorm) rn = v.rnorm[row];
            key = make_key(finalize<M>(acc, qc, rn), row);
        }
        list_insert(list, thr, key, kth, lane);
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
    }
}

// ---- MFMA batched path --------------------------------------------------------------
// Rows of the exact sample scan that bounds each query's k-th distance.  A sample of S of N rows lets about k*N/S rows through
// the filter per query; the candidate buffer holds kMfmaCandCap (4096), so S grows with N and k to keep that near 1536
// (10M rows or k = 64 with the former fixed 8192 overflowed nearly every query into the exact redo: 256 x 10M x 768 took 108 ms).
uint32_t batched_sample_rows(uint32_t n_rows, uint32_t k) {
    static const int forced = env_int("QV_MFMA_SAMPLE_ROWS", 0);
    if (forced > 0) return std::min<uint32_t>(n_rows, (uint32_t)forced);
    // ~1536 expected candidates (3 sigma of the k-th order statistic at k = 10 stays under the 4096 slots), in whole multiples
    // of 8192 rows = 128 tiles: with 16 query groups that is one full round of the 2048 scan waves per multiple
    const uint64_t want = ((uint64_t)n_rows * std::max(k, 1u) / 1536 + 8191) / 8192 * 8192;
    return (uint32_t)std::min<uint64_t>(n_rows, std::max<uint64_t>(8192, want));
}
bool batched_supported(const IndexView& v, uint32_t nq, uint32_t k) {
    // Measured crossover against the exact multi-query scans (256 queries x 768 dims, host pointers for the filter): 12k-16k
    // rows 0.49-0.50 vs 0.36-0.40 ms, 32k 0.51 vs 0.67, 64k 0.64 vs 1.34, 128k 0.85 vs 2.44, 200k 1.19 vs 3.16 — the filter's
    // fixed cost (sample scan, prep, re-score: ~0.4 ms) pays off from about 8M query-rows.
    static const int min_rows = env_int("QV_MFMA_MIN_ROWS", 32768), min_q = env_int("QV_MFMA_MIN_QUERIES", 32);
    static const int min_work_m = env_int("QV_MFMA_MIN_MROWS", 8);                       // millions of query-rows
    return (v.metric == QV_COSINE || v.metric == QV_DOT || v.metric == QV_L2 || v.metric == QV_L2SQ) && k <= (uint32_t)kMaxFusedK && nq >= (uint32_t)min_q &&
           v.n_rows >= (uint32_t)min_rows && (uint64_t)nq * v.n_rows >= (uint64_t)min_work_m * 1000000ull;
}
// queries are padded (zero vector, +inf threshold: nothing passes) to 1, 2 or a multiple of 4 blocks of 64: the four waves of a
// workgroup then work on the same row group (its tiles are fetched once and hit L1/L2 for the other three) and the wave count
// divides evenly over the blocks with one workgroup per CU.  Measured, 1M x 768: 160-192 queries as 3 blocks took 4.8 ms
// (258 workgroups on 256 CUs: a second round), as 4 blocks 3.4 ms.
static uint32_t batched_nq_pad(uint32_t nq) { return nq <= 64 ? 64u : (nq <= 128 ? 128u : (nq + 255) / 256 * 256); }

size_t batched_workspace_bytes(const IndexView& v, const ScanPlan& p, uint32_t nq, uint32_t k) {
    const uint32_t nq_pad = batched_nq_pad(nq);
    size_t b = scan_workspace_bytes(p, nq, k) + (size_t)(nq + 16) * v.dim4 * 4 * sizeof(double);   // sample scan (partials + query blocks)
    b = (b + 255) / 256 * 256;
    b += (size_t)nq_pad * (v.dim4 + 1) * 16;                 // Qt (chunk count padded to even)
    b += (size_t)nq_pad * 8;                                 // cq, mq
    b += (size_t)nq * kMfmaCandCap * 8;                      // candidates: rows + fp32 scores
    b += (size_t)nq * 8;                                     // counters, overflow flags
    b += (size_t)nq * k * 8;                                 // sample rows/dist
    return b + 1024;
}

hipError_t launch_batched(const IndexView& v, const ScanPlan& p, const float* d_queries, uint32_t nq, uint32_t k, void* d_ws,
                          uint32_t* d_rows_out, float* d_dist_out, uint32_t** d_overflow_out, int cus, hipStream_t s,
                          hipEvent_t ev0, hipEvent_t ev1) {
    const uint32_t nq_pad = batched_nq_pad(nq);
    char* w = static_cast<char*>(d_ws);
    size_t off = scan_workspace_bytes(p, nq, k) + (size_t)(nq + 16) * v.dim4 * 4 * sizeof(double);
    off = (off + 255) / 256 * 256;
    float* Qt = reinterpret_cast<float*>(w + off); off += (size_t)nq_pad * (v.dim4 + 1) * 16;
    float* cq = reinterpret_cast<float*>(w + off); off += (size_t)nq_pad * 4;
    float* mq = reinterpret_cast<float*>(w + off); off += (size_t)nq_pad * 4;
    uint32_t* cand = reinterpret_cast<uint32_t*>(w + off); off += (size_t)nq * kMfmaCandCap * 4;
    float* cscore = reinterpret_cast<float*>(w + off); off += (size_t)nq * kMfmaCandCap * 4;
    uint32_t* cnt = reinterpret_cast<uint32_t*>(w + off); off += (size_t)nq * 4;
    uint32_t* ovf = reinterpret_cast<uint32_t*>(w + off); off += (size_t)nq * 4;
    uint32_t* srows = reinterpret_cast<uint32_t*>(w + off); off += (size_t)nq * k * 4;
    float* sdist = reinterpret_cast<float*>(w + off); off += (size_t)nq * k * 4;
    // 1. exact top-k over a sample (first rows) -> per-query upper bound U_q of the k-th distance
    IndexView vs = v;
    vs.n_rows = batched_sample_rows(v.n_rows, k);
    vs.n_tiles = (vs.n_rows + 63) / 64;
    ScanPlan ps = plan_scan(vs.n_tiles, cus);
    hipError_t e = launch_flat_topk(vs, ps, d_queries, nq, k, d_ws, srows, sdist, s);
    if (e != hipSuccess) return e;
    // 2. query re-layout + filter constants
    hipLaunchKernelGGL(k_mfma_prep, dim3(nq_pad), dim3(64), 0, s, d_queries, nq, nq_pad, v.dim, v.dim4, sdist, k, v.metric, Qt, cq, mq, cnt, ovf);
    // 3. MFMA filter
    const uint32_t nqb64 = nq_pad / 64;
    uint32_t grid = (uint32_t)cus;                                     // one 4-wave workgroup per CU (512-register waves)
    while ((grid * 4) % nqb64) grid++;                                 // every query block gets the same number of waves
    if (ev0) (void)hipEventRecord(ev0, s);
    if (v.metric == QV_COSINE) hipLaunchKernelGGL(k_mfma_filter<QV_COSINE>, dim3(grid), dim3(256), 0, s, v, Qt, cq, mq, nq_pad, cand, cscore, cnt);
    else if (v.metric == QV_DOT) hipLaunchKernelGGL(k_mfma_filter<QV_DOT>, dim3(grid), dim3(256), 0, s, v, Qt, cq, mq, nq_pad, cand, cscore, cnt);
    else hipLaunchKernelGGL(k_mfma_filter<QV_L2>, dim3(grid), dim3(256), 0, s, v, Qt, cq, mq, nq_pad, cand, cscore, cnt);   // L2 and L2SQ share the filter
    if (ev1) (void)hipEventRecord(ev1, s);
    // 4. exact re-scoring + selection
    const size_t lds = query_lds_bytes(v.metric, v.dim4) + 4 * 64 * sizeof(uint64_t) + (size_t)kMfmaCandCap * sizeof(uint32_t);
#define QV_RS(MMM) { e = set_lds(k_rescore_select<MMM, 8>, lds); if (e != hipSuccess) return e;                                   \
        hipLaunchKernelGGL((k_rescore_select<MMM, 8>), dim3(nq), dim3(256), lds, s, v, d_queries, cand, cscore, cnt, k, d_rows_out, d_dist_out, ovf); }
    if (v.metric == QV_COSINE) QV_RS(QV_COSINE) else if (v.metric == QV_DOT) QV_RS(QV_DOT) else if (v.metric == QV_L2) QV_RS(QV_L2) else QV_RS(QV_L2SQ)
#undef QV_RS
    *d_overflow_out = ovf;
    return hipGetLastError();
}


}  // namespace qv
