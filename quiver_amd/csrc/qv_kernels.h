// qv_kernels.h — shared device helpers of gfx950 (MI355X, CDNA4) kernels of the similarity-search hot path.
//
// What the kernels replace in the reference (paths relative to the reference tree):
//   k_flat_scan     ExactIndex.Search's distance loop + sort.Sort + truncate
//                   (pkg/hybrid/exact.go:115-129) with vectortypes' distance
//                   arithmetic fused in (pkg/vectortypes/distances.go:12-104,
//                   pkg/hnsw/adapter.go:105-167)
//   k_merge_lists   the tail of the same sort: merge of per-workgroup top-k lists
//   k_distance_rows the neighbour loop of HNSW.searchLayer (pkg/hnsw/hnsw.go:536-563)
//                   and the re-rank loops (pkg/hybrid/hybrid_index.go:536-546)
//   k_distance_pairs one vectortypes.DistanceFunc call per pair (surface.go:8)
//   k_ingest / k_generate   copy-on-insert (exact.go:53-56) into the tile layout
//
// Arithmetic contract: every distance is computed by ONE lane walking its row's
// dimensions 0..D-1 in order, in the precision the reference uses (float64
// accumulation of exact float32 products, or float32 unfused for the *_F32 metrics),
// so results are bit-identical to the reference's scalar Go loops — there is no
// cross-lane partial-sum reduction to reorder the additions.  Cross-lane work
// (ballot / readlane / wave shifts) is used only for top-k selection on 64-bit
// (distance, row) keys, which is exact integer work.
//
// Built with -ffp-contract=off: float32 paths must NOT be fused; float64 paths use
// explicit fma(), which is bit-identical to mul+add there because the products of
// float32-valued doubles are exact.
#pragma once
#include "qv_device.h"
#include "../../include/qv.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace qv {

typedef float f4 __attribute__((ext_vector_type(4)));   // native vector: lets the nontemporal builtin emit global_load_dwordx4 nt

// ---------------------------------------------------------------- wave helpers -----
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ uint64_t readlane64(uint64_t x, uint32_t src /*uniform*/) {
    uint32_t lo = __builtin_amdgcn_readlane((uint32_t)x, src);
    uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(x >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
// lane i <- lane i-1 (lane 0 keeps its value); full-wave shift right by one
__device__ __forceinline__ uint64_t wave_shr1(uint64_t x) {
    // DPP wave_shr:1 (0x138) is a gfx9-family control; bound_ctrl=0 keeps lane 0's old value
    uint32_t lo = __builtin_amdgcn_update_dpp((uint32_t)x, (uint32_t)x, 0x138, 0xf, 0xf, false);
    uint32_t hi = __builtin_amdgcn_update_dpp((uint32_t)(x >> 32), (uint32_t)(x >> 32), 0x138, 0xf, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}

// float32 -> uint32 whose unsigned order is the float order; NaN sorts after +inf
__device__ __forceinline__ uint32_t ord_f32(float f) {
    if (f != f) return 0xFFFFFFFEu;                  // canonical NaN key (below the dead sentinel)
    if (f == 0.0f) f = 0.0f;                         // -0 -> +0 (Go compares them equal)
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord_f32(uint32_t k) {
    if (k == 0xFFFFFFFEu) return __uint_as_float(0x7FC00000u);
    uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}
__device__ __forceinline__ uint64_t make_key(float dist, uint32_t row) { return ((uint64_t)ord_f32(dist) << 32) | row; }

// Sorted wave-resident list: lane i holds the i-th smallest key seen so far
// (kDeadKey = empty).  Inserts every lane's `key` that beats the current k-th key.
__device__ __forceinline__ void list_insert(uint64_t& list, uint64_t& thr, uint64_t key, uint32_t kth_lane, uint32_t lane) {
    uint64_t mask = __ballot(key < thr);
    while (mask) {
        uint32_t src = (uint32_t)__builtin_ctzll(mask);
        mask &= mask - 1;
        uint64_t c = readlane64(key, src);
        if (c >= thr) continue;                      // threshold tightened since the ballot
        uint32_t pos = (uint32_t)__builtin_popcountll(__ballot(list < c));
        uint64_t up = wave_shr1(list);
        list = lane > pos ? up : (lane == pos ? c : list);
        thr = readlane64(list, kth_lane);
    }
}

// The same list with R keys per lane: slot r * 64 + lane holds the (r * 64 + lane)-th smallest key seen so far — 64 R slots for
// k up to 64 R (k_flat_scan_wide: 64 < k <= 256).  An insert shifts every slot from its position up by one: one DPP shift per
// register, lane 0 of register r taking lane 63 of register r - 1.  kth = k - 1, the slot whose key is the admission threshold.
template <int R>
__device__ __forceinline__ void wide_insert(uint64_t (&list)[R], uint64_t& thr, uint64_t key, uint32_t kth, uint32_t lane) {
    uint64_t mask = __ballot(key < thr);
    while (mask) {
        const uint32_t src = (uint32_t)__builtin_ctzll(mask);
        mask &= mask - 1;
        const uint64_t c = readlane64(key, src);
        if (c >= thr) continue;                      // threshold tightened since the ballot
        uint32_t pos = 0;
#pragma unroll
        for (int r = 0; r < R; r++) pos += (uint32_t)__builtin_popcountll(__ballot(list[r] < c));
#pragma unroll
        for (int r = R - 1; r >= 0; r--) {           // top register first: register r - 1 is still the old one when its lane 63 is read
            uint64_t up = wave_shr1(list[r]);
            if (r > 0) { const uint64_t carry = readlane64(list[r - 1], 63); if (lane == 0) up = carry; }
            const uint32_t slot = (uint32_t)r * 64 + lane;
            list[r] = slot > pos ? up : (slot == pos ? c : list[r]);
        }
#pragma unroll
        for (int r = 0; r < R; r++) if ((uint32_t)r == (kth >> 6)) thr = readlane64(list[r], kth & 63);
    }
}

// ascending bitonic sort of one key per lane across the wave (21 compare-exchange steps)
__device__ __forceinline__ uint64_t wave_sort64(uint64_t key, uint32_t lane) {
#pragma unroll
    for (uint32_t k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
            uint32_t lo = __shfl_xor((uint32_t)key, (int)j), hi = __shfl_xor((uint32_t)(key >> 32), (int)j);
            uint64_t other = ((uint64_t)hi << 32) | lo;
            bool up = (lane & k2) == 0, lower = (lane & j) == 0;
            uint64_t mn = key < other ? key : other, mx = key < other ? other : key;
            key = (lower == up) ? mn : mx;
        }
    }
    return key;
}

// ---------------------------------------------------------------- metric traits ----
template <int M> struct MT;
// f64-accumulating metrics take the query as double, the rest as float
template <> struct MT<QV_COSINE>     { using Q = double; using A = double; static constexpr bool needs_rnorm = true;  };
template <> struct MT<QV_L2>         { using Q = float;  using A = double; static constexpr bool needs_rnorm = false; };
template <> struct MT<QV_L2SQ>       { using Q = float;  using A = float;  static constexpr bool needs_rnorm = false; };
template <> struct MT<QV_DOT>        { using Q = double; using A = double; static constexpr bool needs_rnorm = false; };
template <> struct MT<QV_L1>         { using Q = float;  using A = double; static constexpr bool needs_rnorm = false; };
template <> struct MT<QV_COSINE_F32> { using Q = float;  using A = float;  static constexpr bool needs_rnorm = true;  };
template <> struct MT<QV_L2_F32>     { using Q = float;  using A = float;  static constexpr bool needs_rnorm = false; };
template <> struct MT<QV_DOT_F32>    { using Q = float;  using A = float;  static constexpr bool needs_rnorm = false; };
template <> struct MT<QV_L2SQ_F64>   { using Q = double; using A = double; static constexpr bool needs_rnorm = false; };

// one element: a = query element (already in the metric's Q type), b = row element
template <int M> __host__ __device__ __forceinline__ void acc1(typename MT<M>::A& acc, typename MT<M>::Q a, float b) {
    if constexpr (M == QV_COSINE || M == QV_DOT) {
        acc = __builtin_fma(a, (double)b, acc);                       // distances.go:19 / :84
    } else if constexpr (M == QV_L2) {
        double d = (double)(a - b);                                   // float32 subtract, widen (distances.go:50)
        acc = __builtin_fma(d, d, acc);
    } else if constexpr (M == QV_L2SQ_F64) {
        double d = a - (double)b; double sq = d * d; acc = acc + sq;  // arrow_hnsw.go:128-129 (float64, unfused)
    } else if constexpr (M == QV_L1) {
        acc = acc + __builtin_fabs((double)(a - b));                  // distances.go:100
    } else if constexpr (M == QV_L2SQ || M == QV_L2_F32) {
        float d = a - b; float sq = d * d; acc = acc + sq;            // distances.go:67-68 / adapter.go:146-147 (unfused)
    } else {                                                          // QV_COSINE_F32, QV_DOT_F32
        float p = a * b; acc = acc + p;                               // adapter.go:117 / :161 (unfused)
    }
}

// per-query constants: for cosine metrics the query's own norm, computed once per
// wave in the reference's element order (distances.go:20: magnitudeA += a*a)
struct QConst { double qn; float qn32; };

template <int M, typename T> __device__ __forceinline__ QConst query_const(const T* q, uint32_t dim) {
    using Q = typename MT<M>::Q;
    QConst c; c.qn = 0.0; c.qn32 = 0.0f;
    if constexpr (M == QV_COSINE) {
        double ma = 0.0;
        for (uint32_t i = 0; i < dim; i++) { const Q a = (Q)q[i]; ma = __builtin_fma(a, a, ma); }
        c.qn = __builtin_sqrt(ma);                                    // sqrt(ma) == 0  <=>  ma == 0
    } else if constexpr (M == QV_COSINE_F32) {
        float na = 0.0f;
        for (uint32_t i = 0; i < dim; i++) { const Q a = (Q)q[i]; float p = a * a; na = na + p; }
        c.qn32 = (float)__builtin_sqrt((double)na);                   // adapter.go:128
        c.qn = (double)na;                                            // zero test is on na itself (adapter.go:122)
    }
    return c;
}

// rn = stored per-row norm (see k_ingest): sqrt(mb) for COSINE; for COSINE_F32 the
// float32 value float32(sqrt(float64(nb))) widened, negative if nb == 0 cannot occur,
// so rn == 0 <=> nb == 0 only when the sqrt underflows; we store nb's zero-ness in the sign bit
template <int M> __host__ __device__ __forceinline__ float finalize(typename MT<M>::A acc, const QConst& qc, double rn) {
    if constexpr (M == QV_COSINE) {
        if (qc.qn == 0.0 || rn == 0.0) return 1.0f;                   // distances.go:25-27
        double sim = acc / (qc.qn * rn);                              // :30
        if (sim > 1.0) sim = 1.0; else if (sim < -1.0) sim = -1.0;    // :32-36
        return (float)(1.0 - sim);                                    // :39
    } else if constexpr (M == QV_L2) {
        return (float)__builtin_sqrt(acc);                            // :54
    } else if constexpr (M == QV_DOT) {
        return (float)(1.0 - acc);                                    // :89
    } else if constexpr (M == QV_L1 || M == QV_L2SQ_F64) {
        return (float)acc;                                            // :103 / arrow_hnsw.go:131
    } else if constexpr (M == QV_L2SQ) {
        return acc;                                                   // :71
    } else if constexpr (M == QV_COSINE_F32) {
        if (qc.qn == 0.0 || rn < 0.0) return 1.0f;                    // adapter.go:122-124 (rn < 0 encodes nb == 0)
        float den = qc.qn32 * (float)rn;                              // :128
        float sim = acc / den;
        if (sim > 1.0f) sim = 1.0f; else if (sim < -1.0f) sim = -1.0f;
        return 1.0f - sim;                                            // :135
    } else if constexpr (M == QV_L2_F32) {
        return (float)__builtin_sqrt((double)acc);                    // adapter.go:150
    } else {
        return 1.0f - acc;                                            // adapter.go:164
    }
}

// One pair (a, b) of plain row-major vectors: the whole DistanceFunc (pkg/vectortypes/surface.go:8) in the reference's element
// order — the body of k_distance_pairs on the device, and of the host entry point qv_distance_pair (the same source
// compiled for the host: ONE statement of the arithmetic).
template <int M> __host__ __device__ __forceinline__ float pair_distance(const float* pa, const float* pb, uint32_t dim) {
    using Q = typename MT<M>::Q;
    typename MT<M>::A acc = 0;
    QConst qc; qc.qn = 0.0; qc.qn32 = 0.0f;
    double rn = 0.0;
    if constexpr (M == QV_COSINE) {
        double ma = 0.0, mb = 0.0;
        for (uint32_t j = 0; j < dim; j++) {                           // distances.go:18-22, one pass, element order
            double x = pa[j], y = pb[j];
            acc = __builtin_fma(x, y, acc); ma = __builtin_fma(x, x, ma); mb = __builtin_fma(y, y, mb);
        }
        qc.qn = __builtin_sqrt(ma); rn = __builtin_sqrt(mb);
    } else if constexpr (M == QV_COSINE_F32) {
        float na = 0.0f, nb = 0.0f;
        for (uint32_t j = 0; j < dim; j++) {                           // adapter.go:116-120
            float x = pa[j], y = pb[j];
            float p0 = x * y; acc = acc + p0; float p1 = x * x; na = na + p1; float p2 = y * y; nb = nb + p2;
        }
        qc.qn = (double)na; qc.qn32 = (float)__builtin_sqrt((double)na);
        rn = nb == 0.0f ? -1.0 : (double)(float)__builtin_sqrt((double)nb);
    } else {
        for (uint32_t j = 0; j < dim; j++) acc1<M>(acc, (Q)pa[j], pb[j]);
    }
    return finalize<M>(acc, qc, rn);
}

// stage the query into LDS in the metric's Q type, zero-padded to dim4*4
template <int M> __device__ __forceinline__ void stage_query(typename MT<M>::Q* q_lds, const float* q, uint32_t dim, uint32_t dim4) {
    for (uint32_t i = threadIdx.x; i < dim4 * 4; i += blockDim.x) q_lds[i] = i < dim ? (typename MT<M>::Q)q[i] : (typename MT<M>::Q)0;
}

// distance of the query (in LDS) to the row whose chunk c lives at p[c * stride4]
// QN: also accumulate the query's own squared norm in the reference's element order
// (distances.go:20 magnitudeA += a*a; adapter.go:118 normA += a*a).  It is an independent
// dependency chain, so riding along with a row's dot product costs no time; a wave does it
// on its first tile only.
// NT: non-temporal requests (a corpus streamed once per query must not displace the query and the partial lists from L2); false
// for collections small enough to STAY in L2 between calls (k_flat_scan_small)
template <int M, int U, bool QN, bool BAR = false, bool NT = true>
__device__ __forceinline__ typename MT<M>::A row_accumulate(const f4* __restrict__ p, uint32_t stride4,
                                                            const typename MT<M>::Q* __restrict__ q_lds, uint32_t dim4,
                                                            typename MT<M>::A* qnorm2 = nullptr) {
    using Q = typename MT<M>::Q;
    using A = typename MT<M>::A;
    A acc = 0, qa = 0;
    auto qn1 = [&](Q a) {
        if constexpr (QN && M == QV_COSINE) qa = __builtin_fma(a, a, qa);
        else if constexpr (QN && M == QV_COSINE_F32) { float pp = a * a; qa = qa + pp; }
    };
    // one block of B chunks: all B loads issued together, then consumed in order
    auto block = [&](auto Bc, uint32_t c0) {
        constexpr int B = decltype(Bc)::value;
        f4 v[B];
#pragma unroll
        for (int u = 0; u < B; u++) { if constexpr (NT) v[u] = __builtin_nontemporal_load(&p[(size_t)(c0 + u) * stride4]); else v[u] = p[(size_t)(c0 + u) * stride4]; }
        // Keep all loads of the block ahead of the arithmetic: left alone, hipcc (ROCm 7.2) sinks them next to their uses for
        // every metric but cosine — 2 loads in flight instead of 16: dot / Euclidean 78 %, squared Euclidean / Manhattan 52 % of
        // the HBM peak instead of 88 %.  For cosine its own schedule (a rolling window of ~8 loads) beats the hard barrier
        // (89.7 vs 86.6-87.9 % at any batch size 8..32), so the barrier is left out there; tests/test_isa_guard.py checks the
        // compiled loops (>= 16 loads in flight for the barriered metrics, >= 8 for cosine).
        if constexpr (M != QV_COSINE || BAR) __builtin_amdgcn_sched_barrier(0);   // (BAR: a caller whose loads are gathers, not a stream — k_rescore_select)
#pragma unroll
        for (int u = 0; u < B; u++) {
            const Q* qq = q_lds + (size_t)(c0 + u) * 4;
            acc1<M>(acc, qq[0], v[u].x); qn1(qq[0]); acc1<M>(acc, qq[1], v[u].y); qn1(qq[1]);
            acc1<M>(acc, qq[2], v[u].z); qn1(qq[2]); acc1<M>(acc, qq[3], v[u].w); qn1(qq[3]);
        }
    };
    uint32_t c0 = 0;
    for (; c0 + U <= dim4; c0 += U) block(std::integral_constant<int, U>{}, c0);
    // the remainder in blocks of 8, 4, 2, 1 loads (a chunk-at-a-time tail ran dim 16-48 at 40-45 % and dim 100 at 69 % of the HBM peak)
    if constexpr (U > 8) { for (; c0 + 8 <= dim4; c0 += 8) block(std::integral_constant<int, 8>{}, c0); }
    if constexpr (U > 4) { if (c0 + 4 <= dim4) { block(std::integral_constant<int, 4>{}, c0); c0 += 4; } }
    if (c0 + 2 <= dim4) { block(std::integral_constant<int, 2>{}, c0); c0 += 2; }
    if (c0 < dim4) block(std::integral_constant<int, 1>{}, c0);
    if constexpr (QN) *qnorm2 = qa;       // zero padding of the query adds +0 terms: exact
    return acc;
}

template <int M> __device__ __forceinline__ QConst qconst_from_norm2(typename MT<M>::A n2) {
    QConst c; c.qn = 0.0; c.qn32 = 0.0f;
    if constexpr (M == QV_COSINE) c.qn = __builtin_sqrt(n2);
    else if constexpr (M == QV_COSINE_F32) { c.qn32 = (float)__builtin_sqrt((double)n2); c.qn = (double)n2; }
    return c;
}

// Output: partial[(q*gridDim.x + blockIdx.x)*k + i] = workgroup's i-th best key.
constexpr int kScanBlock = 256;
constexpr int kScanWaves = kScanBlock / 64;

// ---------------------------------------------------------------- launcher helpers --
constexpr int kUnroll = 16;

// one tile for QB queries: acc[j] = Σ_d f(q_j[d], row[d]); FIRST also accumulates the query norms
template <int M, int U, int QB, bool FIRST>
__device__ __forceinline__ void mq_tile(const f4* __restrict__ p, const typename MT<M>::Q* __restrict__ q_lds, uint32_t dim4,
                                        typename MT<M>::A (&acc)[QB], typename MT<M>::A (&qa)[QB]) {
    using Q = typename MT<M>::Q;
    constexpr int VW = 16 / sizeof(Q);                          // queries per 16-byte LDS read (2 doubles or 4 floats)
    typedef Q qvec __attribute__((ext_vector_type(VW)));
    static_assert(QB % VW == 0, "QB must be a multiple of the LDS vector width");
#pragma unroll
    for (int j = 0; j < QB; j++) { acc[j] = 0; if constexpr (FIRST) qa[j] = 0; }
    auto chunk = [&](uint32_t c, f4 x) {
        const qvec* qq = reinterpret_cast<const qvec*>(q_lds + (size_t)c * 4 * QB);
        const float e[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int d = 0; d < 4; d++) {
#pragma unroll
            for (int g = 0; g < QB / VW; g++) {
                const qvec a = qq[d * (QB / VW) + g];             // one ds_read_b128, broadcast to the wave
#pragma unroll
                for (int t = 0; t < VW; t++) {
                    const int j = g * VW + t;
                    acc1<M>(acc[j], a[t], e[d]);
                    if constexpr (FIRST && M == QV_COSINE) qa[j] = __builtin_fma(a[t], a[t], qa[j]);
                    else if constexpr (FIRST && M == QV_COSINE_F32) { float pp = a[t] * a[t]; qa[j] = qa[j] + pp; }
                }
            }
        }
    };
    uint32_t c0 = 0;
    for (; c0 + U <= dim4; c0 += U) {
        f4 x[U];
#pragma unroll
        for (int u = 0; u < U; u++) x[u] = __builtin_nontemporal_load(&p[(size_t)(c0 + u) * 64]);
#pragma unroll
        for (int u = 0; u < U; u++) chunk(c0 + u, x[u]);
    }
    for (; c0 < dim4; c0++) chunk(c0, __builtin_nontemporal_load(&p[(size_t)c0 * 64]));
}


// A switch of the MEASUREMENT build only (make VARIANTS=1 -> libqv_dev.so): the product library takes the default — its dispatch
// is what the oracle-checked tests cover (tools/kernel_coverage.py), and an operator has no use for a tuning knob of one kernel.
// The product's own switches (INTEGRATION.md "Environment") go through env_int.
#ifdef QV_VARIANTS
#define dev_env_int(name, dflt) env_int(name, dflt)
#else
#define dev_env_int(name, dflt) (dflt)
#endif
static inline int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    if (!e || !*e) return dflt;
    int v = atoi(e);
    return v > 0 ? v : dflt;
}

static inline size_t query_lds_bytes(int metric, uint32_t dim4) {
    size_t q = (metric == QV_COSINE || metric == QV_DOT || metric == QV_L2SQ_F64) ? sizeof(double) : sizeof(float);
    return ((size_t)dim4 * 4 * q + 15) / 16 * 16;
}

template <typename F> static inline hipError_t set_lds(F f, size_t bytes) {
    if (bytes > 48 * 1024) return hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return hipSuccess;
}

#define QV_DISPATCH_METRIC(metric, CALL)                         \
    switch (metric) {                                            \
        case QV_COSINE:     { constexpr int MM = QV_COSINE;     CALL; } break; \
        case QV_L2:         { constexpr int MM = QV_L2;         CALL; } break; \
        case QV_L2SQ:       { constexpr int MM = QV_L2SQ;       CALL; } break; \
        case QV_DOT:        { constexpr int MM = QV_DOT;        CALL; } break; \
        case QV_L1:         { constexpr int MM = QV_L1;         CALL; } break; \
        case QV_COSINE_F32: { constexpr int MM = QV_COSINE_F32; CALL; } break; \
        case QV_L2_F32:     { constexpr int MM = QV_L2_F32;     CALL; } break; \
        case QV_DOT_F32:    { constexpr int MM = QV_DOT_F32;    CALL; } break; \
        case QV_L2SQ_F64:   { constexpr int MM = QV_L2SQ_F64;   CALL; } break; \
        default: return hipErrorInvalidValue;                    \
    }


}  // namespace qv
