// qv_device.hip — gfx950 (MI355X, CDNA4) kernels of the similarity-search hot path.
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
#include "qv_device.h"
#include "../../include/qv.h"
#include <stdlib.h>
#include <algorithm>

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
template <int M> __device__ __forceinline__ void acc1(typename MT<M>::A& acc, typename MT<M>::Q a, float b) {
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

template <int M> __device__ __forceinline__ QConst query_const(const typename MT<M>::Q* q, uint32_t dim) {
    QConst c; c.qn = 0.0; c.qn32 = 0.0f;
    if constexpr (M == QV_COSINE) {
        double ma = 0.0;
        for (uint32_t i = 0; i < dim; i++) ma = __builtin_fma(q[i], q[i], ma);
        c.qn = __builtin_sqrt(ma);                                    // sqrt(ma) == 0  <=>  ma == 0
    } else if constexpr (M == QV_COSINE_F32) {
        float na = 0.0f;
        for (uint32_t i = 0; i < dim; i++) { float p = q[i] * q[i]; na = na + p; }
        c.qn32 = (float)__builtin_sqrt((double)na);                   // adapter.go:128
        c.qn = (double)na;                                            // zero test is on na itself (adapter.go:122)
    }
    return c;
}

// rn = stored per-row norm (see k_ingest): sqrt(mb) for COSINE; for COSINE_F32 the
// float32 value float32(sqrt(float64(nb))) widened, negative if nb == 0 cannot occur,
// so rn == 0 <=> nb == 0 only when the sqrt underflows; we store nb's zero-ness in the sign bit
template <int M> __device__ __forceinline__ float finalize(typename MT<M>::A acc, const QConst& qc, double rn) {
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

// stage the query into LDS in the metric's Q type, zero-padded to dim4*4
template <int M> __device__ __forceinline__ void stage_query(typename MT<M>::Q* q_lds, const float* q, uint32_t dim, uint32_t dim4) {
    for (uint32_t i = threadIdx.x; i < dim4 * 4; i += blockDim.x) q_lds[i] = i < dim ? (typename MT<M>::Q)q[i] : (typename MT<M>::Q)0;
}

// distance of the query (in LDS) to the row whose chunk c lives at p[c * stride4]
// QN: also accumulate the query's own squared norm in the reference's element order
// (distances.go:20 magnitudeA += a*a; adapter.go:118 normA += a*a).  It is an independent
// dependency chain, so riding along with a row's dot product costs no time; a wave does it
// on its first tile only.
template <int M, int U, bool QN>
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
    uint32_t c0 = 0;
    for (; c0 + U <= dim4; c0 += U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(&p[(size_t)(c0 + u) * stride4]);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const Q* qq = q_lds + (size_t)(c0 + u) * 4;
            acc1<M>(acc, qq[0], v[u].x); qn1(qq[0]); acc1<M>(acc, qq[1], v[u].y); qn1(qq[1]);
            acc1<M>(acc, qq[2], v[u].z); qn1(qq[2]); acc1<M>(acc, qq[3], v[u].w); qn1(qq[3]);
        }
    }
    for (; c0 < dim4; c0++) {
        f4 v = __builtin_nontemporal_load(&p[(size_t)c0 * stride4]);
        const Q* qq = q_lds + (size_t)c0 * 4;
        acc1<M>(acc, qq[0], v.x); qn1(qq[0]); acc1<M>(acc, qq[1], v.y); qn1(qq[1]);
        acc1<M>(acc, qq[2], v.z); qn1(qq[2]); acc1<M>(acc, qq[3], v.w); qn1(qq[3]);
    }
    if constexpr (QN) *qnorm2 = qa;       // zero padding of the query adds +0 terms: exact
    return acc;
}

template <int M> __device__ __forceinline__ QConst qconst_from_norm2(typename MT<M>::A n2) {
    QConst c; c.qn = 0.0; c.qn32 = 0.0f;
    if constexpr (M == QV_COSINE) c.qn = __builtin_sqrt(n2);
    else if constexpr (M == QV_COSINE_F32) { c.qn32 = (float)__builtin_sqrt((double)n2); c.qn = (double)n2; }
    return c;
}

// ---------------------------------------------------------------- flat scan --------
// grid = (workgroups, nq); each wave walks tiles gw, gw+tw, ... ; lane == row.
// Output: partial[(q*gridDim.x + blockIdx.x)*k + i] = workgroup's i-th best key.
constexpr int kScanBlock = 256;
constexpr int kScanWaves = kScanBlock / 64;

template <int M, int U>
__global__ void __launch_bounds__(kScanBlock)
k_flat_scan(IndexView v, const float* __restrict__ queries, uint32_t k, uint64_t* __restrict__ partial) {
    using Q = typename MT<M>::Q;
    extern __shared__ __align__(16) unsigned char smem[];
    Q* q_lds = reinterpret_cast<Q*>(smem);
    uint64_t* wl = reinterpret_cast<uint64_t*>(smem + (((size_t)v.dim4 * 4 * sizeof(Q)) + 15) / 16 * 16);  // [kScanWaves][64]

    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t qi = blockIdx.y;
    stage_query<M>(q_lds, queries + (size_t)qi * v.dim, v.dim, v.dim4);
    __syncthreads();

    const uint32_t tw = gridDim.x * kScanWaves;
    const uint32_t kth = k - 1;
    uint64_t list = kDeadKey, thr = kDeadKey;
    const f4* tiles = reinterpret_cast<const f4*>(v.tiles);
    QConst qc; qc.qn = 0.0; qc.qn32 = 0.0f;

    auto finish_tile = [&](uint32_t t, typename MT<M>::A acc, bool first) {
        const uint32_t row = t * 64 + lane;
        double rn = 0.0;
        if constexpr (MT<M>::needs_rnorm) rn = v.rnorm[row];
        float dist = finalize<M>(acc, qc, rn);
        uint64_t am = v.alive[t];                                     // wave-uniform
        uint64_t key = ((am >> lane) & 1ull) ? make_key(dist, row) : kDeadKey;
        if (first) { list = wave_sort64(key, lane); thr = readlane64(list, kth); }   // empty list: sort the tile outright
        else list_insert(list, thr, key, kth, lane);
    };

    uint32_t t = blockIdx.x * kScanWaves + wave;
    if (t < v.n_tiles) {                                              // first tile: query norm rides along
        typename MT<M>::A qn2 = 0;
        typename MT<M>::A acc = row_accumulate<M, U, true>(tiles + (size_t)t * v.dim4 * 64 + lane, 64, q_lds, v.dim4, &qn2);
        qc = qconst_from_norm2<M>(qn2);
        finish_tile(t, acc, true);
        t += tw;
    }
    for (; t < v.n_tiles; t += tw) {
        typename MT<M>::A acc = row_accumulate<M, U, false>(tiles + (size_t)t * v.dim4 * 64 + lane, 64, q_lds, v.dim4);
        finish_tile(t, acc, false);
    }

    // workgroup merge: waves 1.. hand their lists to wave 0 through LDS
    wl[wave * 64 + lane] = list;
    __syncthreads();
    if (wave == 0) {
        for (uint32_t w = 1; w < kScanWaves; w++) {
            uint64_t key = lane < k ? wl[w * 64 + lane] : kDeadKey;
            list_insert(list, thr, key, kth, lane);
        }
        if (lane < k) partial[((size_t)qi * gridDim.x + blockIdx.x) * k + lane] = list;
    }
}

// ---------------------------------------------------------------- multi-query scan --
// QB queries share ONE pass over the corpus (HybridIndex.BatchSearch, hybrid_index.go:677-811,
// is Q independent exact searches; here every 16-byte row chunk a lane loads is used for QB
// dot products).  Same arithmetic contract: lane == row, each (row, query) distance is one
// sequential chain over dims 0..D-1.  The query block sits in LDS interleaved by query
// (q_lds[dim][QB]) so one ds_read_b128 feeds two (f64) or four (f32) queries of one dim.
// grid = (workgroups, ceil(nq/QB)); partial layout identical to k_flat_scan.
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

// query blocks for the scalar-operand variant: qblk[group][dim4*4][QB] in the metric's Q type
template <int M, int QB>
__global__ void k_prep_qblk(const float* __restrict__ queries, uint32_t nq, uint32_t dim, uint32_t dim4, typename MT<M>::Q* __restrict__ qblk) {
    using Q = typename MT<M>::Q;
    const uint32_t per = dim4 * 4 * QB;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per) return;
    const uint32_t d = i / QB, qq = i % QB, q0 = blockIdx.y * QB;
    const uint32_t qi = q0 + qq < nq ? q0 + qq : nq - 1;
    qblk[(size_t)blockIdx.y * per + i] = d < dim ? (Q)queries[(size_t)qi * dim + d] : (Q)0;
}

// SQ = true: the query block is read from GLOBAL memory at wave-uniform addresses, which the
// compiler turns into scalar loads (s_load) and SGPR operands of v_fma_f64 — the LDS, which
// bounds the LDS-staged form (one broadcast ds_read_b128 per 2 query values), is not touched.
template <int M, int U, int QB, bool SQ>
__global__ void __launch_bounds__(kScanBlock, 2)
k_flat_scan_mq(IndexView v, const float* __restrict__ queries, const typename MT<M>::Q* __restrict__ qblk, uint32_t nq, uint32_t k,
               uint64_t* __restrict__ partial) {
    using Q = typename MT<M>::Q;
    using A = typename MT<M>::A;
    extern __shared__ __align__(16) unsigned char smem[];
    const size_t q_bytes = SQ ? 0 : (((size_t)v.dim4 * 4 * QB * sizeof(Q)) + 15) / 16 * 16;
    uint64_t* wl = reinterpret_cast<uint64_t*>(smem + q_bytes);                  // [waves][QB][64]
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t q0 = blockIdx.y * QB;
    const Q* q_lds;
    if constexpr (SQ) {
        q_lds = qblk + (size_t)blockIdx.y * v.dim4 * 4 * QB;                     // global, uniform -> scalar loads
    } else {
        Q* ql = reinterpret_cast<Q*>(smem);                                      // [dim4*4][QB]
        // stage QB queries, zero-padded in dim; query slots past nq replicate the last query (results dropped)
        for (uint32_t i = threadIdx.x; i < v.dim4 * 4 * QB; i += blockDim.x) {
            uint32_t d = i / QB, qq = i % QB;
            uint32_t qi = q0 + qq < nq ? q0 + qq : nq - 1;
            ql[i] = d < v.dim ? (Q)queries[(size_t)qi * v.dim + d] : (Q)0;
        }
        __syncthreads();
        q_lds = ql;
    }

    const uint32_t tw = gridDim.x * kScanWaves;
    const uint32_t kth = k - 1;
    uint64_t list[QB], thr[QB];
    QConst qc[QB];
#pragma unroll
    for (int j = 0; j < QB; j++) { list[j] = kDeadKey; thr[j] = kDeadKey; qc[j].qn = 0.0; qc[j].qn32 = 0.0f; }
    const f4* tiles = reinterpret_cast<const f4*>(v.tiles);

    uint32_t t = blockIdx.x * kScanWaves + wave;
    if (t < v.n_tiles) {                                                         // first tile: sort outright, query norms ride along
        A acc[QB], qa[QB];
        mq_tile<M, U, QB, true>(tiles + (size_t)t * v.dim4 * 64 + lane, q_lds, v.dim4, acc, qa);
        const uint32_t row = t * 64 + lane;
        double rn = 0.0;
        if constexpr (MT<M>::needs_rnorm) rn = v.rnorm[row];
        const bool live = (v.alive[t] >> lane) & 1ull;
#pragma unroll
        for (int j = 0; j < QB; j++) {
            qc[j] = qconst_from_norm2<M>(qa[j]);
            float dist = finalize<M>(acc[j], qc[j], rn);
            list[j] = wave_sort64(live ? make_key(dist, row) : kDeadKey, lane);
            thr[j] = readlane64(list[j], kth);
        }
        t += tw;
    }
    for (; t < v.n_tiles; t += tw) {
        A acc[QB], qa[QB];
        mq_tile<M, U, QB, false>(tiles + (size_t)t * v.dim4 * 64 + lane, q_lds, v.dim4, acc, qa);
        const uint32_t row = t * 64 + lane;
        double rn = 0.0;
        if constexpr (MT<M>::needs_rnorm) rn = v.rnorm[row];
        const bool live = (v.alive[t] >> lane) & 1ull;
#pragma unroll
        for (int j = 0; j < QB; j++) {
            float dist = finalize<M>(acc[j], qc[j], rn);
            list_insert(list[j], thr[j], live ? make_key(dist, row) : kDeadKey, kth, lane);
        }
    }

#pragma unroll
    for (int j = 0; j < QB; j++) wl[((size_t)wave * QB + j) * 64 + lane] = list[j];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < QB; j++) {
            for (uint32_t w = 1; w < kScanWaves; w++) {
                uint64_t key = lane < k ? wl[((size_t)w * QB + j) * 64 + lane] : kDeadKey;
                list_insert(list[j], thr[j], key, kth, lane);
            }
            if (q0 + j < nq && lane < k) partial[((size_t)(q0 + j) * gridDim.x + blockIdx.x) * k + lane] = list[j];
        }
    }
}

// One workgroup per query merges n_lists sorted lists of k keys into the final top-k.
// Bound trick: the smallest k-th entry over all lists, B, is an upper bound of the final
// k-th key (that list alone holds k keys <= B), so only keys <= B can be in the answer.
// Typically a few dozen of the n_lists*k keys survive; one wave insertion-sorts them.
constexpr int kMergeBlock = 1024;
constexpr int kMergeCap = 2048;                       // survivors kept in LDS; more -> general path
constexpr int kMergeHeads = 128;                      // sampled list heads ranked in LDS

__device__ __forceinline__ uint64_t wave_min64(uint64_t x) {
#pragma unroll
    for (int off = 32; off; off >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)x, off), hi = __shfl_xor((uint32_t)(x >> 32), off);
        uint64_t y = ((uint64_t)hi << 32) | lo;
        x = y < x ? y : x;
    }
    return x;
}

__global__ void __launch_bounds__(kMergeBlock)
k_merge_lists(const uint64_t* __restrict__ partial, uint32_t n_lists, uint32_t k,
              uint32_t* __restrict__ rows_out, float* __restrict__ dist_out) {
    __shared__ uint64_t wl[kMergeBlock / 64][64];
    __shared__ uint64_t surv[kMergeCap];
    __shared__ uint32_t hd[kMergeHeads], hlt[kMergeHeads], hle[kMergeHeads];
    __shared__ uint64_t s_bound, s_b1;
    __shared__ uint32_t s_nsurv;
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nw = blockDim.x >> 6;
    const uint32_t qi = blockIdx.x;
    const uint64_t* src = partial + (size_t)qi * n_lists * k;
    const uint32_t total = n_lists * k;
    const uint32_t kth = k - 1;

    // every global load of the common case is issued up front (one HBM/L2 latency, not three):
    // this thread's <= 8 keys, one list's k-th key, one sampled list head
    const bool small = total <= blockDim.x * 8;
    uint64_t mine[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { uint32_t i = u * blockDim.x + threadIdx.x; mine[u] = (small && i < total) ? src[i] : kDeadKey; }
    // sampled heads: m = min(n_lists, kMergeHeads) lists at a fixed stride.  Any k different
    // lists each hold a key <= the k-th smallest of their heads, so a subset still gives a
    // valid (slightly looser) bound, and the O(m^2) rank count stays ~0.5 us on one CU.
    const uint32_t m = n_lists < (uint32_t)kMergeHeads ? n_lists : (uint32_t)kMergeHeads;
    const uint32_t hstride = n_lists / m;
    const bool use_heads = m >= k;
    uint64_t b = kDeadKey;
    for (uint32_t w = threadIdx.x; w < n_lists; w += blockDim.x) { uint64_t x = src[(size_t)w * k + kth]; b = x < b ? x : b; }
    if (use_heads)
        for (uint32_t w = threadIdx.x; w < m; w += blockDim.x) { hd[w] = (uint32_t)(src[(size_t)w * hstride * k] >> 32); hlt[w] = 0; hle[w] = 0; }

    // phase A: two upper bounds of the final k-th key.
    //   B0 = min over lists of their k-th key (that list alone has k keys <= B0);
    //   B1 = from the k-th smallest sampled HEAD — the tight one when the winners are spread
    //        over many lists, which is the common case.  Rank counting on the 32 distance bits:
    //        head i qualifies when #{j: d_j < d_i} <= k-1 < #{j: d_j <= d_i}; then every key
    //        with distance <= d_i is kept.
    b = wave_min64(b);
    if (lane == 0) wl[wave][0] = b;
    if (threadIdx.x == 0) { s_nsurv = 0; s_b1 = kDeadKey; }
    __syncthreads();
    if (use_heads) {
        const uint32_t segs = blockDim.x >= m ? blockDim.x / m : 1;   // thread -> (head i, segment of j)
        const uint32_t per = (m + segs - 1) / segs;
        for (uint32_t i = threadIdx.x % m, sgm = blockDim.x >= m ? threadIdx.x / m : 0; sgm < segs && i < m; i += blockDim.x) {
            const uint32_t h = hd[i];
            uint32_t clt = 0, cle = 0;
            const uint32_t j0 = sgm * per, j1 = min(j0 + per, m);
            uint32_t j = j0;
            for (; j + 16 <= j1; j += 16) {                           // batch the (broadcast) LDS reads
                uint32_t x[16];
#pragma unroll
                for (int u = 0; u < 16; u++) x[u] = hd[j + u];
#pragma unroll
                for (int u = 0; u < 16; u++) { clt += x[u] < h ? 1u : 0u; cle += x[u] <= h ? 1u : 0u; }
            }
            for (; j < j1; j++) { uint32_t x = hd[j]; clt += x < h ? 1u : 0u; cle += x <= h ? 1u : 0u; }
            if (clt) atomicAdd(&hlt[i], clt);
            if (cle) atomicAdd(&hle[i], cle);
            if (blockDim.x >= m) break;
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < m; i += blockDim.x)
            if (hlt[i] <= kth && kth < hle[i] && hd[i] != 0xFFFFFFFFu) s_b1 = ((uint64_t)hd[i] << 32) | 0xFFFFFFFFull;
    }
    if (wave == 0) {
        uint64_t x = lane < nw ? wl[lane][0] : kDeadKey;
        x = wave_min64(x);
        if (lane == 0) s_bound = x;
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_b1 < s_bound) s_bound = s_b1;
    __syncthreads();
    const uint64_t bound = s_bound;
    // phase B: keep keys <= bound
    auto keep = [&](uint64_t key) {
        if (key != kDeadKey && key <= bound) {
            uint32_t pos = atomicAdd(&s_nsurv, 1u);
            if (pos < (uint32_t)kMergeCap) surv[pos] = key;
        }
    };
    if (small) {
#pragma unroll
        for (int u = 0; u < 8; u++) keep(mine[u]);
    } else {
        for (uint32_t base = 0; base < total; base += blockDim.x * 8) {
            uint64_t key[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { uint32_t i = base + u * blockDim.x + threadIdx.x; key[u] = i < total ? src[i] : kDeadKey; }
#pragma unroll
            for (int u = 0; u < 8; u++) keep(key[u]);
        }
    }
    __syncthreads();
    const uint32_t ns = s_nsurv;
    uint64_t list = kDeadKey, thr = kDeadKey;
    if (ns <= 64) {
        // phase C (common): one wave bitonic-sorts the survivors
        if (wave != 0) return;
        list = wave_sort64(lane < ns ? surv[lane] : kDeadKey, lane);
    } else if (ns <= (uint32_t)kMergeCap) {
        if (wave != 0) return;
        list = wave_sort64(surv[lane], lane);
        thr = readlane64(list, kth);
        for (uint32_t base = 64; base < ns; base += 64) {
            uint32_t i = base + lane;
            uint64_t key = i < ns ? surv[i] : kDeadKey;
            list_insert(list, thr, key, kth, lane);
        }
    } else {
        // general path (tiny indexes whose lists are mostly shorter than k): every wave
        // reduces a slice, wave 0 merges the waves
        for (uint32_t base = wave * 64; base < total; base += nw * 64) {
            uint32_t i = base + lane;
            uint64_t key = i < total ? src[i] : kDeadKey;
            list_insert(list, thr, key, kth, lane);
        }
        wl[wave][lane] = list;
        __syncthreads();
        if (wave != 0) return;
        for (uint32_t w = 1; w < nw; w++) {
            uint64_t key = lane < k ? wl[w][lane] : kDeadKey;
            list_insert(list, thr, key, kth, lane);
        }
    }
    if (lane < k) {
        bool dead = list == kDeadKey;
        rows_out[(size_t)qi * k + lane] = dead ? 0xFFFFFFFFu : (uint32_t)list;
        dist_out[(size_t)qi * k + lane] = dead ? __uint_as_float(0x7F800000u) : unord_f32((uint32_t)(list >> 32));
    }
}

// merge of (distance, row) pair lists, e.g. the all-gathered per-shard top-k of a sharded scan
__global__ void __launch_bounds__(kMergeBlock)
k_merge_pairs(const float* __restrict__ dist, const uint32_t* __restrict__ rows, uint32_t total, uint32_t k,
              uint32_t* __restrict__ rows_out, float* __restrict__ dist_out) {
    __shared__ uint64_t wl[kMergeBlock / 64][64];
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nw = blockDim.x >> 6;
    const uint32_t kth = k - 1;
    uint64_t list = kDeadKey, thr = kDeadKey;
    for (uint32_t base = wave * 64; base < total; base += nw * 64) {
        uint32_t i = base + lane;
        uint64_t key = kDeadKey;
        if (i < total && rows[i] != 0xFFFFFFFFu) key = make_key(dist[i], rows[i]);
        list_insert(list, thr, key, kth, lane);
    }
    wl[wave][lane] = list;
    __syncthreads();
    if (wave == 0) {
        for (uint32_t w = 1; w < nw; w++) {
            uint64_t key = lane < k ? wl[w][lane] : kDeadKey;
            list_insert(list, thr, key, kth, lane);
        }
        if (lane < k) {
            bool dead = list == kDeadKey;
            rows_out[lane] = dead ? 0xFFFFFFFFu : (uint32_t)list;
            dist_out[lane] = dead ? __uint_as_float(0x7F800000u) : unord_f32((uint32_t)(list >> 32));
        }
    }
}

// ---------------------------------------------------------------- full ranking -----
// all keys: keys[row] = (ord(dist), row) or dead
template <int M, int U>
__global__ void __launch_bounds__(kScanBlock)
k_flat_keys(IndexView v, const float* __restrict__ query, uint64_t* __restrict__ keys) {
    using Q = typename MT<M>::Q;
    extern __shared__ __align__(16) unsigned char smem[];
    Q* q_lds = reinterpret_cast<Q*>(smem);
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    stage_query<M>(q_lds, query, v.dim, v.dim4);
    __syncthreads();
    const QConst qc = query_const<M>(q_lds, v.dim);
    const uint32_t tw = gridDim.x * kScanWaves;
    const f4* tiles = reinterpret_cast<const f4*>(v.tiles);
    for (uint32_t t = blockIdx.x * kScanWaves + wave; t < v.n_tiles; t += tw) {
        const f4* p = tiles + (size_t)t * v.dim4 * 64 + lane;
        typename MT<M>::A acc = row_accumulate<M, U, false>(p, 64, q_lds, v.dim4);
        const uint32_t row = t * 64 + lane;
        double rn = 0.0;
        if constexpr (MT<M>::needs_rnorm) rn = v.rnorm[row];
        float dist = finalize<M>(acc, qc, rn);
        uint64_t am = v.alive[t];
        keys[row] = ((am >> lane) & 1ull) ? make_key(dist, row) : kDeadKey;
    }
}

// LSD radix sort of 64-bit keys, 8 bits per pass over the 32 distance bits only (the
// row bits are already ascending in the input and every pass is stable, so equal
// distances stay in row order).  Three kernels per pass: histogram, scan, scatter.
constexpr int kRadixBlock = 256;
constexpr int kRadixItems = 16;                       // keys per thread
constexpr int kRadixTile = kRadixBlock * kRadixItems; // keys per workgroup

__global__ void __launch_bounds__(kRadixBlock)
k_radix_hist(const uint64_t* __restrict__ keys, uint32_t n, uint32_t shift, uint32_t* __restrict__ hist /*[256][nblocks]*/) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kRadixTile;
    for (int i = 0; i < kRadixItems; i++) {
        uint32_t idx = base + i * kRadixBlock + threadIdx.x;
        if (idx < n) atomicAdd(&h[(uint32_t)(keys[idx] >> shift) & 0xFF], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of the digit-major histogram [256][nblocks], two small kernels:
// (1) one workgroup per digit turns its row into within-digit exclusive prefixes and a digit total,
// (2) one workgroup scans the 256 totals.  (A single-workgroup scan of the whole table was 92 us
// per pass at 1M keys — most of the sort.)
__global__ void __launch_bounds__(256)
k_radix_scan_digits(uint32_t* __restrict__ hist, uint32_t nblocks, uint32_t* __restrict__ dtot) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t s_run;
    uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t x = i < nblocks ? row[i] : 0;
        uint32_t inc = x;                                              // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { uint32_t y = __shfl_up(inc, off); if ((int)lane >= off) inc += y; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t before = s_run;
        for (uint32_t w = 0; w < wave; w++) before += wsum[w];
        if (i < nblocks) row[i] = before + inc - x;
        __syncthreads();
        if (threadIdx.x == 0) s_run += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) dtot[blockIdx.x] = s_run;
}
__global__ void __launch_bounds__(256)
k_radix_scan_totals(const uint32_t* __restrict__ dtot, uint32_t* __restrict__ dbase) {
    __shared__ uint32_t wsum[4];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t x = dtot[threadIdx.x];
    uint32_t inc = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { uint32_t y = __shfl_up(inc, off); if ((int)lane >= off) inc += y; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; w++) before += wsum[w];
    dbase[threadIdx.x] = before + inc - x;
}

// stable scatter: within a workgroup keys are ranked in index order
__global__ void __launch_bounds__(kRadixBlock)
k_radix_scatter(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t n, uint32_t shift,
                const uint32_t* __restrict__ hist, const uint32_t* __restrict__ dbase) {
    __shared__ uint32_t digit_base[256];                // global offset of this block's first key of each digit
    __shared__ uint32_t wave_cnt[kRadixBlock / 64][256]; // per-wave digit counts within one round
    __shared__ uint32_t running[256];                   // keys of each digit already placed by earlier rounds
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    digit_base[threadIdx.x] = dbase[threadIdx.x] + hist[(size_t)threadIdx.x * gridDim.x + blockIdx.x];
    running[threadIdx.x] = 0;
    const uint32_t base = blockIdx.x * kRadixTile;
    for (int i = 0; i < kRadixItems; i++) {             // rounds go in index order: round i covers base + i*256 ..
        for (int w = 0; w < kRadixBlock / 64; w++) wave_cnt[w][threadIdx.x] = 0;
        __syncthreads();
        uint32_t idx = base + i * kRadixBlock + threadIdx.x;
        bool valid = idx < n;
        uint64_t key = valid ? in[idx] : 0;
        uint32_t d = (uint32_t)(key >> shift) & 0xFF;
        // rank among lanes of this wave with the same digit (match-any by 8 ballots)
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            uint64_t m = __ballot((d >> b) & 1);
            peers &= ((d >> b) & 1) ? m : ~m;
        }
        uint32_t rank_in_wave = (uint32_t)__builtin_popcountll(peers & ((1ull << lane) - 1));
        uint32_t wave_total = (uint32_t)__builtin_popcountll(peers);
        if (valid && rank_in_wave == 0) wave_cnt[wave][d] = wave_total;
        __syncthreads();
        if (valid) {
            uint32_t before = 0;
            for (uint32_t w = 0; w < wave; w++) before += wave_cnt[w][d];
            out[digit_base[d] + running[d] + before + rank_in_wave] = key;
        }
        __syncthreads();
        uint32_t tot = 0;
        for (int w = 0; w < kRadixBlock / 64; w++) tot += wave_cnt[w][threadIdx.x];
        running[threadIdx.x] += tot;
        __syncthreads();
    }
}

__global__ void k_emit_topk(const uint64_t* __restrict__ keys, uint32_t n, uint32_t k, uint32_t* rows_out, float* dist_out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    uint64_t key = i < n ? keys[i] : kDeadKey;
    bool dead = key == kDeadKey;
    rows_out[i] = dead ? 0xFFFFFFFFu : (uint32_t)key;
    dist_out[i] = dead ? __uint_as_float(0x7F800000u) : unord_f32((uint32_t)(key >> 32));
}

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

// ---------------------------------------------------------------- gathers ----------
// lane == listed row; one wave per 64 listed rows
template <int M, int U>
__global__ void __launch_bounds__(64)
k_distance_rows(IndexView v, const float* __restrict__ query, const uint32_t* __restrict__ rows, uint32_t n, float* __restrict__ out) {
    using Q = typename MT<M>::Q;
    extern __shared__ __align__(16) unsigned char smem[];
    Q* q_lds = reinterpret_cast<Q*>(smem);
    stage_query<M>(q_lds, query, v.dim, v.dim4);
    __syncthreads();
    const QConst qc = query_const<M>(q_lds, v.dim);
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const uint32_t row = rows[i];
    if (row >= v.n_rows) { out[i] = __uint_as_float(0x7FC00000u); return; }
    const f4* p = reinterpret_cast<const f4*>(v.tiles) + (size_t)(row >> 6) * v.dim4 * 64 + (row & 63);
    typename MT<M>::A acc = row_accumulate<M, U, false>(p, 64, q_lds, v.dim4);
    double rn = 0.0;
    if constexpr (MT<M>::needs_rnorm) rn = v.rnorm[row];
    out[i] = finalize<M>(acc, qc, rn);
}

// lane == pair; a, b row-major [n][dim]; plain scalar walk (dim need not be a multiple of 4)
template <int M>
__global__ void __launch_bounds__(64)
k_distance_pairs(const float* __restrict__ a, const float* __restrict__ b, uint32_t n, uint32_t dim, float* __restrict__ out) {
    using Q = typename MT<M>::Q;
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float* pa = a + (size_t)i * dim;
    const float* pb = b + (size_t)i * dim;
    typename MT<M>::A acc = 0;
    QConst qc; qc.qn = 0.0; qc.qn32 = 0.0f;
    double rn = 0.0;
    if constexpr (M == QV_COSINE) {
        double ma = 0.0, mb = 0.0;
        for (uint32_t j = 0; j < dim; j++) {
            double x = pa[j], y = pb[j];
            acc = __builtin_fma(x, y, acc); ma = __builtin_fma(x, x, ma); mb = __builtin_fma(y, y, mb);
        }
        qc.qn = __builtin_sqrt(ma); rn = __builtin_sqrt(mb);
    } else if constexpr (M == QV_COSINE_F32) {
        float na = 0.0f, nb = 0.0f;
        for (uint32_t j = 0; j < dim; j++) {
            float x = pa[j], y = pb[j];
            float p0 = x * y; acc = acc + p0; float p1 = x * x; na = na + p1; float p2 = y * y; nb = nb + p2;
        }
        qc.qn = (double)na; qc.qn32 = (float)__builtin_sqrt((double)na);
        rn = nb == 0.0f ? -1.0 : (double)(float)__builtin_sqrt((double)nb);
    } else {
        for (uint32_t j = 0; j < dim; j++) acc1<M>(acc, (Q)pa[j], pb[j]);
    }
    out[i] = finalize<M>(acc, qc, rn);
}

// ---------------------------------------------------------------- HNSW traversal ----
// Device-resident restatement of hnsw.HNSW.Search (pkg/hnsw/hnsw.go:602-713) and searchLayer
// (:471-580): one wavefront walks the graph for one query; many queries are in flight (the
// path is latency-bound per query).  To return exactly what the reference
// returns, the two heaps are the reference's binary heaps with its own sift loops
// (hnsw.go:101-196), executed by lane 0 on LDS arrays, and a hop's neighbours are admitted
// one by one in adjacency order (:536-563) after their distances have been computed together:
// lane i scores the i-th unvisited neighbour with the same sequential-over-dims arithmetic
// as every other kernel here, so distances — hence every heap decision — are bit-identical
// to the CPU restatement.
struct HRes { float dist; uint32_t idx; };
constexpr int kHnswCandCap = 2048;     // candidate min-heap slots per query (overflow -> host path)
constexpr int kHnswEfMax = 512;
constexpr int kHnswMaxDeg = 64;

__device__ __forceinline__ void h_min_up(HRes* rs, int j) {                      // hnsw.go:118-128
    for (;;) { int i = (j - 1) / 2; if (i == j || rs[j].dist >= rs[i].dist) break; HRes t = rs[i]; rs[i] = rs[j]; rs[j] = t; j = i; }
}
__device__ __forceinline__ void h_min_down(HRes* rs, int i0, int n) {            // hnsw.go:130-148
    int i = i0;
    for (;;) {
        int j1 = 2 * i + 1; if (j1 >= n || j1 < 0) break;
        int j = j1, j2 = j1 + 1; if (j2 < n && rs[j2].dist < rs[j1].dist) j = j2;
        if (rs[i].dist <= rs[j].dist) break;
        HRes t = rs[i]; rs[i] = rs[j]; rs[j] = t; i = j;
    }
}
__device__ __forceinline__ void h_max_up(HRes* rs, int j) {                      // hnsw.go:172-181
    for (;;) { int i = (j - 1) / 2; if (i == j || rs[j].dist <= rs[i].dist) break; HRes t = rs[i]; rs[i] = rs[j]; rs[j] = t; j = i; }
}
__device__ __forceinline__ void h_max_down(HRes* rs, int i0, int n) {            // hnsw.go:183-200
    int i = i0;
    for (;;) {
        int j1 = 2 * i + 1; if (j1 >= n || j1 < 0) break;
        int j = j1, j2 = j1 + 1; if (j2 < n && rs[j2].dist > rs[j1].dist) j = j2;
        if (rs[i].dist >= rs[j].dist) break;
        HRes t = rs[i]; rs[i] = rs[j]; rs[j] = t; i = j;
    }
}

// one wave (64-thread workgroup) per query stream
template <int M, int U>
__global__ void __launch_bounds__(64)
k_hnsw_search(IndexView v, GraphView g, const float* __restrict__ queries, uint32_t nq, uint32_t k, uint32_t ef_search,
              uint32_t* __restrict__ visited /*[gridDim.x][n_nodes]*/, uint32_t epoch0,
              uint32_t* __restrict__ rows_out, float* __restrict__ dist_out, uint32_t* __restrict__ count_out, uint32_t* __restrict__ evals_out) {
    using Q = typename MT<M>::Q;
    extern __shared__ __align__(16) unsigned char smem[];
    Q* q_lds = reinterpret_cast<Q*>(smem);
    unsigned char* base = smem + (((size_t)v.dim4 * 4 * sizeof(Q)) + 15) / 16 * 16;
    HRes* cand = reinterpret_cast<HRes*>(base);                       // [kHnswCandCap]
    HRes* res = cand + kHnswCandCap;                                  // [kHnswEfMax + 1]
    uint32_t* batch = reinterpret_cast<uint32_t*>(res + kHnswEfMax + 1);   // [kHnswMaxDeg]
    float* bd = reinterpret_cast<float*>(batch + kHnswMaxDeg);        // [kHnswMaxDeg]
    __shared__ int s_ncand, s_nres, s_state;                           // state: 0 run, 1 done, 2 overflow
    __shared__ uint32_t s_cur;
    const uint32_t lane = threadIdx.x;
    uint32_t* vis = visited + (size_t)blockIdx.x * g.n_nodes;
    uint32_t epoch = epoch0;
    const bool use_rm = v.rowmaj != nullptr && (v.dim & 3) == 0;

    auto alive = [&](uint32_t n) -> bool { return n < g.n_nodes && g.level[n] >= 0; };
    // distances of batch[0..n) -> bd[0..n); lane i scores batch[i]
    QConst qc;
    auto eval = [&](uint32_t n) {
        if (lane < n) {
            const uint32_t row = batch[lane];
            typename MT<M>::A acc;
            if (use_rm) acc = row_accumulate<M, U, false>(reinterpret_cast<const f4*>(v.rowmaj + (size_t)row * v.dim), 1, q_lds, v.dim4);
            else acc = row_accumulate<M, U, false>(reinterpret_cast<const f4*>(v.tiles) + (size_t)(row >> 6) * v.dim4 * 64 + (row & 63), 64, q_lds, v.dim4);
            double rn = 0.0;
            if constexpr (MT<M>::needs_rnorm) rn = v.rnorm[row];
            bd[lane] = finalize<M>(acc, qc, rn);
        }
        __syncthreads();
    };
    // searchLayer (hnsw.go:471-580); result: res[0..s_nres) ascending; returns false on overflow
    uint32_t n_eval = 0;
    auto search_layer = [&](uint32_t entry, int ef, int level) -> bool {
        epoch++;
        if (lane == 0) { vis[entry] = epoch; batch[0] = entry; }
        __syncthreads();
        eval(1); n_eval += 1;                                          // :492
        if (lane == 0) {
            cand[0] = {bd[0], entry}; res[0] = {bd[0], entry};        // :498-506
            s_ncand = 1; s_nres = 1; s_state = 0;
        }
        __syncthreads();
        for (;;) {
            if (lane == 0) {
                int nc = s_ncand, nr = s_nres;
                if (nc == 0) s_state = 1;                              // :509
                else {
                    nc--; HRes t = cand[0]; cand[0] = cand[nc]; cand[nc] = t; h_min_down(cand, 0, nc); HRes cur = cand[nc];   // :511 pop
                    s_ncand = nc;
                    if (nr >= ef && cur.dist > res[0].dist) s_state = 1;   // :514-516
                    else s_cur = cur.idx;
                }
            }
            __syncthreads();
            if (s_state != 0) break;
            const uint32_t cur = s_cur;
            // neighbours of cur at `level` (:523-534)
            uint32_t deg = 0; const uint32_t* links = nullptr;
            if (alive(cur) && level <= (int)g.level[cur]) {
                if (level == 0) { deg = g.l0_deg[cur]; links = g.l0_links + (size_t)cur * g.max_m0; }
                else { const uint32_t* blk = g.up_links + (size_t)(g.up_off[cur] + (uint32_t)(level - 1)) * (1 + g.max_m); deg = blk[0]; links = blk + 1; }
            }
            uint32_t c = 0xFFFFFFFFu; bool fresh = false;
            if (lane < deg) {
                c = links[lane];
                fresh = alive(c) && vis[c] != epoch;                   // :539-543
            }
            // a list may hold the same node twice (the self-link quirk): only its first occurrence is new
            for (uint32_t j = 0; j + 1 < deg; j++) {
                uint32_t cj = __builtin_amdgcn_readlane(c, j);
                if (lane > j && c == cj) fresh = false;
            }
            const uint64_t fm = __ballot(fresh);
            const uint32_t n = (uint32_t)__builtin_popcountll(fm);
            if (fresh) { vis[c] = epoch; batch[__builtin_popcountll(fm & ((1ull << lane) - 1))] = c; }   // :544, adjacency order kept
            __syncthreads();
            if (n == 0) continue;
            eval(n); n_eval += n;                                      // :548 (batched)
            if (lane == 0) {
                int nc = s_ncand, nr = s_nres;
                for (uint32_t i = 0; i < n; i++) {
                    const float cd = bd[i];
                    if (nr < ef || cd < res[0].dist) {                 // :553
                        if (nc >= kHnswCandCap) { s_state = 2; break; }
                        cand[nc] = {cd, batch[i]}; h_min_up(cand, nc); nc++;          // :554
                        res[nr] = {cd, batch[i]}; h_max_up(res, nr); nr++;            // :555
                        if (nr > ef) { nr--; HRes t = res[0]; res[0] = res[nr]; res[nr] = t; h_max_down(res, 0, nr); }   // :558-560
                    }
                }
                s_ncand = nc; s_nres = nr;
            }
            __syncthreads();
            if (s_state == 2) return false;
        }
        if (lane == 0) {                                               // :566-577 heap -> ascending slice, in place
            int nr = s_nres;
            for (int m = nr; m > 1; m--) { HRes t = res[0]; res[0] = res[m - 1]; res[m - 1] = t; h_max_down(res, 0, m - 1); }
        }
        __syncthreads();
        return true;
    };

    for (uint32_t qi = blockIdx.x; qi < nq; qi += gridDim.x) {
        stage_query<M>(q_lds, queries + (size_t)qi * v.dim, v.dim, v.dim4);
        __syncthreads();
        qc = query_const<M>(q_lds, v.dim);
        n_eval = 0;
        uint32_t entry = g.entry;
        bool ok = true;
        for (int level = g.cur_level; level > 0 && ok; level--) {       // :649-657
            ok = search_layer(entry, 1, level);
            if (ok && s_nres > 0) entry = res[0].idx;
            __syncthreads();
        }
        const int ef = (int)ef_search > (int)k ? (int)ef_search : (int)k;   // :660-663
        if (ok) ok = search_layer(entry, ef, 0);                        // :664
        uint32_t cnt = 0xFFFFFFFFu;                                     // overflow marker
        if (ok) {
            cnt = (uint32_t)s_nres < k ? (uint32_t)s_nres : k;          // :670-672 (under-filled: the caller tops up, :676-710)
            for (uint32_t i = lane; i < k; i += 64) {
                rows_out[(size_t)qi * k + i] = i < cnt ? res[i].idx : 0xFFFFFFFFu;
                dist_out[(size_t)qi * k + i] = i < cnt ? res[i].dist : __uint_as_float(0x7F800000u);
            }
        }
        if (lane == 0) { count_out[qi] = cnt; if (evals_out) evals_out[qi] = n_eval; }
        __syncthreads();
        epoch += 64;                                                    // distinct epochs for the next query of this wave
    }
}

// ---------------------------------------------------------------- HNSW traversal, wave-resident form
// Same traversal, without the serial LDS heaps.  Observation (no two entries of equal distance):
//   * a node enters the candidate heap exactly when it enters the result heap (hnsw.go:553-555);
//   * it leaves the result heap only by eviction, and an evicted node (distance > results.top from
//     then on) is never expanded: when it is popped the loop stops (hnsw.go:514-516) and every
//     other remaining candidate is no better;
//   * sequential admission of a hop's neighbours (hnsw.go:553-560) leaves the ef smallest of
//     (old results + neighbours), whatever the order.
// So the whole searchLayer state is ONE ascending list of <= ef (distance, node) keys with an
// "expanded" bit each: pop = first unexpanded entry; admit = sorted insert, drop the (ef+1)-th.
// The list lives in registers, S keys per lane (ef <= 64*S), and is updated with ballots, popcounts,
// readlanes and DPP wave shifts — no LDS, so ~4x more queries are resident and a hop's serial part
// shrinks from ~40 us to ~1 us.  With equal distances the binary heaps' pop order depends on their
// layout, so a query that ever sees two equal distances in the list (or a NaN) is flagged and
// re-run by k_hnsw_search (the exact-heap form): results stay identical to the reference in all cases.
constexpr uint32_t kHnswTieFlag = 0xFFFFFFFEu;
constexpr int kHnswStageRows = 8;         // neighbour rows staged in LDS per round

// a row staged in LDS (contiguous f4 chunks), same sequential-over-dims arithmetic as row_accumulate
template <int M>
__device__ __forceinline__ typename MT<M>::A row_accumulate_lds(const f4* p, const typename MT<M>::Q* __restrict__ q_lds, uint32_t dim4) {
    typename MT<M>::A acc = 0;
#pragma unroll 4
    for (uint32_t c = 0; c < dim4; c++) {
        const f4 x = p[c];
        const typename MT<M>::Q* qq = q_lds + (size_t)c * 4;
        acc1<M>(acc, qq[0], x.x); acc1<M>(acc, qq[1], x.y); acc1<M>(acc, qq[2], x.z); acc1<M>(acc, qq[3], x.w);
    }
    return acc;
}

template <int M, int U, int S>
__global__ void __launch_bounds__(64)
k_hnsw_search_wave(IndexView v, GraphView g, const float* __restrict__ queries, uint32_t nq, uint32_t k, uint32_t ef_search,
                   uint32_t* __restrict__ visited, uint32_t epoch0,
                   uint32_t* __restrict__ rows_out, float* __restrict__ dist_out, uint32_t* __restrict__ count_out, uint32_t* __restrict__ evals_out) {
    using Q = typename MT<M>::Q;
    extern __shared__ __align__(16) unsigned char smem[];
    Q* q_lds = reinterpret_cast<Q*>(smem);
    uint32_t* batch = reinterpret_cast<uint32_t*>(smem + (((size_t)v.dim4 * 4 * sizeof(Q)) + 15) / 16 * 16);   // [64]
    f4* stage = reinterpret_cast<f4*>(batch + 64);                    // [kHnswStageRows][dim4 + 1] (row-major index only)
    const uint32_t lane = threadIdx.x;
    uint32_t* vis = visited + (size_t)blockIdx.x * g.n_nodes;
    uint32_t epoch = epoch0;
    const bool use_rm = v.rowmaj != nullptr && (v.dim & 3) == 0;
    auto alive = [&](uint32_t n) -> bool { return n < g.n_nodes && g.level[n] >= 0; };

    QConst qc;
    uint64_t key[S];          // ascending over index e = s*64 + lane; kDeadKey = empty
    uint64_t expd[S];         // wave-uniform: bit l of expd[s] = entry (s,l) already expanded
    uint32_t n_list = 0; bool tie = false;
    uint32_t n_eval = 0;

    // distance of the query to batch[lane] for lane < n  ->  64-bit key (all lanes return; dead beyond n).
    // With the row-major copy the neighbour rows are fetched COOPERATIVELY — all 64 lanes read one
    // row's 16-byte chunks side by side (1 KiB per instruction, every load of a round in flight
    // together) into LDS, padded by one chunk per row so the per-lane reads below spread over the
    // banks — and then lane r walks ITS row sequentially from LDS.  (Each lane pulling its own row
    // straight from memory meant 24 dependent latency rounds of 13-way scattered 16-byte loads per hop.)
    auto eval_keys = [&](uint32_t n) -> uint64_t {
        uint64_t kx = kDeadKey;
        if (use_rm) {
            const uint32_t pitch = v.dim4 + 1;
            for (uint32_t base = 0; base < n; base += kHnswStageRows) {
                const uint32_t cnt = n - base < (uint32_t)kHnswStageRows ? n - base : (uint32_t)kHnswStageRows;
                __syncthreads();
                for (uint32_t r = 0; r < cnt; r++) {
                    const f4* src = reinterpret_cast<const f4*>(v.rowmaj + (size_t)batch[base + r] * v.dim);
                    for (uint32_t c = lane; c < v.dim4; c += 64) stage[(size_t)r * pitch + c] = src[c];
                }
                __syncthreads();
                if (lane >= base && lane < base + cnt) {
                    const uint32_t row = batch[lane];
                    typename MT<M>::A acc = row_accumulate_lds<M>(stage + (size_t)(lane - base) * pitch, q_lds, v.dim4);
                    double rn = 0.0;
                    if constexpr (MT<M>::needs_rnorm) rn = v.rnorm[row];
                    kx = make_key(finalize<M>(acc, qc, rn), row);
                }
            }
            return kx;
        }
        if (lane < n) {
            const uint32_t row = batch[lane];
            typename MT<M>::A acc = row_accumulate<M, U, false>(reinterpret_cast<const f4*>(v.tiles) + (size_t)(row >> 6) * v.dim4 * 64 + (row & 63), 64, q_lds, v.dim4);
            double rn = 0.0;
            if constexpr (MT<M>::needs_rnorm) rn = v.rnorm[row];
            kx = make_key(finalize<M>(acc, qc, rn), row);
        }
        return kx;
    };
    // sorted insert of x (distance part xd), list capacity ef
    auto insert = [&](uint64_t x, uint32_t ef) {
        const uint32_t xd = (uint32_t)(x >> 32);
        if (xd == 0xFFFFFFFEu) tie = true;                              // NaN: heap order is not a function of distances
        if (n_list >= ef) {
            const uint32_t e = ef - 1;
            uint64_t top = 0;
#pragma unroll
            for (int s2 = 0; s2 < S; s2++) if ((int)(e >> 6) == s2) top = readlane64(key[s2], e & 63);
            if (xd >= (uint32_t)(top >> 32)) { if (xd == (uint32_t)(top >> 32)) tie = true; return; }   // hnsw.go:553 strict <
        }
        uint32_t p = 0;
#pragma unroll
        for (int s2 = 0; s2 < S; s2++) {
            const uint32_t kd = (uint32_t)(key[s2] >> 32);
            p += (uint32_t)__builtin_popcountll(__ballot(kd < xd));
            if (__ballot(kd == xd && key[s2] != kDeadKey)) tie = true;
        }
        const uint32_t s0 = p >> 6, l0 = p & 63;
        uint64_t carry = 0; uint64_t cbit = 0;
#pragma unroll
        for (int s2 = 0; s2 < S; s2++) {
            if ((uint32_t)s2 < s0) continue;
            const uint64_t last = readlane64(key[s2], 63);
            const uint64_t lastbit = (expd[s2] >> 63) & 1ull;
            const uint64_t up = wave_shr1(key[s2]);
            if ((uint32_t)s2 == s0) {
                key[s2] = lane < l0 ? key[s2] : (lane == l0 ? x : up);
                const uint64_t low = (1ull << l0) - 1;                   // bits below the insertion lane stay
                expd[s2] = (expd[s2] & low) | ((expd[s2] & ~low) << 1);  // the rest move up; bit l0 = 0: new entry unexpanded
            } else {
                key[s2] = lane == 0 ? carry : up;
                expd[s2] = (expd[s2] << 1) | cbit;
            }
            carry = last; cbit = lastbit;
        }
        if (n_list < ef) n_list++;
        // drop whatever sits past the capacity
        if (n_list == ef && ef < (uint32_t)(S * 64)) {
            const uint32_t e = ef;
#pragma unroll
            for (int s2 = 0; s2 < S; s2++) if ((int)(e >> 6) == s2) { if (lane == (e & 63)) key[s2] = kDeadKey; expd[s2] &= ~(1ull << (e & 63)); }
        }
    };

    // searchLayer (hnsw.go:471-580)
    auto search_layer = [&](uint32_t entry, uint32_t ef, int level) {
        epoch++;
#pragma unroll
        for (int s2 = 0; s2 < S; s2++) { key[s2] = kDeadKey; expd[s2] = 0; }
        n_list = 0;
        if (lane == 0) { vis[entry] = epoch; batch[0] = entry; }
        __syncthreads();
        uint64_t k0 = eval_keys(1); n_eval += 1;
        insert(readlane64(k0, 0), ef);
        for (;;) {
            // pop: first unexpanded entry
            uint32_t cur = 0xFFFFFFFFu;
#pragma unroll
            for (int s2 = 0; s2 < S; s2++) {
                if (cur != 0xFFFFFFFFu) continue;
                const uint64_t m = __ballot(key[s2] != kDeadKey) & ~expd[s2];
                if (m) { const uint32_t l = (uint32_t)__builtin_ctzll(m); cur = (uint32_t)readlane64(key[s2], l); expd[s2] |= 1ull << l; }
            }
            if (cur == 0xFFFFFFFFu) break;
            uint32_t deg = 0; const uint32_t* links = nullptr;
            if (alive(cur) && level <= (int)g.level[cur]) {
                if (level == 0) { deg = g.l0_deg[cur]; links = g.l0_links + (size_t)cur * g.max_m0; }
                else { const uint32_t* blk = g.up_links + (size_t)(g.up_off[cur] + (uint32_t)(level - 1)) * (1 + g.max_m); deg = blk[0]; links = blk + 1; }
            }
            uint32_t c = 0xFFFFFFFFu; bool fresh = false;
            if (lane < deg) { c = links[lane]; fresh = alive(c) && vis[c] != epoch; }
            for (uint32_t j = 0; j + 1 < deg; j++) {                     // repeated node in one list: first occurrence only
                uint32_t cj = __builtin_amdgcn_readlane(c, j);
                if (lane > j && c == cj) fresh = false;
            }
            const uint64_t fm = __ballot(fresh);
            const uint32_t nb = (uint32_t)__builtin_popcountll(fm);
            __syncthreads();                                             // previous hop's batch[] reads are done
            if (fresh) { vis[c] = epoch; batch[__builtin_popcountll(fm & ((1ull << lane) - 1))] = c; }
            __syncthreads();
            if (nb == 0) continue;
            const uint64_t kx = eval_keys(nb); n_eval += nb;
            for (uint32_t i = 0; i < nb; i++) insert(readlane64(kx, i), ef);
        }
    };

    for (uint32_t qi = blockIdx.x; qi < nq; qi += gridDim.x) {
        __syncthreads();
        stage_query<M>(q_lds, queries + (size_t)qi * v.dim, v.dim, v.dim4);
        __syncthreads();
        qc = query_const<M>(q_lds, v.dim);
        n_eval = 0; tie = false;
        uint32_t entry = g.entry;
        for (int level = g.cur_level; level > 0; level--) {             // hnsw.go:649-657
            search_layer(entry, 1, level);
            if (n_list > 0) entry = (uint32_t)readlane64(key[0], 0);
        }
        const uint32_t ef = ef_search > k ? ef_search : k;              // :660-663
        search_layer(entry, ef, 0);                                     // :664
        uint32_t cnt = n_list < k ? n_list : k;                         // :670-672
        if (tie) cnt = kHnswTieFlag;
        else {
#pragma unroll
            for (int s2 = 0; s2 < S; s2++) {
                const uint32_t e = (uint32_t)s2 * 64 + lane;
                if (e < k) {
                    const bool has = e < cnt;
                    rows_out[(size_t)qi * k + e] = has ? (uint32_t)key[s2] : 0xFFFFFFFFu;
                    dist_out[(size_t)qi * k + e] = has ? unord_f32((uint32_t)(key[s2] >> 32)) : __uint_as_float(0x7F800000u);
                }
            }
        }
        if (lane == 0) { count_out[qi] = cnt; if (evals_out) evals_out[qi] = n_eval; }
        epoch += 64;
    }
}

// ---------------------------------------------------------------- ingest -----------
// one wave per touched tile; lane == row within the tile
__global__ void __launch_bounds__(64)
k_ingest(IndexView v, const float* __restrict__ src, uint32_t row0, uint32_t n, uint32_t tile0) {
    const uint32_t t = tile0 + blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint32_t row = t * 64 + lane;
    const bool mine = row >= row0 && row < row0 + n;
    if (mine) {
        const float* s = src + (size_t)(row - row0) * v.dim;
        float4* dst = reinterpret_cast<float4*>(v.tiles) + (size_t)t * v.dim4 * 64 + lane;
        double mb = 0.0; float nb = 0.0f;
        const bool vec_ok = (v.dim & 3) == 0;
        for (uint32_t c = 0; c < v.dim4; c++) {
            float4 x;
            if (vec_ok) x = *reinterpret_cast<const float4*>(s + 4 * c);
            else {
                uint32_t j = 4 * c;
                x.x = j < v.dim ? s[j] : 0.f; x.y = j + 1 < v.dim ? s[j + 1] : 0.f;
                x.z = j + 2 < v.dim ? s[j + 2] : 0.f; x.w = j + 3 < v.dim ? s[j + 3] : 0.f;
            }
            dst[(size_t)c * 64] = x;
            if (v.metric == QV_COSINE || v.metric == QV_DOT || v.metric == QV_L2 || v.metric == QV_L2SQ) {   // distances.go:21 magnitudeB += b*b (non-cosine: |r| for the MFMA filter)
                mb = __builtin_fma((double)x.x, (double)x.x, mb); mb = __builtin_fma((double)x.y, (double)x.y, mb);
                mb = __builtin_fma((double)x.z, (double)x.z, mb); mb = __builtin_fma((double)x.w, (double)x.w, mb);
            } else if (v.metric == QV_COSINE_F32) {                       // adapter.go:119 normB += b*b (unfused)
                float p; p = x.x * x.x; nb = nb + p; p = x.y * x.y; nb = nb + p; p = x.z * x.z; nb = nb + p; p = x.w * x.w; nb = nb + p;
            }
        }
        if (v.metric == QV_COSINE || v.metric == QV_DOT || v.metric == QV_L2 || v.metric == QV_L2SQ) v.rnorm[row] = __builtin_sqrt(mb);
        else if (v.metric == QV_COSINE_F32) v.rnorm[row] = nb == 0.0f ? -1.0 : (double)(float)__builtin_sqrt((double)nb);
    }
    uint64_t m = __ballot(mine);
    if (lane == 0 && m) atomicOr(reinterpret_cast<unsigned long long*>(&v.alive[t]), (unsigned long long)m);
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ int32_t gen_int(uint64_t row_key, uint32_t col) {
    uint64_t h = splitmix64(row_key + (uint64_t)col);
    int32_t s = (int32_t)(h & 0xFFFF) + (int32_t)((h >> 16) & 0xFFFF) + (int32_t)((h >> 32) & 0xFFFF) + (int32_t)(h >> 48);
    return s - 131070;
}

// the synthetic-corpus generator of DESIGN.md (SplitMix64 -> Irwin-Hall(4) integers -> unit L2),
// written straight into the tile layout; integer arithmetic plus correctly rounded float64
// sqrt/div only, so any IEEE host reproduces it bit for bit
__global__ void __launch_bounds__(64)
k_generate(IndexView v, uint64_t seed, uint64_t gen_row0, uint32_t row0, uint32_t n, uint32_t tile0) {
    const uint32_t t = tile0 + blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint32_t row = t * 64 + lane;
    const bool mine = row >= row0 && row < row0 + n;
    if (mine) {
        const uint64_t g = gen_row0 + (row - row0);
        const uint64_t row_key = splitmix64(seed ^ (g * 0xD1342543DE82EF95ull));
        double sumsq = 0.0;
        for (uint32_t c = 0; c < v.dim; c++) { double x = (double)gen_int(row_key, c); sumsq = __builtin_fma(x, x, sumsq); }
        const double norm = sumsq > 0.0 ? __builtin_sqrt(sumsq) : 1.0;
        float4* dst = reinterpret_cast<float4*>(v.tiles) + (size_t)t * v.dim4 * 64 + lane;
        double mb = 0.0; float nb = 0.0f;
        for (uint32_t c = 0; c < v.dim4; c++) {
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t col = 4 * c + j;
                e[j] = col < v.dim ? (float)((double)gen_int(row_key, col) / norm) : 0.0f;
                if (v.metric == QV_COSINE || v.metric == QV_DOT || v.metric == QV_L2 || v.metric == QV_L2SQ) mb = __builtin_fma((double)e[j], (double)e[j], mb);
                else if (v.metric == QV_COSINE_F32) { float p = e[j] * e[j]; nb = nb + p; }
            }
            dst[(size_t)c * 64] = make_float4(e[0], e[1], e[2], e[3]);
        }
        if (v.metric == QV_COSINE || v.metric == QV_DOT || v.metric == QV_L2 || v.metric == QV_L2SQ) v.rnorm[row] = __builtin_sqrt(mb);
        else if (v.metric == QV_COSINE_F32) v.rnorm[row] = nb == 0.0f ? -1.0 : (double)(float)__builtin_sqrt((double)nb);
        if (v.rowmaj) {
            float* rm = v.rowmaj + (size_t)row * v.dim;
            for (uint32_t c = 0; c < v.dim; c++) rm[c] = (float)((double)gen_int(row_key, c) / norm);
        }
    }
    uint64_t m = __ballot(mine);
    if (lane == 0 && m) atomicOr(reinterpret_cast<unsigned long long*>(&v.alive[t]), (unsigned long long)m);
}

__global__ void k_set_alive(IndexView v, const uint32_t* __restrict__ rows, uint32_t n, int alive) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r = rows[i];
    if (r >= v.n_rows) return;
    unsigned long long bit = 1ull << (r & 63);
    if (alive) atomicOr(reinterpret_cast<unsigned long long*>(&v.alive[r >> 6]), bit);
    else atomicAnd(reinterpret_cast<unsigned long long*>(&v.alive[r >> 6]), ~bit);
}

__global__ void k_fetch_row(IndexView v, uint32_t row, float* __restrict__ out) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= v.dim) return;
    out[j] = v.tiles[((size_t)(row >> 6) * v.dim4 + (j >> 2)) * 256 + (row & 63) * 4 + (j & 3)];
}

// ---------------------------------------------------------------- launchers --------
constexpr int kUnroll = 16;

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    if (!e || !*e) return dflt;
    int v = atoi(e);
    return v > 0 ? v : dflt;
}

ScanPlan plan_scan(uint32_t n_tiles, int cus) {
    ScanPlan p;
    p.block = kScanBlock;
    uint32_t want = (n_tiles + kScanWaves - 1) / kScanWaves;          // one tile per wave at most
    static const int wg_per_cu = env_int("QV_SCAN_WG_PER_CU", 2);     // 2 workgroups = 8 waves per CU: measured best (profiles/r01_sweep.txt)
    uint32_t cap = (uint32_t)cus * (uint32_t)wg_per_cu;
    p.grid = want < cap ? want : cap;
    if (p.grid == 0) p.grid = 1;
    p.n_lists = p.grid;
    return p;
}

size_t scan_workspace_bytes(const ScanPlan& p, uint32_t nq, uint32_t k) { return ((size_t)p.n_lists * 4 * nq * k * sizeof(uint64_t) + 255) / 256 * 256; }   // x4: the multi-query scan may use up to 8 WG/CU

static size_t query_lds_bytes(int metric, uint32_t dim4) {
    size_t q = (metric == QV_COSINE || metric == QV_DOT || metric == QV_L2SQ_F64) ? sizeof(double) : sizeof(float);
    return ((size_t)dim4 * 4 * q + 15) / 16 * 16;
}

template <typename F> static hipError_t set_lds(F f, size_t bytes) {
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

hipError_t launch_merge_pairs(const float* d_dist, const uint32_t* d_rows, uint32_t n_lists, uint32_t k,
                              uint32_t* d_rows_out, float* d_dist_out, hipStream_t s) {
    if (k == 0 || k > (uint32_t)kMaxFusedK || n_lists == 0) return hipErrorInvalidValue;
    uint32_t total = n_lists * k;
    uint32_t mblock = total >= 16 * 64 * 4 ? kMergeBlock : (total >= 4 * 64 ? 256 : 64);
    hipLaunchKernelGGL(k_merge_pairs, dim3(1), dim3(mblock), 0, s, d_dist, d_rows, total, k, d_rows_out, d_dist_out);
    return hipGetLastError();
}

hipError_t launch_flat_topk(const IndexView& v, const ScanPlan& p, const float* d_queries, uint32_t nq, uint32_t k,
                            void* d_ws, uint32_t* d_rows_out, float* d_dist_out, hipStream_t s,
                            hipEvent_t ev0, hipEvent_t ev1) {
    if (k == 0 || k > (uint32_t)kMaxFusedK || nq == 0) return hipErrorInvalidValue;
    const size_t lds = query_lds_bytes(v.metric, v.dim4) + (size_t)kScanWaves * 64 * sizeof(uint64_t);
    uint64_t* partial = static_cast<uint64_t*>(d_ws);
    hipError_t e = hipSuccess;
    static const int mq_min = env_int("QV_MQ_MIN", 2);                // nq >= this: queries share a corpus pass
    if ((int)nq >= mq_min) {
        // QB queries per corpus pass.  Measured on MI355X, 256 x 1M x 768 cosine (profiles/r01_sweep_mq.txt):
        // QB=8 is HBM-bound (0.454 ms/pass), QB=16 is f64-VALU-bound (0.75 ms/pass, 12.0 ms per 256 queries).
        static const int mq_qb_env = env_int("QV_MQ_QB", 0), mq_wg = env_int("QV_MQ_WG_PER_CU", 2);
        const int qb = mq_qb_env ? mq_qb_env : (nq >= 9 ? 16 : (nq >= 5 ? 8 : 4));
        const uint32_t want = (v.n_tiles + kScanWaves - 1) / kScanWaves;
        const uint32_t grid = std::max(1u, std::min(want, (uint32_t)mq_wg * (p.grid / 2 ? p.grid / 2 : 1)));   // p.grid = 2 WG/CU * CUs
        void* qblk = static_cast<char*>(d_ws) + scan_workspace_bytes(p, nq, k);   // tail of the workspace
#define QV_MQ_LAUNCH(MMM, QQ)                                                                                              \
        {                                                                                                                     \
            using QT = typename MT<MMM>::Q;                                                                                   \
            const uint32_t groups = (nq + QQ - 1) / QQ;                                                                       \
            const uint32_t per = v.dim4 * 4 * QQ;                                                                             \
            hipLaunchKernelGGL((k_prep_qblk<MMM, QQ>), dim3((per + 255) / 256, groups), dim3(256), 0, s, d_queries, nq, v.dim, v.dim4, static_cast<QT*>(qblk)); \
            const size_t lds_mq = (size_t)kScanWaves * QQ * 64 * sizeof(uint64_t);                                            \
            if (ev0) (void)hipEventRecord(ev0, s);                                                                            \
            hipLaunchKernelGGL((k_flat_scan_mq<MMM, 4, QQ, true>), dim3(grid, groups), dim3(p.block), lds_mq, s, v, d_queries, static_cast<const QT*>(qblk), nq, k, partial); \
            if (ev1) (void)hipEventRecord(ev1, s);                                                                            \
        }
        if (qb == 16) { QV_DISPATCH_METRIC(v.metric, { QV_MQ_LAUNCH(MM, 16) }); }
        else if (qb == 8) { QV_DISPATCH_METRIC(v.metric, { QV_MQ_LAUNCH(MM, 8) }); }
        else if (qb == 4) { QV_DISPATCH_METRIC(v.metric, { QV_MQ_LAUNCH(MM, 4) }); }
        else return hipErrorInvalidValue;
#undef QV_MQ_LAUNCH
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        // partial lists are laid out with stride `grid` lists per query
        uint32_t total = grid * k;
        uint32_t mblock = total >= 16 * 64 * 4 ? kMergeBlock : (total >= 4 * 64 ? 256 : 64);
        hipLaunchKernelGGL(k_merge_lists, dim3(nq), dim3(mblock), 0, s, partial, grid, k, d_rows_out, d_dist_out);
        return hipGetLastError();
    }
    static const int unroll = env_int("QV_SCAN_UNROLL", kUnroll);     // tuning knob (cosine only): loads in flight per wave
    if (v.metric == QV_COSINE && unroll != kUnroll) {
#define QV_SCAN_U(UU)                                                                                             \
        case UU: e = set_lds(k_flat_scan<QV_COSINE, UU>, lds); if (e != hipSuccess) return e;                    \
            if (ev0) (void)hipEventRecord(ev0, s);                                                                \
            hipLaunchKernelGGL((k_flat_scan<QV_COSINE, UU>), dim3(p.grid, nq), dim3(p.block), lds, s, v, d_queries, k, partial); \
            if (ev1) (void)hipEventRecord(ev1, s); break;
        switch (unroll) { QV_SCAN_U(4) QV_SCAN_U(8) QV_SCAN_U(12) QV_SCAN_U(24) QV_SCAN_U(32) default: return hipErrorInvalidValue; }
#undef QV_SCAN_U
    } else
    QV_DISPATCH_METRIC(v.metric, {
        e = set_lds(k_flat_scan<MM, kUnroll>, lds);
        if (e != hipSuccess) return e;
        if (ev0) (void)hipEventRecord(ev0, s);
        hipLaunchKernelGGL((k_flat_scan<MM, kUnroll>), dim3(p.grid, nq), dim3(p.block), lds, s, v, d_queries, k, partial);
        if (ev1) (void)hipEventRecord(ev1, s);
    });
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    uint32_t total = p.n_lists * k;
    uint32_t mblock = total >= 16 * 64 * 4 ? kMergeBlock : (total >= 4 * 64 ? 256 : 64);
    hipLaunchKernelGGL(k_merge_lists, dim3(nq), dim3(mblock), 0, s, partial, p.n_lists, k, d_rows_out, d_dist_out);
    return hipGetLastError();
}

// ---- MFMA batched path --------------------------------------------------------------
uint32_t batched_sample_rows(uint32_t n_rows) {
    static const int s = env_int("QV_MFMA_SAMPLE_ROWS", 8192);
    return std::min<uint32_t>(n_rows, (uint32_t)s);
}
bool batched_supported(const IndexView& v, uint32_t nq, uint32_t k) {
    static const int min_rows = env_int("QV_MFMA_MIN_ROWS", 262144), min_q = env_int("QV_MFMA_MIN_QUERIES", 32);
    return (v.metric == QV_COSINE || v.metric == QV_DOT || v.metric == QV_L2 || v.metric == QV_L2SQ) && k <= (uint32_t)kMaxFusedK && nq >= (uint32_t)min_q && v.n_rows >= (uint32_t)min_rows;
}
size_t batched_workspace_bytes(const IndexView& v, const ScanPlan& p, uint32_t nq, uint32_t k) {
    const uint32_t nq_pad = (nq + 63) / 64 * 64;
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
    const uint32_t nq_pad = (nq + 63) / 64 * 64;
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
    vs.n_rows = batched_sample_rows(v.n_rows);
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

// ---- HNSW traversal ------------------------------------------------------------------
size_t hnsw_lds_bytes(int metric, uint32_t dim4) {
    return query_lds_bytes(metric, dim4) + (size_t)(kHnswCandCap + kHnswEfMax + 1) * sizeof(HRes) + (size_t)kHnswMaxDeg * 8 + 64;
}
uint32_t hnsw_grid(int cus, int metric, uint32_t dim4, uint32_t nq) {
    const size_t lds = hnsw_lds_bytes(metric, dim4);
    uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(8, (size_t)(160 * 1024) / lds));
    return std::max(1u, std::min(nq, (uint32_t)cus * per_cu));
}
hipError_t launch_hnsw_search(const IndexView& v, const GraphView& g, const float* d_queries, uint32_t nq, uint32_t k, uint32_t ef,
                              uint32_t* d_visited, uint32_t grid, uint32_t epoch0, uint32_t* d_rows_out, float* d_dist_out,
                              uint32_t* d_count_out, uint32_t* d_evals_out, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    if (k == 0 || k > (uint32_t)kHnswEfMax || ef > (uint32_t)kHnswEfMax || g.max_m0 > (uint32_t)kHnswMaxDeg || g.max_m > (uint32_t)kHnswMaxDeg) return hipErrorInvalidValue;
    const size_t lds = hnsw_lds_bytes(v.metric, v.dim4);
    hipError_t e = hipSuccess;
    QV_DISPATCH_METRIC(v.metric, {
        e = set_lds(k_hnsw_search<MM, 16>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_hnsw_search<MM, 16>), dim3(grid), dim3(64), lds, s, v, g, d_queries, nq, k, ef, d_visited, epoch0,
                           d_rows_out, d_dist_out, d_count_out, d_evals_out);
    });
    return hipGetLastError();
}

// wave-resident form: registers only (plus the staged query); tie-flagged queries report kHnswTieFlag
size_t hnsw_wave_lds_bytes(int metric, uint32_t dim4) { return query_lds_bytes(metric, dim4) + 64 * sizeof(uint32_t) + (size_t)kHnswStageRows * (dim4 + 1) * 16 + 64; }
uint32_t hnsw_wave_grid(int cus, int metric, uint32_t dim4) {
    const size_t lds = hnsw_wave_lds_bytes(metric, dim4);
    uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(16, (size_t)(160 * 1024) / lds));
    static const int cap = env_int("QV_HNSW_WAVES_PER_CU", 16);
    per_cu = std::min<uint32_t>(per_cu, (uint32_t)cap);
    return (uint32_t)cus * per_cu;
}
hipError_t launch_hnsw_search_wave(const IndexView& v, const GraphView& g, const float* d_queries, uint32_t nq, uint32_t k, uint32_t ef,
                                   uint32_t* d_visited, uint32_t grid, uint32_t epoch0, uint32_t* d_rows_out, float* d_dist_out,
                                   uint32_t* d_count_out, uint32_t* d_evals_out, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    const uint32_t efx = ef > k ? ef : k;
    if (k == 0 || efx > (uint32_t)kHnswEfMax || g.max_m0 > (uint32_t)kHnswMaxDeg || g.max_m > (uint32_t)kHnswMaxDeg) return hipErrorInvalidValue;
    const size_t lds = hnsw_wave_lds_bytes(v.metric, v.dim4);
    hipError_t e = hipSuccess;
#define QV_HW(SS) QV_DISPATCH_METRIC(v.metric, {                                                                     \
        e = set_lds(k_hnsw_search_wave<MM, 8, SS>, lds);                                                              \
        if (e != hipSuccess) return e;                                                                                \
        hipLaunchKernelGGL((k_hnsw_search_wave<MM, 8, SS>), dim3(grid), dim3(64), lds, s, v, g, d_queries, nq, k, ef, d_visited, epoch0, \
                           d_rows_out, d_dist_out, d_count_out, d_evals_out);                                         \
    })
    if (efx <= 64) { QV_HW(1); } else if (efx <= 128) { QV_HW(2); } else if (efx <= 256) { QV_HW(4); } else { QV_HW(8); }
#undef QV_HW
    return hipGetLastError();
}

size_t full_sort_workspace_bytes(uint32_t n_tiles) {
    size_t n = (size_t)n_tiles * 64;
    size_t nblocks = (n + kRadixTile - 1) / kRadixTile;
    return 2 * n * sizeof(uint64_t) + (256 * nblocks + 512) * sizeof(uint32_t) + 256;
}

hipError_t launch_flat_fullsort(const IndexView& v, const ScanPlan& p, const float* d_query, uint32_t k,
                                void* d_ws, uint32_t* d_rows_out, float* d_dist_out, hipStream_t s) {
    const uint32_t n = v.n_tiles * 64;
    uint64_t* ka = static_cast<uint64_t*>(d_ws);
    uint64_t* kb = ka + n;
    uint32_t* hist = reinterpret_cast<uint32_t*>(kb + n);
    const uint32_t nblocks = (n + kRadixTile - 1) / kRadixTile;
    uint32_t* dtot = hist + (size_t)256 * nblocks;
    uint32_t* dbase = dtot + 256;
    const size_t lds = query_lds_bytes(v.metric, v.dim4);
    hipError_t e = hipSuccess;
    QV_DISPATCH_METRIC(v.metric, {
        e = set_lds(k_flat_keys<MM, kUnroll>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_flat_keys<MM, kUnroll>), dim3(p.grid), dim3(p.block), lds, s, v, d_query, ka);
    });
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    uint64_t* in = ka; uint64_t* out = kb;
    for (uint32_t shift = 32; shift < 64; shift += 8) {
        hipLaunchKernelGGL(k_radix_hist, dim3(nblocks), dim3(kRadixBlock), 0, s, in, n, shift, hist);
        hipLaunchKernelGGL(k_radix_scan_digits, dim3(256), dim3(256), 0, s, hist, nblocks, dtot);
        hipLaunchKernelGGL(k_radix_scan_totals, dim3(1), dim3(256), 0, s, dtot, dbase);
        hipLaunchKernelGGL(k_radix_scatter, dim3(nblocks), dim3(kRadixBlock), 0, s, in, out, n, shift, hist, dbase);
        uint64_t* t = in; in = out; out = t;
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_emit_topk, dim3((k + 255) / 256), dim3(256), 0, s, in, n, k, d_rows_out, d_dist_out);
    return hipGetLastError();
}

hipError_t launch_distance_rows(const IndexView& v, const float* d_query, const uint32_t* d_rows, uint32_t n,
                                float* d_dist_out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const size_t lds = query_lds_bytes(v.metric, v.dim4);
    hipError_t e = hipSuccess;
    QV_DISPATCH_METRIC(v.metric, {
        e = set_lds(k_distance_rows<MM, kUnroll>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_distance_rows<MM, kUnroll>), dim3((n + 63) / 64), dim3(64), lds, s, v, d_query, d_rows, n, d_dist_out);
    });
    return hipGetLastError();
}

hipError_t launch_distance_pairs(int metric, const float* d_a, const float* d_b, uint32_t n, uint32_t dim, float* d_out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    QV_DISPATCH_METRIC(metric, {
        hipLaunchKernelGGL((k_distance_pairs<MM>), dim3((n + 63) / 64), dim3(64), 0, s, d_a, d_b, n, dim, d_out);
    });
    return hipGetLastError();
}

hipError_t launch_ingest(const IndexView& v, const float* d_rows, uint32_t row0, uint32_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint32_t t0 = row0 / 64, t1 = (row0 + n - 1) / 64;
    hipLaunchKernelGGL(k_ingest, dim3(t1 - t0 + 1), dim3(64), 0, s, v, d_rows, row0, n, t0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (v.rowmaj) return hipMemcpyAsync(v.rowmaj + (size_t)row0 * v.dim, d_rows, (size_t)n * v.dim * sizeof(float), hipMemcpyDeviceToDevice, s);
    return hipSuccess;
}

hipError_t launch_generate(const IndexView& v, uint64_t seed, uint64_t gen_row0, uint32_t row0, uint32_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint32_t t0 = row0 / 64, t1 = (row0 + n - 1) / 64;
    hipLaunchKernelGGL(k_generate, dim3(t1 - t0 + 1), dim3(64), 0, s, v, seed, gen_row0, row0, n, t0);
    return hipGetLastError();
}

hipError_t launch_set_alive(const IndexView& v, const uint32_t* d_rows, uint32_t n, int alive, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_set_alive, dim3((n + 255) / 256), dim3(256), 0, s, v, d_rows, n, alive);
    return hipGetLastError();
}

hipError_t launch_fetch_row(const IndexView& v, uint32_t row, float* d_out, hipStream_t s) {
    hipLaunchKernelGGL(k_fetch_row, dim3((v.dim + 255) / 256), dim3(256), 0, s, v, row, d_out);
    return hipGetLastError();
}

}  // namespace qv
