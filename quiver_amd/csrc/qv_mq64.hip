// qv_mq64.hip — exact multi-query flat scan on the float64 matrix cores (cosine / dot).
//
// What it replaces: HybridIndex.BatchSearch's per-query ExactIndex.Search calls
// (pkg/hybrid/hybrid_index.go:677-811 -> exact.go:92-133), 16 or 32 queries per corpus pass, with the
// reference's arithmetic: float64 accumulation of exact float32 products over dims 0..D-1 IN ORDER
// (pkg/vectortypes/distances.go:17-22, :82-86).
//
// Why a matrix instruction can be bit-exact here: v_mfma_f64_16x16x4_f64 computes, for every output,
//     d = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, c))))
// — a sequential chain of individually rounded IEEE FMAs over k = 0..3 (tools/ubench/mfma_f64_order.hip:
// 51,200 of 51,200 outputs equal to that chain on MI355X; the reverse order or a singly rounded sum would
// not match).  Chaining one MFMA per 16-byte chunk with C = the previous D therefore reproduces the scalar
// loop's rounding sequence exactly, at 77.5 TFLOP/s measured (tools/ubench/mfma_f64_rate.hip) against
// ~37 TFLOP/s for the same chain written with v_fma_f64 (k_flat_scan_mq, f64-VALU-issue-bound).
//
// Operand mapping (lane l of the wave):
//   A[i][k] = query (16*nb + i), dim 4c+k      lane l supplies i = l%16, k = l/16   (from LDS, f32 -> f64)
//   B[k][j] = row (64t + 16g + j), dim 4c+k    lane l supplies k = l/16, j = l%16
//   D: lane l, register r  <->  query 16*nb + 4r + l/16,  row 64t + 16g + l%16
// The corpus stays in its tile layout ([dim4][64 rows][4 dims]): lane l loads the float at (row 16g + l%16, dim 4c + l/16) of
// the chunk directly — the B operand's own arrangement; the four loads of a chunk each cover one 256-byte run of 16 rows.
#include "qv_kernels.h"

namespace qv {

typedef double d4 __attribute__((ext_vector_type(4)));

// Shapes of the workgroup (template parameters of the kernel):
//   W  waves that share out the tiles;  H  such wave sets per workgroup, set h computing query blocks h*NB.. for the SAME tiles;
//   NB 16-query blocks per wave;  U chunks per register block (double-buffered);  WGS workgroups per CU
struct Mq64Shape { int h, nb, u, w, wgs; };
constexpr int kMq64Cap = 1024;      // candidates a workgroup of k_mq64_bounded can hold
static Mq64Shape mq64_shape(uint32_t nq) {
    // 0: 16 queries/pass, 2 x 4 waves per CU on distinct tiles; 1: 32 queries/pass as two 4-wave sets sharing tiles; 2: 32
    // queries/pass, 8 waves per CU with two blocks per wave.  Measured at 1M x 768 cosine with the bounded kernel
    // (tools/sweep_shapes.py, profiles/r02_batch_shapes.txt): a lone 16-query pass 0.98 ms, further ones 0.75 ms each; 32-query
    // passes 1.15 ms, then 0.85-0.99 ms each — so only a batch of 16 or fewer queries goes in the 16-query shape.
    static const int forced = dev_env_int("QV_MQ64_MODE", 0) - 1;        // env value 1..3 -> mode 0..2 (0 = choose per batch)
    int mode = forced;
    if (mode < 0) mode = nq <= 16 ? 0 : 2;
    if (mode == 0) return {1, 1, 8, 4, 2};
    if (mode == 1) return {2, 1, 8, 4, 1};
    return {1, 2, 4, 8, 1};
}

// query fragments for one group of 16*NB queries: qfrag[nb][c][lane] = q_{16nb + lane%16}[4c + lane/16] (float32,
// zero beyond dim; query slots past nq replicate the last query, their results are dropped), and per-query constants
template <int M, int NB>
__global__ void k_mq64_prep(const float* __restrict__ queries, uint32_t nq, uint32_t dim, uint32_t dim4,
                            float* __restrict__ qfrag, double* __restrict__ qconst) {
    const uint32_t grp = blockIdx.y;
    const uint32_t per = (uint32_t)NB * dim4 * 64;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x) {
        const uint32_t lane = i & 63, c = (i >> 6) % dim4, nb = (i >> 6) / dim4;
        uint32_t qi = grp * 16 * NB + nb * 16 + (lane & 15); if (qi >= nq) qi = nq - 1;
        const uint32_t d = c * 4 + (lane >> 4);
        qfrag[(size_t)grp * per + i] = d < dim ? queries[(size_t)qi * dim + d] : 0.0f;
    }
    if (blockIdx.x == 0) {
        // the queries' own norms in the element order of distances.go:20 — one thread per query walks the chain, the block
        // stages 64 dimensions of all 16*NB queries at a time so that the reads are whole lines
        __shared__ __align__(16) float stage[16 * NB][68];      // 64 dims + 4 floats of padding: rows stay 16-byte aligned
        double ma = 0.0;
        if constexpr (M == QV_COSINE) {
            for (uint32_t d0 = 0; d0 < dim; d0 += 64) {
                for (uint32_t i = threadIdx.x; i < 16 * NB * 64; i += blockDim.x) {
                    uint32_t qi = grp * 16 * NB + (i >> 6); if (qi >= nq) qi = nq - 1;
                    const uint32_t d = d0 + (i & 63);
                    stage[i >> 6][i & 63] = d < dim ? queries[(size_t)qi * dim + d] : 0.0f;
                }
                __syncthreads();
                if (threadIdx.x < 16 * NB) {
                    f4 x[16];                                      // all 64 values first: the chain then runs from registers
#pragma unroll
                    for (int c = 0; c < 16; c++) x[c] = reinterpret_cast<const f4*>(stage[threadIdx.x])[c];
#pragma unroll
                    for (int c = 0; c < 16; c++) {                 // values beyond dim are 0: fma(0, 0, ma) == ma
                        double a = (double)x[c].x; ma = __builtin_fma(a, a, ma);
                        a = (double)x[c].y; ma = __builtin_fma(a, a, ma);
                        a = (double)x[c].z; ma = __builtin_fma(a, a, ma);
                        a = (double)x[c].w; ma = __builtin_fma(a, a, ma);
                    }
                }
                __syncthreads();
            }
        }
        if (threadIdx.x < 16 * NB) qconst[(size_t)grp * 16 * NB + threadIdx.x] = __builtin_sqrt(ma);     // dot: unused (0)
    }
}

// list maintenance is rare (see the bound below); kept out of line so that the 16*NB lists (distinct registers) do not each
// inline a bitonic network and an insertion loop
__attribute__((noinline)) __device__ uint64_t sort_out_of_line(uint64_t key, uint32_t lane) { return wave_sort64(key, lane); }
// inserts every lane's candidate below `thr`; thr <- min(thr, the list's k-th key)
__attribute__((noinline)) __device__ uint64_t insert_out_of_line(uint64_t list, uint64_t cand, uint64_t& thr, uint32_t kth, uint32_t lane) {
    uint64_t mask = __ballot(cand < thr);
    while (mask) {
        const uint32_t src = (uint32_t)__builtin_ctzll(mask);
        mask &= mask - 1;
        const uint64_t c = readlane64(cand, src);
        if (c >= thr) continue;                      // threshold tightened since the ballot
        const uint32_t pos = (uint32_t)__builtin_popcountll(__ballot(list < c));
        const uint64_t up = wave_shr1(list);
        list = lane > pos ? up : (lane == pos ? c : list);
        const uint64_t kk = readlane64(list, kth);
        thr = kk < thr ? kk : thr;
    }
    return list;
}

// The k-th key of a SAMPLE of the corpus bounds every query's final k-th key from above.  k_flat_scan_mq64 runs twice:
// over a few hundred tiles spread across the corpus (bound == nullptr: lists start from a sorted first tile), then — after
// k_mq64_bound has turned the sample's lists into bound[q] = its k-th key + 1 — over all tiles with every list empty and every
// threshold at bound[q].  Why: a wave sees only n_tiles / 2048 tiles of a pass (7.6 at 1M rows), so a threshold learnt from
// its own rows never gets tight: 2 of its 64 rows beat it per query and tile, and the kernel spent more VALU instructions
// maintaining lists (4.0e8 per 256 x 1M x 768 batch, SQ_INSTS_VALU) and sorting first tiles than on the scan itself
// (profiles/r02_mq64_breakdown.txt).  With the sample's bound, 0.04 rows per query and tile pass the one-sided test below.
__global__ void k_mq64_bound(const uint64_t* __restrict__ partial, uint32_t n_lists, uint32_t k, uint64_t* __restrict__ bound, uint32_t* __restrict__ overflow) {
    const uint32_t q = blockIdx.x, lane = threadIdx.x, kth = k - 1;
    uint64_t list = kDeadKey, thr = kDeadKey;
    const uint64_t* p = partial + (size_t)q * n_lists * k;
    for (uint32_t i = 0; i < n_lists * k; i += 64) {
        const uint64_t key = i + lane < n_lists * k ? p[i + lane] : kDeadKey;
        list = insert_out_of_line(list, key, thr, kth, lane);
    }
    // fewer than k live rows in the sample: no bound (kDeadKey admits everything)
    if (lane == 0) bound[q] = thr == kDeadKey ? kDeadKey : thr + 1;
    if (q == 0 && lane == 0) *overflow = 0;
}

// A single wave per SIMD issues f64 MFMAs at half the pipe's rate (mfma_f64_rate.hip: 38.9 vs 77.5 TFLOP/s), and every
// other VALU instruction of the SIMD waits while one runs (mfma_f64_mix.hip, mfma_f64_intcvt.hip), so the shapes trade
// registers for waves per SIMD and MFMAs per row convert: see Mq64Shape above.  With H > 1, set h computes its query
// blocks for the SAME tiles as the other sets (the second read of a tile is an L2 hit, HBM sees it once).
// Tiles visited: tile_step * i for i < n_iter (the sample pass strides, the full pass has tile_step 1, n_iter = n_tiles).
// wave_lists: every wave writes its own lists (partial[(q * grid * W + wg * W + wave) * k + i]) instead of the workgroup's merged
// ones — the sample pass, whose waves hold full lists: merging them on one wave was most of that pass's time.
template <int M, int H, int NB, int U, int W, int WGS>
__global__ void __launch_bounds__(64 * W * H, WGS)
k_flat_scan_mq64(IndexView v, const float* __restrict__ qfrag_g, const double* __restrict__ qconst_g, uint32_t nq, uint32_t k,
                 const uint64_t* __restrict__ bound, uint32_t n_iter, uint32_t tile_step, uint32_t wave_lists, const uint32_t* __restrict__ run_if,
                 uint64_t* __restrict__ partial) {
    static_assert(M == QV_COSINE || M == QV_DOT, "the f64 matrix path covers the fma(q, x, acc) metrics");
    if (run_if && *run_if == 0) return;       // the fallback launch after k_mq64_bounded: nothing overflowed
    constexpr int Q = 16 * NB;       // queries per wave (NB 16-query blocks)
    extern __shared__ __align__(16) unsigned char smem[];
    typedef __attribute__((address_space(3))) float lds_f32;
    typedef __attribute__((address_space(3))) uint64_t lds_u64;
    float* qf = reinterpret_cast<float*>(smem);                                   // [NB][dim4][64]
    const lds_f32* qf3 = (const lds_f32*)smem;                                    // same, as an LDS-address-space pointer (ds_read)
    const size_t qf_bytes = (size_t)H * NB * v.dim4 * 64 * sizeof(float);
    lds_u64* scratch = (lds_u64*)((__attribute__((address_space(3))) unsigned char*)smem + qf_bytes);   // [waves][4][64] regrouping
    const uint32_t lane = lane_id();
    const uint32_t wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t half = wave_all / W, wave = wave_all % W;      // query block of this wave, tile slot of this wave
    const uint32_t grp = blockIdx.y, q0 = (grp * H + half) * Q;
    const uint32_t blk = lane >> 4, j16 = lane & 15;

    {   // stage this group's query fragments (all H blocks)
        const f4* src = reinterpret_cast<const f4*>(qfrag_g + (size_t)grp * H * NB * v.dim4 * 64);
        f4* dst = reinterpret_cast<f4*>(qf);
        for (uint32_t i = threadIdx.x; i < (uint32_t)H * NB * v.dim4 * 16; i += blockDim.x) dst[i] = src[i];
    }
    qf3 += half * NB * v.dim4 * 64;                                                      // this wave's query block
    // per-lane query constants: register (nb, r) belongs to query 16nb + 4r + blk
    double qn[NB][4];
    uint64_t thrv[NB][4];                  // per lane: current threshold key of "its" query for register (nb, r)
#pragma unroll
    for (int nb = 0; nb < NB; nb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t qi = q0 + nb * 16 + r * 4 + blk;
            qn[nb][r] = qconst_g[qi];                                                    // qconst is laid out by query slot
            thrv[nb][r] = bound ? bound[qi < nq ? qi : nq - 1] : kDeadKey;
        }
    __syncthreads();

    const uint32_t kth = k - 1;
    uint64_t list[Q];                      // wave-resident ascending top-k list per query (one key per lane)
#pragma unroll
    for (int i = 0; i < Q; i++) list[i] = kDeadKey;

    const f4* tiles = reinterpret_cast<const f4*>(v.tiles);
    const uint32_t tw = gridDim.x * W;
    bool first = bound == nullptr;         // sample pass: the first tile's 64 rows per query are sorted into the list outright
    lds_u64* my_scratch = scratch + wave_all * 4 * 64;

    // this lane's B elements of tile t: row 16g + lane%16, dim 4c + lane/16
    auto tile_ptr = [&](uint32_t t) { return reinterpret_cast<const float*>(tiles + (size_t)t * v.dim4 * 64) + j16 * 4 + blk; };
    float xa[U][4], xb[U][4];              // two register blocks of row elements and query values, used alternately
    float qa[U][NB], qb2[U][NB];
    const uint32_t npair = v.dim4 / (2 * U);          // pairs of U-chunk register blocks; the remaining chunks go one at a time
    auto load_x_from = [&](const float* pb, float (&x)[U][4], uint32_t c0) {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int g = 0; g < 4; g++) x[u][g] = pb[(size_t)(c0 + u) * 256 + g * 64];
    };
    auto load_q = [&](float (&q)[U][NB], uint32_t c0) {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int nb = 0; nb < NB; nb++) q[u][nb] = qf3[((uint32_t)nb * v.dim4 + c0 + u) * 64 + lane];
    };
    const uint32_t it0 = blockIdx.x * W + wave;
    if (it0 < n_iter && npair) { load_x_from(tile_ptr(it0 * tile_step), xa, 0); load_q(qa, 0); }

    for (uint32_t it = it0; it < n_iter; it += tw) {
        const uint32_t t = it * tile_step;
        const float* pb = tile_ptr(t);
        d4 acc[NB][4];
#pragma unroll
        for (int nb = 0; nb < NB; nb++)
#pragma unroll
            for (int g = 0; g < 4; g++) acc[nb][g] = (d4){0.0, 0.0, 0.0, 0.0};
        // per-row constants of this tile, in the output mapping (row 16g + j16)
        double rn[4];
        const uint64_t alive_word = v.alive[t];
#pragma unroll
        for (int g = 0; g < 4; g++) { rn[g] = 0.0; if constexpr (MT<M>::needs_rnorm) rn[g] = v.rnorm[(size_t)t * 64 + g * 16 + j16]; }

        // one chunk: B operands = this lane's (row 16g + lane%16, dim 4c + lane/16) elements, read straight from the tile in
        // that arrangement (each of the four loads of a chunk covers one 256-byte run of 16 rows x 4 dims, so the tile is
        // still fetched in whole lines, once); A operands = the queries' values for this chunk
        auto step = [&](const float (&y)[4], const float (&qa)[NB]) {
            const double b0 = (double)y[0], b1 = (double)y[1], b2 = (double)y[2], b3 = (double)y[3];
#pragma unroll
            for (int nb = 0; nb < NB; nb++) {
                const double a = (double)qa[nb];
                acc[nb][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc[nb][0], 0, 0, 0);
                acc[nb][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc[nb][1], 0, 0, 0);
                acc[nb][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, acc[nb][2], 0, 0, 0);
                acc[nb][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b3, acc[nb][3], 0, 0, 0);
            }
        };
        auto load_x = [&](float (&x)[U][4], uint32_t c0) { load_x_from(pb, x, c0); };
        auto consume = [&](const float (&x)[U][4], const float (&qa)[U][NB]) {
#pragma unroll
            for (int u = 0; u < U; u++) step(x[u], qa[u]);
            // keep the accumulators in the accumulation registers across the back-edge
#pragma unroll
            for (int nb = 0; nb < NB; nb++)
#pragma unroll
                for (int g = 0; g < 4; g++) asm volatile("" : "+v"(acc[nb][g]));
        };
        // chunks in order, U at a time, register blocks in pairs with every load of the steady state unconditional (see
        // k_mq64_bounded); the last pair requests block 0 of the wave's next tile before the epilogue
        for (uint32_t pr = 0; pr + 1 < npair; pr++) {
            load_x(xb, (2 * pr + 1) * U); load_q(qb2, (2 * pr + 1) * U);
            __builtin_amdgcn_sched_barrier(0);
            consume(xa, qa);
            __builtin_amdgcn_sched_barrier(0);
            load_x(xa, (2 * pr + 2) * U); load_q(qa, (2 * pr + 2) * U);
            __builtin_amdgcn_sched_barrier(0);
            consume(xb, qb2);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (npair) {
            load_x(xb, (2 * npair - 1) * U); load_q(qb2, (2 * npair - 1) * U);
            __builtin_amdgcn_sched_barrier(0);
            consume(xa, qa);
            __builtin_amdgcn_sched_barrier(0);
            load_x_from(tile_ptr((it + tw < n_iter ? it + tw : it) * tile_step), xa, 0); load_q(qa, 0);
            __builtin_amdgcn_sched_barrier(0);
            consume(xb, qb2);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (uint32_t c = 2 * npair * U; c < v.dim4; c++) {
            float q1[NB], x1[4];
#pragma unroll
            for (int nb = 0; nb < NB; nb++) q1[nb] = qf3[((uint32_t)nb * v.dim4 + c) * 64 + lane];
#pragma unroll
            for (int g = 0; g < 4; g++) x1[g] = pb[(size_t)c * 256 + g * 64];
            step(x1, q1);
        }

        // epilogue: lane (blk, j16), register (nb, g, r) = (query 16nb+4r+blk, row 64t+16g+j16)
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                QConst qc; qc.qn = qn[nb][r]; qc.qn32 = 0.0f;
                // Almost no row beats its query's threshold.  A one-sided test on the accumulator settles that without the
                // division, the rounding to float32 and the key: with Tu the next float32 above the threshold's distance, a
                // row whose float64 value is provably above Tu cannot enter the list whatever its row id.
                //   dot:    d = 1 - acc is the reference's own float64 value (distances.go:89); d > Tu => float32(d) >= Tu.
                //   cosine: sim = acc / (qn*rn) (distances.go:30).  acc < P - |P| 2^-48 with P = ((1 - Tu) - 2^-50) qn rn
                //           puts the rounded quotient below 1 - Tu - 2^-51, so 1 - sim rounds to >= Tu.  (Tu > 2 — a
                //           threshold at the clamp — zero norms and NaN all fall through to the exact path.)
                if (!first) {
                    const uint32_t tk = (uint32_t)(thrv[nb][r] >> 32);
                    bool cand = tk >= 0xFFFFFFFDu;                       // no threshold yet, or a NaN distance as threshold
                    const double tu = (double)unord_f32(cand ? 0x80000000u : tk + 1);
                    if constexpr (M == QV_DOT) {
#pragma unroll
                        for (int g = 0; g < 4; g++) cand |= !((1.0 - acc[nb][g][r]) > tu);
                    } else {
                        cand |= tu > 2.0;
                        const double s0q = ((1.0 - tu) - 0x1p-50) * qc.qn;
#pragma unroll
                        for (int g = 0; g < 4; g++) {
                            const double pth = s0q * rn[g];
                            cand |= !(acc[nb][g][r] < pth - __builtin_fabs(pth) * 0x1p-48);
                        }
                    }
                    if (__ballot(cand) == 0) continue;
                }
                uint64_t key[4];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const uint32_t row = t * 64 + g * 16 + j16;
                    const bool live = (alive_word >> (g * 16 + j16)) & 1ull;
                    key[g] = live ? make_key(finalize<M>(acc[nb][g][r], qc, rn[g]), row) : kDeadKey;
                }
                // the exact keys against the thresholds; then regroup the 4 x 64 candidates of these four queries through
                // LDS to one per lane and hand each query's 64 to its list
                uint64_t hits = ~0ull;
                if (!first) {
                    hits = __ballot(key[0] < thrv[nb][r] || key[1] < thrv[nb][r] || key[2] < thrv[nb][r] || key[3] < thrv[nb][r]);
                    if (hits == 0) continue;
                }
#pragma unroll
                for (int g = 0; g < 4; g++) my_scratch[blk * 64 + g * 16 + j16] = key[g];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int qb = 0; qb < 4; qb++) {
                    if (((hits >> (16 * qb)) & 0xFFFFull) == 0) continue;
                    const uint64_t cand = my_scratch[qb * 64 + lane];
                    uint64_t& l = list[nb * 16 + r * 4 + qb];
                    uint64_t thr = readlane64(thrv[nb][r], qb * 16);          // this query's threshold (lanes of block qb hold it)
                    if (first) { l = sort_out_of_line(cand, lane); thr = readlane64(l, kth); }
                    else l = insert_out_of_line(l, cand, thr, kth, lane);
                    if (blk == (uint32_t)qb) thrv[nb][r] = thr;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        first = false;
    }

    if (wave_lists) {
#pragma unroll
        for (int i = 0; i < Q; i++)
            if (q0 + i < nq && lane < k) partial[((size_t)(q0 + i) * gridDim.x * W + blockIdx.x * W + wave) * k + lane] = list[i];
        return;
    }
    // merge each set's per-wave lists per query (the query fragments are dead: reuse their LDS)
    __syncthreads();
    lds_u64* wl = (lds_u64*)smem + (size_t)half * (W - 1) * Q * 64;      // [waves-1][Q][64] per set
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < Q; i++) wl[((wave - 1) * Q + i) * 64 + lane] = list[i];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < Q; i++) {
            uint64_t thr = readlane64(list[i], kth);
            for (uint32_t w = 0; w + 1 < (uint32_t)W; w++) {
                const uint64_t key = lane < k ? wl[(w * Q + i) * 64 + lane] : kDeadKey;
                if (__ballot(key < thr) == 0) continue;
                list[i] = insert_out_of_line(list[i], key, thr, kth, lane);
            }
            if (q0 + i < nq && lane < k) partial[((size_t)(q0 + i) * gridDim.x + blockIdx.x) * k + lane] = list[i];
        }
    }
}

// The full pass once every query has a bound: no lists at all.  A row whose key is below its query's bound is appended to a
// small candidate buffer in LDS ((key, query slot); 0.04 rows per query and tile with a 256-tile sample), and after the last tile
// each wave selects the top-k of a few queries from that buffer.  Without 16*NB lists, thresholds and per-query constants in
// registers the kernel fits 3 waves per SIMD (12 per CU) instead of 2 — and a SIMD with a single runnable wave issues f64
// MFMAs at half rate (mfma_f64_rate.hip), which is what every stall of one of two waves cost.  If a workgroup's buffer
// overflows (a bound that admits too much: a sample with fewer than k live rows, or near-duplicates of a query's neighbours
// concentrated in one workgroup's tiles) it raises *overflow and the register-list kernel redoes the pass.
template <int M, int NB, int U, int WV>
__global__ void __launch_bounds__(64 * WV, 1)
k_mq64_bounded(IndexView v, const float* __restrict__ qfrag_g, const double* __restrict__ qconst_g, uint32_t nq, uint32_t k,
               const uint64_t* __restrict__ bound, uint32_t cap, uint32_t* __restrict__ overflow, uint64_t* __restrict__ partial) {
    static_assert(M == QV_COSINE || M == QV_DOT, "the f64 matrix path covers the fma(q, x, acc) metrics");
    constexpr int Q = 16 * NB;
    extern __shared__ __align__(16) unsigned char smem[];
    typedef __attribute__((address_space(3))) float lds_f32;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((address_space(3))) uint64_t lds_u64;
    typedef __attribute__((address_space(3))) double lds_f64;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    float* qf = reinterpret_cast<float*>(smem);                                   // [NB][dim4][64]
    const lds_f32* qf3 = (const lds_f32*)smem;
    const size_t qf_bytes = (size_t)NB * v.dim4 * 64 * sizeof(float);
    lds_u8* base3 = (lds_u8*)smem + qf_bytes;
    lds_u64* bnd = (lds_u64*)base3;                                               // [Q] bound keys
    lds_f64* qnl = (lds_f64*)(base3 + Q * 8);                                     // [Q] query norms
    lds_u32* cnt = (lds_u32*)(base3 + Q * 16);                                    // candidates appended (may exceed cap)
    lds_u64* ckey = (lds_u64*)(base3 + Q * 16 + 16);                              // [cap]
    lds_u32* cq = (lds_u32*)(base3 + Q * 16 + 16 + (size_t)cap * 8);              // [cap] query slot of the candidate
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t grp = blockIdx.y, q0 = grp * Q;
    const uint32_t blk = lane >> 4, j16 = lane & 15;
    {
        const f4* src = reinterpret_cast<const f4*>(qfrag_g + (size_t)grp * NB * v.dim4 * 64);
        f4* dst = reinterpret_cast<f4*>(qf);
        for (uint32_t i = threadIdx.x; i < (uint32_t)NB * v.dim4 * 16; i += blockDim.x) dst[i] = src[i];
        if (threadIdx.x < Q) {
            const uint32_t qi = q0 + threadIdx.x;
            bnd[threadIdx.x] = bound[qi < nq ? qi : nq - 1];
            qnl[threadIdx.x] = qconst_g[qi];
        }
        if (threadIdx.x == 0) *cnt = 0;
    }
    __syncthreads();

    const f4* tiles = reinterpret_cast<const f4*>(v.tiles);
    const uint32_t tw = gridDim.x * WV;
    auto tile_ptr = [&](uint32_t t) { return reinterpret_cast<const float*>(tiles + (size_t)t * v.dim4 * 64) + j16 * 4 + blk; };
    float xa[U][4], xb[U][4];
    float qa[U][NB], qb2[U][NB];
    const uint32_t npair = v.dim4 / (2 * U);          // pairs of U-chunk register blocks; the remaining chunks go one at a time
    auto load_x_from = [&](const float* pb, float (&x)[U][4], uint32_t c0) {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int g = 0; g < 4; g++) x[u][g] = pb[(size_t)(c0 + u) * 256 + g * 64];
    };
    auto load_q = [&](float (&q)[U][NB], uint32_t c0) {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int nb = 0; nb < NB; nb++) q[u][nb] = qf3[((uint32_t)nb * v.dim4 + c0 + u) * 64 + lane];
    };
    const uint32_t t0 = blockIdx.x * WV + wave;
    if (t0 < v.n_tiles && npair) { load_x_from(tile_ptr(t0), xa, 0); load_q(qa, 0); }

    for (uint32_t t = t0; t < v.n_tiles; t += tw) {
        const float* pb = tile_ptr(t);
        d4 acc[NB][4];
#pragma unroll
        for (int nb = 0; nb < NB; nb++)
#pragma unroll
            for (int g = 0; g < 4; g++) acc[nb][g] = (d4){0.0, 0.0, 0.0, 0.0};
        auto step = [&](const float (&y)[4], const float (&qv)[NB]) {
            const double b0 = (double)y[0], b1 = (double)y[1], b2 = (double)y[2], b3 = (double)y[3];
#pragma unroll
            for (int nb = 0; nb < NB; nb++) {
                const double a = (double)qv[nb];
                acc[nb][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc[nb][0], 0, 0, 0);
                acc[nb][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc[nb][1], 0, 0, 0);
                acc[nb][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, acc[nb][2], 0, 0, 0);
                acc[nb][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b3, acc[nb][3], 0, 0, 0);
            }
        };
        auto consume = [&](const float (&x)[U][4], const float (&qv)[U][NB]) {
#pragma unroll
            for (int u = 0; u < U; u++) step(x[u], qv[u]);
#pragma unroll
            for (int nb = 0; nb < NB; nb++)
#pragma unroll
                for (int g = 0; g < 4; g++) asm volatile("" : "+v"(acc[nb][g]));
        };
        // Register blocks in pairs, every load of the steady state unconditional: the compiler's wait counts are then exact
        // (vmcnt(16) before a block is consumed = only the block just requested may be outstanding).  With a load under a
        // condition in the loop it merged the two paths' counters and waited for the block it had just requested, every
        // second block (profiles/r02_mq64_breakdown.txt).  The last pair requests block 0 of the wave's next tile (its own
        // tile again when there is none: a harmless reload) before the epilogue.
        for (uint32_t pr = 0; pr + 1 < npair; pr++) {
            load_x_from(pb, xb, (2 * pr + 1) * U); load_q(qb2, (2 * pr + 1) * U);
            __builtin_amdgcn_sched_barrier(0);        // the scheduler would sink the requests below the block they overlap
            consume(xa, qa);
            __builtin_amdgcn_sched_barrier(0);
            load_x_from(pb, xa, (2 * pr + 2) * U); load_q(qa, (2 * pr + 2) * U);
            __builtin_amdgcn_sched_barrier(0);
            consume(xb, qb2);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (npair) {
            load_x_from(pb, xb, (2 * npair - 1) * U); load_q(qb2, (2 * npair - 1) * U);
            __builtin_amdgcn_sched_barrier(0);
            consume(xa, qa);
            __builtin_amdgcn_sched_barrier(0);
            load_x_from(tile_ptr(t + tw < v.n_tiles ? t + tw : t), xa, 0); load_q(qa, 0);
            __builtin_amdgcn_sched_barrier(0);
            consume(xb, qb2);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (uint32_t c = 2 * npair * U; c < v.dim4; c++) {
            float q1[NB], x1[4];
#pragma unroll
            for (int nb = 0; nb < NB; nb++) q1[nb] = qf3[((uint32_t)nb * v.dim4 + c) * 64 + lane];
#pragma unroll
            for (int g = 0; g < 4; g++) x1[g] = pb[(size_t)c * 256 + g * 64];
            step(x1, q1);
        }

        // epilogue: lane (blk, j16), register (nb, g, r) = (query slot 16nb+4r+blk, row 64t+16g+j16); the one-sided test of
        // k_flat_scan_mq64 against the query's bound, then the exact key for the few that may pass
        double rn[4];
        const uint64_t alive_word = v.alive[t];
#pragma unroll
        for (int g = 0; g < 4; g++) { rn[g] = 0.0; if constexpr (MT<M>::needs_rnorm) rn[g] = v.rnorm[(size_t)t * 64 + g * 16 + j16]; }
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t qs = (uint32_t)(nb * 16 + r * 4) + blk;
                const uint64_t bk = bnd[qs];
                QConst qc; qc.qn = qnl[qs]; qc.qn32 = 0.0f;
                const uint32_t tk = (uint32_t)(bk >> 32);
                bool cand = tk >= 0xFFFFFFFDu;
                const double tu = (double)unord_f32(cand ? 0x80000000u : tk + 1);
                if constexpr (M == QV_DOT) {
#pragma unroll
                    for (int g = 0; g < 4; g++) cand |= !((1.0 - acc[nb][g][r]) > tu);
                } else {
                    cand |= tu > 2.0;
                    const double s0q = ((1.0 - tu) - 0x1p-50) * qc.qn;
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const double pth = s0q * rn[g];
                        cand |= !(acc[nb][g][r] < pth - __builtin_fabs(pth) * 0x1p-48);
                    }
                }
                if (__ballot(cand) == 0) continue;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const uint32_t row = t * 64 + g * 16 + j16;
                    const bool live = (alive_word >> (g * 16 + j16)) & 1ull;
                    const uint64_t key = live ? make_key(finalize<M>(acc[nb][g][r], qc, rn[g]), row) : kDeadKey;
                    if (key < bk) {
                        const uint32_t pos = __hip_atomic_fetch_add((uint32_t*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (pos < cap) { ckey[pos] = key; cq[pos] = qs; }
                    }
                }
            }
        }
    }

    __syncthreads();
    const uint32_t total = *cnt;
    if (total > cap) { if (threadIdx.x == 0) *overflow = 1; return; }             // the fallback launch redoes the pass
    const uint32_t kth = k - 1;
    for (uint32_t i = wave; i < (uint32_t)Q; i += WV) {
        uint64_t list = kDeadKey, thr = kDeadKey;
        for (uint32_t b = 0; b < total; b += 64) {
            const uint32_t j = b + lane;
            const uint64_t key = (j < total && cq[j] == i) ? ckey[j] : kDeadKey;
            if (__ballot(key < thr) == 0) continue;
            list = insert_out_of_line(list, key, thr, kth, lane);
        }
        if (q0 + i < nq && lane < k) partial[((size_t)(q0 + i) * gridDim.x + blockIdx.x) * k + lane] = list;
    }
}

static size_t mq64_lds_bytes(uint32_t dim4, const Mq64Shape& sh) {
    const size_t qf = (size_t)sh.h * sh.nb * dim4 * 64 * sizeof(float);
    const size_t merge = (size_t)sh.h * (sh.w - 1) * 16 * sh.nb * 64 * sizeof(uint64_t);
    return std::max(qf + (size_t)sh.h * sh.w * 4 * 64 * sizeof(uint64_t), merge);
}
// 0 = not applicable (metric / dimension); else queries per pass
int mq64_blocks(int metric, uint32_t dim4, uint32_t nq) {
    static const int enabled = env_int("QV_MQ64", 1);
    if (enabled != 1 || (metric != QV_COSINE && metric != QV_DOT)) return 0;
    const Mq64Shape sh = mq64_shape(nq);
    if (mq64_lds_bytes(dim4, sh) * sh.wgs > 158 * 1024) return 0;
    return 16 * sh.h * sh.nb;
}
// query fragments + query constants + the sample pass's bounds
size_t mq64_workspace_bytes(uint32_t nq, uint32_t dim4) {
    return (size_t)(nq + 32) * dim4 * 4 * sizeof(float) + 256 + (size_t)(nq + 32) * sizeof(double) + 256 + (size_t)(nq + 32) * sizeof(uint64_t) + 256 + 256;
}

// partial[(q * grid + wg) * k + i]; returns the grid used through *grid_out.  ev0/ev1 bracket the full pass.
hipError_t launch_flat_scan_mq64(const IndexView& v, int cus, const float* d_queries, uint32_t nq, uint32_t k, void* d_qws, uint64_t* partial,
                                 uint32_t* grid_out, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (mq64_blocks(v.metric, v.dim4, nq) == 0) return hipErrorInvalidValue;
    const Mq64Shape sh = mq64_shape(nq);
    const uint32_t B = (uint32_t)(sh.h * sh.nb), Q = 16u * B, groups = (nq + Q - 1) / Q;
    auto up256 = [](size_t x) { return (x + 255) / 256 * 256; };
    unsigned char* ws = static_cast<unsigned char*>(d_qws);
    float* qfrag = reinterpret_cast<float*>(ws);
    double* qconst = reinterpret_cast<double*>(ws + up256((size_t)groups * Q * v.dim4 * 4 * sizeof(float)));
    uint64_t* bound = reinterpret_cast<uint64_t*>(reinterpret_cast<unsigned char*>(qconst) + up256((size_t)groups * Q * sizeof(double)));
    uint32_t* overflow = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(bound) + up256((size_t)groups * Q * sizeof(uint64_t)));
    const uint32_t want = (v.n_tiles + (uint32_t)sh.w - 1) / (uint32_t)sh.w;
    uint32_t grid = std::max(1u, std::min(want, (uint32_t)cus * (uint32_t)sh.wgs));
    // the sample: one tile per wave of `sgrid` workgroups, spread evenly over the corpus (QV_MQ64_SAMPLE tiles, 0 = no sample
    // pass: every wave learns its thresholds from its own tiles, the round-1 behaviour)
    static const int sample_env = dev_env_int("QV_MQ64_SAMPLE", 256);
    const uint32_t stiles = std::min<uint32_t>(std::min<uint32_t>((uint32_t)std::max(sample_env, 0), v.n_tiles / 8), 4u * (uint32_t)cus);   // the partial buffer holds 8 lists per CU and query
    const uint32_t sgrid = stiles / (uint32_t)sh.w;
    const uint32_t n_s = sgrid * (uint32_t)sh.w, step_s = n_s ? v.n_tiles / n_s : 1;
    const size_t lds = mq64_lds_bytes(v.dim4, sh);
    hipError_t e = hipSuccess;
#define QV_MQ64B(MMM, BNB_, BU_, BWV_)                                                                                           \
    e = set_lds(k_mq64_bounded<MMM, BNB_, BU_, BWV_>, lds_b);                                                                     \
    if (e != hipSuccess) return e;                                                                                                \
    hipLaunchKernelGGL((k_mq64_bounded<MMM, BNB_, BU_, BWV_>), dim3(grid, groups), dim3(64 * BWV_), lds_b, s, v, qfrag, qconst, nq, k, \
                       (const uint64_t*)bound, (uint32_t)kMq64Cap, overflow, partial);
#define QV_MQ64(MMM, HH, NBB, UU, WW, GG)                                                                                         \
    {                                                                                                                             \
        const uint32_t per = (uint32_t)(HH * NBB) * v.dim4 * 64;                                                                   \
        hipLaunchKernelGGL((k_mq64_prep<MMM, HH * NBB>), dim3(std::min<uint32_t>((per + 255) / 256, 64), groups), dim3(256), 0, s, d_queries, nq, v.dim, v.dim4, qfrag, qconst); \
        e = set_lds(k_flat_scan_mq64<MMM, HH, NBB, UU, WW, GG>, lds);                                                             \
        if (e != hipSuccess) return e;                                                                                            \
        if (ev0) (void)hipEventRecord(ev0, s);                                                                                    \
        if (sgrid) {                                                                                                              \
            constexpr int BNB = HH * NBB, BU = BNB == 2 ? 4 : 8, BWV = 12;                                                         \
            hipLaunchKernelGGL((k_flat_scan_mq64<MMM, HH, NBB, UU, WW, GG>), dim3(sgrid, groups), dim3(64 * WW * HH), lds, s, v, qfrag, qconst, nq, k, \
                               (const uint64_t*)nullptr, n_s, step_s, 1u, (const uint32_t*)nullptr, partial);                       \
            hipLaunchKernelGGL(k_mq64_bound, dim3(nq), dim3(64), 0, s, partial, sgrid * (uint32_t)(WW), k, bound, overflow);       \
            const size_t lds_b = (size_t)BNB * v.dim4 * 64 * sizeof(float) + 16 * BNB * 16 + 16 + (size_t)kMq64Cap * 12;          \
            grid = (uint32_t)cus;                                                                                                 \
            QV_MQ64B(MMM, BNB, BU, BWV)                                                                                           \
            hipLaunchKernelGGL((k_flat_scan_mq64<MMM, HH, NBB, UU, WW, GG>), dim3(grid, groups), dim3(64 * WW * HH), lds, s, v, qfrag, qconst, nq, k, \
                               (const uint64_t*)nullptr, v.n_tiles, 1u, 0u, (const uint32_t*)overflow, partial);                   \
        } else {                                                                                                                  \
            hipLaunchKernelGGL((k_flat_scan_mq64<MMM, HH, NBB, UU, WW, GG>), dim3(grid, groups), dim3(64 * WW * HH), lds, s, v, qfrag, qconst, nq, k, \
                               (const uint64_t*)nullptr, v.n_tiles, 1u, 0u, (const uint32_t*)nullptr, partial);                     \
        }                                                                                                                         \
        if (ev1) (void)hipEventRecord(ev1, s);                                                                                    \
    }
#ifdef QV_VARIANTS                                        // (the two-sets-sharing-tiles shape: QV_MQ64_MODE=2 of the measurement build only)
#define QV_MQ64_SHAPES(MMM)                                                                  \
    if (sh.h == 1 && sh.nb == 1) QV_MQ64(MMM, 1, 1, 8, 4, 2)                                  \
    else if (sh.h == 2) QV_MQ64(MMM, 2, 1, 8, 4, 1)                                           \
    else QV_MQ64(MMM, 1, 2, 4, 8, 1)
#else
#define QV_MQ64_SHAPES(MMM)                                                                  \
    if (sh.h == 1 && sh.nb == 1) QV_MQ64(MMM, 1, 1, 8, 4, 2)                                  \
    else QV_MQ64(MMM, 1, 2, 4, 8, 1)
#endif
    if (v.metric == QV_COSINE) { QV_MQ64_SHAPES(QV_COSINE) } else { QV_MQ64_SHAPES(QV_DOT) }
#undef QV_MQ64_SHAPES
#undef QV_MQ64
    *grid_out = grid;
    static const int trace = env_int("QV_TRACE", 0);
    if (trace && sgrid) {                           // debugging aid only: a synchronous read of the overflow flag
        uint32_t h = 0;
        if (hipMemcpyAsync(&h, overflow, sizeof h, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess)
            fprintf(stderr, "qv: k_mq64_bounded %s\n", h ? "overflowed: the register-list kernel redid the pass" : "held every candidate");
    }
    return hipGetLastError();
}

}  // namespace qv
