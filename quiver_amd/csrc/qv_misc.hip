// qv_misc.hip — row gathers, pair distances, ingest, synthetic generator, tombstones
// (shared helpers, the arithmetic contract and the build flags: qv_kernels.h)
#include <algorithm>
#include "qv_kernels.h"

namespace qv {

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
    // a lane walks ITS row: from the row-major copy when the index keeps one (consecutive 16-byte chunks share cache lines:
    // a quarter of the line transactions of the tile layout, where every chunk of a row sits in its own line)
    typename MT<M>::A acc;
    if (v.rowmaj != nullptr && (v.dim & 3) == 0) acc = row_accumulate<M, U, false>(reinterpret_cast<const f4*>(v.rowmaj + (size_t)row * v.dim), 1, q_lds, v.dim4);
    else acc = row_accumulate<M, U, false>(reinterpret_cast<const f4*>(v.tiles) + (size_t)(row >> 6) * v.dim4 * 64 + (row & 63), 64, q_lds, v.dim4);
    double rn = 0.0;
    if constexpr (MT<M>::needs_rnorm) rn = v.rnorm[row];
    out[i] = finalize<M>(acc, qc, rn);
}

// lane == pair; a, b row-major [n][dim]; plain scalar walk (dim need not be a multiple of 4)
template <int M>
__global__ void __launch_bounds__(64)
k_distance_pairs(const float* __restrict__ a, const float* __restrict__ b, uint32_t n, uint32_t dim, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    out[i] = pair_distance<M>(a + (size_t)i * dim, b + (size_t)i * dim, dim);
}

// the same arithmetic on the host, for ONE pair: qv_distance_pair (include/qv.h)
float host_pair_distance(int metric, const float* a, const float* b, uint32_t dim) {
    QV_DISPATCH_METRIC(metric, { return pair_distance<MM>(a, b, dim); });
    return 0.0f;
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

// Rows whose dimension is a multiple of 4 (every 16-byte chunk whole): one 256-thread workgroup per touched tile moves 64 rows x
// 16 chunks at a time through LDS — the source is read in 256-byte runs of a row, the tile layout ([chunk][64 rows][4 dims]) is
// written in 1 KiB runs of a chunk, and the row-major copy (when the index keeps one) is written from the registers that read
// the source instead of by a second pass over it.  (k_ingest below reads 16 bytes per lane at a stride of one row: every
// line of the source is fetched eight times over; measured 1M x 768: 5.7 ms, 8.1 ms with the row-major copy.)
// The per-row constants are computed by one lane per row in element order, as k_ingest does (distances.go:21, adapter.go:119).
constexpr int kIngestCB = 16;                               // chunks per LDS block
__global__ void __launch_bounds__(256)
k_ingest_tiled(IndexView v, const float* __restrict__ src, uint32_t row0, uint32_t n, uint32_t tile0) {
    __shared__ float4 stage[64][kIngestCB + 1];             // +1: the transposed reads of the store phase hit distinct banks
    const uint32_t t = tile0 + blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* tiles = reinterpret_cast<float4*>(v.tiles) + (size_t)t * v.dim4 * 64;
    float4* rm4 = v.rowmaj ? reinterpret_cast<float4*>(v.rowmaj) : nullptr;
    const bool f64norm = v.metric == QV_COSINE || v.metric == QV_DOT || v.metric == QV_L2 || v.metric == QV_L2SQ;
    const bool f32norm = v.metric == QV_COSINE_F32;
    double mb = 0.0; float nb = 0.0f;
    const uint32_t my_row = t * 64 + tid;                   // wave 0: the row whose constants this lane accumulates
    const bool my_mine = tid < 64 && my_row >= row0 && my_row < row0 + n;
    for (uint32_t cb = 0; cb < v.dim4; cb += kIngestCB) {
#pragma unroll
        for (int i = 0; i < 4; i++) {                       // 16 consecutive threads read 256 contiguous bytes of one row
            const uint32_t idx = tid + 256u * i, r = idx / kIngestCB, ch = idx % kIngestCB;
            const uint32_t row = t * 64 + r;
            float4 x = {0.f, 0.f, 0.f, 0.f};
            if (row >= row0 && row < row0 + n && cb + ch < v.dim4) {
                x = s4[(size_t)(row - row0) * v.dim4 + cb + ch];
                if (rm4) rm4[(size_t)row * v.dim4 + cb + ch] = x;
            }
            stage[r][ch] = x;
        }
        __syncthreads();
        if (my_mine) {
            const uint32_t nch = v.dim4 - cb < (uint32_t)kIngestCB ? v.dim4 - cb : (uint32_t)kIngestCB;
            for (uint32_t ch = 0; ch < nch; ch++) {
                const float4 x = stage[tid][ch];
                if (f64norm) {
                    mb = __builtin_fma((double)x.x, (double)x.x, mb); mb = __builtin_fma((double)x.y, (double)x.y, mb);
                    mb = __builtin_fma((double)x.z, (double)x.z, mb); mb = __builtin_fma((double)x.w, (double)x.w, mb);
                } else if (f32norm) {
                    float p; p = x.x * x.x; nb = nb + p; p = x.y * x.y; nb = nb + p; p = x.z * x.z; nb = nb + p; p = x.w * x.w; nb = nb + p;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {                       // 64 consecutive threads write the 1 KiB of one chunk
            const uint32_t idx = tid + 256u * i, ch = idx / 64, r = idx % 64;
            const uint32_t row = t * 64 + r;
            if (row >= row0 && row < row0 + n && cb + ch < v.dim4) tiles[(size_t)(cb + ch) * 64 + r] = stage[r][ch];
        }
        __syncthreads();
    }
    if (my_mine) {
        if (f64norm) v.rnorm[my_row] = __builtin_sqrt(mb);
        else if (f32norm) v.rnorm[my_row] = nb == 0.0f ? -1.0 : (double)(float)__builtin_sqrt((double)nb);
    }
    if (tid < 64) {
        const uint64_t m = __ballot(my_mine);
        if (tid == 0 && m) atomicOr(reinterpret_cast<unsigned long long*>(&v.alive[t]), (unsigned long long)m);
    }
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

// The bfloat16 copy of tiles [t0, t1] (QV_FLAG_BF16_ROWS): [tile][16-dim step][32-row block][8-dim half][row of the block][8 values] — the
// 64 lanes of a wave that wants (step, block) as its MFMA B operand read ONE contiguous KiB; value = bf16(float32 row value), round to
// nearest even — what the batched filter would compute on the fly.  One thread per (tile, 8-dim group, row); refreshed by every
// launcher that writes rows, on the same stream, so the plane is never behind the tiles.
__global__ void __launch_bounds__(256)
k_bf16_plane(IndexView v, uint32_t t0, uint32_t n_tiles) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const uint32_t dim8 = (v.dim4 + 1) / 2;
    const uint64_t total = (uint64_t)n_tiles * dim8 * 64;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i & 63), c8 = (uint32_t)((i >> 6) % dim8), t = t0 + (uint32_t)((i >> 6) / dim8);
        const f4* src = reinterpret_cast<const f4*>(v.tiles) + (size_t)t * v.dim4 * 64 + r;
        const f4 a = src[(size_t)(2 * c8) * 64];
        f4 b = {0.f, 0.f, 0.f, 0.f};
        if (2 * c8 + 1 < v.dim4) b = src[(size_t)(2 * c8 + 1) * 64];
        const bf2 p0 = {(__bf16)a.x, (__bf16)a.y}, p1 = {(__bf16)a.z, (__bf16)a.w}, p2 = {(__bf16)b.x, (__bf16)b.y}, p3 = {(__bf16)b.z, (__bf16)b.w};
        uint4 o;
        o.x = __builtin_bit_cast(uint32_t, p0); o.y = __builtin_bit_cast(uint32_t, p1); o.z = __builtin_bit_cast(uint32_t, p2); o.w = __builtin_bit_cast(uint32_t, p3);
        const uint32_t steps8 = (dim8 + 1) / 2;
        reinterpret_cast<uint4*>(v.bf16)[((((size_t)t * steps8 + (c8 >> 1)) * 2 + (r >> 5)) * 2 + (c8 & 1)) * 32 + (r & 31)] = o;
    }
}
// |r - bf16(r)| of the rows of tiles [t0, t0 + n_tiles): what the one-term bfloat16 filter drops of a row (qv_batched.hip: its
// margin is |q - qh||r| + |qh||r - rh|, a third of the worst case 2 * 2^-8 |q||r| on ordinary data).  Lane == row, one wave per
// tile, 1 KiB per request; the value only feeds a bound, so it is rounded UP (float64 sum, correctly rounded sqrt, one more
// float32 step than round-to-nearest) and the order of the additions is free.  Elements below 2^-126 in magnitude count whole:
// the matrix core may flush such an operand.
__global__ void __launch_bounds__(64)
k_row_residual(IndexView v, uint32_t t0) {
    const uint32_t t = t0 + blockIdx.x, lane = threadIdx.x;
    const f4* src = reinterpret_cast<const f4*>(v.tiles) + (size_t)t * v.dim4 * 64 + lane;
    double s2 = 0.0;
    auto one = [&](float x) {
        const __bf16 hb = (__bf16)x;
        float h = (float)hb;
        if (__builtin_fabsf(h) < 1.17549435e-38f) h = 0.f;
        const double d = (double)x - (double)h;
        s2 = __builtin_fma(d, d, s2);
    };
    for (uint32_t c = 0; c < v.dim4; c += 4) {
        f4 x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) x[u] = c + u < v.dim4 ? src[(size_t)(c + u) * 64] : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; u++) { one(x[u].x); one(x[u].y); one(x[u].z); one(x[u].w); }
    }
    const float r = (float)(__builtin_sqrt(s2) * (1.0 + 1e-12));
    // next float up (NaN / inf stay: such a row's norm is NaN / inf too and the filter passes it on)
    float up = r;
    if (r == r && r != __builtin_inff()) up = r == 0.f ? (s2 == 0.0 ? 0.f : __uint_as_float(1u)) : __uint_as_float(__float_as_uint(r) + 1u);
    v.rres[(size_t)t * 64 + lane] = up;
}
static hipError_t refresh_bf16(const IndexView& v, uint32_t t0, uint32_t t1, hipStream_t s) {
    if (v.rres && (v.metric == QV_COSINE || v.metric == QV_DOT || v.metric == QV_L2 || v.metric == QV_L2SQ)) {
        hipLaunchKernelGGL(k_row_residual, dim3(t1 - t0 + 1), dim3(64), 0, s, v, t0);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (!v.bf16) return hipSuccess;
    const uint64_t total = (uint64_t)(t1 - t0 + 1) * ((v.dim4 + 1) / 2) * 64;
    hipLaunchKernelGGL(k_bf16_plane, dim3((uint32_t)std::min<uint64_t>((total + 255) / 256, 65536)), dim3(256), 0, s, v, t0, t1 - t0 + 1);
    return hipGetLastError();
}

hipError_t launch_ingest(const IndexView& v, const float* d_rows, uint32_t row0, uint32_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint32_t t0 = row0 / 64, t1 = (row0 + n - 1) / 64;
    static const int tiled = dev_env_int("QV_INGEST_TILED", 1);
    if ((v.dim & 3) == 0 && tiled == 1 && (reinterpret_cast<uintptr_t>(d_rows) & 15) == 0) {
        hipLaunchKernelGGL(k_ingest_tiled, dim3(t1 - t0 + 1), dim3(256), 0, s, v, d_rows, row0, n, t0);
        hipError_t e0 = hipGetLastError();
        return e0 != hipSuccess ? e0 : refresh_bf16(v, t0, t1, s);
    }
    hipLaunchKernelGGL(k_ingest, dim3(t1 - t0 + 1), dim3(64), 0, s, v, d_rows, row0, n, t0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (v.rowmaj) e = hipMemcpyAsync(v.rowmaj + (size_t)row0 * v.dim, d_rows, (size_t)n * v.dim * sizeof(float), hipMemcpyDeviceToDevice, s);
    return e != hipSuccess ? e : refresh_bf16(v, t0, t1, s);
}

hipError_t launch_generate(const IndexView& v, uint64_t seed, uint64_t gen_row0, uint32_t row0, uint32_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint32_t t0 = row0 / 64, t1 = (row0 + n - 1) / 64;
    hipLaunchKernelGGL(k_generate, dim3(t1 - t0 + 1), dim3(64), 0, s, v, seed, gen_row0, row0, n, t0);
    hipError_t e = hipGetLastError();
    return e != hipSuccess ? e : refresh_bf16(v, t0, t1, s);
}

hipError_t launch_set_alive(const IndexView& v, const uint32_t* d_rows, uint32_t n, int alive, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_set_alive, dim3((n + 255) / 256), dim3(256), 0, s, v, d_rows, n, alive);
    return hipGetLastError();
}

// n listed rows -> row-major [n][dim]: blockIdx.y = listed row, one thread per element (a row's elements sit 64 floats apart in
// its tile: a gather by nature; this is the Save / GetVector path, not a scan)
__global__ void k_fetch_rows(IndexView v, const uint32_t* __restrict__ rows, float* __restrict__ out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= v.dim) return;
    const uint32_t row = rows[blockIdx.y];
    out[(size_t)blockIdx.y * v.dim + j] = v.tiles[((size_t)(row >> 6) * v.dim4 + (j >> 2)) * 256 + (row & 63) * 4 + (j & 3)];
}

hipError_t launch_fetch_rows(const IndexView& v, const uint32_t* d_rows, uint32_t n, float* d_out, hipStream_t s) {
    for (uint32_t done = 0; done < n; done += 65535) {                    // gridDim.y limit
        const uint32_t m = n - done < 65535 ? n - done : 65535;
        hipLaunchKernelGGL(k_fetch_rows, dim3((v.dim + 255) / 256, m), dim3(256), 0, s, v, d_rows + done, d_out + (size_t)done * v.dim);
    }
    return hipGetLastError();
}

hipError_t launch_fetch_row(const IndexView& v, uint32_t row, float* d_out, hipStream_t s) {
    hipLaunchKernelGGL(k_fetch_row, dim3((v.dim + 255) / 256), dim3(256), 0, s, v, row, d_out);
    return hipGetLastError();
}


}  // namespace qv
