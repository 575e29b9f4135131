// qv_scan.hip — flat scan + fused top-k (single- and multi-query), list merges
// (shared helpers, the arithmetic contract and the build flags: qv_kernels.h)
#include "qv_kernels.h"

namespace qv {

// ---------------------------------------------------------------- flat scan --------
// grid = (workgroups, nq); each wave walks tiles gw, gw+tw, ... ; lane == row.
// Output: partial[(q*gridDim.x + blockIdx.x)*k + i] = workgroup's i-th best key.

__device__ void merge_lists_last_workgroup(const uint64_t* src, uint32_t n_lists, uint32_t k, uint32_t* rows_out, float* dist_out, uint32_t* done_flag = nullptr, uint32_t done_seq = 0);

// FUSE: the workgroups publish their lists with returning atomic exchanges and take a ticket; the LAST one to finish merges all of
// them (k_merge_lists' own code) and writes the final rows / distances — one launch per query instead of two (round 5: a single
// query over 1M x 768 is 0.44 ms of scan; the second launch and the gap in front of it were 2 % of it).  tickets[qi]: zero before
// the first launch, left zero.  Nothing crosses workgroups through a fence (see k_flat_scan_small).
template <int M, int U, bool FUSE = false>
__global__ void __launch_bounds__(kScanBlock)
k_flat_scan(IndexView v, const float* __restrict__ queries, uint32_t k, uint64_t* __restrict__ partial,
            uint32_t* __restrict__ tickets = nullptr, uint32_t* __restrict__ rows_out = nullptr, float* __restrict__ dist_out = nullptr,
            uint32_t* done_flag = nullptr, uint32_t done_seq = 0) {
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
        if constexpr (!FUSE) {
            if (lane < k) partial[((size_t)qi * gridDim.x + blockIdx.x) * k + lane] = list;
        } else {
            uint64_t* mine = partial + ((size_t)qi * gridDim.x + blockIdx.x) * k;
            if (lane < k) (void)atomicExch(reinterpret_cast<unsigned long long*>(&mine[lane]), (unsigned long long)list);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every lane's exchange has returned: the list is at the memory side
            uint32_t last = 0;
            if (lane == 0) last = atomicAdd(&tickets[qi], 1u) == gridDim.x - 1 ? 1u : 0u;
            last = __builtin_amdgcn_readfirstlane(last);
            if (last && lane == 0) tickets[qi] = 0;                    // for the next launch on this workspace (stream order)
            if (lane == 0) wl[0] = last;
        }
    }
    if constexpr (FUSE) {
        __syncthreads();
        if (!(uint32_t)wl[0]) return;
        __syncthreads();                                               // (wl is the merge's scratch next)
        merge_lists_last_workgroup(partial + (size_t)qi * gridDim.x * k, gridDim.x, k, rows_out + (size_t)qi * k, dist_out + (size_t)qi * k, done_flag, done_seq);
    }
}

// ---------------------------------------------------------------- flat scan of a short corpus: a tile over several waves --
// k_flat_scan gives a wave whole tiles: 64 rows x dim x 4 bytes (196 KB at 768 dimensions) streamed through 16 loads in flight and a
// dim-step float64 chain — ~30 us per tile whatever the corpus, which IS the scan below a few hundred thousand rows (measured, one
// query, 768 dimensions: 10 k rows 31 us, 30 k 57 us, 100 k 72 us of kernel against 4 / 13 / 43 us of HBM time).  Here the eight waves
// of a workgroup share a tile: wave w requests columns [w, w + 1) * dim / 8 of its 64 rows (every load in flight at once) and walks
// them, lane == row, as a partial chain; the tile's consumer wave (they take turns) adds the eight partial sums and takes the float32
// from the CERTIFICATE of qv_hnsw.hip ("a row's sum over several lanes, certified"): the reference's single chain lies within
// B = (2 dim + 128) u |q| |r| of the partial chains' sum, the float32 is a monotone function of that sum, so when both ends of the
// interval give the same float32 it is the reference's.  For cosine the query's norm is itself a sum here (the reference's is a chain
// over the same squares): its relative uncertainty d goes into the interval as (|S| + B) 2 d — the quotient S / (|q| |r|) is the same
// real number whether the error sits in S or in |q|.  A row that fails (a few in a million; every row at distance ~0) is walked again
// as ONE chain by row_accumulate, with the query's norm as a chain as well: the reference's arithmetic, unchanged.
// Lists are published and merged exactly as in k_flat_scan<., ., true> (returning exchanges, a ticket, the last workgroup merges).
constexpr int kSplitWaves = 8;
constexpr int kSplitBlock = 64 * kSplitWaves;
template <int M> struct ScanSplitOK { static constexpr bool value = M == QV_COSINE || M == QV_DOT || M == QV_L2 || M == QV_L1 || M == QV_L2SQ_F64; };
template <int M>
__global__ void __launch_bounds__(kSplitBlock)
k_flat_scan_split(IndexView v, const float* __restrict__ queries, uint32_t k, uint64_t* __restrict__ partial,
                  uint32_t* __restrict__ tickets, uint32_t* __restrict__ rows_out, float* __restrict__ dist_out, uint32_t* done_flag, uint32_t done_seq) {
    using Q = typename MT<M>::Q;
    using A = typename MT<M>::A;
    extern __shared__ __align__(16) unsigned char smem[];
    Q* q_lds = reinterpret_cast<Q*>(smem);
    const size_t q_bytes = (((size_t)v.dim4 * 4 * sizeof(Q)) + 15) / 16 * 16;
    double* part = reinterpret_cast<double*>(smem + q_bytes);                      // [2][kSplitWaves][64]
    uint64_t* wl = reinterpret_cast<uint64_t*>(smem + q_bytes + 2 * kSplitWaves * 64 * sizeof(double));   // [kSplitWaves][64]
    __shared__ double s_red[kSplitWaves];
    __shared__ uint32_t s_last;
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t qi = blockIdx.y;
    const float* q = queries + (size_t)qi * v.dim;
    stage_query<M>(q_lds, q, v.dim, v.dim4);
    __syncthreads();
    // |q|^2 as a sum over the workgroup (cosine, dot): its place in the certificate, see above.  (From the staged copy: the query
    // may live in host memory.)
    double qn_s = 0.0;
    if constexpr (M == QV_COSINE || M == QV_DOT) {
        double sq = 0.0;
        for (uint32_t i = threadIdx.x; i < v.dim4 * 4; i += kSplitBlock) { const double a = (double)q_lds[i]; sq = __builtin_fma(a, a, sq); }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq = sq + __shfl_xor(sq, off);
        if (lane == 0) s_red[wave] = sq;
        __syncthreads();
        sq = s_red[0];
#pragma unroll
        for (int w = 1; w < kSplitWaves; w++) sq = sq + s_red[w];
        qn_s = __builtin_sqrt(sq);
    }
    const double k_u = ((double)(2u * v.dim) + 128.0) * 0x1p-53;
    QConst qc; qc.qn = qn_s; qc.qn32 = 0.0f;                                         // (cosine: used with the widened interval; the fallback makes its own)
    QConst qc_exact; qc_exact.qn = 0.0; qc_exact.qn32 = 0.0f; bool have_exact = false;

    const uint32_t kth = k - 1;
    uint64_t list = kDeadKey, thr = kDeadKey;
    bool first = true;
    const f4* tiles = reinterpret_cast<const f4*>(v.tiles);
    const uint32_t c_lo = wave * v.dim4 / kSplitWaves, c_hi = (wave + 1) * v.dim4 / kSplitWaves;
    uint32_t it = 0;
    for (uint32_t t = blockIdx.x; t < v.n_tiles; t += gridDim.x, it++) {
        const uint32_t consumer = it % kSplitWaves;
        const uint32_t row = t * 64 + lane;
        double rn = 0.0; uint64_t am = 0;
        if (wave == consumer) {                                                      // (requested before the columns: they have arrived by the time they are used)
            if constexpr (MT<M>::needs_rnorm || M == QV_DOT) rn = v.rnorm[row];
            am = v.alive[t];
        }
        A acc = 0;
        if (c_lo < c_hi) acc = row_accumulate<M, 24, false, true, false>(tiles + ((size_t)t * v.dim4 + c_lo) * 64 + lane, 64, q_lds + (size_t)c_lo * 4, c_hi - c_lo);
        double* pb = part + (size_t)(it & 1u) * kSplitWaves * 64;
        pb[wave * 64 + lane] = (double)acc;
        __syncthreads();                                                             // one barrier per tile: the buffer written two tiles ahead is free by then
        if (wave != consumer) continue;
        double sum = pb[lane];
#pragma unroll
        for (int w = 1; w < kSplitWaves; w++) sum = sum + pb[w * 64 + lane];
        double b;
        if constexpr (M == QV_COSINE || M == QV_DOT) b = k_u * qn_s * rn; else b = k_u * sum;
        if constexpr (M == QV_COSINE) b = b + (__builtin_fabs(sum) + b) * (2.0 * k_u);    // the norm's own uncertainty (<= k_u / 2 + 3 u relative), twice over
        const float d_lo = finalize<M>((A)(sum - b), qc, rn), d_hi = finalize<M>((A)(sum + b), qc, rn);
        float dist = d_lo;
        const bool live = (am >> lane) & 1ull;
        const bool ok = __float_as_uint(d_lo) == __float_as_uint(d_hi) && d_lo == d_lo;
        if (__ballot(live && !ok)) {
            // the reference's own arithmetic for this tile: one chain per row (and for the query's norm, once per workgroup)
            A qn2 = 0;
            A ex;
            if (!have_exact) { ex = row_accumulate<M, 16, true, true, false>(tiles + (size_t)t * v.dim4 * 64 + lane, 64, q_lds, v.dim4, &qn2); qc_exact = qconst_from_norm2<M>(qn2); have_exact = true; }
            else ex = row_accumulate<M, 16, false, true, false>(tiles + (size_t)t * v.dim4 * 64 + lane, 64, q_lds, v.dim4);
            const float de = finalize<M>(ex, qc_exact, rn);
            if (!ok) dist = de;
        }
        const uint64_t key = live ? make_key(dist, row) : kDeadKey;
        if (first) { list = wave_sort64(key, lane); thr = readlane64(list, kth); first = false; }
        else list_insert(list, thr, key, kth, lane);
    }
    // the waves' lists -> wave 0 -> published; the last workgroup to finish merges (k_flat_scan<., ., true>'s protocol)
    wl[wave * 64 + lane] = list;
    __syncthreads();
    if (wave == 0) {
        if (first) { list = kDeadKey; thr = kDeadKey; }
        for (uint32_t w = 1; w < kSplitWaves; w++) {
            const uint64_t key = lane < k ? wl[w * 64 + lane] : kDeadKey;
            if (first) { list = wave_sort64(key, lane); thr = readlane64(list, kth); first = false; }
            else list_insert(list, thr, key, kth, lane);
        }
        uint64_t* mine = partial + ((size_t)qi * gridDim.x + blockIdx.x) * k;
        if (lane < k) (void)atomicExch(reinterpret_cast<unsigned long long*>(&mine[lane]), (unsigned long long)list);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t last = 0;
        if (lane == 0) last = atomicAdd(&tickets[qi], 1u) == gridDim.x - 1 ? 1u : 0u;
        last = __builtin_amdgcn_readfirstlane(last);
        if (last && lane == 0) tickets[qi] = 0;
        if (lane == 0) s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    merge_lists_last_workgroup(partial + (size_t)qi * gridDim.x * k, gridDim.x, k, rows_out + (size_t)qi * k, dist_out + (size_t)qi * k, done_flag, done_seq);
}

// The same for the few queries of a shared pass (concurrent callers on a short corpus: qv_coalesce.h): up to QB queries per workgroup row
// (blockIdx.y = group of QB).  k_flat_scan_mq gives a wave whole tiles and QB chains of dim steps per row — 16 queries over 10 k x 768
// took 280 us, all of it chains.  Here a wave converts each element of its columns once and feeds QB partial chains (768 fma + 96
// converts per tile and wave at 768 dimensions and QB = 8: ~3 us, what the tile's bytes take to arrive); wave j is query j's consumer
// on every tile, so its list is the workgroup's list for that query: no merge inside the workgroup.  Lists go to partial[q][grid][k];
// k_merge_lists follows as for the other multi-query scans.
template <int M, int QB>
__global__ void __launch_bounds__(kSplitBlock)
k_flat_scan_split_mq(IndexView v, const float* __restrict__ queries, uint32_t nq, uint32_t k, uint64_t* __restrict__ partial) {
    static_assert(QB <= kSplitWaves, "one consumer wave per query");
    using Q = typename MT<M>::Q;
    using A = typename MT<M>::A;
    extern __shared__ __align__(16) unsigned char smem[];
    const size_t q_stride = (size_t)v.dim4 * 4;                                    // elements per staged query
    Q* q_lds = reinterpret_cast<Q*>(smem);                                         // [QB][dim4 * 4]
    const size_t q_bytes = ((size_t)QB * q_stride * sizeof(Q) + 15) / 16 * 16;
    double* part = reinterpret_cast<double*>(smem + q_bytes);                      // [QB][kSplitWaves][64]
    __shared__ double s_red[QB][kSplitWaves];
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t q0 = blockIdx.y * QB;
    const uint32_t nqg = nq - q0 < (uint32_t)QB ? nq - q0 : (uint32_t)QB;          // queries of this group
    for (uint32_t j = 0; j < (uint32_t)QB; j++) {
        const float* q = queries + (size_t)(q0 + (j < nqg ? j : 0)) * v.dim;       // (slots past the group repeat its first query: computed, never written)
        for (uint32_t i = threadIdx.x; i < q_stride; i += kSplitBlock) q_lds[j * q_stride + i] = i < v.dim ? (Q)q[i] : (Q)0;
    }
    __syncthreads();
    double qn_s = 0.0;                                                             // of THIS wave's query (wave j consumes query j)
    if constexpr (M == QV_COSINE || M == QV_DOT) {
        for (uint32_t j = 0; j < (uint32_t)QB; j++) {
            double sq = 0.0;
            for (uint32_t i = threadIdx.x; i < q_stride; i += kSplitBlock) { const double a = (double)q_lds[j * q_stride + i]; sq = __builtin_fma(a, a, sq); }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sq = sq + __shfl_xor(sq, off);
            if (lane == 0) s_red[j][wave] = sq;
        }
        __syncthreads();
        if (wave < (uint32_t)QB) {
            double sq = s_red[wave][0];
#pragma unroll
            for (int w = 1; w < kSplitWaves; w++) sq = sq + s_red[wave][w];
            qn_s = __builtin_sqrt(sq);
        }
    }
    const double k_u = ((double)(2u * v.dim) + 128.0) * 0x1p-53;
    QConst qc; qc.qn = qn_s; qc.qn32 = 0.0f;
    QConst qc_exact; qc_exact.qn = 0.0; qc_exact.qn32 = 0.0f; bool have_exact = false;
    const uint32_t kth = k - 1;
    uint64_t list = kDeadKey, thr = kDeadKey;
    bool first = true;
    const f4* tiles = reinterpret_cast<const f4*>(v.tiles);
    const uint32_t c_lo = wave * v.dim4 / kSplitWaves, c_hi = (wave + 1) * v.dim4 / kSplitWaves;
    const bool consumer = wave < nqg;
    const Q* my_q = q_lds + (size_t)(wave < (uint32_t)QB ? wave : 0) * q_stride;
    for (uint32_t t = blockIdx.x; t < v.n_tiles; t += gridDim.x) {
        const uint32_t row = t * 64 + lane;
        double rn = 0.0; uint64_t am = 0;
        if (consumer) {
            if constexpr (MT<M>::needs_rnorm || M == QV_DOT) rn = v.rnorm[row];
            am = v.alive[t];
        }
        A acc[QB];
#pragma unroll
        for (int j = 0; j < QB; j++) acc[j] = 0;
        const f4* p = tiles + ((size_t)t * v.dim4 + c_lo) * 64 + lane;
        for (uint32_t c = c_lo; c < c_hi; c += 8) {                                  // eight chunks requested together, each element widened once for all queries
            f4 x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) x[u] = c + (uint32_t)u < c_hi ? p[(size_t)(c - c_lo + u) * 64] : f4{0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (c + (uint32_t)u < c_hi) {
#pragma unroll
                    for (int j = 0; j < QB; j++) {
                        const Q* qq = q_lds + (size_t)j * q_stride + (size_t)(c + u) * 4;
                        acc1<M>(acc[j], qq[0], x[u].x); acc1<M>(acc[j], qq[1], x[u].y); acc1<M>(acc[j], qq[2], x[u].z); acc1<M>(acc[j], qq[3], x[u].w);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < QB; j++) part[((size_t)j * kSplitWaves + wave) * 64 + lane] = (double)acc[j];
        __syncthreads();
        if (consumer) {
            const double* pb = part + (size_t)wave * kSplitWaves * 64;
            double sum = pb[lane];
#pragma unroll
            for (int w = 1; w < kSplitWaves; w++) sum = sum + pb[w * 64 + lane];
            double b;
            if constexpr (M == QV_COSINE || M == QV_DOT) b = k_u * qn_s * rn; else b = k_u * sum;
            if constexpr (M == QV_COSINE) b = b + (__builtin_fabs(sum) + b) * (2.0 * k_u);
            const float d_lo = finalize<M>((A)(sum - b), qc, rn), d_hi = finalize<M>((A)(sum + b), qc, rn);
            float dist = d_lo;
            const bool live = (am >> lane) & 1ull;
            const bool ok = __float_as_uint(d_lo) == __float_as_uint(d_hi) && d_lo == d_lo;
            if (__ballot(live && !ok)) {                                             // the reference's own chain for this tile and query
                A qn2 = 0; A ex;
                if (!have_exact) { ex = row_accumulate<M, 16, true, true, false>(tiles + (size_t)t * v.dim4 * 64 + lane, 64, my_q, v.dim4, &qn2); qc_exact = qconst_from_norm2<M>(qn2); have_exact = true; }
                else ex = row_accumulate<M, 16, false, true, false>(tiles + (size_t)t * v.dim4 * 64 + lane, 64, my_q, v.dim4);
                const float de = finalize<M>(ex, qc_exact, rn);
                if (!ok) dist = de;
            }
            const uint64_t key = live ? make_key(dist, row) : kDeadKey;
            if (first) { list = wave_sort64(key, lane); thr = readlane64(list, kth); first = false; }
            else list_insert(list, thr, key, kth, lane);
        }
        __syncthreads();                                                             // the partial sums are read before the next tile's are written
    }
    if (consumer && lane < k) partial[((size_t)(q0 + wave) * gridDim.x + blockIdx.x) * k + lane] = list;
}


// ---------------------------------------------------------------- queries a filter handed back, redone WITHOUT the host --
// The matrix-core filter hands a query back when its candidate buffer overflows (flags[q] != 0: a loose sample bound, e.g. a corpus
// stored cluster by cluster); such a query needs the exact scan.  Until round 5 the callers read the flags on the host — one round trip
// per batch (and per shard: qv_sharded_search_device was synchronous for 9+ queries because of it).  Here: k_redo_compact lists the
// flagged queries and zeroes their tickets; k_flat_scan_redo — a FIXED launch of grid workgroups, each of which leaves at once when the
// list is empty — walks the listed queries one after another: per query the flat scan's own loop over the workgroup's tiles, the
// lists published with returning atomic exchanges, a ticket, and the LAST workgroup merges and writes the query's k results in place
// of what the filter left there (k_flat_scan<., ., true>'s protocol, once per listed query).  A launch serves kRedoSlots list entries
// from `first` on (its lists live in [slot][grid][k] of the workspace); the launcher issues ceil(nq / kRedoSlots) of them, all but the
// needed ones empty: nothing is decided on the host.
constexpr uint32_t kRedoSlots = 64;
__global__ void k_redo_compact(const uint32_t* __restrict__ flags, uint32_t nq, uint32_t* __restrict__ list, uint32_t* __restrict__ count /* [0] count, [1 ..] tickets [nq] */) {
    // one workgroup, in query order (a wave at a time: ballot + prefix popcount)
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nq; base += blockDim.x) {
        const uint32_t q = base + threadIdx.x;
        const bool f = q < nq && flags[q] != 0;
        const uint64_t m = __ballot(f);
        const uint32_t lane = lane_id();
        __shared__ uint32_t w_off[16];
        const uint32_t wave = threadIdx.x >> 6;
        if (lane == 0) w_off[wave] = (uint32_t)__builtin_popcountll(m);
        __syncthreads();
        uint32_t before = s_n;
        for (uint32_t w = 0; w < wave; w++) before += w_off[w];
        if (f) list[before + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = q;
        if (q < nq) count[1 + q] = 0;                                  // tickets
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t t = 0; for (uint32_t w = 0; w < (blockDim.x >> 6); w++) t += w_off[w]; s_n += t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) count[0] = s_n;
}

template <int M, int U>
__global__ void __launch_bounds__(kScanBlock)
k_flat_scan_redo(IndexView v, const float* __restrict__ queries, uint32_t k, uint32_t k_stride, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                 uint32_t first, uint64_t* __restrict__ partial /* [kRedoSlots][grid][k] */, uint32_t* __restrict__ tickets /* [nq] by list position */,
                 uint32_t* __restrict__ rows_out, float* __restrict__ dist_out) {
    using Q = typename MT<M>::Q;
    const uint32_t n = count[0];
    if (first >= n) return;                                           // (the usual case: nothing was handed back)
    extern __shared__ __align__(16) unsigned char smem[];
    Q* q_lds = reinterpret_cast<Q*>(smem);
    uint64_t* wl = reinterpret_cast<uint64_t*>(smem + (((size_t)v.dim4 * 4 * sizeof(Q)) + 15) / 16 * 16);  // [kScanWaves][64]
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t tw = gridDim.x * kScanWaves, kth = k - 1;
    const f4* tiles = reinterpret_cast<const f4*>(v.tiles);
    const uint32_t last_j = n < first + kRedoSlots ? n : first + kRedoSlots;
    for (uint32_t j = first; j < last_j; j++) {
        const uint32_t qi = list[j];
        __syncthreads();                                               // (q_lds and wl are reused from the previous entry)
        stage_query<M>(q_lds, queries + (size_t)qi * v.dim, v.dim, v.dim4);
        __syncthreads();
        uint64_t lst = kDeadKey, thr = kDeadKey;
        QConst qc; qc.qn = 0.0; qc.qn32 = 0.0f;
        bool first_tile = true;
        for (uint32_t t = blockIdx.x * kScanWaves + wave; t < v.n_tiles; t += tw) {
            typename MT<M>::A qn2 = 0, acc;
            if (first_tile) { acc = row_accumulate<M, U, true>(tiles + (size_t)t * v.dim4 * 64 + lane, 64, q_lds, v.dim4, &qn2); qc = qconst_from_norm2<M>(qn2); }
            else acc = row_accumulate<M, U, false>(tiles + (size_t)t * v.dim4 * 64 + lane, 64, q_lds, v.dim4);
            const uint32_t row = t * 64 + lane;
            double rn = 0.0;
            if constexpr (MT<M>::needs_rnorm) rn = v.rnorm[row];
            const float dist = finalize<M>(acc, qc, rn);
            const uint64_t am = v.alive[t];
            const uint64_t key = ((am >> lane) & 1ull) ? make_key(dist, row) : kDeadKey;
            if (first_tile) { lst = wave_sort64(key, lane); thr = readlane64(lst, kth); first_tile = false; }
            else list_insert(lst, thr, key, kth, lane);
        }
        wl[wave * 64 + lane] = lst;
        __syncthreads();
        uint64_t* slot = partial + (size_t)(j - first) * gridDim.x * k;
        if (wave == 0) {
            for (uint32_t w = 1; w < kScanWaves; w++) list_insert(lst, thr, lane < k ? wl[w * 64 + lane] : kDeadKey, kth, lane);
            uint64_t* mine = slot + (size_t)blockIdx.x * k;
            if (lane < k) (void)atomicExch(reinterpret_cast<unsigned long long*>(&mine[lane]), (unsigned long long)lst);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every lane's exchange has returned: the list is at the memory side
            uint32_t last = 0;
            if (lane == 0) last = atomicAdd(&tickets[j], 1u) == gridDim.x - 1 ? 1u : 0u;
            last = __builtin_amdgcn_readfirstlane(last);
            if (lane == 0) wl[0] = last;
        }
        __syncthreads();
        if ((uint32_t)wl[0]) {
            __syncthreads();
            merge_lists_last_workgroup(slot, gridDim.x, k, rows_out + (size_t)qi * k_stride, dist_out + (size_t)qi * k_stride);
        }
    }
}

__device__ __forceinline__ uint64_t wave_min64(uint64_t x) {
#pragma unroll
    for (int off = 32; off; off >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)x, off), hi = __shfl_xor((uint32_t)(x >> 32), off);
        uint64_t y = ((uint64_t)hi << 32) | lo;
        x = y < x ? y : x;
    }
    return x;
}

// ---------------------------------------------------------------- small collections: scan + merge in ONE launch --
// configs[0] of BASELINE.json is 10k x 128: 5 MB of rows, a scan of a few microseconds — what a query costs there is launches and
// the wait for the stream (31 us per query in round 3: scan launch + merge launch + hipStreamSynchronize; the reference's own
// ExactIndex.Search bench at 1000 x 64 is 38 us on a laptop core, final_bench.txt:28).  Up to kSmallTiles tiles the scan's
// workgroups hand their lists to the LAST one to finish (a ticket), which merges them and writes the final rows / distances
// itself — and, for the host-pointer entry point, a sequence number into pinned memory that the host polls instead of waiting
// for the stream.  Same arithmetic and (distance, row) order as k_flat_scan + k_merge_lists.
// Nothing crosses workgroups through a fence (an agent-scope fence writes the XCD's L2 back, qv_select.h): the lists are published
// with returning atomic exchanges (returned = performed at the memory side), then the ticket; the last workgroup reads them with
// agent-scope atomic loads.  grid (workgroups, nq); partial [nq][grid][k]; tickets [nq], zero before the first launch and left zero.
constexpr uint32_t kSmallTiles = 256;                  // 16k rows: 64 workgroups of one tile per wave (at 30k rows the two-launch path measured faster: 25.7 against 32.6 us)
constexpr uint32_t kSmallPerThread = 64 * 16 / kScanBlock;    // keys a thread of the last workgroup holds: 64 lists of up to 16 keys over 256 threads
constexpr uint32_t kSmallSurv = 512;                   // survivors of the bound ranked by counting (more: one wave inserts the lists)
template <int M, int U>
__global__ void __launch_bounds__(kScanBlock)
k_flat_scan_small(IndexView v, const float* __restrict__ queries, uint32_t k, uint64_t* __restrict__ partial, uint32_t* __restrict__ tickets,
                  uint32_t* __restrict__ rows_out, float* __restrict__ dist_out, uint32_t* done_flag, uint32_t done_seq) {
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
    bool first = true;
    for (uint32_t t = blockIdx.x * kScanWaves + wave; t < v.n_tiles; t += tw) {
        typename MT<M>::A qn2 = 0, acc;
        // the row's norm and the tile's live word first, then all of a 128-dimension row's chunks in ONE round of requests (U = 32,
        // pinned), temporal: a collection this small stays in its XCD's L2 from call to call (the same workgroup, hence the same XCD,
        // reads the same tiles every time) — what a query costs here is round trips, not bytes
        const uint32_t row = t * 64 + lane;
        double rn = 0.0;
        if constexpr (MT<M>::needs_rnorm) rn = v.rnorm[row];
        const uint64_t am = v.alive[t];
        if (first) { acc = row_accumulate<M, U, true, true, false>(tiles + (size_t)t * v.dim4 * 64 + lane, 64, q_lds, v.dim4, &qn2); qc = qconst_from_norm2<M>(qn2); }
        else acc = row_accumulate<M, U, false, true, false>(tiles + (size_t)t * v.dim4 * 64 + lane, 64, q_lds, v.dim4);
        const float dist = finalize<M>(acc, qc, rn);
        const uint64_t key = ((am >> lane) & 1ull) ? make_key(dist, row) : kDeadKey;
        if (first) { list = wave_sort64(key, lane); thr = readlane64(list, kth); first = false; }
        else list_insert(list, thr, key, kth, lane);
    }
    wl[wave * 64 + lane] = list;
    __syncthreads();
    __shared__ uint32_t s_last, s_ns;
    __shared__ unsigned long long s_bound;
    __shared__ uint64_t surv[kSmallSurv];
    uint64_t* mine = partial + ((size_t)qi * gridDim.x + blockIdx.x) * k;
    if (wave == 0) {
        for (uint32_t w = 1; w < kScanWaves; w++) list_insert(list, thr, lane < k ? wl[w * 64 + lane] : kDeadKey, kth, lane);
        uint32_t last = 1;
        if (gridDim.x > 1) {
            if (lane < k) (void)atomicExch(reinterpret_cast<unsigned long long*>(&mine[lane]), (unsigned long long)list);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every lane's exchange has returned: the list is at the memory side
            if (lane == 0) last = atomicAdd(&tickets[qi], 1u) == gridDim.x - 1 ? 1u : 0u;
            last = __builtin_amdgcn_readfirstlane(last);
            if (last && lane == 0) tickets[qi] = 0;                    // for the next launch on this workspace (stream order)
        }
        if (lane == 0) { s_last = last; s_ns = 0; s_bound = kDeadKey; }
    }
    __syncthreads();
    if (!s_last) return;
    if (gridDim.x > 1) {
        // The last workgroup merges, all four waves: (1) B = the smallest k-th key of any list — that list alone holds k keys <= B, so
        // nothing above B is in the answer; (2) the keys <= B (a few dozen of grid * k) go to LDS; (3) every survivor counts the
        // survivors below it: that count is its place.  (One wave inserting 400 keys one by one was 8 of the kernel's 20 us.)
        const uint64_t* all = partial + (size_t)qi * gridDim.x * k;
        const uint32_t total = gridDim.x * k;
        uint64_t b = kDeadKey;
        for (uint32_t w = threadIdx.x; w < gridDim.x; w += kScanBlock) { const uint64_t x = __hip_atomic_load(&all[(size_t)w * k + kth], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); b = x < b ? x : b; }
        uint64_t keys[kSmallPerThread];
#pragma unroll
        for (uint32_t u = 0; u < kSmallPerThread; u++) {
            const uint32_t i = threadIdx.x + u * kScanBlock;
            keys[u] = i < total ? __hip_atomic_load(&all[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kDeadKey;
        }
        b = wave_min64(b);
        if (lane == 0 && b != kDeadKey) atomicMin(&s_bound, (unsigned long long)b);
        __syncthreads();
        const uint64_t bound = s_bound;
#pragma unroll
        for (uint32_t u = 0; u < kSmallPerThread; u++)
            if (keys[u] != kDeadKey && keys[u] <= bound) { const uint32_t pos = atomicAdd(&s_ns, 1u); if (pos < kSmallSurv) surv[pos] = keys[u]; }
        __syncthreads();
        const uint32_t ns = s_ns < kSmallSurv ? s_ns : kSmallSurv;     // (more than kSmallSurv keys <= B needs > 32 lists with k-th keys tied at B: the first kSmallSurv still hold the answer's
                                                                      //  keys only if none is lost — so that case takes the serial path below)
        if (s_ns <= kSmallSurv) {
            list = kDeadKey;
            for (uint32_t i = threadIdx.x; i < ns; i += kScanBlock) {
                const uint64_t me = surv[i];
                uint32_t rank = 0;
                for (uint32_t j = 0; j < ns; j++) rank += surv[j] < me ? 1u : 0u;
                if (rank < k) wl[rank] = me;                           // keys are distinct: ranks are too
            }
            __syncthreads();
            if (wave != 0) return;
            list = lane < (ns < k ? ns : k) ? wl[lane] : kDeadKey;
        } else {
            if (wave != 0) return;
            for (uint32_t base = 0; base < total; base += 64) {
                const uint32_t i = base + lane;
                uint64_t key = kDeadKey;
                if (i < total && i / k != blockIdx.x) key = __hip_atomic_load(&all[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                list_insert(list, thr, key, kth, lane);
            }
        }
    } else if (wave != 0) return;
    const bool dead = list == kDeadKey;
    const uint32_t r_out = dead ? 0xFFFFFFFFu : (uint32_t)list;
    const float d_out = dead ? __uint_as_float(0x7F800000u) : unord_f32((uint32_t)(list >> 32));
    if (!done_flag) {
        if (lane < k) { rows_out[(size_t)qi * k + lane] = r_out; dist_out[(size_t)qi * k + lane] = d_out; }
    } else {
        // The host polls a sequence number instead of waiting for the stream (nq == 1 for this use).  The results go out as
        // SYSTEM-scope stores (write-through whatever the page's cache policy: plain stores sat in L2 until the kernel ended and
        // the host read the previous call's rows); once they are acknowledged the sequence number follows on the same path.
        if (lane < k) {
            __hip_atomic_store(&rows_out[(size_t)qi * k + lane], r_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&dist_out[(size_t)qi * k + lane], d_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------- flat scan, 64 < k <= 128 --
// The negative-example branches of the reference fetch max(2k, 30) results (hybrid_index.go:516-522, hnsw/adapter.go:353-359):
// k = 50 is a 100-key search.  Same stream, same arithmetic as k_flat_scan; the wave's list holds R keys per lane
// (wide_insert), and every WAVE writes its 64 R slots — a workgroup merge through one wave's serial inserts would cost more
// than the scan of a short corpus.  (Four keys per lane, k <= 256, measured SLOWER than a key per row + selection: 5.05 against 4.80 ms
// at 10M x 768 — ~1000 inserts per wave, each a serial ~100-instruction sequence on a wave that has one partner on its SIMD.)
// The radix selection (qv_select.hip) then takes the k best of the waves' lists: all keys a
// wave saw beyond its 64 R best are beaten by 64 R >= k others, so the lists hold the answer.
// Output: lists[((q * gridDim.x + blockIdx.x) * kScanWaves + wave) * 64 R + r * 64 + lane].
template <int M, int U, int R>
__global__ void __launch_bounds__(kScanBlock)
k_flat_scan_wide(IndexView v, const float* __restrict__ queries, uint32_t k, uint64_t* __restrict__ lists) {
    using Q = typename MT<M>::Q;
    extern __shared__ __align__(16) unsigned char smem[];
    Q* q_lds = reinterpret_cast<Q*>(smem);
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t qi = blockIdx.y;
    stage_query<M>(q_lds, queries + (size_t)qi * v.dim, v.dim, v.dim4);
    __syncthreads();

    const uint32_t tw = gridDim.x * kScanWaves;
    const uint32_t kth = k - 1;
    uint64_t list[R], thr = kDeadKey;
#pragma unroll
    for (int r = 0; r < R; r++) list[r] = kDeadKey;
    const f4* tiles = reinterpret_cast<const f4*>(v.tiles);
    QConst qc; qc.qn = 0.0; qc.qn32 = 0.0f;

    auto finish_tile = [&](uint32_t t, typename MT<M>::A acc, bool first) {
        const uint32_t row = t * 64 + lane;
        double rn = 0.0;
        if constexpr (MT<M>::needs_rnorm) rn = v.rnorm[row];
        float dist = finalize<M>(acc, qc, rn);
        uint64_t am = v.alive[t];                                     // wave-uniform
        uint64_t key = ((am >> lane) & 1ull) ? make_key(dist, row) : kDeadKey;
        if (first) list[0] = wave_sort64(key, lane);                  // empty list: sort the tile outright (k > 64: the threshold stays open)
        else wide_insert<R>(list, thr, key, kth, lane);
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
    uint64_t* out = lists + (((size_t)qi * gridDim.x + blockIdx.x) * kScanWaves + wave) * (64 * R);
#pragma unroll
    for (int r = 0; r < R; r++) out[r * 64 + lane] = list[r];
}

// ---------------------------------------------------------------- multi-query scan --
// QB queries share ONE pass over the corpus (HybridIndex.BatchSearch, hybrid_index.go:677-811,
// is Q independent exact searches; here every 16-byte row chunk a lane loads is used for QB
// dot products).  Same arithmetic contract: lane == row, each (row, query) distance is one
// sequential chain over dims 0..D-1.  The query block sits in LDS interleaved by query
// (q_lds[dim][QB]) so one ds_read_b128 feeds two (f64) or four (f32) queries of one dim.
// grid = (workgroups, ceil(nq/QB)); partial layout identical to k_flat_scan.
// (mq_tile — one tile for QB queries — lives in qv_kernels.h: the traversal's hub table uses it too)
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

// The multi-query scan writing ONE KEY PER ROW AND QUERY instead of keeping lists: the front end of the radix selection
// (qv_select.hip) for batches that ask for more than 64 results per query and do not go through the matrix-core filter (2 - 8
// queries, or a metric the filter does not take).  QB queries share a corpus pass, their values arrive as scalar operands
// (k_flat_scan_mq's SQ form); keys_all [nq][n_tiles * 64].  grid = (workgroups, ceil(nq / QB)).
template <int M, int U, int QB>
__global__ void __launch_bounds__(kScanBlock, 2)
k_flat_keys_mq(IndexView v, const typename MT<M>::Q* __restrict__ qblk, uint32_t nq, uint64_t* __restrict__ keys_all,
               const uint32_t* __restrict__ active /* null, or a device word: query groups from *active on leave at once (launch_flat_select_redo) */) {
    using Q = typename MT<M>::Q;
    using A = typename MT<M>::A;
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t q0 = blockIdx.y * QB;
    if (active != nullptr && q0 >= *active) return;
    const Q* q_lds = qblk + (size_t)blockIdx.y * v.dim4 * 4 * QB;                // global, uniform -> scalar loads
    const uint32_t tw = gridDim.x * kScanWaves;
    const size_t n = (size_t)v.n_tiles * 64;
    QConst qc[QB];
#pragma unroll
    for (int j = 0; j < QB; j++) { qc[j].qn = 0.0; qc[j].qn32 = 0.0f; }
    const f4* tiles = reinterpret_cast<const f4*>(v.tiles);
    bool first = true;
    for (uint32_t t = blockIdx.x * kScanWaves + wave; t < v.n_tiles; t += tw) {
        A acc[QB], qa[QB];
        if (first) {
            mq_tile<M, U, QB, true>(tiles + (size_t)t * v.dim4 * 64 + lane, q_lds, v.dim4, acc, qa);
#pragma unroll
            for (int j = 0; j < QB; j++) qc[j] = qconst_from_norm2<M>(qa[j]);
            first = false;
        } else mq_tile<M, U, QB, false>(tiles + (size_t)t * v.dim4 * 64 + lane, q_lds, v.dim4, acc, qa);
        const uint32_t row = t * 64 + lane;
        double rn = 0.0;
        if constexpr (MT<M>::needs_rnorm) rn = v.rnorm[row];
        const bool live = (v.alive[t] >> lane) & 1ull;
#pragma unroll
        for (int j = 0; j < QB; j++) {
            const uint64_t key = live ? make_key(finalize<M>(acc[j], qc[j], rn), row) : kDeadKey;
            if (q0 + j < nq) __builtin_nontemporal_store(key, &keys_all[(size_t)(q0 + j) * n + row]);
        }
    }
}

size_t flat_keys_mq_workspace_bytes(uint32_t nq, uint32_t dim4) { return ((size_t)(nq + 8) * dim4 * 4 * sizeof(double) + 255) / 256 * 256; }
// keys of nq >= 2 queries in shared corpus passes; d_qws: flat_keys_mq_workspace_bytes
hipError_t launch_flat_keys_mq(const IndexView& v, const ScanPlan& p, const float* d_queries, uint32_t nq, uint64_t* d_keys, void* d_qws, hipStream_t s, const uint32_t* d_active) {
    const uint32_t want = (v.n_tiles + kScanWaves - 1) / kScanWaves;
    const uint32_t grid = std::max(1u, std::min(want, p.grid));
#define QV_KMQ(MMM, QQ)                                                                                                   \
    {                                                                                                                     \
        using QT = typename MT<MMM>::Q;                                                                                   \
        const uint32_t groups = (nq + QQ - 1) / QQ;                                                                       \
        const uint32_t per = v.dim4 * 4 * QQ;                                                                             \
        hipLaunchKernelGGL((k_prep_qblk<MMM, QQ>), dim3((per + 255) / 256, groups), dim3(256), 0, s, d_queries, nq, v.dim, v.dim4, static_cast<QT*>(d_qws)); \
        hipLaunchKernelGGL((k_flat_keys_mq<MMM, 4, QQ>), dim3(grid, groups), dim3(p.block), 0, s, v, static_cast<const QT*>(d_qws), nq, d_keys, d_active); \
    }
    if (nq >= 5) { QV_DISPATCH_METRIC(v.metric, { QV_KMQ(MM, 8) }); }
    else { QV_DISPATCH_METRIC(v.metric, { QV_KMQ(MM, 4) }); }
#undef QV_KMQ
    return hipGetLastError();
}

// One workgroup per query merges n_lists sorted lists of k keys into the final top-k.
// Bound trick: the smallest k-th entry over all lists, B, is an upper bound of the final
// k-th key (that list alone holds k keys <= B), so only keys <= B can be in the answer.
// Typically a few dozen of the n_lists*k keys survive; one wave insertion-sorts them.
constexpr int kMergeBlock = 1024;
constexpr int kMergeCap = 2048;                       // survivors kept in LDS; more -> general path
constexpr int kMergeHeads = 128;                      // sampled list heads ranked in LDS


// The merge itself, for one query: src = its n_lists lists of k keys, rows_out / dist_out = its k results.  AT: the lists were
// published by other workgroups of the SAME launch (returning atomic exchanges, then a ticket: k_flat_scan<., ., true>) and are read
// with agent-scope atomic loads; otherwise by an earlier launch, and plain loads do.
template <bool AT>
__device__ void merge_lists_body(const uint64_t* __restrict__ src, uint32_t n_lists, uint32_t k, uint32_t* __restrict__ rows_out, float* __restrict__ dist_out,
                                 uint32_t* done_flag = nullptr, uint32_t done_seq = 0) {
    auto ld = [](const uint64_t* p) -> uint64_t {
        if constexpr (AT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return *p;
    };

    __shared__ uint64_t wl[kMergeBlock / 64][64];
    __shared__ uint64_t surv[kMergeCap];
    __shared__ uint32_t hd[kMergeHeads], hlt[kMergeHeads], hle[kMergeHeads];
    __shared__ uint64_t s_bound, s_b1;
    __shared__ uint32_t s_nsurv;
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nw = blockDim.x >> 6;
    const uint32_t total = n_lists * k;
    const uint32_t kth = k - 1;

    // every global load of the common case is issued up front (one HBM/L2 latency, not three):
    // this thread's <= 8 keys, one list's k-th key, one sampled list head
    const bool small = total <= blockDim.x * 8;
    uint64_t mine[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { uint32_t i = u * blockDim.x + threadIdx.x; mine[u] = (small && i < total) ? ld(src + i) : kDeadKey; }
    // sampled heads: m = min(n_lists, kMergeHeads) lists at a fixed stride.  Any k different
    // lists each hold a key <= the k-th smallest of their heads, so a subset still gives a
    // valid (slightly looser) bound, and the O(m^2) rank count stays ~0.5 us on one CU.
    const uint32_t m = n_lists < (uint32_t)kMergeHeads ? n_lists : (uint32_t)kMergeHeads;
    const uint32_t hstride = n_lists / m;
    const bool use_heads = m >= k;
    uint64_t b = kDeadKey;
    for (uint32_t w = threadIdx.x; w < n_lists; w += blockDim.x) { uint64_t x = ld(src + (size_t)w * k + kth); b = x < b ? x : b; }
    if (use_heads)
        for (uint32_t w = threadIdx.x; w < m; w += blockDim.x) { hd[w] = (uint32_t)(ld(src + (size_t)w * hstride * k) >> 32); hlt[w] = 0; hle[w] = 0; }

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
            for (int u = 0; u < 8; u++) { uint32_t i = base + u * blockDim.x + threadIdx.x; key[u] = i < total ? ld(src + i) : kDeadKey; }
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
            uint64_t key = i < total ? ld(src + i) : kDeadKey;
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
    const bool dead = list == kDeadKey;
    const uint32_t r_out = dead ? 0xFFFFFFFFu : (uint32_t)list;
    const float d_out = dead ? __uint_as_float(0x7F800000u) : unord_f32((uint32_t)(list >> 32));
    if (!done_flag) {
        if (lane < k) { rows_out[lane] = r_out; dist_out[lane] = d_out; }
    } else {
        // the host polls a sequence number instead of waiting for the stream (k_flat_scan_small's hand-over: system-scope stores for
        // the results, and once they are acknowledged the sequence number on the same path)
        if (lane < k) {
            __hip_atomic_store(&rows_out[lane], r_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&dist_out[lane], d_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}


__global__ void __launch_bounds__(kMergeBlock)
k_merge_lists(const uint64_t* __restrict__ partial, uint32_t n_lists, uint32_t k,
              uint32_t* __restrict__ rows_out, float* __restrict__ dist_out) {
    const uint32_t qi = blockIdx.x;
    merge_lists_body<false>(partial + (size_t)qi * n_lists * k, n_lists, k, rows_out + (size_t)qi * k, dist_out + (size_t)qi * k);
}
// (out of line: the scan's loop keeps its own register allocation and schedule — tests/test_isa_guard.py counts its loads in flight)
__device__ __noinline__ void merge_lists_last_workgroup(const uint64_t* src, uint32_t n_lists, uint32_t k, uint32_t* rows_out, float* dist_out, uint32_t* done_flag, uint32_t done_seq) {
    merge_lists_body<true>(src, n_lists, k, rows_out, dist_out, done_flag, done_seq);
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

// the same merge for the packed exchange buffer of the sharded scan: per shard and query [k local rows][k distance bits], global
// row = shard base + local row (so a shard's scan output goes into the all-gather as it is: one collective, no kernel in
// between).  packed = [n_lists][nq][2][k]; one workgroup per query.
__global__ void __launch_bounds__(kMergeBlock)
k_merge_shards(const uint32_t* __restrict__ packed, const uint32_t* __restrict__ bases, uint32_t n_lists, uint32_t nq, uint32_t k,
               uint32_t* __restrict__ rows_out, float* __restrict__ dist_out, uint32_t planes) {
    __shared__ uint64_t wl[kMergeBlock / 64][64];
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nw = blockDim.x >> 6;
    const uint32_t q = blockIdx.x;
    const uint32_t kth = k - 1, total = n_lists * k;
    uint64_t list = kDeadKey, thr = kDeadKey;
    for (uint32_t base = wave * 64; base < total; base += nw * 64) {
        const uint32_t i = base + lane;
        uint64_t key = kDeadKey;
        if (i < total) {
            const uint32_t g = i / k, j = i - g * k;
            // planes == 0, interleaved: per shard [nq][2][k];  planes >= 2, planar: per shard [planes][nq][k] (rows of all
            // queries, then distances of all queries, then payload planes the merge does not read — a shard's multi-query
            // scan writes both halves in place, no repacking before the all-gather)
            const uint32_t* blk = packed + (planes ? (size_t)g * planes * nq * k + (size_t)q * k : ((size_t)g * nq + q) * 2 * k);
            const uint32_t row = blk[j];
            if (row != 0xFFFFFFFFu) key = make_key(__uint_as_float(blk[(planes ? (size_t)nq * k : k) + j]), bases[g] + row);
        }
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
            rows_out[(size_t)q * k + lane] = dead ? 0xFFFFFFFFu : (uint32_t)list;
            dist_out[(size_t)q * k + lane] = dead ? __uint_as_float(0x7F800000u) : unord_f32((uint32_t)(list >> 32));
        }
    }
}

ScanPlan plan_scan(uint32_t n_tiles, int cus) {
    ScanPlan p;
    p.block = kScanBlock;
    p.cus = (uint32_t)(cus > 0 ? cus : 1);
    uint32_t want = (n_tiles + kScanWaves - 1) / kScanWaves;          // one tile per wave at most
    static const int wg_per_cu = dev_env_int("QV_SCAN_WG_PER_CU", 2);     // 2 workgroups = 8 waves per CU: measured best (profiles/r01_sweep.txt)
    uint32_t cap = (uint32_t)cus * (uint32_t)wg_per_cu;
    p.grid = want < cap ? want : cap;
    if (p.grid == 0) p.grid = 1;
    // even shares: with T tiles per wave at most, use only as many waves as leave nobody a tile short
    // (1M rows on 256 CUs: 15625 tiles over 2048 waves = 7 or 8 each -> 1954 waves x 8)
    static const int balance = dev_env_int("QV_SCAN_BALANCE", 1);
    if (balance == 1 && want > cap) {
        const uint32_t waves = p.grid * kScanWaves;
        const uint32_t per = (n_tiles + waves - 1) / waves;
        const uint32_t need = (n_tiles + per - 1) / per;
        // measured: 1M x 768: 0.4501 -> 0.4426 ms (85.5 -> 87.0 % of HBM peak); at 77 tiles per wave (10M) the spare
        // waves matter more than the last tile (89.7 -> 89.2 %), so only short shares are evened out
        if (per <= 32) p.grid = (need + kScanWaves - 1) / kScanWaves;
    }
    p.n_lists = p.grid;
    return p;
}

size_t scan_workspace_bytes(const ScanPlan& p, uint32_t nq, uint32_t k) { return ((size_t)p.n_lists * 4 * nq * k * sizeof(uint64_t) + 255) / 256 * 256; }   // x4: the multi-query scan may use up to 8 WG/CU

hipError_t launch_merge_pairs(const float* d_dist, const uint32_t* d_rows, uint32_t n_lists, uint32_t k,
                              uint32_t* d_rows_out, float* d_dist_out, hipStream_t s) {
    if (k == 0 || k > (uint32_t)kMaxFusedK || n_lists == 0) return hipErrorInvalidValue;
    uint32_t total = n_lists * k;
    uint32_t mblock = total >= 16 * 64 * 4 ? kMergeBlock : (total >= 4 * 64 ? 256 : 64);
    hipLaunchKernelGGL(k_merge_pairs, dim3(1), dim3(mblock), 0, s, d_dist, d_rows, total, k, d_rows_out, d_dist_out);
    return hipGetLastError();
}

hipError_t launch_merge_shards(const uint32_t* d_packed, const uint32_t* d_bases, uint32_t n_lists, uint32_t nq, uint32_t k,
                               uint32_t* d_rows_out, float* d_dist_out, hipStream_t s, uint32_t planes) {
    if (k == 0 || k > (uint32_t)kMaxFusedK || n_lists == 0 || nq == 0 || planes == 1) return hipErrorInvalidValue;
    uint32_t total = n_lists * k;
    uint32_t mblock = total >= 16 * 64 * 4 ? kMergeBlock : (total >= 4 * 64 ? 256 : 64);
    hipLaunchKernelGGL(k_merge_shards, dim3(nq), dim3(mblock), 0, s, d_packed, d_bases, n_lists, nq, k, d_rows_out, d_dist_out, planes);
    return hipGetLastError();
}

// (k <= 16: with 64 keys per list the bound lets hundreds through and the two-launch path's 1024-thread merge is faster — 55 against ~30 us at 10k x 128)
bool flat_small_applies(const IndexView& v, uint32_t nq, uint32_t k) { return v.n_tiles <= kSmallTiles && nq >= 1 && nq <= 4 && k >= 1 && k <= 16; }
size_t flat_small_workspace_bytes(uint32_t nq, uint32_t k) { return (size_t)nq * 64 * std::min(k, 16u) * sizeof(uint64_t); }
// d_tickets: 64 words of the caller's own (one set per stream), zero before the first launch; the kernel leaves them zero.
hipError_t launch_flat_small(const IndexView& v, const float* d_queries, uint32_t nq, uint32_t k, void* d_ws, uint32_t* d_tickets, uint32_t* d_rows_out, float* d_dist_out,
                             uint32_t* done_flag, uint32_t done_seq, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (!flat_small_applies(v, nq, k)) return hipErrorInvalidValue;
    const uint32_t grid = std::max(1u, (v.n_tiles + kScanWaves - 1) / kScanWaves);
    uint32_t* tickets = d_tickets;
    uint64_t* partial = static_cast<uint64_t*>(d_ws);
    const size_t lds = query_lds_bytes(v.metric, v.dim4) + (size_t)kScanWaves * 64 * sizeof(uint64_t);
    hipError_t e = hipSuccess;
    QV_DISPATCH_METRIC(v.metric, {
        e = set_lds(k_flat_scan_small<MM, 32>, lds);
        if (e != hipSuccess) return e;
        if (ev0) (void)hipEventRecord(ev0, s);
        hipLaunchKernelGGL((k_flat_scan_small<MM, 32>), dim3(grid, nq), dim3(kScanBlock), lds, s, v, d_queries, k, partial, tickets, d_rows_out, d_dist_out, done_flag, done_seq);
        if (ev1) (void)hipEventRecord(ev1, s);
    });
    return hipGetLastError();
}

// 64 < kk <= kMaxWideK: the scan with R keys per lane, then the selection over the waves' lists
static uint32_t wide_regs(uint32_t) { return 2u; }
size_t flat_wide_workspace_bytes(const ScanPlan& p, uint32_t nq, uint32_t kk) {
    const size_t n = (size_t)p.grid * kScanWaves * 64 * wide_regs(kk);
    return (size_t)nq * n * sizeof(uint64_t) + 256 + select_workspace_bytes(nq, kk);
}
hipError_t launch_flat_wide(const IndexView& v, const ScanPlan& p, const float* d_queries, uint32_t nq, uint32_t kk, uint32_t k_stride,
                            void* d_ws, uint32_t* d_rows_out, float* d_dist_out, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (kk <= (uint32_t)kMaxFusedK || kk > (uint32_t)kMaxWideK || nq == 0 || kk > k_stride) return hipErrorInvalidValue;
    const uint32_t R = wide_regs(kk);
    const uint32_t n = p.grid * kScanWaves * 64 * R;
    uint64_t* lists = static_cast<uint64_t*>(d_ws);
    void* sel_ws = static_cast<char*>(d_ws) + ((size_t)nq * n * sizeof(uint64_t) + 255) / 256 * 256;
    const size_t lds = query_lds_bytes(v.metric, v.dim4);
    hipError_t e = hipSuccess;
#define QV_WIDE(RR)                                                                                              \
    QV_DISPATCH_METRIC(v.metric, {                                                                                \
        e = set_lds(k_flat_scan_wide<MM, kUnroll, RR>, lds);                                                      \
        if (e != hipSuccess) return e;                                                                            \
        if (ev0) (void)hipEventRecord(ev0, s);                                                                    \
        hipLaunchKernelGGL((k_flat_scan_wide<MM, kUnroll, RR>), dim3(p.grid, nq), dim3(p.block), lds, s, v, d_queries, kk, lists); \
        if (ev1) (void)hipEventRecord(ev1, s);                                                                    \
    })
    QV_WIDE(2);
#undef QV_WIDE
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // the lists are not in row order: ties beyond the selection's capacity are settled on the row bits (ordered = false)
    return launch_select_topk(lists, n, n, nq, kk, k_stride, sel_ws, d_rows_out, d_dist_out, s, false, false);
}

// A short corpus in the tile-over-eight-waves form (k_flat_scan_split): one query, a metric whose chain can be split and certified, rows
// wide enough that a wave's own tile is the scan (it pays from 256 dimensions; at 128 the kernels are level and the single polled launch
// wins 16 k - 60 k rows; at 64 nothing), short enough
// that the fixed costs matter (beyond ~160 k rows both forms run at the memory's rate).  QV_SCAN_SPLIT=2: never.
constexpr uint32_t kSplitMaxTiles = 2560;
bool flat_split_applies(const IndexView& v, uint32_t nq, uint32_t k) {
    static const int split = dev_env_int("QV_SCAN_SPLIT", 1), split_max = dev_env_int("QV_SCAN_SPLIT_MAX_TILES", (int)kSplitMaxTiles);
    const bool split_metric = v.metric == QV_COSINE || v.metric == QV_DOT || v.metric == QV_L2 || v.metric == QV_L1 || v.metric == QV_L2SQ_F64;
    const size_t lds = query_lds_bytes(v.metric, v.dim4) + (size_t)3 * kSplitWaves * 64 * sizeof(double) + 40 * 1024;   // + the merge's own arrays
    static const int min_dim4 = dev_env_int("QV_SCAN_SPLIT_MIN_DIM4", 32);       // (128 dimensions: 16 k / 30 k rows 30.8 / 35.6 -> 26.9 / 28.9 us, the rest equal; 64 dimensions: no gain)
    return split == 1 && split_metric && nq == 1 && k >= 1 && k <= (uint32_t)kMaxFusedK && v.dim4 >= (uint32_t)min_dim4 && lds <= (size_t)160 * 1024 &&
           v.n_tiles >= 2 && v.n_tiles <= (uint32_t)split_max;
}
// ... and for the 2 .. 32 queries of a shared pass (k_flat_scan_split_mq, groups of up to 8 per workgroup row, every group reading the corpus
// again): while the groups' reads stay under ~600 MB — measured against the multi-query kernels it replaces, us per call of 8 / 16 / 32 / 64
// queries: 10 k x 768 66 / 87 / 96 / 143 against 119 / 208 / 214 / 218; 30 k x 768 81 / 111 / 162 / 269 against 119 / 212 / 215 / 241;
// 100 k x 768 131 / 181 / 195 / 202 against 169 / 181 / 190 / 197; 10 k x 1536 93 / 99 / 135 / 234 against 182 / 346 / 350 / 349.
// (QV_SCAN_SPLIT_MQ_MAX=1: never.)
bool flat_split_mq_applies(const IndexView& v, uint32_t nq, uint32_t k) {
    static const int split = dev_env_int("QV_SCAN_SPLIT", 1), split_max = dev_env_int("QV_SCAN_SPLIT_MAX_TILES", (int)kSplitMaxTiles), nq_max = dev_env_int("QV_SCAN_SPLIT_MQ_MAX", 32);
    const bool split_metric = v.metric == QV_COSINE || v.metric == QV_DOT || v.metric == QV_L2 || v.metric == QV_L1 || v.metric == QV_L2SQ_F64;
    const uint32_t qsz = (v.metric == QV_COSINE || v.metric == QV_DOT || v.metric == QV_L2SQ_F64) ? 8u : 4u;
    const uint32_t qb = nq <= 4 ? 4u : 8u;
    const size_t lds = (size_t)qb * v.dim4 * 4 * qsz + (size_t)qb * kSplitWaves * 64 * sizeof(double) + 1024;
    const uint64_t reads = (uint64_t)((nq + qb - 1) / qb) * v.n_tiles * v.dim4 * 1024ull;
    static const int min_dim4 = dev_env_int("QV_SCAN_SPLIT_MQ_MIN_DIM4", 16);   // (from 64 dimensions: 10 k x 128, 8 / 16 / 32 queries per call 36 / 40 / 47 us against 64 / 98 / 98)
    return split == 1 && split_metric && nq >= 2 && nq <= (uint32_t)nq_max && k >= 1 && k <= (uint32_t)kMaxFusedK && v.dim4 >= (uint32_t)min_dim4 && lds <= (size_t)160 * 1024 &&
           v.n_tiles >= 2 && v.n_tiles <= (uint32_t)split_max && reads <= 600ull * 1000 * 1000;
}
hipError_t launch_flat_topk(const IndexView& v, const ScanPlan& p, const float* d_queries, uint32_t nq, uint32_t k,
                            void* d_ws, uint32_t* d_rows_out, float* d_dist_out, hipStream_t s,
                            hipEvent_t ev0, hipEvent_t ev1, uint32_t* d_tickets, uint32_t* done_flag, uint32_t done_seq, bool* flag_used) {
    if (k == 0 || k > (uint32_t)kMaxFusedK || nq == 0) return hipErrorInvalidValue;
    const size_t lds = query_lds_bytes(v.metric, v.dim4) + (size_t)kScanWaves * 64 * sizeof(uint64_t);
    uint64_t* partial = static_cast<uint64_t*>(d_ws);
    hipError_t e = hipSuccess;
    static const int mq_min = dev_env_int("QV_MQ_MIN", 2);                // nq >= this: queries share a corpus pass
    if ((int)nq >= mq_min && flat_split_mq_applies(v, nq, k)) {
        // a shared pass of a few queries over a short corpus of wide rows: the tile-over-eight-waves form, up to 8 queries per workgroup row
        const uint32_t grid = std::min<uint32_t>(v.n_tiles, std::min<uint32_t>((uint32_t)p.cus, p.n_lists * 4u));
        const uint32_t qsz = (v.metric == QV_COSINE || v.metric == QV_DOT || v.metric == QV_L2SQ_F64) ? 8u : 4u;
#define QV_SMQ(MMM, QQ)                                                                                                     \
        {                                                                                                                     \
            const size_t lds_q = ((size_t)QQ * v.dim4 * 4 * qsz + 15) / 16 * 16 + (size_t)QQ * kSplitWaves * 64 * sizeof(double); \
            e = set_lds(k_flat_scan_split_mq<MMM, QQ>, lds_q);                                                                \
            if (e != hipSuccess) return e;                                                                                    \
            if (ev0) (void)hipEventRecord(ev0, s);                                                                            \
            hipLaunchKernelGGL((k_flat_scan_split_mq<MMM, QQ>), dim3(grid, (nq + QQ - 1) / QQ), dim3(kSplitBlock), lds_q, s, v, d_queries, nq, k, partial); \
            if (ev1) (void)hipEventRecord(ev1, s);                                                                            \
        }
        QV_DISPATCH_METRIC(v.metric, {
            if constexpr (ScanSplitOK<MM>::value) {
                if (nq <= 4) QV_SMQ(MM, 4) else QV_SMQ(MM, 8)
            }
        });
#undef QV_SMQ
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        const uint32_t total = grid * k;
        const uint32_t mblock = total >= 16 * 64 * 4 ? kMergeBlock : (total >= 4 * 64 ? 256 : 64);
        hipLaunchKernelGGL(k_merge_lists, dim3(nq), dim3(mblock), 0, s, partial, grid, k, d_rows_out, d_dist_out);
        return hipGetLastError();
    }
    if ((int)nq >= mq_min) {
        // QB queries per corpus pass.  Measured on MI355X, 256 x 1M x 768 cosine (profiles/r01_sweep_mq.txt):
        // QB=8 is HBM-bound (0.454 ms/pass), QB=16 is f64-VALU-bound (0.75 ms/pass, 12.0 ms per 256 queries).
        static const int mq_qb_env = dev_env_int("QV_MQ_QB", 0), mq_wg = dev_env_int("QV_MQ_WG_PER_CU", 2);
        // the all-float32 metrics spill ~330 SGPRs at 16 queries per pass (16 x 4 scalar query values per chunk on top of the
        // list state): 16 queries take 1.73 ms in one pass and 1.02 ms in two passes of 8 (1M x 768) -> 8 per pass for them
        const bool f32_acc = v.metric == QV_L2SQ || v.metric == QV_COSINE_F32 || v.metric == QV_L2_F32 || v.metric == QV_DOT_F32;
        const int qb = mq_qb_env ? (f32_acc && mq_qb_env == 16 ? 8 : mq_qb_env) : (nq >= 9 ? (f32_acc ? 8 : 16) : (nq >= 5 ? 8 : 4));
        const uint32_t want = (v.n_tiles + kScanWaves - 1) / kScanWaves;
        const uint32_t grid = std::max(1u, std::min(want, (uint32_t)mq_wg * (p.grid / 2 ? p.grid / 2 : 1)));   // p.grid = 2 WG/CU * CUs
        void* qblk = static_cast<char*>(d_ws) + scan_workspace_bytes(p, nq, k);   // tail of the workspace
        // >= 9 cosine/dot queries: the same exact arithmetic on the f64 matrix cores (qv_mq64.hip), 16 or 32 queries per pass
        static const int mq64_min = dev_env_int("QV_MQ64_MIN", 9);
        static const int trace = env_int("QV_TRACE", 0);                  // QV_TRACE=1: name the scan kernel chosen, on stderr
        // (not for short scans: its first tile per wave pays 16 out-of-line sorts; measured slower below ~4 tiles per wave)
        if ((int)nq >= mq64_min && mq_qb_env == 0 && v.n_tiles >= 16u * p.grid && mq64_blocks(v.metric, v.dim4, nq) != 0) {
            // a last pass of 8 queries or fewer costs a whole 32-query pass of the matrix kernel (1.15 ms at 1M x 768) but only an
            // HBM-bound pass of k_flat_scan_mq (0.45-0.65 ms): split it off.  Same stream, so the workspace is reused in order.
            const uint32_t rem = nq & 31u;
            if (nq > 32 && rem >= 1 && rem <= 8) {
                e = launch_flat_topk(v, p, d_queries, nq - rem, k, d_ws, d_rows_out, d_dist_out, s, ev0, ev1, nullptr, nullptr, 0, nullptr);
                if (e != hipSuccess) return e;
                return launch_flat_topk(v, p, d_queries + (size_t)(nq - rem) * v.dim, rem, k, d_ws, d_rows_out + (size_t)(nq - rem) * k,
                                        d_dist_out + (size_t)(nq - rem) * k, s, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
            }
            uint32_t g64 = 0;
            if (trace) fprintf(stderr, "qv: scan kernel = k_flat_scan_mq64 (nq=%u, tiles=%u)\n", nq, v.n_tiles);
            e = launch_flat_scan_mq64(v, (int)p.cus, d_queries, nq, k, qblk, partial, &g64, s, ev0, ev1);
            if (e != hipSuccess) return e;
            const uint32_t total64 = g64 * k;
            const uint32_t mb64 = total64 >= 16 * 64 * 4 ? kMergeBlock : (total64 >= 4 * 64 ? 256 : 64);
            hipLaunchKernelGGL(k_merge_lists, dim3(nq), dim3(mb64), 0, s, partial, g64, k, d_rows_out, d_dist_out);
            return hipGetLastError();
        }
        if (trace) fprintf(stderr, "qv: scan kernel = k_flat_scan_mq QB=%d (nq=%u, tiles=%u)\n", qb, nq, v.n_tiles);
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
        // (16 per pass never for the all-float32 metrics, see f32_acc above: their instantiations are not built)
        if (qb == 16) { QV_DISPATCH_METRIC(v.metric, { if constexpr (MM == QV_L2SQ || MM == QV_COSINE_F32 || MM == QV_L2_F32 || MM == QV_DOT_F32) return hipErrorInvalidValue; else QV_MQ_LAUNCH(MM, 16) }); }
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
#ifdef QV_VARIANTS
    static const int unroll = env_int("QV_SCAN_UNROLL", kUnroll);     // tuning knob of the measurement build (cosine only): loads in flight per wave
    if (v.metric == QV_COSINE && unroll != kUnroll) {
#define QV_SCAN_U(UU)                                                                                             \
        case UU: e = set_lds(k_flat_scan<QV_COSINE, UU>, lds); if (e != hipSuccess) return e;                    \
            if (ev0) (void)hipEventRecord(ev0, s);                                                                \
            hipLaunchKernelGGL((k_flat_scan<QV_COSINE, UU>), dim3(p.grid, nq), dim3(p.block), lds, s, v, d_queries, k, partial, (uint32_t*)nullptr, (uint32_t*)nullptr, (float*)nullptr); \
            if (ev1) (void)hipEventRecord(ev1, s); break;
        switch (unroll) { QV_SCAN_U(4) QV_SCAN_U(8) QV_SCAN_U(12) QV_SCAN_U(24) QV_SCAN_U(32) default: return hipErrorInvalidValue; }
#undef QV_SCAN_U
    } else
#endif
    {
        // one query, the caller's tickets at hand: scan + merge in ONE launch (the last workgroup merges).  QV_SCAN_FUSE=0 (read once)
        // keeps the two launches, for measurements.
        static const int fuse = dev_env_int("QV_SCAN_FUSE", 1);
        if (d_tickets && fuse && flat_split_applies(v, nq, k)) {
            static const int split_grid = dev_env_int("QV_SCAN_SPLIT_GRID", 0);
            const uint32_t grid = std::min<uint32_t>(v.n_tiles, std::min<uint32_t>(split_grid ? (uint32_t)split_grid : (uint32_t)p.cus, p.n_lists * 4u));
            const size_t lds_s = query_lds_bytes(v.metric, v.dim4) + (size_t)3 * kSplitWaves * 64 * sizeof(double);
            QV_DISPATCH_METRIC(v.metric, {
                if constexpr (ScanSplitOK<MM>::value) {
                    e = set_lds(k_flat_scan_split<MM>, lds_s);
                    if (e != hipSuccess) return e;
                    if (ev0) (void)hipEventRecord(ev0, s);
                    hipLaunchKernelGGL((k_flat_scan_split<MM>), dim3(grid, nq), dim3(kSplitBlock), lds_s, s, v, d_queries, k, partial, d_tickets, d_rows_out, d_dist_out, done_flag, done_seq);
                    if (ev1) (void)hipEventRecord(ev1, s);
                }
            });
            if (flag_used) *flag_used = done_flag != nullptr;
            return hipGetLastError();
        }
        if (d_tickets && nq == 1 && fuse && p.grid > 1) {
            QV_DISPATCH_METRIC(v.metric, {
                e = set_lds((k_flat_scan<MM, kUnroll, true>), lds);
                if (e != hipSuccess) return e;
                if (ev0) (void)hipEventRecord(ev0, s);
                hipLaunchKernelGGL((k_flat_scan<MM, kUnroll, true>), dim3(p.grid, nq), dim3(p.block), lds, s, v, d_queries, k, partial, d_tickets, d_rows_out, d_dist_out, done_flag, done_seq);
                if (ev1) (void)hipEventRecord(ev1, s);
            });
            if (flag_used) *flag_used = done_flag != nullptr;
            return hipGetLastError();
        }
    }
    QV_DISPATCH_METRIC(v.metric, {
        e = set_lds(k_flat_scan<MM, kUnroll>, lds);
        if (e != hipSuccess) return e;
        if (ev0) (void)hipEventRecord(ev0, s);
        hipLaunchKernelGGL((k_flat_scan<MM, kUnroll>), dim3(p.grid, nq), dim3(p.block), lds, s, v, d_queries, k, partial, (uint32_t*)nullptr, (uint32_t*)nullptr, (float*)nullptr);
        if (ev1) (void)hipEventRecord(ev1, s);
    });
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    uint32_t total = p.n_lists * k;
    uint32_t mblock = total >= 16 * 64 * 4 ? kMergeBlock : (total >= 4 * 64 ? 256 : 64);
    hipLaunchKernelGGL(k_merge_lists, dim3(nq), dim3(mblock), 0, s, partial, p.n_lists, k, d_rows_out, d_dist_out);
    return hipGetLastError();
}



// the exact scan for the queries whose flags are set, decided and listed on the device (see k_flat_scan_redo)
size_t redo_workspace_bytes(const ScanPlan& p, uint32_t nq, uint32_t k) {
    return ((size_t)kRedoSlots * p.grid * k * sizeof(uint64_t) + 255) / 256 * 256 + ((size_t)nq * 4 + 255) / 256 * 256 + ((size_t)(nq + 1) * 4 + 255) / 256 * 256;
}
hipError_t launch_flat_redo_flagged(const IndexView& v, const ScanPlan& p, const float* d_queries, uint32_t nq, uint32_t k, uint32_t k_stride, const uint32_t* d_flags,
                                    void* d_ws, uint32_t* d_rows_out, float* d_dist_out, hipStream_t s) {
    if (k == 0 || k > (uint32_t)kMaxFusedK || nq == 0) return hipErrorInvalidValue;
    char* w = static_cast<char*>(d_ws);
    uint64_t* partial = reinterpret_cast<uint64_t*>(w); w += ((size_t)kRedoSlots * p.grid * k * sizeof(uint64_t) + 255) / 256 * 256;
    uint32_t* list = reinterpret_cast<uint32_t*>(w); w += ((size_t)nq * 4 + 255) / 256 * 256;
    uint32_t* count = reinterpret_cast<uint32_t*>(w);
    hipLaunchKernelGGL(k_redo_compact, dim3(1), dim3(1024), 0, s, d_flags, nq, list, count);
    const size_t lds = query_lds_bytes(v.metric, v.dim4) + (size_t)kScanWaves * 64 * sizeof(uint64_t);
    hipError_t e = hipSuccess;
    for (uint32_t first = 0; first < nq; first += kRedoSlots) {
        QV_DISPATCH_METRIC(v.metric, {
            // (the flags come from the batched filter path: its four metrics)
            if constexpr (MM == QV_COSINE || MM == QV_L2 || MM == QV_L2SQ || MM == QV_DOT) {
                e = set_lds((k_flat_scan_redo<MM, kUnroll>), lds);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL((k_flat_scan_redo<MM, kUnroll>), dim3(p.grid), dim3(p.block), lds, s, v, d_queries, k, k_stride, list, count, first, partial, count + 1, d_rows_out, d_dist_out);
            } else return hipErrorInvalidValue;
        });
    }
    return hipGetLastError();
}

}  // namespace qv
