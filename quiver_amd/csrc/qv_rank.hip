// qv_rank.hip — full ranking: all keys + stable LSD radix sort
// (shared helpers, the arithmetic contract and the build flags: qv_kernels.h)
#include "qv_select.h"

namespace qv {

// ---------------------------------------------------------------- full ranking -----
// all keys: keys[row] = (ord(dist), row) or dead
// grid = (workgroups, queries): query blockIdx.y, its keys at keys + blockIdx.y * n_tiles * 64
// HIST: the kernel also counts the first window of the radix selection (qv_select.h) — a tile's 64 keys go into the workgroup's
// LDS histogram as they are made, so the selection does not read all the keys once more for it (107 us at 10M rows)
//
// The keys leave in BATCHES.  A store per tile cost 0.31 ms of a 4.8 ms pass at 10M x 768 (measured with the store compiled out:
// 4.48 ms) although it is 0.26 % of the bytes: vector-memory operations retire in issue order per wave, so the next tile's first
// rows cannot be consumed before the store is acknowledged — one such bubble per tile and wave.  A wave parks kKeyBatch tiles'
// keys in LDS and writes them back to back, non-temporally (4.70 ms), and the end-of-window step no longer fences (qv_select.h:
// an agent-scope fence behind 80 MB of dirty keys was another 0.2 ms): 4.47 ms, the pass's rate without any store.
#ifndef QV_KEY_BATCH
#define QV_KEY_BATCH 8
#endif
constexpr int kKeyBatch = QV_KEY_BATCH;
// tiles a wave parks: kKeyBatch, fewer when a very wide query (up to 128 KiB of LDS at 16384 float64 dimensions) leaves less room
static uint32_t key_batch(int metric, uint32_t dim4, bool hist) {
    const size_t room = (size_t)160 * 1024 - 1024 - query_lds_bytes(metric, dim4) - (hist ? (size_t)kSelBins * sizeof(uint32_t) : 0);
    const size_t per = (size_t)kScanWaves * 64 * sizeof(uint64_t);
    return (uint32_t)std::max<size_t>(1, std::min<size_t>(kKeyBatch, room / per));
}
template <int M, int U, bool HIST>
__global__ void __launch_bounds__(kScanBlock)
k_flat_keys(IndexView v, const float* __restrict__ queries, uint64_t* __restrict__ keys_all, SelState* __restrict__ st, uint32_t* __restrict__ hist,
            uint32_t k, uint32_t cap, uint32_t batch /* tiles parked per wave: key_batch() */) {
    using Q = typename MT<M>::Q;
    extern __shared__ __align__(16) unsigned char smem[];
    Q* q_lds = reinterpret_cast<Q*>(smem);
    uint64_t* kb_all = reinterpret_cast<uint64_t*>(smem + (((size_t)v.dim4 * 4 * sizeof(Q)) + 15) / 16 * 16);   // [kScanWaves][batch][64]
    uint32_t* h = reinterpret_cast<uint32_t*>(kb_all + (size_t)kScanWaves * batch * 64);                         // [kSelBins] (HIST)
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint64_t* kb = kb_all + (size_t)wave * batch * 64;
    const float* query = queries + (size_t)blockIdx.y * v.dim;
    uint64_t* keys = keys_all + (size_t)blockIdx.y * v.n_tiles * 64;
    stage_query<M>(q_lds, query, v.dim, v.dim4);
    if constexpr (HIST) for (uint32_t b = threadIdx.x; b < (uint32_t)kSelBins; b += blockDim.x) h[b] = 0;
    __syncthreads();
    const QConst qc = query_const<M>(q_lds, v.dim);
    const uint32_t tw = gridDim.x * kScanWaves;
    const f4* tiles = reinterpret_cast<const f4*>(v.tiles);
    uint32_t t_first = blockIdx.x * kScanWaves + wave, parked = 0;    // first tile of the batch in LDS, tiles parked
    SelRun run{0u, 0u};
    auto flush = [&]() {
        // (non-temporal: 4.47 against 4.52 ms at 10M x 768 — the keys are read back by other CUs, nothing of them is worth keeping in this L2)
        for (uint32_t j = 0; j < parked; j++) __builtin_nontemporal_store(kb[j * 64 + lane], &keys[(size_t)(t_first + j * tw) * 64 + lane]);
        t_first += parked * tw; parked = 0;
    };
    for (uint32_t t = blockIdx.x * kScanWaves + wave; t < v.n_tiles; t += tw) {
        const f4* p = tiles + (size_t)t * v.dim4 * 64 + lane;
        // The tile loop has k_flat_scan's shape on purpose: the row walk first (cosine on hipcc's own rolling window of ~8 requests,
        // the other metrics pinned per block — row_accumulate), the row's norm and the tile's live word after it.  Measured at
        // 10M x 768 by compiling the differences out one by one (round 4): pinned blocks + norm requested first 4.47 ms, this shape
        // 4.40, the same without writing any key 4.29 (= k_flat_scan).  tests/test_isa_guard.py holds the compiled loop to it.
        typename MT<M>::A acc = row_accumulate<M, U, false>(p, 64, q_lds, v.dim4);
        const uint32_t row = t * 64 + lane;
        double rn = 0.0;
        if constexpr (MT<M>::needs_rnorm) rn = v.rnorm[row];
        const uint64_t am = v.alive[t];
        float dist = finalize<M>(acc, qc, rn);
        const uint64_t key = ((am >> lane) & 1ull) ? make_key(dist, row) : kDeadKey;
        kb[parked * 64 + lane] = key;                                  // (a lane reads back only what it wrote: no barrier)
        if (++parked == batch) flush();
        if constexpr (HIST) sel_count<0>(h, run, key, true);
    }
    flush();
    if constexpr (HIST) sel_flush(h, run);
    if constexpr (HIST) sel_finish_window<0>(h, hist + (size_t)blockIdx.y * kSelBins, st + blockIdx.y, gridDim.x, k, cap);
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

// stable sort of n 64-bit keys by their upper 32 bits (4 passes); *sorted_out = the buffer holding the result (a)
size_t radix_hist_words(uint32_t n) { return (size_t)256 * ((n + kRadixTile - 1) / kRadixTile) + 512; }
hipError_t launch_radix_sort_hi32(uint64_t* a, uint64_t* b, uint32_t n, uint32_t* hist, uint64_t** sorted_out, hipStream_t s) {
    const uint32_t nblocks = (n + kRadixTile - 1) / kRadixTile;
    uint32_t* dtot = hist + (size_t)256 * nblocks;
    uint32_t* dbase = dtot + 256;
    uint64_t* in = a; uint64_t* out = b;
    for (uint32_t shift = 32; shift < 64; shift += 8) {
        hipLaunchKernelGGL(k_radix_hist, dim3(nblocks), dim3(kRadixBlock), 0, s, in, n, shift, hist);
        hipLaunchKernelGGL(k_radix_scan_digits, dim3(256), dim3(256), 0, s, hist, nblocks, dtot);
        hipLaunchKernelGGL(k_radix_scan_totals, dim3(1), dim3(256), 0, s, dtot, dbase);
        hipLaunchKernelGGL(k_radix_scatter, dim3(nblocks), dim3(kRadixBlock), 0, s, in, out, n, shift, hist, dbase);
        uint64_t* t = in; in = out; out = t;
    }
    *sorted_out = in;
    return hipGetLastError();
}

// ---------------------------------------------------------------- merge of ranked shard lists (any k) -----
// keys of query q from the gathered per-shard lists [G][planes][nq][kcap] (plane 0: local rows, 0xFFFFFFFF = padding; plane 1:
// distance bits).  Key i = entry i % kcap of shard i / kcap: every shard's list is ascending in (distance, local row) and the
// shards' bases ascend, so equal distances already appear in global-row order and the stable sort on the distance bits
// alone leaves the full (distance, global row) order a single index produces.
__global__ void __launch_bounds__(256)
k_shard_keys(const uint32_t* __restrict__ packed, const uint32_t* __restrict__ bases, uint32_t n_lists, uint32_t nq, uint32_t q, uint32_t kcap,
             uint32_t planes, uint64_t* __restrict__ keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)n_lists * kcap) return;
    const uint32_t g = (uint32_t)(i / kcap), j = (uint32_t)(i - (uint64_t)g * kcap);
    const uint32_t* blk = packed + (size_t)g * planes * nq * kcap + (size_t)q * kcap;
    const uint32_t row = blk[j];
    keys[i] = row == 0xFFFFFFFFu ? kDeadKey : make_key(__uint_as_float(blk[(size_t)nq * kcap + j]), bases[g] + row);
}

// the same keys for every query of the batch at once (grid.y = query), each query's keys padded to an even count `stride`:
// what the radix SELECT takes (k <= kMaxSelectK).  The order inside a query's keys is (shard, place in the shard's list), i.e.
// ascending global row among equal distances — the order the selection's tie rule needs.
__global__ void __launch_bounds__(256)
k_shard_keys_all(const uint32_t* __restrict__ packed, const uint32_t* __restrict__ bases, uint32_t n_lists, uint32_t nq, uint32_t kcap,
                 uint32_t planes, uint32_t stride, uint64_t* __restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y;
    if (i >= stride) return;
    uint64_t key = kDeadKey;
    if (i < n_lists * kcap) {
        const uint32_t g = i / kcap, j = i - g * kcap;
        const uint32_t* blk = packed + (size_t)g * planes * nq * kcap + (size_t)q * kcap;
        const uint32_t row = blk[j];
        if (row != 0xFFFFFFFFu) key = make_key(__uint_as_float(blk[(size_t)nq * kcap + j]), bases[g] + row);
    }
    keys[(size_t)q * stride + i] = key;
}

size_t merge_select_workspace_bytes(uint32_t n_lists, uint32_t nq, uint32_t kcap, uint32_t k_out) {
    const size_t stride = ((size_t)n_lists * kcap + 1) & ~(size_t)1;
    return (size_t)nq * stride * sizeof(uint64_t) + 256 + select_workspace_bytes(nq, k_out);
}

// merge of the gathered per-shard lists for kMaxFusedK < k_out <= kMaxSelectK, all nq queries in one go: keys, then the radix
// selection (qv_select.hip) of the min(k_out, valid entries) smallest; outputs [nq][k_out], padded
hipError_t launch_merge_select(const uint32_t* d_packed, const uint32_t* d_bases, uint32_t n_lists, uint32_t nq, uint32_t kcap, uint32_t planes,
                               uint32_t k_out, uint32_t kk, void* d_ws, uint32_t* d_rows_out, float* d_dist_out, hipStream_t s) {
    const uint64_t n64 = (uint64_t)n_lists * kcap;
    if (n_lists == 0 || kcap == 0 || k_out == 0 || kk == 0 || kk > k_out || planes < 2 || n64 > 0x7FFFFF00ull || kk > n64) return hipErrorInvalidValue;
    const uint32_t stride = ((uint32_t)n64 + 1u) & ~1u;
    uint64_t* keys = static_cast<uint64_t*>(d_ws);
    void* sel_ws = static_cast<char*>(d_ws) + ((size_t)nq * stride * sizeof(uint64_t) + 255) / 256 * 256;
    hipLaunchKernelGGL(k_shard_keys_all, dim3((stride + 255) / 256, nq), dim3(256), 0, s, d_packed, d_bases, n_lists, nq, kcap, planes, stride, keys);
    return launch_select_topk(keys, stride, stride, nq, kk, k_out, sel_ws, d_rows_out, d_dist_out, s);
}

size_t merge_ranked_workspace_bytes(uint64_t n_keys) {
    return 2 * n_keys * sizeof(uint64_t) + radix_hist_words((uint32_t)n_keys) * sizeof(uint32_t) + 256;
}

hipError_t launch_merge_ranked(const uint32_t* d_packed, const uint32_t* d_bases, uint32_t n_lists, uint32_t nq, uint32_t q, uint32_t kcap, uint32_t planes,
                               uint32_t k_out, void* d_ws, uint32_t* d_rows_out, float* d_dist_out, hipStream_t s) {
    const uint64_t n64 = (uint64_t)n_lists * kcap;
    if (n_lists == 0 || kcap == 0 || k_out == 0 || planes < 2 || n64 > 0xFFFFFF00ull) return hipErrorInvalidValue;
    const uint32_t n = (uint32_t)n64;
    uint64_t* ka = static_cast<uint64_t*>(d_ws);
    uint64_t* kb = ka + n;
    uint32_t* hist = reinterpret_cast<uint32_t*>(kb + n);
    hipLaunchKernelGGL(k_shard_keys, dim3((n + 255) / 256), dim3(256), 0, s, d_packed, d_bases, n_lists, nq, q, kcap, planes, ka);
    uint64_t* in = nullptr;
    hipError_t e = launch_radix_sort_hi32(ka, kb, n, hist, &in, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_emit_topk, dim3((k_out + 255) / 256), dim3(256), 0, s, in, n, k_out, d_rows_out, d_dist_out);
    return hipGetLastError();
}

// merged global row -> the payload its shard computed for it (e.g. the distance to a negative example): thread i finds
// row rows[i] in the list of the shard whose id range holds it
__global__ void __launch_bounds__(256)
k_lookup_payload(const uint32_t* __restrict__ packed, const uint32_t* __restrict__ bases, uint32_t n_lists, uint32_t nq, uint32_t q, uint32_t kcap,
                 uint32_t planes, uint32_t plane, const uint32_t* __restrict__ rows, uint32_t n, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = rows[i];
    float v = __uint_as_float(0x7F800000u);
    if (r != 0xFFFFFFFFu) {
        uint32_t g = 0;
        while (g + 1 < n_lists && bases[g + 1] <= r) g++;
        const uint32_t local = r - bases[g];
        const uint32_t* blk = packed + (size_t)g * planes * nq * kcap + (size_t)q * kcap;
        for (uint32_t j = 0; j < kcap; j++)
            if (blk[j] == local) { v = __uint_as_float(blk[(size_t)plane * nq * kcap + j]); break; }
    }
    out[i] = v;
}

hipError_t launch_lookup_payload(const uint32_t* d_packed, const uint32_t* d_bases, uint32_t n_lists, uint32_t nq, uint32_t q, uint32_t kcap, uint32_t planes,
                                 uint32_t plane, const uint32_t* d_rows, uint32_t n, float* d_out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_lookup_payload, dim3((n + 255) / 256), dim3(256), 0, s, d_packed, d_bases, n_lists, nq, q, kcap, planes, plane, d_rows, n, d_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------- scan + selection (kMaxFusedK < k <= kMaxSelectK) -----
// keys of a group of queries live together: 1 GiB of them at most (128 queries of 1M rows, 13 of 10M)
static uint32_t flat_select_group(uint32_t n_tiles, uint32_t nq) {
    const size_t per_query = (size_t)n_tiles * 64 * sizeof(uint64_t);
    return (uint32_t)std::max<size_t>(1, std::min<size_t>(nq, ((size_t)1 << 30) / per_query));
}
size_t flat_select_workspace_bytes(uint32_t n_tiles, uint32_t nq, uint32_t kk, uint32_t dim4) {
    const uint32_t g = flat_select_group(n_tiles, nq);
    return (size_t)g * n_tiles * 64 * sizeof(uint64_t) + 256 + select_workspace_bytes(g, kk) + 256 + (g >= 2 ? flat_keys_mq_workspace_bytes(g, dim4) : 0);
}
static hipError_t flat_select_impl(const IndexView& v, const ScanPlan& p, const float* d_queries, uint32_t nq, uint32_t kk, uint32_t k_stride,
                                   void* d_ws, uint32_t* d_rows_out, float* d_dist_out, hipStream_t s, const uint32_t* d_active);
hipError_t launch_flat_select(const IndexView& v, const ScanPlan& p, const float* d_queries, uint32_t nq, uint32_t kk, uint32_t k_stride,
                              void* d_ws, uint32_t* d_rows_out, float* d_dist_out, hipStream_t s) {
    return flat_select_impl(v, p, d_queries, nq, kk, k_stride, d_ws, d_rows_out, d_dist_out, s, nullptr);
}
// d_active (a device word, or null): only the first *d_active queries are worked on — every kernel's other workgroups leave at once
// (launch_flat_select_redo: the count is decided on the device).  With it the queries go through the shared passes (nq >= 2, one group).
static hipError_t flat_select_impl(const IndexView& v, const ScanPlan& p, const float* d_queries, uint32_t nq, uint32_t kk, uint32_t k_stride,
                                   void* d_ws, uint32_t* d_rows_out, float* d_dist_out, hipStream_t s, const uint32_t* d_active) {
    const uint32_t n = v.n_tiles * 64;
    const uint32_t g = flat_select_group(v.n_tiles, nq);
    uint64_t* keys = static_cast<uint64_t*>(d_ws);
    void* sel_ws = static_cast<char*>(d_ws) + ((size_t)g * n * sizeof(uint64_t) + 255) / 256 * 256;
    const uint32_t batch = key_batch(v.metric, v.dim4, true);
    const size_t lds = query_lds_bytes(v.metric, v.dim4) + (size_t)kScanWaves * batch * 64 * sizeof(uint64_t) + (size_t)kSelBins * sizeof(uint32_t);
    void* qws = static_cast<char*>(sel_ws) + (select_workspace_bytes(g, kk) + 255) / 256 * 256;
    for (uint32_t q0 = 0; q0 < nq; q0 += g) {
        const uint32_t m = std::min(g, nq - q0);
        if (m >= 2) {
            // several queries: they share corpus passes (4 or 8 per pass) instead of reading the corpus once each — 8 queries at
            // k = 100 over 1M x 768: one ~0.6 ms pass instead of eight 0.46 ms ones; the selection counts its first window itself
            hipError_t e2 = launch_flat_keys_mq(v, p, d_queries + (size_t)q0 * v.dim, m, keys, qws, s, d_active);
            if (e2 != hipSuccess) return e2;
            e2 = launch_select_topk(keys, n, n, m, kk, k_stride, sel_ws, d_rows_out + (size_t)q0 * k_stride, d_dist_out + (size_t)q0 * k_stride, s, false, true, d_active);
            if (e2 != hipSuccess) return e2;
            continue;
        }
        if (d_active != nullptr) return hipErrorInvalidValue;             // (the lone-query kernel takes no device-side count)
        SelState* st = nullptr; uint32_t* hist = nullptr;
        hipError_t e = select_prepare(sel_ws, m, kk, &st, &hist, s);     // states and histograms zeroed: the keys kernel counts window 0
        if (e != hipSuccess) return e;
        QV_DISPATCH_METRIC(v.metric, {
            e = set_lds(k_flat_keys<MM, kUnroll, true>, lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((k_flat_keys<MM, kUnroll, true>), dim3(p.grid, m), dim3(p.block), lds, s, v, d_queries + (size_t)q0 * v.dim, keys, st, hist, kk, select_cap(kk), batch);
        });
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        e = launch_select_topk(keys, n, n, m, kk, k_stride, sel_ws, d_rows_out + (size_t)q0 * k_stride, d_dist_out + (size_t)q0 * k_stride, s, true);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ---------------------------------------------------------------- hand-backs above 64 results per query, redone WITHOUT the host -----
// The k <= 64 scheme (qv_scan.hip: k_redo_compact + k_flat_scan_redo) for the selection path: the flagged queries are listed on the device,
// and for every kLkRedoSlots list entries the exact path runs on a gathered copy of their vectors — shared corpus passes writing a key per
// row (k_flat_keys_mq), the radix selection, the results scattered back to the queries' own slots — with every kernel's workgroups
// leaving at once where the list has no entry for them.  ceil(nq / slots) rounds of ten launches are issued whatever the flags say: ~0.1 ms
// of empty launches per 256 queries, the price of a call that never waits for its own filter.
constexpr uint32_t kLkRedoSlots = 64;
static uint32_t lk_redo_slots(uint32_t n_tiles, uint32_t nq) { return flat_select_group(n_tiles, std::min(nq, kLkRedoSlots)); }
__global__ void __launch_bounds__(1024)
k_lk_redo_list(const uint32_t* __restrict__ flags, uint32_t nq, uint32_t slots, uint32_t* __restrict__ list, uint32_t* __restrict__ count /* [0] flagged, [1 + r] entries of round r */) {
    __shared__ uint32_t s_n;
    __shared__ uint32_t w_off[16];
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < nq; base += blockDim.x) {             // in query order (a wave at a time: ballot + prefix popcount)
        const uint32_t q = base + threadIdx.x;
        const bool f = q < nq && flags[q] != 0;
        const uint64_t m = __ballot(f);
        if (lane == 0) w_off[wave] = (uint32_t)__builtin_popcountll(m);
        __syncthreads();
        uint32_t before = s_n;
        for (uint32_t w = 0; w < wave; w++) before += w_off[w];
        if (f) list[before + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = q;
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t t = 0; for (uint32_t w = 0; w < (blockDim.x >> 6); w++) t += w_off[w]; s_n += t; }
        __syncthreads();
    }
    const uint32_t total = s_n, rounds = (nq + slots - 1) / slots;
    if (threadIdx.x == 0) count[0] = total;
    for (uint32_t r = threadIdx.x; r < rounds; r += blockDim.x) count[1 + r] = total > r * slots ? (total - r * slots < slots ? total - r * slots : slots) : 0u;
}
// slot j of the round takes the vector of the list's entry first + j (zeros where the list has none: the query blocks are laid out for every slot)
__global__ void __launch_bounds__(256)
k_lk_redo_gather(const float* __restrict__ queries, uint32_t dim, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, uint32_t first, float* __restrict__ rq) {
    const uint32_t j = blockIdx.x;
    const bool on = first + j < count[0];
    const float* src = queries + (size_t)(on ? list[first + j] : 0u) * dim;
    for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x) rq[(size_t)j * dim + i] = on ? src[i] : 0.0f;
}
__global__ void __launch_bounds__(256)
k_lk_redo_scatter(const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, uint32_t first, const uint32_t* __restrict__ t_rows, const float* __restrict__ t_dist,
                  uint32_t k_stride, uint32_t* __restrict__ rows_out, float* __restrict__ dist_out) {
    const uint32_t j = blockIdx.x;
    if (first + j >= count[0]) return;
    const uint32_t q = list[first + j];
    for (uint32_t i = threadIdx.x; i < k_stride; i += blockDim.x) {
        rows_out[(size_t)q * k_stride + i] = t_rows[(size_t)j * k_stride + i];
        dist_out[(size_t)q * k_stride + i] = t_dist[(size_t)j * k_stride + i];
    }
}
static size_t up256(size_t b) { return (b + 255) / 256 * 256; }
size_t flat_select_redo_workspace_bytes(uint32_t n_tiles, uint32_t nq, uint32_t kk, uint32_t dim, uint32_t dim4) {
    const uint32_t slots = lk_redo_slots(n_tiles, nq);
    return up256((size_t)nq * 4) + up256((size_t)(nq + 2) * 4) + up256((size_t)slots * dim * 4) + 2 * up256((size_t)slots * kk * 4) + flat_select_workspace_bytes(n_tiles, slots, kk, dim4) + 256;
}
hipError_t launch_flat_select_redo(const IndexView& v, const ScanPlan& p, const float* d_queries, uint32_t nq, uint32_t kk, uint32_t k_stride, const uint32_t* d_flags,
                                   void* d_ws, uint32_t* d_rows_out, float* d_dist_out, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    const uint32_t slots = lk_redo_slots(v.n_tiles, nq);
    if (kk <= (uint32_t)kMaxFusedK || kk > (uint32_t)kMaxSelectK || kk != k_stride || slots < 2) return hipErrorNotSupported;   // (a corpus whose keys leave room for one query at a time: the caller's host path)
    char* w = static_cast<char*>(d_ws);
    uint32_t* list = reinterpret_cast<uint32_t*>(w); w += up256((size_t)nq * 4);
    uint32_t* count = reinterpret_cast<uint32_t*>(w); w += up256((size_t)(nq + 2) * 4);
    float* rq = reinterpret_cast<float*>(w); w += up256((size_t)slots * v.dim * 4);
    uint32_t* t_rows = reinterpret_cast<uint32_t*>(w); w += up256((size_t)slots * kk * 4);
    float* t_dist = reinterpret_cast<float*>(w); w += up256((size_t)slots * kk * 4);
    hipLaunchKernelGGL(k_lk_redo_list, dim3(1), dim3(1024), 0, s, d_flags, nq, slots, list, count);
    uint32_t round = 0;
    for (uint32_t first = 0; first < nq; first += slots, round++) {
        hipLaunchKernelGGL(k_lk_redo_gather, dim3(slots), dim3(256), 0, s, d_queries, v.dim, list, count, first, rq);
        hipError_t e = flat_select_impl(v, p, rq, slots, kk, k_stride, w, t_rows, t_dist, s, count + 1 + round);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_lk_redo_scatter, dim3(slots), dim3(256), 0, s, list, count, first, t_rows, t_dist, k_stride, d_rows_out, d_dist_out);
    }
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
    const uint32_t batch = key_batch(v.metric, v.dim4, false);
    const size_t lds = query_lds_bytes(v.metric, v.dim4) + (size_t)kScanWaves * batch * 64 * sizeof(uint64_t);
    hipError_t e = hipSuccess;
    QV_DISPATCH_METRIC(v.metric, {
        e = set_lds(k_flat_keys<MM, kUnroll, false>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_flat_keys<MM, kUnroll, false>), dim3(p.grid), dim3(p.block), lds, s, v, d_query, ka, (SelState*)nullptr, (uint32_t*)nullptr, 0u, 0u, batch);
    });
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    uint64_t* in = nullptr;
    e = launch_radix_sort_hi32(ka, kb, n, hist, &in, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_emit_topk, dim3((k + 255) / 256), dim3(256), 0, s, in, n, k, d_rows_out, d_dist_out);
    return hipGetLastError();
}


}  // namespace qv
