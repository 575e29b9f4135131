// qv_hnsw.hip — device-resident HNSW traversal (exact-heap and wave-resident forms; each as a wave per query for thousands of
// traversals in flight and, for batches of at most one query per CU, as a workgroup of eight waves per query: "the latency form")
// (shared helpers, the arithmetic contract and the build flags: qv_kernels.h)
#include "qv_kernels.h"

namespace qv {

// ---------------------------------------------------------------- HNSW traversal ----
// Device-resident restatement of hnsw.HNSW.Search (pkg/hnsw/hnsw.go:602-713) and searchLayer
// (:471-580): one wavefront walks the graph for one query; many queries are in flight (the
// path is latency-bound per query).  To return exactly what the reference
// returns, the two heaps are the reference's binary heaps with its own sift loops
// (hnsw.go:101-196), executed by lane 0 on LDS arrays, and a hop's neighbours are admitted
// one by one in adjacency order (:536-563) after their distances have been computed together:
// lane i scores the i-th unvisited neighbour with the same sequential-over-dims arithmetic
// as every other kernel here, so distances — hence every heap decision — are bit-identical
// to the CPU restatement.  (The latency form adds a row's products as partial chains and takes the float32 from a certificate —
// or, where that does not decide it, from the same single chain: see "a row's sum over several lanes, certified" below.)
struct HRes { float dist; uint32_t idx; };
constexpr int kHnswCandCap = 2048;     // candidate min-heap slots per query (overflow -> host path)
constexpr int kHnswEfMax = 512;
constexpr int kHnswMaxDeg = 64;

// ---- visited sets ---------------------------------------------------------------------------------------------------
// searchLayer's visited map (hnsw.go:483-488) is per call and holds only the nodes the call evaluated, so its device form
// is sized by the SEARCH, not by the graph:
//   wave kernel  : an open-addressed hash table of `vis_cap` node ids per wave slot in global memory (L2/MALL-resident:
//                  32-64 KiB per slot), test-and-set by one atomicCAS per probe — lanes of a hop insert their neighbours
//                  concurrently, a duplicate inside one adjacency list (the self-link quirk) resolves itself: one lane
//                  gets EMPTY back, the other the node id.  Cleared by the wave at the start of every searchLayer.  A
//                  search that fills 3/4 of the table is flagged for the exact-heap kernel like a tie.
//   heap kernel  : one bit per node per slot (n/8 bytes), atomicOr test-and-set: never overflows, and the few slots of this
//                  kernel keep it small (1.25 MB per slot at 10M nodes).
// (Round 1 kept a uint32 stamp per (slot, node): 16 GB at 1M nodes x 4096 slots, unusable at 10M.)
constexpr uint32_t kVisEmpty = 0xFFFFFFFFu;
constexpr uint32_t kVisGreedyCap = 1024;   // table entries used by an ef = 1 (greedy, upper-level) search
__device__ __forceinline__ void vis_hash_clear(uint32_t* tab, uint32_t cap, uint32_t lane) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 e = {kVisEmpty, kVisEmpty, kVisEmpty, kVisEmpty};
    for (uint32_t i = lane * 4; i < cap; i += 256) *reinterpret_cast<u4*>(tab + i) = e;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the clears are in L2 before the first atomic probes it
}
// true if `node` was not in the table (and now is)
__device__ __forceinline__ bool vis_hash_insert(uint32_t* tab, uint32_t mask, uint32_t shift, uint32_t node) {
    uint32_t h = (node * 0x9E3779B1u) >> shift;
    for (;;) {
        const uint32_t old = atomicCAS(&tab[h], kVisEmpty, node);
        if (old == kVisEmpty) return true;
        if (old == node) return false;
        h = (h + 1) & mask;
    }
}
__device__ __forceinline__ void vis_bits_clear(uint32_t* bm, uint32_t words, uint32_t lane) {
    for (uint32_t i = lane; i < words; i += 64) bm[i] = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ bool vis_bits_insert(uint32_t* bm, uint32_t node) {
    const uint32_t bit = 1u << (node & 31);
    return (atomicOr(&bm[node >> 5], bit) & bit) == 0;
}

// The reference's container/heap sifts (hnsw.go:101-196), comparison for comparison; the moving element is held in
// registers and only the displaced child / parent is written per level ("hole" form: same comparisons, same final array).
__device__ __forceinline__ void h_min_up(HRes* rs, int j) {                      // hnsw.go:118-128
    const HRes x = rs[j];
    for (;;) { int i = (j - 1) / 2; if (i == j) break; const HRes p = rs[i]; if (x.dist >= p.dist) break; rs[j] = p; j = i; }
    rs[j] = x;
}
__device__ __forceinline__ void h_min_down(HRes* rs, int i0, int n) {            // hnsw.go:130-148
    int i = i0;
    const HRes x = rs[i0];
    for (;;) {
        int j1 = 2 * i + 1; if (j1 >= n || j1 < 0) break;
        HRes c = rs[j1]; int j = j1;
        if (j1 + 1 < n) { const HRes c2 = rs[j1 + 1]; if (c2.dist < c.dist) { c = c2; j = j1 + 1; } }
        if (x.dist <= c.dist) break;
        rs[i] = c; i = j;
    }
    rs[i] = x;
}
__device__ __forceinline__ void h_max_up(HRes* rs, int j) {                      // hnsw.go:172-181
    const HRes x = rs[j];
    for (;;) { int i = (j - 1) / 2; if (i == j) break; const HRes p = rs[i]; if (x.dist <= p.dist) break; rs[j] = p; j = i; }
    rs[j] = x;
}
__device__ __forceinline__ void h_max_down(HRes* rs, int i0, int n) {            // hnsw.go:183-200
    int i = i0;
    const HRes x = rs[i0];
    for (;;) {
        int j1 = 2 * i + 1; if (j1 >= n || j1 < 0) break;
        HRes c = rs[j1]; int j = j1;
        if (j1 + 1 < n) { const HRes c2 = rs[j1 + 1]; if (c2.dist > c.dist) { c = c2; j = j1 + 1; } }
        if (x.dist >= c.dist) break;
        rs[i] = c; i = j;
    }
    rs[i] = x;
}

// The same two sifts done by the whole wave (round 3).  The exact-heap kernel is a lone wave whose lane 0 walked these loops level
// by level through LDS (~130 cycles a level, ~20 sifts per hop: 11.5 us of a 30 us hop).  Same comparisons, same final array:
//   up(j):   the ancestors of j are known in advance (a_l = ((j + 1) >> l) - 1): lane l reads ancestor l, one ballot finds the first
//            parent the new element does NOT pass (hnsw.go:121 / :175), the parents below it move down one slot, the element lands.
//            One LDS read and one LDS write in all.
//   down():  the element sinks along the path of PREFERRED children (the smaller / larger of each pair, hnsw.go:137-139 / :190-192),
//            which does not depend on the element.  62 lanes read the five levels below the hole at once, each inner node picks
//            its preferred child by two lane shuffles, one ballot + a five-step scalar walk gives the path, a second ballot the
//            first path node the element does not pass (hnsw.go:140 / :193); the path above it moves up one slot.  Five levels per
//            round: two rounds for a 512-entry result heap, three for 4096 candidates.
template <bool MAXH>
__device__ __forceinline__ void wave_heap_up(HRes* rs, int j, HRes x, uint32_t lane) {
    const int D = 31 - __builtin_clz((uint32_t)j + 1u);              // ancestors a_1 .. a_D (a_D = the root)
    const int l = (int)lane;
    HRes p = {0.f, 0u};
    const bool anc = l >= 1 && l <= D;
    if (anc) p = rs[(int)(((uint32_t)j + 1u) >> l) - 1];
    const bool stop = anc && (MAXH ? x.dist <= p.dist : x.dist >= p.dist);
    const uint64_t m = __ballot(stop);
    const int L = m ? (int)__builtin_ctzll(m) : D + 1;               // the element stays at ancestor L - 1
    if (anc && l < L) rs[(int)(((uint32_t)j + 1u) >> (l - 1)) - 1] = p;
    if (l == 0) rs[(int)(((uint32_t)j + 1u) >> (L - 1)) - 1] = x;
}
// x sinks from the root of rs[0..n) (rs[0] is the hole)
template <bool MAXH>
__device__ __forceinline__ void wave_heap_down(HRes* rs, int n, HRes x, uint32_t lane) {
    uint32_t p = 0;                                                   // the hole (wave-uniform)
    const uint32_t t = lane + 2;                                      // lane m holds local node m + 2 of the subtree below the hole (local 1 = the hole)
    const uint32_t d = 31u - (uint32_t)__builtin_clz(t);
    const uint32_t lc = (2 * t - 2) & 63u;                            // lane of this node's left child (nodes of the first four levels: lanes 0..29)
    for (;;) {
        if (2 * p + 1 >= (uint32_t)n) break;                          // the hole is a leaf
        const uint32_t gi = ((p + 1u) << d) - 1u + (t - (1u << d));
        const bool have = lane < 62 && gi < (uint32_t)n;
        HRes c = {0.f, 0u};
        if (have) c = rs[gi];
        const uint64_t HM = __ballot(have);
        const float dl = __shfl(c.dist, (int)lc), dr = __shfl(c.dist, (int)((lc + 1) & 63u));
        const bool right = lane <= 29 && ((HM >> ((lc + 1) & 63u)) & 1ull) && (MAXH ? dr > dl : dr < dl);
        const uint64_t RB = __ballot(right);
        const float d0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c.dist), 0));
        const float d1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c.dist), 1));
        const bool right1 = ((HM >> 1) & 1ull) && (MAXH ? d1 > d0 : d1 < d0);
        uint64_t path = 0; uint32_t tcur = 1, last_lane = 0;
#pragma unroll
        for (int step = 0; step < 5; step++) {
            const uint32_t left = 2 * tcur - 2;
            if (tcur > 31 || !((HM >> left) & 1ull)) break;           // no child in the heap (or below the five levels read)
            const uint32_t r = tcur == 1 ? (right1 ? 1u : 0u) : (uint32_t)((RB >> (tcur - 2)) & 1ull);
            last_lane = left + r;
            path |= 1ull << last_lane;
            tcur = last_lane + 2;
        }
        const bool onpath = (path >> lane) & 1ull;
        const bool stop = onpath && (MAXH ? x.dist >= c.dist : x.dist <= c.dist);
        const uint64_t SM = __ballot(stop);
        const uint32_t first = SM ? (uint32_t)__builtin_ctzll(SM) : 64u;
        if (onpath && lane < first) rs[(gi - 1u) >> 1] = c;           // the path above the stopping point moves up one slot
        if (SM) { p = ((uint32_t)__builtin_amdgcn_readlane((int)gi, (int)first) - 1u) >> 1; break; }
        p = (uint32_t)__builtin_amdgcn_readlane((int)gi, (int)last_lane);
    }
    if (lane == 0) rs[p] = x;
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
// readlanes and DPP wave shifts — no LDS heaps, so a hop's serial part shrinks from ~40 us to a few us
// and the LDS is left to the row stream below.  With equal distances the binary heaps' pop order depends on their
// layout, so a query that ever sees two equal distances in the list (or a NaN) is flagged and
// re-run by k_hnsw_search (the exact-heap form): results stay identical to the reference in all cases.
constexpr uint32_t kHnswTieFlag = 0xFFFFFFFEu;
// Neighbour rows of one hop are streamed through LDS in column slabs by LDS-DMA
// (global_load_lds_dwordx4: no VGPR destination, every piece of a slab in flight at once):
//   slab   = kHnswSlab consecutive 16-byte chunks of up to kHnswRound rows, two slab buffers;
//   piece  = one DMA instruction = 8 rows x 128 contiguous bytes (lane L: row L/8, 16-byte slot L%8),
//            the shape the memory system serves at full rate for gathered rows;
//   image  = [group of 8 rows][piece of 8 chunks][row][slot]; slot j of row r holds chunk j ^ sw(r),
//            sw = (r%8) ^ ((r/8)&1) — applied on the SOURCE address, the LDS destination of a DMA is
//            lane-linear — so that the per-lane reads "chunk c of MY row" of 16 rows cover all 64 banks.
// While lane r walks slab s of its row sequentially (same arithmetic order as everywhere else), slab
// s+1 is landing in the other buffer.  The query is NOT in LDS: it is pre-converted to the metric's
// Q type in global memory (k_hnsw_prep_queries) and read at wave-uniform addresses, i.e. by scalar
// loads straight into SGPR operands of v_fma_f64.
// What bounds it (tools/ubench/f64chain.hip, MI355X): a dependent v_fma_f64 chain runs at 10.6 cycles per
// element, v_cvt_f64_f32 issues in ~12.6 and v_fma_f64 in ~8.6, so one wave-hop costs 768 x 21 cycles of
// VALU issue whatever the lane count: with 4 waves per SIMD the kernel is VALU-issue-bound on the
// convert + fma pair, which is why occupancy (small slabs, no query in LDS) is what pays.
#ifndef QV_HNSW_SLAB
#define QV_HNSW_SLAB 8
#endif
#ifndef QV_HNSW_LAT_WAVES
#define QV_HNSW_LAT_WAVES 8
#endif
constexpr int kHnswSlab = QV_HNSW_SLAB;   // chunks per slab (128 B of each row): 2 x 4 KiB of slab buffers per wave -> 16 waves per CU
                                          // (measured 5k x 768, efSearch 128: slab 8 -> 554k QPS, 16 -> 430k, 32 -> 209k: occupancy wins)
constexpr int kHnswRound = 32;            // rows per round (MaxM0 = 32 by default: one round per hop)
constexpr int kHnswSlabBytes = kHnswRound * kHnswSlab * 16;
// The exact-heap kernel runs a handful of flagged queries, one wave per CU: nothing else covers a slab's DMA latency (~1.7 us
// against 0.3 us of arithmetic per 128-byte slab), so it takes 1 KiB of every row per slab: 3 slabs per 768-d hop, arithmetic-
// bound.  Measured per hop at efSearch 512 (QV_HNSW_PROF): evaluation 16 -> 14 us, heap inserts 11.5 us, links + visited 2.7 us,
// pop 1.7 us (profiles/r02_hnsw_build_sweeps.txt).
constexpr int kHnswHeapSlab = 64;
template <int SL> constexpr int slab_bytes() { return kHnswRound * SL * 16; }

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

__device__ __forceinline__ void glds16(const float* g, lds_u8* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

// what this lane fetches in the DMA pieces of a round: for each group of 8 rows, row (group*8 + lane/8), slot lane%8.
// src* already include the swizzled slot offset; lanes of rows beyond the round's count point at the round's first
// row (valid memory) and fill LDS slots nobody reads, so a full piece needs no exec masking.
struct DmaRole {
    const float* src0; const float* src1; const float* src2; const float* src3;
    uint32_t sw0, sw1;      // slot -> chunk swizzle for even / odd groups (tail pieces only)
    uint32_t ng;
};
__device__ __forceinline__ void dma_role(DmaRole& r, const float* rowmaj, uint32_t dim, const lds_u32* batch, uint32_t cnt, uint32_t lane) {
    const uint32_t drow = lane >> 3, dslot = lane & 7;
    r.ng = (cnt + 7) >> 3;
    r.sw0 = dslot ^ drow; r.sw1 = dslot ^ drow ^ 1u;
    r.src0 = rowmaj + (size_t)batch[drow < cnt ? drow : 0u] * dim + r.sw0 * 4;
    r.src1 = rowmaj + (size_t)batch[8 + drow < cnt ? 8 + drow : 0u] * dim + r.sw1 * 4;
    r.src2 = rowmaj + (size_t)batch[16 + drow < cnt ? 16 + drow : 0u] * dim + r.sw0 * 4;
    r.src3 = rowmaj + (size_t)batch[24 + drow < cnt ? 24 + drow : 0u] * dim + r.sw1 * 4;
}
template <bool FULL, int SL>
__device__ __forceinline__ void dma_issue_group(const float* src, uint32_t sw, uint32_t chunk0, uint32_t dim4, lds_u8* dst) {
#pragma unroll
    for (int p = 0; p < SL / 8; p++) {
        const uint32_t cbase = chunk0 + (uint32_t)p * 8;
        if (FULL) glds16(src + (size_t)cbase * 4, dst + p * 1024);
        else if (cbase + sw < dim4) glds16(src + (size_t)cbase * 4, dst + p * 1024);
    }
}
template <int SL>
__device__ __forceinline__ void dma_issue_slab(const DmaRole& r, uint32_t sl, uint32_t dim4, lds_u8* buf) {
    const uint32_t chunk0 = sl * SL;
    constexpr int G = SL / 8 * 1024;
    if (chunk0 + SL <= dim4) {
        dma_issue_group<true, SL>(r.src0, r.sw0, chunk0, dim4, buf);
        if (r.ng > 1) dma_issue_group<true, SL>(r.src1, r.sw1, chunk0, dim4, buf + G);
        if (r.ng > 2) dma_issue_group<true, SL>(r.src2, r.sw0, chunk0, dim4, buf + 2 * G);
        if (r.ng > 3) dma_issue_group<true, SL>(r.src3, r.sw1, chunk0, dim4, buf + 3 * G);
    } else {
        dma_issue_group<false, SL>(r.src0, r.sw0, chunk0, dim4, buf);
        if (r.ng > 1) dma_issue_group<false, SL>(r.src1, r.sw1, chunk0, dim4, buf + G);
        if (r.ng > 2) dma_issue_group<false, SL>(r.src2, r.sw0, chunk0, dim4, buf + 2 * G);
        if (r.ng > 3) dma_issue_group<false, SL>(r.src3, r.sw1, chunk0, dim4, buf + 3 * G);
    }
}
// The query values of one slab, one per lane: lane L (0..31; the upper half mirrors it) holds element 32*sl + L of the query
// in the metric's Q type, fetched with ONE coalesced vector load per slab beside the slab's row DMA.  Element i is broadcast
// to the arithmetic by v_readlane (wave-uniform -> an SGPR operand of the fma), so the query costs neither LDS bandwidth nor
// scalar-cache traffic.  (Round 1 read it with s_load_dwordx16 at wave-uniform addresses: 16 waves x 6 KiB of queries cycle
// through a 16 KiB scalar cache, four 64-byte loads per 32 elements, each drained with lgkmcnt(0) together with the row
// reads.  Measured 1M x 768: efSearch 64 / 128 / 256 +6 % each; what bounds the traversal is the gather of 3 KiB rows itself,
// 4.5-5.2 TB/s against 6.6 TB/s for the bare LDS-DMA row stream — profiles/r02_hnsw_query_operand.txt.)
template <typename Q, int SL> struct QSlab;
template <int SL> struct QSlab<double, SL> {
    static constexpr int R = (SL * 4 + 63) / 64;          // registers (pairs) per lane: a slab's SL*4 query elements over 64 lanes
    uint32_t lo[R], hi[R];
    __device__ __forceinline__ void load(const double* __restrict__ q, uint32_t sl, uint32_t n_el, uint32_t lane) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t e = sl * (uint32_t)(SL * 4) + (uint32_t)r * 64 + (SL * 4 < 64 ? (lane & (uint32_t)(SL * 4 - 1)) : lane);
            const double v = e < n_el ? q[e] : 0.0;
            lo[r] = (uint32_t)__double2loint(v); hi[r] = (uint32_t)__double2hiint(v);
        }
    }
    __device__ __forceinline__ double at(uint32_t i) const {
        // the register is chosen AFTER the cross-lane read (a scalar select): a vector select under a partial exec mask would
        // leave the source lane undefined when that lane is not active.  With a constant i only one pair of reads remains.
        uint32_t h = __builtin_amdgcn_readlane(hi[0], i & 63), l = __builtin_amdgcn_readlane(lo[0], i & 63);
#pragma unroll
        for (int r = 1; r < R; r++)
            if ((i >> 6) == (uint32_t)r) { h = __builtin_amdgcn_readlane(hi[r], i & 63); l = __builtin_amdgcn_readlane(lo[r], i & 63); }
        return __hiloint2double((int)h, (int)l);
    }
};
template <int SL> struct QSlab<float, SL> {
    static constexpr int R = (SL * 4 + 63) / 64;
    uint32_t w[R];
    __device__ __forceinline__ void load(const float* __restrict__ q, uint32_t sl, uint32_t n_el, uint32_t lane) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t e = sl * (uint32_t)(SL * 4) + (uint32_t)r * 64 + (SL * 4 < 64 ? (lane & (uint32_t)(SL * 4 - 1)) : lane);
            w[r] = __float_as_uint(e < n_el ? q[e] : 0.0f);
        }
    }
    __device__ __forceinline__ float at(uint32_t i) const {
        uint32_t x = __builtin_amdgcn_readlane(w[0], i & 63);
#pragma unroll
        for (int r = 1; r < R; r++)
            if ((i >> 6) == (uint32_t)r) x = __builtin_amdgcn_readlane(w[r], i & 63);
        return __uint_as_float(x);
    }
};
static_assert(((kHnswSlab * 4) & (kHnswSlab * 4 - 1)) == 0 && kHnswSlab % 8 == 0 && kHnswHeapSlab % 8 == 0, "slabs are whole 128-byte pieces");

// lane walks slab `sl` of row r (0..31) of the round, sequentially over the dims
template <int M, int SL>
__device__ __forceinline__ void slab_accumulate(typename MT<M>::A& acc, const lds_u8* buf, uint32_t r, const QSlab<typename MT<M>::Q, SL>& qs,
                                                uint32_t sl, uint32_t dim4) {
    typedef const __attribute__((address_space(3))) f4* lds_f4p;
    const uint32_t mg = r >> 3, mr = r & 7, msw = mr ^ (mg & 1);
    const lds_u8* mine = buf + mg * (SL / 8 * 1024) + mr * 128;
    const uint32_t c0 = sl * SL, c1 = c0 + SL < dim4 ? c0 + SL : dim4;
    uint32_t c = c0;
#pragma unroll
    for (int pi = 0; pi < SL / 8; pi++) {
        if (c + 8 > c1) break;
        const lds_u8* pc = mine + pi * 1024;
        const uint32_t e0 = (uint32_t)pi * 32;                         // first query element of this piece within the slab
        f4 x0 = *(lds_f4p)(pc + ((0u ^ msw) << 4)), x1 = *(lds_f4p)(pc + ((1u ^ msw) << 4)), x2 = *(lds_f4p)(pc + ((2u ^ msw) << 4)), x3 = *(lds_f4p)(pc + ((3u ^ msw) << 4));
        f4 x4 = *(lds_f4p)(pc + ((4u ^ msw) << 4)), x5 = *(lds_f4p)(pc + ((5u ^ msw) << 4)), x6 = *(lds_f4p)(pc + ((6u ^ msw) << 4)), x7 = *(lds_f4p)(pc + ((7u ^ msw) << 4));
#define QV_ACC4(X, O) acc1<M>(acc, qs.at(e0 + O), X.x); acc1<M>(acc, qs.at(e0 + O + 1), X.y); acc1<M>(acc, qs.at(e0 + O + 2), X.z); acc1<M>(acc, qs.at(e0 + O + 3), X.w);
        QV_ACC4(x0, 0) QV_ACC4(x1, 4) QV_ACC4(x2, 8) QV_ACC4(x3, 12) QV_ACC4(x4, 16) QV_ACC4(x5, 20) QV_ACC4(x6, 24) QV_ACC4(x7, 28)
        c += 8;
    }
    for (; c < c1; c++) {
        const f4 x = *(lds_f4p)(mine + ((c - c0) >> 3) * 1024 + ((((c - c0) & 7) ^ msw) << 4));
        const uint32_t e0 = (c - c0) * 4;
        QV_ACC4(x, 0)
    }
#undef QV_ACC4
}

// distance of the wave's query to the rows batch[0..n) (n <= 64): lane i returns distance(query, batch[i]).
// Row-major copy present: LDS-DMA slabs as described above; otherwise each lane pulls its row from the tile layout.
template <int M, int U, int SL = kHnswSlab>
__device__ __forceinline__ float hnsw_eval_rows(const IndexView& v, const lds_u32* batch_l, lds_u8* slabs_l,
                                                const typename MT<M>::Q* __restrict__ q_g, const QConst& qc, uint32_t n, uint32_t lane) {
    float out = 0.0f;
    const bool use_rm = v.rowmaj != nullptr && (v.dim & 3) == 0;
    if (use_rm) {
        const uint32_t nslab = (v.dim4 + SL - 1) / SL;
        for (uint32_t base = 0; base < n; base += kHnswRound) {
            const uint32_t cnt = n - base < (uint32_t)kHnswRound ? n - base : (uint32_t)kHnswRound;
            const bool me = lane >= base && lane < base + cnt;       // row r of this round sits on lane base + r
            const uint32_t myrow = me ? batch_l[lane] : 0u;
            double rn = 0.0;
            if constexpr (MT<M>::needs_rnorm) { if (me) rn = v.rnorm[myrow]; }
            DmaRole role;
            dma_role(role, v.rowmaj, v.dim, batch_l + base, cnt, lane);
            typename MT<M>::A acc = 0;
            QSlab<typename MT<M>::Q, SL> q_cur, q_nxt;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            dma_issue_slab<SL>(role, 0, v.dim4, slabs_l);
            q_cur.load(q_g, 0, v.dim4 * 4, lane);
            for (uint32_t sl = 0; sl < nslab; sl++) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // slab sl and its query values have landed
                q_nxt = q_cur;
                if (sl + 1 < nslab) {
                    dma_issue_slab<SL>(role, sl + 1, v.dim4, slabs_l + ((sl + 1) & 1) * slab_bytes<SL>());   // lands while slab sl is consumed
                    q_nxt.load(q_g, sl + 1, v.dim4 * 4, lane);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (me) slab_accumulate<M, SL>(acc, slabs_l + (sl & 1) * slab_bytes<SL>(), (lane - base) & 31u, q_cur, sl, v.dim4);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this buffer's reads are done before it is refilled
                q_cur = q_nxt;
            }
            if (me) out = finalize<M>(acc, qc, rn);
        }
        return out;
    }
    if (lane < n) {
        const uint32_t row = batch_l[lane];
        typename MT<M>::A acc = row_accumulate<M, U, false>(reinterpret_cast<const f4*>(v.tiles) + (size_t)(row >> 6) * v.dim4 * 64 + (row & 63), 64, q_g, v.dim4);
        double rn = 0.0;
        if constexpr (MT<M>::needs_rnorm) rn = v.rnorm[row];
        out = finalize<M>(acc, qc, rn);
    }
    return out;
}

// ---- the query through LDS ---------------------------------------------------------------------------------------------
// The traversal kernels' form of hnsw_eval_rows (row-major copy, dim a multiple of 32).  Measured with QV_HNSW_PROF at 1M x 768,
// efSearch 128: a hop spends 79 k of its 110 k cycles evaluating rows, 3.5 x the 22 k cycles its 768-step chain takes alone —
// the four waves of a SIMD queue for the vector ALU, which is ~80 % busy.  (Keeping a second slab in flight per wave, by
// reading a slab into registers at once and re-requesting its buffer early, changed nothing: 589 k vs 581 k QPS.)  So the
// instructions per dimension are what counts: convert + fma, and — in the plain form — two v_readlane to broadcast the query
// value from its one-per-lane register.  Here the query's 32 values of a slab arrive by LDS-DMA beside the rows (two 256-byte
// buffers) and are read with uniform-address ds_reads one 16-byte chunk ahead of the arithmetic: the broadcast costs LDS
// bandwidth instead of VALU issue slots, and no load with a register destination is left in the loop.
constexpr int kHnswQBufBytes = kHnswSlab * 4 * 8;          // one slab of query values (32 x float64; float32 metrics use half)
// LDS of one wave of the wave-per-query form: batch[64] | 2 row slabs | 2 query slabs | pos[32] + pad | (resident query)
constexpr int kHnswWaveFixedLds = 64 * 4 + 2 * kHnswSlabBytes + 2 * kHnswQBufBytes + 32 * 4 + 64;
static inline bool hnsw_qlds_ok(const IndexView& v) { return v.rowmaj != nullptr && (v.dim & 31u) == 0 && v.dim >= 32; }

template <typename Q>
__device__ __forceinline__ void dma_issue_query(const Q* __restrict__ q_g, uint32_t sl, lds_u8* qbuf, uint32_t lane) {
    constexpr uint32_t lanes = kHnswSlab * 4 * sizeof(Q) / 16;                    // 16 (float64) or 8 (float32) lanes x 16 bytes
    if (lane < lanes) glds16(reinterpret_cast<const float*>(q_g + (size_t)sl * (kHnswSlab * 4)) + lane * 4, qbuf);
}
// lane walks its row's 8 chunks of the slab; chunk c+1 and the query's values for it are requested before chunk c is consumed
template <int M>
__device__ __forceinline__ void slab_accumulate_qlds(typename MT<M>::A& acc, const lds_u8* buf, uint32_t r, const lds_u8* qbuf) {
    typedef const __attribute__((address_space(3))) f4* lds_f4p;
    typedef typename MT<M>::Q Q;
    typedef const __attribute__((address_space(3))) Q* lds_qp;
    const uint32_t mg = r >> 3, mr = r & 7, msw = mr ^ (mg & 1);
    const lds_u8* mine = buf + mg * 1024 + mr * 128;
    lds_qp q = (lds_qp)qbuf;
    f4 x[2]; Q qq[2][4];
    x[0] = *(lds_f4p)(mine + ((0u ^ msw) << 4));
#pragma unroll
    for (int e = 0; e < 4; e++) qq[0][e] = q[e];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int cur = c & 1, nxt = cur ^ 1;
        if (c + 1 < 8) {
            x[nxt] = *(lds_f4p)(mine + (((uint32_t)(c + 1) ^ msw) << 4));
#pragma unroll
            for (int e = 0; e < 4; e++) qq[nxt][e] = q[4 * (c + 1) + e];
        }
        acc1<M>(acc, qq[cur][0], x[cur].x); acc1<M>(acc, qq[cur][1], x[cur].y); acc1<M>(acc, qq[cur][2], x[cur].z); acc1<M>(acc, qq[cur][3], x[cur].w);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// slabs_l: 2 x kHnswSlabBytes of row buffers followed by 2 x kHnswQBufBytes of query buffers.  Requires hnsw_qlds_ok(v).
template <int M, int U>
__device__ __forceinline__ float hnsw_eval_rows_qlds(const IndexView& v, const lds_u32* batch_l, lds_u8* slabs_l,
                                                     const typename MT<M>::Q* __restrict__ q_g, const QConst& qc, uint32_t n, uint32_t lane) {
    static_assert(kHnswSlab == 8, "one 128-byte piece per row and slab");
    lds_u8* qbufs = slabs_l + 2 * kHnswSlabBytes;
    float out = 0.0f;
    const uint32_t nslab = v.dim4 >> 3;
    for (uint32_t base = 0; base < n; base += kHnswRound) {
        const uint32_t cnt = n - base < (uint32_t)kHnswRound ? n - base : (uint32_t)kHnswRound;
        const bool me = lane >= base && lane < base + cnt;       // row r of this round sits on lane base + r
        const uint32_t myrow = me ? batch_l[lane] : 0u;
        double rn = 0.0;
        if constexpr (MT<M>::needs_rnorm) { if (me) rn = v.rnorm[myrow]; }
        DmaRole role;
        dma_role(role, v.rowmaj, v.dim, batch_l + base, cnt, lane);
        const uint32_t r = (lane - base) & 31u;
        typename MT<M>::A acc = 0;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        dma_issue_slab<kHnswSlab>(role, 0, v.dim4, slabs_l); dma_issue_query(q_g, 0, qbufs, lane);
        for (uint32_t sl = 0; sl < nslab; sl++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // slab sl and its query values have landed
            if (sl + 1 < nslab) {                                  // the next slab lands while this one is consumed
                dma_issue_slab<kHnswSlab>(role, sl + 1, v.dim4, slabs_l + ((sl + 1) & 1) * kHnswSlabBytes);
                dma_issue_query(q_g, sl + 1, qbufs + ((sl + 1) & 1) * kHnswQBufBytes, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (me) slab_accumulate_qlds<M>(acc, slabs_l + (sl & 1) * kHnswSlabBytes, r, qbufs + (sl & 1) * kHnswQBufBytes);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this slab's buffers are read before they are refilled
        }
        if (me) out = finalize<M>(acc, qc, rn);
    }
    return out;
}

// ---- a hop whose first slab is already there ------------------------------------------------------------------------------
// The wave-per-query traversal requests slab 0 of EVERY link of the popped node before it knows which of them are new (beside the
// visited test, which is the other round trip it would otherwise wait for): the rows sit at their adjacency positions in slab
// buffer 0.  Then the new ones are compacted (batch[0..n)), slabs 1.. are requested for those only, and lane r walks slab 0 of its
// row at the row's adjacency position `pos` and the later slabs at position r.  Same arithmetic, same order, one round trip less.
//
// The query is RESIDENT in LDS as the caller's float32 words (3 KiB at 768 dimensions; widening a float32 to the metric's Q type is
// exact, so k_hnsw_prep_queries' converted block and this give the same operand): a lane reads ONE word of it per slab (lane i:
// element 32 sl + i) and each step takes its operand by v_readlane.  Round 5 streamed the converted query slab by slab (an LDS-DMA
// per slab and wave) and read it with two uniform-address ds_read_b128 per four steps: 16 LDS instructions per slab for the query
// against 8 for the rows, on an LDS that also takes every row byte from the DMA.  tools/ubench/gather_mix.hip, 12 waves per CU:
// the bare row stream 6.63 TB/s; with the chain and the round-5 query path 6.13; with this 6.54.
template <int M>
__device__ __forceinline__ void slab_accumulate_qreg(typename MT<M>::A& acc, const lds_u8* buf, uint32_t r, uint32_t qw) {
    typedef const __attribute__((address_space(3))) f4* lds_f4p;
    typedef typename MT<M>::Q Q;
    const uint32_t mg = r >> 3, mr = r & 7, msw = mr ^ (mg & 1);
    const lds_u8* mine = buf + mg * 1024 + mr * 128;
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
}
// The caller has waited for slab 0 (vmcnt(0)).  n <= 32.  qres: the query's float32 words in LDS.
template <int M>
__device__ __forceinline__ float hnsw_eval_hop_front(const IndexView& v, const lds_u32* batch_l, lds_u8* slabs_l, const QConst& qc, uint32_t n, uint32_t pos, uint32_t lane,
                                                     const lds_u32* qres) {
    const uint32_t nslab = v.dim4 >> 3;
    const bool me = lane < n;
    const uint32_t myrow = me ? batch_l[lane] : 0u;
    double rn = 0.0;
    if constexpr (MT<M>::needs_rnorm) { if (me) rn = v.rnorm[myrow]; }          // (used at the end: lands beside slab 1)
    DmaRole role;
    dma_role(role, v.rowmaj, v.dim, batch_l, n, lane);
    typename MT<M>::A acc = 0;
    uint32_t qw = qres[lane & 31u];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (uint32_t sl = 0; sl < nslab; sl++) {
        if (sl) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // slab sl has landed
        uint32_t qn = 0;
        if (sl + 1 < nslab) {                                           // the next slab lands while this one is consumed
            dma_issue_slab<kHnswSlab>(role, sl + 1, v.dim4, slabs_l + ((sl + 1) & 1) * kHnswSlabBytes);
            qn = qres[(sl + 1) * 32u + (lane & 31u)];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (me) slab_accumulate_qreg<M>(acc, slabs_l + (sl & 1) * kHnswSlabBytes, sl ? lane : pos, qw);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // this slab's buffer is read before it is refilled
        qw = qn;
    }
    return me ? finalize<M>(acc, qc, rn) : 0.0f;
}

#ifdef QV_HNSW_P512
// EXPERIMENT of the measurement build (tools/build_variant.sh p512 qv_hnsw.hip -DQV_HNSW_P512), TIMING ONLY: a hop's rows in groups of 8,
// 512 contiguous bytes of each row per slab (a quarter of the address translations and DRAM pages of the 128-byte pieces), eight lanes
// per row each walking 16 elements of the slab, the eight partial chains added WITHOUT the certificate — so the distances are not the
// reference's bits and no test may run on this build.  What it answers: is the 512-byte shape worth building the certificate for?
template <int M>
__device__ __forceinline__ float hnsw_eval_hop_p512(const IndexView& v, const lds_u32* batch_l, lds_u8* slabs_l, const QConst& qc, uint32_t n, uint32_t lane, const lds_u32* qres) {
    typedef const __attribute__((address_space(3))) f4* lds_f4p;
    const uint32_t drow = lane >> 3, dslot = lane & 7;
    const uint32_t ncs = v.dim >> 7, ngr = (n + 7) >> 3, total = ngr * ncs;
    double rn = 0.0;
    if constexpr (MT<M>::needs_rnorm) { if (lane < n) rn = v.rnorm[batch_l[lane]]; }
    float mine = 0.0f;
    double acc = 0.0;
    auto issue = [&](uint32_t t) {
        const uint32_t gr = t / ncs, cs = t - gr * ncs, r = gr * 8 + drow;
        const uint32_t row = batch_l[r < n ? r : 0u];
        const float* src = v.rowmaj + (size_t)row * v.dim + cs * 128 + dslot * 4;
        lds_u8* b = slabs_l + (t & 1) * kHnswSlabBytes;
#pragma unroll
        for (int j = 0; j < 4; j++) glds16(src + j * 32, b + j * 1024);
    };
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    issue(0);
    for (uint32_t t = 0; t < total; t++) {
        const uint32_t gr = t / ncs, cs = t - gr * ncs;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t + 1 < total) issue(t + 1);
        __builtin_amdgcn_sched_barrier(0);
        const lds_u8* b = slabs_l + (t & 1) * kHnswSlabBytes;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const f4 x = *(lds_f4p)(b + j * 1024 + lane * 16);
            const f4 q = *(lds_f4p)((const lds_u8*)qres + (cs * 128 + j * 32 + dslot * 4) * 4);
            acc = __builtin_fma((double)q.x, (double)x.x, acc); acc = __builtin_fma((double)q.y, (double)x.y, acc);
            acc = __builtin_fma((double)q.z, (double)x.z, acc); acc = __builtin_fma((double)q.w, (double)x.w, acc);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (cs == ncs - 1) {
            double s = acc;
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
            const double got = __shfl(s, (int)(((lane - gr * 8) & 7u) * 8u));
            if (lane >= gr * 8 && lane < gr * 8 + 8 && lane < n) mine = finalize<M>(got, qc, rn);
            acc = 0.0;
        }
    }
    return mine;
}
#endif

// ---- visited set in buckets (wave-per-query traversal) ------------------------------------------------------------------------
// The table of a wave slot as buckets of 16 words (64 bytes: one request).  A test reads the node's whole bucket: ONE round trip
// says present / absent and where the first free word is.  The open-addressed form took one dependent round trip per probe, and
// the slowest of a hop's 32 lanes decided: at the end of an efSearch-128 search (4.7 k of 8 k words used) two to three per hop.
// A bucket fills from word 0 up; a full bucket sends its keys on to the next one, so a lookup goes on exactly while the bucket it
// read has no free word.  The table belongs to ONE wave slot and the lanes of a hop carry distinct nodes (repeats inside an
// adjacency list are masked before), so "absent" is final when it is read; claiming the free word is an atomicCAS only because two
// lanes of one hop can pick the same word — the loser tests and claims again (vis_bucket_settle) — and its answer is not needed
// before the hop's rows have been walked.
__device__ __forceinline__ bool vis_bucket_test(uint32_t* tab, uint32_t bmask, uint32_t bshift, uint32_t node, bool active, uint32_t*& word) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    uint32_t b = (node * 0x9E3779B1u) >> bshift;
    bool fresh = false, pend = active;
    word = nullptr;
    while (__ballot(pend)) {
        if (pend) {
            // sc1: served by L2, where the claims (atomics) are made — a plain load could hit a line this CU's L1 kept from an earlier
            // hop, without the words claimed since.  (The wait also covers the slab-0 requests issued before the test.)
            const uint32_t* p = tab + (size_t)b * 16;
            u4 w0, w1, w2, w3;
            asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                         "global_load_dwordx4 %2, %4, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:48 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"(p) : "memory");
            const uint32_t w[16] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w};
            uint32_t hit = 0, empty = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) { hit |= (w[i] == node) ? 1u : 0u; empty |= (w[i] == kVisEmpty) ? (1u << i) : 0u; }
            if (hit) pend = false;
            else if (empty) { fresh = true; word = tab + (size_t)b * 16 + (uint32_t)__builtin_ctz(empty); pend = false; }
            else b = (b + 1) & bmask;
        }
    }
    return fresh;
}
// the claims of a hop (old = what atomicCAS(word, empty, node) returned): a lane that lost its word to another lane of the hop goes again
__device__ __forceinline__ void vis_bucket_settle(uint32_t* tab, uint32_t bmask, uint32_t bshift, uint32_t node, bool fresh, uint32_t old) {
    bool lost = fresh && old != kVisEmpty;
    while (__ballot(lost)) {
        uint32_t* word;
        (void)vis_bucket_test(tab, bmask, bshift, node, lost, word);       // (absent, by construction: finds the next free word)
        uint32_t o2 = kVisEmpty;
        if (lost) o2 = atomicCAS(word, kVisEmpty, node);
        lost = lost && o2 != kVisEmpty;
    }
}

// ---- a row's sum over several lanes, certified -----------------------------------------------------------------------
// The reference's value is ONE chain: 768 dependent rounded additions per row (distances.go:18-22), 21 cycles each on one wave
// (convert + fma, tools/ubench/f64chain.hip) — 9 us per hop however few rows the hop has.
// A different summation order gives a different float64, but the distance is that float64 pushed through a MONOTONE function
// (finalize: divide by a positive constant, clamp, subtract from one / square root, round to float32 — every step monotone, so
// their composition is), and both orders lie within a computable distance of the exact sum of the (exactly representable)
// products: |computed - exact| <= g(h) * sum|p_i| with g(h) = h u / (1 - h u), u = 2^-53, h = the longest chain of additions
// (Higham, Accuracy and Stability of Numerical Algorithms, 4.2).  So with S = the sum of P partial chains and
// B >= (g(dim) + g(dim / P + log2 P)) * sum|p_i|, the reference's float64 lies in [S - B, S + B]; when finalize(S - B) and
// finalize(S + B) are the same float32 — all but a few in a million evaluations: B is ~2e-13 of the operands' size, a float32
// step is 6e-8 — that float32 IS the reference's, bit for bit.  Otherwise the row is walked again as one chain.
// sum|p_i| <= |q| |r| (Cauchy-Schwarz; the cached norms) for cosine and dot; for the metrics whose terms are >= 0 it is the sum itself.
// kSplitSlack covers the norms' own rounding, the rounding of S -+ B and of B, and the second-order terms.
// Used where a traversal has lanes and waves to spare — the latency form below.  (In the wave-per-query form, two / four / eight
// lanes per row measured 3.1 -> 2.8 ms for a lone query and -5 % at full occupancy, where the 19 extra registers cost the fourth
// wave per SIMD at efSearch 128: not kept there.)
template <int M> struct SplitOK { static constexpr bool value = M == QV_COSINE || M == QV_DOT || M == QV_L2 || M == QV_L1 || M == QV_L2SQ_F64; };
constexpr double kSplitSlack = 128.0;
template <int M> __device__ __forceinline__ double split_bound(double s, double k_u, const QConst& qc, double rn) {
    if constexpr (M == QV_COSINE || M == QV_DOT) return k_u * qc.qn * rn;    // (dot: the query's norm comes from k_hnsw_prep_queries, the row's is cached for every float64 metric)
    else return k_u * s;
}

// ---- the latency form: one workgroup of W waves per query ----------------------------------------------------------------
// A batch that leaves most of the device idle (a lone caller; a shared batch of a few dozen callers) is bound by ONE query's chain
// of hops, and a hop by its parts in sequence.  Measured at 1M x 768, efSearch 128, four queries in flight (QV_HNSW_PROF):
// 150 hops of 16.5 us = 2.5 ms — evaluation 9.2 us (24 slabs, each a DMA round trip nothing else covers, ~0.4 us), adjacency list
// + visited table 4.2 us (two dependent round trips to L2), list inserts 2.8 us.  With a CU to itself a query takes the whole
// LDS instead: every row of the hop is requested at once (32 x 3 KiB in flight, one round trip), the query stays resident,
// the visited table is in LDS — and W waves evaluate the hop: wave w requests and walks columns [w, w + 1) * dim / W of every
// row, its 64 lanes split those columns 2 / 4 / 8 ways per row (cnt <= 32 / 16 / 8), so a row's 768-step chain becomes chains of
// 48 / 24 / 12 steps (eight waves) whose sum is certified as above (rows that fail are walked again from LDS as ONE chain:
// the reference's order).  Wave 0 keeps the list and drives; the others wait at a barrier between hops.
// workgroup barrier of the latency form: LDS traffic only (a __syncthreads would also wait for the vector-memory queue — the row
// norms and the adjacency prefetch the driver has in flight — ~1.5 k cycles per hop)
__device__ __forceinline__ void lat_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
struct LatLds {
    lds_u32* batch;     // [64] the hop's rows
    lds_u32* ctrl;      // [0] rows of the round (0xFFFFFFFF: leave), [1] first row of the round in batch[]
    lds_u8*  part;      // [W][32] float64 partial sums
    lds_u8*  q;         // the query in the metric's Q type, dim4 * 4 elements
    lds_u8*  rows;      // [4 groups of 8 rows][dim4 / 8 pieces][8 rows][8 slots] x 16 bytes (slot j of row r: chunk j ^ sw(r))
};
constexpr int kLatWaves = QV_HNSW_LAT_WAVES;                            // waves per query in the latency form
constexpr uint32_t kLatQOff = 512 + kLatWaves * 256 + 512;              // batch, ctrl, partial sums below it (a multiple of 1 KiB)
static_assert(kLatQOff % 1024 == 0, "LDS-DMA blocks are 1 KiB");
__host__ __device__ inline uint32_t lat_q_bytes(uint32_t dim4, uint32_t qsize) { return (dim4 * 4 * qsize + 1023u) & ~1023u; }
// rows of a round (32, or 16 / 8 where 32 rows of this dimension do not fit the LDS) x dim4 chunks; never below 8 KiB: between hops the
// buffer is the scratch the list's batched admissions scatter through (576 entries x 12 bytes at most)
__host__ __device__ inline uint32_t lat_rows_bytes(uint32_t dim4, uint32_t round_rows) { const uint32_t b = (round_rows >> 3) * (dim4 >> 3) * 1024u; return b < 8192u ? 8192u : b; }

// chunks [c_lo, c_hi) of row r of the round, in order (c_lo = 0, c_hi = dim4: the reference's chain)
template <int M>
__device__ __forceinline__ typename MT<M>::A lat_row_chain(const lds_u8* rows, const lds_u8* qb, uint32_t nP, uint32_t r, uint32_t c_lo, uint32_t c_hi) {
    typedef const __attribute__((address_space(3))) f4* lds_f4p;
    typedef typename MT<M>::Q Q;
    typedef const __attribute__((address_space(3))) Q* lds_qp;
    const uint32_t mg = r >> 3, mr = r & 7, msw = mr ^ (mg & 1);
    const lds_u8* mine = rows + mg * nP * 1024 + mr * 128;
    lds_qp q = (lds_qp)qb;
    typename MT<M>::A acc = 0;
    if (c_lo >= c_hi) return acc;
    // two register sets, each loaded a chunk ahead of its use (an index past the range is clamped: a load nobody uses)
    auto row_at = [&](uint32_t c) -> f4 { return *(lds_f4p)(mine + (c >> 3) * 1024 + (((c & 7) ^ msw) << 4)); };
    const uint32_t c_last = c_hi - 1;
    uint32_t c = c_lo;
    f4 xa = row_at(c), xb;
    Q a0 = q[4 * c], a1 = q[4 * c + 1], a2 = q[4 * c + 2], a3 = q[4 * c + 3], b0, b1, b2, b3;
    for (; c + 1 < c_hi; c += 2) {
        const uint32_t cb = c + 1, ca = c + 2 < c_hi ? c + 2 : c_last;
        xb = row_at(cb); b0 = q[4 * cb]; b1 = q[4 * cb + 1]; b2 = q[4 * cb + 2]; b3 = q[4 * cb + 3];
        acc1<M>(acc, a0, xa.x); acc1<M>(acc, a1, xa.y); acc1<M>(acc, a2, xa.z); acc1<M>(acc, a3, xa.w);
        xa = row_at(ca); a0 = q[4 * ca]; a1 = q[4 * ca + 1]; a2 = q[4 * ca + 2]; a3 = q[4 * ca + 3];
        acc1<M>(acc, b0, xb.x); acc1<M>(acc, b1, xb.y); acc1<M>(acc, b2, xb.z); acc1<M>(acc, b3, xb.w);
    }
    if (c < c_hi) { acc1<M>(acc, a0, xa.x); acc1<M>(acc, a1, xa.y); acc1<M>(acc, a2, xa.z); acc1<M>(acc, a3, xa.w); }
    return acc;
}
// one wave's share of a round (every wave of the workgroup calls it between the two barriers of the round)
template <int M, int W>
__device__ __forceinline__ void lat_round_part(const IndexView& v, const LatLds& L, uint32_t base, uint32_t cnt, uint32_t wave, uint32_t lane, uint64_t* st = nullptr) {
#ifdef QV_HNSW_PROF
    uint64_t t_l = __builtin_readcyclecounter();
#define LTICK(i) if (st) { const uint64_t t_n = __builtin_readcyclecounter(); st[i] += t_n - t_l; t_l = t_n; }
#else
#define LTICK(i)
#endif
    const uint32_t nP = v.dim4 >> 3;
    const uint32_t p_lo = wave * nP / W, p_hi = (wave + 1) * nP / W;     // this wave's pieces of every row
    const uint32_t drow = lane >> 3, dslot = lane & 7, ng = (cnt + 7) >> 3;
    uint32_t rowid[4];
#pragma unroll
    for (uint32_t gi = 0; gi < 4; gi++) {                                // (one LDS round trip for the four groups' row numbers)
        const uint32_t rr = gi * 8 + drow;
        rowid[gi] = L.batch[base + (rr < cnt ? rr : 0u)];               // lanes past the round's rows fetch its first row into slots nobody reads
    }
#pragma unroll
    for (uint32_t gi = 0; gi < 4; gi++) {
        if (gi >= ng) break;
        const float* src = v.rowmaj + (size_t)rowid[gi] * v.dim + ((dslot ^ drow ^ (gi & 1u)) << 2) + p_lo * 32;
        lds_u8* dst = L.rows + (gi * nP + p_lo) * 1024;
        for (uint32_t p = p_lo; p < p_hi; p++, src += 32, dst += 1024) glds16(src, dst);
    }
    const uint32_t rows_cap = cnt <= 8 ? 8u : (cnt <= 16 ? 16u : 32u), lpr = 64u / rows_cap;
    const uint32_t r = lane & (rows_cap - 1), sub = lane / rows_cap;
    const uint32_t c0 = p_lo * 8, n = (p_hi - p_lo) * 8;
    const uint32_t my_lo = c0 + sub * n / lpr, my_hi = c0 + (sub + 1) * n / lpr;
    LTICK(8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // this wave's columns have landed (it reads no others)
    LTICK(9);
    double s = 0.0;
    if (r < cnt) s = (double)lat_row_chain<M>(L.rows, L.q, nP, r, my_lo, my_hi);
    LTICK(10);
    for (uint32_t off = rows_cap; off < 64; off <<= 1) s = s + __shfl_xor(s, (int)off);
    if (lane < rows_cap && r < cnt) *(__attribute__((address_space(3))) double*)(L.part + (wave * 32 + r) * 8) = s;
    LTICK(11);
#undef LTICK
}
// the driver's side of a hop: distance(query, batch[i]) on lane i for i < n
template <int M, int W>
__device__ __forceinline__ float lat_eval_rows(const IndexView& v, const LatLds& L, const QConst& qc, uint32_t n, uint32_t lane, uint32_t round_rows, uint64_t* st = nullptr) {
#ifdef QV_HNSW_PROF
    uint64_t t_l = __builtin_readcyclecounter();
#define LTICK(i) if (st) { const uint64_t t_n = __builtin_readcyclecounter(); st[i] += t_n - t_l; t_l = t_n; }
#else
#define LTICK(i)
#endif
    float out = 0.0f;
    const uint32_t nP = v.dim4 >> 3;
    for (uint32_t base = 0; base < n; base += round_rows) {
        const uint32_t cnt = n - base < round_rows ? n - base : round_rows;
        const bool me = lane >= base && lane < base + cnt;
        double rn = 0.0;
        if constexpr (MT<M>::needs_rnorm || M == QV_DOT) { if (me) rn = v.rnorm[L.batch[lane]]; }
        if (lane == 0) { L.ctrl[0] = cnt; L.ctrl[1] = base; }
        lat_barrier();                                                  // the round is posted (and batch[] is visible to every wave)
        LTICK(12);
        lat_round_part<M, W>(v, L, base, cnt, 0, lane, st);
#ifdef QV_HNSW_PROF
        t_l = __builtin_readcyclecounter();
#endif
        lat_barrier();                                                  // every wave's columns are in LDS, every partial sum written
        LTICK(13);
        // the two ends of the interval on the two halves of the wave: lane base + r takes S - B, lane (base + r) ^ 32 takes S + B
        float d = 0.0f; bool ok = true;
        {
            typedef const __attribute__((address_space(3))) double* lds_dp;
            const uint32_t r = (lane - base) & 31u;                       // (the same row on lane and lane ^ 32)
            const bool has = r < cnt;
            const bool low = ((lane - base) & 63u) < 32u;
            double rn2 = __shfl_xor(rn, 32);
            if (low) rn2 = rn;
            double s = 0.0;
            if (has) {
                lds_dp pp = (lds_dp)L.part + r;
                s = pp[0];
#pragma unroll
                for (int w = 1; w < W; w++) s = s + pp[w * 32];
            }
            const double k_u = ((double)(2u * v.dim) + kSplitSlack) * 0x1p-53;
            const double b = split_bound<M>(s, k_u, qc, rn2);
            const float de = finalize<M>((typename MT<M>::A)(low ? s - b : s + b), qc, rn2);
            const float other = __shfl_xor(de, 32);
            ok = !me || (__float_as_uint(de) == __float_as_uint(other) && de == de);
            d = de;
        }
        if (__ballot(!ok)) { if (me && !ok) d = finalize<M>(lat_row_chain<M>(L.rows, L.q, nP, lane - base, 0, v.dim4), qc, rn); }   // the rows are all here: one chain
        if (me) out = d;
        LTICK(14);
    }
#undef LTICK
    return out;
}
// visited table in LDS (the latency form, when it fits): the same open-addressed table
typedef __attribute__((address_space(3))) uint32_t* lds_u32w;
__device__ __forceinline__ void vis_hash_clear_lds(lds_u32w tab, uint32_t cap, uint32_t lane) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 e = {kVisEmpty, kVisEmpty, kVisEmpty, kVisEmpty};
    for (uint32_t i = lane * 4; i < cap; i += 256) *reinterpret_cast<__attribute__((address_space(3))) u4*>(tab + i) = e;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
__device__ __forceinline__ bool vis_hash_insert_lds(lds_u32w tab, uint32_t mask, uint32_t shift, uint32_t node) {
    uint32_t h = (node * 0x9E3779B1u) >> shift;
    for (;;) {
        uint32_t expected = kVisEmpty;
        if (__hip_atomic_compare_exchange_strong(tab + h, &expected, node, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return true;
        if (expected == node) return false;
        h = (h + 1) & mask;
    }
}

// lanes whose node already stands on a LOWER lane of the same adjacency list (the self-link quirk: rare).  c = the list, one node
// per lane, 0xFFFFFFFF past its end; every lane index is a constant, so a step is v_readlane + v_cmp + two scalar operations
// and no branch (the loop over a register lane index took 3.7 k cycles per hop: 1.5 us of a lone traversal's 10).
__device__ __forceinline__ uint64_t repeats_in_list(uint32_t c, uint32_t deg) {
    uint64_t rep = 0;
#pragma unroll
    for (int j = 0; j < kHnswMaxDeg - 1; j++) {
        if ((j & 7) == 0 && (uint32_t)j + 1 >= deg) break;             // (wave-uniform)
        const uint32_t cj = __builtin_amdgcn_readlane(c, j);
        rep |= __ballot(c == cj) & ~((2ull << j) - 1ull);
    }
    return rep;
}

#ifdef QV_HNSW_PROF
#define HTICK(ph) tick(ph)
#else
#define HTICK(ph)
#endif

// one wave (64-thread workgroup) per query stream
// Build mode (o.qlevel != null, connectNode's searches hnsw.go:367-385): query i stands for node o.qnode0 + i, descends
// greedily to min(level, cur_level) and searches THAT level with ef exactly; the ascending result is re-ordered inside
// equal-distance runs by node index (selectNeighbors' order, hnsw.go:589-594) and cut to the level's degree bound.
template <int M, int U, int W = 1>
__global__ void __launch_bounds__(64 * W)
k_hnsw_search(IndexView v, GraphView g, const typename MT<M>::Q* __restrict__ qblk, const double* __restrict__ qconst, uint32_t nq, uint32_t k, uint32_t ef_search,
              HnswOpts o, uint32_t cand_cap,
              uint32_t* __restrict__ rows_out, float* __restrict__ dist_out, uint32_t* __restrict__ count_out, uint32_t* __restrict__ evals_out) {
    using Q = typename MT<M>::Q;
    extern __shared__ __align__(16) unsigned char smem[];
    // LDS, W == 1: two row slabs (hnsw_eval_rows) | candidate heap | result heap | the hop's batch | its distances
    //      W  > 1 (the latency form, as in k_hnsw_search_wave: the other waves only take their share of each hop's rows):
    //             batch, round, partial sums | query | the hop's rows | candidate heap | result heap | distances
    LatLds L;
    L.batch = (lds_u32*)smem; L.ctrl = (lds_u32*)smem + 64; L.part = (lds_u8*)smem + 512; L.q = (lds_u8*)smem + kLatQOff;
    L.rows = L.q + lat_q_bytes(v.dim4, (uint32_t)sizeof(Q));
    lds_u8* slabs_l = (lds_u8*)smem;
    unsigned char* base = W > 1 ? smem + kLatQOff + lat_q_bytes(v.dim4, (uint32_t)sizeof(Q)) + lat_rows_bytes(v.dim4, o.lat_rows) : smem + 2 * slab_bytes<kHnswHeapSlab>();
    HRes* cand = reinterpret_cast<HRes*>(base);                       // [cand_cap]
    HRes* res = cand + cand_cap;                                      // [kHnswEfMax + 1]
    uint32_t* batch = W > 1 ? reinterpret_cast<uint32_t*>(smem) : reinterpret_cast<uint32_t*>(res + kHnswEfMax + 1);   // [kHnswMaxDeg]
    float* bd = W > 1 ? reinterpret_cast<float*>(res + kHnswEfMax + 1) : reinterpret_cast<float*>(batch + kHnswMaxDeg);   // [kHnswMaxDeg]
    const lds_u32* batch_l = W > 1 ? (const lds_u32*)smem
                                   : (const lds_u32*)((lds_u8*)smem + 2 * slab_bytes<kHnswHeapSlab>() + (size_t)(cand_cap + kHnswEfMax + 1) * sizeof(HRes));
    const Q* q_g = qblk;
    __shared__ int s_ncand, s_nres;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* bm = o.vis + (size_t)blockIdx.x * o.vis_cap;            // one bit per node
    const uint32_t bm_words = (g.n_nodes + 31) >> 5;
    const bool build = o.qlevel != nullptr;
    if constexpr (W > 1) {
        const uint32_t wave = threadIdx.x >> 6;
        if (wave != 0) {
            for (;;) {
                lat_barrier();
                const uint32_t cnt = L.ctrl[0], rbase = L.ctrl[1];
                if (cnt == 0xFFFFFFFFu) return;
                lat_round_part<M, W>(v, L, rbase, cnt, wave, lane);
                lat_barrier();
            }
        }
    }
    auto wsync = [&]() { if constexpr (W > 1) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } else __syncthreads(); };

    auto alive = [&](uint32_t n) -> bool { return n < g.n_nodes && g.level[n] >= 0; };
    // distances of batch[0..n) -> bd[0..n); lane i scores batch[i]
    QConst qc;
    auto eval = [&](uint32_t n) {
        float dd;
        if constexpr (W > 1) dd = lat_eval_rows<M, W>(v, L, qc, n, lane, o.lat_rows);
        else dd = hnsw_eval_rows<M, U, kHnswHeapSlab>(v, batch_l, slabs_l, q_g, qc, n, lane);
        if (lane < n) bd[lane] = dd;
        wsync();
    };
    // searchLayer (hnsw.go:471-580); result: res[0..s_nres) ascending; returns false on overflow
    uint32_t n_eval = 0;
#ifdef QV_HNSW_PROF
    uint64_t T[6] = {0, 0, 0, 0, 0, 0}; uint64_t t_last = wall_clock64(); uint64_t hops = 0;
    auto tk = [&](int ph) { uint64_t t = wall_clock64(); T[ph] += t - t_last; t_last = t; };
#define HTK(ph) tk(ph)
#else
#define HTK(ph)
#endif
    auto search_layer = [&](uint32_t entry, int ef, int level) -> bool {
        vis_bits_clear(bm, bm_words, lane);                            // :483-488
        if (lane == 0) { (void)vis_bits_insert(bm, entry); batch[0] = entry; }
        wsync();
        eval(1); n_eval += 1;                                          // :492
        if (lane == 0) {
            cand[0] = {bd[0], entry}; res[0] = {bd[0], entry};        // :498-506
            s_ncand = 1; s_nres = 1;
        }
        wsync();
        for (;;) {
            HTK(5);
            // every lane carries the heap sizes and walks the same branches (the sifts are wave operations)
            int nc = s_ncand, nr = s_nres;
            if (nc == 0) break;                                        // :509
            nc--;
            const HRes top = cand[0], lastc = cand[nc];                // :511 pop: the last element sinks from the root
            wsync();
            if (nc > 0) wave_heap_down<false>(cand, nc, lastc, lane);
            wsync();
            if (lane == 0) s_ncand = nc;
            if (nr >= ef && top.dist > res[0].dist) break;             // :514-516
            HTK(0);
            const uint32_t cur = top.idx;
            // neighbours of cur at `level` (:523-534).  On level 0 the degree and the (fixed-width) list are requested together,
            // and a graph without tombstones needs no level[] lookups at all: two dependent round trips less per hop for a
            // wave that has a CU to itself.
            uint32_t deg = 0; const uint32_t* links = nullptr;
            uint32_t c = 0xFFFFFFFFu; bool fresh = false;
            if (level == 0 && !g.has_dead && cur < g.n_nodes) {
                deg = g.l0_deg[cur];
                const uint32_t cl = lane < g.max_m0 ? g.l0_links[(size_t)cur * g.max_m0 + lane] : 0xFFFFFFFFu;
                if (lane < deg) { c = cl; fresh = c < g.n_nodes; }
            } else {
                if (alive(cur) && level <= (int)g.level[cur]) {
                    if (level == 0) { deg = g.l0_deg[cur]; links = g.l0_links + (size_t)cur * g.max_m0; }
                    else { const uint32_t* blk = g.up_links + (size_t)(g.up_off[cur] + (uint32_t)(level - 1)) * (1 + g.max_m); deg = blk[0]; links = blk + 1; }
                }
                if (lane < deg) { c = links[lane]; fresh = alive(c); }     // :539-541
            }
            // a list may hold the same node twice (the self-link quirk): only its first occurrence is new
            if ((repeats_in_list(c, deg) >> lane) & 1ull) fresh = false;
            if (fresh) fresh = vis_bits_insert(bm, c);                 // :543-544
            const uint64_t fm = __ballot(fresh);
            const uint32_t n = (uint32_t)__builtin_popcountll(fm);
            if (fresh) batch[__builtin_popcountll(fm & ((1ull << lane) - 1))] = c;   // adjacency order kept
            wsync();
            HTK(1);
            if (n == 0) continue;
            eval(n); n_eval += n;                                      // :548 (batched)
            HTK(2);
#ifdef QV_HNSW_PROF
            hops++;
#endif
            {
                bool overflow = false;
                // once the result heap is full its worst value only decreases: a neighbour that is not below it now never will be,
                // so one ballot drops those up front (most of a hop) and the admission test (:553) runs for the rest, in adjacency order
                const float mine = lane < n ? bd[lane] : 0.f;
                uint64_t pend = __ballot(lane < n && (nr < ef || mine < res[0].dist));
                while (pend) {
                    const uint32_t i = (uint32_t)__builtin_ctzll(pend);
                    pend &= pend - 1;
                    const float cd = bd[i];
                    if (nr < ef || cd < res[0].dist) {                 // :553
                        if (nc >= (int)cand_cap) { overflow = true; break; }
                        const HRes x = {cd, batch[i]};
                        wave_heap_up<false>(cand, nc, x, lane); nc++;                 // :554
                        wave_heap_up<true>(res, nr, x, lane); nr++;                   // :555
                        wsync();
                        if (nr > ef) {                                                // :558-560: the last element sinks from the root
                            nr--;
                            const HRes lastr = res[nr];
                            wsync();
                            wave_heap_down<true>(res, nr, lastr, lane);
                            wsync();
                        }
                    }
                }
                wsync();
                if (lane == 0) { s_ncand = nc; s_nres = nr; }
                wsync();
                if (overflow) return false;
            }
            HTK(3);
        }
        {                                                              // :566-577 heap -> ascending slice, in place: pop after pop
            const int nr = s_nres;
            for (int m = nr; m > 1; m--) {
                const HRes t = res[0], lastr = res[m - 1];
                wsync();
                if (lane == 0) res[m - 1] = t;
                if (m - 1 > 0) wave_heap_down<true>(res, m - 1, lastr, lane);
                wsync();
            }
        }
        wsync();
        return true;
    };

    const uint32_t n_run = o.redo_n ? *o.redo_n : nq;
    for (uint32_t ri = blockIdx.x; ri < n_run; ri += gridDim.x) {
        const uint32_t qi = o.redo_idx ? o.redo_idx[ri] : ri;
        q_g = qblk + (size_t)qi * v.dim4 * 4;
        qc.qn = qconst[(size_t)qi * 2]; qc.qn32 = (float)qconst[(size_t)qi * 2 + 1];
        n_eval = 0;
        if constexpr (W > 1) {                                          // the query into LDS (the other waves are at their barrier)
            const uint32_t qbytes = v.dim4 * 4 * (uint32_t)sizeof(Q);
            for (uint32_t off = 0; off < qbytes; off += 1024)
                if (off + lane * 16 < qbytes) glds16(reinterpret_cast<const float*>(q_g) + (off >> 2) + lane * 4, L.q + off);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        uint32_t entry = g.entry;
        bool ok = true;
        int stop = 0;
        if (build) { stop = (int)o.qlevel[qi]; if (stop > g.cur_level) stop = g.cur_level; }
        for (int level = g.cur_level; level > stop && ok; level--) {    // :649-657 / :367-380
            ok = search_layer(entry, 1, level);
            if (ok && s_nres > 0) entry = res[0].idx;
            wsync();
        }
        const int ef = build ? (int)ef_search : ((int)ef_search > (int)k ? (int)ef_search : (int)k);   // :660-663 / :385
        if (ok) ok = search_layer(entry, ef, stop);                     // :664
        uint32_t cnt = 0xFFFFFFFFu;                                     // overflow marker
        if (ok) {
            uint32_t kq = k;
            if (build) {
                kq = stop == 0 ? g.max_m0 : g.max_m;                    // :395-398
                if (lane == 0) {                                        // selectNeighbors' (Distance, VectorIndex) order, :589-594
                    const int nr = s_nres;
                    for (int i = 1; i < nr; i++) {
                        const HRes x = res[i]; int j = i;
                        while (j > 0 && res[j - 1].dist == x.dist && res[j - 1].idx > x.idx) { res[j] = res[j - 1]; j--; }
                        res[j] = x;
                    }
                }
                wsync();
            }
            cnt = (uint32_t)s_nres < kq ? (uint32_t)s_nres : kq;        // :670-672 (under-filled: the caller tops up, :676-710)
            for (uint32_t i = lane; i < k; i += 64) {
                rows_out[(size_t)qi * k + i] = i < cnt ? res[i].idx : 0xFFFFFFFFu;
                dist_out[(size_t)qi * k + i] = i < cnt ? res[i].dist : __uint_as_float(0x7F800000u);
            }
            if (build && stop >= 1 && o.self_dist) {                    // the node's lower levels link to itself (:463-467): d(node, node)
                wsync();
                if (lane == 0) batch[0] = o.qnode0 + qi;
                wsync();
                eval(1);
                if (lane == 0) o.self_dist[qi] = bd[0];
            }
        }
        if (lane == 0) { count_out[qi] = cnt; if (evals_out) evals_out[qi] = n_eval; }
        wsync();
#ifdef QV_HNSW_PROF
        if (lane == 0 && blockIdx.x == 1) printf("heap kernel q%u: pop %llu links+vis %llu eval %llu insert %llu other %llu (x10 ns) hops %llu evals %u\n", qi, T[0], T[1], T[2], T[3], T[5], hops, n_eval);
#endif
    }
    if constexpr (W > 1) { if (lane == 0) L.ctrl[0] = 0xFFFFFFFFu; lat_barrier(); }      // the other waves leave
}


template <int M, int U, int S, bool QLDS, int W = 1>
__global__ void __launch_bounds__(64 * W)
k_hnsw_search_wave(IndexView v, GraphView g, const typename MT<M>::Q* __restrict__ qblk, const double* __restrict__ qconst, uint32_t nq, uint32_t k, uint32_t ef_search,
                   HnswOpts o,
                   uint32_t* __restrict__ rows_out, float* __restrict__ dist_out, uint32_t* __restrict__ count_out, uint32_t* __restrict__ evals_out) {
    using Q = typename MT<M>::Q;
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* batch = reinterpret_cast<uint32_t*>(smem);                   // [64]
    // the same regions as LDS-address-space pointers (ds_read / LDS-DMA destinations)
    lds_u32* batch_l = (lds_u32*)smem;
    lds_u8* slabs_l = (lds_u8*)(batch_l + 64);                             // 2 x kHnswSlabBytes (row-major index only)
    const Q* q_g = qblk;                                                   // this wave's query, zero-padded to dim4*4, in the metric's Q type
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* tab = o.vis + (size_t)blockIdx.x * o.vis_cap;               // this wave slot's visited hash table
    // wave-per-query form (W == 1), row-major index, dimension a multiple of 32 (launch_hnsw_search_wave sets o.front):
    //   bit 0  the hop's front: slab 0 of every link requested beside the visited test, the visited set in buckets, the query resident
    //          in LDS as float32 words (behind the slab buffers) and broadcast by v_readlane (hnsw_eval_hop_front)
    //   bit 1  the adjacency list of the likely next pop requested a hop ahead
    const bool front = W == 1 && QLDS && (o.front & 1u);
    lds_u32w pos_l = (lds_u32w)(slabs_l + 2 * kHnswSlabBytes + 2 * kHnswQBufBytes);   // [32] adjacency position of the r-th new neighbour
    const lds_u32* qres_l = (const lds_u32*)(slabs_l + kHnswWaveFixedLds - 64 * 4);   // [dim] the query's float32 words (front only)
    // W > 1, the latency form (see LatLds): wave 0 runs everything below, the other waves only take their share of each hop's rows
    LatLds L;
    L.batch = (lds_u32*)batch_l; L.ctrl = (lds_u32*)batch_l + 64; L.part = (lds_u8*)smem + 512; L.q = (lds_u8*)smem + kLatQOff;
    L.rows = L.q + lat_q_bytes(v.dim4, (uint32_t)sizeof(Q));
    lds_u32w tab_l = (lds_u32w)(L.rows + lat_rows_bytes(v.dim4, o.lat_rows));          // (used when o.vis_lds)
    const bool vis_lds = W > 1 && o.vis_lds;
    if constexpr (W > 1) {
        const uint32_t wave = threadIdx.x >> 6;
        if (wave != 0) {
            for (;;) {
                lat_barrier();
                const uint32_t cnt = L.ctrl[0], base = L.ctrl[1];
                if (cnt == 0xFFFFFFFFu) return;
                lat_round_part<M, W>(v, L, base, cnt, wave, lane);
                lat_barrier();
            }
        }
    }
    // wave 0's own LDS traffic (batch[] written by some lanes, read by others): LDS operations of one wave execute in order
    // (one wave's LDS operations execute in order: no barrier, and — unlike __syncthreads — no wait for the vector-memory queue)
    auto wfence = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    auto wsync = [&]() { if constexpr (W > 1) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } else __syncthreads(); };
    const bool build = o.qlevel != nullptr;
    // a graph without tombstones (every graph built on the device) needs no level[] lookup to know a node is there
    auto alive = [&](uint32_t n) -> bool { return n < g.n_nodes && (!g.has_dead || g.level[n] >= 0); };
#ifdef QV_HNSW_PROF
    uint64_t T[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; uint64_t t_last = __builtin_readcyclecounter(); const uint64_t wc0 = wall_clock64(); uint64_t hops = 0;
    uint64_t full_hops = 0, full_rows = 0, surv_hops = 0, surv_rows = 0;
    auto tick = [&](int ph) { uint64_t t = __builtin_readcyclecounter(); T[ph] += t - t_last; t_last = t; };
#endif

    const bool kSpec = W > 1 || (W == 1 && QLDS && (o.front & 2u));   // the likely next adjacency list requested a hop ahead
    QConst qc;
    uint64_t key[S];          // ascending over index e = s*64 + lane; kDeadKey = empty
    uint64_t expd[S];         // wave-uniform: bit l of expd[s] = entry (s,l) already expanded
    uint32_t n_list = 0; bool tie = false;
    uint32_t n_eval = 0;

    // distance of the query to batch[lane] for lane < n  ->  64-bit key (all lanes return; dead beyond n)
    auto eval_keys = [&](uint32_t n) -> uint64_t {
        float dd;
#ifdef QV_HNSW_PROF
        if constexpr (W > 1) dd = lat_eval_rows<M, W>(v, L, qc, n, lane, o.lat_rows, T);
#else
        if constexpr (W > 1) dd = lat_eval_rows<M, W>(v, L, qc, n, lane, o.lat_rows);
#endif
        else dd = QLDS ? hnsw_eval_rows_qlds<M, U>(v, batch_l, slabs_l, q_g, qc, n, lane) : hnsw_eval_rows<M, U>(v, batch_l, slabs_l, q_g, qc, n, lane);
        return lane < n ? make_key(dd, batch_l[lane]) : kDeadKey;
    };
    // sorted insert of x (distance part xd) into the list; ef = size of the reference's result heap
    // Equal distances.  The reference's two binary heaps order equal keys by their layout, so a list model is exact only
    // while no decision falls on a tie.  Every decision compares VALUES except two, which pick an ELEMENT:
    //   * the eviction results.pop() (hnsw.go:558-560) picks a maximum.  When the two largest of the ef + 1 entries are equal,
    //     WHICH of them leaves the result heap is the heap's choice — but the one that leaves stays in the candidate heap
    //     at a distance that is not above the new worst, so it is still expanded when its turn comes (:514 is a strict >).
    //     The list therefore keeps the whole TIE GROUP at the worst value w = (ef-th smallest distance): ef or more entries,
    //     everything above w dropped, admission tested against w.  Which members of the group the result heap holds matters
    //     only if the output reaches into the group (checked at the end) — or if the group does not fit the registers;
    //   * candidates.pop() (:511) picks a minimum: ambiguous iff another UNEXPANDED entry has the popped entry's distance
    //     (checked at the pop).
    // Only those situations (and a NaN, and for a plain Search equal distances among the returned entries, whose order is
    // the heap's) flag the query for the exact-heap kernel.  Equal distances elsewhere in the list are harmless.
    auto entry_at = [&](uint32_t e) -> uint64_t {
        uint64_t r = kDeadKey;
#pragma unroll
        for (int s2 = 0; s2 < S; s2++) if ((int)(e >> 6) == s2) r = readlane64(key[s2], e & 63);
        return r;
    };
    auto insert = [&](uint64_t x, uint32_t ef) {
        const uint32_t xd = (uint32_t)(x >> 32);
        constexpr uint32_t kCap = (uint32_t)S * 64;
        if (xd == 0xFFFFFFFEu) tie = true;                              // NaN: heap order is not a function of distances
        if (n_list >= ef && xd >= (uint32_t)(entry_at(ef - 1) >> 32)) return;   // hnsw.go:553 strict <: not below the worst of the result heap
        const uint64_t lost = n_list == kCap ? entry_at(kCap - 1) : kDeadKey;  // falls off the end of the registers
        uint32_t p = 0;
#pragma unroll
        for (int s2 = 0; s2 < S; s2++)
            p += (uint32_t)__builtin_popcountll(__ballot(key[s2] < x));      // full (distance, node) order: equal distances sit in node order,
                                                                             // which is selectNeighbors' order (hnsw.go:589-594) when the build cuts the list
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
        if (n_list < kCap) n_list++;
        if (n_list > ef || lost != kDeadKey) {
            // the result heap holds ef entries: everything above its worst value w goes, the entries AT w (the tie group) stay
            const uint32_t w = (uint32_t)(entry_at(ef - 1) >> 32);
            uint32_t dropped = 0;
#pragma unroll
            for (int s2 = 0; s2 < S; s2++) {
                const uint32_t e = (uint32_t)s2 * 64 + lane;
                const bool drop = e >= ef && key[s2] != kDeadKey && (uint32_t)(key[s2] >> 32) > w;
                const uint64_t dm = __ballot(drop);
                if (drop) key[s2] = kDeadKey;
                expd[s2] &= ~dm;
                dropped += (uint32_t)__builtin_popcountll(dm);
            }
            n_list -= dropped;
            if (lost != kDeadKey && (uint32_t)(lost >> 32) == w) tie = true;   // a member of the tie group did not fit: which members the heap holds is unknown
        }
    };

    // A hop's admissions at once (latency form).  One by one, each admission shifts the S registers (~1.6 k cycles: 4 per hop were
    // the largest single part of a lone traversal's hop).  Write T(Z) = the entries of Z at or below its ef-th smallest distance
    // (Z itself while it has fewer than ef): an admission is L -> T(L + x) unless L is full and d(x) >= w(L), the ef-th smallest —
    // for d(x) > w(L) that IS T(L + x), and T(T(Z) + Y) = T(Z + Y), so the hop leaves T(L + X) unless some x met d(x) == w(L) and
    // was turned away (hnsw.go:553 is a strict <) where T would keep it.  Such an x has d(x) >= the final w'; above w' it is dropped
    // either way; so only a survivor AT the final w' (with more than ef entries in all) can make the two differ — then, as for a
    // NaN or a list that would outgrow its registers, the hop is admitted one by one as before (false; nothing has been touched).
    // Positions: an old entry moves up by the survivors below it, a survivor goes to (old entries below it) + (survivors below it) —
    // one compare per (survivor, register) serves both; the scatter goes through LDS (the row buffer is idle between hops).
    auto insert_batch = [&](uint64_t kx, uint64_t pend, uint32_t ef) -> bool {
        constexpr uint32_t kCap = (uint32_t)S * 64;
        const uint32_t ns = (uint32_t)__builtin_popcountll(pend);
        if (n_list + ns > kCap) return false;
        const bool mine = (pend >> lane) & 1ull;
        if (__ballot(mine && (uint32_t)(kx >> 32) == 0xFFFFFFFEu)) return false;
        uint32_t up[S];
#pragma unroll
        for (int s2 = 0; s2 < S; s2++) up[s2] = 0;
        uint32_t pos = 0;
        const uint32_t n_dead = kCap - n_list;                          // dead registers compare above every key
        for (uint64_t m = pend; m; m &= m - 1) {
            const uint32_t i = (uint32_t)__builtin_ctzll(m);
            const uint64_t xi = readlane64(kx, i);
            uint32_t above = 0;
#pragma unroll
            for (int s2 = 0; s2 < S; s2++) {
                const bool gt = key[s2] > xi;
                up[s2] += gt ? 1u : 0u;
                above += (uint32_t)__builtin_popcountll(__ballot(gt));
            }
            if (mine && kx > xi) pos++;
            if (lane == i) pos += n_list - (above - n_dead);
        }
        const uint32_t merged = n_list + ns;
        uint32_t w = 0;
        if (merged > ef) {
            uint64_t wk = 0;
#pragma unroll
            for (int s2 = 0; s2 < S; s2++) {
                const uint32_t e = (uint32_t)s2 * 64 + lane;
                const uint64_t bal = __ballot(e < n_list && e + up[s2] == ef - 1);
                if (bal) wk = readlane64(key[s2], (uint32_t)__builtin_ctzll(bal));
            }
            const uint64_t bal = __ballot(mine && pos == ef - 1);
            if (bal) wk = readlane64(kx, (uint32_t)__builtin_ctzll(bal));
            w = (uint32_t)(wk >> 32);
            if (__ballot(mine && (uint32_t)(kx >> 32) == w)) return false;
        }
        typedef __attribute__((address_space(3))) uint64_t* lds_u64w;
        typedef __attribute__((address_space(3))) uint32_t* lds_f;
        lds_u64w kb = (lds_u64w)L.rows; lds_f fb = (lds_f)(L.rows + kCap * 8);
#pragma unroll
        for (int s2 = 0; s2 < S; s2++) {
            const uint32_t e = (uint32_t)s2 * 64 + lane;
            if (e < n_list) { kb[e + up[s2]] = key[s2]; fb[e + up[s2]] = (uint32_t)((expd[s2] >> lane) & 1ull); }
        }
        if (mine) { kb[pos] = kx; fb[pos] = 0; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        uint32_t dropped = 0;
#pragma unroll
        for (int s2 = 0; s2 < S; s2++) {
            const uint32_t e = (uint32_t)s2 * 64 + lane;
            uint64_t kk = kDeadKey; uint32_t f = 0;
            if (e < merged) { kk = kb[e]; f = fb[e]; }
            const bool drop = merged > ef && e >= ef && e < merged && (uint32_t)(kk >> 32) > w;   // above the ef-th smallest: out (the entries AT it stay)
            if (drop) kk = kDeadKey;
            key[s2] = kk;
            expd[s2] = __ballot(f != 0 && !drop);
            dropped += (uint32_t)__builtin_popcountll(__ballot(drop));
        }
        n_list = merged - dropped;
        return true;
    };

    // the slot's next query: the call's counter (HnswOpts::next) when there is one, else a fixed stride
    auto next_query = [&](uint32_t qi) -> uint32_t {
        if (!o.next) return qi + gridDim.x;
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(o.next, 1u);
        return gridDim.x + (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    };
    for (uint32_t qi = blockIdx.x; qi < nq; qi = next_query(qi)) {
        q_g = qblk + (size_t)qi * v.dim4 * 4;
        qc.qn = qconst[(size_t)qi * 2]; qc.qn32 = (float)qconst[(size_t)qi * 2 + 1];
        n_eval = 0; tie = false;
        if (front) {                                                    // the query's float32 words into LDS, once per traversal
            const uint32_t qbytes = v.dim * 4u;
            const float* q32 = o.q32 + (size_t)qi * v.dim;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the previous traversal's reads of it are done)
            for (uint32_t off = 0; off < qbytes; off += 1024)
                if (off + lane * 16 < qbytes) glds16(q32 + (off >> 2) + lane * 4, (lds_u8*)qres_l + off);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (W > 1) {                                          // the query into LDS (the other waves are at their barrier)
            const uint32_t qbytes = v.dim4 * 4 * (uint32_t)sizeof(Q);
            for (uint32_t off = 0; off < qbytes; off += 1024)
                if (off + lane * 16 < qbytes) glds16(reinterpret_cast<const float*>(q_g) + (off >> 2) + lane * 4, L.q + off);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        uint32_t entry = g.entry;
        int stop = 0;                                                   // build: connectNode stops at min(level, graphLevel), hnsw.go:383
        if (build) { stop = (int)o.qlevel[qi]; if (stop > g.cur_level) stop = g.cur_level; }
        // Search (hnsw.go:649-664): ef = 1 on the upper levels, max(efSearch, k) on level 0.  One loop body serves every
        // level and the entry-point evaluation (a "hop" whose only neighbour is the entry), so each piece of the
        // traversal is instantiated once.
        for (int level = g.cur_level; level >= stop && !tie; level--) {
            const uint32_t ef = level > stop ? 1u : (build ? ef_search : (ef_search > k ? ef_search : k));
            // searchLayer (hnsw.go:471-580)
            const uint32_t hcap = level > stop && o.vis_cap > kVisGreedyCap ? kVisGreedyCap : o.vis_cap;
            const uint32_t hmask = hcap - 1, hshift = (uint32_t)__builtin_clz(hcap) + 1;   // 32 - log2(hcap)
            const uint32_t hlimit = hcap - (hcap >> 2);
            uint32_t n_vis = 1;
            // level 0 of a graph without tombstones and at most 32 links per node: the hop's front (slab 0 beside the visited test,
            // visited set in buckets — one scheme per searchLayer: its table starts empty)
            const bool bucketed = front && level == 0 && !g.has_dead && g.max_m0 <= 32u;
            const uint32_t bmask = (hcap >> 4) - 1, bshift = hshift + 4;
            if (vis_lds) vis_hash_clear_lds(tab_l, hcap, lane); else vis_hash_clear(tab, hcap, lane);   // :483-488
#pragma unroll
            for (int s2 = 0; s2 < S; s2++) { key[s2] = kDeadKey; expd[s2] = 0; }
            n_list = 0;
            bool first = true;
            // latency form: the adjacency list of the entry most likely to be popped next, requested a hop ahead (see below)
            uint32_t spec = 0xFFFFFFFFu, spec_deg = 0, spec_cl = 0xFFFFFFFFu, spec_hl = 0xFFFFu;
            for (;;) {
                uint32_t nb;
                bool fronted = false, ffresh = false, fhub = false; uint32_t fc = 0xFFFFFFFFu, fold = kVisEmpty, fpos = 0, nrow = 0, frank = 0; uint32_t* fword = nullptr;
                uint64_t fmask = 0; float fdtab = 0.0f;
                if (first) {                                                 // :492-506: the entry point itself
                    first = false;
                    wsync();
                    if (lane == 0) {
                        if (vis_lds) (void)vis_hash_insert_lds(tab_l, hmask, hshift, entry);
                        else if (bucketed) (void)atomicCAS(tab + (size_t)((entry * 0x9E3779B1u) >> bshift) * 16, kVisEmpty, entry);   // (word 0 of its bucket: the table is empty)
                        else (void)vis_hash_insert(tab, hmask, hshift, entry);
                        batch[0] = entry;
                    }
                    wsync();
                    nb = 1;
                } else {
                    // pop: first unexpanded entry
                    uint32_t cur = 0xFFFFFFFFu, cur_d = 0;
#pragma unroll
                    for (int s2 = 0; s2 < S; s2++) {
                        if (cur != 0xFFFFFFFFu) continue;
                        const uint64_t m = __ballot(key[s2] != kDeadKey) & ~expd[s2];
                        if (m) { const uint32_t l = (uint32_t)__builtin_ctzll(m); const uint64_t kc = readlane64(key[s2], l); cur = (uint32_t)kc; cur_d = (uint32_t)(kc >> 32); expd[s2] |= 1ull << l; }
                    }
                    if (cur == 0xFFFFFFFFu) break;
                    {   // candidates.pop() among equal minima is the heap's choice: another unexpanded entry at the popped distance
                        uint64_t same = 0;
#pragma unroll
                        for (int s2 = 0; s2 < S; s2++) same |= __ballot((uint32_t)(key[s2] >> 32) == cur_d && key[s2] != kDeadKey) & ~expd[s2];
                        if (same) { tie = true; break; }
                    }
                    HTICK(0);
                    if (bucketed) {
                        if (cur >= g.n_nodes) continue;
                        uint32_t fdeg, cl, hl;
                        if (kSpec && cur == spec) { fdeg = spec_deg; cl = spec_cl; hl = spec_hl; }
                        else {
                            fdeg = g.l0_deg[cur]; cl = lane < g.max_m0 ? g.l0_links[(size_t)cur * g.max_m0 + lane] : 0xFFFFFFFFu;
                            hl = (o.l0_hub && lane < g.max_m0) ? (uint32_t)o.l0_hub[(size_t)cur * g.max_m0 + lane] : 0xFFFFu;
                        }
                        if (fdeg > 32u) fdeg = 32u;
                        fc = lane < fdeg ? cl : 0xFFFFFFFFu;
                        bool valid = fc < g.n_nodes;
                        if ((repeats_in_list(fc, fdeg) >> lane) & 1ull) valid = false;   // (a repeated node counts at its first occurrence, see below)
                        const uint64_t vm = __ballot(valid);
                        if (!vm) continue;
                        // a HUB's distance comes from the call's hub table (HnswOpts::hub_tab, see k_hub_table): no row is read for it
                        fhub = valid && hl < o.hub_H;
                        const uint64_t vrow = __ballot(valid && !fhub);       // links whose rows are read
                        if (vrow) {
                            const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)fc, (int)__builtin_ctzll(vrow));
                            wfence();                                        // the previous hop's reads of batch[] / pos[] are done
                            if (lane < 32u) batch[32 + lane] = (valid && !fhub) ? fc : c0;   // every adjacency position names a row that exists
                            wfence();
                            HTICK(2);
                            DmaRole r0;
                            dma_role(r0, v.rowmaj, v.dim, batch_l + 32, fdeg, lane);
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef QV_HNSW_P512
                            dma_issue_slab<kHnswSlab>(r0, 0, v.dim4, slabs_l);       // slab 0 of every such link, beside the visited test
#endif
                        }
                        float dtab = 0.0f;
                        if (fhub) dtab = o.hub_tab[(size_t)qi * o.hub_H + hl];   // (asked for before it is known to be new: it lands with the test)
                        ffresh = vis_bucket_test(tab, bmask, bshift, fc, valid, fword);
                        HTICK(3);
                        fmask = __ballot(ffresh);
                        nb = (uint32_t)__builtin_popcountll(fmask);
                        n_vis += nb;
                        if (n_vis > hlimit) { tie = true; break; }
                        if (nb == 0) continue;                               // (slab 0 has landed with the test's answer: nothing left in flight)
                        // Claiming the free words WITHOUT atomics: the table is this wave slot's alone, so the only contenders for a word are the
                        // hop's own lanes — those whose nodes fall into one bucket all saw the same first free word.  They take consecutive
                        // words in lane order (a bucket fills from word 0 up, so the words behind the first free one are free too); a lane
                        // that would run past its bucket's end is left to vis_bucket_settle (test again + atomicCAS, after the rows).
                        // 39 M returning atomics per 8192 traversals cost 4.5 % of the call (profiles/r06_hnsw_front.txt).
                        fold = kVisEmpty;
#ifdef QV_HNSW_CAS_CLAIMS
                        if (ffresh) fold = atomicCAS(fword, kVisEmpty, fc);  // (measurement build: the claim as a returning atomic, as first built)
#else
                        {
                            const uint32_t bid = (uint32_t)(reinterpret_cast<uintptr_t>(fword) >> 6);
                            uint32_t brank = 0;
                            for (uint64_t mm = fmask; mm; mm &= mm - 1) {
                                const uint32_t j = (uint32_t)__builtin_ctzll(mm);
                                const uint32_t bj = (uint32_t)__builtin_amdgcn_readlane((int)bid, (int)j);
                                brank += (bid == bj && lane > j) ? 1u : 0u;
                            }
                            if (ffresh) {
                                const uint32_t fe = (uint32_t)(reinterpret_cast<uintptr_t>(fword) >> 2) & 15u;
                                if (fe + brank < 16u) __builtin_nontemporal_store(fc, fword + brank);
                                else fold = 0u;                              // (any value but "empty": settled later)
                            }
                        }
#endif
                        if (o.hist && ffresh) atomicAdd(&o.hist[fc], 1u);    // (the sampling pass that chooses the hubs)
                        const uint64_t frow = fmask & vrow;                  // new AND read from its row: compacted for slabs 1..
                        nrow = (uint32_t)__builtin_popcountll(frow);
                        frank = (uint32_t)__builtin_popcountll(frow & ((1ull << lane) - 1));
                        if ((frow >> lane) & 1ull) { batch[frank] = fc; pos_l[frank] = lane; }
                        wfence();
                        fpos = lane < nrow ? pos_l[lane] : 0u;
                        fdtab = dtab;
                        fronted = true;
                        HTICK(1);
                    } else {
                    uint32_t deg = 0; const uint32_t* links = nullptr;
                    uint32_t c = 0xFFFFFFFFu; bool fresh = false;
                    if (level == 0 && !g.has_dead && cur < g.n_nodes) {
                        // level 0 of a graph without tombstones (every graph built on the device): the degree and the fixed-width list
                        // are requested together — one round trip instead of two dependent ones per hop
                        uint32_t cl;
                        if (kSpec && cur == spec) { deg = spec_deg; cl = spec_cl; }
                        else { deg = g.l0_deg[cur]; cl = lane < g.max_m0 ? g.l0_links[(size_t)cur * g.max_m0 + lane] : 0xFFFFFFFFu; }
                        if (lane < deg) { c = cl; fresh = c < g.n_nodes; }
#ifdef QV_HNSW_PROF
                        asm volatile("" : "+v"(c));
                        HTICK(2);
#endif
                    } else {
                        if (alive(cur) && (level == 0 ? (!g.has_dead || g.level[cur] >= 0) : level <= (int)g.level[cur])) {
                            if (level == 0) { deg = g.l0_deg[cur]; links = g.l0_links + (size_t)cur * g.max_m0; }
                            else { const uint32_t* blk = g.up_links + (size_t)(g.up_off[cur] + (uint32_t)(level - 1)) * (1 + g.max_m); deg = blk[0]; links = blk + 1; }
                        }
                        if (lane < deg) { c = links[lane]; fresh = alive(c); }
                    }
                    // a node repeated inside one list (the self-link quirk) is new at its FIRST occurrence: admissions of a hop go
                    // in adjacency order (:537-560), and between two nodes of equal distance that order decides which one a full
                    // result set keeps — so the hash's test-and-set must not pick the winner among a node's repeats
                    if ((repeats_in_list(c, deg) >> lane) & 1ull) fresh = false;
                    HTICK(6);
                    if (fresh) fresh = vis_lds ? vis_hash_insert_lds(tab_l, hmask, hshift, c) : vis_hash_insert(tab, hmask, hshift, c);
                    HTICK(3);
                    const uint64_t fm = __ballot(fresh);
                    nb = (uint32_t)__builtin_popcountll(fm);
                    n_vis += nb;
                    if (n_vis > hlimit) { tie = true; break; }               // table 3/4 full: hand the query to the exact-heap kernel
                    wsync();                                         // previous hop's batch[] reads are done
                    if (fresh) batch[__builtin_popcountll(fm & ((1ull << lane) - 1))] = c;
                    wsync();
                    HTICK(1);
                    if (nb == 0) continue;
                    }
                }
                if (kSpec) {
                    // The next pop is the first unexpanded entry — the one the list shows NOW unless this hop admits something
                    // closer.  Its adjacency list is requested before the hop's rows are evaluated and has arrived long before
                    // the pop: a hop's two dependent round trips (list, then rows) become one on most hops.
                    spec = 0xFFFFFFFFu;
                    if (level == 0 && !g.has_dead) {
#pragma unroll
                        for (int s2 = 0; s2 < S; s2++) {
                            if (spec != 0xFFFFFFFFu) continue;
                            const uint64_t m = __ballot(key[s2] != kDeadKey) & ~expd[s2];
                            if (m) spec = (uint32_t)readlane64(key[s2], (uint32_t)__builtin_ctzll(m));
                        }
                        if (spec < g.n_nodes) {
                            spec_deg = g.l0_deg[spec]; spec_cl = lane < g.max_m0 ? g.l0_links[(size_t)spec * g.max_m0 + lane] : 0xFFFFFFFFu;
                            spec_hl = (o.l0_hub && lane < g.max_m0) ? (uint32_t)o.l0_hub[(size_t)spec * g.max_m0 + lane] : 0xFFFFu;
                        }
                        else spec = 0xFFFFFFFFu;
                    }
                }
                uint64_t kx;
                uint64_t have;                                               // the lanes of kx that hold a neighbour (adjacency order = lane order)
                if (fronted) {
                    float dd = 0.0f;
                    if (nrow) {
#ifdef QV_HNSW_P512
                        if constexpr (M == QV_COSINE || M == QV_DOT) dd = hnsw_eval_hop_p512<M>(v, batch_l, slabs_l, qc, nrow, lane, qres_l);
                        else dd = hnsw_eval_hop_front<M>(v, batch_l, slabs_l, qc, nrow, fpos, lane, qres_l);
#else
                        dd = hnsw_eval_hop_front<M>(v, batch_l, slabs_l, qc, nrow, fpos, lane, qres_l);
#endif
                    }
                    // back to the adjacency positions: the lane of link p takes its row's distance from compacted lane frank(p), or the hub table's
                    const float drow = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute((int)(frank << 2), (int)__float_as_uint(dd)));
                    kx = ffresh ? make_key(fhub ? fdtab : drow, fc) : kDeadKey;
                    have = fmask;
                    vis_bucket_settle(tab, bmask, bshift, fc, ffresh, fold);
                } else { kx = eval_keys(nb); have = __ballot(lane < nb); }
                n_eval += nb;
                HTICK(7);
                {   // admissions in adjacency order (:553-560).  The worst value only ever decreases, so a neighbour that is not below
                    // it now never will be: one ballot drops those up front (most of a hop once the list is full)
                    uint64_t pend = have;
                    if (n_list >= ef) {
                        const uint32_t w = (uint32_t)(entry_at(ef - 1) >> 32);
                        pend = have & __ballot((uint32_t)(kx >> 32) < w);
#ifdef QV_HNSW_PROF
                        full_hops++; full_rows += nb; surv_hops += pend != 0; surv_rows += (uint64_t)__builtin_popcountll(pend);
#endif
                    }
                    bool done = false;
                    if constexpr (W > 1) { if (pend) done = insert_batch(kx, pend, ef); }   // (measured in the wave-per-query form as well, through the slab buffers: no difference at full occupancy)
                    while (!done && pend) {
                        const uint32_t i = (uint32_t)__builtin_ctzll(pend);
                        pend &= pend - 1;
                        insert(readlane64(kx, i), ef);
                    }
                }
                HTICK(4);
                if (tie) break;                                              // flagged: the exact-heap kernel redoes it from scratch
#ifdef QV_HNSW_PROF
                hops++;
#endif
            }
            if (level > stop && n_list > 0) entry = (uint32_t)readlane64(key[0], 0);   // :649-657
        }
        const uint32_t kq = build ? (stop == 0 ? g.max_m0 : g.max_m) : k;   // build: the level's degree bound (:395-398)
        const uint32_t ef_last = build ? ef_search : (ef_search > k ? ef_search : k);
        const uint32_t n_res = n_list < ef_last ? n_list : ef_last;     // size of the result heap (the list may also hold the rest of a tie group)
        uint32_t cnt = n_res < kq ? n_res : kq;                         // :670-672
        if (!tie && n_list > ef_last) {
            // a tie group straddles the end of the result heap: which of its members the heap holds is the heap's choice.
            // Harmless unless the output reaches into the group.
            const uint32_t w = (uint32_t)(entry_at(ef_last - 1) >> 32);
            uint32_t below = 0;
#pragma unroll
            for (int s2 = 0; s2 < S; s2++) below += (uint32_t)__builtin_popcountll(__ballot((uint32_t)(key[s2] >> 32) < w));
            if (cnt > below) tie = true;
        }
        if (!build && !tie) {
            // results[:k] comes out of the max-heap in pop order (:566-577): among equal distances that order — and, at the cut,
            // which of them is kept — is the heap's.  (The build re-sorts by (distance, node), selectNeighbors :589-594.)
#pragma unroll
            for (int s2 = 0; s2 < S; s2++) {
                const uint32_t e = (uint32_t)s2 * 64 + lane;
                uint64_t nxt = __shfl_down(key[s2], 1);
                if (lane == 63) nxt = s2 + 1 < S ? readlane64(key[s2 + 1 < S ? s2 + 1 : s2], 0) : kDeadKey;
                if (__ballot(e < cnt && e + 1 < n_list && (uint32_t)(key[s2] >> 32) == (uint32_t)(nxt >> 32))) tie = true;
            }
        }
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
            if (build && stop >= 1 && o.self_dist) {                    // lower levels link the node to itself (:463-467): d(node, node)
                wsync();
                if (lane == 0) batch[0] = o.qnode0 + qi;
                wsync();
                const uint64_t ks = eval_keys(1);
                if (lane == 0) o.self_dist[qi] = unord_f32((uint32_t)(ks >> 32));
            }
        }
        if (lane == 0) { count_out[qi] = cnt; if (evals_out) evals_out[qi] = n_eval; }
#ifdef QV_HNSW_PROF
        if (lane == 0 && (blockIdx.x == 3 || blockIdx.x == 777) && qi >= nq - gridDim.x)
        {
            if (W > 1) printf("lat eval: issue %llu dma-wait %llu chain %llu shuffle+write %llu | post+barrier %llu barrier B %llu certify %llu\n", T[8], T[9], T[10], T[11], T[12], T[13], T[14]);
            printf("blk %u: pop %llu links+vis %llu dma-wait %llu issue %llu compute %llu insert %llu other %llu/%llu hops %llu evals(last q) %u wall(10ns) %llu cyc %llu | list-full hops %llu rows %llu, with a row below the worst: hops %llu rows %llu\n", blockIdx.x, T[0], T[1], T[2], T[6], T[3], T[4], T[5], T[7], hops, n_eval,
                   (unsigned long long)(wall_clock64() - wc0), (unsigned long long)(T[0]+T[1]+T[2]+T[3]+T[4]+T[5]+T[6]+T[7]), full_hops, full_rows, surv_hops, surv_rows);
        }
#endif
    }
    if constexpr (W > 1) { if (lane == 0) L.ctrl[0] = 0xFFFFFFFFu; lat_barrier(); }      // the other waves leave
}

// ---- link distances of an uploaded graph ---------------------------------------------------
// A graph that was built on the host arrives without the per-link distances the device-side construction works on
// (qv_build.hip).  One wavefront per adjacency list of nodes [n0, n0 + nq): the node's own vector is the query (converted by
// k_hnsw_prep_queries like any other), lane i scores link i with the same arithmetic as a traversal hop —
// computeDistance(node.Vector, conn.Vector), what the reference's prune would compute (hnsw.go:438).
template <int M, int U>
__global__ void __launch_bounds__(64)
k_graph_link_dists(IndexView v, GraphView g, const typename MT<M>::Q* __restrict__ qblk, const double* __restrict__ qconst, uint32_t n0, uint32_t nq,
                   float* __restrict__ l0_dist, float* __restrict__ up_dist) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* batch = reinterpret_cast<uint32_t*>(smem);
    lds_u32* batch_l = (lds_u32*)smem;
    lds_u8* slabs_l = (lds_u8*)(batch_l + 64);
    const uint32_t lane = threadIdx.x;
    for (uint32_t qi = blockIdx.x; qi < nq; qi += gridDim.x) {
        const uint32_t node = n0 + qi;
        const int lv = (int)g.level[node];
        if (lv < 0) continue;                                               // tombstone: its lists are never read
        const typename MT<M>::Q* q_g = qblk + (size_t)qi * v.dim4 * 4;
        QConst qc; qc.qn = qconst[(size_t)qi * 2]; qc.qn32 = (float)qconst[(size_t)qi * 2 + 1];
        for (int level = 0; level <= lv; level++) {
            uint32_t deg; const uint32_t* links; float* out;
            if (level == 0) { deg = g.l0_deg[node]; links = g.l0_links + (size_t)node * g.max_m0; out = l0_dist + (size_t)node * g.max_m0; if (deg > g.max_m0) deg = g.max_m0; }
            else {
                const uint32_t blk = g.up_off[node] + (uint32_t)(level - 1);
                const uint32_t* b = g.up_links + (size_t)blk * (1 + g.max_m);
                deg = b[0] < g.max_m ? b[0] : g.max_m; links = b + 1; out = up_dist + (size_t)blk * g.max_m;
            }
            if (deg == 0) continue;
            __syncthreads();
            if (lane < deg) { const uint32_t c = links[lane]; batch[lane] = c < g.n_nodes ? c : node; }
            __syncthreads();
            const float d = hnsw_eval_rows<M, U>(v, batch_l, slabs_l, q_g, qc, deg, lane);
            if (lane < deg) out[lane] = d;
        }
    }
}

// ---- HNSW traversal ------------------------------------------------------------------
// the wave kernel keeps no query in LDS: queries are pre-converted to the metric's Q type (zero-padded to dim4*4)
// in global memory and read at wave-uniform addresses, i.e. by scalar loads into SGPR operands of v_fma_f64
template <int M>
__global__ void k_hnsw_prep_queries(const float* __restrict__ queries, uint32_t dim, uint32_t dim4, typename MT<M>::Q* __restrict__ qblk, double* __restrict__ qconst,
                                    uint32_t* __restrict__ next = nullptr) {
    if (next && blockIdx.x == 0 && threadIdx.x == 0) *next = 0;     // the traversal's query counter (HnswOpts::next)
    using Q = typename MT<M>::Q;
    const float* q = queries + (size_t)blockIdx.x * dim;
    Q* out = qblk + (size_t)blockIdx.x * dim4 * 4;
    for (uint32_t i = threadIdx.x; i < dim4 * 4; i += blockDim.x) out[i] = i < dim ? (Q)q[i] : (Q)0;
    if (threadIdx.x == 0) {
        QConst c = query_const<M>(q, dim);                // same element order as everywhere else (distances.go:20)
        if constexpr (M == QV_DOT) {                      // not part of the distance: the bound of the latency form's certificate (split_bound)
            double ma = 0.0;
            for (uint32_t i = 0; i < dim; i++) { const double a = (double)q[i]; ma = __builtin_fma(a, a, ma); }
            c.qn = __builtin_sqrt(ma);
        }
        qconst[(size_t)blockIdx.x * 2] = c.qn; qconst[(size_t)blockIdx.x * 2 + 1] = (double)c.qn32;
    }
}
size_t hnsw_qblk_bytes(uint32_t nq, uint32_t dim4) { return (size_t)nq * dim4 * 4 * sizeof(double) + (size_t)nq * 2 * sizeof(double) + 64; }   // + the query counter
// candidate min-heap slots of the exact-heap kernel: admissions of one searchLayer grow like ef * (1 + ln(visited / ef))
static uint32_t hnsw_cand_cap(uint32_t ef) { return ef > 256 ? 2 * (uint32_t)kHnswCandCap : (uint32_t)kHnswCandCap; }
size_t hnsw_lds_bytes(uint32_t ef) {
    return 2 * (size_t)slab_bytes<kHnswHeapSlab>() + (size_t)(hnsw_cand_cap(ef) + kHnswEfMax + 1) * sizeof(HRes) + (size_t)kHnswMaxDeg * 8 + 64;
}
uint32_t hnsw_grid(int cus, uint32_t ef, uint32_t nq) {
    const size_t lds = hnsw_lds_bytes(ef);
    uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(8, (size_t)(160 * 1024) / lds));
    return std::max(1u, std::min(nq, (uint32_t)cus * per_cu));
}
// visited hash entries per wave slot: ~64 x ef (a search with ef = 128 evaluates ~4-5 k nodes of a 1M-node graph), 3/4 usable
uint32_t hnsw_vis_hash_cap(uint32_t ef) {
    static const int mult = dev_env_int("QV_HNSW_VIS_MULT", 64);
    uint32_t cap = 4096;
    while (cap < (uint32_t)mult * ef && cap < 262144u) cap <<= 1;
    return cap;
}
hipError_t launch_hnsw_search(const IndexView& v, const GraphView& g, const float* d_queries, void* d_qblk, uint32_t nq, uint32_t k, uint32_t ef,
                              const HnswOpts& o, uint32_t grid, bool prep, uint32_t* d_rows_out, float* d_dist_out,
                              uint32_t* d_count_out, uint32_t* d_evals_out, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    if (k == 0 || k > (uint32_t)kHnswEfMax || ef > (uint32_t)kHnswEfMax || g.max_m0 > (uint32_t)kHnswMaxDeg || g.max_m > (uint32_t)kHnswMaxDeg) return hipErrorInvalidValue;
    if (!o.vis || (size_t)o.vis_cap * 32 < g.n_nodes) return hipErrorInvalidValue;
    const uint32_t efx = o.qlevel ? ef : std::max(ef, k);
    const size_t lds = hnsw_lds_bytes(efx);
    const uint32_t cand_cap = hnsw_cand_cap(efx);
    hipError_t e = hipSuccess;
    double* d_qconst = reinterpret_cast<double*>(static_cast<unsigned char*>(d_qblk) + (size_t)nq * v.dim4 * 4 * sizeof(double));
    QV_DISPATCH_METRIC(v.metric, {
        if (prep) hipLaunchKernelGGL((k_hnsw_prep_queries<MM>), dim3(nq), dim3(64), 0, s, d_queries, v.dim, v.dim4, static_cast<typename MT<MM>::Q*>(d_qblk), d_qconst);
        bool lat = false;
        if constexpr (SplitOK<MM>::value) {
            // the latency form (eight waves per query, the hop's rows requested at once and evaluated as certified partial chains):
            // this kernel only ever runs a handful of queries, each a chain of ~200 hops (QV_HNSW_LAT=2: never)
            static const int lat_env = env_int("QV_HNSW_LAT", 1);
            const size_t heaps = (size_t)(cand_cap + kHnswEfMax + 1) * sizeof(HRes) + (size_t)kHnswMaxDeg * 4 + 64;
            HnswOpts ol = o;
            size_t lds_lat = 0;
            for (ol.lat_rows = 32; ol.lat_rows >= 8; ol.lat_rows >>= 1) {      // the hop's rows at once; half / a quarter of them where the dimension asks for it
                lds_lat = (size_t)kLatQOff + lat_q_bytes(v.dim4, (uint32_t)sizeof(typename MT<MM>::Q)) + lat_rows_bytes(v.dim4, ol.lat_rows) + heaps;
                if (lds_lat <= (size_t)160 * 1024) break;
            }
            if (lat_env == 1 && hnsw_qlds_ok(v) && ol.lat_rows >= 8) {
                lat = true;
                e = set_lds(k_hnsw_search<MM, 16, kLatWaves>, lds_lat);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL((k_hnsw_search<MM, 16, kLatWaves>), dim3(grid), dim3(64 * kLatWaves), lds_lat, s, v, g, static_cast<const typename MT<MM>::Q*>(d_qblk),
                                   static_cast<const double*>(d_qconst), nq, k, ef, ol, cand_cap, d_rows_out, d_dist_out, d_count_out, d_evals_out);
            }
        }
        if (!lat) {
            e = set_lds(k_hnsw_search<MM, 16>, lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((k_hnsw_search<MM, 16>), dim3(grid), dim3(64), lds, s, v, g, static_cast<const typename MT<MM>::Q*>(d_qblk),
                               static_cast<const double*>(d_qconst), nq, k, ef, o, cand_cap, d_rows_out, d_dist_out, d_count_out, d_evals_out);
        }
    });
    return hipGetLastError();
}

// wave-resident form: the list in registers, the LDS for the row slabs; tie-flagged queries report kHnswTieFlag
// What the wave-per-query form does beyond round 5's (bits of HnswOpts::front; QV_HNSW_FRONT, default 3: all of it; 0: as before):
//   1  the hop's front — slab 0 of every link requested beside the visited test, the visited set in buckets of 16 words, the query
//      resident in LDS and broadcast by v_readlane
//   2  the adjacency list of the likely next pop requested a hop ahead
// Measured on the 1M x 768 graph, efSearch 128 (profiles/r06_hnsw_front.txt); the pieces were sized with tools/ubench/gather_mix.hip.
static uint32_t hnsw_front_bits() { static const int b = dev_env_int("QV_HNSW_FRONT", 3); return (uint32_t)b & 3u; }
size_t hnsw_wave_lds_bytes(int /*metric*/, uint32_t dim4) {
    const bool qres = (hnsw_front_bits() & 1u) && (dim4 & 7u) == 0 && dim4 >= 8;      // (a row-major index is the launcher's other condition; without one the room stays unused)
    return (size_t)kHnswWaveFixedLds + (qres ? (size_t)dim4 * 16 : 0);
}
uint32_t hnsw_wave_grid(int cus, int metric, uint32_t dim4) {
    const size_t lds = hnsw_wave_lds_bytes(metric, dim4);
    uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(16, (size_t)(160 * 1024) / lds));
    // Resident traversals per CU.  The gather saturates the memory system well below 16: measured 8 / 10 / 12 / 14 / 16 per CU on the
    // 1M x 768 graph, efSearch 128, 32768 queries per call: 354 k / 370 k / 339 k / 341 k / 331 k queries/s (round 5's form; the bare
    // row stream of tools/ubench/gather_mix.hip: 6.6 TB/s at 8 and 12 per CU, 5.8 at 16) — and every one fewer is 4 KiB of rows less in
    // flight competing for the same DRAM pages.  (With the query resident a wave takes 12 KiB of LDS at 768 dimensions: 13 fit.)
    static const int cap = env_int("QV_HNSW_WAVES_PER_CU", 12);
    per_cu = std::min<uint32_t>(per_cu, (uint32_t)cap);
    return (uint32_t)cus * per_cu;
}
hipError_t launch_hnsw_search_wave(const IndexView& v, const GraphView& g, const float* d_queries, void* d_qblk, uint32_t nq, uint32_t k, uint32_t ef,
                                   const HnswOpts& o, uint32_t grid, uint32_t* d_rows_out, float* d_dist_out,
                                   uint32_t* d_count_out, uint32_t* d_evals_out, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    const uint32_t efx = o.qlevel ? std::max(ef, 64u) : (ef > k ? ef : k);       // build: the list must also hold the k = MaxM0 outputs
    if (k == 0 || efx > (uint32_t)kHnswEfMax || k > (uint32_t)kHnswEfMax || g.max_m0 > (uint32_t)kHnswMaxDeg || g.max_m > (uint32_t)kHnswMaxDeg) return hipErrorInvalidValue;
    if (!o.vis || o.vis_cap < 1024 || (o.vis_cap & (o.vis_cap - 1))) return hipErrorInvalidValue;
    const size_t lds = hnsw_wave_lds_bytes(v.metric, v.dim4);
    hipError_t e = hipSuccess;
    double* d_qconst = reinterpret_cast<double*>(static_cast<unsigned char*>(d_qblk) + (size_t)nq * v.dim4 * 4 * sizeof(double));
    // Queries are handed out through a counter: a batch of 8192 is two traversals per wave slot, their lengths differ (evaluations per
    // query: 5th / 95th percentile 0.8 / 1.25 of the median at efSearch 128), and with a fixed stride of queries per slot the call lasts
    // as long as its unluckiest pair while the other slots idle (QV_HNSW_DYN=0: fixed stride, as before round 6)
    static const int dyn_env = dev_env_int("QV_HNSW_DYN", 1);
    uint32_t* d_next = dyn_env == 1 ? reinterpret_cast<uint32_t*>(d_qconst + (size_t)nq * 2) : nullptr;
    HnswOpts on = o; on.next = d_next;
    on.front = hnsw_qlds_ok(v) ? hnsw_front_bits() : 0u;
    on.q32 = d_queries;
    QV_DISPATCH_METRIC(v.metric, {
        hipLaunchKernelGGL((k_hnsw_prep_queries<MM>), dim3(nq), dim3(64), 0, s, d_queries, v.dim, v.dim4, static_cast<typename MT<MM>::Q*>(d_qblk), d_qconst, d_next);
    });
#define QV_HWD(SS, DD) QV_DISPATCH_METRIC(v.metric, {                                                                \
        e = set_lds(k_hnsw_search_wave<MM, 4, SS, DD>, lds);                                                          \
        if (e != hipSuccess) return e;                                                                                \
        hipLaunchKernelGGL((k_hnsw_search_wave<MM, 4, SS, DD>), dim3(grid), dim3(64), lds, s, v, g, static_cast<const typename MT<MM>::Q*>(d_qblk), \
                           static_cast<const double*>(d_qconst), nq, k, ef, on,                                      \
                           d_rows_out, d_dist_out, d_count_out, d_evals_out);                                         \
    })
    // the query through LDS for row-major indexes whose dimension is a multiple of 32 (QV_HNSW_QLDS=2: never)
    static const int qlds_env = dev_env_int("QV_HNSW_QLDS", 1);
    const bool deep = qlds_env == 1 && hnsw_qlds_ok(v);
    // The latency form (a workgroup of four waves and a CU's LDS per query) for batches that would leave most CUs idle anyway:
    // at most one query per CU, a metric whose chain can be split and certified, rows of a hop + query within the LDS.
    static const int lat_env = env_int("QV_HNSW_LAT", 1);                    // (QV_HNSW_LAT=2: never)
    static const int lat_cus = [] { int d = 0, c = 0; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d); return c; }();
    const uint32_t qsize = (v.metric == QV_COSINE || v.metric == QV_DOT || v.metric == QV_L2SQ_F64) ? 8u : 4u;
    HnswOpts ol = on; ol.front = 0;
    size_t lat_fixed = 0;
    // two tiers: up to one query per CU — 32 rows of a hop at once, the visited table in LDS, a CU per query; up to three per CU
    // (QV_HNSW_LAT_TIER2, default 768 queries) — 16 rows at once and the visited table in global memory, so that two workgroups share a CU:
    // one call of 384 / 512 / 768 / 1024 queries 6.75 / 7.13 / 7.47 / 7.89 ms a wave per query, 4.75 / 5.17 / 6.94 / 7.52 ms in this form (with the
    // exact-heap pass of their flagged queries; 8192 queries: 319 k QPS a wave per query, 264 k in this form, 183 k one workgroup per CU)
    static const int lat_tier2 = dev_env_int("QV_HNSW_LAT_TIER2", 768);
    const bool tier2 = nq > (uint32_t)lat_cus;
    for (ol.lat_rows = tier2 ? 16u : 32u; ol.lat_rows >= 8; ol.lat_rows >>= 1) {   // the hop's rows at once; half / a quarter of them where the dimension asks for it
        lat_fixed = (size_t)kLatQOff + lat_q_bytes(v.dim4, qsize) + lat_rows_bytes(v.dim4, ol.lat_rows);
        if (lat_fixed <= (size_t)(tier2 ? 80 : 160) * 1024) break;
    }
    const bool lat_metric = v.metric == QV_COSINE || v.metric == QV_DOT || v.metric == QV_L2 || v.metric == QV_L1 || v.metric == QV_L2SQ_F64;
    const uint32_t lat_nq_cap = std::max<uint32_t>((uint32_t)lat_cus, (uint32_t)lat_tier2);
    if (lat_env == 1 && lat_metric && hnsw_qlds_ok(v) && nq <= lat_nq_cap && ol.lat_rows >= 8) {
        ol.vis_lds = !tier2 && lat_fixed + (size_t)o.vis_cap * 4 <= (size_t)160 * 1024 ? 1u : 0u;
        const size_t lds_lat = lat_fixed + (ol.vis_lds ? (size_t)o.vis_cap * 4 : 0);
        const uint32_t lat_grid = std::min<uint32_t>(nq, std::min<uint32_t>(grid, (uint32_t)lat_cus * (tier2 ? 2u : 1u)));
#define QV_HWL(SS) QV_DISPATCH_METRIC(v.metric, {                                                                     \
        if constexpr (SplitOK<MM>::value) {                                                                           \
            e = set_lds(k_hnsw_search_wave<MM, 4, SS, true, kLatWaves>, lds_lat);                                             \
            if (e != hipSuccess) return e;                                                                            \
            hipLaunchKernelGGL((k_hnsw_search_wave<MM, 4, SS, true, kLatWaves>), dim3(lat_grid), dim3(64 * kLatWaves), lds_lat, s, v, g, static_cast<const typename MT<MM>::Q*>(d_qblk), \
                               static_cast<const double*>(d_qconst), nq, k, ef, ol,                                  \
                               d_rows_out, d_dist_out, d_count_out, d_evals_out);                                     \
        }                                                                                                             \
    })
        if (efx < 128) { QV_HWL(2); } else if (efx < 256) { QV_HWL(4); } else if (efx < 320) { QV_HWL(5); } else if (efx < 512) { QV_HWL(8); } else { QV_HWL(9); }
#undef QV_HWL
        return e != hipSuccess ? e : hipGetLastError();
    }
#define QV_HW(SS) if (deep) { QV_HWD(SS, true); } else { QV_HWD(SS, false); }
    // list registers: S x 64 entries.  One notch more than efx needs where that is free (<= 128 VGPRs either way), so that a tie
    // group at the end of the result heap has room (ef <= 127 -> S = 2, ef = 128..256 -> S = 4; S = 8 would cost a wave per SIMD)
    if (efx < 128) { QV_HW(2); } else if (efx < 256) { QV_HW(4); } else if (efx < 320) { QV_HW(5); } else if (efx < 512) { QV_HW(8); } else { QV_HW(9); }
#undef QV_HW
#undef QV_HWD
    return hipGetLastError();
}


// ---- hubs ---------------------------------------------------------------------------------------------------------------------
// MEASUREMENT BUILD ONLY (make VARIANTS=1, QV_HNSW_HUBS=1): built in round 6, measured, not shipped.  The idea: rows that most traversals
// of a batch read are not gathered at all —
//   * a sampling pass (the call's first 1024 queries, HnswOpts::hist) counts reads per row; rows read by more than one sampled query in
//     sixteen become hubs — a copy of them in the tile layout (k_hub_gather) and, beside every level-0 adjacency list, the hub slot of each
//     link (k_hub_links), kept with the graph until it changes;
//   * per call, k_hub_table computes distance(query, hub) for EVERY query and hub as a dense pass — mq_tile: the multi-query scan's loop,
//     one lane per row, eight queries per pass, the same chain in the same order — at the f64 vector rate (~30 TFLOP/s) instead of the
//     gather's;
//   * a hop looks a hub's distance up (4 bytes) and reads rows only for the rest.
// Results cannot change (a table entry is the distance the hop would have computed, bit for bit; a hub is still "visited", admitted and
// counted: tests/test_gpu_graph.py::test_a_large_call_equals_the_latency_form_the_oracle_and_sees_updates passes with it on) — but the
// premise does not hold on the corpora of the benchmark: on the 1M x 768 graphs (i.i.d. rows and the 16-dimensional subspace alike) the
// 8192 most-read rows take 6-7 % of a batch's reads and only ~50 rows are read by more than one query in twelve (QV_TRACE=1 prints the
// distribution of the sampling pass), so the table would cost twice what it saves and the selection rule declines.  It stays here for
// corpora with real hubs; the product library neither samples nor builds it (hubs_possible, qv_graph_api.cpp).
#ifdef QV_VARIANTS
template <int M, int QB>
__global__ void __launch_bounds__(256, 2)
k_hub_table(const float* __restrict__ hub_tiles, const double* __restrict__ hub_rnorm, uint32_t n_hub_tiles, uint32_t dim, uint32_t dim4,
            const float* __restrict__ queries, const double* __restrict__ qconst, uint32_t nq, float* __restrict__ table) {
    using Q = typename MT<M>::Q;
    using A = typename MT<M>::A;
    extern __shared__ __align__(16) unsigned char smem[];
    Q* ql = reinterpret_cast<Q*>(smem);                                          // [dim4*4][QB]
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t q0 = blockIdx.y * QB;
    for (uint32_t i = threadIdx.x; i < dim4 * 4 * QB; i += blockDim.x) {
        const uint32_t d = i / QB, qq = i % QB;
        const uint32_t qi = q0 + qq < nq ? q0 + qq : nq - 1;
        ql[i] = d < dim ? (Q)queries[(size_t)qi * dim + d] : (Q)0;
    }
    __syncthreads();
    QConst qc[QB];
#pragma unroll
    for (int j = 0; j < QB; j++) { const uint32_t qi = q0 + j < nq ? q0 + j : nq - 1; qc[j].qn = qconst[(size_t)qi * 2]; qc[j].qn32 = (float)qconst[(size_t)qi * 2 + 1]; }
    const f4* tiles = reinterpret_cast<const f4*>(hub_tiles);
    const size_t H = (size_t)n_hub_tiles * 64;
    for (uint32_t t = blockIdx.x * 4 + wave; t < n_hub_tiles; t += gridDim.x * 4) {
        A acc[QB], qa[QB];
        mq_tile<M, 4, QB, false>(tiles + (size_t)t * dim4 * 64 + lane, ql, dim4, acc, qa);
        double rn = 0.0;
        if constexpr (MT<M>::needs_rnorm) rn = hub_rnorm[(size_t)t * 64 + lane];
#pragma unroll
        for (int j = 0; j < QB; j++)
            if (q0 + j < nq) table[(size_t)(q0 + j) * H + (size_t)t * 64 + lane] = finalize<M>(acc[j], qc[j], rn);
    }
}
// hub h's row (tile layout) into slot h of the hub tiles; one block per hub, one thread per 16-byte chunk
__global__ void k_hub_gather(IndexView v, const uint32_t* __restrict__ hub_rows, uint32_t H, float* __restrict__ hub_tiles, double* __restrict__ hub_rnorm) {
    const uint32_t h = blockIdx.x;
    if (h >= H) return;
    const uint32_t row = hub_rows[h];
    const f4* src = reinterpret_cast<const f4*>(v.tiles) + (size_t)(row >> 6) * v.dim4 * 64 + (row & 63u);
    f4* dst = reinterpret_cast<f4*>(hub_tiles) + (size_t)(h >> 6) * v.dim4 * 64 + (h & 63u);
    for (uint32_t c = threadIdx.x; c < v.dim4; c += blockDim.x) dst[(size_t)c * 64] = src[(size_t)c * 64];
    if (threadIdx.x == 0) hub_rnorm[h] = v.rnorm ? v.rnorm[row] : 0.0;
}
__global__ void k_hub_slots(const uint32_t* __restrict__ hub_rows, uint32_t H, uint16_t* __restrict__ slot_of) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h < H) slot_of[hub_rows[h]] = (uint16_t)h;
}
__global__ void k_hub_links(const uint32_t* __restrict__ l0_links, const uint32_t* __restrict__ l0_deg, uint32_t n_nodes, uint32_t max_m0,
                            const uint16_t* __restrict__ slot_of, uint16_t* __restrict__ l0_hub) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_nodes * max_m0) return;
    const uint32_t node = (uint32_t)(i / max_m0), j = (uint32_t)(i % max_m0);
    const uint32_t c = j < l0_deg[node] ? l0_links[i] : 0xFFFFFFFFu;
    l0_hub[i] = c < n_nodes ? slot_of[c] : (uint16_t)0xFFFFu;
}

#endif
// the hubs' copies and the link -> hub slot map (d_slot_of: [n_nodes] scratch)
hipError_t launch_hub_build(const IndexView& v, const GraphView& g, const uint32_t* d_hub_rows, uint32_t H, float* d_hub_tiles, double* d_hub_rnorm,
                            uint16_t* d_slot_of, uint16_t* d_l0_hub, hipStream_t s) {
#ifndef QV_VARIANTS
    (void)v; (void)g; (void)d_hub_rows; (void)H; (void)d_hub_tiles; (void)d_hub_rnorm; (void)d_slot_of; (void)d_l0_hub; (void)s;
    return hipErrorNotSupported;
#else
    if (H == 0 || (H & 63u) || H > 65535u - 63u) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(d_slot_of, 0xFF, (size_t)g.n_nodes * sizeof(uint16_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_hub_gather, dim3(H), dim3(256), 0, s, v, d_hub_rows, H, d_hub_tiles, d_hub_rnorm);
    hipLaunchKernelGGL(k_hub_slots, dim3((H + 255) / 256), dim3(256), 0, s, d_hub_rows, H, d_slot_of);
    const size_t n = (size_t)g.n_nodes * g.max_m0;
    hipLaunchKernelGGL(k_hub_links, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, g.l0_links, g.l0_deg, g.n_nodes, g.max_m0, d_slot_of, d_l0_hub);
    return hipGetLastError();
#endif
}
// table[q][h] = distance(query q, hub h) for the call's nq queries; d_qblk as for launch_hnsw_search_wave (the queries are converted here
// for their constants; the traversal's launcher converts them again: 20 us)
hipError_t launch_hub_table(const IndexView& v, const float* d_hub_tiles, const double* d_hub_rnorm, uint32_t H, const float* d_queries, void* d_qblk, uint32_t nq,
                            float* d_table, int cus, hipStream_t s) {
#ifndef QV_VARIANTS
    (void)v; (void)d_hub_tiles; (void)d_hub_rnorm; (void)H; (void)d_queries; (void)d_qblk; (void)nq; (void)d_table; (void)cus; (void)s;
    return hipErrorNotSupported;
#else
    if (nq == 0 || H == 0) return hipSuccess;
    double* d_qconst = reinterpret_cast<double*>(static_cast<unsigned char*>(d_qblk) + (size_t)nq * v.dim4 * 4 * sizeof(double));
    hipError_t e = hipSuccess;
    constexpr int QB = 8;
    const uint32_t n_tiles = H / 64, groups = (nq + QB - 1) / QB;
    const uint32_t gx = std::max(1u, std::min((n_tiles + 3) / 4, std::max(1u, (uint32_t)cus * 2u / std::max(1u, std::min(groups, (uint32_t)cus * 2u)))));
    QV_DISPATCH_METRIC(v.metric, {
        using Q = typename MT<MM>::Q;
        hipLaunchKernelGGL((k_hnsw_prep_queries<MM>), dim3(nq), dim3(64), 0, s, d_queries, v.dim, v.dim4, static_cast<Q*>(d_qblk), d_qconst, (uint32_t*)nullptr);
        const size_t lds = (size_t)v.dim4 * 4 * QB * sizeof(Q);
        e = set_lds((k_hub_table<MM, QB>), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_hub_table<MM, QB>), dim3(gx, groups), dim3(256), lds, s, d_hub_tiles, d_hub_rnorm, n_tiles, v.dim, v.dim4, d_queries,
                           static_cast<const double*>(d_qconst), nq, d_table);
    });
    return hipGetLastError();
#endif
}

// distances of every link of nodes [n0, n0 + nq) (see k_graph_link_dists); d_qblk: hnsw_qblk_bytes(nq, dim4)
hipError_t launch_graph_link_dists(const IndexView& v, const GraphView& g, void* d_qblk, uint32_t n0, uint32_t nq, float* d_l0_dist, float* d_up_dist,
                                   uint32_t grid, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    if (!v.rowmaj || g.max_m0 > (uint32_t)kHnswMaxDeg || g.max_m > (uint32_t)kHnswMaxDeg) return hipErrorInvalidValue;
    const size_t lds = hnsw_wave_lds_bytes(v.metric, v.dim4);
    double* d_qconst = reinterpret_cast<double*>(static_cast<unsigned char*>(d_qblk) + (size_t)nq * v.dim4 * 4 * sizeof(double));
    QV_DISPATCH_METRIC(v.metric, {
        hipLaunchKernelGGL((k_hnsw_prep_queries<MM>), dim3(nq), dim3(64), 0, s, v.rowmaj + (size_t)n0 * v.dim, v.dim, v.dim4, static_cast<typename MT<MM>::Q*>(d_qblk), d_qconst);
        hipLaunchKernelGGL((k_graph_link_dists<MM, 4>), dim3(std::min(grid, nq)), dim3(64), lds, s, v, g, static_cast<const typename MT<MM>::Q*>(d_qblk),
                           static_cast<const double*>(d_qconst), n0, nq, d_l0_dist, d_up_dist);
    });
    return hipGetLastError();
}

}  // namespace qv
