// qv_build.hip — device-resident HNSW construction: the link phase of a batch of Inserts
// (shared helpers: qv_kernels.h; the search phase is qv_hnsw.hip's traversal kernels in build mode)
//
// What it replaces: the loop of hnsw.HNSW.Insert / connectNode over a batch of new nodes (pkg/hnsw/hnsw.go:266-468).
// The reference releases its lock before connectNode (hnsw.go:313-315), i.e. concurrent Inserts — each searching a graph in
// which the others are not linked yet — are its own contract.  A batch here is the deterministic member of that family
// (the test suite holds a CPU restatement of exactly these semantics, with the reference's own re-scoring prune):
//   search  every batch node runs connectNode's searches (:367-385) against the graph as it was before the batch;
//   link    forward links (:404-409), self-links below the connected level (:463-467) and back-links with the prune
//           (:413-460) are applied as if node by node in index order.
// A batch of one node is exactly Insert.
//
// Why no distance is evaluated in the link phase: the prune re-scores a neighbour's whole list,
// computeDistance(neighbor.Vector, conn.Vector) (hnsw.go:438).  Every metric here is bitwise symmetric in its arguments
// (products, squared / absolute differences and the norm product commute; the accumulation order over dimensions is the
// same), so that value is the distance the search phase computed when the link was made.  Each link therefore carries its
// distance (l0_dist / up_dist, 4 B per link), a back-link brings the distance its own search measured, and the prune is
// pure integer work on 64-bit (distance, node) keys — selectNeighbors' order (hnsw.go:589-594) is the key order.
// tests/test_gpu_build.py checks the built graph against that CPU restatement, which re-scores like the reference.
//
// Pipeline of one batch of B nodes (all on one stream, no host synchronisation):
//   k_hnsw_search_wave<build> -> k_build_compact_redo -> k_hnsw_search<build> (tie-flagged queries only)
//   -> k_build_link: forward + self links of the new nodes; one 64-bit key (target list << 32 | edge) per back-link
//   -> stable radix sort of the keys by target list (qv_rank.hip)  -> k_build_heads: one segment per target list
//   -> k_build_merge: one wavefront per target list replays the appends / prunes of its incoming links in node order.
#include "qv_kernels.h"

namespace qv {

constexpr uint32_t kBuildTie = 0xFFFFFFFEu;

__global__ void k_build_compact_redo(const uint32_t* __restrict__ count, uint32_t n, uint32_t* __restrict__ redo_idx, uint32_t* __restrict__ redo_n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && count[i] == kBuildTie) { redo_idx[atomicAdd(redo_n, 1u)] = i; atomicAdd(redo_n + 3, 1u); }   // [3]: running total
}

// the list a back-link at `level` to node nb goes into: level 0 -> nb, level l >= 1 -> cap_nodes + block(nb, l)
__device__ __forceinline__ uint32_t list_id(const BuildView& b, uint32_t nb, int level) {
    return level == 0 ? nb : b.cap_nodes + b.up_off[nb] + (uint32_t)(level - 1);
}

// one wavefront per new node: its own lists + the keys of its back-links
__global__ void __launch_bounds__(64)
k_build_link(BuildView b, uint32_t first, uint32_t n, int cur_level, const uint32_t* __restrict__ rows, const float* __restrict__ dist,
             const uint32_t* __restrict__ count, const float* __restrict__ self_dist, uint64_t* __restrict__ keys, uint32_t* __restrict__ status) {
    const uint32_t xi = blockIdx.x, lane = threadIdx.x;
    if (xi >= n) return;
    const uint32_t x = first + xi;
    const int lv = (int)b.level[x];
    const int stop = lv < cur_level ? lv : cur_level;                    // hnsw.go:383 min(level, graphLevel)
    uint32_t cnt = count[xi];
    if (cnt > 64u) { if (lane == 0) atomicOr(status, cnt == kBuildTie ? 2u : 1u); cnt = 0; }   // unresolved tie flag / candidate-heap overflow
    const uint32_t nb = lane < cnt ? rows[(size_t)xi * b.max_m0 + lane] : 0xFFFFFFFFu;
    const float d = lane < cnt ? dist[(size_t)xi * b.max_m0 + lane] : 0.0f;
    if (stop == 0) {                                                     // forward links, :404-409
        if (lane < cnt) { b.l0_links[(size_t)x * b.max_m0 + lane] = nb; b.l0_dist[(size_t)x * b.max_m0 + lane] = d; }
        if (lane == 0) b.l0_deg[x] = cnt;
    } else {
        const uint32_t blk = b.up_off[x] + (uint32_t)(stop - 1);
        if (lane < cnt) { b.up_links[(size_t)blk * (1 + b.max_m) + 1 + lane] = nb; b.up_dist[(size_t)blk * b.max_m + lane] = d; }
        if (lane == 0) b.up_links[(size_t)blk * (1 + b.max_m)] = cnt;
        // below the connected level the search re-enters from the node itself and finds only it (:463-467): the node links
        // to itself (:407) and gets itself back as a back-link (:426)
        const float sd = self_dist[xi];
        if (cnt > 0) {
            for (int l = (int)lane; l < stop; l += 64) {
                if (l == 0) {
                    b.l0_links[(size_t)x * b.max_m0] = x; b.l0_links[(size_t)x * b.max_m0 + 1] = x;
                    b.l0_dist[(size_t)x * b.max_m0] = sd; b.l0_dist[(size_t)x * b.max_m0 + 1] = sd;
                    b.l0_deg[x] = 2;
                } else {
                    const uint32_t bl = b.up_off[x] + (uint32_t)(l - 1);
                    b.up_links[(size_t)bl * (1 + b.max_m)] = 2; b.up_links[(size_t)bl * (1 + b.max_m) + 1] = x; b.up_links[(size_t)bl * (1 + b.max_m) + 2] = x;
                    b.up_dist[(size_t)bl * b.max_m] = sd; b.up_dist[(size_t)bl * b.max_m + 1] = sd;
                }
            }
        }
    }
    if (lane < b.max_m0)                                                 // back-links, :413-426: (target list, edge) keys, dead beyond cnt
        keys[(size_t)xi * b.max_m0 + lane] = lane < cnt ? ((uint64_t)list_id(b, nb, stop) << 32) | (uint64_t)(xi * b.max_m0 + lane) : kDeadKey;
}

__global__ void k_build_heads(const uint64_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ seg_start, uint32_t* __restrict__ seg_n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = keys[i];
    if (k == kDeadKey) return;
    if (i == 0 || (uint32_t)(keys[i - 1] >> 32) != (uint32_t)(k >> 32)) seg_start[atomicAdd(seg_n, 1u)] = i;
}

// one wavefront per target list: replay "append; if over the bound, drop dead links, sort by (distance, node), keep the
// bound" (hnsw.go:426-457) for its incoming links in node order.  The list lives in the lanes as 64-bit keys.
__global__ void __launch_bounds__(64)
k_build_merge(BuildView b, const uint64_t* __restrict__ keys, uint32_t n_keys, const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ seg_n,
              uint32_t first, const float* __restrict__ dist) {
    const uint32_t lane = threadIdx.x;
    const uint32_t ns = *seg_n;
    for (uint32_t s = blockIdx.x; s < ns; s += gridDim.x) {
        const uint32_t i0 = seg_start[s];
        const uint32_t lid = (uint32_t)(keys[i0] >> 32);
        uint32_t* degp; uint32_t* linkp; float* distp; uint32_t cap;
        if (lid < b.cap_nodes) { degp = b.l0_deg + lid; linkp = b.l0_links + (size_t)lid * b.max_m0; distp = b.l0_dist + (size_t)lid * b.max_m0; cap = b.max_m0; }
        else { const uint32_t blk = lid - b.cap_nodes; degp = b.up_links + (size_t)blk * (1 + b.max_m); linkp = degp + 1; distp = b.up_dist + (size_t)blk * b.max_m; cap = b.max_m; }
        uint32_t len = *degp;
        if (len > cap) len = cap;
        uint64_t ent = lane < len ? make_key(distp[lane], linkp[lane]) : kDeadKey;
        for (uint32_t e = i0; e < n_keys; e++) {
            const uint64_t ke = keys[e];
            if ((uint32_t)(ke >> 32) != lid) break;
            const uint32_t eidx = (uint32_t)ke;
            const uint64_t nk = make_key(dist[eidx], first + eidx / b.max_m0);   // the distance its own search measured (symmetric)
            if (len < cap) { if (lane == len) ent = nk; len++; }                 // :426
            else {                                                               // :429-457
                const uint32_t node = (uint32_t)ent;
                const bool live = lane < len && node < b.cap_nodes && b.level[node] >= 0;   // :433-436 nil nodes are dropped
                uint64_t kk = wave_sort64(live ? ent : kDeadKey, lane);
                const uint64_t worst = readlane64(kk, cap - 1);
                if (nk < worst) { if (lane == cap - 1) kk = nk; kk = wave_sort64(kk, lane); }
                ent = kk;
                len = (uint32_t)__builtin_popcountll(__ballot(kk != kDeadKey));
            }
        }
        if (lane < len) { linkp[lane] = (uint32_t)ent; distp[lane] = unord_f32((uint32_t)(ent >> 32)); }
        if (lane == 0) *degp = len;
    }
}

hipError_t launch_build_links(const BuildView& b, uint32_t first, uint32_t n, int cur_level, const uint32_t* d_rows, const float* d_dist,
                              const uint32_t* d_count, const float* d_self, uint64_t* d_keys_a, uint64_t* d_keys_b, uint32_t* d_hist,
                              uint32_t* d_seg_start, uint32_t* d_counters /*[0]=redo_n [1]=seg_n [2]=status*/, uint32_t merge_grid, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint32_t nk = n * b.max_m0;
    hipLaunchKernelGGL(k_build_link, dim3(n), dim3(64), 0, s, b, first, n, cur_level, d_rows, d_dist, d_count, d_self, d_keys_a, d_counters + 2);
    uint64_t* sorted = nullptr;
    hipError_t e = launch_radix_sort_hi32(d_keys_a, d_keys_b, nk, d_hist, &sorted, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_build_heads, dim3((nk + 255) / 256), dim3(256), 0, s, sorted, nk, d_seg_start, d_counters + 1);
    hipLaunchKernelGGL(k_build_merge, dim3(std::min(merge_grid, nk)), dim3(64), 0, s, b, sorted, nk, d_seg_start, d_counters + 1, first, d_dist);
    return hipGetLastError();
}

hipError_t launch_build_compact_redo(const uint32_t* d_count, uint32_t n, uint32_t* d_redo_idx, uint32_t* d_redo_n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_build_compact_redo, dim3((n + 255) / 256), dim3(256), 0, s, d_count, n, d_redo_idx, d_redo_n);
    return hipGetLastError();
}

}  // namespace qv
