// qv_device.h — internal interface between the C-ABI layer (qv_api.cpp) and the
// gfx950 kernels (qv_scan/qv_rank/qv_batched/qv_hnsw/qv_misc.hip; shared helpers in qv_kernels.h).  Not installed; include/qv.h is the public ABI.
//
// HBM layout of an index ("row tiles", the SoA layout of DESIGN.md §3):
//   tiles   float  [n_tiles][dim4][64][4]   tile t holds rows 64t..64t+63; within a
//                                            tile, 16-byte chunk c (dims 4c..4c+3) of
//                                            row r sits at ((t*dim4 + c)*64 + r)*16 B.
//                                            One wave-wide dwordx4 load = one contiguous
//                                            KiB; lane == row, so a lane accumulates its
//                                            row's distance over dims 0..D-1 in exactly
//                                            the reference's element order.
//   rnorm   double [n_tiles*64]              per-row sqrt(sum b_i^2) in the metric's
//                                            precision (cosine metrics only)
//   alive   u64    [n_tiles]                 bit r of word t = row 64t+r is live
//   rres    float  [n_tiles*64]              per-row |r - bf16(r)|, rounded up: what the one-term bfloat16 filter loses of a row
//                                            (qv_batched.hip: its error bound is per row and per query, not a worst case)
//   rowmaj  float  [rows][dim]               optional row-major copy (QV_FLAG_ROWMAJOR)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qv {

constexpr int kTileRows = 64;          // one wavefront
constexpr int kMaxFusedK = 64;         // wave-resident top-k list: one key per lane
constexpr int kMaxWideK = 128;         // wave-resident list of 2 keys per lane (k_flat_scan_wide) + selection over the waves' lists
constexpr int kMaxBatchedK = 4096;     // filter + re-score batches: beyond 64 results per query their selections are radix selections (qv_batched.hip)
constexpr int kMaxSelectK = 8192;      // one key per row + radix select (qv_select.hip): above it, the full ranking (qv_rank.hip)
constexpr uint64_t kDeadKey = ~0ull;

struct IndexView {
    float*    tiles;
    double*   rnorm;
    uint64_t* alive;
    float*    rres;       // |r - bf16(r)| per row (k_row_residual, after every write of rows)
    float*    rowmaj;     // may be null
    uint16_t* bf16;       // may be null: bfloat16 copy of the rows, [tile][ceil(dim/16)][2 row blocks][2 halves][32 rows][8 values] (QV_FLAG_BF16_ROWS)
    uint32_t  dim;
    uint32_t  dim4;       // ceil(dim/4)
    uint32_t  n_rows;     // rows ever added
    uint32_t  n_tiles;    // ceil(n_rows/64)
    int       metric;
    int       filter;     // batched path's filter kernel: 0 = the library's rule, 1 fp32 MFMA chain, 2 bfloat16 x 3, 3 bfloat16 x 1 (qv_index_set_filter)
};

struct GraphView {
    const int8_t*   level;      // [n] node level, -1 = tombstone
    const uint32_t* l0_deg;     // [n]
    const uint32_t* l0_links;   // [n][max_m0]
    const uint32_t* up_off;     // [n] first block of the node's upper levels (level 1 -> block up_off[n])
    const uint32_t* up_links;   // blocks of (1 + max_m): degree, links
    uint32_t n_nodes, max_m0, max_m;
    uint32_t entry; int cur_level;
    bool has_dead;              // some level[] entry is -1 (a graph built on the device has none)
};

// Extras of the traversal kernels beyond a plain Search: the build's per-query stop level, the device-side redo list of
// the exact-heap pass, and the visited-set storage (qv_hnsw.hip).
struct HnswOpts {
    const int8_t*   qlevel    = nullptr;  // build: level of the node query i stands for; the query stops at
                                          // min(qlevel[i], cur_level) and searches THAT level with ef (hnsw.go:383-385)
    uint32_t        qnode0    = 0;        // build: node index of query 0 (query i is node qnode0 + i)
    float*          self_dist = nullptr;  // build: [nq] distance(node, node), written when the stop level is >= 1
    const uint32_t* redo_idx  = nullptr;  // heap kernel: the queries to run (null = all nq)
    const uint32_t* redo_n    = nullptr;  // heap kernel: how many of them (device word)
    uint32_t*       vis       = nullptr;  // visited storage: [slots][vis_cap] hash entries / [slots][vis_cap] bitmap words
    uint32_t        vis_cap   = 0;        // per slot: hash entries (power of two) / bitmap words
    uint32_t        vis_lds   = 0;        // set by launch_hnsw_search_wave for its latency form: the hash table is in LDS
    uint32_t        lat_rows  = 32;       // set by the launchers for the latency form: rows of a hop requested at once (32 / 16 / 8 by dimension)
    // the hubs (qv_graph_api.cpp graph_hubs): rows most traversals read are not read at all — their distances to every query of the call
    // are computed up front by a dense pass (k_hub_table) and looked up
    uint32_t*       hist      = nullptr;  // [n_nodes] += 1 per evaluated row (the sampling pass that chooses the hubs)
    const float*    hub_tab   = nullptr;  // [nq][hub_H] distance(query, hub)
    const uint16_t* l0_hub    = nullptr;  // [n_nodes][max_m0] the hub slot of every level-0 link (0xFFFF: not a hub)
    uint32_t        hub_H     = 0;        // hubs (0: none)
    const float*    q32       = nullptr;  // set by launch_hnsw_search_wave: the caller's queries as they came ([nq][dim] float32), what the front keeps in LDS
    uint32_t        front     = 0;        // set by launch_hnsw_search_wave for its wave-per-query form: which parts of the round-6 hop are on (qv_hnsw.hip)
    uint32_t*       next      = nullptr;  // set by launch_hnsw_search_wave: the call's query counter (zeroed by k_hnsw_prep_queries): a wave slot that
                                          // finishes its traversal takes the next unclaimed query instead of a fixed stride of them
};

// the mutable arrays of a graph under construction (qv_build.hip); every link carries its distance
struct BuildView {
    int8_t*   level;      // [cap_nodes]
    uint32_t* l0_deg;     // [cap_nodes]
    uint32_t* l0_links;   // [cap_nodes][max_m0]
    float*    l0_dist;    // [cap_nodes][max_m0]
    uint32_t* up_off;     // [cap_nodes]
    uint32_t* up_links;   // [blocks][1 + max_m]
    float*    up_dist;    // [blocks][max_m]
    uint32_t  cap_nodes, max_m0, max_m;
};

struct ScanPlan {
    uint32_t grid;        // workgroups
    uint32_t block;       // threads (multiple of 64)
    uint32_t n_lists;     // partial lists the scan writes (= grid)
    uint32_t cus;         // compute units of the device the plan was made for
};

// how many workgroups a scan over n_tiles uses on a device with `cus` CUs
ScanPlan plan_scan(uint32_t n_tiles, int cus);

// exact multi-query scan on the f64 matrix cores (qv_mq64.hip): 0 = not applicable, else 16-query blocks per pass
int mq64_blocks(int metric, uint32_t dim4, uint32_t nq);
size_t mq64_workspace_bytes(uint32_t nq, uint32_t dim4);
hipError_t launch_flat_scan_mq64(const IndexView& v, int cus, const float* d_queries, uint32_t nq, uint32_t k, void* d_qws, uint64_t* partial,
                                 uint32_t* grid_out, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);

// bytes of workspace for nq queries at list length k (partial lists)
size_t scan_workspace_bytes(const ScanPlan& p, uint32_t nq, uint32_t k);

// --- ingest ---------------------------------------------------------------------
// d_rows [n][dim] row-major on device -> tiles at rows [row0, row0+n); computes rnorm;
// sets alive bits; also fills rowmaj if present.
hipError_t launch_ingest(const IndexView& v, const float* d_rows, uint32_t row0, uint32_t n, hipStream_t s);
// synthetic rows straight into tile layout (the synthetic-corpus generator of DESIGN.md)
hipError_t launch_generate(const IndexView& v, uint64_t seed, uint64_t gen_row0, uint32_t row0, uint32_t n, hipStream_t s);
// tombstone / revive
hipError_t launch_set_alive(const IndexView& v, const uint32_t* d_rows, uint32_t n, int alive, hipStream_t s);
// tile layout -> row-major (one row)
hipError_t launch_fetch_row(const IndexView& v, uint32_t row, float* d_out, hipStream_t s);
// tile layout -> row-major, n listed rows -> d_out [n][dim]
hipError_t launch_fetch_rows(const IndexView& v, const uint32_t* d_rows, uint32_t n, float* d_out, hipStream_t s);

// --- search ---------------------------------------------------------------------
// Flat scan + fused top-k for k <= kMaxFusedK.  d_queries [nq][dim] float32 on device.
// d_ws: workspace of scan_workspace_bytes().  Writes d_rows_out/d_dist_out [nq][k]
// (padded with 0xFFFFFFFF / +inf).
// ev0/ev1 (optional): recorded immediately before/after the scan kernel on `s`.
// d_tickets (optional): 64 zeroed words that belong to the caller's stream alone — with them a single query is ONE launch (the last
// workgroup of the scan merges the lists); without, scan + k_merge_lists.
hipError_t launch_flat_topk(const IndexView& v, const ScanPlan& p, const float* d_queries, uint32_t nq, uint32_t k,
                            void* d_ws, uint32_t* d_rows_out, float* d_dist_out, hipStream_t s,
                            hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, uint32_t* d_tickets = nullptr,
                            uint32_t* done_flag = nullptr, uint32_t done_seq = 0, bool* flag_used = nullptr);   // done_flag: see launch_flat_small (k_flat_scan_split only)
bool flat_split_applies(const IndexView& v, uint32_t nq, uint32_t k);   // launch_flat_topk will take the tile-over-eight-waves form (given tickets)
// The exact scan (k <= kMaxFusedK) for exactly the queries whose d_flags word is non-zero — the ones a filter handed back —, listed and
// scanned on the device: nothing is read back.  Results replace rows / distances [q][k_stride] of those queries.  d_ws: redo_workspace_bytes.
size_t redo_workspace_bytes(const ScanPlan& p, uint32_t nq, uint32_t k);
hipError_t launch_flat_redo_flagged(const IndexView& v, const ScanPlan& p, const float* d_queries, uint32_t nq, uint32_t k, uint32_t k_stride, const uint32_t* d_flags,
                                    void* d_ws, uint32_t* d_rows_out, float* d_dist_out, hipStream_t s);
// Small collections (<= 256 tiles, <= 4 queries, k <= 16): scan + merge in ONE launch (the last workgroup to finish merges).
// d_ws: flat_small_workspace_bytes (partial lists); d_tickets: 64 zeroed words that belong to the caller's stream alone (the kernel
// leaves them zero).  done_flag (optional, device-visible host memory): receives done_seq after the results have been written.
bool flat_small_applies(const IndexView& v, uint32_t nq, uint32_t k);
size_t flat_small_workspace_bytes(uint32_t nq, uint32_t k);
hipError_t launch_flat_small(const IndexView& v, const float* d_queries, uint32_t nq, uint32_t k, void* d_ws, uint32_t* d_tickets, uint32_t* d_rows_out, float* d_dist_out,
                             uint32_t* done_flag, uint32_t done_seq, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
// merge n_lists lists of k (dist, row) pairs -> k best by (dist, row)
hipError_t launch_merge_shards(const uint32_t* d_packed, const uint32_t* d_bases, uint32_t n_lists, uint32_t nq, uint32_t k,
                               uint32_t* d_rows_out, float* d_dist_out, hipStream_t s, uint32_t planes = 0 /* 0: [nq][2][k] per shard; >= 2: [planes][nq][k] */);
// The same merge for lists of any length (k > kMaxFusedK: a filtered search asks every shard for its full ranking): the valid
// entries of the G sorted lists [planes][nq][kcap] of query q become 64-bit keys (distance, global row), one stable radix sort
// orders them, the first k_out are written (padded with 0xFFFFFFFF / +inf).  d_ws: merge_ranked_workspace_bytes(G * kcap).
size_t merge_ranked_workspace_bytes(uint64_t n_keys);
hipError_t launch_merge_ranked(const uint32_t* d_packed, const uint32_t* d_bases, uint32_t n_lists, uint32_t nq, uint32_t q, uint32_t kcap, uint32_t planes,
                               uint32_t k_out, void* d_ws, uint32_t* d_rows_out, float* d_dist_out, hipStream_t s);
// The merge for kMaxFusedK < k_out <= kMaxSelectK, every query of the batch in one go: keys of all gathered entries, then the radix
// selection of the kk = min(k_out, valid entries) smallest (qv_select.hip) — 6 launches for the batch instead of 18 per query.
size_t merge_select_workspace_bytes(uint32_t n_lists, uint32_t nq, uint32_t kcap, uint32_t k_out);
hipError_t launch_merge_select(const uint32_t* d_packed, const uint32_t* d_bases, uint32_t n_lists, uint32_t nq, uint32_t kcap, uint32_t planes,
                               uint32_t k_out, uint32_t kk, void* d_ws, uint32_t* d_rows_out, float* d_dist_out, hipStream_t s);
// payload lookup after a merge: out[i] = payload plane `plane` of the list entry that produced merged global row rows[i]
// (shard = the one whose base range holds the row; entry found by its local row), +inf for padding rows
hipError_t launch_lookup_payload(const uint32_t* d_packed, const uint32_t* d_bases, uint32_t n_lists, uint32_t nq, uint32_t q, uint32_t kcap, uint32_t planes,
                                 uint32_t plane, const uint32_t* d_rows, uint32_t n, float* d_out, hipStream_t s);
hipError_t launch_merge_pairs(const float* d_dist, const uint32_t* d_rows, uint32_t n_lists, uint32_t k,
                              uint32_t* d_rows_out, float* d_dist_out, hipStream_t s);

// Batched path: fp32-MFMA filter + exact re-scoring (results identical to launch_flat_topk).
// *d_overflow_out -> [nq] flags (device): 1 = candidate buffer overflowed, caller must redo that
// query with launch_flat_topk.
bool   batched_supported(const IndexView& v, uint32_t nq, uint32_t k);
size_t batched_workspace_bytes(const IndexView& v, const ScanPlan& p, uint32_t nq, uint32_t k);
hipError_t launch_batched(const IndexView& v, const ScanPlan& p, const float* d_queries, uint32_t nq, uint32_t k, void* d_ws,
                          uint32_t* d_rows_out, float* d_dist_out, uint32_t** d_overflow_out, int cus, hipStream_t s,
                          hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);

// the one-term bfloat16 filter with the query operands in registers (qv_qreg.hip): same arguments and candidates as the eight-wave kernels
bool qreg_filter_applies(const IndexView& v, uint32_t nq_pad, bool bfrows);
hipError_t launch_qreg_filter(const IndexView& v, const uint4* Qbf, const float* cq, const float* mq, uint32_t nq_pad, uint32_t* cand, float* cscore,
                              uint32_t* cnt, bool bfrows, int cus, hipStream_t s);

// Device-resident HNSW traversal (hnsw.go:471-713), one wave per query.
size_t   hnsw_lds_bytes(uint32_t ef);
uint32_t hnsw_grid(int cus, uint32_t ef, uint32_t nq);
// prep = also convert the queries into d_qblk (false: d_qblk already holds them, e.g. the redo pass of the same batch)
hipError_t launch_hnsw_search(const IndexView& v, const GraphView& g, const float* d_queries, void* d_qblk, uint32_t nq, uint32_t k, uint32_t ef,
                              const HnswOpts& o, uint32_t grid, bool prep, uint32_t* d_rows_out, float* d_dist_out,
                              uint32_t* d_count_out, uint32_t* d_evals_out, hipStream_t s);
uint32_t hnsw_vis_hash_cap(uint32_t ef);                 // hash entries per wave slot for a search with this ef

// wave-resident form (no LDS heaps); count_out = 0xFFFFFFFE for queries that met equal distances / NaN
uint32_t hnsw_wave_grid(int cus, int metric, uint32_t dim4);
hipError_t launch_hub_build(const IndexView& v, const GraphView& g, const uint32_t* d_hub_rows, uint32_t H, float* d_hub_tiles, double* d_hub_rnorm,
                            uint16_t* d_slot_of, uint16_t* d_l0_hub, hipStream_t s);
hipError_t launch_hub_table(const IndexView& v, const float* d_hub_tiles, const double* d_hub_rnorm, uint32_t H, const float* d_queries, void* d_qblk, uint32_t nq,
                            float* d_table, int cus, hipStream_t s);
size_t hnsw_qblk_bytes(uint32_t nq, uint32_t dim4);   // workspace for the converted queries + per-query constants
hipError_t launch_hnsw_search_wave(const IndexView& v, const GraphView& g, const float* d_queries, void* d_qblk, uint32_t nq, uint32_t k, uint32_t ef,
                                   const HnswOpts& o, uint32_t grid, uint32_t* d_rows_out, float* d_dist_out,
                                   uint32_t* d_count_out, uint32_t* d_evals_out, hipStream_t s);

// Selection path, kMaxFusedK < k <= kMaxSelectK (qv_select.hip): the kk smallest of each query's n keys (d_keys + q * stride; n and
// stride even), written as [nq][k_stride] rows / distances (padded).  ordered: among keys of equal distance, place in the array
// ascends with the low word (true for the keys of a scan, where place == row) — ties beyond the sort's capacity are then taken
// front to back; otherwise they are settled by three more radix windows on the low word.
// window0_counted: the kernel that made the keys also counted the selection's first window (select_prepare before it: qv_select.h)
struct SelState;
uint32_t select_cap(uint32_t kk);
size_t select_workspace_bytes(uint32_t nq, uint32_t kk);
hipError_t select_prepare(void* d_ws, uint32_t nq, uint32_t kk, SelState** st_out, uint32_t** hist_out, hipStream_t s);
hipError_t launch_select_topk(const uint64_t* d_keys, size_t stride, uint32_t n, uint32_t nq, uint32_t kk, uint32_t k_stride, void* d_ws,
                              uint32_t* d_rows_out, float* d_dist_out, hipStream_t s, bool window0_counted = false, bool ordered = true,
                              const uint32_t* d_active = nullptr /* device word: queries from *d_active on are left alone */);
// 64 < kk <= kMaxWideK: the scan with a list of 2 keys per lane (same stream as launch_flat_topk, no key per row written),
// then the selection over the waves' lists.  ev0 / ev1 as launch_flat_topk.
size_t flat_wide_workspace_bytes(const ScanPlan& p, uint32_t nq, uint32_t kk);
hipError_t launch_flat_wide(const IndexView& v, const ScanPlan& p, const float* d_queries, uint32_t nq, uint32_t kk, uint32_t k_stride,
                            void* d_ws, uint32_t* d_rows_out, float* d_dist_out, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
// keys [nq][n_tiles * 64] of nq >= 2 queries, QB (4 or 8) queries per corpus pass (k_flat_keys_mq); d_qws: flat_keys_mq_workspace_bytes
size_t flat_keys_mq_workspace_bytes(uint32_t nq, uint32_t dim4);
hipError_t launch_flat_keys_mq(const IndexView& v, const ScanPlan& p, const float* d_queries, uint32_t nq, uint64_t* d_keys, void* d_qws, hipStream_t s,
                               const uint32_t* d_active = nullptr /* device word: only the first *d_active queries are worked on */);
// scan + selection for nq queries at list length kk <= kMaxSelectK: every query's keys (one 8-byte key per row, k_flat_keys), then
// launch_select_topk; queries go in groups so that the keys of a group stay under flat_select_group_bytes
size_t flat_select_workspace_bytes(uint32_t n_tiles, uint32_t nq, uint32_t kk, uint32_t dim4);
hipError_t launch_flat_select(const IndexView& v, const ScanPlan& p, const float* d_queries, uint32_t nq, uint32_t kk, uint32_t k_stride,
                              void* d_ws, uint32_t* d_rows_out, float* d_dist_out, hipStream_t s);
// The same for the queries whose flags are set (64 < kk <= kMaxSelectK: the batched filter's hand-backs above the wave lists' range), decided
// and listed on the device: nothing is read back.  Results replace rows / distances [q][k_stride] of those queries.
size_t flat_select_redo_workspace_bytes(uint32_t n_tiles, uint32_t nq, uint32_t kk, uint32_t dim, uint32_t dim4);
hipError_t launch_flat_select_redo(const IndexView& v, const ScanPlan& p, const float* d_queries, uint32_t nq, uint32_t kk, uint32_t k_stride, const uint32_t* d_flags,
                                   void* d_ws, uint32_t* d_rows_out, float* d_dist_out, hipStream_t s);

// Full ranking path (any k): all distances -> 64-bit keys -> stable radix sort -> first k.
// d_keys_a/d_keys_b: two buffers of n_tiles*64 u64; d_hist: radix histogram workspace.
size_t  full_sort_workspace_bytes(uint32_t n_tiles);
hipError_t launch_flat_fullsort(const IndexView& v, const ScanPlan& p, const float* d_query, uint32_t k,
                                void* d_ws, uint32_t* d_rows_out, float* d_dist_out, hipStream_t s);

// per-link distances of nodes [n0, n0 + nq) of an uploaded graph (what the device-side construction needs to extend it)
hipError_t launch_graph_link_dists(const IndexView& v, const GraphView& g, void* d_qblk, uint32_t n0, uint32_t nq, float* d_l0_dist, float* d_up_dist,
                                   uint32_t grid, hipStream_t s);

// stable sort of n 64-bit keys by their upper 32 bits; hist: radix_hist_words(n) words; *sorted_out = a or b
size_t radix_hist_words(uint32_t n);
hipError_t launch_radix_sort_hi32(uint64_t* a, uint64_t* b, uint32_t n, uint32_t* hist, uint64_t** sorted_out, hipStream_t s);

// HNSW construction, link phase of one batch (qv_build.hip).  d_counters: [0] redo count, [1] segment count, [2] status bits
hipError_t launch_build_compact_redo(const uint32_t* d_count, uint32_t n, uint32_t* d_redo_idx, uint32_t* d_redo_n, hipStream_t s);
hipError_t launch_build_links(const BuildView& b, uint32_t first, uint32_t n, int cur_level, const uint32_t* d_rows, const float* d_dist,
                              const uint32_t* d_count, const float* d_self, uint64_t* d_keys_a, uint64_t* d_keys_b, uint32_t* d_hist,
                              uint32_t* d_seg_start, uint32_t* d_counters, uint32_t merge_grid, hipStream_t s);

// distance of one query to n listed rows (lane == listed row, sequential accumulation)
hipError_t launch_distance_rows(const IndexView& v, const float* d_query, const uint32_t* d_rows, uint32_t n,
                                float* d_dist_out, hipStream_t s);
// one pair on the HOST (qv_distance_pair): the kernels' own pair_distance<M> compiled for the CPU
float host_pair_distance(int metric, const float* a, const float* b, uint32_t dim);
// n independent pairs a[i], b[i] (row-major [n][dim])
hipError_t launch_distance_pairs(int metric, const float* d_a, const float* d_b, uint32_t n, uint32_t dim,
                                 float* d_out, hipStream_t s);

}  // namespace qv
