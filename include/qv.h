/*
 * qv.h — C ABI of libqv, the MI355X (gfx950) similarity-search hot path for Quiver.
 *
 * This is the drop-in boundary: the entry points below are exactly what a cgo
 * binding for the reference's flat-scan / neighbour-distance path would bind
 * (INTEGRATION.md shows the Go side).  No torch types, no C++ types: plain
 * pointers and sizes.  Each entry point cites the reference interface it
 * replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - every function returns QV_OK (0) or a negative qv_status; it never aborts.
 *     qv_last_error() returns a thread-local message whose wording follows the
 *     reference's error strings (pkg/hybrid/exact.go:45,49,101,105).
 *   - the device sees dense uint32 row numbers (the analogue of
 *     hnsw.Node.VectorIndex, pkg/hnsw/hnsw.go:94); string ids stay in the host
 *     language.
 *   - vectors are COPIED on add (copy-on-insert, pkg/hybrid/exact.go:53-56); no
 *     caller pointer is retained after a call returns (cgo pointer rule).
 *   - result ordering: distance ascending, ties by row ascending.  The reference
 *     leaves tie order unspecified (Go map iteration + unstable sort,
 *     pkg/hybrid/exact.go:115,124); this is a deterministic refinement of it.
 *   - qv_index_search*, qv_distance_rows* are thread-safe and may run
 *     concurrently (the reference runs Index.Search under a read lock,
 *     pkg/core/collection.go:647); add / remove / reserve / destroy need external
 *     exclusion (the reference holds c.Lock there, pkg/core/collection.go:139).
 *   - concurrent SMALL host-pointer searches on one handle (qv_index_search,
 *     qv_sharded_search, qv_graph_search) share device passes: one query per call
 *     from many threads is all the reference's host ever sends (collection.go:647;
 *     DB.BatchSearch's batch branch type-asserts the reference's own wrapper,
 *     pkg/core/db.go:726-727, and otherwise fans out single searches, :805-828).
 *     A call that finds the handle busy joins the calls that arrived during the
 *     running pass; together they are the next pass — one multi-query launch.  No
 *     timer: a lone caller runs at once, exactly as if there were no sharing.
 *     A scan of 256 MiB or more runs one pass at a time; a smaller index (where a
 *     single-query pass leaves most of the device idle) up to four side by side.
 *     Results are the same bits either way (every path is exact).
 *   - "_device" variants take device pointers and a hipStream_t (passed as
 *     void*), enqueue work and return without synchronising, so a caller can keep
 *     queries and results resident in HBM and time with HIP events.
 *   - rows are float32 at this seam.  arrowindex.Graph keeps float64 vectors
 *     (pkg/arrowindex/graph.go:136-140, :796-858); its only producer in the
 *     reference, ArrowHNSWIndex, widens float32 columns (index/arrow_hnsw.go:
 *     222-225), for which QV_L2SQ_F64 over float32 rows is lossless.  Genuine
 *     float64 input is narrowed on add: the host layer warns
 *     (quiver_amd/arrowindex.py), nothing fails silently.
 *   - platform: the library is built for MI355X hosts — Linux on x86-64.  The
 *     front that lets concurrent callers share passes waits on futex words and
 *     spins with the x86 PAUSE instruction (quiver_amd/csrc/qv_coalesce.h); the
 *     library never modifies the process environment (GPU_MAX_HW_QUEUES is the
 *     host's to set: INTEGRATION.md).
 */
#ifndef QV_H
#define QV_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QV_ABI_VERSION 4

typedef struct qv_index qv_index; /* opaque; owns device memory */

/* Distance metrics.  0..4 restate pkg/vectortypes/distances.go:12-104 (float64
 * accumulation, one rounding to float32; SquaredEuclidean is all-float32).
 * 5..7 restate pkg/hnsw/adapter.go:105-167 (float32 sequential accumulation),
 * the functions a reloaded collection uses (pkg/core/db.go:181-188). */
typedef enum qv_metric {
    QV_COSINE      = 0, /* vectortypes.CosineDistance            distances.go:12-40  */
    QV_L2          = 1, /* vectortypes.EuclideanDistance         distances.go:43-55  */
    QV_L2SQ        = 2, /* vectortypes.SquaredEuclideanDistance  distances.go:60-72  */
    QV_DOT         = 3, /* vectortypes.DotProductDistance        distances.go:77-90  */
    QV_L1          = 4, /* vectortypes.ManhattanDistance         distances.go:93-104 */
    QV_COSINE_F32  = 5, /* hnsw.CosineDistanceFunc               adapter.go:105-136  */
    QV_L2_F32      = 6, /* hnsw.EuclideanDistanceFunc            adapter.go:139-151  */
    QV_DOT_F32     = 7, /* hnsw.DotProductDistanceFunc           adapter.go:154-165  */
    QV_L2SQ_F64    = 8, /* ArrowHNSWIndex.Search re-score        index/arrow_hnsw.go:124-132
                           (float64 difference, float64 unfused square-accumulate)   */
    QV_METRIC_COUNT = 9
} qv_metric;

typedef enum qv_status {
    QV_OK                 =  0,
    QV_ERR_INVALID_ARG    = -1,
    QV_ERR_DIM_MISMATCH   = -2, /* "vector dimension mismatch" / "query dimension mismatch" */
    QV_ERR_K_NOT_POSITIVE = -3, /* "k must be positive" (exact.go:104-106; adapter.go:42-44) */
    QV_ERR_OUT_OF_RANGE   = -4, /* row id >= size */
    QV_ERR_NO_DEVICE      = -5, /* no HIP device / HIP runtime failure at create time */
    QV_ERR_DEVICE         = -6, /* HIP error during a call (message carries hipGetErrorString) */
    QV_ERR_OOM            = -7,
    QV_ERR_UNSUPPORTED    = -8
} qv_status;

/* flags for qv_index_create */
#define QV_FLAG_NONE        0ull
#define QV_FLAG_ROWMAJOR    1ull /* also keep a row-major copy: fast single-row gathers for
                                    qv_distance_rows / HNSW traversal (hnsw.go:536-563) */
#define QV_FLAG_BF16_ROWS   2ull /* also keep a bfloat16 copy of the rows (+50 % memory): the batched path's filter reads it
                                    instead of the float32 rows (half the bytes); results are unchanged — the filter only
                                    selects candidates for the exact re-score */

/* ---- lifecycle ------------------------------------------------------------------ */

/* Replaces hybrid.NewExactIndex(distFunc) (pkg/hybrid/exact.go:29-35) plus the
 * dimension lock-in of the first Insert (exact.go:43-47): dim is fixed here.
 * device = HIP device ordinal the index lives on (one index = one GPU = one shard). */
int qv_index_create(qv_index** out, uint32_t dim, qv_metric metric, int device, uint64_t flags);
void qv_index_destroy(qv_index* idx);

/* Pre-size device storage for `rows` rows (amortises growth; optional). */
int qv_index_reserve(qv_index* idx, uint64_t rows);

/* ---- mutation ------------------------------------------------------------------- */

/* Replaces ExactIndex.Insert / HybridIndex.InsertBatch data movement
 * (exact.go:38-58; hybrid_index.go:132-242): append n host rows [n][dim] (row-major
 * float32), copied to the device.  *first_row_out = row number of rows[0]; the
 * others follow contiguously. */
int qv_index_add(qv_index* idx, const float* rows, uint32_t n, uint32_t* first_row_out);

/* Same, rows already on the device (Arrow IPC values buffer -> device path,
 * index/arrow_hnsw.go:222-225).  Synchronous w.r.t. `stream`. */
int qv_index_add_device(qv_index* idx, const float* d_rows, uint32_t n, uint32_t* first_row_out, void* stream);

/* Append n synthetic unit rows generated on the device by the counter-based
 * generator documented in DESIGN.md (synthetic corpus); global row number of the first
 * generated row is `gen_row0` (so shards of one corpus agree).  Benchmark/test
 * helper: keeps 30 GB corpora off PCIe. */
int qv_index_add_synthetic(qv_index* idx, uint64_t seed, uint64_t gen_row0, uint32_t n, uint32_t* first_row_out);

/* Replaces ExactIndex.Delete (exact.go:61-70) / HNSW tombstoning (hnsw.go:829):
 * rows are tombstoned, never renumbered.  Removing a dead row is not an error
 * (exact.go:65 never errors). */
int qv_index_remove(qv_index* idx, const uint32_t* rows, uint32_t n);

/* Overwrite one live or dead row in place and mark it live (Collection.Update
 * path, pkg/core/collection.go:~400: delete + insert under one lock). */
int qv_index_update(qv_index* idx, uint32_t row, const float* vec);

/* ---- queries -------------------------------------------------------------------- */

/* Number of rows ever added (dead rows included) and number of live rows
 * (= ExactIndex.Size, exact.go:136-141). */
uint32_t qv_index_rows(const qv_index* idx);
uint32_t qv_index_size(const qv_index* idx);
uint32_t qv_index_dim(const qv_index* idx);
int      qv_index_metric(const qv_index* idx);

/* Replaces ExactIndex.Search (exact.go:92-133) for nq queries at once
 * (HybridIndex.BatchSearch, hybrid_index.go:677-811, is nq independent searches).
 *   queries  [nq][dim] host float32
 *   k        > 0; clamped to the live size (exact.go:109-111); may equal the size
 *            (filtered search asks for a full ranking, collection.go:679-682)
 *   rows_out [nq][k], dist_out [nq][k]  caller-allocated; entries past count are
 *            row = 0xFFFFFFFF, dist = +inf
 *   count_out[nq] = min(k, live size); empty index -> 0 results, QV_OK (exact.go:96-98)
 * Order of checks follows exact.go:96-106: empty -> ok; then k <= 0 -> error.
 * How k is served (same result whichever applies): up to 64 results the scan keeps a sorted list per wavefront; up to 128 —
 * the negative-example branches fetch max(2k, 30), hybrid_index.go:516-522 — two keys per lane in that list; up to 8192 one
 * key per row and a radix SELECTION of the k smallest; beyond (k = Size() of a filtered search) the full radix ranking.
 * Batches of 9+ queries (32+ with the fp32 filter) over a large corpus go through the matrix-core filter + exact re-score up to
 * 4096 results per query. */
int qv_index_search(qv_index* idx, const float* queries, uint32_t nq, uint32_t k,
                    uint32_t* rows_out, float* dist_out, uint32_t* count_out);

/* How the calls of qv_index_search were served since the index was created.  out[0] solo = calls that ran at once in their own
 * context; out[1] led / out[2] rode = calls that ran a group / had their results written by its leader; out[3] groups, out[4]
 * group_queries = passes that carried a group and the queries in them; out[5] lingers, out[6] linger_ns = groups held open for
 * callers the previous pass had just released, and the time that took in all; out[7] group_pass_ns = time inside the groups'
 * device calls.  For reports and tests. */
int qv_index_coalesce_stats(qv_index* idx, uint64_t out[8]);
/* One more counter of the same front (kept out of out[8] so that ABI 4 callers' arrays stay the size they are): groups whose first
 * pass handed part of their members their results before the rest were redone (the filter's hand-backs inside a group). */
int qv_index_coalesce_early_rounds(qv_index* idx, uint64_t* out);

/* Same with device-resident queries/results; enqueues on `stream`, no sync. */
int qv_index_search_device(qv_index* idx, const float* d_queries, uint32_t nq, uint32_t k,
                           uint32_t* d_rows_out, float* d_dist_out, void* stream);

/* Filtered exact search: only rows whose bit is set in `mask` (bit r%64 of word r/64; host memory,
 * ceil(qv_index_rows/64) words) are candidates.  This is the row-bitmap form of a filtered
 * Collection.Search (collection.go:679-759 ranks ALL rows — searchK = Index.Size() — and keeps the first k
 * whose metadata matches; when the match set is known up front the same k results come from a top-k over
 * the matching rows, without producing or downloading the full ranking).  count_out[q] =
 * min(k, live rows selected by mask); other arguments and ordering as qv_index_search. */
int qv_index_search_masked(qv_index* idx, const float* queries, uint32_t nq, uint32_t k, const uint64_t* mask,
                           uint32_t* rows_out, float* dist_out, uint32_t* count_out);

/* Search with a negative example, the device part of HybridIndex.searchWithStrategy's exact branch
 * (hybrid_index.go:517-570; the HNSW adapter's is adapter.go:345-437): the k_fetch = max(2k, 30) nearest rows of `query`
 * (as qv_index_search), and for exactly those rows the distance to `negative` (as qv_distance_rows) — one call, one
 * synchronisation, the row ids never leave the device in between.  The host forms score = d - w * d_neg in float32
 * (:549) and sorts the <= k_fetch records by (score, id) (:552-557).  Outputs are [k_fetch]; *count_out = min(k_fetch, size). */
int qv_index_search_negative(qv_index* idx, const float* query, const float* negative, uint32_t k_fetch,
                             uint32_t* rows_out, float* dist_out, float* neg_dist_out, uint32_t* count_out);

/* Batched-query path: approximate scores by a GEMM on the matrix cores with a proven error margin (one bfloat16 MFMA
 * term up to 1536 dimensions, three exact-product bfloat16 terms above; qv_index_set_filter chooses otherwise) and
 * fused per-tile candidate selection, then exact re-scoring of the candidates with the same arithmetic as
 * qv_index_search, so results are identical to it.  Same arguments as qv_index_search. */
int qv_index_search_batched(qv_index* idx, const float* queries, uint32_t nq, uint32_t k,
                            uint32_t* rows_out, float* dist_out, uint32_t* count_out);

/* Which filter kernel the batched path of this index uses: QV_FILTER_AUTO (the rule above), QV_FILTER_FP32_MFMA (the dense
 * fp32 GEMM on v_mfma_f32_32x32x2_f32, BASELINE configs[2] as written), QV_FILTER_BF16X3, QV_FILTER_BF16X1, or QV_FILTER_OFF
 * (qv_index_search never takes the batched path; qv_index_search_batched* answer QV_ERR_UNSUPPORTED / fall back).  Results are
 * identical whichever runs (the filter only selects candidates for the exact re-score).  Needs external exclusion against
 * running searches, like the mutations.  The environment variable QV_MFMA_FILTER (read once per process) sets the default of
 * indexes that never call this. */
#define QV_FILTER_AUTO      0
#define QV_FILTER_FP32_MFMA 1
#define QV_FILTER_BF16X3    2
#define QV_FILTER_BF16X1    3
#define QV_FILTER_OFF       4   /* no filter: every batch of this index takes the exact scans (what tests of those scans choose) */
int qv_index_set_filter(qv_index* idx, int filter);

/* Device-pointer form of the batched path: enqueues on `stream`, no sync.  d_redo_flags_out[nq]
 * (uint32) is set to 1 for queries whose candidate buffer overflowed or — 16 or more results per query over 131 072 rows or
 * more, where the k-th distance's bound is a guess from a sample of 32 768 rows or more — whose guess did not hold (fewer than k candidates
 * within it): the caller must redo those with qv_index_search_device (their result rows are unspecified).  Returns QV_ERR_UNSUPPORTED when
 * the MFMA path does not apply (metric, k > 4096, small corpus, too few queries): use qv_index_search_device. */
int qv_index_search_batched_device(qv_index* idx, const float* d_queries, uint32_t nq, uint32_t k,
                                   uint32_t* d_rows_out, float* d_dist_out, uint32_t* d_redo_flags_out, void* stream);

/* Replaces the neighbour loop of HNSW.searchLayer (hnsw.go:536-563) and the
 * re-rank loops (hybrid_index.go:536-546; adapter.go:387-415): distance of one
 * query to n listed rows.  dist_out[i] = distance(query, row rows[i]); dead rows
 * are still evaluated (the reference skips nil nodes before calling). */
int qv_distance_rows(qv_index* idx, const float* query, const uint32_t* rows, uint32_t n, float* dist_out);
int qv_distance_rows_device(qv_index* idx, const float* d_query, const uint32_t* d_rows, uint32_t n,
                            float* d_dist_out, void* stream);

/* Replaces a vectortypes.DistanceFunc call (pkg/vectortypes/surface.go:8) for n
 * independent pairs a[i], b[i] (each [dim]); computed on the device `device`. */
int qv_distance_pairs(qv_metric metric, const float* a, const float* b, uint32_t n, uint32_t dim,
                      float* dist_out, int device);

/* The DistanceFunc contract for ONE pair, on the HOST: what a Go `vectortypes.DistanceFunc` wrapper calls
 * (surface.go:8; 78 ns per call in the reference, final_bench.txt:47 — no device round trip can serve a per-pair call).  Same
 * arithmetic, bit for bit, as qv_distance_pairs and the scans: it is the kernels' own per-pair routine compiled for the CPU.
 * Not a fallback: no search, scan or batch entry point uses it, and those fail without a GPU.  The length check (the reference
 * panics on len(a) != len(b), distances.go:13-15) stays in the host-language wrapper, which knows both lengths. */
int qv_distance_pair(qv_metric metric, const float* a, const float* b, uint32_t dim, float* out);

/* Deterministic merge of per-shard top-k lists (the exchange step of the sharded flat
 * scan: every rank all-gathers its k (distance, global row) pairs over RCCL, then
 * merges).  d_dist_lists / d_row_lists are [n_lists][k] device arrays, each list
 * sorted or not; output = the k smallest by (distance, row).  n_lists * k <= 65536.
 * No reference counterpart: the reference is single-process. */
int qv_merge_topk_device(const float* d_dist_lists, const uint32_t* d_row_lists, uint32_t n_lists, uint32_t k,
                         uint32_t* d_rows_out, float* d_dist_out, void* stream);

/* The same merge for the PACKED exchange buffer of a batch of nq queries: d_packed_lists = [n_lists][nq][2][k] 32-bit
 * words — per shard and query, k shard-local rows (0xFFFFFFFF = no result) followed by the k float32 distances —
 * exactly what a shard's qv_index_search_device wrote for the batch into one [nq][2][k] buffer, so that buffer goes
 * into a single all-gather untouched.  d_bases[n_lists] = first global row of each shard; outputs are [nq][k],
 * rows global.  n_lists * k <= 65536. */
int qv_merge_topk_shards_device(const uint32_t* d_packed_lists, const uint32_t* d_bases, uint32_t n_lists, uint32_t nq, uint32_t k,
                                uint32_t* d_rows_out, float* d_dist_out, void* stream);

/* Measurement aid: when enabled, every flat-scan kernel launched through
 * qv_index_search_device is bracketed by HIP events on the caller's stream;
 * qv_index_profile_read synchronises those events and returns the summed kernel
 * time and the number of launches since the last read.  Off by default. */
int qv_index_profile(qv_index* idx, int enable);
int qv_index_profile_read(qv_index* idx, double* scan_ms_sum_out, uint64_t* launches_out);

/* ---- device-resident HNSW traversal ------------------------------------------------
 * Replaces hnsw.HNSW.Search (pkg/hnsw/hnsw.go:602-713) for a batch of queries: the graph a
 * host-side HNSW built (levels + per-level adjacency, hnsw.go:44-56) is uploaded once, then
 * whole queries are walked on the GPU, one wavefront (or, few queries at a time, one workgroup) per query.  Node index
 * == row of `idx` (the index must hold the nodes' vectors; QV_FLAG_ROWMAJOR makes the
 * per-hop row gathers contiguous).  Results equal the reference's searchLayer semantics
 * exactly (same heaps, same admission order, bit-identical distances).
 *   levels     [n_nodes]            node level, -1 = tombstone (hnsw.go:829 Nodes[idx] = nil)
 *   l0_deg     [n_nodes]            level-0 degree;  l0_links [n_nodes][max_m0]
 *   up_off     [n_nodes]            first upper-level block of the node (level 1)
 *   up_links   [n_up_blocks][1+max_m]  per (node, level>=1): degree, then links
 * qv_graph_search: count_out[q] = results written (< k when the graph search under-filled:
 * the caller tops up like hnsw.go:676-710) or 0xFFFFFFFF when the candidate heap overflowed
 * (the caller must fall back to its host traversal).  evals_out (optional) = distance
 * evaluations per query. */
typedef struct qv_graph qv_graph;
int qv_graph_create(qv_graph** out, qv_index* idx, uint32_t n_nodes, const int8_t* levels, uint32_t max_m0, uint32_t max_m,
                    const uint32_t* l0_deg, const uint32_t* l0_links, const uint32_t* up_off, const uint32_t* up_links,
                    uint32_t n_up_blocks, uint32_t entry, int cur_level);
int qv_graph_search(qv_graph* g, const float* queries, uint32_t nq, uint32_t k, uint32_t ef_search,
                    uint32_t* rows_out, float* dist_out, uint32_t* count_out, uint32_t* evals_out);
/* qv_graph_search is thread-safe: the reference searches under a read lock (hnsw.go:602-606), one goroutine per query
 * (adapter.go:253-279).  Every call works in a context of its own (stream, visited sets, buffers: a pool); calls of up to 64
 * queries with the same k and ef_search that find two traversal batches in flight wait and form the next batch together
 * (256 queries at most).  A batch of at most one query per compute unit — a lone call, a shared batch — runs in the latency
 * form of the kernels: a workgroup of eight wavefronts per query, ~1.1 ms for one traversal of a 1M x 768 graph at efSearch 128
 * (3.1 ms as one wavefront); larger batches one wavefront per query, thousands in flight.  Same results either way.
 * qv_graph_insert / qv_graph_make_buildable / qv_graph_destroy need external exclusion against searches (the reference's
 * write lock).  qv_graph_coalesce_stats: as qv_index_coalesce_stats. */
int qv_graph_coalesce_stats(qv_graph* g, uint64_t out[8]);
int qv_graph_coalesce_early_rounds(qv_graph* g, uint64_t* out);
/* Device-pointer form: queries, results, counts (and optional evals) live on the device; the traversal is
 * enqueued on `stream` (0 = the graph's own stream) with no synchronisation.  One pass only: a query that
 * met two equal distances or a NaN on its way (or visited more nodes than its visited table holds: ~48 x ef)
 * reports count 0xFFFFFFFE and must be redone through qv_graph_search (which runs the exact-heap kernel for
 * those); everything else is final.  DEVICE-FORM traversals on one graph are ordered one after another whatever
 * their streams (they share the graph's own visited tables; host-pointer calls have a context each). */
int qv_graph_search_device(qv_graph* g, const float* d_queries, uint32_t nq, uint32_t k, uint32_t ef_search,
                           uint32_t* d_rows_out, float* d_dist_out, uint32_t* d_count_out, uint32_t* d_evals_out, void* stream);
void qv_graph_destroy(qv_graph* g);

/* ---- device-resident HNSW construction ----------------------------------------------
 * Replaces a loop of hnsw.HNSW.Insert (pkg/hnsw/hnsw.go:266-334: connectNode :337-468, selectNeighbors :583-599)
 * over rows that are already in `idx` (node index == row; the index must have been created with QV_FLAG_ROWMAJOR).
 * The reference connects concurrently (it releases its lock before connectNode, hnsw.go:313-315); the device build
 * inserts in BATCHES with the deterministic form of that: every node of a batch runs connectNode's searches
 * (greedy descent :367-380, then searchLayer(efConstruction) :385 on level min(level, CurrentLevel)) against the
 * graph as it was before the batch — one wavefront per node, the same traversal kernels as qv_graph_search — and
 * forward links, the self-links below the connected level (:463-467) and the back-links with their prune
 * (:413-460) are then applied as if node by node in index order.  A batch of ONE node is exactly Insert, so
 * batch_max = 1 reproduces the reference's sequential graph (the oracle's, tests/test_gpu_build.py).
 *   levels[i]   level of row first_row + i: the caller draws them in node order (randomLevel, hnsw.go:716-738),
 *               which keeps the RNG — seeded from the wall clock in the reference, hnsw.go:248 — on the host side
 *   m, max_m0, ef_construction   0 = the reference's defaults 16 / 2m / 200 (hnsw.go:223-231); <= 64 / 64 / 512
 *   batch_max   largest batch (0 = 16384, the maximum);  ramp_div: a batch never exceeds (nodes already linked) / ramp_div, so
 *               early nodes are inserted nearly one by one (0 = no ramp).  qv_graph_batch_size is the rule.
 * qv_graph_insert appends rows [first_row, first_row + n) (first_row must equal the graph's node count) and returns
 * when they are linked; the graph can be searched (qv_graph_search*) between and after calls. */
uint32_t qv_graph_batch_size(uint32_t nodes_linked, uint32_t batch_max, uint32_t ramp_div);
int qv_graph_create_empty(qv_graph** out, qv_index* idx, uint32_t capacity_nodes, uint32_t m, uint32_t max_m0, uint32_t ef_construction);
int qv_graph_insert(qv_graph* g, uint32_t first_row, uint32_t n, const int8_t* levels, uint32_t batch_max, uint32_t ramp_div);
/* A graph uploaded with qv_graph_create (built on the host) carries no per-link distances, which the device-side
 * construction works on: this scores every existing link once (one wavefront per adjacency list, the traversal's
 * arithmetic: computeDistance(node.Vector, conn.Vector), hnsw.go:438), after which qv_graph_insert can extend the graph.
 * ef_construction as above.  No-op on a graph made by qv_graph_create_empty / qv_graph_build (apart from setting ef). */
int qv_graph_make_buildable(qv_graph* g, uint32_t ef_construction);
/* create_empty + insert of rows [0, n_nodes) */
int qv_graph_build(qv_graph** out, qv_index* idx, uint32_t n_nodes, const int8_t* levels, uint32_t m, uint32_t max_m0,
                   uint32_t ef_construction, uint32_t batch_max, uint32_t ramp_div);
/* Shape of a graph (any output may be null), and a copy of it in the flat form qv_graph_create takes:
 * levels [n_nodes], l0_deg [n_nodes], l0_links [n_nodes][max_m0], up_off [n_nodes], up_links [n_up_blocks][1 + max_m]
 * (any output may be null).  What HNSW.Nodes[i].Connections holds (hnsw.go:44-56), for the host side to keep. */
int qv_graph_info(const qv_graph* g, uint32_t* n_nodes, uint32_t* n_up_blocks, uint32_t* max_m0, uint32_t* max_m, uint32_t* entry, int* cur_level);
int qv_graph_export(qv_graph* g, int8_t* levels, uint32_t* l0_deg, uint32_t* l0_links, uint32_t* up_off, uint32_t* up_links);
/* Counters for reports: seconds spent in qv_graph_insert, batches, construction searches redone by the exact-heap
 * kernel (equal distances), qv_graph_search queries redone for the same reason. */
int qv_graph_stats(const qv_graph* g, double* build_seconds, uint64_t* build_batches, uint64_t* build_redo, uint64_t* search_redo);

/* ---- one corpus over the GPUs of a node --------------------------------
 * No reference counterpart (the reference is one process on CPU cores): this is what lets the Go host reach all the
 * GPUs of a node through ONE handle — what core.Index (pkg/core/collection.go:78-96) is for one GPU (qv_index),
 * qv_sharded is for n, with the same surface: add / remove / update / get / search with any k / filtered search /
 * search with a negative example / listed-row distances.  One host process; devices[g] holds shard g (an exact index
 * of its own) and runs its flat scan on a stream of its own; ONE RCCL all-gather per search carries every shard's
 * result list (nq*k*8 bytes per shard for a top-k, over xGMI between the GPUs of a node: a latency collective); the merge
 * on the first device orders by (distance, global row) like a single index.
 *   global row ids   shard g owns [g * span, (g+1) * span), span = qv_sharded_span(n): "global row = shard base +
 *                    local row" without knowing the corpus size up front; ids are stable as shards grow
 *   qv_sharded_add   cuts a batch into one contiguous piece per shard so that the shards' fill evens out
 *                    (qv_sharded_plan_add is the rule); global_rows_out[i] = id of rows[i] (the host maps string ids);
 *                    all-or-nothing like qv_index_add
 *   any k            k <= 64: per-shard fused top-k + one wavefront-list merge.  Up to 8192 every shard SELECTS its
 *                    min(k, rows) best (as qv_index_search_device does) and one radix selection on the first device takes
 *                    the k best of the gathered lists, the whole batch at once.  Beyond (a filtered Collection.Search asks
 *                    for k = Index.Size(), collection.go:679-682): every shard ranks its rows (radix sort), the sorted runs
 *                    are exchanged and one stable radix sort on the first device merges them
 *   flags            QV_FLAG_ROWMAJOR and QV_FLAG_BF16_ROWS pass through to the shards; QV_SHARDED_PEER_COPY replaces the collective with
 *                    point-to-point copies into the first device (and lets several shards share one device, which
 *                    RCCL does not allow: how the tests exercise 3 and 8 shards on a 1-GPU box)
 * Threading: as qv_index — searches, qv_sharded_distance_rows and qv_sharded_get_row(s) may run concurrently from many
 * threads (the reference searches under a read lock, collection.go:647): every call works in a context of its own
 * (streams, staging and exchange buffers from a pool); add / remove / update / reserve take the handle exclusively
 * (they wait for running searches); destroy needs external exclusion.
 * What has run on hardware (state of round 4; no box with more than one GPU was available to the authors): concurrent callers
 * with the point-to-point exchange (QV_SHARDED_PEER_COPY, shards co-located) and with RCCL at ONE rank.  With RCCL over several
 * devices every context enqueues its grouped all-gather on the shared per-device communicators under one lock, each on streams of
 * its own — within RCCL's rules, but first exercised by tests/test_gpu_sharded_abi.py::test_concurrent_callers_on_one_rccl_handle_*,
 * which needs two GPUs.  Until that has passed on the target node, callers that want no exposure serialise searches on an RCCL
 * handle or create it with QV_SHARDED_PEER_COPY. */
typedef struct qv_sharded qv_sharded;
#define QV_SHARDED_PEER_COPY (1ull << 32)
uint32_t qv_sharded_span(int n_shards);
int qv_sharded_plan_add(const uint64_t* rows_per_shard, int n_shards, uint64_t n, uint64_t* give_out);
int qv_sharded_create(qv_sharded** out, uint32_t dim, qv_metric metric, const int* devices, int n_devices, uint64_t flags);
void qv_sharded_destroy(qv_sharded* s);
int qv_sharded_shards(const qv_sharded* s);
uint64_t qv_sharded_size(const qv_sharded* s);                 /* live rows over all shards */
uint64_t qv_sharded_rows(const qv_sharded* s);                 /* rows ever added over all shards (tombstones included) */
uint32_t qv_sharded_dim(const qv_sharded* s);
int qv_sharded_shard_info(const qv_sharded* s, int shard, int* device, uint32_t* base_row, uint32_t* rows, uint32_t* live);
int qv_sharded_reserve(qv_sharded* s, uint64_t rows_total);
int qv_sharded_add(qv_sharded* s, const float* rows, uint32_t n, uint32_t* global_rows_out);
/* n synthetic rows (the generator of qv_index_add_synthetic), shard g taking the contiguous block [g*n/G, (g+1)*n/G) */
int qv_sharded_add_synthetic(qv_sharded* s, uint64_t seed, uint64_t gen_row0, uint64_t n);
int qv_sharded_remove(qv_sharded* s, const uint32_t* global_rows, uint32_t n);
/* qv_index_update / qv_index_get_row / qv_index_get_rows on the shard that owns the row (Collection.Update,
 * pkg/core/collection.go:417-465; hybrid_index.go:537 reads idx.vectors[id]) */
int qv_sharded_update(qv_sharded* s, uint32_t global_row, const float* vec);
int qv_sharded_get_row(qv_sharded* s, uint32_t global_row, float* vec_out);
int qv_sharded_get_rows(qv_sharded* s, const uint32_t* global_rows, uint32_t n, float* out /* [n][dim] */);
/* Same contract as qv_index_search (check order, clamping, padding, ordering, any k); rows_out holds global row ids. */
int qv_sharded_search(qv_sharded* s, const float* queries, uint32_t nq, uint32_t k, uint32_t* rows_out, float* dist_out, uint32_t* count_out);
/* qv_index_search_masked over the shards.  The candidates are LISTED (n_selected global row ids, any order, duplicates
 * allowed) rather than given as a bitmap, because global row ids are sparse (one id range per shard); dead rows in the list
 * are ignored; an id outside every shard is QV_ERR_OUT_OF_RANGE.  count_out[q] = min(k, live selected rows). */
int qv_sharded_search_masked(qv_sharded* s, const float* queries, uint32_t nq, uint32_t k, const uint32_t* selected_global_rows, uint32_t n_selected,
                             uint32_t* rows_out, float* dist_out, uint32_t* count_out);
/* qv_index_search_negative over the shards (hybrid_index.go:517-570): every shard also evaluates distance(row, negative)
 * for its own k_fetch candidates before the exchange, so the merged list carries both distances after ONE exchange. */
int qv_sharded_search_negative(qv_sharded* s, const float* query, const float* negative, uint32_t k_fetch,
                               uint32_t* rows_out, float* dist_out, float* neg_dist_out, uint32_t* count_out);
/* qv_distance_rows with global row ids: every shard evaluates the rows it owns, all shards in flight together. */
int qv_sharded_distance_rows(qv_sharded* s, const float* query, const uint32_t* global_rows, uint32_t n, float* dist_out);
/* Queries and results resident on the FIRST device.  The work is enqueued on the handle's own streams and ordered after
 * what `stream` (a stream of the first device; null = the null stream, as in qv_index_search_device) held at the call and
 * before anything `stream` runs afterwards; there is no host synchronisation for any nq: queries the matrix-core filter
 * hands back (nq >= 9) are listed and redone by the exact scan on the device (k <= 64: round 5; 64 < k <= 4096: round 6, the
 * listed queries in groups of 64 through shared corpus passes + radix selection; until then their flags were read on the
 * host, one round trip per shard and batch; the first batch with k > 64 allocates that redo's keys per shard: 64 queries x
 * rows x 8 bytes, 1 GiB at most).  The one exception: a shard of more than ~67 M rows at k > 64, where a query's
 * keys leave room for one query at a time, still reads the flags on the host.  Any k. */
int qv_sharded_search_device(qv_sharded* s, const float* d_queries, uint32_t nq, uint32_t k, uint32_t* d_rows_out, float* d_dist_out, void* stream);
int qv_sharded_sync(qv_sharded* s);                            /* wait for every stream of the handle */
/* Measurement aid: with profiling on, every search is synchronous and its phases are timed with HIP events on the first
 * device's stream: its own scan, the exchange (incl. waiting for the slowest shard), merge + download; and every shard's
 * scan KERNEL is bracketed by events on its own stream (qv_index_profile): qv_sharded_profile_read_shard returns the summed
 * kernel time and launch count of one shard since the last read — the per-GPU roofline numerator. */
int qv_sharded_set_filter(qv_sharded* s, int filter);          /* qv_index_set_filter on every shard */
int qv_sharded_profile(qv_sharded* s, int enable);
int qv_sharded_profile_read(qv_sharded* s, double* scan_ms_sum, double* exchange_ms_sum, double* merge_ms_sum, uint64_t* searches);
int qv_sharded_profile_read_shard(qv_sharded* s, int shard, double* scan_kernel_ms_sum, uint64_t* launches);

/* Copy row `row` back to the host (ExactIndex keeps vectors readable,
 * hybrid_index.go:537 reads idx.vectors[id] for the re-rank). */
int qv_index_get_row(qv_index* idx, uint32_t row, float* vec_out);
/* n rows in one device pass (ArrowHNSWIndex.Save writes every vector, index/arrow_hnsw.go:138-198): out = [n][dim] */
int qv_index_get_rows(qv_index* idx, const uint32_t* rows, uint32_t n, float* out);

/* ---- misc ----------------------------------------------------------------------- */
const char* qv_last_error(void);          /* thread-local */
int         qv_abi_version(void);
int         qv_device_count(void);
/* One line naming the HIP runtime and the RCCL this process bound (version + file): "hip_runtime=7.2.x lib=...; rccl=2.27.7
 * lib=...".  Both resolve by soname to whatever the process loaded first (PyTorch bundles its own pair); qv_sharded_create
 * refuses an RCCL exchange when the two come from different installations.  For reports of multi-GPU runs.  Also the process's
 * GPU_MAX_HW_QUEUES (hardware queues per device; the HIP runtime reads it at its first call and defaults to 4): a host that serves
 * concurrent callers sets it to 8 BEFORE its first HIP call — libqv never modifies the environment. */
int         qv_runtime_info(char* out, size_t cap);
/* Timing of the last qv_index_search_device-style launch is the caller's business
 * (HIP events on its stream); this returns static facts for reports. */
int         qv_device_info(int device, char* name_out, size_t name_cap, int* cu_count, uint64_t* hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* QV_H */
