/*
 * qv_oracle.h — CPU restatement of the reference's similarity-search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under quiver_amd/ or include/ may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / the timed CPU baseline.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference tree, TFMV/quiver @ 2026-04-24).  The reference is Go and cannot be
 * compiled here (no Go toolchain); the restatement is pinned by the reference's
 * own known-answer tests, transcribed as data in tests/golden/ref_kats.json.
 */
#ifndef QV_ORACLE_H
#define QV_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* same numbering as include/qv.h qv_metric (the oracle does not include qv.h on purpose) */
enum { QVO_COSINE = 0, QVO_L2 = 1, QVO_L2SQ = 2, QVO_DOT = 3, QVO_L1 = 4,
       QVO_COSINE_F32 = 5, QVO_L2_F32 = 6, QVO_DOT_F32 = 7, QVO_L2SQ_F64 = 8, QVO_METRIC_COUNT = 9 };

/* pkg/vectortypes/distances.go:12-104, pkg/hnsw/adapter.go:105-167, index/arrow_hnsw.go:124-132 */
float qvo_distance(int metric, const float* a, const float* b, uint32_t dim);

/* counter-based synthetic unit vectors, integer arithmetic +
 * correctly-rounded f64 sqrt/div only, so CPU and GPU agree bit for bit */
void qvo_gen_rows(uint64_t seed, uint64_t row0, uint32_t n, uint32_t dim, float* out);

/* pkg/hybrid/exact.go:92-133 with the declared tie-break (distance asc, row asc).
 * alive may be NULL (all live).  Returns the number of results written
 * (min(k, live)), or -1 for k == 0 ("k must be positive"). */
int64_t qvo_exact_search(int metric, const float* rows, const uint8_t* alive, uint32_t n, uint32_t dim,
                         const float* query, uint32_t k, uint32_t* rows_out, float* dist_out);

/* all n distances (no selection); dist_out[n] */
void qvo_all_distances(int metric, const float* rows, uint32_t n, uint32_t dim, const float* query, float* dist_out);

/* pkg/hybrid/hybrid_index.go:517-570: fetch max(2k,30) by exact search, re-rank by
 * d - w*d_neg (float32), stable sort by (score, id) where id order is given by
 * id_rank[row] (the rank of the row's string id in lexicographic order; NULL =
 * row order), truncate to k.  dist_out holds the re-ranked score. */
int64_t qvo_exact_search_negative(int metric, const float* rows, const uint8_t* alive, uint32_t n, uint32_t dim,
                                  const float* query, const float* negative, float neg_weight, uint32_t k,
                                  const uint32_t* id_rank, uint32_t* rows_out, float* dist_out);

/* ---- reference-faithful CPU baseline (what ExactIndex.Search costs) ------------
 * rows individually heap-allocated behind a string-keyed hash map (exact.go:16),
 * one scalar distance call per row through a function pointer (exact.go:116),
 * full sort of all N (id, distance) records by distance (exact.go:124), truncate. */
typedef struct qvo_faithful qvo_faithful;
qvo_faithful* qvo_faithful_create(int metric, uint32_t dim);
void          qvo_faithful_destroy(qvo_faithful*);
int           qvo_faithful_insert(qvo_faithful*, const char* id, const float* vec); /* copies */
uint32_t      qvo_faithful_size(const qvo_faithful*);
/* ids_out[k] receives pointers to the stored id strings (owned by the index) */
int64_t       qvo_faithful_search(qvo_faithful*, const float* query, uint32_t k, const char** ids_out, float* dist_out);

/* ---- CPU baselines beyond one thread (qv_cpu_baselines.c) ---------------------------------
 * nq faithful searches over `threads` threads, one whole search per thread at a time: what HybridIndex.BatchSearch does
 * with that many cores (hybrid_index.go:703-705).  dist_out [nq][k] optional.  Returns wall seconds. */
double qvo_faithful_search_many(qvo_faithful*, uint32_t dim, const float* queries, uint32_t nq, uint32_t k, int threads, float* dist_out);
/* NOT the reference: contiguous rows, cached norms, 16-wide float32 FMA lanes (AVX-512 where present), rows split over
 * `threads` pinned threads (each slice allocated by its own thread: NUMA-local), per-thread partial top-k, spinning
 * barriers.  Returns wall seconds for the nq queries, one after another (thread start-up excluded). */
typedef struct qvo_opt qvo_opt;
qvo_opt* qvo_opt_create(const float* rows, uint32_t n, uint32_t dim, int threads);   /* per-thread NUMA-local slices + cached norms */
void   qvo_opt_destroy(qvo_opt*);
double qvo_opt_cosine_scan(qvo_opt*, const float* queries, uint32_t nq, uint32_t k, uint32_t* rows_out, float* dist_out);
int    qvo_opt_simd_bits(void);

/* ---- HNSW restatement (pkg/hnsw/hnsw.go) ------------------------------------------ */
typedef struct qvo_hnsw qvo_hnsw;
/* NewHNSW hnsw.go:222-252; defaults M=16, MaxM0=2M, efC=200, efS=100, MaxLevel=16.
 * The reference seeds its level RNG from the wall clock (hnsw.go:248); the
 * restatement takes an explicit seed so a graph can be rebuilt identically. */
qvo_hnsw* qvo_hnsw_create(int metric, uint32_t dim, int M, int maxM0, int efConstruction, int efSearch,
                          int maxLevel, uint64_t seed);
void      qvo_hnsw_destroy(qvo_hnsw*);
/* Insert hnsw.go:266-334 (+ connectNode :337-468, selectNeighbors :583-599,
 * randomLevel :716-738).  Returns the node index. */
int64_t   qvo_hnsw_insert(qvo_hnsw*, const float* vec);
/* Delete hnsw.go:741-842 (tombstone + unlink + entry-point repair) */
int       qvo_hnsw_delete(qvo_hnsw*, uint32_t node);
/* Search hnsw.go:602-713; returns count; n_eval_out (optional) = distance evaluations */
int64_t   qvo_hnsw_search(qvo_hnsw*, const float* query, uint32_t k, uint32_t* rows_out, float* dist_out,
                          uint64_t* n_eval_out);
void      qvo_hnsw_set_ef_search(qvo_hnsw*, int ef);
uint32_t  qvo_hnsw_size(const qvo_hnsw*);      /* live nodes */
uint32_t  qvo_hnsw_nodes(const qvo_hnsw*);     /* nodes ever inserted */
int       qvo_hnsw_entry_point(const qvo_hnsw*, uint32_t* ep_out, int* cur_level_out);
int       qvo_hnsw_node_level(const qvo_hnsw*, uint32_t node); /* -1 if deleted */
/* copy out node's links at `level`; returns count (<= cap) or -1 */
int       qvo_hnsw_links(const qvo_hnsw*, uint32_t node, int level, uint32_t* out, uint32_t cap);
/* n Inserts "at once" (the reference connects concurrently, hnsw.go:313-315): levels drawn in node order, every node's
 * searches see the graph as it was before the batch, links applied node by node in index order with the reference's
 * append / re-scoring prune.  n == 1 is qvo_hnsw_insert.  This is the CPU statement of what qv_graph_insert does on the
 * device.  Returns the index of the first node, -1 on error. */
int64_t   qvo_hnsw_insert_batch(qvo_hnsw*, const float* vecs, uint32_t n);
/* test scaffolding: install a multi-level graph in include/qv.h's flat form (qv_graph_export); rows are borrowed */
int       qvo_hnsw_load_graph(qvo_hnsw*, uint32_t n, const float* rows, const int8_t* levels, uint32_t max_m0, uint32_t max_m,
                              const uint32_t* l0_deg, const uint32_t* l0_links, const uint32_t* up_off, const uint32_t* up_links,
                              uint32_t entry, int cur_level);
/* test scaffolding (not a reference function): install a ready-made single-layer graph; rows are borrowed */
int       qvo_hnsw_load_flat(qvo_hnsw*, uint32_t n, const float* rows, const uint32_t* deg, const uint32_t* links,
                             uint32_t stride, uint32_t entry);
/* test scaffolding (not a reference function): the next n level draws return levels[0..n) (borrowed until consumed) */
void      qvo_hnsw_force_levels(qvo_hnsw*, const int8_t* levels, uint32_t n);
/* the level law alone, for property tests: p(level>=l+1 | level>=l) = 0.25, <= min(MaxLevel,10) draws */
int       qvo_hnsw_random_level(qvo_hnsw*);

#ifdef __cplusplus
}
#endif
#endif
