/*
 * qv_oracle_hnsw.c — CPU restatement of pkg/hnsw/hnsw.go (graph build, searchLayer,
 * Search, Delete).  TEST INFRASTRUCTURE ONLY (see qv_oracle.h).
 *
 * The restatement keeps the reference's behaviour statement for statement,
 * including its quirks, because traversal order decides results under ties:
 *   - both heaps compare on Distance only, with the reference's own sift loops
 *     (hnsw.go:101-196), restated verbatim so equal-distance pops come out in the
 *     same order;
 *   - connectNode re-enters lower levels from the new node itself
 *     (hnsw.go:465-467), so a node with level >= 1 links to itself below its top
 *     connected level (the search from an unlinked node returns only that node);
 *   - the admission threshold tightens inside a hop (hnsw.go:553-560);
 *   - Delete may leave EntryPoint on a node it then tombstones (hnsw.go:796-805),
 *     Search repairs by taking the first live node (hnsw.go:620-628).
 * Differences, both declared in DESIGN.md: the level RNG takes a seed (the
 * reference seeds from the wall clock, hnsw.go:248), and the under-fill top-up
 * sort breaks distance ties by node index (the reference: by string id,
 * hnsw.go:699-704; node index == insertion order of the ids).
 */
#include "qv_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef struct { float dist; uint32_t idx; } res_t;

typedef struct {
    float* vec;
    int level;
    uint32_t** conn;   /* [level+1] */
    uint32_t* conn_len;
    uint32_t* conn_cap;
    int alive;
    int borrowed;      /* vec points into the caller's array (qvo_hnsw_load_flat) */
} node_t;

struct qvo_hnsw {
    int metric; uint32_t dim;
    int M, maxM0, efC, efS, maxLevel;
    node_t* nodes; uint32_t n_nodes, cap_nodes;
    uint32_t entry; int cur_level; uint32_t size;
    uint64_t rng;
    uint32_t* visited; uint32_t visited_cap; uint32_t epoch;
    uint64_t n_eval;
    const int8_t* forced_levels; uint32_t forced_n, forced_i;         /* test scaffolding: qvo_hnsw_force_levels */
};

/* ---- heaps, hnsw.go:101-196 ------------------------------------------------------ */
typedef struct { res_t* a; int n, cap; } heap_t;
static void heap_reserve(heap_t* h) { if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 128; h->a = (res_t*)realloc(h->a, (size_t)h->cap * sizeof(res_t)); } }

static void min_up(res_t* rs, int j) {                                /* :118-128 */
    for (;;) { int i = (j - 1) / 2; if (i == j || rs[j].dist >= rs[i].dist) break; res_t t = rs[i]; rs[i] = rs[j]; rs[j] = t; j = i; }
}
static void min_down(res_t* rs, int i0, int n) {                      /* :130-148 */
    int i = i0;
    for (;;) {
        int j1 = 2 * i + 1; if (j1 >= n || j1 < 0) break;
        int j = j1; int j2 = j1 + 1; if (j2 < n && rs[j2].dist < rs[j1].dist) j = j2;
        if (rs[i].dist <= rs[j].dist) break;
        res_t t = rs[i]; rs[i] = rs[j]; rs[j] = t; i = j;
    }
}
static void min_push(heap_t* h, res_t x) { heap_reserve(h); h->a[h->n++] = x; min_up(h->a, h->n - 1); }   /* :103-106 */
static res_t min_pop(heap_t* h) {                                     /* :108-116 */
    int n = h->n - 1; res_t t = h->a[0]; h->a[0] = h->a[n]; h->a[n] = t; min_down(h->a, 0, n); h->n = n; return h->a[n];
}
static void max_up(res_t* rs, int j) {                                /* :172-181 */
    for (;;) { int i = (j - 1) / 2; if (i == j || rs[j].dist <= rs[i].dist) break; res_t t = rs[i]; rs[i] = rs[j]; rs[j] = t; j = i; }
}
static void max_down(res_t* rs, int i0, int n) {                      /* :183-200 */
    int i = i0;
    for (;;) {
        int j1 = 2 * i + 1; if (j1 >= n || j1 < 0) break;
        int j = j1; int j2 = j1 + 1; if (j2 < n && rs[j2].dist > rs[j1].dist) j = j2;
        if (rs[i].dist >= rs[j].dist) break;
        res_t t = rs[i]; rs[i] = rs[j]; rs[j] = t; i = j;
    }
}
static void max_push(heap_t* h, res_t x) { heap_reserve(h); h->a[h->n++] = x; max_up(h->a, h->n - 1); }
static res_t max_pop(heap_t* h) { int n = h->n - 1; res_t t = h->a[0]; h->a[0] = h->a[n]; h->a[n] = t; max_down(h->a, 0, n); h->n = n; return h->a[n]; }

/* ---- helpers --------------------------------------------------------------------- */
static inline uint64_t sm64_next(uint64_t* s) {
    uint64_t x = (*s += 0x9E3779B97F4A7C15ull);
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
static inline double rng_float64(qvo_hnsw* h) { return (double)(sm64_next(&h->rng) >> 11) * (1.0 / 9007199254740992.0); }

static inline int node_ok(const qvo_hnsw* h, uint32_t i) { return i < h->n_nodes && h->nodes[i].alive; }
static inline float dist(qvo_hnsw* h, const float* a, const float* b) { h->n_eval++; return qvo_distance(h->metric, a, b, h->dim); }

static void conn_append(node_t* nd, int lc, uint32_t v) {
    if (nd->conn_len[lc] == nd->conn_cap[lc]) { nd->conn_cap[lc] = nd->conn_cap[lc] ? nd->conn_cap[lc] * 2 : 8; nd->conn[lc] = (uint32_t*)realloc(nd->conn[lc], nd->conn_cap[lc] * sizeof(uint32_t)); }
    nd->conn[lc][nd->conn_len[lc]++] = v;
}

qvo_hnsw* qvo_hnsw_create(int metric, uint32_t dim, int M, int maxM0, int efC, int efS, int maxLevel, uint64_t seed) {
    qvo_hnsw* h = (qvo_hnsw*)calloc(1, sizeof(*h));
    h->metric = metric; h->dim = dim;
    h->M = M > 0 ? M : 16;                                /* hnsw.go:223-225 */
    h->maxM0 = maxM0 > 0 ? maxM0 : h->M * 2;              /* :226-228 */
    h->efC = efC > 0 ? efC : 200;                         /* :229-231 */
    h->efS = efS > 0 ? efS : 100;                         /* :232-234 */
    h->maxLevel = maxLevel > 0 ? maxLevel : 16;           /* :235-237 */
    h->cur_level = -1; h->entry = 0; h->rng = seed;
    return h;
}
void qvo_hnsw_destroy(qvo_hnsw* h) {
    if (!h) return;
    for (uint32_t i = 0; i < h->n_nodes; i++) {
        node_t* nd = &h->nodes[i];
        for (int l = 0; l <= nd->level; l++) free(nd->conn[l]);
        free(nd->conn); free(nd->conn_len); free(nd->conn_cap); if (!nd->borrowed) free(nd->vec);
    }
    free(h->nodes); free(h->visited); free(h);
}
void qvo_hnsw_set_ef_search(qvo_hnsw* h, int ef) { if (ef > 0) h->efS = ef; }
uint32_t qvo_hnsw_size(const qvo_hnsw* h) { return h->size; }
uint32_t qvo_hnsw_nodes(const qvo_hnsw* h) { return h->n_nodes; }
int qvo_hnsw_entry_point(const qvo_hnsw* h, uint32_t* ep, int* lvl) { if (ep) *ep = h->entry; if (lvl) *lvl = h->cur_level; return 0; }
int qvo_hnsw_node_level(const qvo_hnsw* h, uint32_t n) { return node_ok(h, n) ? h->nodes[n].level : -1; }
int qvo_hnsw_links(const qvo_hnsw* h, uint32_t n, int level, uint32_t* out, uint32_t cap) {
    if (!node_ok(h, n) || level < 0 || level > h->nodes[n].level) return -1;
    uint32_t c = h->nodes[n].conn_len[level]; if (c > cap) c = cap;
    if (c) memcpy(out, h->nodes[n].conn[level], c * sizeof(uint32_t));   /* (a list that was never appended to has no array: memcpy from NULL is undefined even for 0 bytes) */
    return (int)c;
}

/* randomLevel, hnsw.go:716-738 */
int qvo_hnsw_random_level(qvo_hnsw* h) {
    if (h->forced_i < h->forced_n) return h->forced_levels[h->forced_i++];   /* (scaffolding: the caller drew the levels) */
    int level = 0;
    int maxAttempts = h->maxLevel < 10 ? h->maxLevel : 10;
    for (int i = 0; i < maxAttempts; i++) { if (rng_float64(h) < 0.25) level++; else break; }
    if (level >= h->maxLevel) level = h->maxLevel - 1;
    return level;
}

/* Test scaffolding, not a reference function: the next n level draws return levels[0..n) instead of consuming the RNG — for a
 * caller that keeps the RNG on its own side (qv_graph_insert takes the levels of its nodes as an argument, include/qv.h).
 * `levels` is BORROWED until consumed. */
void qvo_hnsw_force_levels(qvo_hnsw* h, const int8_t* levels, uint32_t n) { h->forced_levels = levels; h->forced_n = n; h->forced_i = 0; }

/* searchLayer, hnsw.go:471-580.  Returns count in *out (ascending), -1 on invalid entry. */
static int search_layer(qvo_hnsw* h, const float* q, uint32_t entry, int ef, int level, res_t** out_p, int* out_cap) {
    if (h->n_nodes == 0) return 0;                                    /* :473-475 */
    if (!node_ok(h, entry)) return -1;                                /* :478-480 */
    if (h->visited_cap < h->n_nodes) {                                /* visited set :483-488 (epoch array == cleared map) */
        h->visited = (uint32_t*)realloc(h->visited, (size_t)h->cap_nodes * sizeof(uint32_t));
        memset(h->visited + h->visited_cap, 0, (size_t)(h->cap_nodes - h->visited_cap) * sizeof(uint32_t));
        h->visited_cap = h->cap_nodes;
    }
    if (++h->epoch == 0) { memset(h->visited, 0, (size_t)h->visited_cap * sizeof(uint32_t)); h->epoch = 1; }
    h->visited[entry] = h->epoch;
    float d0 = dist(h, q, h->nodes[entry].vec);                       /* :492 */
    heap_t cand = {0}, res = {0};
    res_t e = { d0, entry };
    min_push(&cand, e); max_push(&res, e);                            /* :498-506 */
    while (cand.n > 0) {                                              /* :509 */
        res_t cur = min_pop(&cand);                                   /* :511 */
        if (res.n >= ef && cur.dist > res.a[0].dist) break;           /* :514-516 */
        if (!node_ok(h, cur.idx)) continue;                           /* :519-521 */
        node_t* nd = &h->nodes[cur.idx];
        if (level >= nd->level + 1) continue;                         /* :528-531 */
        uint32_t nconn = nd->conn_len[level]; const uint32_t* conns = nd->conn[level];
        for (uint32_t ci = 0; ci < nconn; ci++) {                     /* :537 */
            uint32_t c = conns[ci];
            if (!node_ok(h, c)) continue;                             /* :539-541 */
            if (h->visited[c] != h->epoch) {                          /* :543 */
                h->visited[c] = h->epoch;
                float cd = dist(h, q, h->nodes[c].vec);               /* :548 */
                if (res.n < ef || cd < res.a[0].dist) {               /* :553 */
                    res_t r = { cd, c };
                    min_push(&cand, r); max_push(&res, r);            /* :554-555 */
                    if (res.n > ef) (void)max_pop(&res);              /* :558-560 */
                }
            }
        }
    }
    int n = res.n;                                                    /* :566-577 pop max-heap into the slice back to front */
    if (*out_cap < n) { *out_cap = n; *out_p = (res_t*)realloc(*out_p, (size_t)n * sizeof(res_t)); }
    for (int i = n - 1; i >= 0; i--) (*out_p)[i] = max_pop(&res);
    free(cand.a); free(res.a);
    return n;
}

/* selectNeighbors, hnsw.go:583-599: sort by (Distance, VectorIndex), keep k */
static int sel_cmp(const void* pa, const void* pb) {
    const res_t* a = (const res_t*)pa; const res_t* b = (const res_t*)pb;
    if (a->dist == b->dist) return (a->idx > b->idx) - (a->idx < b->idx);
    if (a->dist < b->dist) return -1;
    if (a->dist > b->dist) return 1;
    return (a->idx > b->idx) - (a->idx < b->idx);                     /* NaN: fall back to index */
}
static int select_neighbors(res_t* c, int n, int k) {
    if (k <= 0 || n == 0) return 0;
    qsort(c, (size_t)n, sizeof(res_t), sel_cmp);
    return n > k ? k : n;
}

/* connectNode, hnsw.go:337-468 */
static int connect_node(qvo_hnsw* h, uint32_t nodeIdx, const float* vec, int level, int graphLevel) {
    if (level >= h->maxLevel) level = h->maxLevel - 1;                /* :342-344 */
    if (h->n_nodes == 1) { h->entry = nodeIdx; h->cur_level = level; return 0; }   /* :347-351 */
    uint32_t entry = h->entry;
    if (!node_ok(h, entry)) {                                         /* :356-364 */
        for (uint32_t i = 0; i < h->n_nodes; i++) if (i != nodeIdx && h->nodes[i].alive) { entry = i; break; }
    }
    res_t* buf = NULL; int buf_cap = 0;
    for (int lc = graphLevel; lc > level; lc--) {                     /* :367-380 */
        if (!node_ok(h, entry)) break;                                /* the reference would panic on a nil entry here */
        if (lc >= h->nodes[entry].level + 1) continue;                /* :369-371 */
        int n = search_layer(h, vec, entry, 1, lc, &buf, &buf_cap);
        if (n < 0) { free(buf); return -1; }
        if (n > 0) entry = buf[0].idx;
    }
    int top = level < graphLevel ? level : graphLevel;
    for (int lc = top; lc >= 0; lc--) {                               /* :383 */
        if (!node_ok(h, entry)) break;
        int n = search_layer(h, vec, entry, h->efC, lc, &buf, &buf_cap);   /* :385 */
        if (n < 0) { free(buf); return -1; }
        if (n == 0) continue;                                         /* :391-393 */
        int maxConn = lc == 0 ? h->maxM0 : h->M;                      /* :395-398 */
        int nsel = select_neighbors(buf, n, maxConn < n ? maxConn : n);    /* :401 */
        node_t* nn = &h->nodes[nodeIdx];
        for (int i = 0; i < nsel; i++) conn_append(nn, lc, buf[i].idx);    /* :407-409 */
        for (int i = 0; i < nsel; i++) {                              /* :413 back-links */
            uint32_t nb = buf[i].idx;
            if (!node_ok(h, nb)) continue;                            /* :415-417 */
            node_t* nbn = &h->nodes[nb];
            if (lc > nbn->level) continue;                            /* :420-422 */
            conn_append(nbn, lc, nodeIdx);                            /* :426 */
            if ((int)nbn->conn_len[lc] > maxConn) {                   /* :429 prune */
                int cn = (int)nbn->conn_len[lc];
                res_t* nd = (res_t*)malloc((size_t)cn * sizeof(res_t)); int m = 0;
                for (int j = 0; j < cn; j++) {                        /* :432-448 */
                    uint32_t ci = nbn->conn[lc][j];
                    if (!node_ok(h, ci)) continue;
                    nd[m].dist = dist(h, nbn->vec, h->nodes[ci].vec); nd[m].idx = ci; m++;
                }
                int keep = select_neighbors(nd, m, maxConn);          /* :451 */
                nbn->conn_len[lc] = 0;                                /* :454-457 */
                for (int j = 0; j < keep; j++) conn_append(nbn, lc, nd[j].idx);
                free(nd);
            }
        }
        if (nsel > 0) entry = nodeIdx;                                /* :463-465 (the self-entry quirk) */
    }
    free(buf);
    return 0;
}

/* Insert, hnsw.go:266-334 */
int64_t qvo_hnsw_insert(qvo_hnsw* h, const float* vec) {
    int level = qvo_hnsw_random_level(h);                             /* :275 */
    int oldLevel = h->cur_level;                                      /* :276 */
    if (h->n_nodes == h->cap_nodes) { h->cap_nodes = h->cap_nodes ? h->cap_nodes * 2 : 1024; h->nodes = (node_t*)realloc(h->nodes, (size_t)h->cap_nodes * sizeof(node_t)); }
    uint32_t idx = h->n_nodes;
    node_t* nd = &h->nodes[idx];
    nd->vec = (float*)malloc(h->dim * sizeof(float)); memcpy(nd->vec, vec, h->dim * sizeof(float));  /* :281-282 copy */
    nd->level = level; nd->alive = 1; nd->borrowed = 0;               /* (the node array grows by realloc: every field is set here) */
    nd->conn = (uint32_t**)calloc((size_t)level + 1, sizeof(uint32_t*));
    nd->conn_len = (uint32_t*)calloc((size_t)level + 1, sizeof(uint32_t));
    nd->conn_cap = (uint32_t*)calloc((size_t)level + 1, sizeof(uint32_t));
    h->n_nodes++; h->size++;                                          /* :302-304 */
    if (h->n_nodes == 1) { h->entry = 0; h->cur_level = level; return idx; }   /* :307-312 */
    if (connect_node(h, idx, vec, level, oldLevel) != 0) {            /* :316-323 rollback */
        nd->alive = 0; h->size--; return -1;
    }
    if (level > oldLevel) {                                           /* :325-332 */
        if (level > h->cur_level) { h->entry = idx; h->cur_level = level; }
    }
    return idx;
}

/* ---- batched insertion ------------------------------------------------------------
 * The reference's Insert releases the index lock BEFORE connectNode ("Connect the new node to the graph
 * concurrently", hnsw.go:313-315), so n goroutines calling Insert at once is part of its contract: every one of them
 * searches a graph in which the others are not linked yet.  qvo_hnsw_insert_batch is the deterministic member of that
 * family which the device build (qv_graph_insert) implements:
 *   1. levels are drawn in node order (:275) and the nodes appended (:279-303);
 *   2. every node of the batch runs connectNode's searches (:367-385) against the graph AS IT WAS BEFORE THE BATCH
 *      (snapshot entry point / CurrentLevel; batch nodes are unreachable: nothing links to them yet);
 *   3. links are applied node by node, in index order, with the reference's own statements: forward links :404-409,
 *      back-links with the re-scoring prune :413-460, then the entry-point update :325-332.
 * A batch of ONE node is exactly qvo_hnsw_insert (tests/test_oracle_hnsw.py asserts graph equality). */
typedef struct { int top; int n_lv; uint32_t** sel; int* nsel; } plan_t;

static int connect_search(qvo_hnsw* h, uint32_t nodeIdx, const float* vec, int level, int graphLevel, uint32_t entry, plan_t* pl) {
    if (level >= h->maxLevel) level = h->maxLevel - 1;                /* :342-344 */
    res_t* buf = NULL; int buf_cap = 0;
    for (int lc = graphLevel; lc > level; lc--) {                     /* :367-380 */
        if (!node_ok(h, entry)) break;
        if (lc >= h->nodes[entry].level + 1) continue;
        int n = search_layer(h, vec, entry, 1, lc, &buf, &buf_cap);
        if (n < 0) { free(buf); return -1; }
        if (n > 0) entry = buf[0].idx;
    }
    int top = level < graphLevel ? level : graphLevel;
    pl->top = top; pl->n_lv = top + 1;
    pl->sel = (uint32_t**)calloc((size_t)(top + 1 > 0 ? top + 1 : 1), sizeof(uint32_t*));
    pl->nsel = (int*)calloc((size_t)(top + 1 > 0 ? top + 1 : 1), sizeof(int));
    for (int lc = top; lc >= 0; lc--) {                               /* :383 */
        if (!node_ok(h, entry)) break;
        int n = search_layer(h, vec, entry, h->efC, lc, &buf, &buf_cap);   /* :385 */
        if (n < 0) { free(buf); return -1; }
        if (n == 0) continue;
        int maxConn = lc == 0 ? h->maxM0 : h->M;
        int nsel = select_neighbors(buf, n, maxConn < n ? maxConn : n);    /* :401 */
        pl->sel[lc] = (uint32_t*)malloc((size_t)(nsel ? nsel : 1) * sizeof(uint32_t));
        for (int i = 0; i < nsel; i++) pl->sel[lc][i] = buf[i].idx;
        pl->nsel[lc] = nsel;
        if (nsel > 0) entry = nodeIdx;                                /* :463-465: the node's lower-level lists are still empty,
                                                                         so the next search returns the node itself */
    }
    free(buf);
    return 0;
}

static void connect_apply(qvo_hnsw* h, uint32_t nodeIdx, const plan_t* pl) {
    for (int lc = pl->top; lc >= 0; lc--) {
        int nsel = pl->nsel[lc];
        if (nsel == 0) continue;
        int maxConn = lc == 0 ? h->maxM0 : h->M;
        node_t* nn = &h->nodes[nodeIdx];
        for (int i = 0; i < nsel; i++) conn_append(nn, lc, pl->sel[lc][i]);      /* :407-409 */
        for (int i = 0; i < nsel; i++) {                              /* :413 */
            uint32_t nb = pl->sel[lc][i];
            if (!node_ok(h, nb)) continue;
            node_t* nbn = &h->nodes[nb];
            if (lc > nbn->level) continue;                            /* :420-422 */
            conn_append(nbn, lc, nodeIdx);                            /* :426 */
            if ((int)nbn->conn_len[lc] > maxConn) {                   /* :429 */
                int cn = (int)nbn->conn_len[lc];
                res_t* nd = (res_t*)malloc((size_t)cn * sizeof(res_t)); int m = 0;
                for (int j = 0; j < cn; j++) {
                    uint32_t ci = nbn->conn[lc][j];
                    if (!node_ok(h, ci)) continue;
                    nd[m].dist = dist(h, nbn->vec, h->nodes[ci].vec); nd[m].idx = ci; m++;   /* :438 */
                }
                int keep = select_neighbors(nd, m, maxConn);          /* :451 */
                nbn->conn_len[lc] = 0;
                for (int j = 0; j < keep; j++) conn_append(nbn, lc, nd[j].idx);
                free(nd);
            }
        }
    }
}

int64_t qvo_hnsw_insert_batch(qvo_hnsw* h, const float* vecs, uint32_t n) {
    if (n == 0) return (int64_t)h->n_nodes;
    if (h->n_nodes == 0) {                                            /* the first node has nothing to search (:306-311) */
        if (qvo_hnsw_insert(h, vecs) < 0) return -1;
        if (n == 1) return 0;
        int64_t r = qvo_hnsw_insert_batch(h, vecs + h->dim, n - 1);
        return r < 0 ? r : 0;
    }
    const uint32_t first = h->n_nodes;
    const int snapLevel = h->cur_level;                               /* every batch node's oldCurrentLevel (:276) */
    uint32_t snapEntry = h->entry;
    if (!node_ok(h, snapEntry)) {                                     /* :356-364 */
        for (uint32_t i = 0; i < first; i++) if (h->nodes[i].alive) { snapEntry = i; break; }
    }
    while (h->n_nodes + n > h->cap_nodes) { h->cap_nodes = h->cap_nodes ? h->cap_nodes * 2 : 1024; h->nodes = (node_t*)realloc(h->nodes, (size_t)h->cap_nodes * sizeof(node_t)); }
    for (uint32_t i = 0; i < n; i++) {                                /* :275-303, in node order */
        int level = qvo_hnsw_random_level(h);
        node_t* nd = &h->nodes[first + i];
        nd->vec = (float*)malloc(h->dim * sizeof(float)); memcpy(nd->vec, vecs + (size_t)i * h->dim, h->dim * sizeof(float));
        nd->level = level; nd->alive = 1; nd->borrowed = 0;
        nd->conn = (uint32_t**)calloc((size_t)level + 1, sizeof(uint32_t*));
        nd->conn_len = (uint32_t*)calloc((size_t)level + 1, sizeof(uint32_t));
        nd->conn_cap = (uint32_t*)calloc((size_t)level + 1, sizeof(uint32_t));
    }
    h->n_nodes += n; h->size += n;
    plan_t* plans = (plan_t*)calloc(n, sizeof(plan_t));
    int rc = 0;
    for (uint32_t i = 0; i < n && rc == 0; i++)                       /* searches: the graph before the batch */
        rc = connect_search(h, first + i, h->nodes[first + i].vec, h->nodes[first + i].level, snapLevel, snapEntry, &plans[i]);
    for (uint32_t i = 0; i < n && rc == 0; i++) {                     /* links: node by node */
        connect_apply(h, first + i, &plans[i]);
        int level = h->nodes[first + i].level;
        if (level > snapLevel && level > h->cur_level) { h->entry = first + i; h->cur_level = level; }   /* :325-332 */
    }
    for (uint32_t i = 0; i < n; i++) { for (int l = 0; l < plans[i].n_lv; l++) free(plans[i].sel[l]); free(plans[i].sel); free(plans[i].nsel); }
    free(plans);
    return rc == 0 ? (int64_t)first : -1;
}

/* Test scaffolding, not a reference function: install a ready-made MULTI-level graph in the flat form of
 * include/qv.h (qv_graph_export): levels[n], level-0 degree/links, and per (node, level >= 1) blocks of
 * (1 + max_m) words — so the CPU traversal can walk the very graph the device built.  `rows` is BORROWED. */
int qvo_hnsw_load_graph(qvo_hnsw* h, uint32_t n, const float* rows, const int8_t* levels, uint32_t max_m0, uint32_t max_m,
                        const uint32_t* l0_deg, const uint32_t* l0_links, const uint32_t* up_off, const uint32_t* up_links,
                        uint32_t entry, int cur_level) {
    if (!h || h->n_nodes != 0 || n == 0 || entry >= n) return -1;
    h->nodes = (node_t*)calloc(n, sizeof(node_t)); h->cap_nodes = n;
    uint32_t live = 0;
    for (uint32_t i = 0; i < n; i++) {
        node_t* nd = &h->nodes[i];
        nd->vec = (float*)(rows + (size_t)i * h->dim); nd->borrowed = 1;
        int lv = levels[i];
        nd->alive = lv >= 0; if (lv < 0) lv = 0; else live++;
        nd->level = lv;
        nd->conn = (uint32_t**)calloc((size_t)lv + 1, sizeof(uint32_t*));
        nd->conn_len = (uint32_t*)calloc((size_t)lv + 1, sizeof(uint32_t));
        nd->conn_cap = (uint32_t*)calloc((size_t)lv + 1, sizeof(uint32_t));
        for (int l = 0; l <= lv; l++) {
            uint32_t d; const uint32_t* src;
            if (l == 0) { d = l0_deg[i] < max_m0 ? l0_deg[i] : max_m0; src = l0_links + (size_t)i * max_m0; }
            else { const uint32_t* blk = up_links + (size_t)(up_off[i] + (uint32_t)(l - 1)) * (1 + max_m); d = blk[0] < max_m ? blk[0] : max_m; src = blk + 1; }
            if (!nd->alive) d = 0;
            nd->conn[l] = (uint32_t*)malloc((d ? d : 1) * sizeof(uint32_t));
            memcpy(nd->conn[l], src, d * sizeof(uint32_t));
            nd->conn_len[l] = d; nd->conn_cap[l] = d ? d : 1;
        }
    }
    h->n_nodes = n; h->size = live; h->entry = entry; h->cur_level = cur_level;
    return 0;
}

/* Test scaffolding, not a reference function: install a ready-made single-layer graph (every node at
 * level 0, links as given) so that Search (hnsw.go:602-713) can be run on graphs that were not built by
 * Insert — e.g. an exact k-NN graph at a size where the sequential reference build is impractical.
 * `rows` is BORROWED (must outlive the index).  The index must be empty. */
int qvo_hnsw_load_flat(qvo_hnsw* h, uint32_t n, const float* rows, const uint32_t* deg, const uint32_t* links, uint32_t stride, uint32_t entry) {
    if (!h || h->n_nodes != 0 || n == 0 || entry >= n) return -1;
    h->nodes = (node_t*)calloc(n, sizeof(node_t)); h->cap_nodes = n;
    for (uint32_t i = 0; i < n; i++) {
        node_t* nd = &h->nodes[i];
        nd->vec = (float*)(rows + (size_t)i * h->dim); nd->borrowed = 1; nd->level = 0; nd->alive = 1;
        nd->conn = (uint32_t**)calloc(1, sizeof(uint32_t*)); nd->conn_len = (uint32_t*)calloc(1, sizeof(uint32_t)); nd->conn_cap = (uint32_t*)calloc(1, sizeof(uint32_t));
        uint32_t d = deg[i] < stride ? deg[i] : stride;
        nd->conn[0] = (uint32_t*)malloc((d ? d : 1) * sizeof(uint32_t));
        memcpy(nd->conn[0], links + (size_t)i * stride, d * sizeof(uint32_t));
        nd->conn_len[0] = d; nd->conn_cap[0] = d ? d : 1;
    }
    h->n_nodes = n; h->size = n; h->entry = entry; h->cur_level = 0;
    return 0;
}

/* Delete, hnsw.go:741-842 */
int qvo_hnsw_delete(qvo_hnsw* h, uint32_t idx) {
    if (!node_ok(h, idx)) return -1;                                  /* :745-755 */
    node_t* nd = &h->nodes[idx];
    for (int level = 0; level <= nd->level; level++) {                /* :762 */
        uint32_t n = nd->conn_len[level];
        uint32_t* snap = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));   /* Go ranges over the slice as it was */
        if (n) memcpy(snap, nd->conn[level], n * sizeof(uint32_t));
        for (uint32_t j = 0; j < n; j++) {
            uint32_t ci = snap[j];
            if (!node_ok(h, ci)) continue;                            /* :770-772 */
            node_t* cn = &h->nodes[ci];
            if (level < cn->level + 1) {                              /* :778 */
                uint32_t w = 0;
                for (uint32_t t = 0; t < cn->conn_len[level]; t++) if (cn->conn[level][t] != idx) cn->conn[level][w++] = cn->conn[level][t];
                cn->conn_len[level] = w;
            }
        }
        free(snap);
    }
    if (h->entry == idx) {                                            /* :796 */
        if (h->n_nodes == 1) { h->entry = 0; h->cur_level = -1; }     /* :797-800 */
        else {
            int found = 0;
            for (int level = nd->level; level >= 0 && !found; level--) {   /* :805-816 */
                if (nd->conn_len[level] > 0) {
                    uint32_t c = nd->conn[level][0];
                    if (node_ok(h, c)) { h->entry = c; h->cur_level = level; found = 1; }
                }
            }
            if (!found) {                                             /* :819-827 */
                for (uint32_t i = 0; i < h->n_nodes; i++) if (i != idx && h->nodes[i].alive) { h->entry = i; h->cur_level = h->nodes[i].level; break; }
            }
        }
    }
    nd->alive = 0;                                                    /* :832 tombstone */
    h->size--;
    return 0;
}

/* top-up sort: (Distance, VectorID) hnsw.go:699-704.  VectorID is a STRING in the reference; this restatement has no ids and
 * breaks ties by node index, which is the reference's order exactly when the ids sort like the indices (zero-padded
 * decimal ids in the tests); the host mirror (csrc/host/qvhost.cpp) compares the real id strings. */
static int topup_cmp(const void* pa, const void* pb) { return sel_cmp(pa, pb); }

/* Search, hnsw.go:602-713 */
int64_t qvo_hnsw_search(qvo_hnsw* h, const float* q, uint32_t k, uint32_t* rows_out, float* dist_out, uint64_t* n_eval_out) {
    uint64_t ev0 = h->n_eval;
    if (n_eval_out) *n_eval_out = 0;
    if (h->n_nodes == 0) return 0;                                    /* :606-608 */
    if (k == 0) return -1;                                            /* :610-612 */
    if (k > h->n_nodes) k = h->n_nodes;                               /* :615-617 (len(Nodes), tombstones included) */
    uint32_t entry = h->entry;
    if (!node_ok(h, entry)) {                                         /* :621-629 */
        uint32_t i; for (i = 0; i < h->n_nodes; i++) if (h->nodes[i].alive) { entry = i; break; }
        if (i == h->n_nodes) return 0;                                /* :632-634 */
    }
    (void)dist(h, q, h->nodes[entry].vec);                            /* :637 entryDistance (computed, then unused) */
    res_t* buf = NULL; int cap = 0;
    for (int level = h->cur_level; level > 0; level--) {              /* :649-657 */
        int n = search_layer(h, q, entry, 1, level, &buf, &cap);
        if (n <= 0) continue;
        entry = buf[0].idx;
    }
    int ef = h->efS; if (ef < (int)k) ef = (int)k;                    /* :660-663 */
    int n = search_layer(h, q, entry, ef, 0, &buf, &cap);             /* :664 */
    if (n < 0) { free(buf); return -2; }
    if (n > (int)k) n = (int)k;                                       /* :670-672 */
    if (n < (int)k) {                                                 /* :676 under-filled: exact top-up */
        res_t* all = (res_t*)malloc((size_t)h->n_nodes * sizeof(res_t)); int m = 0;
        uint8_t* have = (uint8_t*)calloc(h->n_nodes, 1);
        for (int i = 0; i < n; i++) { all[m++] = buf[i]; have[buf[i].idx] = 1; }
        for (uint32_t i = 0; i < h->n_nodes; i++) {                   /* :682-697 */
            if (!h->nodes[i].alive || have[i]) continue;
            all[m].dist = dist(h, q, h->nodes[i].vec); all[m].idx = i; m++;
        }
        qsort(all, (size_t)m, sizeof(res_t), topup_cmp);              /* :699-704 */
        if (m > (int)k) m = (int)k;
        for (int i = 0; i < m; i++) { rows_out[i] = all[i].idx; dist_out[i] = all[i].dist; }
        free(all); free(have); free(buf);
        if (n_eval_out) *n_eval_out = h->n_eval - ev0;
        return m;
    }
    for (int i = 0; i < n; i++) { rows_out[i] = buf[i].idx; dist_out[i] = buf[i].dist; }
    free(buf);
    if (n_eval_out) *n_eval_out = h->n_eval - ev0;
    return n;
}
