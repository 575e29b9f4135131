/*
 * qv_oracle.c — CPU restatement of the reference's distance + exact-scan path.
 *
 * TEST INFRASTRUCTURE ONLY (see qv_oracle.h).  Plain C, scalar, element order and
 * precision exactly as the reference's Go code; built with
 *   gcc -O2 -fno-fast-math -ffp-contract=off -fno-tree-vectorize
 * so the compiler neither reassociates nor fuses nor vectorises (the Go compiler
 * does none of those on amd64, which is what the reference's Dockerfile builds).
 *
 * Parity pinning: tests/test_oracle_kats.py runs every literal known-answer test
 * of the reference (tests/golden/ref_kats.json, transcribed from
 * pkg/vectortypes/distances_test.go, pkg/hnsw/hnsw_test.go, pkg/hybrid/exact_test.go,
 * ...) through these functions, and cross-checks them against an independent
 * numpy restatement (oracle/oracle_np.py).
 *
 * Note on fused multiply-add: for QV_COSINE / QV_L2 / QV_DOT / QV_L1 every product
 * float64(a)*float64(b) of two float32 values is exact in float64, so fusing or
 * not fusing (Go fuses on arm64, not on amd64) cannot change a bit.  For the
 * all-float32 forms (QV_L2SQ and the pkg/hnsw *_F32 variants) the reference is
 * platform-dependent; this restatement is the amd64 (unfused) one.
 */
#define _POSIX_C_SOURCE 200809L
#include "qv_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ distances --- */

/* pkg/vectortypes/distances.go:12-40 */
static float cosine_f64(const float* a, const float* b, uint32_t n) {
    double dot = 0.0, ma = 0.0, mb = 0.0;
    for (uint32_t i = 0; i < n; i++) {               /* :18-22, element order */
        dot += (double)a[i] * (double)b[i];
        ma  += (double)a[i] * (double)a[i];
        mb  += (double)b[i] * (double)b[i];
    }
    if (ma == 0.0 || mb == 0.0) return 1.0f;          /* :25-27 */
    double sim = dot / (sqrt(ma) * sqrt(mb));         /* :30 */
    if (sim > 1.0) sim = 1.0; else if (sim < -1.0) sim = -1.0; /* :32-36 */
    return (float)(1.0 - sim);                        /* :39 */
}

/* distances.go:43-55 — subtract in float32, widen, square-accumulate in float64 */
static float l2_f64(const float* a, const float* b, uint32_t n) {
    double sum = 0.0;
    for (uint32_t i = 0; i < n; i++) {
        volatile float df = a[i] - b[i];              /* float32 subtraction (:50) */
        double diff = (double)df;
        sum += diff * diff;
    }
    return (float)sqrt(sum);
}

/* distances.go:60-72 — all float32 */
static float l2sq_f32(const float* a, const float* b, uint32_t n) {
    float sum = 0.0f;
    for (uint32_t i = 0; i < n; i++) {
        float diff = a[i] - b[i];
        volatile float sq = diff * diff;              /* rounded product, then rounded add */
        sum = sum + sq;
    }
    return sum;
}

/* distances.go:77-90 */
static float dot_f64(const float* a, const float* b, uint32_t n) {
    double dot = 0.0;
    for (uint32_t i = 0; i < n; i++) dot += (double)a[i] * (double)b[i];
    return (float)(1.0 - dot);
}

/* distances.go:93-104 */
static float l1_f64(const float* a, const float* b, uint32_t n) {
    double sum = 0.0;
    for (uint32_t i = 0; i < n; i++) {
        volatile float df = a[i] - b[i];
        sum += fabs((double)df);
    }
    return (float)sum;
}

/* pkg/hnsw/adapter.go:105-136 */
static float cosine_f32(const float* a, const float* b, uint32_t n) {
    float dot = 0.0f, na = 0.0f, nb = 0.0f;
    for (uint32_t i = 0; i < n; i++) {
        volatile float p0 = a[i] * b[i]; dot = dot + p0;
        volatile float p1 = a[i] * a[i]; na = na + p1;
        volatile float p2 = b[i] * b[i]; nb = nb + p2;
    }
    if (na == 0.0f || nb == 0.0f) return 1.0f;
    /* :128  float32(math.Sqrt(float64(normA))) * float32(math.Sqrt(float64(normB))) */
    float sa = (float)sqrt((double)na), sb = (float)sqrt((double)nb);
    volatile float den = sa * sb;
    float sim = dot / den;
    if (sim > 1.0f) sim = 1.0f; else if (sim < -1.0f) sim = -1.0f;
    return 1.0f - sim;
}

/* adapter.go:139-151 */
static float l2_f32(const float* a, const float* b, uint32_t n) {
    float sum = 0.0f;
    for (uint32_t i = 0; i < n; i++) {
        float diff = a[i] - b[i];
        volatile float sq = diff * diff;
        sum = sum + sq;
    }
    return (float)sqrt((double)sum);
}

/* adapter.go:154-165 */
static float dot_f32(const float* a, const float* b, uint32_t n) {
    float dot = 0.0f;
    for (uint32_t i = 0; i < n; i++) { volatile float p = a[i] * b[i]; dot = dot + p; }
    return 1.0f - dot;
}

/* index/arrow_hnsw.go:124-132 — ArrowHNSWIndex.Search re-scores every hit itself: both
 * vectors widened to float64 (arrow_hnsw.go:73-75, 109-112), d = q[j]-vec[j] in float64,
 * dist += d*d sequentially (mul then add: unfused on amd64), float32(dist) */
static float l2sq_f64(const float* a, const float* b, uint32_t n) {
    double dist = 0.0;
    for (uint32_t i = 0; i < n; i++) {
        double d = (double)a[i] - (double)b[i];
        volatile double sq = d * d;
        dist = dist + sq;
    }
    return (float)dist;
}

typedef float (*dist_fn)(const float*, const float*, uint32_t);
static dist_fn metric_fn(int metric) {
    switch (metric) {
        case QVO_COSINE: return cosine_f64;
        case QVO_L2: return l2_f64;
        case QVO_L2SQ: return l2sq_f32;
        case QVO_DOT: return dot_f64;
        case QVO_L1: return l1_f64;
        case QVO_COSINE_F32: return cosine_f32;
        case QVO_L2_F32: return l2_f32;
        case QVO_DOT_F32: return dot_f32;
        case QVO_L2SQ_F64: return l2sq_f64;
        default: return cosine_f64;                   /* types.go:46-47: unknown -> cosine */
    }
}

float qvo_distance(int metric, const float* a, const float* b, uint32_t dim) {
    return metric_fn(metric)(a, b, dim);
}

/* ------------------------------------------------------------------ generator --- */

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* element (row, col) before normalisation: an Irwin-Hall(4) integer in
 * [-131070, 131070] — bell-shaped, exactly representable, no transcendental */
static inline int32_t gen_int(uint64_t row_key, uint32_t col) {
    uint64_t h = splitmix64(row_key + (uint64_t)col);
    int32_t s = (int32_t)(h & 0xFFFF) + (int32_t)((h >> 16) & 0xFFFF) +
                (int32_t)((h >> 32) & 0xFFFF) + (int32_t)(h >> 48);
    return s - 131070;
}

void qvo_gen_rows(uint64_t seed, uint64_t row0, uint32_t n, uint32_t dim, float* out) {
    for (uint32_t r = 0; r < n; r++) {
        uint64_t row_key = splitmix64(seed ^ ((row0 + r) * 0xD1342543DE82EF95ull));
        double sumsq = 0.0;                           /* exact: integers < 2^53 */
        for (uint32_t c = 0; c < dim; c++) {
            double v = (double)gen_int(row_key, c);
            sumsq += v * v;
        }
        double norm = sumsq > 0.0 ? sqrt(sumsq) : 1.0;
        float* o = out + (size_t)r * dim;
        for (uint32_t c = 0; c < dim; c++) o[c] = (float)((double)gen_int(row_key, c) / norm);
    }
}

/* ------------------------------------------------------------------ exact scan --- */

typedef struct { float dist; uint32_t row; uint32_t aux; float aux_f; } rec_t;

/* total order used everywhere: distance ascending (NaN last), then row ascending */
static int rec_cmp(const void* pa, const void* pb) {
    const rec_t* a = (const rec_t*)pa; const rec_t* b = (const rec_t*)pb;
    int an = a->dist != a->dist, bn = b->dist != b->dist;
    if (an != bn) return an - bn;
    if (!an) { if (a->dist < b->dist) return -1; if (a->dist > b->dist) return 1; }
    return (a->row > b->row) - (a->row < b->row);
}

void qvo_all_distances(int metric, const float* rows, uint32_t n, uint32_t dim, const float* query, float* dist_out) {
    dist_fn f = metric_fn(metric);
    for (uint32_t i = 0; i < n; i++) dist_out[i] = f(query, rows + (size_t)i * dim, dim); /* exact.go:116 distFunc(query, vec) */
}

int64_t qvo_exact_search(int metric, const float* rows, const uint8_t* alive, uint32_t n, uint32_t dim,
                         const float* query, uint32_t k, uint32_t* rows_out, float* dist_out) {
    uint32_t live = 0;
    for (uint32_t i = 0; i < n; i++) live += (!alive || alive[i]);
    if (live == 0) return 0;                          /* exact.go:96-98 */
    if (k == 0) return -1;                            /* exact.go:104-106 */
    if (k > live) k = live;                           /* exact.go:109-111 */
    rec_t* recs = (rec_t*)malloc((size_t)live * sizeof(rec_t));
    dist_fn f = metric_fn(metric);
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; i++) {                /* exact.go:115-121 */
        if (alive && !alive[i]) continue;
        recs[m].dist = f(query, rows + (size_t)i * dim, dim);
        recs[m].row = i; m++;
    }
    qsort(recs, live, sizeof(rec_t), rec_cmp);        /* exact.go:124 (full sort) */
    for (uint32_t i = 0; i < k; i++) { rows_out[i] = recs[i].row; dist_out[i] = recs[i].dist; } /* :127-129 */
    free(recs);
    return k;
}

/* stable sort key for the re-rank: (score asc, id asc) — hybrid_index.go:552-560 */
static int rerank_cmp(const void* pa, const void* pb) {
    const rec_t* a = (const rec_t*)pa; const rec_t* b = (const rec_t*)pb;
    if (a->dist == b->dist) return (a->aux > b->aux) - (a->aux < b->aux);
    if (a->dist < b->dist) return -1;
    if (a->dist > b->dist) return 1;
    /* NaN involved: keep input order (what a stable sort with an always-false less does) */
    return (a->row > b->row) - (a->row < b->row);
}

int64_t qvo_exact_search_negative(int metric, const float* rows, const uint8_t* alive, uint32_t n, uint32_t dim,
                                  const float* query, const float* negative, float neg_weight, uint32_t k,
                                  const uint32_t* id_rank, uint32_t* rows_out, float* dist_out) {
    uint32_t live = 0;
    for (uint32_t i = 0; i < n; i++) live += (!alive || alive[i]);
    if (k == 0 && live) return -1;
    uint32_t retrieve = 2 * k > 30 ? 2 * k : 30;      /* hybrid_index.go:518 maxInt(2*k, 30) */
    if (retrieve > live) retrieve = live;             /* :519-521 */
    if (retrieve == 0) return 0;
    uint32_t* r = (uint32_t*)malloc(retrieve * sizeof(uint32_t));
    float* d = (float*)malloc(retrieve * sizeof(float));
    int64_t got = qvo_exact_search(metric, rows, alive, n, dim, query, retrieve, r, d);
    rec_t* recs = (rec_t*)malloc((size_t)got * sizeof(rec_t));
    dist_fn f = metric_fn(metric);
    for (int64_t i = 0; i < got; i++) {
        float dneg = f(rows + (size_t)r[i] * dim, negative, dim);   /* :543 distFunc(vector, negExample) */
        volatile float prod = neg_weight * dneg;                    /* :549 float32 arithmetic */
        recs[i].dist = d[i] - prod;
        recs[i].row = (uint32_t)i;                                  /* input position (stability) */
        recs[i].aux = id_rank ? id_rank[r[i]] : r[i];
        recs[i].aux_f = 0;
    }
    /* qsort is not stable; the comparator is a total order on (score, id) with ids unique,
     * so stability only matters for NaN, handled by the input-position fallback */
    qsort(recs, (size_t)got, sizeof(rec_t), rerank_cmp);
    int64_t outn = got < (int64_t)k ? got : (int64_t)k;             /* :566-568 */
    for (int64_t i = 0; i < outn; i++) { rows_out[i] = r[recs[i].row]; dist_out[i] = recs[i].dist; }
    free(recs); free(r); free(d);
    return outn;
}

/* -------------------------------------------------- reference-faithful baseline --- */

typedef struct { char* id; float* vec; } slot_t;
struct qvo_faithful {
    int metric; uint32_t dim; dist_fn fn;
    slot_t* slots; uint32_t cap; uint32_t size;
};
typedef struct { const char* id; float dist; } idrec_t;

static uint64_t str_hash(const char* s) { uint64_t h = 1469598103934665603ull; while (*s) { h ^= (unsigned char)*s++; h *= 1099511628211ull; } return h; }

qvo_faithful* qvo_faithful_create(int metric, uint32_t dim) {
    qvo_faithful* f = (qvo_faithful*)calloc(1, sizeof(*f));
    f->metric = metric; f->dim = dim; f->fn = metric_fn(metric);
    f->cap = 1024; f->slots = (slot_t*)calloc(f->cap, sizeof(slot_t));
    return f;
}
void qvo_faithful_destroy(qvo_faithful* f) {
    if (!f) return;
    for (uint32_t i = 0; i < f->cap; i++) { free(f->slots[i].id); free(f->slots[i].vec); }
    free(f->slots); free(f);
}
static void faithful_put(slot_t* slots, uint32_t cap, char* id, float* vec) {
    uint32_t i = (uint32_t)(str_hash(id) & (cap - 1));
    while (slots[i].id) i = (i + 1) & (cap - 1);
    slots[i].id = id; slots[i].vec = vec;
}
int qvo_faithful_insert(qvo_faithful* f, const char* id, const float* vec) {
    uint32_t i = (uint32_t)(str_hash(id) & (f->cap - 1));
    while (f->slots[i].id) { if (!strcmp(f->slots[i].id, id)) return -1; i = (i + 1) & (f->cap - 1); } /* exact.go:48-50 */
    if ((uint64_t)(f->size + 1) * 2 > f->cap) {
        uint32_t ncap = f->cap * 2; slot_t* ns = (slot_t*)calloc(ncap, sizeof(slot_t));
        for (uint32_t j = 0; j < f->cap; j++) if (f->slots[j].id) faithful_put(ns, ncap, f->slots[j].id, f->slots[j].vec);
        free(f->slots); f->slots = ns; f->cap = ncap;
    }
    float* copy = (float*)malloc(f->dim * sizeof(float));           /* exact.go:53-56 per-row allocation + copy */
    memcpy(copy, vec, f->dim * sizeof(float));
    faithful_put(f->slots, f->cap, strdup(id), copy);
    f->size++;
    return 0;
}
uint32_t qvo_faithful_size(const qvo_faithful* f) { return f->size; }

static int idrec_cmp(const void* pa, const void* pb) {              /* exact.go:76 Less: Distance only */
    float a = ((const idrec_t*)pa)->dist, b = ((const idrec_t*)pb)->dist;
    return (a > b) - (a < b);
}
int64_t qvo_faithful_search(qvo_faithful* f, const float* query, uint32_t k, const char** ids_out, float* dist_out) {
    if (f->size == 0) return 0;
    if (k == 0) return -1;
    if (k > f->size) k = f->size;
    idrec_t* recs = (idrec_t*)malloc((size_t)f->size * sizeof(idrec_t));  /* exact.go:114 make(resultHeap, 0, N) */
    uint32_t m = 0;
    dist_fn volatile fn = f->fn;                                           /* indirect call per row, as in Go */
    for (uint32_t i = 0; i < f->cap; i++) {                                /* map iteration order */
        if (!f->slots[i].id) continue;
        recs[m].id = f->slots[i].id; recs[m].dist = fn(query, f->slots[i].vec, f->dim); m++;
    }
    qsort(recs, m, sizeof(idrec_t), idrec_cmp);                            /* exact.go:124 */
    for (uint32_t i = 0; i < k; i++) { ids_out[i] = recs[i].id; dist_out[i] = recs[i].dist; }
    free(recs);
    return k;
}
