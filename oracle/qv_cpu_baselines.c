/*
 * qv_cpu_baselines.c — CPU baselines beyond the single-thread faithful scan.  TEST / BENCH INFRASTRUCTURE ONLY (see
 * qv_oracle.h): timed by bench.py's cpu_baseline leg beside the GPU numbers, never on a product path.
 *
 *  1. qvo_faithful_search_many: what HybridIndex.BatchSearch does with T cores (pkg/hybrid/hybrid_index.go:703-705 starts
 *     one goroutine per query): T threads, each running whole reference-faithful ExactIndex.Search calls
 *     (qvo_faithful_search: rows behind a string-keyed map, one scalar f64 distance call per row, full sort).
 *  2. qvo_opt_cosine_scan: NOT the reference — the scan a CPU engineer would write for the same job, so that the GPU
 *     speed-up is not flattered by the reference's scalar loop: contiguous rows, cached row norms, 16-wide float32 FMA
 *     lanes (AVX-512 where the host has it, chosen at run time), rows split over T threads, per-thread partial top-k,
 *     merged.  float32 accumulation: distances are NOT bit-identical to the reference's (bench.py reports the overlap
 *     of its top-k with the exact one).
 * Built with -O3 (vectorisation allowed) in its own translation unit; the faithful restatement keeps -fno-tree-vectorize.
 */
#define _GNU_SOURCE
#include "qv_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sched.h>
#include <time.h>
#include <unistd.h>

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

/* ---- 1. one faithful search per thread at a time ------------------------------------------------------------------ */
typedef struct { qvo_faithful* f; const float* qs; uint32_t dim, nq, k; volatile uint32_t* next; float* dist_out; } many_t;
static void* many_worker(void* p) {
    many_t* a = (many_t*)p;
    const char** ids = (const char**)malloc((size_t)a->k * sizeof(char*));
    float* d = (float*)malloc((size_t)a->k * sizeof(float));
    for (;;) {
        uint32_t q = __atomic_fetch_add(a->next, 1, __ATOMIC_RELAXED);     /* goroutines are scheduled as cores free up */
        if (q >= a->nq) break;
        int64_t got = qvo_faithful_search(a->f, a->qs + (size_t)q * a->dim, a->k, ids, d);
        if (a->dist_out) for (int64_t i = 0; i < got; i++) a->dist_out[(size_t)q * a->k + i] = d[i];
    }
    free(ids); free(d);
    return NULL;
}
double qvo_faithful_search_many(qvo_faithful* f, uint32_t dim, const float* queries, uint32_t nq, uint32_t k, int threads, float* dist_out) {
    if (threads < 1) threads = 1;
    volatile uint32_t next = 0;
    many_t a = { f, queries, dim, nq, k, &next, dist_out };
    pthread_t* th = (pthread_t*)malloc((size_t)threads * sizeof(pthread_t));
    double t0 = now_s();
    for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, many_worker, &a);
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    double dt = now_s() - t0;
    free(th);
    return dt;
}

/* ---- 2. optimised scan -------------------------------------------------------------------------------------------- */
typedef float v16f __attribute__((vector_size(64), aligned(4)));

__attribute__((target_clones("avx512f", "avx2", "default")))
static float dot16(const float* a, const float* b, uint32_t dim) {
    v16f acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    uint32_t i = 0;
    for (; i + 64 <= dim; i += 64) {
        acc0 += *(const v16f*)(a + i) * *(const v16f*)(b + i);
        acc1 += *(const v16f*)(a + i + 16) * *(const v16f*)(b + i + 16);
        acc2 += *(const v16f*)(a + i + 32) * *(const v16f*)(b + i + 32);
        acc3 += *(const v16f*)(a + i + 48) * *(const v16f*)(b + i + 48);
    }
    for (; i + 16 <= dim; i += 16) acc0 += *(const v16f*)(a + i) * *(const v16f*)(b + i);
    v16f acc = (acc0 + acc1) + (acc2 + acc3);
    float s = 0.f;
    for (int j = 0; j < 16; j++) s += acc[j];
    for (; i < dim; i++) s += a[i] * b[i];
    return s;
}

/* The corpus as a tuned CPU scan would hold it: one contiguous slice per thread, allocated and first-touched BY that thread
 * while pinned to its core (NUMA-local on a multi-socket host), row norms cached.  Threads are pinned the same way when
 * they scan, meet at a spinning sense-reversing barrier twice per query, and keep a partial top-k each. */
typedef struct { float* rows; float* norms; uint32_t lo, hi; } slice_t;
struct qvo_opt { uint32_t n, dim; int threads; slice_t* sl; };

static void pin_to(int tid) {
    long nc = sysconf(_SC_NPROCESSORS_ONLN);
    if (nc <= 0) return;
    cpu_set_t set; CPU_ZERO(&set); CPU_SET((int)(tid % nc), &set);
    (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
}

typedef struct { struct qvo_opt* o; const float* src; int tid; } prep_t;
static void* prep_worker(void* p) {
    prep_t* a = (prep_t*)p; struct qvo_opt* o = a->o; slice_t* s = &o->sl[a->tid];
    pin_to(a->tid);
    const size_t cnt = (size_t)(s->hi - s->lo);
    s->rows = (float*)malloc((cnt ? cnt : 1) * o->dim * sizeof(float));
    s->norms = (float*)malloc((cnt ? cnt : 1) * sizeof(float));
    memcpy(s->rows, a->src + (size_t)s->lo * o->dim, cnt * o->dim * sizeof(float));
    for (size_t r = 0; r < cnt; r++) s->norms[r] = sqrtf(dot16(s->rows + r * o->dim, s->rows + r * o->dim, o->dim));
    return NULL;
}
qvo_opt* qvo_opt_create(const float* rows, uint32_t n, uint32_t dim, int threads) {
    if (threads < 1) threads = 1;
    struct qvo_opt* o = (struct qvo_opt*)calloc(1, sizeof(*o));
    o->n = n; o->dim = dim; o->threads = threads; o->sl = (slice_t*)calloc((size_t)threads, sizeof(slice_t));
    pthread_t* th = (pthread_t*)malloc((size_t)threads * sizeof(pthread_t)); prep_t* args = (prep_t*)malloc((size_t)threads * sizeof(prep_t));
    for (int t = 0; t < threads; t++) {
        o->sl[t].lo = (uint32_t)((uint64_t)n * (uint64_t)t / (uint64_t)threads); o->sl[t].hi = (uint32_t)((uint64_t)n * (uint64_t)(t + 1) / (uint64_t)threads);
        args[t] = (prep_t){ o, rows, t };
        pthread_create(&th[t], NULL, prep_worker, &args[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th); free(args);
    return o;
}
void qvo_opt_destroy(qvo_opt* o) {
    if (!o) return;
    for (int t = 0; t < o->threads; t++) { free(o->sl[t].rows); free(o->sl[t].norms); }
    free(o->sl); free(o);
}

typedef struct { volatile int count; volatile int sense; int n; } spin_t;
static void spin_wait(spin_t* b, int* local) {
    *local = !*local;
    if (__atomic_add_fetch(&b->count, 1, __ATOMIC_ACQ_REL) == b->n) { b->count = 0; __atomic_store_n(&b->sense, *local, __ATOMIC_RELEASE); }
    else while (__atomic_load_n(&b->sense, __ATOMIC_ACQUIRE) != *local) __builtin_ia32_pause();
}

typedef struct {
    struct qvo_opt* o; const float* qs; uint32_t nq, k; int tid;
    float* part_d; uint32_t* part_r;         /* [threads][k] */
    uint32_t* rows_out; float* dist_out;     /* [nq][k] */
    spin_t* bar; double* t_scan;             /* wall seconds between the first and the last barrier, by thread 0 */
} opt_t;

static void topk_insert(float* d, uint32_t* r, uint32_t k, uint32_t* len, float x, uint32_t row) {
    if (*len == k && !(x < d[k - 1])) return;
    uint32_t i = *len < k ? (*len)++ : k - 1;
    while (i > 0 && (d[i - 1] > x || (d[i - 1] == x && r[i - 1] > row))) { d[i] = d[i - 1]; r[i] = r[i - 1]; i--; }
    d[i] = x; r[i] = row;
}

static void* opt_worker(void* p) {
    opt_t* a = (opt_t*)p; struct qvo_opt* o = a->o; const slice_t* s = &o->sl[a->tid];
    pin_to(a->tid);
    int sense = 0;
    const uint32_t k = a->k, dim = o->dim;
    float* d = a->part_d + (size_t)a->tid * k; uint32_t* r = a->part_r + (size_t)a->tid * k;
    spin_wait(a->bar, &sense);                                         /* everyone is up: the clock starts */
    double t0 = a->tid == 0 ? now_s() : 0.0;
    for (uint32_t q = 0; q < a->nq; q++) {
        const float* qv = a->qs + (size_t)q * dim;
        const float qn = sqrtf(dot16(qv, qv, dim));
        uint32_t len = 0;
        for (uint32_t i = 0; i < s->hi - s->lo; i++) {
            const float den = qn * s->norms[i];
            float dist = 1.0f;
            if (den != 0.0f) { float sim = dot16(qv, s->rows + (size_t)i * dim, dim) / den; if (sim > 1.f) sim = 1.f; else if (sim < -1.f) sim = -1.f; dist = 1.0f - sim; }
            topk_insert(d, r, k, &len, dist, s->lo + i);
        }
        for (uint32_t i = len; i < k; i++) { d[i] = INFINITY; r[i] = 0xFFFFFFFFu; }
        spin_wait(a->bar, &sense);
        if (a->tid == 0) {                                             /* merge the per-thread lists */
            float* od = a->dist_out + (size_t)q * k; uint32_t* orow = a->rows_out + (size_t)q * k;
            uint32_t olen = 0;
            for (int t = 0; t < o->threads; t++)
                for (uint32_t i = 0; i < k; i++) if (a->part_r[(size_t)t * k + i] != 0xFFFFFFFFu) topk_insert(od, orow, k, &olen, a->part_d[(size_t)t * k + i], a->part_r[(size_t)t * k + i]);
            for (uint32_t i = olen; i < k; i++) { od[i] = INFINITY; orow[i] = 0xFFFFFFFFu; }
        }
        spin_wait(a->bar, &sense);
    }
    if (a->tid == 0) *a->t_scan = now_s() - t0;
    return NULL;
}

/* nq queries one after another, each scanned by all the threads; returns the wall seconds of the scans (thread start-up excluded) */
double qvo_opt_cosine_scan(qvo_opt* o, const float* queries, uint32_t nq, uint32_t k, uint32_t* rows_out, float* dist_out) {
    const int threads = o->threads;
    spin_t bar = { 0, 0, threads };
    float* pd = (float*)malloc((size_t)threads * k * sizeof(float)); uint32_t* pr = (uint32_t*)malloc((size_t)threads * k * sizeof(uint32_t));
    opt_t* args = (opt_t*)malloc((size_t)threads * sizeof(opt_t));
    pthread_t* th = (pthread_t*)malloc((size_t)threads * sizeof(pthread_t));
    double t_scan = 0.0;
    for (int t = 0; t < threads; t++) {
        args[t] = (opt_t){ o, queries, nq, k, t, pd, pr, rows_out, dist_out, &bar, &t_scan };
        pthread_create(&th[t], NULL, opt_worker, &args[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(pd); free(pr); free(args); free(th);
    return t_scan;
}

/* which clone of the inner loop this host runs: 512 / 256 / 128-bit lanes */
int qvo_opt_simd_bits(void) {
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512f")) return 512;
    if (__builtin_cpu_supports("avx2")) return 256;
    return 128;
}
