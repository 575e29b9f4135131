/* Per-call latency of qv_index_search (host pointers) on small collections, from plain C — what a cgo caller sees, without
 * Python's per-call allocations:  ExactIndex.Search at the sizes the reference itself benchmarks (1000 x 64: 38 us on a laptop
 * core, final_bench.txt:28) and BASELINE configs[0] (10k x 128).
 *   gcc -O2 -std=c11 -I include tools/ubench/abi_latency.c -L quiver_amd/lib -lqv -lm -lpthread -Wl,-rpath,$PWD/quiver_amd/lib -o /tmp/abi_latency
 *   /tmp/abi_latency [rows=10000] [dim=128] [k=10] [calls=20000] [threads=8]
 * Prints p50 / p99 / mean per call for one caller, and the aggregate rate of `threads` concurrent callers (the reference searches
 * under a read lock: many callers at once, collection.go:647). */
#define _POSIX_C_SOURCE 200809L
#include "qv.h"
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_us(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }
static int cmp(const void* a, const void* b) { double x = *(const double*)a, y = *(const double*)b; return x < y ? -1 : x > y; }

static qv_index* g_idx; static float* g_q; static int g_dim, g_k, g_calls;
static void* worker(void* arg) {
    (void)arg;
    uint32_t rows[64], count; float dist[64];
    for (int i = 0; i < g_calls; i++) qv_index_search(g_idx, g_q + (size_t)(i % 64) * g_dim, 1, (uint32_t)g_k, rows, dist, &count);
    return NULL;
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 10000, dim = argc > 2 ? atoi(argv[2]) : 128, k = argc > 3 ? atoi(argv[3]) : 10;
    const int calls = argc > 4 ? atoi(argv[4]) : 20000, threads = argc > 5 ? atoi(argv[5]) : 8;
    if (k > 64) return 2;
    qv_index* idx = NULL;
    if (qv_index_create(&idx, (uint32_t)dim, QV_COSINE, 0, 0) != QV_OK) { fprintf(stderr, "create: %s\n", qv_last_error()); return 1; }
    if (qv_index_add_synthetic(idx, 20260424, 0, (uint32_t)rows, NULL) != QV_OK) { fprintf(stderr, "add: %s\n", qv_last_error()); return 1; }
    float* q = malloc(sizeof(float) * 64 * (size_t)dim);
    for (int i = 0; i < 64; i++) qv_index_get_row(idx, (uint32_t)(i * 7 % rows), q + (size_t)i * dim);
    uint32_t out_rows[64], count = 0; float out_dist[64];
    for (int i = 0; i < 200; i++) qv_index_search(idx, q + (size_t)(i % 64) * dim, 1, (uint32_t)k, out_rows, out_dist, &count);
    if (count != (uint32_t)k || out_rows[0] != (uint32_t)(199 % 64 * 7 % rows)) { fprintf(stderr, "unexpected result (row %u)\n", out_rows[0]); return 3; }
    double* t = malloc(sizeof(double) * (size_t)calls);
    const double t0 = now_us();
    for (int i = 0; i < calls; i++) {
        const double a = now_us();
        qv_index_search(idx, q + (size_t)(i % 64) * dim, 1, (uint32_t)k, out_rows, out_dist, &count);
        t[i] = now_us() - a;
    }
    const double total = now_us() - t0;
    qsort(t, (size_t)calls, sizeof(double), cmp);
    g_idx = idx; g_q = q; g_dim = dim; g_k = k; g_calls = calls / 4;
    pthread_t th[64];
    const double m0 = now_us();
    for (int i = 0; i < threads && i < 64; i++) pthread_create(&th[i], NULL, worker, NULL);
    for (int i = 0; i < threads && i < 64; i++) pthread_join(th[i], NULL);
    const double mt = now_us() - m0;
    printf("{\"rows\": %d, \"dim\": %d, \"k\": %d, \"calls\": %d, \"p50_us\": %.2f, \"p99_us\": %.2f, \"mean_us\": %.2f, \"callers\": %d, \"aggregate_qps\": %.0f, \"us_per_search_aggregate\": %.2f}\n",
           rows, dim, k, calls, t[calls / 2], t[(int)(calls * 0.99)], total / calls, threads, (double)threads * g_calls / (mt * 1e-6), mt / ((double)threads * g_calls));
    qv_index_destroy(idx);
    return 0;
}
