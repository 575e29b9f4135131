/* Pure C11 consumer of include/qv.h, the way a cgo preamble sees it: proves the header is C (not C++), links against
 * libqv.so alone, and — when a GPU is present — runs create / add / search / remove / destroy, a device HNSW build + search and
 * a one-shard qv_sharded_* round trip through the C ABI.
 * Without a GPU every entry point must fail LOUDLY with QV_ERR_NO_DEVICE (no CPU path).
 *   gcc -std=c11 -Wall -Werror -I include tests/c/abi_smoke.c -L quiver_amd/lib -lqv -lm -Wl,-rpath,$PWD/quiver_amd/lib -o abi_smoke */
#include "qv.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

int main(void) {
    if (qv_abi_version() != QV_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 2; }
    /* the DistanceFunc contract for one pair is host code: works with or without a device.  distances_test.go:151-156 */
    { const float a[3] = {2, 0, 0}, b[3] = {3, 0, 0}; float d = 0;
      if (qv_distance_pair(QV_DOT, a, b, 3, &d) != QV_OK || d != -5.0f) { fprintf(stderr, "distance_pair: %g\n", d); return 30; } }
    qv_index* idx = NULL;
    int rc = qv_index_create(&idx, 4, QV_L2, 0, 0);
    if (qv_device_count() <= 0) {
        if (rc != QV_ERR_NO_DEVICE || idx != NULL) { fprintf(stderr, "expected QV_ERR_NO_DEVICE, got %d\n", rc); return 3; }
        printf("no device: %s\n", qv_last_error());
        return 0;
    }
    if (rc != QV_OK) { fprintf(stderr, "create: %s\n", qv_last_error()); return 4; }
    /* exact_test.go:101-156: three axis vectors, q = (0.9, 0.1, 0) -> vec1 then vec2 */
    const float rows[3][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}};
    uint32_t first = 99;
    if (qv_index_add(idx, &rows[0][0], 3, &first) != QV_OK || first != 0) { fprintf(stderr, "add: %s\n", qv_last_error()); return 5; }
    const float q[4] = {0.9f, 0.1f, 0, 0};
    uint32_t out_rows[2], count = 0; float out_dist[2];
    if (qv_index_search(idx, q, 1, 2, out_rows, out_dist, &count) != QV_OK) { fprintf(stderr, "search: %s\n", qv_last_error()); return 6; }
    if (count != 2 || out_rows[0] != 0 || out_rows[1] != 1 || !(out_dist[0] < out_dist[1])) { fprintf(stderr, "unexpected result\n"); return 7; }
    if (qv_index_search(idx, q, 1, 0, out_rows, out_dist, &count) != QV_ERR_K_NOT_POSITIVE) { fprintf(stderr, "k=0 must be refused\n"); return 8; }
    const uint32_t dead = 0;
    if (qv_index_remove(idx, &dead, 1) != QV_OK || qv_index_size(idx) != 2) { fprintf(stderr, "remove: %s\n", qv_last_error()); return 9; }
    if (qv_index_search(idx, q, 1, 2, out_rows, out_dist, &count) != QV_OK || out_rows[0] != 1) { fprintf(stderr, "search after remove\n"); return 10; }
    qv_index_destroy(idx);

    /* HNSW.Insert x n on the device (hnsw.go:266-468), then HNSW.Search (:602-713): 200 points on a circle, the query's
     * nearest neighbours are its angular neighbours */
    enum { N = 200, D = 4 };
    static float pts[N][D]; static int8_t levels[N];
    for (int i = 0; i < N; i++) { double a = 6.283185307179586 * i / N; pts[i][0] = (float)cos(a); pts[i][1] = (float)sin(a); pts[i][2] = 0; pts[i][3] = 0; levels[i] = 0; }
    qv_index* gi = NULL; qv_graph* g = NULL;
    if (qv_index_create(&gi, D, QV_L2, 0, QV_FLAG_ROWMAJOR) != QV_OK || qv_index_add(gi, &pts[0][0], N, &first) != QV_OK) { fprintf(stderr, "graph index: %s\n", qv_last_error()); return 11; }
    if (qv_graph_build(&g, gi, N, levels, 8, 16, 50, 16, 4) != QV_OK) { fprintf(stderr, "graph build: %s\n", qv_last_error()); return 12; }
    uint32_t nn = 0, ep = 0; int lvl = -2;
    if (qv_graph_info(g, &nn, NULL, NULL, NULL, &ep, &lvl) != QV_OK || nn != N || lvl != 0) { fprintf(stderr, "graph info\n"); return 13; }
    const float gq[D] = {pts[50][0], pts[50][1], 0, 0};
    uint32_t gr[3], gc = 0; float gd[3];
    if (qv_graph_search(g, gq, 1, 3, 32, gr, gd, &gc, NULL) != QV_OK || gc != 3) { fprintf(stderr, "graph search: %s\n", qv_last_error()); return 14; }
    if (gr[0] != 50 || gd[0] != 0.0f || !((gr[1] == 49 && gr[2] == 51) || (gr[1] == 51 && gr[2] == 49))) { fprintf(stderr, "graph result [%u %u %u]\n", gr[0], gr[1], gr[2]); return 15; }
    { uint64_t st[8];                                      /* how the calls were served: a lone caller runs at once, in its own context */
      if (qv_graph_coalesce_stats(g, st) != QV_OK || st[0] != 1 || st[1] + st[2] + st[3] != 0) { fprintf(stderr, "graph coalesce stats\n"); return 26; } }
    qv_graph_destroy(g); qv_index_destroy(gi);

    /* one corpus behind one handle: a single shard here, a real RCCL communicator all the same */
    qv_sharded* sh = NULL; const int devs[1] = {0};
    if (qv_sharded_create(&sh, 4, QV_L2, devs, 1, 0) != QV_OK) { fprintf(stderr, "sharded create: %s\n", qv_last_error()); return 16; }
    uint32_t gids[3];
    if (qv_sharded_add(sh, &rows[0][0], 3, gids) != QV_OK || qv_sharded_size(sh) != 3) { fprintf(stderr, "sharded add: %s\n", qv_last_error()); return 17; }
    if (qv_sharded_search(sh, q, 1, 2, out_rows, out_dist, &count) != QV_OK || count != 2 || out_rows[0] != gids[0] || out_rows[1] != gids[1]) { fprintf(stderr, "sharded search: %s\n", qv_last_error()); return 18; }
    /* the rest of core.Index on the sharded handle: update (Collection.Update), get, filtered search, a ranking deeper than
     * the fused top-k, negative example, listed-row distances */
    { const float moved[4] = {0.5f, 0.5f, 0, 0}; float back[4]; uint32_t big_rows[70], cnt70 = 0; float big_dist[70], nd[3], dr[2];
      if (qv_sharded_update(sh, gids[2], moved) != QV_OK || qv_sharded_get_row(sh, gids[2], back) != QV_OK || back[0] != 0.5f) { fprintf(stderr, "sharded update/get: %s\n", qv_last_error()); return 19; }
      if (qv_sharded_search(sh, q, 1, 70, big_rows, big_dist, &cnt70) != QV_OK || cnt70 != 3 || big_rows[0] != gids[0] || big_rows[1] != gids[2] || big_rows[3] != 0xFFFFFFFFu) { fprintf(stderr, "sharded k=70: %s\n", qv_last_error()); return 20; }
      const uint32_t sel[2] = {gids[1], gids[2]};
      if (qv_sharded_search_masked(sh, q, 1, 2, sel, 2, out_rows, out_dist, &count) != QV_OK || count != 2 || out_rows[0] != gids[2] || out_rows[1] != gids[1]) { fprintf(stderr, "sharded masked: %s\n", qv_last_error()); return 21; }
      const float neg[4] = {0, 1, 0, 0}; uint32_t nrows[3], ncnt = 0; float ndist[3];
      if (qv_sharded_search_negative(sh, q, neg, 3, nrows, ndist, nd, &ncnt) != QV_OK || ncnt != 3 || nrows[0] != gids[0] || !(fabsf(nd[0] - 1.41421354f) < 1e-6f)) { fprintf(stderr, "sharded negative: %s\n", qv_last_error()); return 22; }
      if (qv_sharded_distance_rows(sh, q, sel, 2, dr) != QV_OK || dr[0] != out_dist[1] || dr[1] != out_dist[0]) { fprintf(stderr, "sharded distance_rows: %s\n", qv_last_error()); return 23; }
      char rt[512];
      if (qv_runtime_info(rt, sizeof rt) != QV_OK || !strstr(rt, "rccl=")) { fprintf(stderr, "runtime info\n"); return 24; }
      if (qv_sharded_set_filter(sh, QV_FILTER_BF16X3) != QV_OK || qv_sharded_set_filter(sh, 9) != QV_ERR_INVALID_ARG) { fprintf(stderr, "set_filter\n"); return 25; } }
    qv_sharded_destroy(sh);
    printf("ok: [%u %u] d0=%g graph [%u %u %u]\n", 1u, out_rows[1], out_dist[0], gr[0], gr[1], gr[2]);
    return 0;
}
