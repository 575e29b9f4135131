/* Pure C11 consumer of include/qv.h, the way a cgo preamble sees it: proves the header is C (not C++), links against
 * libqv.so alone, and — when a GPU is present — runs create / add / search / remove / destroy through the C ABI.
 * Without a GPU every entry point must fail LOUDLY with QV_ERR_NO_DEVICE (no CPU path).
 *   gcc -std=c11 -Wall -Werror -I include tests/c/abi_smoke.c -L quiver_amd/lib -lqv -Wl,-rpath,$PWD/quiver_amd/lib -o abi_smoke */
#include "qv.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(void) {
    if (qv_abi_version() != QV_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 2; }
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
    printf("ok: [%u %u] d0=%g\n", 1u, out_rows[1], out_dist[0]);
    return 0;
}
