// What a pure streaming read reaches on this GPU: every lane loads 16 bytes per step (nontemporal), no arithmetic beyond an
// xor to keep the loads alive.  Context for roofline.frac of k_flat_scan (8 TB/s is the nominal peak).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ void __launch_bounds__(256) k_read(const u4* __restrict__ p, size_t n16, unsigned* out) {
    u4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        u4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(&p[i + u * stride]);
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= v[u];
    }
    for (; i < n16; i += stride) acc ^= __builtin_nontemporal_load(&p[i]);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
// the scan's own shape: a wave owns whole 192 KiB tiles (192 steps of 1 KiB) and takes every (waves)-th tile
template <int U>
__global__ void __launch_bounds__(256) k_read_tiles(const u4* __restrict__ p, size_t n_tiles, unsigned* out) {
    u4 acc = {0, 0, 0, 0};
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t tw = (size_t)gridDim.x * 4;
    for (size_t t = (size_t)blockIdx.x * 4 + wave; t < n_tiles; t += tw) {
        const u4* q = p + t * 192 * 64 + lane;
        for (int c0 = 0; c0 < 192; c0 += U) {
            u4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(&q[(size_t)(c0 + u) * 64]);
#pragma unroll
            for (int u = 0; u < U; u++) acc ^= v[u];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
template <int U> void run_tiles(const u4* d, size_t bytes, unsigned* out, int wg_per_cu) {
    const int grid = 256 * wg_per_cu;
    const size_t n_tiles = bytes / (192 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_read_tiles<U>, dim3(grid), dim3(256), 0, 0, d, n_tiles, out);
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_read_tiles<U>, dim3(grid), dim3(256), 0, 0, d, n_tiles, out);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("tiles U=%2d  %d WG/CU (%2d waves/CU): %.3f ms per pass, %.2f TB/s\n", U, wg_per_cu, wg_per_cu * 4, ms / 5, n_tiles * 192.0 * 1024 / (ms / 5 * 1e-3) / 1e12);
}
template <int U> void run(const u4* d, size_t bytes, unsigned* out, int wg_per_cu) {
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_read<U>, dim3(grid), dim3(256), 0, 0, d, bytes / 16, out);
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_read<U>, dim3(grid), dim3(256), 0, 0, d, bytes / 16, out);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("U=%2d  %d WG/CU (%2d waves/CU): %.3f ms per pass, %.2f TB/s\n", U, wg_per_cu, wg_per_cu * 4, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
}
int main() {
    const size_t bytes = (size_t)30720 << 20;     // 30 GiB, the 10M x 768 corpus size
    u4* d; unsigned* out; hipMalloc(&d, bytes); hipMalloc(&out, 64); hipMemset(d, 1, bytes);
    for (int w : {2, 4, 8}) { run<4>(d, bytes, out, w); run<8>(d, bytes, out, w); run<16>(d, bytes, out, w); }
    for (int w : {1, 2, 3, 4}) { run_tiles<8>(d, bytes, out, w); run_tiles<16>(d, bytes, out, w); run_tiles<32>(d, bytes, out, w); }
    return 0;
}
