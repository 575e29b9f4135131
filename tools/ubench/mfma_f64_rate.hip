// throughput of v_mfma_f64_16x16x4_f64 (4 independent accumulator chains per wave), 1..4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, double* D, unsigned long long* cyc, int n) {
    const int lane = threadIdx.x & 63;
    double a = A[lane], b = A[64 + lane];
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    d4 s = c0 + c1 + c2 + c3;
    D[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double* A; double* D; unsigned long long* cyc;
    hipMalloc(&A, 4096); hipMalloc(&D, 1 << 24); hipMalloc(&cyc, 1 << 16); hipMemset(A, 0, 4096);
    const int n = 2000;
    for (int w = 0; w < 50; w++) hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, A, D, cyc, n);
    hipDeviceSynchronize();
    for (int threads : {64, 256, 512, 1024}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256 * 4), dim3(threads), 0, 0, A, D, cyc, n);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[1024]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 1024; i++) s += h[i];
        double flops = 2.0 * 16 * 16 * 4 * 4.0 * n * (threads / 64) * 1024;
        printf("%4d threads/WG x 1024 WGs: %.1f ticks per MFMA per wave, %.2f ms, %.1f TFLOP/s f64\n", threads, s / 1024 / (4.0 * n), ms, flops / ms / 1e9);
    }
    return 0;
}
