// Sustained rate of v_mfma_f32_32x32x2_f32 (the fp32-input MFMA the batched filter uses): 4 independent accumulators per wave,
// 1..4 waves per SIMD, ~50 ms of continuous issue so the clock settles under load.  Context for the filter's fraction of the
// nominal 157.3 TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) k(const float* A, float* D, int n) {
    const int lane = threadIdx.x & 63;
    float a = A[lane], b = A[64 + lane];
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < n; i++) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    f16v s = c0 + c1 + c2 + c3;
    float t = 0; for (int i = 0; i < 16; i++) t += s[i];
    D[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
int main() {
    float* A; float* D; hipMalloc(&A, 4096); hipMalloc(&D, 1 << 24); { float h[1024]; srand(7); for (int i = 0; i < 1024; i++) h[i] = (float)rand() / RAND_MAX - 0.5f; hipMemcpy(A, h, 4096, hipMemcpyHostToDevice); }   // real operand bits: zeros draw less power
    for (int wg_per_cu : {1, 2, 4}) {
        const int grid = 256 * wg_per_cu, n = 60000;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, A, D, n);      // warm-up / clock settle
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, A, D, n);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2.0 * 32 * 32 * 2 * 4.0 * n * (double)grid * 4;
        printf("%d wave(s)/SIMD: %.1f ms, %.1f TFLOP/s fp32 MFMA\n", wg_per_cu, ms, flops / (ms * 1e-3) / 1e12);
    }
    return 0;
}
