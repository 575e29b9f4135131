// issue rate of v_mfma_f32_32x32x16_bf16 from one wave per SIMD: 8, 4 independent accumulators, 2 accumulators used alternately
// (every instruction depends on the one two places back), and the same two-accumulator stream with 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ void __launch_bounds__(512) k(const uint4* A, float* D, unsigned long long* cyc, int n) {
    uint4 ua = A[threadIdx.x & 63], ub = A[64 + (threadIdx.x & 63)];
    bf8 a = __builtin_bit_cast(bf8, ua), b = __builtin_bit_cast(bf8, ub);
    f16v c[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) c[i][r] = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int rep = 0; rep < 24 / NACC; rep++)
#pragma unroll
            for (int i = 0; i < NACC; i++) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += c[i][r];
    D[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC> void run(const char* what, int threads, uint4* A, float* D, unsigned long long* cyc) {
    const int n = 2000;
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, A, D, cyc, n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, A, D, cyc, n);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; i++) s += h[i];
    double flops = 2.0 * 32 * 32 * 16 * 24.0 * n * (threads / 64) * 256;
    printf("%-52s %3d threads/WG: %.1f ticks per MFMA per wave, %.3f ms, %.0f TFLOP/s\n", what, threads, s / 256 / (24.0 * n), ms, flops / ms / 1e9);
}
int main() {
    uint4* A; float* D; unsigned long long* cyc;
    hipMalloc(&A, 4096); hipMalloc(&D, 1 << 24); hipMalloc(&cyc, 1 << 16); hipMemset(A, 0x3c, 4096);
    run<8>("8 accumulators in turn", 256, A, D, cyc);
    run<4>("4 accumulators in turn", 256, A, D, cyc);
    run<2>("2 accumulators alternately (dependent two back)", 256, A, D, cyc);
    run<1>("1 accumulator (every MFMA depends on the last)", 256, A, D, cyc);
    run<2>("2 accumulators alternately, 2 waves per SIMD", 512, A, D, cyc);
    run<8>("8 accumulators in turn, 2 waves per SIMD", 512, A, D, cyc);
    return 0;
}
