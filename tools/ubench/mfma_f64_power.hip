// Does the f64 matrix rate depend on the operand values (power management)?  v_mfma_f64_16x16x4_f64, 4 independent chains per
// wave, 4 waves per SIMD, run for ~50 ms with (a) all-zero operands (what mfma_f64_rate.hip measures), (b) random values in [-1, 1).
// The shader clock is read off as s_memtime ticks / s_memrealtime ticks x 100 MHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, double* D, unsigned long long* cyc, int n) {
    const int lane = threadIdx.x & 63;
    double a = A[lane], b = A[64 + lane];
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; i++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        a = -a;                                    // keep the sums bounded and the operands changing
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    d4 s = c0 + c1 + c2 + c3;
    D[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
    double* A; double* D; unsigned long long* cyc;
    hipMalloc(&A, 4096); hipMalloc(&D, 1 << 24); hipMalloc(&cyc, 1 << 16);
    const int n = 40000, threads = 1024;
    for (int mode = 0; mode < 2; mode++) {
        double h[512];
        for (int i = 0; i < 512; i++) h[i] = mode ? (double)rand() / RAND_MAX * 2.0 - 1.0 : 0.0;
        hipMemcpy(A, h, sizeof h, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 4; rep++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256 * 4), dim3(threads), 0, 0, A, D, cyc, n);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c[2048]; hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
            double st = 0, sr = 0; for (int i = 0; i < 1024; i++) { st += c[2 * i]; sr += c[2 * i + 1]; }
            double flops = 2.0 * 16 * 16 * 4 * 4.0 * n * (threads / 64) * 1024;
            printf("%s operands, run %d: %.2f ms, %.1f TFLOP/s f64, s_memtime/s_memrealtime = %.3f (x 100 MHz)\n", mode ? "random" : "zero  ", rep, ms, flops / ms / 1e9, st / sr);
        }
    }
    return 0;
}
