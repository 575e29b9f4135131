// Can the float32 -> float64 widening of MFMA operands leave the DP pipe?  v_cvt_f64_f32 costs DP-pipe time that adds to the
// f64 MFMAs' (mfma_f64_mix.hip).  The same widening done with 32-bit integer VALU ops (exact for normal numbers and zero;
// denormal / inf / nan inputs need the real convert) — does IT overlap with the wave's own f64 MFMAs?
// Per iteration: 4 independent v_mfma_f64_16x16x4_f64 + NV widenings (mode 0: v_cvt_f64_f32, mode 1: integer ops).
// Reported: ticks per iteration per wave for 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double widen_int(float f) {
    const unsigned x = __float_as_uint(f);
    const unsigned mag = x & 0x7FFFFFFFu;
    unsigned hi = (x & 0x80000000u) | ((mag >> 3) + 0x38000000u);
    if (mag == 0) hi = x & 0x80000000u;
    const unsigned lo = x << 29;
    return __hiloint2double((int)hi, (int)lo);
}
template <int NV, int MODE>
__global__ void k(const double* A, double* D, unsigned long long* cyc, int n) {
    const int lane = threadIdx.x & 63;
    double a = A[lane], b = A[64 + lane];
    float f = (float)A[128 + lane] + 1.5f;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double s = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; v++) {
            float ff = f; asm volatile("" : "+v"(ff));
            double d = MODE == 0 ? (double)ff : widen_int(ff);
            asm volatile("" : "+v"(d)); s = d;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    d4 r = c0 + c1 + c2 + c3;
    D[blockIdx.x * blockDim.x + threadIdx.x] = r[0] + r[1] + r[2] + r[3] + s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int NV, int MODE> void run(const double* A, double* D, unsigned long long* cyc) {
    const int n = 2000;
    for (int threads : {256, 512, 1024}) {
        hipLaunchKernelGGL((k<NV, MODE>), dim3(256), dim3(threads), 0, 0, A, D, cyc, n);
        hipDeviceSynchronize();
        static unsigned long long h[4096]; const int nw = 256 * threads / 64; hipMemcpy(h, cyc, nw * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < nw; i++) s += h[i];
        printf("%s NV=%2d, %d wave(s)/SIMD: %.0f ticks per iteration (4 MFMA + %d widenings) per wave\n", MODE ? "int ops " : "v_cvt   ", NV, threads / 256, s / nw / n, NV);
    }
}
int main() {
    double* A; double* D; unsigned long long* cyc;
    hipMalloc(&A, 4096); hipMalloc(&D, 1 << 24); hipMalloc(&cyc, 1 << 16); hipMemset(A, 0, 4096);
    for (int w = 0; w < 50; w++) hipLaunchKernelGGL((k<0, 0>), dim3(1024), dim3(256), 0, 0, A, D, cyc, 2000);
    hipDeviceSynchronize();
    run<0, 0>(A, D, cyc); run<6, 0>(A, D, cyc); run<6, 1>(A, D, cyc); run<12, 0>(A, D, cyc); run<12, 1>(A, D, cyc);
    return 0;
}
