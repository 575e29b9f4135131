// Does a v_cvt_f64_f32 + v_fma_f64 pair cost less when fewer lanes are active?  (A searchLayer hop keeps <= 32 of the 64
// lanes busy: if the DP pipe's time followed the number of active lanes, the idle half would be free.)
// One wave per SIMD and four waves per SIMD; active lanes = 64 / 32 (low half) / 16 / 1.  Ticks per element.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* x, const double* q, double* out, unsigned long long* cyc, int n, int active) {
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    float a = x[lane]; double b = q[lane];
    unsigned long long t0 = __builtin_readcyclecounter();
    if (lane < active) {
        for (int i = 0; i < n; i += 8) {
#pragma unroll
            for (int u = 0; u < 8; u++) { float aa = a; asm volatile("" : "+v"(aa)); acc = __builtin_fma((double)aa, b, acc); }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
int main() {
    float* x; double* q; double* out; unsigned long long* cyc;
    hipMalloc(&x, 4096); hipMalloc(&q, 4096); hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 1 << 16);
    hipMemset(x, 0, 4096); hipMemset(q, 0, 4096);
    const int n = 8192;
    for (int w = 0; w < 20; w++) hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, x, q, out, cyc, n, 64);
    hipDeviceSynchronize();
    for (int threads : {256, 1024}) {
        for (int active : {64, 32, 16, 1}) {
            hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, x, q, out, cyc, n, active);
            hipDeviceSynchronize();
            static unsigned long long h[4096]; const int nw = 256 * threads / 64; hipMemcpy(h, cyc, nw * 8, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < nw; i++) s += h[i];
            printf("%d wave(s)/SIMD, %2d active lanes: %.2f ticks per element (cvt + fma) per wave\n", threads / 256, active, s / nw / n);
        }
    }
    return 0;
}
