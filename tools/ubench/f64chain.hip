// micro-benchmark: latency of a dependent v_fmac_f64 chain (one wave), with and without the f32->f64 convert,
// and of ds_read_b128-fed chains.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off f64chain.hip -o f64chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_chain(const float* x, const double* q, double* out, unsigned long long* cyc, int n, int mode) {
    extern __shared__ __align__(16) unsigned char smem[];
    double acc = 0.0;
    const int lane = threadIdx.x;
    // stage q (f64) and x (f32 per lane row) in LDS for mode 2
    double* ql = reinterpret_cast<double*>(smem);
    f4* xl = reinterpret_cast<f4*>(smem + 8192);
    for (int i = lane; i < 1024; i += 64) ql[i] = q[i];
    for (int i = lane; i < 256 * 8; i += 64) xl[i] = reinterpret_cast<const f4*>(x)[i % 256];
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    if (mode == 0) {          // pure dependent fma chain, operands in registers
        double a = q[lane], b = q[lane + 64];
        for (int i = 0; i < n; i += 8) {
#pragma unroll
            for (int u = 0; u < 8; u++) { acc = __builtin_fma(a, b, acc); asm volatile("" : "+v"(acc)); }
        }
    } else if (mode == 1) {   // convert + fma, operands in registers
        float a = x[lane]; double b = q[lane];
        for (int i = 0; i < n; i += 8) {
#pragma unroll
            for (int u = 0; u < 8; u++) { float aa = a; asm volatile("" : "+v"(aa)); acc = __builtin_fma((double)aa, b, acc); }
        }
    } else if (mode == 2) {   // LDS-fed: x chunk (per-lane address) + q (uniform address), like the traversal kernel
        const f4* mine = xl + (lane & 7) * 8;
        for (int c = 0; c < n / 4; c++) {
            const f4 xx = mine[(c & 7)];
            const double* qq = ql + (c & 255) * 4;
            acc = __builtin_fma((double)xx.x, qq[0], acc); acc = __builtin_fma((double)xx.y, qq[1], acc);
            acc = __builtin_fma((double)xx.z, qq[2], acc); acc = __builtin_fma((double)xx.w, qq[3], acc);
        }
    }
    else if (mode == 3) {   // throughput: 8 independent fma chains
        double a = q[lane], b = q[lane + 64];
        double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
        for (int i = 0; i < n; i += 8) {
            c0 = __builtin_fma(a, b, c0); c1 = __builtin_fma(a, b, c1); c2 = __builtin_fma(a, b, c2); c3 = __builtin_fma(a, b, c3);
            c4 = __builtin_fma(a, b, c4); c5 = __builtin_fma(a, b, c5); c6 = __builtin_fma(a, b, c6); c7 = __builtin_fma(a, b, c7);
            asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));
        }
        acc = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    } else if (mode == 4) {   // throughput: independent converts
        float a = x[lane];
        double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
        for (int i = 0; i < n; i += 8) {
            float a0 = a, a1 = a, a2 = a, a3 = a, a4 = a, a5 = a, a6 = a, a7 = a;
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            c0 = (double)a0; c1 = (double)a1; c2 = (double)a2; c3 = (double)a3; c4 = (double)a4; c5 = (double)a5; c6 = (double)a6; c7 = (double)a7;
            asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));
        }
        acc = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float* x; double* q; double* out; unsigned long long* cyc;
    hipMalloc(&x, 1 << 20); hipMalloc(&q, 1 << 20); hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 1 << 16);
    hipMemset(x, 0, 1 << 20); hipMemset(q, 0, 1 << 20);
    const int n = 4096;
    for (int w = 0; w < 300; w++) hipLaunchKernelGGL(k_chain, dim3(4096), dim3(64), 32768, 0, x, q, out, cyc, n, 2);   // clock warm-up
    hipDeviceSynchronize();
    for (int mode = 0; mode < 5; mode++)
        for (int blocks : {1, 1024, 1280}) {
            hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(64), 32768, 0, x, q, out, cyc, n, mode);
            hipDeviceSynchronize();
            unsigned long long h[4096]; hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < blocks; i++) s += h[i];
            printf("mode %d blocks %4d: %.2f cycles per element (avg over waves)\n", mode, blocks, s / blocks / n);
        }
    return 0;
}
