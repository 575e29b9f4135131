// Probe of v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks per instruction, one f64 per lane for A, B, C and D):
//   1. which lane supplies A[i][k] / B[k][j] of which block and which lane holds D[i][j] (found by matching against a host model);
//   2. whether D is the sequential chain fma(a3,b3,fma(a2,b2,fma(a1,b1,fma(a0,b0,c)))) bit for bit;
//   3. its issue rate per SIMD with 1..4 waves, alone and with two v_cvt_f64_f32 per instruction beside it (the HNSW hop's shape).
// make -C tools/ubench && tools/ubench/bin/mfma_f64_4x4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
__global__ void k_one(const double* A, const double* B, const double* C, double* D) {
    const int lane = threadIdx.x;
    D[lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[lane], B[lane], C[lane], 0, 0, 0);
}
template <int CVT>
__global__ void k_rate(const float* src, double* out, int iters) {
    const int lane = threadIdx.x & 63;
    double c0 = 0.0, c1 = 0.0;
    float f0 = src[lane], f1 = src[lane + 64];
    const double b = (double)src[lane + 128];
    double a0 = (double)f0, a1 = (double)f1;
    for (int i = 0; i < iters; i++) {
        if (CVT) { a0 = (double)f0; a1 = (double)f1; f0 += 1.0f; f1 += 1.0f; }
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b, c1, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1;
}
// the scalar form of the same work: lane = row, 4 dims per step: 4 cvt + 4 fma
__global__ void k_rate_valu(const float* src, double* out, int iters) {
    const int lane = threadIdx.x & 63;
    double c = 0.0;
    float f[4] = {src[lane], src[lane + 64], src[lane + 128], src[lane + 192]};
    const double q0 = src[1], q1 = src[2], q2 = src[3], q3 = src[4];
    for (int i = 0; i < iters; i++) {
        c = fma((double)f[0], q0, c); c = fma((double)f[1], q1, c); c = fma((double)f[2], q2, c); c = fma((double)f[3], q3, c);
        f[0] += 1.0f; f[1] += 1.0f; f[2] += 1.0f; f[3] += 1.0f;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c;
}
int main() {
    double hA[64], hB[64], hC[64], hD[64];
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 512); hipMalloc(&dD, 512);
    srand(7);
    // 1. mapping: one-hot probes.  A one-hot at lane la (value 2), B one-hot at lane lb (value 3), C = 0: D lanes holding 6 tell which (la, lb) meet
    int a_blk[64], a_i[64], a_k[64], b_blk[64], b_k[64], b_j[64], d_blk[64], d_i[64], d_j[64];
    for (int l = 0; l < 64; l++) a_blk[l] = a_i[l] = a_k[l] = b_blk[l] = b_k[l] = b_j[l] = d_blk[l] = d_i[l] = d_j[l] = -1;
    static int meets[64][64][64];
    for (int la = 0; la < 64; la++) for (int lb = 0; lb < 64; lb++) {
        for (int i = 0; i < 64; i++) { hA[i] = i == la ? 2.0 : 0.0; hB[i] = i == lb ? 3.0 : 0.0; hC[i] = 0.0; }
        hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice); hipMemcpy(dC, hC, 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; l++) meets[la][lb][l] = hD[l] == 6.0;
    }
    // A lane la and B lane lb share (block, k) iff some D lane sees them; D lanes seen = the 1 output (i of la, j of lb)
    printf("A lane -> D lanes it reaches with B lane 0..63 (first 20 A lanes):\n");
    for (int la = 0; la < 20; la++) {
        printf("  A lane %2d:", la);
        for (int lb = 0; lb < 64; lb++) for (int l = 0; l < 64; l++) if (meets[la][lb][l]) printf(" (B%d->D%d)", lb, l);
        printf("\n");
    }
    // 2. chain order under the hypothesis block = lane/16, A: i = lane%4, k = (lane/4)%4; B: j = lane%4, k = (lane/4)%4; D: j = lane%4, i = (lane/4)%4
    long total = 0, fwd = 0, rev = 0;
    for (int trial = 0; trial < 500; trial++) {
        for (int i = 0; i < 64; i++) { hA[i] = (double)(float)((rand() / (double)RAND_MAX - 0.5) * 2.0); hB[i] = (double)(float)((rand() / (double)RAND_MAX - 0.5) * 0.1); hC[i] = (rand() / (double)RAND_MAX - 0.5) * 3.0; }
        hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice); hipMemcpy(dC, hC, 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; l++) {
            const int blk = l / 16, j = l % 4, i = (l / 4) % 4;
            double f = hC[l], r = hC[l];
            for (int kk = 0; kk < 4; kk++) f = fma(hA[blk * 16 + kk * 4 + i], hB[blk * 16 + kk * 4 + j], f);
            for (int kk = 3; kk >= 0; kk--) r = fma(hA[blk * 16 + kk * 4 + i], hB[blk * 16 + kk * 4 + j], r);
            total++; fwd += hD[l] == f; rev += hD[l] == r;
        }
    }
    printf("hypothesis (block = lane/16; A[i][k]: lane 16b + 4k + i; B[k][j]: lane 16b + 4k + j; D[i][j]: lane 16b + 4i + j; C like D):\n"
           "  outputs %ld: equal to the sequential k=0..3 fma chain %ld, to the k=3..0 chain %ld\n", total, fwd, rev);
    // 3. rate
    float* dsrc; double* dout;
    hipMalloc(&dsrc, 4096); hipMalloc(&dout, 8 * 1024 * 1024);
    hipMemset(dsrc, 0, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int waves = 1; waves <= 4; waves++) {
        const int blocks = 256, threads = 256 * waves;            // one wave per SIMD per "waves"
        float ms[3];
        for (int variant = 0; variant < 3; variant++) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0, 0);
                if (variant == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(threads), 0, 0, dsrc, dout, iters);
                else if (variant == 1) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(threads), 0, 0, dsrc, dout, iters);
                else hipLaunchKernelGGL(k_rate_valu, dim3(blocks), dim3(threads), 0, 0, dsrc, dout, iters);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[variant], e0, e1);
            }
        }
        // per SIMD: waves x iters x 2 instructions (variants 0, 1) in ms -> ns per instruction pair per wave
        printf("%d wave(s) per SIMD: 2 x mfma_4x4x4 per step %.1f ns/step/wave, with 2 cvt + 2 fadd beside them %.1f, scalar 4 x (cvt + fma) %.1f  (a step = 4 dims of 32 / 32 / 64 rows)\n",
               waves, ms[0] * 1e6 / iters, ms[1] * 1e6 / iters, ms[2] * 1e6 / iters);
    }
    return 0;
}
