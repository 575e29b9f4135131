// Probe: does v_mfma_f64_16x16x4_f64 accumulate its 4 k-steps as a sequential chain of IEEE FMAs
// (k = 0,1,2,3, each rounded), i.e. is D bit-identical to fma(a3,b3,fma(a2,b2,fma(a1,b1,fma(a0,b0,c))))?
// make -C tools/ubench && tools/ubench/bin/mfma_f64_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
// A: 16x4 (row i = lane%16, k = lane/16), B: 4x16 (k = lane/16, col j = lane%16), C/D: 16x16, lane holds D[4*(lane/16)+r][lane%16], r=0..3
__global__ void k(const double* A, const double* B, const double* Cin, double* D) {
    const int lane = threadIdx.x;
    double a = A[(lane % 16) * 4 + lane / 16];
    double b = B[(lane / 16) * 16 + lane % 16];
    d4 c;
    for (int r = 0; r < 4; r++) c[r] = Cin[0];   // uniform C: layout-independent
    d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[lane * 4 + r] = d[r];
}
int main() {
    double hA[64], hB[64], hC[256], hD[256];
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
    srand(1);
    long seq_fwd = 0, seq_rev = 0, total = 0, other = 0, exact_single = 0;
    for (int trial = 0; trial < 200; trial++) {
        // float32-valued doubles (exact products), accumulator a full double: the scan's situation
        for (int i = 0; i < 64; i++) { hA[i] = (double)(float)((rand() / (double)RAND_MAX - 0.5) * 2.0); hB[i] = (double)(float)((rand() / (double)RAND_MAX - 0.5) * 0.1); }
        for (int i = 0; i < 256; i++) hC[i] = (rand() / (double)RAND_MAX - 0.5) * 3.0;
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice); hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        static int map_i[256], map_j[256]; 
        for (int o = 0; o < 256; o++) {          // o = lane*4 + r
            int found = 0;
            for (int i = 0; i < 16 && !found; i++) for (int j = 0; j < 16 && !found; j++) {
                double f = hC[0], r = hC[0], ex = hC[0];
                for (int kk = 0; kk < 4; kk++) f = fma(hA[i * 4 + kk], hB[kk * 16 + j], f);
                for (int kk = 3; kk >= 0; kk--) r = fma(hA[i * 4 + kk], hB[kk * 16 + j], r);
                long double e = hC[0]; for (int kk = 0; kk < 4; kk++) e += (long double)hA[i * 4 + kk] * (long double)hB[kk * 16 + j]; ex = (double)e;
                if (fabs(hD[o] - f) < 1e-12) {    // this is output (i,j)
                    found = 1; total++;
                    if (hD[o] == f) seq_fwd++;
                    if (hD[o] == r) seq_rev++;
                    if (hD[o] != f && hD[o] != r) { other++; if (hD[o] == ex) exact_single++; }
                    if (trial == 0) { map_i[o] = i; map_j[o] = j; }
                }
            }
        }
        if (trial == 0) { for (int o = 0; o < 16; o++) printf("lane %d r %d -> D[%d][%d]\n", o / 4, o % 4, map_i[o], map_j[o]); printf("lane 16 r0 -> D[%d][%d], lane 32 r0 -> D[%d][%d]\n", map_i[64], map_j[64], map_i[128], map_j[128]); }
    }
    printf("outputs %ld: equal to sequential k=0..3 chain %ld, equal to k=3..0 chain %ld, neither %ld (of which equal to the singly-rounded exact sum: %ld)\n", total, seq_fwd, seq_rev, other, exact_single);
    return 0;
}
