// Where does the 4.9-5.0 TB/s ceiling of k_bf16rows_filter_q64 come from?  The kernel's memory pattern in isolation: eight (or W)
// free-running waves per workgroup, each streaming its own 128-row groups of a bfloat16 plane (two tiles x STEPS steps x 2 KiB,
// one 16-byte request per lane and 32-row block and step, R steps in flight), with M matrix instructions per step fed by the
// loaded registers (B) and an operand block in LDS (A) — M = 0 .. 8, so the same loop runs as a pure stream, as the filter's
// arithmetic, and in between.  Prints TB/s of the plane per configuration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int W, int R, int M, int NT>
__global__ void __launch_bounds__(W * 64, 1)
k_stream(const u4* __restrict__ plane, uint32_t n_tiles, uint32_t steps, float* out) {
    extern __shared__ __align__(16) unsigned char smem[];
    u4* s_a = reinterpret_cast<u4*>(smem);
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (uint32_t i = threadIdx.x; i < steps * 128; i += W * 64) s_a[i] = u4{i * 2654435761u, i, i ^ 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    const uint32_t half = lane >> 5, l31 = lane & 31;
    const uint32_t n_groups = (n_tiles + 1) / 2;
    const uint32_t gw = blockIdx.x * W + wave, tw = gridDim.x * W;
    u4 rb[R][4];
    const u4* p[4];
    auto bases = [&](uint32_t g_, const u4* (&o)[4]) {
        const uint32_t ta = 2 * g_, tb = 2 * g_ + 1 < n_tiles ? 2 * g_ + 1 : ta;
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = plane + (size_t)(j < 2 ? ta : tb) * steps * 128 + 64 * (j & 1) + 32 * half + l31;
    };
    auto load_step = [&](u4 (&o)[4]) {
#pragma unroll
        for (int j = 0; j < 4; j++) { o[j] = NT ? __builtin_nontemporal_load(p[j]) : *p[j]; p[j] += 128; }
    };
    f16v acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    u4 x = {0, 0, 0, 0};
    bool primed = false;
    for (uint32_t g = gw; g < n_groups; g += tw) {
        const u4* nb[4];
        bases(g + tw < n_groups ? g + tw : g, nb);
        if (!primed) {
            primed = true;
            bases(g, p);
#pragma unroll
            for (int i = 0; i < R; i++) load_step(rb[i]);
        }
        for (uint32_t st = 0; st < steps; st += R) {
#pragma unroll
            for (int k = 0; k < R; k++) {
                if constexpr (M > 0) {
                    const u4 a0 = s_a[(st + k) * 128 + lane], a1 = s_a[(st + k) * 128 + 64 + lane];
                    const bf8 ah0 = __builtin_bit_cast(bf8, a0), ah1 = __builtin_bit_cast(bf8, a1);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const bf8 bh = __builtin_bit_cast(bf8, rb[k][j]);
                        if (2 * j < M) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh, acc[0][j], 0, 0, 0);
                        else x ^= rb[k][j];
                        if (2 * j + 1 < M) acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh, acc[1][j], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) x ^= rb[k][j];
                }
                if (st + k + R == steps) {
#pragma unroll
                    for (int j = 0; j < 4; j++) p[j] = nb[j];
                }
                load_step(rb[k]);
            }
        }
    }
    float s = __builtin_bit_cast(float, x.x ^ x.y ^ x.z ^ x.w);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) s += acc[i][j][0] + acc[i][j][7];
    if (s == 1.2345f) out[0] = s;
}

template <int W, int R, int M, int NT>
void run(const u4* d, uint32_t n_tiles, uint32_t steps, float* out, int wg_per_cu, const char* label) {
    const int grid = 256 * wg_per_cu;
    const size_t lds = (size_t)steps * 128 * 16;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_stream<W, R, M, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_stream<W, R, M, NT>), dim3(grid), dim3(W * 64), lds, 0, d, n_tiles, steps, out);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_stream<W, R, M, NT>), dim3(grid), dim3(W * 64), lds, 0, d, n_tiles, steps, out);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)n_tiles * steps * 2048;
    printf("%-10s W=%2d R=%2d M=%d nt=%d wg/cu=%d: %8.3f ms  %.2f TB/s  (%s)\n", label, W, R, M, NT, wg_per_cu, ms / reps, bytes / (ms / reps * 1e-3) / 1e12,
           hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
    const uint32_t steps = 48;                                    // 768 dimensions
    for (uint32_t rows : {1000000u, 10000000u}) {
        const uint32_t n_tiles = (rows + 63) / 64;
        const size_t bytes = (size_t)n_tiles * steps * 2048;
        u4* d; float* out; hipMalloc(&d, bytes); hipMalloc(&out, 64); hipMemset(d, 0x3c, bytes);
        printf("--- plane of %u rows x 768 bfloat16 = %.2f GB\n", rows, bytes / 1e9);
        run<8, 4, 0, 1>(d, n_tiles, steps, out, 1, "stream");
        run<8, 4, 2, 1>(d, n_tiles, steps, out, 1, "mfma2");
        run<8, 4, 4, 1>(d, n_tiles, steps, out, 1, "mfma4");
        run<8, 4, 8, 1>(d, n_tiles, steps, out, 1, "mfma8");
        run<8, 8, 0, 1>(d, n_tiles, steps, out, 1, "stream");
        run<8, 8, 8, 1>(d, n_tiles, steps, out, 1, "mfma8");
        run<8, 4, 0, 0>(d, n_tiles, steps, out, 1, "stream");
        run<8, 4, 8, 0>(d, n_tiles, steps, out, 1, "mfma8");
        run<4, 8, 0, 1>(d, n_tiles, steps, out, 1, "stream");
        run<4, 8, 8, 1>(d, n_tiles, steps, out, 1, "mfma8");
        run<4, 4, 0, 1>(d, n_tiles, steps, out, 2, "stream");
        run<4, 4, 8, 1>(d, n_tiles, steps, out, 2, "mfma8");
        run<16, 2, 0, 1>(d, n_tiles, steps, out, 1, "stream");
        run<16, 2, 8, 1>(d, n_tiles, steps, out, 1, "mfma8");
        hipFree(d); hipFree(out);
    }
    return 0;
}
