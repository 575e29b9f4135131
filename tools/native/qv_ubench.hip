// qv_ubench.hip — on-box ceilings that bench.py reports NEXT TO the kernels' rates (measurement library: quiver_amd/lib/libqvubench.so,
// loaded by bench.py / tests/bench only; nothing in libqv or libqvhost links it and it touches no index).
//   qvu_mfma_f32_rate   the bare v_mfma_f32_32x32x2_f32 issue rate of this chip under load (register-resident operands, four
//                       independent accumulators per wave, 1 or 2 waves per SIMD) with the shader clock it held — what the fp32-MFMA
//                       filter (k_mfma_filter, BASELINE configs[2]) can reach at most with its one wave per SIMD
//   qvu_gather_rate     the bare random-row stream the HNSW traversal is built around: 3 KiB rows of a table far larger than the caches,
//                       32 rows x 128-byte pieces per slab by LDS-DMA, two slab buffers per wave, no arithmetic and no bookkeeping
//                       (tools/ubench/gather_mix.hip's "bare" line) — the ceiling of `gathered GB/s`
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) unsigned char lds_u8;

__global__ void __launch_bounds__(256) k_mfma_rate(const float* A, float* D, int n, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    float a = A[lane], b = A[64 + lane];
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < n; i++) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - t0; clk[1] = wall_clock64() - w0; }
    f16v s = c0 + c1 + c2 + c3;
    float t = 0; for (int i = 0; i < 16; i++) t += s[i];
    D[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

__global__ void __launch_bounds__(64) k_gather_rate(const float* __restrict__ rows, const uint32_t* __restrict__ ids, uint32_t hops, uint32_t rows_per_hop, uint32_t dim, unsigned* out) {
    extern __shared__ __align__(16) unsigned char smem[];
    lds_u8* slabs = (lds_u8*)smem;
    const uint32_t lane = threadIdx.x, drow = lane >> 3, dslot = lane & 7;
    const uint32_t nslab = dim / 32, ng = (rows_per_hop + 7) / 8;
    unsigned acc = 0;
    for (uint32_t h = 0; h < hops; h++) {
        const uint32_t* my = ids + ((size_t)blockIdx.x * hops + h) * 32;
        const float* src[4];
        for (int g = 0; g < 4; g++) { const uint32_t r = g * 8 + drow; src[g] = rows + (size_t)my[r < rows_per_hop ? r : 0] * dim + ((dslot ^ drow ^ (g & 1)) * 4); }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        auto issue = [&](uint32_t sl) {
            for (uint32_t g = 0; g < ng; g++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[g] + (size_t)sl * 32),
                                                 (__attribute__((address_space(3))) void*)(slabs + (sl & 1) * 4096 + g * 1024), 16, 0, 0);
        };
        issue(0);
        for (uint32_t sl = 0; sl < nslab; sl++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (sl + 1 < nslab) issue(sl + 1);
            acc += *(const __attribute__((address_space(3))) unsigned*)(slabs + (sl & 1) * 4096 + lane * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (acc == 0x12345678u) out[0] = 1;
}

extern "C" {

// waves_per_simd 1 or 2; returns 0 and TFLOP/s + the shader clock (GHz) the chip held, or a HIP error code
int qvu_mfma_f32_rate(int waves_per_simd, double* tflops, double* clock_ghz) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    float *A = nullptr, *D = nullptr; unsigned long long* clk = nullptr;
    if (hipMalloc(&A, 4096) != hipSuccess || hipMalloc(&D, (size_t)cus * 2 * 256 * 4) != hipSuccess || hipMalloc(&clk, 16) != hipSuccess) return -2;
    float h[1024]; uint32_t s = 7; for (int i = 0; i < 1024; i++) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) / 16777216.0f - 0.5f; }   // real operand bits: zeros draw less power
    (void)hipMemcpy(A, h, 4096, hipMemcpyHostToDevice);
    const int grid = cus * (waves_per_simd >= 2 ? 2 : 1), n = 40000;
    hipLaunchKernelGGL(k_mfma_rate, dim3(grid), dim3(256), 0, 0, A, D, n, clk);          // the clock settles under load
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_mfma_rate, dim3(grid), dim3(256), 0, 0, A, D, n, clk);
    (void)hipEventRecord(e1, 0);
    const hipError_t e = hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2] = {0, 1}; (void)hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    if (tflops) *tflops = 2.0 * 32 * 32 * 2 * 4.0 * n * (double)grid * 4 / (ms * 1e-3) / 1e12;
    if (clock_ghz) *clock_ghz = (double)c[0] / ((double)c[1] * 10.0);                     // wall_clock64 ticks at 100 MHz
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(A); (void)hipFree(D); (void)hipFree(clk);
    return e == hipSuccess ? 0 : (int)e;
}

// rows: a [n_rows][dim] float32 table ALREADY on the device (the index's row-major copy or any buffer of that size); waves_per_cu 8..16
int qvu_gather_rate(const float* d_rows, uint32_t n_rows, uint32_t dim, int waves_per_cu, double* tbps) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    if (dim % 32 || !d_rows || n_rows < 1024) return -3;
    const uint32_t hops = 96, rph = 31, grid = (uint32_t)cus * (uint32_t)waves_per_cu;
    std::vector<uint32_t> ids((size_t)grid * hops * 32);
    uint64_t s = 88172645463325252ull;
    for (auto& x : ids) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint32_t)(s % n_rows); }
    uint32_t* dids = nullptr; unsigned* out = nullptr;
    if (hipMalloc(&dids, ids.size() * 4) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return -2;
    (void)hipMemcpy(dids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_gather_rate, dim3(grid), dim3(64), 8192, 0, d_rows, dids, hops, rph, dim, out);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_gather_rate, dim3(grid), dim3(64), 8192, 0, d_rows, dids, hops, rph, dim, out);
    (void)hipEventRecord(e1, 0);
    const hipError_t e = hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    if (tbps) *tbps = (double)grid * hops * rph * dim * 4 / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(dids); (void)hipFree(out);
    return e == hipSuccess ? 0 : (int)e;
}

}  // extern "C"
