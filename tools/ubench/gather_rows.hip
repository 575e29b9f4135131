// Ceiling for the traversal kernel's row stream: random 3 KiB rows of a 3 GB table (1M x 768 float32) fetched by LDS-DMA in the
// kernel's own shape — per wave 21 rows per "hop", 8-row x 128-byte pieces, slabs of 8 chunks, two slab buffers — with no
// arithmetic and no traversal bookkeeping.  Reports gathered TB/s for 4 / 8 / 16 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((address_space(3))) unsigned char lds_u8;
__device__ __forceinline__ void glds16(const float* g, lds_u8* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__global__ void __launch_bounds__(64) k_gather(const float* __restrict__ rows, const uint32_t* __restrict__ ids, uint32_t hops, uint32_t rows_per_hop,
                                             uint32_t dim, unsigned* out) {
    extern __shared__ __align__(16) unsigned char smem[];
    lds_u8* slabs = (lds_u8*)smem;
    const uint32_t lane = threadIdx.x, drow = lane >> 3, dslot = lane & 7;
    const uint32_t dim4 = dim / 4, nslab = dim4 / 8, ng = (rows_per_hop + 7) / 8;
    unsigned acc = 0;
    for (uint32_t h = 0; h < hops; h++) {
        const uint32_t* my = ids + ((size_t)blockIdx.x * hops + h) * 32;
        const float* src[4];
        for (int g = 0; g < 4; g++) { const uint32_t r = g * 8 + drow; src[g] = rows + (size_t)my[r < rows_per_hop ? r : 0] * dim + ((dslot ^ drow ^ (g & 1)) * 4); }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        for (uint32_t g = 0; g < ng; g++) glds16(src[g], slabs + g * 1024);
        for (uint32_t sl = 0; sl < nslab; sl++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (sl + 1 < nslab) for (uint32_t g = 0; g < ng; g++) glds16(src[g] + (size_t)(sl + 1) * 32, slabs + ((sl + 1) & 1) * 4096 + g * 1024);
            acc += *(const __attribute__((address_space(3))) unsigned*)(slabs + (sl & 1) * 4096 + lane * 16);   // touch the slab
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (acc == 0x12345678u) out[0] = 1;
}
int main() {
    const uint32_t n = 1000000, dim = 768, rows_per_hop = 21, hops = 64;
    float* d; hipMalloc(&d, (size_t)n * dim * 4); hipMemset(d, 0, (size_t)n * dim * 4);
    unsigned* out; hipMalloc(&out, 64);
    for (int wpc : {4, 8, 16}) {
        const uint32_t grid = 256 * wpc;
        std::vector<uint32_t> ids((size_t)grid * hops * 32);
        uint64_t s = 88172645463325252ull;
        for (auto& x : ids) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint32_t)(s % n); }
        uint32_t* dids; hipMalloc(&dids, ids.size() * 4); hipMemcpy(dids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_gather, dim3(grid), dim3(64), 8192, 0, d, dids, hops, rows_per_hop, dim, out);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_gather, dim3(grid), dim3(64), 8192, 0, d, dids, hops, rows_per_hop, dim, out);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)grid * hops * rows_per_hop * dim * 4;
        printf("%2d waves/CU: %.3f ms, %.2f TB/s gathered (%.2f G rows/s)\n", wpc, ms, bytes / (ms * 1e-3) / 1e12, bytes / (dim * 4) / (ms * 1e-3) / 1e9);
        hipFree(dids);
    }
    return 0;
}
