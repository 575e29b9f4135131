// Does WHERE the row table's physical memory comes from matter for the random-row stream?  The bare stream of gather_mix.hip (31 rows per hop,
// 128-byte pieces, 12 waves per CU) over a 3 GB table allocated (a) with hipMalloc, (b) through the virtual-memory API as handles of
// 2 MiB / 64 MiB / 1 GiB mapped into one reserved range.  Run it on a box in the "slow state" (after a few 10 GB processes have come and gone).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((address_space(3))) unsigned char lds_u8;
__global__ void __launch_bounds__(64) k_gather(const float* __restrict__ rows, const uint32_t* __restrict__ ids, uint32_t hops, uint32_t rows_per_hop, uint32_t dim, unsigned* out) {
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
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[g] + (size_t)sl * 32), (__attribute__((address_space(3))) void*)(slabs + (sl & 1) * 4096 + g * 1024), 16, 0, 0);
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
static double run(const float* d, uint32_t n, uint32_t dim) {
    const uint32_t hops = 96, rph = 31, grid = 256 * 12;
    std::vector<uint32_t> ids((size_t)grid * hops * 32);
    uint64_t s = 88172645463325252ull;
    for (auto& x : ids) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint32_t)(s % n); }
    uint32_t* dids; unsigned* out; hipMalloc(&dids, ids.size() * 4); hipMalloc(&out, 64);
    hipMemcpy(dids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_gather, dim3(grid), dim3(64), 8192, 0, d, dids, hops, rph, dim, out);
    hipEventRecord(e0); hipLaunchKernelGGL(k_gather, dim3(grid), dim3(64), 8192, 0, d, dids, hops, rph, dim, out); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(dids); hipFree(out);
    return (double)grid * hops * rph * dim * 4 / (ms * 1e-3) / 1e12;
}
int main() {
    const uint32_t n = 1000000, dim = 768; const size_t bytes = (size_t)n * dim * 4;
    float* d; hipMalloc(&d, bytes); hipMemset(d, 1, bytes);
    printf("hipMalloc                       : %.2f TB/s\n", run(d, n, dim)); hipFree(d);
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gmin = 0, grec = 0; hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum); hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
    printf("allocation granularity: minimum %zu, recommended %zu\n", gmin, grec);
    for (size_t chunk : {(size_t)2 << 20, (size_t)64 << 20, (size_t)1 << 30, (size_t)4 << 30}) {
        const size_t total = (bytes + chunk - 1) / chunk * chunk;
        void* va = nullptr;
        if (hipMemAddressReserve(&va, total, chunk > ((size_t)1 << 30) ? (size_t)1 << 30 : chunk, nullptr, 0) != hipSuccess) { printf("reserve failed for chunk %zu\n", chunk); continue; }
        std::vector<hipMemGenericAllocationHandle_t> hs; bool ok = true;
        for (size_t off = 0; off < total && ok; off += chunk) {
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { ok = false; break; }
            hs.push_back(h);
            if (hipMemMap((char*)va + off, chunk, 0, h, 0) != hipSuccess) ok = false;
        }
        hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        if (ok && hipMemSetAccess(va, total, &acc, 1) != hipSuccess) ok = false;
        if (ok) { hipMemset(va, 1, bytes); printf("virtual-memory API, %4zu MiB handles: %.2f TB/s\n", chunk >> 20, run((const float*)va, n, dim)); }
        else printf("virtual-memory API, %4zu MiB handles: failed (%s)\n", chunk >> 20, hipGetErrorString(hipGetLastError()));
        hipMemUnmap(va, total); for (auto h : hs) hipMemRelease(h); hipMemAddressFree(va, total);
    }
    float* d2; hipMalloc(&d2, bytes); hipMemset(d2, 1, bytes);
    printf("hipMalloc again                 : %.2f TB/s\n", run(d2, n, dim)); hipFree(d2);
    return 0;
}
