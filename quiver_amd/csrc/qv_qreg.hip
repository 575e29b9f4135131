// qv_qreg.hip — the one-term bfloat16 filter with the QUERY operands resident in registers (round 4)
//
// What bounded the eight-wave filter kernels (k_bf16x1_filter_w8x2, k_bf16rows_filter): every wave re-reads its
// query operands from L2 for each row group (3.2 GB of the 4.7 GB that cross L2 -> L1 per launch at 256 x 1M x 768), every wave reads
// every B operand of the group from LDS, and one barrier per 16-dimension step keeps the waves in phase, so a step's parts (requests,
// LDS, matrix instructions) run one after the other.
//
// Here a workgroup is four waves, one per SIMD, 512 registers each.  Wave w keeps the one-term operands of queries 64w .. 64w+63 for
// ALL dimensions: 2 x STEPS 16-byte A operands per lane (384 registers at 768 dimensions; the last LSTEPS steps' operands live in LDS
// instead, which is the valve that keeps the kernel under 512 registers).  Rows arrive by LDS-DMA (global_load_lds_dwordx4: no
// register in between) into a ring of 16-KiB stages — a stage is 64 dimensions of one 64-row tile (float32 rows: 16 chunks of the
// tile layout, contiguous) or 128 dimensions of it (bfloat16 plane, contiguous as well) — several stages ahead of their use, ONE
// barrier per stage (16 or 32 matrix instructions per wave) instead of one per step.  Each wave reads the stage's rows from LDS
// (float32: two 16-byte reads + 4 v_cvt_pk_bf16_f32 per B operand; bfloat16: the read IS the operand) and multiplies them with its
// own 64 queries: 64 queries x 64 rows of accumulators per wave.  No query traffic in the loop at all.
//
// Per tile and CU: 192 matrix instructions per SIMD (6144 cycles), HBM 192 KiB (float32 rows, ~16 500 cycles at 7 TB/s) or 96 KiB.
// The DMA is issued from inline assembly: the compiler would otherwise order EVERY LDS read behind the youngest LDS-DMA (vmcnt(0):
// it cannot tell the stages apart), which would serialize the ring.  The waits are explicit: s_waitcnt vmcnt(N) for the wave's own
// requests of the stage (requests complete in order), then the workgroup barrier for the other waves'.
//
// What it gains and what bounds it (profiles/r04_qreg.md): 570 -> 517 us on float32 rows, 433 -> 388 us on the copy.
// Every schedule of this work lands within 2 % of the others and two waves per SIMD change nothing: the chip runs at 1.7-1.8 GHz under
// the kernel (1.47 under the bare chain of its matrix instructions, 255 us) — it is at its power limit, and what was saved is energy:
// the query operands' L2 -> L1 traffic and half the LDS reads.
#include "qv_filter.h"

namespace qv {

typedef __attribute__((address_space(3))) unsigned char qlds_u8;

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// one 1-KiB request of a stage: lane l moves bytes [16 l, 16 l + 16); src (wave-uniform) and the LDS byte address both take `off`.
// (The same request through a buffer descriptor — base in the descriptor, the piece's offset in a scalar register — measures the same:
// 518 / 393 us against 517 / 388; what a request costs the issuing wave, ~55 ns, is not its address arithmetic.)
template <int OFF>
__device__ __forceinline__ void dma_piece(const void* src, uint32_t lane16, uint32_t lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:%3 nt"
                 :: "v"(lane16), "s"(src), "s"(lds_byte_addr), "n"(OFF) : "memory");   // (m0 is written; the compiler keeps nothing in it across statements in this kernel: no movrel, GWS or LDS-DMA builtins)
}

// 256 bytes, one dword per lane (byte offset voff from the wave-uniform src), to lds_byte_addr + 4 lane: the row constants of a tile
__device__ __forceinline__ void dma_words(const void* src, uint32_t voff, uint32_t lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dword %0, %1"
                 :: "v"(voff), "s"(src), "s"(lds_byte_addr) : "memory");
}

// The matrix instruction from inline assembly: the A operand straight out of the accumulation registers (AG) — the compiler's own
// allocation keeps what does not fit the 256 vector registers there too, but copies it out with four v_accvgpr_read before every
// use — or out of a vector register; ZERO: C = 0 (a tile's first step).  volatile: the statements stay in the order written, which
// is the schedule.  Hazards are the writer's here: NOP = the B operand was written by the vector ALU just before (two wait
// states); the accumulators are read by vector instructions only after the tile's last step (s_nop there).
template <bool AG, bool ZERO, bool NOP>
__device__ __forceinline__ void mfma_bf16(f16v& acc, const u32x4& a, const u32x4& b) {
    if constexpr (ZERO) {
        if constexpr (AG) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(a), "v"(b));
        else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
    } else if constexpr (NOP) {
        if constexpr (AG) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
        else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    } else {
        if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    }
}

constexpr int kQregStageBytes = 16384;
#ifndef QV_QREG_DBG
#define QV_QREG_DBG 0   // measurement builds: 1 no epilogue, 2 no row requests in the loop, 4 no LDS reads of rows in the loop, 8 no waits / barriers
#endif
#ifndef QV_QREG_DMA_GAP
#define QV_QREG_DMA_GAP 2   // the gap of a step (after its 2nd or 4th matrix instruction) that takes the step's row request
#endif
#ifndef QV_QREG_NST
#define QV_QREG_NST 6
#endif
#ifndef QV_QREG_LSTEPS
#define QV_QREG_LSTEPS 4
#endif
#ifndef QV_QREG_ASTEPS
#define QV_QREG_ASTEPS 32
#endif

// STEPS: 16-dimension steps (dim = 16 STEPS exactly).  The A operands of steps [0, ASTEPS) live in accumulation registers, of the last
// LSTEPS steps in LDS, of the steps between in vector registers.  NST ring stages.  BF: rows from the index's bfloat16 plane (8 steps
// per stage), else float32 tiles (4 steps per stage).
template <int METRIC, int STEPS, int ASTEPS, int LSTEPS, int NST, bool BF>
__global__ void __launch_bounds__(256, 1)
k_qreg_filter(IndexView v, const uint4* __restrict__ Qbf, const float* __restrict__ cq, const float* __restrict__ mq, uint32_t nq_pad,
              uint32_t* __restrict__ cand_rows, float* __restrict__ cand_score, uint32_t* __restrict__ cand_cnt) {
    constexpr int SPS = BF ? 8 : 4;                                  // steps per stage
    constexpr int NSTG = STEPS / SPS;                                // stages per tile
    constexpr int VSTEPS = STEPS - ASTEPS - LSTEPS;
    constexpr int RSTEPS = STEPS - LSTEPS;
    static_assert(STEPS % SPS == 0 && NSTG >= 1 && NST >= 4 && LSTEPS >= 0 && VSTEPS >= 0 && ASTEPS >= 0 && ASTEPS <= 32, "shape");
    constexpr bool kStaticSlots = NSTG % NST == 0;
    __shared__ __align__(16) float s_c[256], s_m[512];
    __shared__ __align__(1024) unsigned char s_ring[NST][kQregStageBytes];
    __shared__ __align__(16) u32x4 s_a[4][LSTEPS > 0 ? LSTEPS : 1][2][64];
    // The epilogue's row constants travel with the rows: a vector load for them would sit behind every row request in the wave's
    // in-order queue, and the epilogue would wait for the whole ring to land (1.3 us per tile).  Per tile 1 KiB: norms (64 doubles),
    // residuals (64 floats), the alive word — one 256-byte request per wave, issued with the tile's first stage.
    constexpr int kRcTiles = 8;
    __shared__ __align__(16) uint32_t s_rc[kRcTiles][256];
    QV_CAND_QUEUE(cqu, 4, 128);                                      // 6 KiB
    QV_EPI_DUMP(du, 4, 16);                                          // 5 KiB
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t wgs_per_group = nq_pad >> 8;
    const uint32_t stride = gridDim.x / wgs_per_group;
    // which query block and which tiles: the workgroups that walk the SAME tiles share an XCD's L2 (filter_block_role, qv_filter.h)
    uint32_t qb256, first;
    filter_block_role(wgs_per_group, qb256, first);
    {
        const uint32_t q = 256 * qb256 + threadIdx.x;
        const float c = cq[q], m = mq[q];
        s_c[threadIdx.x] = METRIC == QV_COSINE ? c - m : c;
        s_m[threadIdx.x] = m;
        s_m[256 + threadIdx.x] = mq[nq_pad + q];
    }
    // the wave's query operands: hi plane of 32-query blocks 8 qb256 + 2 wave + i
    u32x4 Aa[ASTEPS > 0 ? ASTEPS : 1][2], Av[VSTEPS > 0 ? VSTEPS : 1][2];
    {
        const u32x4* a0 = reinterpret_cast<const u32x4*>(Qbf) + ((size_t)(8 * qb256 + 2 * wave) * STEPS) * 2 * 64 + lane;
        const u32x4* a1 = a0 + (size_t)STEPS * 2 * 64;
#pragma unroll
        for (int s = 0; s < ASTEPS; s++) { Aa[s][0] = a0[(size_t)s * 128]; Aa[s][1] = a1[(size_t)s * 128]; }
#pragma unroll
        for (int s = 0; s < VSTEPS; s++) { Av[s][0] = a0[(size_t)(ASTEPS + s) * 128]; Av[s][1] = a1[(size_t)(ASTEPS + s) * 128]; }
#pragma unroll
        for (int s = 0; s < LSTEPS; s++) { s_a[wave][s][0][lane] = a0[(size_t)(RSTEPS + s) * 128]; s_a[wave][s][1][lane] = a1[(size_t)(RSTEPS + s) * 128]; }
    }
    // Every operand is "used" here, so that the compiler waits for these loads HERE: it cannot see the row requests below (inline
    // assembly), and its own wait for an operand's load at the operand's first use — inside the tile loop, executed for every
    // tile — would be s_waitcnt vmcnt(0): the whole ring drained once per tile.
#pragma unroll
    for (int s = 0; s < ASTEPS; s++) { asm volatile("" :: "a"(Aa[s][0]), "a"(Aa[s][1])); }
#pragma unroll
    for (int s = 0; s < VSTEPS; s++) { asm volatile("" :: "v"(Av[s][0]), "v"(Av[s][1])); }
    __syncthreads();
    if (stride == 0) return;
    if (first >= v.n_tiles) return;
    const uint32_t n_mine = (v.n_tiles - first + stride - 1) / stride;
    const uint32_t half = lane >> 5, l31 = lane & 31;
    const EpiConsts ec = epi_consts<METRIC>(s_c, s_m, 64 * wave, half);
    const uint32_t ring_addr = (uint32_t)(size_t)(qlds_u8*)&s_ring[0][0];
    const uint32_t lane16 = lane * 16;
    const size_t tile_bytes = (size_t)STEPS * (BF ? 2048 : 4096);
    const unsigned char* rows_base = BF ? reinterpret_cast<const unsigned char*>(v.bf16) : reinterpret_cast<const unsigned char*>(v.tiles);
    // producer cursor: the stage being requested (tile place p_i in this workgroup's sequence, stage p_s of it, ring slot p_slot);
    // a stage is four requests per wave, handed out one at a time (request_piece<0..3>) so that they can sit in different gaps
    uint32_t p_i = 0, p_s = 0, p_slot = 0;
    const unsigned char* p_src = rows_base + (size_t)first * tile_bytes + 4096 * wave;
    uint32_t p_dst = ring_addr + 4096 * wave;
    const uint32_t rc_addr = (uint32_t)(size_t)(qlds_u8*)&s_rc[0][0];
    const unsigned char* rc_base = wave < 2 ? reinterpret_cast<const unsigned char*>(v.rnorm) + 256 * wave
                                 : (wave == 2 ? reinterpret_cast<const unsigned char*>(v.rres) : reinterpret_cast<const unsigned char*>(v.alive));
    const uint32_t rc_mul = wave < 2 ? 512u : (wave == 2 ? 256u : 8u);      // bytes per tile in this wave's array
    const uint32_t rc_voff = wave == 3 ? (lane & 1u) * 4u : lane * 4u;      // (the alive word: two dwords, every other lane pair re-reads them)
    auto request_consts = [&]() {                                    // with the first stage of the producer's tile
        const uint32_t pi = p_i < n_mine ? p_i : n_mine - 1;
        dma_words(rc_base + (size_t)(first + pi * stride) * rc_mul, rc_voff, rc_addr + (p_i & (kRcTiles - 1)) * 1024 + 256 * wave);
    };
    auto request_advance = [&]() {
        p_s++; if (p_s == (uint32_t)NSTG) { p_s = 0; p_i++; }
        p_slot++; if (p_slot == (uint32_t)NST) p_slot = 0;
        const uint32_t pi = p_i < n_mine ? p_i : n_mine - 1;        // past the end: the last tile again (lands in a slot nobody reads)
        p_src = rows_base + (size_t)(first + pi * stride) * tile_bytes + (size_t)p_s * kQregStageBytes + 4096 * wave;
        p_dst = ring_addr + p_slot * kQregStageBytes + 4096 * wave;
    };
#pragma unroll
    for (int i = 0; i < NST - 1; i++) {
        if (i % NSTG == 0) request_consts();
        dma_piece<0>(p_src, lane16, p_dst); dma_piece<1024>(p_src, lane16, p_dst); dma_piece<2048>(p_src, lane16, p_dst); dma_piece<3072>(p_src, lane16, p_dst);
        request_advance();
    }
    // this lane's place in a stage: float32 — chunk 4 s + 2 half (+1) of row 32 j + l31: byte (4 s + 2 half) 1024 + (32 j + l31) 16;
    // bfloat16 — (step s, block j) is one KiB, lane l its 16 bytes
    const uint32_t lane_off = BF ? lane16 : half * 2048 + l31 * 16;
    struct Raw { f4 x[BF ? 1 : 2]; };                                // what one B operand reads from LDS
    auto fetch = [&](const unsigned char* sp, int s_, int j, Raw& r) {
        if constexpr (BF) r.x[0] = *reinterpret_cast<const f4*>(sp + s_ * 2048 + j * 1024);
        else { r.x[0] = *reinterpret_cast<const f4*>(sp + s_ * 4096 + j * 512); r.x[1] = *reinterpret_cast<const f4*>(sp + s_ * 4096 + j * 512 + 1024); }
    };
    auto pack = [&](const Raw& r) {
        u32x4 b;
        if constexpr (BF) b = __builtin_bit_cast(u32x4, r.x[0]);
        else { b.x = pack_bf16(r.x[0].x, r.x[0].y); b.y = pack_bf16(r.x[0].z, r.x[0].w); b.z = pack_bf16(r.x[1].x, r.x[1].y); b.w = pack_bf16(r.x[1].z, r.x[1].w); }
        return b;
    };
    // A stage is certified one stage ahead of its use (the barrier at the top of stage n says stage n + 1 has landed), so that the
    // last step of a stage can already read the first step of the next: NST - 2 stages stay in flight.
    wait_vmcnt<4 * (NST - 2)>();
    __syncthreads();                                                 // the first stage is there
    uint32_t c_slot = 0;
    for (uint32_t ti = 0; ti < n_mine; ti++) {
        const uint32_t t = first + ti * stride;
        f16v acc[2][2];
        double rnd[2]; float rho[2]; uint64_t alv[1];
        Raw raw0, raw1;                                              // float32 rows: what the next step's two operands read
        u32x4 b0, b1;                                                // this step's B operands (row blocks 0 and 1 of the tile)
        u32x4 bn0, bn1;                                              // bfloat16 rows: the next step's
        u32x4 al0, al1;                                              // a step's A operands out of LDS (the last LSTEPS steps)
#pragma unroll
        for (int st = 0; st < NSTG; st++) {
            if constexpr (!(QV_QREG_DBG & 8)) {
            wait_vmcnt<4 * (NST - 3)>();                             // this wave's four requests of the NEXT stage have landed
            __syncthreads();                                         // ... and everybody's; the slot consumed before this one is free
            }
            const uint32_t slot = kStaticSlots ? (uint32_t)(st % NST) : c_slot;
            const uint32_t slot1 = kStaticSlots ? (uint32_t)((st + 1) % NST) : (c_slot + 1 == (uint32_t)NST ? 0u : c_slot + 1);
            const unsigned char* sp = &s_ring[0][0] + slot * kQregStageBytes + lane_off;
            const unsigned char* sp1 = &s_ring[0][0] + slot1 * kQregStageBytes + lane_off;
            if (st == 0) {                                           // (once per tile: the epilogue carries no operands across)
                fetch(sp, 0, 0, raw0); fetch(sp, 0, 1, raw1);
                b0 = pack(raw0);
                if constexpr (BF) b1 = pack(raw1);
                if constexpr (LSTEPS == STEPS) { al0 = s_a[wave][0][0][lane]; al1 = s_a[wave][0][1][lane]; }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < SPS; s++) {
                const int gs = st * SPS + s;
                const bool has_next = !(st == NSTG - 1 && s == SPS - 1) && !(QV_QREG_DBG & 4);
                const unsigned char* np = s + 1 < SPS ? sp : sp1;
                const int ns = s + 1 < SPS ? s + 1 : 0;
                const bool ag = gs < ASTEPS, lds_a = gs >= RSTEPS, lds_an = gs + 1 >= RSTEPS && gs + 1 < STEPS;
                const u32x4& a0 = ag ? Aa[ag ? gs : 0][0] : (lds_a ? al0 : Av[!ag && !lds_a ? gs - ASTEPS : 0][0]);
                const u32x4& a1 = ag ? Aa[ag ? gs : 0][1] : (lds_a ? al1 : Av[!ag && !lds_a ? gs - ASTEPS : 0][1]);
#define QV_MF(ACC, AOP, BOP, NOPP) { if (gs == 0) { if (ag) mfma_bf16<true, true, false>(ACC, AOP, BOP); else mfma_bf16<false, true, false>(ACC, AOP, BOP); } \
                                     else if (ag) mfma_bf16<true, false, NOPP>(ACC, AOP, BOP); else mfma_bf16<false, false, NOPP>(ACC, AOP, BOP); }
                if constexpr (!BF) {
                    // (the next step's first reads BEFORE this matrix instruction — two instructions between the conversion that wrote b0
                    // and its reader instead of the s_nop 1 — measured the same: 520 against 515 us)
                    QV_MF(acc[0][0], a0, b0, true)
                    if (has_next) fetch(np, ns, 0, raw0);            // gap 1: the next step's first operand is requested, this step's second converted
                    b1 = pack(raw1);
                    __builtin_amdgcn_sched_barrier(0);
                    QV_MF(acc[1][0], a1, b0, false)
#define QV_DMA_F32 { if (QV_QREG_DBG & 2) {} \
                    else if (s == 0) { if ((st + NST - 1) % NSTG == 0) request_consts(); dma_piece<0>(p_src, lane16, p_dst); } \
                    else if (s == 1) dma_piece<1024>(p_src, lane16, p_dst); \
                    else if (s == 2) dma_piece<2048>(p_src, lane16, p_dst); \
                    else { dma_piece<3072>(p_src, lane16, p_dst); request_advance(); } }
                    if (QV_QREG_DMA_GAP == 2) QV_DMA_F32                 // one row request per step
                    __builtin_amdgcn_sched_barrier(0);
                    QV_MF(acc[0][1], a0, b1, false)
                    if (has_next) fetch(np, ns, 1, raw1);            // gap 3
                    if (lds_an) al0 = s_a[wave][gs + 1 - RSTEPS >= 0 ? gs + 1 - RSTEPS : 0][0][lane];
                    __builtin_amdgcn_sched_barrier(0);
                    QV_MF(acc[1][1], a1, b1, false)
                    if (lds_an) al1 = s_a[wave][gs + 1 - RSTEPS >= 0 ? gs + 1 - RSTEPS : 0][1][lane];
                    if (has_next) b0 = pack(raw0);                   // gap 4
                    if (QV_QREG_DMA_GAP == 4) QV_DMA_F32
#undef QV_DMA_F32
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    QV_MF(acc[0][0], a0, b0, false)
                    if (has_next) fetch(np, ns, 0, raw0);
                    __builtin_amdgcn_sched_barrier(0);
                    QV_MF(acc[1][0], a1, b0, false)
                    if (has_next) fetch(np, ns, 1, raw1);
#define QV_DMA_BF { if (QV_QREG_DBG & 2) {} \
                    else if (s == 0) { if ((st + NST - 1) % NSTG == 0) request_consts(); dma_piece<0>(p_src, lane16, p_dst); } \
                    else if (s == 2) dma_piece<1024>(p_src, lane16, p_dst); \
                    else if (s == 4) dma_piece<2048>(p_src, lane16, p_dst); \
                    else if (s == 6) { dma_piece<3072>(p_src, lane16, p_dst); request_advance(); } }
                    if (QV_QREG_DMA_GAP == 2) QV_DMA_BF
                    __builtin_amdgcn_sched_barrier(0);
                    QV_MF(acc[0][1], a0, b1, false)
                    if (lds_an) al0 = s_a[wave][gs + 1 - RSTEPS >= 0 ? gs + 1 - RSTEPS : 0][0][lane];
                    __builtin_amdgcn_sched_barrier(0);
                    QV_MF(acc[1][1], a1, b1, false)
                    if (lds_an) al1 = s_a[wave][gs + 1 - RSTEPS >= 0 ? gs + 1 - RSTEPS : 0][1][lane];
                    if (QV_QREG_DMA_GAP == 4) QV_DMA_BF
#undef QV_DMA_BF
                    if (has_next) { bn0 = pack(raw0); bn1 = pack(raw1); b0 = bn0; b1 = bn1; }
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef QV_MF
            }
            if (!kStaticSlots) { c_slot++; if (c_slot == (uint32_t)NST) c_slot = 0; }
        }
        {
            const uint32_t* rc = s_rc[ti & (kRcTiles - 1)];
#pragma unroll
            for (int j = 0; j < 2; j++) { rnd[j] = *reinterpret_cast<const double*>(rc + 2 * (32 * j + l31)); rho[j] = __uint_as_float(rc[128 + 32 * j + l31]); }
            alv[0] = *reinterpret_cast<const uint64_t*>(rc + 192);
        }
        // the last matrix instructions' results, before vector instructions read them (18 wait states; the statement names the
        // accumulators so that nothing that reads them is scheduled ahead of it)
        asm volatile("s_nop 15\n\ts_nop 1" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
#if QV_QREG_DBG & 1                                                      // (measurement build: no epilogue)
        { float sd = 0.f;
#pragma unroll
          for (int e = 0; e < 16; e++) sd += acc[0][0][e] + acc[0][1][e] + acc[1][0][e] + acc[1][1][e];
          if (sd == 1.2345678f) cand_cnt[0] = 1;
          if (sd == sd || sd != sd) continue; }
#endif
        filter_epilogue<METRIC>(v, acc, t, t, s_c, s_m, 64 * wave, half, l31, 256 * qb256 + 64 * wave, filter_tiny_norm(v.dim), ec, rnd, rho, alv, cqu, cqu_n, cqu_out, du);
    }
    wait_vmcnt<0>();                                                 // the requests past the end, before the workgroup's LDS is released
    cand_flush(cqu, cqu_n, cqu_out);
}

// (Measured and not kept: the same kernel with TWO waves per SIMD on the bfloat16 copy — eight waves of 32 queries, 252-256 registers,
// so that one wave's row requests and epilogue run under the other's matrix instructions: 403 us against 393.  Together with the
// schedules above that all land within 2 % of each other this says the kernel is not short of issue slots: the chip is at its power
// limit — 1.7-1.8 GHz under this kernel, 1.47 under the bare matrix chain — and what costs time is what costs energy.)

// Which shapes run here: whole workgroups of 256 queries; dimensions 384, 512 and 768 (16 STEPS exactly: 64 queries' operands are
// STEPS x 8 registers per lane, and the row stages divide the K loop); everything else stays on the eight-wave kernels.  QV_QREG=2
// (read once) turns it off for measurements.
static int qreg_steps(const IndexView& v) { return (v.dim & 15u) == 0 && v.dim4 * 4 == v.dim ? (int)(v.dim / 16) : 0; }
bool qreg_filter_applies(const IndexView& v, uint32_t nq_pad, bool bfrows) {
    static const int env = dev_env_int("QV_QREG", 1);
    if (env != 1 || nq_pad < 256 || (nq_pad & 255u)) return false;
    const int st = qreg_steps(v);
    (void)bfrows;
    return st == 24 || st == 32 || st == 48;                       // (256 dimensions: measured behind the eight-wave kernel, 0.728 against 0.716 ms per 3 GB batch)
}

hipError_t launch_qreg_filter(const IndexView& v, const uint4* Qbf, const float* cq, const float* mq, uint32_t nq_pad, uint32_t* cand, float* cscore,
                              uint32_t* cnt, bool bfrows, int cus, hipStream_t s) {
    const uint32_t wgs = nq_pad >> 8;
    uint32_t grid = (uint32_t)cus / wgs * wgs;
    if (!grid) grid = wgs;
    const int st = qreg_steps(v);
    // <STEPS, in accumulation registers, in LDS>: 768 dimensions keep 32 steps' operands in the 256 accumulation registers, 12 in vector
    // registers and 4 in LDS; up to 512 dimensions everything fits the accumulation registers
#define QV_QR1(MMM, SS, AA, LL) { if (bfrows) hipLaunchKernelGGL((k_qreg_filter<MMM, SS, AA, LL, QV_QREG_NST, true>), dim3(grid), dim3(256), 0, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt); \
                                  else hipLaunchKernelGGL((k_qreg_filter<MMM, SS, AA, LL, QV_QREG_NST, false>), dim3(grid), dim3(256), 0, s, v, Qbf, cq, mq, nq_pad, cand, cscore, cnt); }
#define QV_QR(MMM) { if (st == 48) QV_QR1(MMM, 48, QV_QREG_ASTEPS, QV_QREG_LSTEPS) else if (st == 32) QV_QR1(MMM, 32, 32, 0) else if (st == 24) QV_QR1(MMM, 24, 24, 0) \
                     else return hipErrorInvalidValue; }
    if (v.metric == QV_COSINE) QV_QR(QV_COSINE) else if (v.metric == QV_DOT) QV_QR(QV_DOT) else QV_QR(QV_L2)
#undef QV_QR
#undef QV_QR1
    return hipGetLastError();
}

}  // namespace qv
