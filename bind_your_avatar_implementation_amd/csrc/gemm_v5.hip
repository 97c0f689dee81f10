// Persistent 128x256x64 bf16 GEMM: the tile for row counts that leave the 256x256 kernel's grid half empty.
//
// A rank of the 8-way sequence-parallel step has 2222 token rows: 9 row tiles of 256, so its 3072-wide Linears are 108
// tiles on 256 CUs (the K-split of gemm_v4.hip pays for K = 12288 only) and the 9216-wide q|k|v projection is 1.27 rounds.
// Halving the tile's ROWS doubles the tile count at the same W traffic per tile (216 and 648 tiles) -- the wave keeps its
// 128-column span, so the 16-byte epilogues of gemm_wide_epilogue.h (bias / gate / residual, and the q/k-norm + RoPE one of the
// q|k|v projection) are the ones of gemm_v4.hip with four row blocks per wave instead of eight.
//
// Same operands, same fused epilogues, same summation order (K-tiles in order, two 32-wide k-steps each, first MFMA with
// C = 0) and therefore the same bits as every other tiled kernel here.  What differs from gemm_v4.hip:
//
//  * 4 waves x (64 rows x 128 columns): 128 accumulator registers (AGPRs), 4 + 8 fragments per k-step.
//  * A K-tile is 64 MFMAs -- half the time for an LDS-DMA piece to land.  The ring therefore has THREE 48-KiB stages
//    (144 of the CU's 160 KiB) and the K-tile t + 3 is requested while K-tile t computes: two K-tile periods to land.
//  * ONE barrier per K-tile: behind it every wave holds all fragments of K-tile t (so its stage may be refilled) and K-tile
//    t + 1 has landed for everybody (so its k-step-0 fragments may be read).
//  * The ring runs ACROSS output tiles: the last three K-tiles of a tile request K-tiles 0, 1, 2 of the workgroup's next
//    tile, and the last K-tile reads the next tile's first fragments -- no prologue at all after the first tile.
//
// tools/gen_gemm_v5_schedule.py holds the placement table and rewrites the GENERATED block.  Compiled WITHOUT
// -amdgpu-mfma-vgpr-form (accumulators in AGPRs), like gemm_v4.hip; the remarks there about inline-asm MFMAs apply.
#include "gemm_wide_epilogue.h"
#include "options.h"

// timing-only ablation builds (tools/gemm_p128_ablate.py; never the shipped library): 1 no epilogue, 2 no LDS-DMA inside the
// K-loop, 4 no barriers inside the K-loop, 8 no fragment reads inside the K-loop, 16 every K-tile re-reads K-tile 0 (every request an L2 hit),
// 32 / 64 no A / no W pieces inside the K-loop (the vmcnt waits then wait for less: timing only)
#ifndef BYA_GEMM5_ABLATE
#define BYA_GEMM5_ABLATE 0
#endif

namespace {

struct Tile128 { int z, m0, n0; bool valid; };

template <bool QKN>
__global__ __launch_bounds__(256, 1) void gemm128p_kernel(GemmArgs p, int tiles_m, int tiles_n, int batch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 128, BN = 256, TILE_A = BM * BK * 2, STAGE = (BM + BN) * BK * 2;
    static_assert(TILE_A == 16384 && STAGE == 49152, "three stages fill 144 KiB");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.K / BK;                                   // >= 4 (launcher)
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;

    // ---- this workgroup's output tiles: XCD x owns a contiguous range of the group-M tile order (gemm_v4.hip)
    const int per_z = tiles_m * tiles_n, total = per_z * batch;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int cq = total >> 3, cr = total & 7;
    const int base = (xcd < cr) ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
    const int end = base + cq + (xcd < cr ? 1 : 0);
    auto coord = [&](int seq) {
        Tile128 c;
        const int id = base + slot + seq * slots;
        c.valid = id < end;
        const int idz = c.valid ? id : base;
        c.z = idz / per_z;
        const int idt = idz - c.z * per_z;
        const int GM = p.gm;
        const int per_group = GM * tiles_n;
        const int group = idt / per_group, first_m = group * GM;
        const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
        const int in_g = idt - group * per_group;
        c.m0 = (first_m + in_g % gsz) * BM;
        c.n0 = (in_g / gsz) * BN;
        return c;
    };
    int seq = 0;
    Tile128 cur = coord(seq);
    if (!cur.valid) return;

    // fragment read addresses (XOR swizzle on (row >> 1) & 7; row blocks are 16 rows = 2048 bytes apart).  r?0: k-step 0 of
    // K-tile t + 1, r?1: k-step 1 of K-tile t -- they sit one ring stage apart and advance together.
    const int a_row = wm * 64 + fr, w_row = wn * 128 + fr;
    const int a_sw = (a_row >> 1) & 7, w_sw = (w_row >> 1) & 7;
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
    uint32_t rA0 = lds0 + a_row * 128 + ((fq ^ a_sw) << 4), rA1 = lds0 + a_row * 128 + (((4 + fq) ^ a_sw) << 4);
    uint32_t rW0 = lds0 + TILE_A + w_row * 128 + ((fq ^ w_sw) << 4);
    uint32_t rW1 = lds0 + TILE_A + w_row * 128 + (((4 + fq) ^ w_sw) << 4);
    // this wave's first A / W piece in the stage of K-tile t (the stage that K-tile t + 3 goes to)
    uint32_t fillA = __builtin_amdgcn_readfirstlane(lds0 + wave * 32 * 128);
    uint32_t fillW = __builtin_amdgcn_readfirstlane(lds0 + TILE_A + wave * 64 * 128);
    int st = 0;                                                // ring stage of K-tile t

    // staging: wave w moves tile rows [32 w, 32 w + 32) of A (4 one-KiB pieces of 8 rows) and LDS slot rows [64 w, 64 w + 64)
    // of W (8 pieces); W slot row s = 128 h + 16 i + r holds tile column 128 h + ((r & 3) * 4 + (r >> 2)) * 8 + i.  The lane
    // loads the source chunk that belongs at its linear LDS position (source-side XOR swizzle).  Offsets are relative to the
    // tile origin, which lives in the buffer descriptor: rows past M / N arrive as zeros, an invalid tile's descriptor is empty.
    uint32_t voA[4], voW[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rl = wave * 32 + q * 8 + (lane >> 3);
        voA[q] = (uint32_t)rl * (uint32_t)(p.lda * 2) + ((lane & 7) ^ ((rl >> 1) & 7)) * 16;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int rl = wave * 64 + q * 8 + (lane >> 3);
        const int r = rl & 15, i = (rl >> 4) & 7;
        const int wcol = (rl & 128) + (((r & 3) << 2) | (r >> 2)) * 8 + i;
        voW[q] = (uint32_t)wcol * (uint32_t)(p.ldw * 2) + ((lane & 7) ^ ((rl >> 1) & 7)) * 16;
    }
    auto a_rsrc = [&](const Tile128& c) {
        const long long left = ((long long)(p.M - 1 - c.m0) * p.lda + p.K) * 2;
        return raw_rsrc(p.A + (long long)c.z * p.a_bs + (long long)c.m0 * p.lda, c.valid && left > 0 ? (uint32_t)left : 0u);
    };
    auto w_rsrc = [&](const Tile128& c) {
        const long long left = ((long long)(p.N - 1 - c.n0) * p.ldw + p.K) * 2;
        return raw_rsrc(p.W + (long long)c.n0 * p.ldw, c.valid && left > 0 ? (uint32_t)left : 0u);
    };
    i32x4 rsA = a_rsrc(cur), rsW = w_rsrc(cur);

#define DMA_A(Q, BASE, RS, SOFF) dma_piece<(Q) * 1024>(BASE, voA[Q], RS, SOFF)
#define DMA_W(Q, BASE, RS, SOFF) dma_piece<(Q) * 1024>(BASE, voW[Q], RS, SOFF)
#define ALL4(M, ...) M(0, __VA_ARGS__); M(1, __VA_ARGS__); M(2, __VA_ARGS__); M(3, __VA_ARGS__)
#define ALL8(M, ...) ALL4(M, __VA_ARGS__); M(4, __VA_ARGS__); M(5, __VA_ARGS__); M(6, __VA_ARGS__); M(7, __VA_ARGS__)
    // ---- prologue of the FIRST tile only: K-tiles 0, 1, 2 into stages 0, 1, 2
    ALL4(DMA_A, fillA, rsA, 0u);
    ALL8(DMA_W, fillW, rsW, 0u);
    ALL4(DMA_A, fillA + STAGE, rsA, (uint32_t)(BK * 2));
    ALL8(DMA_W, fillW + STAGE, rsW, (uint32_t)(BK * 2));
    ALL4(DMA_A, fillA + 2 * STAGE, rsA, (uint32_t)(2 * BK * 2));
    ALL8(DMA_W, fillW + 2 * STAGE, rsW, (uint32_t)(2 * BK * 2));

    f32x4 acc[8][4];
    bf16x8 fa[2][4], fw[2][8];

#define RA(S, J) ds_read128<(J) * 2048>(fa[S][J], (S) ? rA1 : rA0)
#define RW(S, I) ds_read128<(I) * 2048>(fw[S][I], (S) ? rW1 : rW0)
#define RA_LOOP(S, J) do { if (!(BYA_GEMM5_ABLATE & 8)) RA(S, J); } while (0)
#define RW_LOOP(S, I) do { if (!(BYA_GEMM5_ABLATE & 8)) RW(S, I); } while (0)
    // K-tile 0 has landed (for everybody, behind the barrier): its k-step-0 fragments; then K-tile 1 for this wave -- the
    // state every later tile starts in (see SYNC_A)
    asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");
    RA(0, 0); RA(0, 1); RA(0, 2); RA(0, 3);
    RW(0, 0); RW(0, 1); RW(0, 2); RW(0, 3); RW(0, 4); RW(0, 5); RW(0, 6); RW(0, 7);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(12)" ::: "memory");
    rA0 += STAGE;
    rW0 += STAGE;

    for (;;) {
        const Tile128 nxt = coord(seq + 1);
        const i32x4 rsAn = a_rsrc(nxt), rsWn = w_rsrc(nxt);

        // One K-tile, variant V; (dA, dW, soff): where the 12 pieces this K-tile requests come from -- K-tile t + 3 of this
        // output tile, or K-tile t + 3 - nk of the next one
        auto ktile = [&](auto v_c, const i32x4& dA, const i32x4& dW, uint32_t soff) {
            constexpr char V = decltype(v_c)::value;
#define MF(S, I, J) \
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[I][J]) : "v"(fw[S][I]), "v"(fa[S][J]))
#define MFZ(S, I, J) \
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[I][J]) : "v"(fw[S][I]), "v"(fa[S][J]))
#if BYA_GEMM5_ABLATE & 2
#define DP(Q) do { (void)dA; (void)dW; (void)soff; } while (0)
#else
#define DP(Q) do { \
                if constexpr ((Q) < 4) { if (!(BYA_GEMM5_ABLATE & 32)) DMA_A((Q) & 3, fillA, dA, (BYA_GEMM5_ABLATE & 16) ? 0u : soff); } \
                else if (!(BYA_GEMM5_ABLATE & 64)) DMA_W(((Q) - 4) & 7, fillW, dW, (BYA_GEMM5_ABLATE & 16) ? 0u : soff); \
            } while (0)
#endif
            // A: K-tile 1 was waited for in front of the previous epilogue (or by the prologue); B: all but the 12 pieces of
            // K-tile t + 2 -- requested during K-tile t - 1 -- have landed, i.e. K-tile t + 1 has
#if BYA_GEMM5_ABLATE & 4
#define SYNC_A() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define SYNC_B() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(12)" ::: "memory")
#else
#define SYNC_A() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#if BYA_GEMM5_ABLATE & 32
#define SYNC_B() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory")
#elif BYA_GEMM5_ABLATE & 64
#define SYNC_B() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory")
#else
#define SYNC_B() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory")
#endif
#endif
#define NEXT() do { \
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
                const int s0 = st == 2 ? 0 : st + 1; \
                const uint32_t d1 = st == 2 ? (uint32_t)(-2 * STAGE) : (uint32_t)STAGE; \
                const uint32_t d0 = s0 == 2 ? (uint32_t)(-2 * STAGE) : (uint32_t)STAGE; \
                rA1 += d1; rW1 += d1; fillA += d1; fillW += d1; rA0 += d0; rW0 += d0; st = s0; \
            } while (0)
            // GENERATED-BEGIN (tools/gen_gemm_v5_schedule.py)
        if constexpr (V == 'A') {
            MFZ(0, 0, 0);  RA_LOOP(1, 0);
            MFZ(0, 0, 1);  RA_LOOP(1, 1);
            MFZ(0, 0, 2);  RA_LOOP(1, 2);
            MFZ(0, 0, 3);  RA_LOOP(1, 3);
            MFZ(0, 1, 0);  RW_LOOP(1, 0);
            MFZ(0, 1, 1);
            MFZ(0, 1, 2);  RW_LOOP(1, 1);
            MFZ(0, 1, 3);
            MFZ(0, 2, 0);  RW_LOOP(1, 2);
            MFZ(0, 2, 1);
            MFZ(0, 2, 2);  RW_LOOP(1, 3);
            MFZ(0, 2, 3);
            MFZ(0, 3, 0);  RW_LOOP(1, 4);
            MFZ(0, 3, 1);
            MFZ(0, 3, 2);  RW_LOOP(1, 5);
            MFZ(0, 3, 3);
            MFZ(0, 4, 0);  RW_LOOP(1, 6);
            MFZ(0, 4, 1);
            MFZ(0, 4, 2);  RW_LOOP(1, 7);
            MFZ(0, 4, 3);
            MFZ(0, 5, 0);
            MFZ(0, 5, 1);
            MFZ(0, 5, 2);
            MFZ(0, 5, 3);
            MFZ(0, 6, 0);
            MFZ(0, 6, 1);
            MFZ(0, 6, 2);
            MFZ(0, 6, 3);  SYNC_A();
            MFZ(0, 7, 0);  DP(0);
            MFZ(0, 7, 1);  RW_LOOP(0, 0);
            MFZ(0, 7, 2);  DP(1);
            MFZ(0, 7, 3);  RW_LOOP(0, 1);
            MF(1, 0, 0);  DP(2);
            MF(1, 0, 1);  RW_LOOP(0, 2);
            MF(1, 0, 2);  DP(3);
            MF(1, 0, 3);  RW_LOOP(0, 3);
            MF(1, 1, 0);  DP(4);
            MF(1, 1, 1);  RW_LOOP(0, 4);
            MF(1, 1, 2);  DP(5);
            MF(1, 1, 3);  RA_LOOP(0, 0);
            MF(1, 2, 0);  DP(6);
            MF(1, 2, 1);  RA_LOOP(0, 1);
            MF(1, 2, 2);  DP(7);
            MF(1, 2, 3);  RA_LOOP(0, 2);
            MF(1, 3, 0);  DP(8);
            MF(1, 3, 1);  RA_LOOP(0, 3);
            MF(1, 3, 2);  DP(9);
            MF(1, 3, 3);  RW_LOOP(0, 5);
            MF(1, 4, 0);  DP(10);
            MF(1, 4, 1);  RW_LOOP(0, 6);
            MF(1, 4, 2);  DP(11);
            MF(1, 4, 3);  RW_LOOP(0, 7);
            MF(1, 5, 0);
            MF(1, 5, 1);
            MF(1, 5, 2);
            MF(1, 5, 3);
            MF(1, 6, 0);
            MF(1, 6, 1);
            MF(1, 6, 2);
            MF(1, 6, 3);
            MF(1, 7, 0);
            MF(1, 7, 1);
            MF(1, 7, 2);
            MF(1, 7, 3);  NEXT();
        } else if constexpr (V == 'B') {
            MF(0, 0, 0);  RA_LOOP(1, 0);
            MF(0, 0, 1);  RA_LOOP(1, 1);
            MF(0, 0, 2);  RA_LOOP(1, 2);
            MF(0, 0, 3);  RA_LOOP(1, 3);
            MF(0, 1, 0);  RW_LOOP(1, 0);
            MF(0, 1, 1);
            MF(0, 1, 2);  RW_LOOP(1, 1);
            MF(0, 1, 3);
            MF(0, 2, 0);  RW_LOOP(1, 2);
            MF(0, 2, 1);
            MF(0, 2, 2);  RW_LOOP(1, 3);
            MF(0, 2, 3);
            MF(0, 3, 0);  RW_LOOP(1, 4);
            MF(0, 3, 1);
            MF(0, 3, 2);  RW_LOOP(1, 5);
            MF(0, 3, 3);
            MF(0, 4, 0);  RW_LOOP(1, 6);
            MF(0, 4, 1);
            MF(0, 4, 2);  RW_LOOP(1, 7);
            MF(0, 4, 3);
            MF(0, 5, 0);
            MF(0, 5, 1);
            MF(0, 5, 2);
            MF(0, 5, 3);
            MF(0, 6, 0);
            MF(0, 6, 1);
            MF(0, 6, 2);
            MF(0, 6, 3);  SYNC_B();
            MF(0, 7, 0);  DP(0);
            MF(0, 7, 1);  RW_LOOP(0, 0);
            MF(0, 7, 2);  DP(1);
            MF(0, 7, 3);  RW_LOOP(0, 1);
            MF(1, 0, 0);  DP(2);
            MF(1, 0, 1);  RW_LOOP(0, 2);
            MF(1, 0, 2);  DP(3);
            MF(1, 0, 3);  RW_LOOP(0, 3);
            MF(1, 1, 0);  DP(4);
            MF(1, 1, 1);  RW_LOOP(0, 4);
            MF(1, 1, 2);  DP(5);
            MF(1, 1, 3);  RA_LOOP(0, 0);
            MF(1, 2, 0);  DP(6);
            MF(1, 2, 1);  RA_LOOP(0, 1);
            MF(1, 2, 2);  DP(7);
            MF(1, 2, 3);  RA_LOOP(0, 2);
            MF(1, 3, 0);  DP(8);
            MF(1, 3, 1);  RA_LOOP(0, 3);
            MF(1, 3, 2);  DP(9);
            MF(1, 3, 3);  RW_LOOP(0, 5);
            MF(1, 4, 0);  DP(10);
            MF(1, 4, 1);  RW_LOOP(0, 6);
            MF(1, 4, 2);  DP(11);
            MF(1, 4, 3);  RW_LOOP(0, 7);
            MF(1, 5, 0);
            MF(1, 5, 1);
            MF(1, 5, 2);
            MF(1, 5, 3);
            MF(1, 6, 0);
            MF(1, 6, 1);
            MF(1, 6, 2);
            MF(1, 6, 3);
            MF(1, 7, 0);
            MF(1, 7, 1);
            MF(1, 7, 2);
            MF(1, 7, 3);  NEXT();
        }
            // GENERATED-END
#define KEEP4(F, S) asm volatile("" :: "v"(F[S][0]), "v"(F[S][1]), "v"(F[S][2]), "v"(F[S][3]))
#define KEEP8(F, S) asm volatile("" :: "v"(F[S][0]), "v"(F[S][1]), "v"(F[S][2]), "v"(F[S][3]), "v"(F[S][4]), \
                                      "v"(F[S][5]), "v"(F[S][6]), "v"(F[S][7]))
            KEEP4(fa, 0); KEEP8(fw, 0); KEEP4(fa, 1); KEEP8(fw, 1);
#undef KEEP4
#undef KEEP8
#undef MF
#undef MFZ
#undef DP
#undef SYNC_A
#undef SYNC_B
#undef NEXT
        };
        ktile(IntTag<'A'>{}, rsA, rsW, (uint32_t)(3 * BK * 2));
        for (int t = 1; t + 3 < nk; ++t) ktile(IntTag<'B'>{}, rsA, rsW, (uint32_t)((t + 3) * (BK * 2)));
        ktile(IntTag<'B'>{}, rsAn, rsWn, 0u);
        ktile(IntTag<'B'>{}, rsAn, rsWn, (uint32_t)(BK * 2));
        ktile(IntTag<'B'>{}, rsAn, rsWn, (uint32_t)(2 * BK * 2));
        // K-tile 1 of the next output tile has landed once only the 12 pieces of its K-tile 2 are in flight (SYNC_A relies on
        // it: the epilogue's own loads and stores go into the same counter); the MFMAs are inline asm, so pad their last
        // results before the epilogue reads them
        asm volatile("s_waitcnt vmcnt(12)\n\ts_nop 15\n\ts_nop 15" ::: "memory");

        if constexpr (BYA_GEMM5_ABLATE & 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" :: "a"(acc[i][0]), "a"(acc[i][1]), "a"(acc[i][2]), "a"(acc[i][3]));
        } else if constexpr (QKN) {
            epilogue_qkn<4>(p, cur.z, cur.m0 + wm * 64, cur.n0 + wn * 128, fr, fq, acc);
        } else {
            auto run = [&](auto act_tag) {
                epilogue_wide<decltype(act_tag)::value, 2, false, false, 4>(p, cur.z, cur.m0 + wm * 64, cur.n0 + wn * 128, fr, fq, acc,
                                                                            wave, lane);
            };
            dispatch_act_big(p.act, run);
        }

        if (!nxt.valid) break;
        ++seq;
        cur = nxt;
        rsA = rsAn;
        rsW = rsWn;
    }
#undef RA
#undef RW
#undef RA_LOOP
#undef RW_LOOP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the (empty-descriptor) pieces requested for the tile after the last
}

template <bool QKN>
int launch128p(const GemmArgs& a0, int batch, hipStream_t s) {
    GemmArgs a = a0;
    a.gm = 2 * gemm_group_m(a);                                // the same rows per group as the 256-row tiles' order
    const int tiles_m = (a.M + 127) / 128, tiles_n = (a.N + 255) / 256;
    const long long total = (long long)tiles_m * tiles_n * batch;
    const int blocks = (int)(total < 256 ? (total + 7) / 8 * 8 : 256);
    const size_t lds = 3 * (128 + 256) * BK * 2;
    static std::atomic<unsigned long long> attr_done{0};
    if (bya_allow_big_lds(reinterpret_cast<const void*>(gemm128p_kernel<QKN>), (int)lds, attr_done) != BYA_OK) return BYA_ERR_LAUNCH;
    BYA_LAUNCH((gemm128p_kernel<QKN>), dim3(blocks), dim3(256), lds, s, a, tiles_m, tiles_n, batch);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

}  // namespace

// callers (gemm.hip) have checked v4_eligible() and K >= 4 K-tiles
int bya_launch_gemm128p(const void* args, int batch, hipStream_t s) { return launch128p<false>(*static_cast<const GemmArgs*>(args), batch, s); }
int bya_launch_gemm128p_qkn(const void* args, int batch, hipStream_t s) { return launch128p<true>(*static_cast<const GemmArgs*>(args), batch, s); }
