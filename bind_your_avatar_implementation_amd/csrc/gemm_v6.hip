// Persistent 128x256x64 bf16 GEMM with LOADER WAVES: gemm_v5.hip's tile, ring and summation order, but the four waves that
// feed the matrix pipe issue no vector-memory instruction inside the K-loop.
//
// gemm_v5.hip's ablations price its LDS-DMA stream at 27 % of a launch, close to additive per piece: ISSUING a 1-KiB
// vector-memory instruction costs the issuing wave 11-18 cycles of matrix-pipe time (docs/history.md, round 3), and a 128-row
// tile needs twelve of them per 64 MFMAs.  A wave is the unit that pays, so here somebody else issues them: the workgroup has
// EIGHT waves, two per SIMD -- waves 0-3 compute (64 rows x 128 columns each, 128 accumulator AGPRs, fragments double-buffered:
// 236 of the 256 registers a wave may have at two waves per SIMD), waves 4-7 only request pieces.  A SIMD issues from both of its
// waves in the same cycle when they want different units, so the loader's buffer_load ... lds goes out beside the compute
// wave's MFMAs instead of between them.
//
//  * One barrier per K-tile for all eight waves.  B_g (g = K-tiles counted across this workgroup's output tiles; ring stage
//    g % 3): every compute wave has all fragments of K-tile g in registers, and every loader wave has seen its pieces of
//    K-tile g + 1 land.  Behind it the loaders refill stage g % 3 with K-tile g + 3 and the compute waves read the k-step-0
//    fragments of K-tile g + 1.
//  * The loaders' vmcnt holds nothing but pieces (the epilogue's loads and stores are the compute waves'): "all but the youngest
//    twelve have landed" is exact at every barrier, also across output tiles.
//  * 256 registers per wave: the next tile's first fragments are NOT read across the epilogue (48 registers); the tile re-reads
//    them behind its epilogue (they have landed: the previous barrier said so), a bubble of one LDS round trip per tile.
//
// Same bits as gemm_v4.hip / gemm_v5.hip.  tools/gen_gemm_v6_schedule.py holds the compute waves' placement table.  Compiled
// WITHOUT -amdgpu-mfma-vgpr-form (accumulators in AGPRs); the remarks in gemm_v4.hip about inline-asm MFMAs apply.
#include "gemm_wide_epilogue.h"
#include "options.h"

// timing-only ablation builds (tools/gemm_p128_ablate.py --v6; never the shipped library): 1 no epilogue, 2 the loader waves
// request nothing inside the K-loop, 16 every K-tile re-reads K-tile 0 (every request an L2 hit)
#ifndef BYA_GEMM6_ABLATE
#define BYA_GEMM6_ABLATE 0
#endif

namespace {

struct Tile128s { int z, m0, n0; bool valid; };

__global__ __launch_bounds__(512, 1) void gemm128s_kernel(GemmArgs p, int tiles_m, int tiles_n, int batch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 128, BN = 256, TILE_A = BM * BK * 2, STAGE = (BM + BN) * BK * 2;
    static_assert(TILE_A == 16384 && STAGE == 49152, "three stages fill 144 KiB");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.K / BK;                                   // >= 4 (launcher)

    // ---- this workgroup's output tiles: XCD x owns a contiguous range of the group-M tile order (gemm_v4.hip)
    const int per_z = tiles_m * tiles_n, total = per_z * batch;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int cq = total >> 3, cr = total & 7;
    const int base = (xcd < cr) ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
    const int end = base + cq + (xcd < cr ? 1 : 0);
    auto coord = [&](int seq) {
        Tile128s c;
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
    Tile128s cur = coord(seq);
    if (!cur.valid) return;                                    // (all eight waves alike)
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);

    if (wave >= 4) {
        // ================================================================================================ loader waves
        // wave 4 + w moves tile rows [32 w, 32 w + 32) of A (4 one-KiB pieces of 8 rows) and LDS slot rows [64 w, 64 w + 64) of W
        // (8 pieces) of every K-tile; layout and source-side swizzle: gemm_v5.hip
        const int lw = wave - 4;
        uint32_t voA[4], voW[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rl = lw * 32 + q * 8 + (lane >> 3);
            voA[q] = (uint32_t)rl * (uint32_t)(p.lda * 2) + ((lane & 7) ^ ((rl >> 1) & 7)) * 16;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int rl = lw * 64 + q * 8 + (lane >> 3);
            const int r = rl & 15, i = (rl >> 4) & 7;
            const int wcol = (rl & 128) + (((r & 3) << 2) | (r >> 2)) * 8 + i;
            voW[q] = (uint32_t)wcol * (uint32_t)(p.ldw * 2) + ((lane & 7) ^ ((rl >> 1) & 7)) * 16;
        }
        auto a_rsrc = [&](const Tile128s& c) {
            const long long left = ((long long)(p.M - 1 - c.m0) * p.lda + p.K) * 2;
            return raw_rsrc(p.A + (long long)c.z * p.a_bs + (long long)c.m0 * p.lda, c.valid && left > 0 ? (uint32_t)left : 0u);
        };
        auto w_rsrc = [&](const Tile128s& c) {
            const long long left = ((long long)(p.N - 1 - c.n0) * p.ldw + p.K) * 2;
            return raw_rsrc(p.W + (long long)c.n0 * p.ldw, c.valid && left > 0 ? (uint32_t)left : 0u);
        };
        uint32_t fillA = __builtin_amdgcn_readfirstlane(lds0 + lw * 32 * 128);
        uint32_t fillW = __builtin_amdgcn_readfirstlane(lds0 + TILE_A + lw * 64 * 128);
        int st = 0;                                            // ring stage of K-tile g = the one to refill behind B_g
        i32x4 rsA = a_rsrc(cur), rsW = w_rsrc(cur);
#define DMA_A(Q, BASE, RS, SOFF) dma_piece<(Q) * 1024>(BASE, voA[Q], RS, SOFF)
#define DMA_W(Q, BASE, RS, SOFF) dma_piece<(Q) * 1024>(BASE, voW[Q], RS, SOFF)
#define ALL4(M, ...) M(0, __VA_ARGS__); M(1, __VA_ARGS__); M(2, __VA_ARGS__); M(3, __VA_ARGS__)
#define ALL8(M, ...) ALL4(M, __VA_ARGS__); M(4, __VA_ARGS__); M(5, __VA_ARGS__); M(6, __VA_ARGS__); M(7, __VA_ARGS__)
        ALL4(DMA_A, fillA, rsA, 0u);
        ALL8(DMA_W, fillW, rsW, 0u);
        ALL4(DMA_A, fillA + STAGE, rsA, (uint32_t)(BK * 2));
        ALL8(DMA_W, fillW + STAGE, rsW, (uint32_t)(BK * 2));
        ALL4(DMA_A, fillA + 2 * STAGE, rsA, (uint32_t)(2 * BK * 2));
        ALL8(DMA_W, fillW + 2 * STAGE, rsW, (uint32_t)(2 * BK * 2));
        asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");        // B_-1: K-tile 0 of the first tile has landed
        // behind B_g: K-tile g + 3 (soff / descriptors say whose) into the stage K-tile g just left
        auto refill = [&](const i32x4& dA, const i32x4& dW, uint32_t soff) {
            asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");    // B_g: my pieces of K-tile g + 1 have landed
            if (!(BYA_GEMM6_ABLATE & 2)) {
                ALL4(DMA_A, fillA, dA, (BYA_GEMM6_ABLATE & 16) ? 0u : soff);
                ALL8(DMA_W, fillW, dW, (BYA_GEMM6_ABLATE & 16) ? 0u : soff);
            }
            const uint32_t d = st == 2 ? (uint32_t)(-2 * STAGE) : (uint32_t)STAGE;
            fillA += d;
            fillW += d;
            st = st == 2 ? 0 : st + 1;
        };
        for (;;) {
            const Tile128s nxt = coord(seq + 1);
            const i32x4 rsAn = a_rsrc(nxt), rsWn = w_rsrc(nxt);
            for (int t = 0; t + 3 < nk; ++t) refill(rsA, rsW, (uint32_t)((t + 3) * (BK * 2)));
            refill(rsAn, rsWn, 0u);
            refill(rsAn, rsWn, (uint32_t)(BK * 2));
            refill(rsAn, rsWn, (uint32_t)(2 * BK * 2));
            if (!nxt.valid) break;
            ++seq;
            rsA = rsAn;
            rsW = rsWn;
        }
#undef DMA_A
#undef DMA_W
#undef ALL4
#undef ALL8
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the (empty-descriptor) pieces requested for the tile after the last
        return;
    }

    // ==================================================================================================== compute waves
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    // fragment read addresses in stage 0 (XOR swizzle on (row >> 1) & 7; row blocks are 16 rows = 2048 bytes apart): k-step 0 / 1
    const int a_row = wm * 64 + fr, w_row = wn * 128 + fr;
    const int a_sw = (a_row >> 1) & 7, w_sw = (w_row >> 1) & 7;
    const uint32_t bA0 = lds0 + a_row * 128 + ((fq ^ a_sw) << 4), bA1 = lds0 + a_row * 128 + (((4 + fq) ^ a_sw) << 4);
    const uint32_t bW0 = lds0 + TILE_A + w_row * 128 + ((fq ^ w_sw) << 4);
    const uint32_t bW1 = lds0 + TILE_A + w_row * 128 + (((4 + fq) ^ w_sw) << 4);
    uint32_t off1 = 0u, off0 = (uint32_t)STAGE;               // ring stage (as a byte offset) of K-tile g, of K-tile g + 1
    uint32_t rA1 = bA1, rW1 = bW1, rA0 = bA0 + STAGE, rW0 = bW0 + STAGE;

    f32x4 acc[8][4];
    bf16x8 fa[2][4], fw[2][8];
    asm volatile("s_barrier" ::: "memory");                    // B_-1

#define RA(S, J) ds_read128<(J) * 2048>(fa[S][J], (S) ? rA1 : rA0)
#define RW(S, I) ds_read128<(I) * 2048>(fw[S][I], (S) ? rW1 : rW0)
    for (;;) {
        const Tile128s nxt = coord(seq + 1);
        // the k-step-0 fragments of this tile's first K-tile (it landed before the previous barrier)
        {
            const uint32_t tA = bA0 + off1, tW = bW0 + off1;
            ds_read128<0 * 2048>(fa[0][0], tA); ds_read128<1 * 2048>(fa[0][1], tA); ds_read128<2 * 2048>(fa[0][2], tA); ds_read128<3 * 2048>(fa[0][3], tA);
            ds_read128<0 * 2048>(fw[0][0], tW); ds_read128<1 * 2048>(fw[0][1], tW); ds_read128<2 * 2048>(fw[0][2], tW); ds_read128<3 * 2048>(fw[0][3], tW);
            ds_read128<4 * 2048>(fw[0][4], tW); ds_read128<5 * 2048>(fw[0][5], tW); ds_read128<6 * 2048>(fw[0][6], tW); ds_read128<7 * 2048>(fw[0][7], tW);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        auto ktile = [&](auto v_c) {
            constexpr char V = decltype(v_c)::value;
#define MF(S, I, J) \
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[I][J]) : "v"(fw[S][I]), "v"(fa[S][J]))
#define MFZ(S, I, J) \
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[I][J]) : "v"(fw[S][I]), "v"(fa[S][J]))
#define SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define NEXT() do { \
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
                off1 = off0; \
                off0 = off0 == (uint32_t)(2 * STAGE) ? 0u : off0 + (uint32_t)STAGE; \
                rA1 = bA1 + off1; rW1 = bW1 + off1; rA0 = bA0 + off0; rW0 = bW0 + off0; \
            } while (0)
            // GENERATED-BEGIN (tools/gen_gemm_v6_schedule.py)
        if constexpr (V == 'A') {
            MFZ(0, 0, 0);  RA(1, 0);
            MFZ(0, 0, 1);  RA(1, 1);
            MFZ(0, 0, 2);  RA(1, 2);
            MFZ(0, 0, 3);  RA(1, 3);
            MFZ(0, 1, 0);  RW(1, 0);
            MFZ(0, 1, 1);
            MFZ(0, 1, 2);  RW(1, 1);
            MFZ(0, 1, 3);
            MFZ(0, 2, 0);  RW(1, 2);
            MFZ(0, 2, 1);
            MFZ(0, 2, 2);  RW(1, 3);
            MFZ(0, 2, 3);
            MFZ(0, 3, 0);  RW(1, 4);
            MFZ(0, 3, 1);
            MFZ(0, 3, 2);  RW(1, 5);
            MFZ(0, 3, 3);
            MFZ(0, 4, 0);  RW(1, 6);
            MFZ(0, 4, 1);
            MFZ(0, 4, 2);  RW(1, 7);
            MFZ(0, 4, 3);
            MFZ(0, 5, 0);
            MFZ(0, 5, 1);
            MFZ(0, 5, 2);
            MFZ(0, 5, 3);
            MFZ(0, 6, 0);
            MFZ(0, 6, 1);
            MFZ(0, 6, 2);
            MFZ(0, 6, 3);  SYNC();
            MFZ(0, 7, 0);
            MFZ(0, 7, 1);  RW(0, 0);
            MFZ(0, 7, 2);
            MFZ(0, 7, 3);  RW(0, 1);
            MF(1, 0, 0);
            MF(1, 0, 1);  RW(0, 2);
            MF(1, 0, 2);
            MF(1, 0, 3);  RW(0, 3);
            MF(1, 1, 0);
            MF(1, 1, 1);  RW(0, 4);
            MF(1, 1, 2);
            MF(1, 1, 3);  RA(0, 0);
            MF(1, 2, 0);
            MF(1, 2, 1);  RA(0, 1);
            MF(1, 2, 2);
            MF(1, 2, 3);  RA(0, 2);
            MF(1, 3, 0);
            MF(1, 3, 1);  RA(0, 3);
            MF(1, 3, 2);
            MF(1, 3, 3);  RW(0, 5);
            MF(1, 4, 0);
            MF(1, 4, 1);  RW(0, 6);
            MF(1, 4, 2);
            MF(1, 4, 3);  RW(0, 7);
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
            MF(0, 0, 0);  RA(1, 0);
            MF(0, 0, 1);  RA(1, 1);
            MF(0, 0, 2);  RA(1, 2);
            MF(0, 0, 3);  RA(1, 3);
            MF(0, 1, 0);  RW(1, 0);
            MF(0, 1, 1);
            MF(0, 1, 2);  RW(1, 1);
            MF(0, 1, 3);
            MF(0, 2, 0);  RW(1, 2);
            MF(0, 2, 1);
            MF(0, 2, 2);  RW(1, 3);
            MF(0, 2, 3);
            MF(0, 3, 0);  RW(1, 4);
            MF(0, 3, 1);
            MF(0, 3, 2);  RW(1, 5);
            MF(0, 3, 3);
            MF(0, 4, 0);  RW(1, 6);
            MF(0, 4, 1);
            MF(0, 4, 2);  RW(1, 7);
            MF(0, 4, 3);
            MF(0, 5, 0);
            MF(0, 5, 1);
            MF(0, 5, 2);
            MF(0, 5, 3);
            MF(0, 6, 0);
            MF(0, 6, 1);
            MF(0, 6, 2);
            MF(0, 6, 3);  SYNC();
            MF(0, 7, 0);
            MF(0, 7, 1);  RW(0, 0);
            MF(0, 7, 2);
            MF(0, 7, 3);  RW(0, 1);
            MF(1, 0, 0);
            MF(1, 0, 1);  RW(0, 2);
            MF(1, 0, 2);
            MF(1, 0, 3);  RW(0, 3);
            MF(1, 1, 0);
            MF(1, 1, 1);  RW(0, 4);
            MF(1, 1, 2);
            MF(1, 1, 3);  RA(0, 0);
            MF(1, 2, 0);
            MF(1, 2, 1);  RA(0, 1);
            MF(1, 2, 2);
            MF(1, 2, 3);  RA(0, 2);
            MF(1, 3, 0);
            MF(1, 3, 1);  RA(0, 3);
            MF(1, 3, 2);
            MF(1, 3, 3);  RW(0, 5);
            MF(1, 4, 0);
            MF(1, 4, 1);  RW(0, 6);
            MF(1, 4, 2);
            MF(1, 4, 3);  RW(0, 7);
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
        } else if constexpr (V == 'L') {
            MF(0, 0, 0);  RA(1, 0);
            MF(0, 0, 1);  RA(1, 1);
            MF(0, 0, 2);  RA(1, 2);
            MF(0, 0, 3);  RA(1, 3);
            MF(0, 1, 0);  RW(1, 0);
            MF(0, 1, 1);
            MF(0, 1, 2);  RW(1, 1);
            MF(0, 1, 3);
            MF(0, 2, 0);  RW(1, 2);
            MF(0, 2, 1);
            MF(0, 2, 2);  RW(1, 3);
            MF(0, 2, 3);
            MF(0, 3, 0);  RW(1, 4);
            MF(0, 3, 1);
            MF(0, 3, 2);  RW(1, 5);
            MF(0, 3, 3);
            MF(0, 4, 0);  RW(1, 6);
            MF(0, 4, 1);
            MF(0, 4, 2);  RW(1, 7);
            MF(0, 4, 3);
            MF(0, 5, 0);
            MF(0, 5, 1);
            MF(0, 5, 2);
            MF(0, 5, 3);
            MF(0, 6, 0);
            MF(0, 6, 1);
            MF(0, 6, 2);
            MF(0, 6, 3);  SYNC();
            MF(0, 7, 0);
            MF(0, 7, 1);
            MF(0, 7, 2);
            MF(0, 7, 3);
            MF(1, 0, 0);
            MF(1, 0, 1);
            MF(1, 0, 2);
            MF(1, 0, 3);
            MF(1, 1, 0);
            MF(1, 1, 1);
            MF(1, 1, 2);
            MF(1, 1, 3);
            MF(1, 2, 0);
            MF(1, 2, 1);
            MF(1, 2, 2);
            MF(1, 2, 3);
            MF(1, 3, 0);
            MF(1, 3, 1);
            MF(1, 3, 2);
            MF(1, 3, 3);
            MF(1, 4, 0);
            MF(1, 4, 1);
            MF(1, 4, 2);
            MF(1, 4, 3);
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
#undef SYNC
#undef NEXT
        };
        ktile(IntTag<'A'>{});
        for (int t = 1; t + 1 < nk; ++t) ktile(IntTag<'B'>{});
        ktile(IntTag<'L'>{});
        // the MFMAs are inline asm: pad their last results before the epilogue reads them
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

        if constexpr (BYA_GEMM6_ABLATE & 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" :: "a"(acc[i][0]), "a"(acc[i][1]), "a"(acc[i][2]), "a"(acc[i][3]));
        } else {
            auto run = [&](auto act_tag) {
                epilogue_wide<decltype(act_tag)::value, 1, false, false, 4>(p, cur.z, cur.m0 + wm * 64, cur.n0 + wn * 128, fr, fq, acc, wave, lane);
            };
            dispatch_act_big(p.act, run);
        }

        if (!nxt.valid) break;
        ++seq;
        cur = nxt;
    }
#undef RA
#undef RW
}

}  // namespace

// callers (gemm.hip) have checked v4_eligible() and K >= 4 K-tiles.  (The q|k|v projection's norm epilogue needs more than the 128
// arch registers a wave has here -- built and measured: 41 spilled registers, 804 against gemm_v5.hip's 937 TFLOP/s at 2222 x 9216 --
// and its 128-row part is the 430-row tail of a row plan: it stays on gemm_v5.hip.)
int bya_launch_gemm128s(const void* args, int batch, hipStream_t s) {
    GemmArgs a = *static_cast<const GemmArgs*>(args);
#ifdef BYA_GEMM6_GM                                             // (side builds of a group-M sweep)
    a.gm = BYA_GEMM6_GM;
#else
    a.gm = 2 * gemm_group_m(a);                                // the same rows per group as the 256-row tiles' order
#endif
    const int tiles_m = (a.M + 127) / 128, tiles_n = (a.N + 255) / 256;
    const long long total = (long long)tiles_m * tiles_n * batch;
    const int blocks = (int)(total < 256 ? (total + 7) / 8 * 8 : 256);
    const size_t lds = 3 * (128 + 256) * BK * 2;
    static std::atomic<unsigned long long> attr_done{0};
    if (bya_allow_big_lds(reinterpret_cast<const void*>(gemm128s_kernel), (int)lds, attr_done) != BYA_OK) return BYA_ERR_LAUNCH;
    BYA_LAUNCH(gemm128s_kernel, dim3(blocks), dim3(512), lds, s, a, tiles_m, tiles_n, batch);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}
