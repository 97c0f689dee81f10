// Persistent 256x256x64 bf16 GEMM: ONE wave per SIMD, LDS-DMA two K-tiles ahead, next output tile prefetched under the drain.
//
// Same operands, fused epilogue and ABI as gemm256_kernel (gemm.hip).  The K-loop is the one of gemm_v3.hip (4 waves,
// 128 x 128 per wave, 256 accumulator registers in AGPRs, both k-steps' fragments in 128 VGPRs, a 2-stage 128 KiB LDS
// ring whose stages are released in HALVES so that an LDS-DMA piece has one and a half K-tile periods to land).  What
// this kernel adds is everything AROUND the K-loop, which at K = 3072 was a fifth of the time:
//
//  * PERSISTENT workgroups: one per CU, each walks its share of the output tiles (every XCD owns a contiguous range of
//    the group-M tile order, so the A / W panels a round of 32 tiles shares are served by that XCD's L2).
//  * NO PROLOGUE BUBBLE: the first two K-tiles of the NEXT output tile are requested while the current one drains --
//    K-tile 0 during the last-but-one K-tile (its stage is free after that K-tile's barrier 3), K-tile 1 during the last
//    K-tile -- so they have landed when the epilogue ends and the next K-loop starts with its fragment reads.
//  * NO ACCUMULATOR ZEROING: the first K-tile's first k-step issues its MFMAs with C = 0.
//  * 16-BYTE EPILOGUE ACCESSES: the W rows of a wave's 128-column span are staged in a permuted order (LDS slot (i, r)
//    holds column ((r & 3) * 4 + (r >> 2)) * 8 + i), which makes a lane's eight accumulator tiles i = 0..7 hold EIGHT
//    CONSECUTIVE output columns: bias / gate / residual are read and C is written 16 bytes per lane, 64 contiguous
//    bytes per row and instruction -- half the memory instructions of the 8-byte epilogue, all requested in bursts
//    (gemm_common.h explains why that matters: one serialized round trip per access otherwise).
//
//  * THE LAST, PARTIAL ROUND IS SPLIT ALONG K (when the caller registered a workspace, bya_set_gemm_workspace, and option
//    gemm_splitk is 1 -- the default of rounds 2-6; since then gemm.hip's row plan sends those rows to 128-row tiles instead): an XCD
//    whose tile count is not a multiple of its 32 workgroups has R < 32 tiles left after the full rounds; each of them is
//    cut into p = min(32 / R, K-tiles / 40) K-ranges that p workgroups compute at the same time (shorter ranges do not
//    pay for the slab exchange: K = 12288 splits, K = 3072 does not).  p - 1 of them write
//    their 256 x 256 fp32 partial sums to a slab (write-through stores, drained, then one agent-scope counter increment);
//    the workgroup with the highest index of the p -- dispatched last, so it never waits for a workgroup that has no
//    CU yet -- polls the counter, adds the slabs inside its epilogue bursts and resets the counter.  17776 x 3072
//    outputs are 3.28 rounds of tiles: the 0.28 used to cost a whole round (or a row split onto the slower 128 x 128
//    kernel); per-rank shapes of the sharded step (2222 rows: 108 tiles on 256 CUs) use every CU this way.
//
// K-tile variants (tools/gen_gemm_v4_schedule.py holds the placement tables and rewrites the GENERATED block):
//   A first (C = 0), B steady, C last-but-one (+ next tile's K-tile 0), D last (+ next tile's K-tile 1).  K >= 192.
//
// Inline-asm MFMAs are invisible to hipcc: (1) nothing tells it that an LDS return must not land in a register a queued
// MFMA still has to read, so every fragment stays allocated to its fragment for the whole K-tile (KEEP8; without it the
// drain tile of gemm_v3 computed wrong sums); (2) it pads no MFMA -> accvgpr_read hazard, so the epilogue starts behind
// explicit s_nops.  This translation unit is compiled WITHOUT -amdgpu-mfma-vgpr-form (accumulators in AGPRs).
#include "gemm_wide_epilogue.h"
#include "options.h"

#ifndef BYA_GEMM_ABLATE
#define BYA_GEMM_ABLATE 0
#endif

namespace {

#ifdef BYA_GEMM_TIMELINE
// Ablation builds only (tools/gemm_epilogue_timeline.py; never the shipped library): wall-clock stamps (100 MHz) of every
// unit of an UNSPLIT launch, written by thread 0 to the last slab of the workspace -- [blockIdx][seq & 1][8] x u64: unit start,
// K-loop end, -, -, unit end, -, K-tiles, valid.  The split instance is not instrumented: the stamps cost it 130 spilled
// registers and what they then measure is scratch traffic.
#define TL_STAMP(K) do { if (!SPLIT && tid == 0 && p.ws_slabs) tl[K] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TL_STAMP(K) do {} while (0)
#endif

// One unit of work: K-tiles [k0, k0 + nkk) of output tile (z, m0, n0).  role 0: the whole tile (ordinary epilogue);
// 1: a partial sum, written to slab `slab`; 2: the finisher of a tile split into `parts` K-ranges (adds slabs
// slab, slab + slab_step, ... then the ordinary epilogue; counter `ctr` says when they are complete).
struct TileCoord { int z, m0, n0; bool valid; int k0, nkk, role, parts, slab, slab_step, ctr; };
// K-tiles per K-range below which a split does not pay (the slab exchange costs ~20 us: measured break-even ~30 K-tiles)
constexpr int DEFAULT_MIN_SPLIT_KTILES = 40;

// SPLIT = false: the instance for launches that will not split a tile (no workspace, short K, or nothing left over): its
// epilogue has no slab branch -- that branch alone costs the ordinary path 1.5-2.5 % on K = 3072 shapes (same-box A/B,
// profiles/history/r2_gemm_epilogue_variants_same_box.txt) because it cuts the unrolled epilogue into blocks.
template <bool SPLIT, bool CONV = false, bool QKN = false>
__global__ __launch_bounds__(256, 1) void gemm256p_kernel(GemmArgs p, int tiles_m, int tiles_n, int batch, int split_arg,
                                                          int min_seg, unsigned epoch) {
    static_assert(!(SPLIT && CONV), "the convolution instance does not split K");
    static_assert(!(QKN && (SPLIT || CONV)), "the q/k-norm instance is a plain, unsplit GEMM");
    const int split = SPLIT ? split_arg : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 256, STAGE = (BM + BN) * BK * 2, TILE_A = BM * BK * 2;
    static_assert(STAGE == 65536, "stage flip uses one address bit");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.K / BK;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;

    // ---- this workgroup's output tiles: XCD x (= blockIdx % 8 under round-robin dispatch; speed only) owns a contiguous
    // range of the tile order, its workgroups take every (gridDim / 8)-th tile of it, round after round
    const int per_z = tiles_m * tiles_n, total = per_z * batch;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int cq = total >> 3, cr = total & 7;
    const int base = (xcd < cr) ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
    const int end = base + cq + (xcd < cr ? 1 : 0);
    // unit `seq` of this workgroup: full rounds first (tile base + slot + seq * slots), then -- with a workspace -- its
    // K-range of one of the R tiles the full rounds leave over (see the top)
    const int n_x = end - base;
    const int full = split ? n_x / slots : 0x7fffffff;
    const int R = split ? n_x - full * slots : 0;
    int parts = 1;
    if (split && R > 0) {
        parts = slots / R;
        if (parts > nk / min_seg) parts = nk / min_seg;
        if (parts < 1) parts = 1;
    }
    auto coord = [&](int seq) {
        TileCoord c;
        c.k0 = 0; c.nkk = nk; c.role = 0; c.parts = 1; c.slab = 0; c.slab_step = 0; c.ctr = 0;
        int id;
        if (seq < full) {
            id = base + slot + seq * slots;
            c.valid = id < end;
        } else if (seq == full && R > 0 && slot < R * parts) {
            const int j = slot / R, r = slot - j * R;
            id = base + full * slots + r;
            c.valid = true;
            if (parts > 1) {
                c.k0 = j * nk / parts;
                c.nkk = (j + 1) * nk / parts - c.k0;
                c.parts = parts;
                c.role = j == parts - 1 ? 2 : 1;
                c.slab = c.role == 1 ? xcd * 32 + j * R + r : xcd * 32 + r;
                c.slab_step = R;
                c.ctr = xcd * 32 + r;
            }
        } else {
            id = base;
            c.valid = false;
        }
        const int idz = c.valid ? id : base;
        c.z = idz / per_z;
        const int idt = idz - c.z * per_z;
        // group-M order: GM row tiles sweep a column tile before the order moves on -- what the 32 concurrent tiles of an XCD share
        // in its L2.  Chosen per SHAPE by the launchers (gemm_group_m, gemm_common.h; round 3 had one compromise, 4): same-box sweep of
        // round 6 (tools/gemm_gm_probe.py, profiles/r6_k_gemm_gm_probe.json): FF2 (K = 12288) 1283 TFLOP/s at 2, 1268 at 4, 1244
        // at 8; FF1 (N = 12288) 1244 / 1265 / 1278; QKV 1250 / 1295 / 1302; 16 loses everywhere.  (-DBYA_GEMM_GM=n pins it: the probe's builds.)
#ifdef BYA_GEMM_GM
        constexpr int GM = BYA_GEMM_GM;
#else
        const int GM = p.gm;
#endif
        const int per_group = GM * tiles_n;
        const int group = idt / per_group, first_m = group * GM;
        const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
        const int in_g = idt - group * per_group;
        c.m0 = (first_m + in_g % gsz) * BM;
        c.n0 = (in_g / gsz) * BN;
        return c;
    };
    int seq = 0;
    TileCoord cur = coord(seq);
    if (!cur.valid) return;

    // fragment read addresses (XOR swizzle on (row >> 1) & 7; row blocks are 16 rows = 2048 bytes apart)
    const int a_row = wm * 128 + fr, w_row = wn * 128 + fr;
    const int a_sw = (a_row >> 1) & 7, w_sw = (w_row >> 1) & 7;
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
    uint32_t cA0 = lds0 + a_row * 128 + ((fq ^ a_sw) << 4), cA1 = lds0 + a_row * 128 + (((4 + fq) ^ a_sw) << 4);
    uint32_t cW0 = lds0 + TILE_A + w_row * 128 + ((fq ^ w_sw) << 4);
    uint32_t cW1 = lds0 + TILE_A + w_row * 128 + (((4 + fq) ^ w_sw) << 4);
    uint32_t fill = __builtin_amdgcn_readfirstlane(lds0 + wave * 64 * 128);     // this wave's first A piece, current stage

    // staging: wave w moves LDS slot rows [64w, 64w + 64) of the A tile and of the W tile, 8 one-KiB pieces (8 rows) each.
    // A slot rows are tile rows; W slot row s = 128 h + 16 i + r holds tile column 128 h + ((r & 3) * 4 + (r >> 2)) * 8 + i
    // (see the top).  The lane loads the source chunk that belongs at its linear LDS position (source-side XOR swizzle).
    // The per-lane byte offsets are relative to the TILE origin and never change; the tile origin lives in the buffer
    // descriptor (base advanced to the tile's first row, size = what is left of the matrix), so rows past M / N fall
    // outside the descriptor and arrive as zeros, and an invalid (past-the-end) tile gets an empty descriptor.
    uint32_t voA[8], voW[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int rl = wave * 64 + q * 8 + (lane >> 3);
        const int r = rl & 15, i = (rl >> 4) & 7;
        const int wcol = (rl & 128) + (((r & 3) << 2) | (r >> 2)) * 8 + i;
        const int chunk16 = ((lane & 7) ^ ((rl >> 1) & 7)) * 16;
        voA[q] = (uint32_t)rl * (uint32_t)(p.lda * 2) + chunk16;
        voW[q] = (uint32_t)wcol * (uint32_t)(p.ldw * 2) + chunk16;
    }
    auto a_rsrc = [&](const TileCoord& c) {
        long long left = CONV ? p.conv_a_bytes - (long long)c.m0 * p.lda * 2
                              : ((long long)(p.M - 1 - c.m0) * p.lda + p.K - c.k0 * BK) * 2;
        if (CONV && left > 0xffffffffLL) left = 0xffffffffLL;          // (a tile reaches < 2^31 bytes past its first row)
        return raw_rsrc(p.A + (long long)c.z * p.a_bs + (long long)c.m0 * p.lda + c.k0 * BK,
                        c.valid && left > 0 ? (uint32_t)left : 0u);
    };
    auto w_rsrc = [&](const TileCoord& c) {
        const long long left = ((long long)(p.N - 1 - c.n0) * p.ldw + p.K - c.k0 * BK) * 2;
        return raw_rsrc(p.W + (long long)c.n0 * p.ldw + c.k0 * BK, c.valid && left > 0 ? (uint32_t)left : 0u);
    };
    i32x4 rsA = a_rsrc(cur), rsW = w_rsrc(cur);

#define DMA_A(Q, BASE, VO, RS, SOFF) dma_piece<(Q) * 1024>(BASE, VO[Q], RS, SOFF)
#define DMA_W(Q, BASE, VO, RS, SOFF) dma_piece<TILE_A + (Q) * 1024>(BASE, VO[Q], RS, SOFF)
#define ALL8(M, ...) M(0, __VA_ARGS__); M(1, __VA_ARGS__); M(2, __VA_ARGS__); M(3, __VA_ARGS__); \
                     M(4, __VA_ARGS__); M(5, __VA_ARGS__); M(6, __VA_ARGS__); M(7, __VA_ARGS__)
    // ---- prologue of the FIRST tile only: K-tiles 0 and 1
    ALL8(DMA_A, fill, voA, rsA, 0u);
    ALL8(DMA_W, fill, voW, rsW, 0u);
    ALL8(DMA_A, fill ^ STAGE, voA, rsA, (uint32_t)(BK * 2));
    ALL8(DMA_W, fill ^ STAGE, voW, rsW, (uint32_t)(BK * 2));
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");

    f32x4 acc[8][8];
    bf16x8 fa[2][8], fw[2][8];

    for (;;) {
#ifdef BYA_GEMM_TIMELINE
        unsigned long long* const tl = reinterpret_cast<unsigned long long*>(p.ws_slabs + (size_t)(GEMM_WS_SLABS - 1) * (GEMM_WS_SLAB_BYTES / 4)) +
                                       ((size_t)blockIdx.x * 2 + (seq & 1)) * 8;
        TL_STAMP(0);
        if (!SPLIT && tid == 0 && p.ws_slabs) { tl[6] = (unsigned long long)cur.nkk; tl[7] = cur.valid ? 1ull : 0ull; }
#endif
        // ---- K-tile 0 of this output tile has landed for this wave (prologue wait / the wait in front of the previous
        // epilogue); make that true for everybody, then fetch its k-step-0 fragments
        asm volatile("s_barrier" ::: "memory");
#define RA(S, J) ds_read128<(J) * 2048>(fa[S][J], (S) ? cA1 : cA0)
#define RW(S, I) ds_read128<(I) * 2048>(fw[S][I], (S) ? cW1 : cW0)
        RA(0, 0); RA(0, 1); RA(0, 2); RA(0, 3); RA(0, 4); RA(0, 5); RA(0, 6); RA(0, 7);
        RW(0, 0); RW(0, 1); RW(0, 2); RW(0, 3); RW(0, 4); RW(0, 5); RW(0, 6); RW(0, 7);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        const TileCoord nxt = coord(seq + 1);
        const i32x4 rsAn = a_rsrc(nxt), rsWn = w_rsrc(nxt);
        const int unk = cur.nkk;                              // K-tiles of this unit (>= 3)

        // One K-tile, variant V (see the top); t = its index inside the output tile.
        auto ktile = [&](int t, auto v_c) {
            constexpr char V = decltype(v_c)::value;
            const uint32_t soff = (BYA_GEMM_ABLATE & 4) ? 0u : (uint32_t)((t + 2) * (BK * 2));     // (ablation 4: every K-tile re-reads K-tile 0)
            uint32_t soffA = soff;                            // (W's K-tiles are plain columns in either case)
            if constexpr (CONV) {
                // K-tile t + 2 = channel group cg of tap (dt, dh, dw): the tile's rows, shifted by the tap (scalar arithmetic)
                const uint32_t kt = (uint32_t)(t + 2), tap = kt >> p.conv_cpg_log2, cg = kt - (tap << p.conv_cpg_log2);
                const uint32_t dt = tap / 9u, r9 = tap - dt * 9u, dh = r9 / 3u, dw = r9 - dh * 3u;
                soffA = (((dt * (uint32_t)p.conv_Hp + dh) * (uint32_t)p.conv_Wp + dw) * (uint32_t)p.lda + cg * 64u) * 2u;
            }
#define MF(S, I, J) \
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[I][J]) : "v"(fw[S][I]), "v"(fa[S][J]))
#define MFZ(S, I, J) \
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[I][J]) : "v"(fw[S][I]), "v"(fa[S][J]))
#if BYA_GEMM_ABLATE & 1                 // timing-only ablation builds (tools/): no LDS-DMA inside the K-loop
#define DA(Q) do {} while (0)
#define DW(Q) do {} while (0)
#else
#define DA(Q) DMA_A(Q, fill, voA, rsA, soffA)
#define DW(Q) DMA_W(Q, fill, voW, rsW, soff)
#endif
#define PA(Q) DMA_A(Q, fill, voA, rsAn, 0u)
#define PW(Q) DMA_W(Q, fill, voW, rsWn, 0u)
#define QA(Q) DMA_A(Q, fill, voA, rsAn, (uint32_t)(BK * 2))
#define QW(Q) DMA_W(Q, fill, voW, rsWn, (uint32_t)(BK * 2))
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#if BYA_GEMM_ABLATE & 2                 // ... no barriers inside the K-loop
#define BAR() do {} while (0)
#else
#define BAR() asm volatile("s_barrier" ::: "memory")
#endif
#define FLIP0() do { cA0 ^= STAGE; cW0 ^= STAGE; } while (0)
#define FLIP1() do { cA1 ^= STAGE; cW1 ^= STAGE; fill ^= STAGE; } while (0)
            // GENERATED-BEGIN (tools/gen_gemm_v4_schedule.py)
        if constexpr (V == 'A') {
            MFZ(0, 0, 0);  RA(1, 0);
            MFZ(0, 0, 1);  RA(1, 1);
            MFZ(0, 0, 2);  RA(1, 2);
            MFZ(0, 0, 3);  RA(1, 3);
            MFZ(0, 0, 4);  RA(1, 4);
            MFZ(0, 0, 5);  RA(1, 5);
            MFZ(0, 0, 6);  RA(1, 6);
            MFZ(0, 0, 7);  RA(1, 7);
            MFZ(0, 1, 0);
            MFZ(0, 1, 1);
            MFZ(0, 1, 2);
            MFZ(0, 1, 3);
            MFZ(0, 1, 4);
            MFZ(0, 1, 5);
            MFZ(0, 1, 6);
            MFZ(0, 1, 7);  WAIT_LGKM0(); BAR();
            MFZ(0, 2, 0);  DA(0);
            MFZ(0, 2, 1);
            MFZ(0, 2, 2);  RW(1, 0);
            MFZ(0, 2, 3);
            MFZ(0, 2, 4);  DA(1);
            MFZ(0, 2, 5);
            MFZ(0, 2, 6);  RW(1, 1);
            MFZ(0, 2, 7);
            MFZ(0, 3, 0);  DA(2);
            MFZ(0, 3, 1);
            MFZ(0, 3, 2);  RW(1, 2);
            MFZ(0, 3, 3);
            MFZ(0, 3, 4);  DA(3);
            MFZ(0, 3, 5);
            MFZ(0, 3, 6);  RW(1, 3);
            MFZ(0, 3, 7);
            MFZ(0, 4, 0);  DA(4);
            MFZ(0, 4, 1);
            MFZ(0, 4, 2);  RW(1, 4);
            MFZ(0, 4, 3);
            MFZ(0, 4, 4);  DA(5);
            MFZ(0, 4, 5);
            MFZ(0, 4, 6);  RW(1, 5);
            MFZ(0, 4, 7);
            MFZ(0, 5, 0);  DA(6);
            MFZ(0, 5, 1);
            MFZ(0, 5, 2);  RW(1, 6);
            MFZ(0, 5, 3);
            MFZ(0, 5, 4);  DA(7);
            MFZ(0, 5, 5);
            MFZ(0, 5, 6);  RW(1, 7);
            MFZ(0, 5, 7);
            MFZ(0, 6, 0);
            MFZ(0, 6, 1);
            MFZ(0, 6, 2);
            MFZ(0, 6, 3);
            MFZ(0, 6, 4);
            MFZ(0, 6, 5);
            MFZ(0, 6, 6);
            MFZ(0, 6, 7);  WAIT_LGKM0(); BAR();
            MFZ(0, 7, 0);  DW(0);
            MFZ(0, 7, 1);
            MFZ(0, 7, 2);
            MFZ(0, 7, 3);
            MFZ(0, 7, 4);  DW(1);
            MFZ(0, 7, 5);
            MFZ(0, 7, 6);
            MFZ(0, 7, 7);
            MF(1, 0, 0);  DW(2);
            MF(1, 0, 1);
            MF(1, 0, 2);
            MF(1, 0, 3);
            MF(1, 0, 4);  DW(3);
            MF(1, 0, 5);
            MF(1, 0, 6);
            MF(1, 0, 7);
            MF(1, 1, 0);  DW(4);
            MF(1, 1, 1);
            MF(1, 1, 2);
            MF(1, 1, 3);  WAIT_VM(13); BAR(); FLIP0();
            MF(1, 1, 4);  RA(0, 0);
            MF(1, 1, 5);
            MF(1, 1, 6);  DW(5);
            MF(1, 1, 7);  RA(0, 1);
            MF(1, 2, 0);
            MF(1, 2, 1);
            MF(1, 2, 2);  RA(0, 2);
            MF(1, 2, 3);
            MF(1, 2, 4);
            MF(1, 2, 5);  RA(0, 3);
            MF(1, 2, 6);
            MF(1, 2, 7);
            MF(1, 3, 0);  RA(0, 4);
            MF(1, 3, 1);
            MF(1, 3, 2);  DW(6);
            MF(1, 3, 3);  RA(0, 5);
            MF(1, 3, 4);
            MF(1, 3, 5);
            MF(1, 3, 6);  RA(0, 6);
            MF(1, 3, 7);
            MF(1, 4, 0);
            MF(1, 4, 1);  RA(0, 7);
            MF(1, 4, 2);
            MF(1, 4, 3);
            MF(1, 4, 4);  RW(0, 0);
            MF(1, 4, 5);
            MF(1, 4, 6);  DW(7);
            MF(1, 4, 7);  RW(0, 1);
            MF(1, 5, 0);
            MF(1, 5, 1);
            MF(1, 5, 2);  RW(0, 2);
            MF(1, 5, 3);
            MF(1, 5, 4);
            MF(1, 5, 5);  RW(0, 3);
            MF(1, 5, 6);
            MF(1, 5, 7);
            MF(1, 6, 0);  RW(0, 4);
            MF(1, 6, 1);
            MF(1, 6, 2);
            MF(1, 6, 3);  RW(0, 5);
            MF(1, 6, 4);
            MF(1, 6, 5);
            MF(1, 6, 6);  RW(0, 6);
            MF(1, 6, 7);
            MF(1, 7, 0);
            MF(1, 7, 1);  RW(0, 7);
            MF(1, 7, 2);
            MF(1, 7, 3);
            MF(1, 7, 4);
            MF(1, 7, 5);
            MF(1, 7, 6);
            MF(1, 7, 7);  WAIT_LGKM0(); FLIP1();
        } else if constexpr (V == 'B') {
            MF(0, 0, 0);  RA(1, 0);
            MF(0, 0, 1);  RA(1, 1);
            MF(0, 0, 2);  RA(1, 2);
            MF(0, 0, 3);  RA(1, 3);
            MF(0, 0, 4);  RA(1, 4);
            MF(0, 0, 5);  RA(1, 5);
            MF(0, 0, 6);  RA(1, 6);
            MF(0, 0, 7);  RA(1, 7);
            MF(0, 1, 0);
            MF(0, 1, 1);
            MF(0, 1, 2);
            MF(0, 1, 3);
            MF(0, 1, 4);
            MF(0, 1, 5);
            MF(0, 1, 6);
            MF(0, 1, 7);  WAIT_LGKM0(); BAR();
            MF(0, 2, 0);  DA(0);
            MF(0, 2, 1);
            MF(0, 2, 2);  RW(1, 0);
            MF(0, 2, 3);
            MF(0, 2, 4);  DA(1);
            MF(0, 2, 5);
            MF(0, 2, 6);  RW(1, 1);
            MF(0, 2, 7);
            MF(0, 3, 0);  DA(2);
            MF(0, 3, 1);
            MF(0, 3, 2);  RW(1, 2);
            MF(0, 3, 3);
            MF(0, 3, 4);  DA(3);
            MF(0, 3, 5);
            MF(0, 3, 6);  RW(1, 3);
            MF(0, 3, 7);
            MF(0, 4, 0);  DA(4);
            MF(0, 4, 1);
            MF(0, 4, 2);  RW(1, 4);
            MF(0, 4, 3);
            MF(0, 4, 4);  DA(5);
            MF(0, 4, 5);
            MF(0, 4, 6);  RW(1, 5);
            MF(0, 4, 7);
            MF(0, 5, 0);  DA(6);
            MF(0, 5, 1);
            MF(0, 5, 2);  RW(1, 6);
            MF(0, 5, 3);
            MF(0, 5, 4);  DA(7);
            MF(0, 5, 5);
            MF(0, 5, 6);  RW(1, 7);
            MF(0, 5, 7);
            MF(0, 6, 0);
            MF(0, 6, 1);
            MF(0, 6, 2);
            MF(0, 6, 3);
            MF(0, 6, 4);
            MF(0, 6, 5);
            MF(0, 6, 6);
            MF(0, 6, 7);  WAIT_LGKM0(); BAR();
            MF(0, 7, 0);  DW(0);
            MF(0, 7, 1);
            MF(0, 7, 2);
            MF(0, 7, 3);
            MF(0, 7, 4);  DW(1);
            MF(0, 7, 5);
            MF(0, 7, 6);
            MF(0, 7, 7);
            MF(1, 0, 0);  DW(2);
            MF(1, 0, 1);
            MF(1, 0, 2);
            MF(1, 0, 3);
            MF(1, 0, 4);  DW(3);
            MF(1, 0, 5);
            MF(1, 0, 6);
            MF(1, 0, 7);
            MF(1, 1, 0);  DW(4);
            MF(1, 1, 1);
            MF(1, 1, 2);
            MF(1, 1, 3);  WAIT_VM(13); BAR(); FLIP0();
            MF(1, 1, 4);  RA(0, 0);
            MF(1, 1, 5);
            MF(1, 1, 6);  DW(5);
            MF(1, 1, 7);  RA(0, 1);
            MF(1, 2, 0);
            MF(1, 2, 1);
            MF(1, 2, 2);  RA(0, 2);
            MF(1, 2, 3);
            MF(1, 2, 4);
            MF(1, 2, 5);  RA(0, 3);
            MF(1, 2, 6);
            MF(1, 2, 7);
            MF(1, 3, 0);  RA(0, 4);
            MF(1, 3, 1);
            MF(1, 3, 2);  DW(6);
            MF(1, 3, 3);  RA(0, 5);
            MF(1, 3, 4);
            MF(1, 3, 5);
            MF(1, 3, 6);  RA(0, 6);
            MF(1, 3, 7);
            MF(1, 4, 0);
            MF(1, 4, 1);  RA(0, 7);
            MF(1, 4, 2);
            MF(1, 4, 3);
            MF(1, 4, 4);  RW(0, 0);
            MF(1, 4, 5);
            MF(1, 4, 6);  DW(7);
            MF(1, 4, 7);  RW(0, 1);
            MF(1, 5, 0);
            MF(1, 5, 1);
            MF(1, 5, 2);  RW(0, 2);
            MF(1, 5, 3);
            MF(1, 5, 4);
            MF(1, 5, 5);  RW(0, 3);
            MF(1, 5, 6);
            MF(1, 5, 7);
            MF(1, 6, 0);  RW(0, 4);
            MF(1, 6, 1);
            MF(1, 6, 2);
            MF(1, 6, 3);  RW(0, 5);
            MF(1, 6, 4);
            MF(1, 6, 5);
            MF(1, 6, 6);  RW(0, 6);
            MF(1, 6, 7);
            MF(1, 7, 0);
            MF(1, 7, 1);  RW(0, 7);
            MF(1, 7, 2);
            MF(1, 7, 3);
            MF(1, 7, 4);
            MF(1, 7, 5);
            MF(1, 7, 6);
            MF(1, 7, 7);  WAIT_LGKM0(); FLIP1();
        } else if constexpr (V == 'C') {
            MF(0, 0, 0);  RA(1, 0);
            MF(0, 0, 1);  RA(1, 1);
            MF(0, 0, 2);  RA(1, 2);
            MF(0, 0, 3);  RA(1, 3);
            MF(0, 0, 4);  RA(1, 4);
            MF(0, 0, 5);  RA(1, 5);
            MF(0, 0, 6);  RA(1, 6);
            MF(0, 0, 7);  RA(1, 7);
            MF(0, 1, 0);
            MF(0, 1, 1);
            MF(0, 1, 2);
            MF(0, 1, 3);
            MF(0, 1, 4);
            MF(0, 1, 5);
            MF(0, 1, 6);
            MF(0, 1, 7);  WAIT_LGKM0();
            MF(0, 2, 0);
            MF(0, 2, 1);
            MF(0, 2, 2);  RW(1, 0);
            MF(0, 2, 3);
            MF(0, 2, 4);
            MF(0, 2, 5);
            MF(0, 2, 6);  RW(1, 1);
            MF(0, 2, 7);
            MF(0, 3, 0);
            MF(0, 3, 1);
            MF(0, 3, 2);  RW(1, 2);
            MF(0, 3, 3);
            MF(0, 3, 4);
            MF(0, 3, 5);
            MF(0, 3, 6);  RW(1, 3);
            MF(0, 3, 7);
            MF(0, 4, 0);
            MF(0, 4, 1);
            MF(0, 4, 2);  RW(1, 4);
            MF(0, 4, 3);
            MF(0, 4, 4);
            MF(0, 4, 5);
            MF(0, 4, 6);  RW(1, 5);
            MF(0, 4, 7);
            MF(0, 5, 0);
            MF(0, 5, 1);
            MF(0, 5, 2);  RW(1, 6);
            MF(0, 5, 3);
            MF(0, 5, 4);
            MF(0, 5, 5);
            MF(0, 5, 6);  RW(1, 7);
            MF(0, 5, 7);
            MF(0, 6, 0);
            MF(0, 6, 1);
            MF(0, 6, 2);
            MF(0, 6, 3);
            MF(0, 6, 4);
            MF(0, 6, 5);
            MF(0, 6, 6);
            MF(0, 6, 7);  WAIT_LGKM0();
            MF(0, 7, 0);
            MF(0, 7, 1);
            MF(0, 7, 2);
            MF(0, 7, 3);
            MF(0, 7, 4);
            MF(0, 7, 5);
            MF(0, 7, 6);
            MF(0, 7, 7);
            MF(1, 0, 0);
            MF(1, 0, 1);
            MF(1, 0, 2);
            MF(1, 0, 3);
            MF(1, 0, 4);
            MF(1, 0, 5);
            MF(1, 0, 6);
            MF(1, 0, 7);
            MF(1, 1, 0);
            MF(1, 1, 1);
            MF(1, 1, 2);
            MF(1, 1, 3);  WAIT_VM(0); BAR(); FLIP0();
            MF(1, 1, 4);  RA(0, 0);
            MF(1, 1, 5);  PA(0);
            MF(1, 1, 6);
            MF(1, 1, 7);  RA(0, 1);
            MF(1, 2, 0);  PA(1);
            MF(1, 2, 1);
            MF(1, 2, 2);  RA(0, 2);
            MF(1, 2, 3);  PA(2);
            MF(1, 2, 4);
            MF(1, 2, 5);  RA(0, 3);
            MF(1, 2, 6);  PA(3);
            MF(1, 2, 7);
            MF(1, 3, 0);  RA(0, 4);
            MF(1, 3, 1);  PA(4);
            MF(1, 3, 2);
            MF(1, 3, 3);  RA(0, 5);
            MF(1, 3, 4);  PA(5);
            MF(1, 3, 5);
            MF(1, 3, 6);  RA(0, 6);
            MF(1, 3, 7);  PA(6);
            MF(1, 4, 0);
            MF(1, 4, 1);  RA(0, 7);
            MF(1, 4, 2);  PA(7);
            MF(1, 4, 3);
            MF(1, 4, 4);  RW(0, 0);
            MF(1, 4, 5);  PW(0);
            MF(1, 4, 6);
            MF(1, 4, 7);  RW(0, 1);
            MF(1, 5, 0);  PW(1);
            MF(1, 5, 1);
            MF(1, 5, 2);  RW(0, 2);
            MF(1, 5, 3);  PW(2);
            MF(1, 5, 4);
            MF(1, 5, 5);  RW(0, 3);
            MF(1, 5, 6);  PW(3);
            MF(1, 5, 7);
            MF(1, 6, 0);  RW(0, 4);
            MF(1, 6, 1);  PW(4);
            MF(1, 6, 2);
            MF(1, 6, 3);  RW(0, 5);
            MF(1, 6, 4);  PW(5);
            MF(1, 6, 5);
            MF(1, 6, 6);  RW(0, 6);
            MF(1, 6, 7);  PW(6);
            MF(1, 7, 0);
            MF(1, 7, 1);  RW(0, 7);
            MF(1, 7, 2);  PW(7);
            MF(1, 7, 3);
            MF(1, 7, 4);
            MF(1, 7, 5);
            MF(1, 7, 6);
            MF(1, 7, 7);  WAIT_LGKM0(); FLIP1();
        } else if constexpr (V == 'D') {
            MF(0, 0, 0);  RA(1, 0);
            MF(0, 0, 1);  RA(1, 1);
            MF(0, 0, 2);  RA(1, 2);
            MF(0, 0, 3);  RA(1, 3);
            MF(0, 0, 4);  RA(1, 4);
            MF(0, 0, 5);  RA(1, 5);
            MF(0, 0, 6);  RA(1, 6);
            MF(0, 0, 7);  RA(1, 7);
            MF(0, 1, 0);
            MF(0, 1, 1);
            MF(0, 1, 2);
            MF(0, 1, 3);
            MF(0, 1, 4);
            MF(0, 1, 5);
            MF(0, 1, 6);
            MF(0, 1, 7);  WAIT_LGKM0();
            MF(0, 2, 0);
            MF(0, 2, 1);
            MF(0, 2, 2);  RW(1, 0);
            MF(0, 2, 3);
            MF(0, 2, 4);
            MF(0, 2, 5);
            MF(0, 2, 6);  RW(1, 1);
            MF(0, 2, 7);
            MF(0, 3, 0);
            MF(0, 3, 1);
            MF(0, 3, 2);  RW(1, 2);
            MF(0, 3, 3);
            MF(0, 3, 4);
            MF(0, 3, 5);
            MF(0, 3, 6);  RW(1, 3);
            MF(0, 3, 7);
            MF(0, 4, 0);
            MF(0, 4, 1);
            MF(0, 4, 2);  RW(1, 4);
            MF(0, 4, 3);
            MF(0, 4, 4);
            MF(0, 4, 5);
            MF(0, 4, 6);  RW(1, 5);
            MF(0, 4, 7);
            MF(0, 5, 0);
            MF(0, 5, 1);
            MF(0, 5, 2);  RW(1, 6);
            MF(0, 5, 3);
            MF(0, 5, 4);
            MF(0, 5, 5);
            MF(0, 5, 6);  RW(1, 7);
            MF(0, 5, 7);
            MF(0, 6, 0);
            MF(0, 6, 1);
            MF(0, 6, 2);
            MF(0, 6, 3);
            MF(0, 6, 4);
            MF(0, 6, 5);
            MF(0, 6, 6);
            MF(0, 6, 7);  WAIT_LGKM0(); BAR();
            MF(0, 7, 0);
            MF(0, 7, 1);  QA(0);
            MF(0, 7, 2);
            MF(0, 7, 3);
            MF(0, 7, 4);  QA(1);
            MF(0, 7, 5);
            MF(0, 7, 6);
            MF(0, 7, 7);  QA(2);
            MF(1, 0, 0);
            MF(1, 0, 1);
            MF(1, 0, 2);  QA(3);
            MF(1, 0, 3);
            MF(1, 0, 4);
            MF(1, 0, 5);  QA(4);
            MF(1, 0, 6);
            MF(1, 0, 7);
            MF(1, 1, 0);  QA(5);
            MF(1, 1, 1);
            MF(1, 1, 2);
            MF(1, 1, 3);  QA(6);
            MF(1, 1, 4);
            MF(1, 1, 5);
            MF(1, 1, 6);  QA(7);
            MF(1, 1, 7);
            MF(1, 2, 0);
            MF(1, 2, 1);  QW(0);
            MF(1, 2, 2);
            MF(1, 2, 3);
            MF(1, 2, 4);  QW(1);
            MF(1, 2, 5);
            MF(1, 2, 6);
            MF(1, 2, 7);  QW(2);
            MF(1, 3, 0);
            MF(1, 3, 1);
            MF(1, 3, 2);  QW(3);
            MF(1, 3, 3);
            MF(1, 3, 4);
            MF(1, 3, 5);  QW(4);
            MF(1, 3, 6);
            MF(1, 3, 7);
            MF(1, 4, 0);  QW(5);
            MF(1, 4, 1);
            MF(1, 4, 2);
            MF(1, 4, 3);  QW(6);
            MF(1, 4, 4);
            MF(1, 4, 5);
            MF(1, 4, 6);  QW(7);
            MF(1, 4, 7);
            MF(1, 5, 0);
            MF(1, 5, 1);
            MF(1, 5, 2);
            MF(1, 5, 3);
            MF(1, 5, 4);
            MF(1, 5, 5);
            MF(1, 5, 6);
            MF(1, 5, 7);
            MF(1, 6, 0);
            MF(1, 6, 1);
            MF(1, 6, 2);
            MF(1, 6, 3);
            MF(1, 6, 4);
            MF(1, 6, 5);
            MF(1, 6, 6);
            MF(1, 6, 7);
            MF(1, 7, 0);
            MF(1, 7, 1);
            MF(1, 7, 2);
            MF(1, 7, 3);
            MF(1, 7, 4);
            MF(1, 7, 5);
            MF(1, 7, 6);
            MF(1, 7, 7);  FLIP0(); FLIP1();
        }
            // GENERATED-END
#define KEEP8(F, S) asm volatile("" :: "v"(F[S][0]), "v"(F[S][1]), "v"(F[S][2]), "v"(F[S][3]), "v"(F[S][4]), \
                                      "v"(F[S][5]), "v"(F[S][6]), "v"(F[S][7]))
            KEEP8(fa, 0); KEEP8(fw, 0); KEEP8(fa, 1); KEEP8(fw, 1);
#undef KEEP8
#undef MF
#undef MFZ
#undef DA
#undef DW
#undef PA
#undef PW
#undef QA
#undef QW
#undef WAIT_LGKM0
#undef WAIT_VM
#undef BAR
#undef FLIP0
#undef FLIP1
        };
        ktile(0, IntTag<'A'>{});
        for (int t = 1; t + 2 < unk; ++t) ktile(t, IntTag<'B'>{});
        ktile(unk - 2, IntTag<'C'>{});
        ktile(unk - 1, IntTag<'D'>{});
#undef RA
#undef RW
        // K-tile 0 of the next output tile (16 pieces, requested during variant C) has landed once all but the 16 younger
        // pieces of its K-tile 1 have; the MFMAs are inline asm, so pad their last results before the epilogue reads them
        asm volatile("s_waitcnt vmcnt(16)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        TL_STAMP(1);

        if (SPLIT && cur.role == 2) {
            // ---- finisher of a split tile: the other K-ranges were started together with this one; wait for their slabs
            if (tid == 0) {
                // A counter word is (launch epoch << 8) | arrivals: only arrivals of THIS launch satisfy the wait, a writer
                // that turns up late from an earlier (timed-out) launch can neither satisfy nor disturb it.
                // Bounded wait (~1 s): a hand-off that never arrives must not hang the GPU; the finisher then counts the
                // event in the last counter word (bya_gemm_workspace_status reports it to the host; ops checks it) and
                // finishes the tile without the missing sums.
                int spins = 0;
                for (;;) {
                    const unsigned w = __hip_atomic_load(p.ws_counters + cur.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((w >> 8) == epoch && (w & 0xffu) >= (unsigned)(cur.parts - 1)) break;
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1 << 22)) {
                        __hip_atomic_fetch_add(p.ws_counters + 1023, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                if (split == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            // add the slabs into the accumulators (back into the AGPRs: the epilogue below is the ordinary one).  (row block,
            // slab) pairs in order, the loads of pair it + 1 in flight while pair it is summed: a slab read is an
            // agent-scope (sc1) load, a memory round trip past the L2s -- other workgroups, maybe on another XCD, wrote it.
            // (r5, measured and NOT adopted: more of the slab in flight.  Two 64-register buffers, two 32-register buffers,
            // a straight-line one-slab form: every variant made hipcc spill 0.3-1 KB per lane around the asm-owned
            // accumulators; the slab by LDS-DMA through the ring -- no registers at all -- built with 700 B of scratch, ran
            // 13-30 % SLOWER on every split shape and failed the split-K parity test.  profiles/history/r5_d_gemm_finisher_*.json)
            {
                const float* const part0 = p.ws_slabs + (size_t)cur.slab * (GEMM_WS_SLAB_BYTES / 4);
                const int nparts = cur.parts - 1, part_step = cur.slab_step;
                u32x4 pf[4][2];
                auto issue = [&](int it) {
                    const int jn = it / nparts, s2 = it - jn * nparts;
                    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(
                        (void*)(part0 + (size_t)s2 * part_step * (GEMM_WS_SLAB_BYTES / 4)), 0, (int)GEMM_WS_SLAB_BYTES, 0x00020000);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int hlf = 0; hlf < 2; ++hlf)
                            pf[e][hlf] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                rsP, (uint32_t)(((wave * 64 + (jn * 4 + e) * 2 + hlf) * 64 + lane) * 16), 0, 16 /* sc1 */));
                };
                issue(0);
#pragma unroll
                for (int jn = 0; jn < 8; ++jn) {
                    float ps[4][8];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < 8; ++i) ps[e][i] = 0.f;
                    for (int s2 = 0; s2 < nparts; ++s2) {
                        f32x4 c[4][2];
                        // whole-vector casts: bit_cast of ONE element of an ext_vector miscompiles (hipcc 7.2 returns element 0)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { c[e][0] = __builtin_bit_cast(f32x4, pf[e][0]); c[e][1] = __builtin_bit_cast(f32x4, pf[e][1]); }
                        const int it = jn * nparts + s2 + 1;
                        if (it < 8 * nparts) issue(it);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int i = 0; i < 8; ++i) ps[e][i] += c[e][i >> 2][i & 3];
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        f32x4 t = acc[i][jn];
                        t[0] += ps[0][i]; t[1] += ps[1][i]; t[2] += ps[2][i]; t[3] += ps[3][i];
                        asm volatile("" : "+a"(t));          // back into accumulator registers, 4 at a time
                        acc[i][jn] = t;
                    }
                }
            }
            auto run = [&](auto act_tag) {
                epilogue_wide<decltype(act_tag)::value, 2, SPLIT, CONV>(p, cur.z, cur.m0 + wm * 128, cur.n0 + wn * 128, fr, fq, acc, wave, lane);
            };
            dispatch_act_big(p.act, run);
            asm volatile("s_barrier" ::: "memory");          // every wave has read the slabs: arrivals back to 0 (same epoch:
                                                             // a hipGraph replays this launch with the same epoch)
            if (tid == 0) __hip_atomic_store(p.ws_counters + cur.ctr, epoch << 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            // ---- whole tile (ordinary epilogue), or (role 1) a split tile's partial sums -> slab, then drained and counted
            float* const raw_out = (SPLIT && cur.role == 1) ? p.ws_slabs + (size_t)cur.slab * (GEMM_WS_SLAB_BYTES / 4) : nullptr;
            if constexpr (QKN) {
                epilogue_qkn(p, cur.z, cur.m0 + wm * 128, cur.n0 + wn * 128, fr, fq, acc);
            } else {
                auto run = [&](auto act_tag) {
                    epilogue_wide<decltype(act_tag)::value, 2, SPLIT, CONV>(p, cur.z, cur.m0 + wm * 128, cur.n0 + wn * 128, fr, fq, acc,
                                                                            wave, lane, raw_out);
                };
                dispatch_act_big(p.act, run);
            }
            if (SPLIT && cur.role == 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
                if (tid == 0) {
                    if (split == 2) {                    // debugging aid: full agent-scope release instead of relying on sc1
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    // arrive: count within this launch's epoch.  A word of another epoch is taken over when it is IDLE
                    // (arrival count 0: every finisher leaves its word that way, and so does the zero-filled initial state)
                    // or older.  Idle words must be claimable whatever their epoch: a hipGraph replays its launches with
                    // the epochs baked in at capture, so the first split launch of a replay finds the words stamped by the
                    // LAST launch of the previous replay (or by an eager launch in between) -- a "newer" epoch.  Only a
                    // newer word with arrivals in it means this writer is the stale one (a timed-out launch) and must not touch it.
                    unsigned* const c = p.ws_counters + cur.ctr;
                    unsigned old = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (;;) {
                        const unsigned oe = old >> 8, age = (epoch - oe) & 0xffffffu;
                        unsigned want;
                        if (oe == epoch) want = old + 1u;
                        else if ((old & 0xffu) == 0u || age < 0x800000u) want = (epoch << 8) | 1u;
                        else break;
                        if (__hip_atomic_compare_exchange_strong(c, &old, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT)) break;
                    }
                }
            }
        }

        TL_STAMP(4);
        if (!nxt.valid) break;
        ++seq;
        cur = nxt;
        rsA = rsAn;
        rsW = rsWn;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the (empty-descriptor) prefetch pieces of the tile after the last
}

}  // namespace

// Implicit-GEMM convolution launch (bya_vae_conv3d, vae.hip): the CONV instance, no K-split.
int bya_launch_conv256p(const void* args, hipStream_t s) {
    const GemmArgs& a = *static_cast<const GemmArgs*>(args);
    const int tiles_m = (a.M + 255) / 256, tiles_n = (a.N + 255) / 256;
    const long long total = (long long)tiles_m * tiles_n;
    const int blocks = (int)(total < 256 ? (total + 7) / 8 * 8 : 256);
    const size_t lds = 2 * 512 * BK * 2;
    static std::atomic<unsigned long long> attr_done{0};
    if (bya_allow_big_lds(reinterpret_cast<const void*>(gemm256p_kernel<false, true>), (int)lds, attr_done) != BYA_OK) return BYA_ERR_LAUNCH;
    BYA_LAUNCH((gemm256p_kernel<false, true>), dim3(blocks), dim3(256), lds, s, a, tiles_m, tiles_n, 1, 0, 1 << 30, 0u);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

// q|k|v projection with the q/k-norm + RoPE epilogue (bya_gemm_qkv_norm_rope): the QKN instance, never split
int bya_launch_gemm256p_qkn(const void* args, int batch, hipStream_t s) {
    GemmArgs a = *static_cast<const GemmArgs*>(args);
    a.gm = gemm_group_m(a);
    const int tiles_m = (a.M + 255) / 256, tiles_n = (a.N + 255) / 256;
    const long long total = (long long)tiles_m * tiles_n * batch;
    const int blocks = (int)(total < 256 ? (total + 7) / 8 * 8 : 256);
    const size_t lds = 2 * 512 * BK * 2;
    static std::atomic<unsigned long long> attr_done{0};
    if (bya_allow_big_lds(reinterpret_cast<const void*>(gemm256p_kernel<false, false, true>), (int)lds, attr_done) != BYA_OK) return BYA_ERR_LAUNCH;
    BYA_LAUNCH((gemm256p_kernel<false, false, true>), dim3(blocks), dim3(256), lds, s, a, tiles_m, tiles_n, batch, 0, 1 << 30, 0u);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

int bya_gemm_split_min_ktiles() {
    const int o = bya_opt(BYA_OPT_GEMM_SPLITK_MIN);           // test / tuning option
    const int v = o ? o : DEFAULT_MIN_SPLIT_KTILES;
    return v < 3 ? 3 : v;
}

int bya_launch_gemm256p(const void* args, int batch, hipStream_t s) {
    GemmArgs a = *static_cast<const GemmArgs*>(args);
    a.gm = gemm_group_m(a);
    const int tiles_m = (a.M + 255) / 256, tiles_n = (a.N + 255) / 256;
    const long long total = (long long)tiles_m * tiles_n * batch;
    // split the last partial round along K when a workspace is registered (BYA_GEMM_SPLITK=0 switches it off, read per call)
    const int sk = bya_opt(BYA_OPT_GEMM_SPLITK);
    const int min_seg = bya_gemm_split_min_ktiles();
    const int split = (a.ws_slabs && a.ws_counters && sk != 0 && a.K / BK >= 2 * min_seg) ? (sk == 2 ? 2 : 1) : 0;
    int blocks = (int)(total < 256 && !split ? (total + 7) / 8 * 8 : 256);
    const size_t lds = 2 * 512 * BK * 2;
    static std::atomic<unsigned long long> attr_done{0}, attr_done_s{0};
    // would any XCD have tiles left over after its full rounds?  (only then the split instance is worth its epilogue branch)
    const bool leftover = split && (total % 256 != 0);
    if (leftover) {
        if (bya_allow_big_lds(reinterpret_cast<const void*>(gemm256p_kernel<true>), (int)lds, attr_done_s) != BYA_OK) return BYA_ERR_LAUNCH;
        // launch epoch, 24 bits, never 0 (0 = the zero-filled initial state of a counter word)
        static std::atomic<unsigned> g_epoch{0};
        unsigned epoch = (g_epoch.fetch_add(1) + 1u) & 0xffffffu;
        if (epoch == 0u) epoch = (g_epoch.fetch_add(1) + 1u) & 0xffffffu;
        BYA_LAUNCH(gemm256p_kernel<true>, dim3(blocks), dim3(256), lds, s, a, tiles_m, tiles_n, batch, split, min_seg, epoch);
    } else {
        if (bya_allow_big_lds(reinterpret_cast<const void*>(gemm256p_kernel<false>), (int)lds, attr_done) != BYA_OK) return BYA_ERR_LAUNCH;
        if (total < 256) blocks = (int)((total + 7) / 8 * 8);
        BYA_LAUNCH(gemm256p_kernel<false>, dim3(blocks), dim3(256), lds, s, a, tiles_m, tiles_n, batch, 0, min_seg, 0u);
    }
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}
