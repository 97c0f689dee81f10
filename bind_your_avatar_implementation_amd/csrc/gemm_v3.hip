// 256x256x64 bf16 GEMM tile, ONE wave per SIMD, LDS-DMA staging running TWO K-tiles ahead inside a 2-stage (128 KiB) ring.
//
// Same operands, epilogue and ABI as gemm256_kernel (gemm.hip); what changes is the pipeline around the matrix core:
//
//  * 4 waves (2 x 2), wave tile 128 x 128: the 64 accumulator tiles (256 registers) live in AGPRs, the fragments of BOTH
//    32-wide k-steps of a K-tile (2 x (8 A + 8 W) x 4 = 128 registers) in the architectural VGPRs.  Every fragment read
//    from LDS feeds 8 MFMAs: 128 KiB of LDS reads per K-tile and block instead of the 192 KiB of the 8-wave kernel.
//  * The stage a K-tile occupies is released in HALVES: as soon as every wave holds the A fragments of both k-steps in
//    registers (barrier 1, after the first 16 of the tile's 128 MFMAs) the A half is re-filled by LDS-DMA with tile
//    t+2, and likewise the W half after barrier 2.  A DMA piece therefore has one and a half tile periods to land
//    (it is only waited for at barrier 3 of the NEXT tile) instead of the half period a "fill the other stage while this
//    one computes" ring gives it -- under full-chip load a fabric round trip is several thousand cycles, and with one
//    wave per SIMD nothing else hides it.
//  * Everything between the barriers is one hand-placed instruction stream (volatile inline asm keeps program order):
//    at most one side instruction -- ds_read_b128, LDS-DMA piece -- per MFMA gap, waits counted by hand
//    (vmcnt(13) = "all but this tile's own 13 pieces have landed").  tools/gen_gemm_v3_schedule.py holds the table
//    and rewrites the block between the GENERATED markers.
//
// This translation unit is compiled WITHOUT -amdgpu-mfma-vgpr-form so the accumulators may live in AGPRs.
#include "gemm_common.h"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 raw_rsrc(const void* base, uint32_t bytes) {
    const unsigned long long b = (unsigned long long)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffu));
    r.y = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

// one 1-KiB LDS-DMA piece: 64 lanes x 16 bytes from per-lane global offsets to LDS [m0 .. m0 + 1024)
template <int LDS_OFF>
__device__ __forceinline__ void dma_piece(uint32_t lds_base, uint32_t voff, const i32x4& rsrc, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_base + LDS_OFF), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

__global__ __launch_bounds__(256, 1) void gemm256v3_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 256, STAGE = (BM + BN) * BK * 2, TILE_A = BM * BK * 2;
    static_assert(STAGE == 65536, "stage flip uses one address bit");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    const int id = xcd_remap(blockIdx.x, nwg);
    constexpr int GM = 4;
    const int per_group = GM * tiles_n;
    const int group = id / per_group, first_m = group * GM;
    const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int in_g = id - group * per_group;
    const int tm = first_m + in_g % gsz, tn = in_g / gsz;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.z;
    const bf16_t* A = p.A + (long long)z * p.a_bs;
    const int nk = p.K / BK;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;

    // fragment read addresses (XOR swizzle on (row >> 1) & 7; row blocks are 16 rows = 2048 bytes apart)
    const int a_row = wm * 128 + fr, w_row = wn * 128 + fr;
    const int a_sw = (a_row >> 1) & 7, w_sw = (w_row >> 1) & 7;
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
    uint32_t cA0 = lds0 + a_row * 128 + ((fq ^ a_sw) << 4), cA1 = lds0 + a_row * 128 + (((4 + fq) ^ a_sw) << 4);
    uint32_t cW0 = lds0 + TILE_A + w_row * 128 + ((fq ^ w_sw) << 4);
    uint32_t cW1 = lds0 + TILE_A + w_row * 128 + (((4 + fq) ^ w_sw) << 4);

    // staging: wave w moves rows [64w, 64w + 64) of the A tile and of the W tile, 8 one-KiB pieces (8 rows) each; the lane
    // loads the source chunk that belongs at its linear LDS position (source-side XOR swizzle).  Rows past M / N fall
    // outside the descriptor and arrive as zeros.
    const i32x4 rsA = raw_rsrc(A, (uint32_t)(((long long)(p.M - 1) * p.lda + p.K) * 2));
    const i32x4 rsW = raw_rsrc(p.W, (uint32_t)(((long long)(p.N - 1) * p.ldw + p.K) * 2));
    uint32_t voA[8], voW[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int rl = wave * 64 + q * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((rl >> 1) & 7);
        voA[q] = (uint32_t)(m0 + rl) * (uint32_t)(p.lda * 2) + chunk * 16;
        voW[q] = (uint32_t)(n0 + rl) * (uint32_t)(p.ldw * 2) + chunk * 16;
    }
    // LDS byte address of this wave's first A piece in the stage being (re)filled
    uint32_t fill = __builtin_amdgcn_readfirstlane(lds0 + wave * 64 * 128);

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 fa[2][8], fw[2][8];

#define DA_T(Q, T) dma_piece<(Q) * 1024>(fill, voA[Q], rsA, (uint32_t)((T) * (BK * 2)))
#define DW_T(Q, T) dma_piece<TILE_A + (Q) * 1024>(fill, voW[Q], rsW, (uint32_t)((T) * (BK * 2)))
#define ALL8(M, T) M(0, T); M(1, T); M(2, T); M(3, T); M(4, T); M(5, T); M(6, T); M(7, T)
    // ---- prologue: tiles 0 and 1 completely
    ALL8(DA_T, 0);
    ALL8(DW_T, 0);
    if (nk > 1) {
        fill ^= STAGE;
        ALL8(DA_T, 1);
        ALL8(DW_T, 1);
        fill ^= STAGE;
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");
#define RA(S, J) ds_read128<(J) * 2048>(fa[S][J], (S) ? cA1 : cA0)
#define RW(S, I) ds_read128<(I) * 2048>(fw[S][I], (S) ? cW1 : cW0)
    RA(0, 0); RA(0, 1); RA(0, 2); RA(0, 3); RA(0, 4); RA(0, 5); RA(0, 6); RA(0, 7);
    RW(0, 0); RW(0, 1); RW(0, 2); RW(0, 3); RW(0, 4); RW(0, 5); RW(0, 6); RW(0, 7);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // One K-tile.  LD: tile t+2 exists (re-fill this stage), NX: tile t+1 exists (read its first fragments).
    auto tile = [&](int t, auto ld_c, auto nx_c) {
        constexpr bool LD = decltype(ld_c)::value != 0, NX = decltype(nx_c)::value != 0;
        const uint32_t soff = (uint32_t)((t + 2) * (BK * 2));
#define MF(S, I, J) \
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[I][J]) : "v"(fw[S][I]), "v"(fa[S][J]))
#define DA(Q) do { if constexpr (LD) dma_piece<(Q) * 1024>(fill, voA[Q], rsA, soff); } while (0)
#define DW(Q) do { if constexpr (LD) dma_piece<TILE_A + (Q) * 1024>(fill, voW[Q], rsW, soff); } while (0)
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define BAR1() do { if constexpr (LD) asm volatile("s_barrier" ::: "memory"); } while (0)
#define BAR2() do { if constexpr (LD) asm volatile("s_barrier" ::: "memory"); } while (0)
#define WAIT_VM_NEXT_TILE() do { if constexpr (NX) { if constexpr (LD) asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); \
                                                      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } } while (0)
#define BAR3() do { if constexpr (NX) asm volatile("s_barrier" ::: "memory"); } while (0)
#define FLIP0() do { cA0 ^= STAGE; cW0 ^= STAGE; } while (0)
#define FLIP1() do { cA1 ^= STAGE; cW1 ^= STAGE; fill ^= STAGE; } while (0)
#undef RA
#undef RW
#define RA(S, J) do { if constexpr ((S) == 1 || NX) ds_read128<(J) * 2048>(fa[S][J], (S) ? cA1 : cA0); } while (0)
#define RW(S, I) do { if constexpr ((S) == 1 || NX) ds_read128<(I) * 2048>(fw[S][I], (S) ? cW1 : cW0); } while (0)
        // GENERATED-BEGIN (tools/gen_gemm_v3_schedule.py)
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
        MF(0, 1, 7);  WAIT_LGKM0(); BAR1();
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
        MF(0, 6, 7);  WAIT_LGKM0(); BAR2();
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
        MF(1, 1, 3);  WAIT_VM_NEXT_TILE(); BAR3(); FLIP0();
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
        // GENERATED-END
        // Every fragment register stays allocated to its fragment for the whole tile: hipcc otherwise hands a k-step-0
        // register to a k-step-1 read as soon as ITS last MFMA has been issued (seen in the drain tile: RA(1, 7) was given
        // the register of fw[0][0] one instruction after MF(0, 0, 7)), and an LDS return may land before a queued MFMA
        // has read its operands -- nothing in an inline-asm MFMA tells the compiler or the hazard recogniser otherwise.
#define KEEP8(F, S) asm volatile("" :: "v"(F[S][0]), "v"(F[S][1]), "v"(F[S][2]), "v"(F[S][3]), "v"(F[S][4]), \
                                      "v"(F[S][5]), "v"(F[S][6]), "v"(F[S][7]))
        KEEP8(fa, 0); KEEP8(fw, 0); KEEP8(fa, 1); KEEP8(fw, 1);
#undef KEEP8
#undef MF
#undef DA
#undef DW
#undef WAIT_LGKM0
#undef BAR1
#undef BAR2
#undef WAIT_VM_NEXT_TILE
#undef BAR3
#undef FLIP0
#undef FLIP1
#undef RA
#undef RW
    };
    int t = 0;
    for (; t + 2 < nk; ++t) tile(t, IntTag<1>{}, IntTag<1>{});       // steady state: no branches inside a tile
    if (t + 1 < nk) { tile(t, IntTag<0>{}, IntTag<1>{}); ++t; }
    tile(t, IntTag<0>{}, IntTag<0>{});
    // the MFMAs are inline asm, invisible to hipcc's hazard recognizer: let the last results land in the AGPRs before
    // the epilogue's v_accvgpr_read
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");

    // ---- epilogue: acc[i][j] -> C[m][n4..n4+3], m = m0 + 128 wm + 16 j + fr, n4 = n0 + 128 wn + 16 i + 4 fq
    auto run = [&](auto act_tag) {
        epilogue_block<decltype(act_tag)::value, 8, 8, 4>(p, z, m0 + wm * 128 + fr, n0 + wn * 128 + fq * 4, acc);
    };
    dispatch_act_big(p.act, run);
}

}  // namespace

int bya_launch_gemm256v3(const void* args, int batch, hipStream_t s) {
    const GemmArgs& a = *static_cast<const GemmArgs*>(args);
    const int tiles_m = (a.M + 255) / 256, tiles_n = (a.N + 255) / 256;
    dim3 grid(tiles_m * tiles_n, 1, batch);
    const size_t lds = 2 * 512 * BK * 2;
    static std::atomic<unsigned long long> attr_done{0};
    if (bya_allow_big_lds(reinterpret_cast<const void*>(gemm256v3_kernel), (int)lds, attr_done) != BYA_OK) return BYA_ERR_LAUNCH;
    BYA_LAUNCH(gemm256v3_kernel, grid, dim3(256), lds, s, a);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}
