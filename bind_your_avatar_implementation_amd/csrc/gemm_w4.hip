// 256x256x64 bf16 GEMM tile with ONE wave per SIMD: 4 waves (2 x 2), wave tile 128x128, the 64 accumulator tiles
// (256 registers) in AGPRs, fragments and addresses in the architectural VGPRs -- the whole 512-entry register file of
// the SIMD belongs to one wave.  Same LDS ring, source-side swizzle, LDS-DMA staging and fused epilogue as
// gemm256_kernel (gemm.hip); what changes is the register-level reuse (every fragment read from LDS feeds 8 MFMAs
// instead of 4-8: 128 KiB of LDS reads per K-tile and block instead of 192) and that no second wave competes for the
// SIMD's issue port or sits at the same barrier.  This translation unit is compiled WITHOUT -amdgpu-mfma-vgpr-form so
// the accumulators may live in AGPRs.
//
// Staging goes through REGISTERS, not LDS-DMA (the vendor's hand-written kernel of the same shape does the same): with
// one wave per SIMD nobody else can issue while an LDS-DMA instruction occupies the wave (~60+ cycles each, 16 per
// K-tile), whereas a buffer_load_dwordx4 + a ds_write_b128 cost ~20.  One set of 16 x 16 bytes per lane is enough for a
// full tile period of latency: piece q of tile t+1 is written to LDS and the same register is immediately re-requested
// for tile t+2.
//
// Per K-tile (two 32-wide k-steps) a wave runs four phases of 32 MFMAs (4 groups of 8); between the groups it issues
// the fragment reads of the next phase and 5-6 write + reload pairs.  All waits are counted by hand (every memory
// instruction here is inline asm): vmcnt(15) before a write leaves the 15 younger loads in flight; lgkmcnt(5|6) at a
// phase start leaves the previous phase's writes in flight.  One s_barrier per K-tile, after the first 8 MFMAs of the
// last phase.
#include "gemm_common.h"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

struct FragA8 { bf16x8 v[8]; };   // 128 rows of the A tile x 32 k  (MFMA B operand: one output row per lane)
struct FragW4 { bf16x8 v[4]; };   // 64 rows of the W tile x 32 k   (MFMA A operand: 4 output columns per lane)
struct Stage16 { u32x4 v[16]; };  // one K-tile's share of this wave in flight between HBM/L2 and LDS (16 x 1 KiB)

__device__ __forceinline__ void read_a8(FragA8& f, uint32_t addr) {
    ds_read128<0 * 2048>(f.v[0], addr); ds_read128<1 * 2048>(f.v[1], addr);
    ds_read128<2 * 2048>(f.v[2], addr); ds_read128<3 * 2048>(f.v[3], addr);
    ds_read128<4 * 2048>(f.v[4], addr); ds_read128<5 * 2048>(f.v[5], addr);
    ds_read128<6 * 2048>(f.v[6], addr); ds_read128<7 * 2048>(f.v[7], addr);
}
template <int HALF>
__device__ __forceinline__ void read_w4(FragW4& f, uint32_t addr) {
    ds_read128<(HALF * 4 + 0) * 2048>(f.v[0], addr); ds_read128<(HALF * 4 + 1) * 2048>(f.v[1], addr);
    ds_read128<(HALF * 4 + 2) * 2048>(f.v[2], addr); ds_read128<(HALF * 4 + 3) * 2048>(f.v[3], addr);
}
template <int N>
__device__ __forceinline__ void wait_aw(FragA8& a, FragW4& w) {
    asm volatile("s_waitcnt lgkmcnt(%12)"
                 : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(a.v[3]), "+v"(a.v[4]), "+v"(a.v[5]), "+v"(a.v[6]),
                   "+v"(a.v[7]), "+v"(w.v[0]), "+v"(w.v[1]), "+v"(w.v[2]), "+v"(w.v[3])
                 : "i"(N));
}
template <int N>
__device__ __forceinline__ void wait_w(FragW4& w) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(w.v[0]), "+v"(w.v[1]), "+v"(w.v[2]), "+v"(w.v[3]) : "i"(N));
}
// 8 MFMAs: W row block I of the half against the 8 row blocks of A.  Inline asm with the accumulators pinned to the
// AGPR class: hipcc otherwise parks part of the 256 accumulator registers in arch VGPRs and shuttles them through
// v_accvgpr_write around every MFMA.
template <int HALF, int I>
__device__ __forceinline__ void mfma8(f32x4 (&acc)[8][8], const FragA8& a, const FragW4& w) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[HALF * 4 + I][j]) : "v"(w.v[I]), "v"(a.v[j]));
}
// global -> registers (raw buffer load: rows past M / N fall outside the descriptor and read as zeros)
__device__ __forceinline__ void gload(u32x4& dst, uint32_t voff, const i32x4& rsrc, uint32_t soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff));
}
template <int OFF>
__device__ __forceinline__ void lwrite(uint32_t addr, const u32x4& data) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(data), "i"(OFF));
}
template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" : : "i"(N)); }

__device__ __forceinline__ i32x4 make_raw_rsrc(const void* base, uint32_t bytes) {
    const unsigned long long b = (unsigned long long)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffu));
    r.y = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

// LDS byte offset (inside a stage) of piece q of this wave: q = 0..7 A rows 64w + 8q, q = 8..15 W rows 64w + 8(q-8)
template <int Q>
constexpr int piece_off() { return (Q >= 8 ? 256 * BK * 2 : 0) + (Q & 7) * 8 * 128; }

template <int Q>
__device__ __forceinline__ void stage_write(uint32_t wr_addr, const Stage16& s) { lwrite<piece_off<Q>()>(wr_addr, s.v[Q]); }

__global__ __launch_bounds__(256, 1) void gemm256w4_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 256, STAGE = (BM + BN) * BK * 2, TILE_A = BM * BK * 2;
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

    // fragment addresses (the XOR swizzle depends on (row >> 1) & 7 only; row blocks are 16 rows = +2048 bytes)
    const int a_row = wm * 128 + fr, w_row = wn * 128 + fr;
    const int a_sw = (a_row >> 1) & 7, w_sw = (w_row >> 1) & 7;
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
    uint32_t cA0 = lds0 + a_row * 128 + ((fq ^ a_sw) << 4), cA1 = lds0 + a_row * 128 + (((4 + fq) ^ a_sw) << 4);
    uint32_t cW0 = lds0 + TILE_A + w_row * 128 + ((fq ^ w_sw) << 4);
    uint32_t cW1 = lds0 + TILE_A + w_row * 128 + (((4 + fq) ^ w_sw) << 4);
    static_assert(STAGE == 65536, "stage flip uses one address bit");

    // staging: wave w moves rows [64w, 64w+64) of the A tile (pieces 0..7) and of the W tile (pieces 8..15); the lane
    // loads the source chunk that belongs at its linear LDS position (source-side XOR swizzle)
    const i32x4 rsA = make_raw_rsrc(A, (uint32_t)(((long long)(p.M - 1) * p.lda + p.K) * 2));
    const i32x4 rsW = make_raw_rsrc(p.W, (uint32_t)(((long long)(p.N - 1) * p.ldw + p.K) * 2));
    const int rl0 = wave * 64 + (lane >> 3);
    uint32_t voA[2], voW[2];                   // ((rl0 + 8r) >> 1) & 7 alternates between two values with r
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int chunk = (lane & 7) ^ (((rl0 >> 1) + 4 * e) & 7);
        voA[e] = (uint32_t)(m0 + rl0) * (uint32_t)(p.lda * 2) + chunk * 16;
        voW[e] = (uint32_t)(n0 + rl0) * (uint32_t)(p.ldw * 2) + chunk * 16;
    }
    const uint32_t stepA = 8u * (uint32_t)(p.lda * 2), stepW = 8u * (uint32_t)(p.ldw * 2);
    uint32_t wr_addr = lds0 + STAGE + wave * 64 * 128 + lane * 16;      // tile t+1 goes to the OTHER stage

    auto load_piece = [&](Stage16& s, int t, int q) {       // q is a compile-time constant after unrolling
        const int r = q & 7;
        if (q >= 8) gload(s.v[q], voW[r & 1] + r * stepW, rsW, (uint32_t)(t * (BK * 2)));
        else gload(s.v[q], voA[r & 1] + r * stepA, rsA, (uint32_t)(t * (BK * 2)));
    };

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: tile 0 by LDS-DMA straight into stage 0 while tile 1 is requested into the staging registers
    Stage16 sr;
    {
        const __amdgpu_buffer_rsrc_t dA = __builtin_amdgcn_make_buffer_rsrc(
            (void*)A, 0, (int)(((long long)(p.M - 1) * p.lda + p.K) * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t dW = __builtin_amdgcn_make_buffer_rsrc(
            (void*)p.W, 0, (int)(((long long)(p.N - 1) * p.ldw + p.K) * 2), 0x00020000);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int r = q & 7;
            char* dst = smem + (q >= 8 ? TILE_A : 0) + (wave * 64 + r * 8) * 128;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(q >= 8 ? dW : dA, LDS_PTR(dst), 16,
                                                     q >= 8 ? voW[r & 1] + r * stepW : voA[r & 1] + r * stepA, 0, 0, 0);
        }
    }
    if (nk > 1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) load_piece(sr, 1, q);
        vm_wait<16>();                        // the 16 LDS-DMA pieces of tile 0 are older than the 16 register loads
    } else {
        vm_wait<0>();
    }
    __builtin_amdgcn_s_barrier();

    FragA8 fa0, fa1;
    FragW4 fw0, fw1;
    read_a8(fa0, cA0);
    read_w4<0>(fw0, cW0);

    // One K-tile; WR: tile t+1 exists (write its staged pieces to LDS), LD: tile t+2 exists (request it).
    auto tile = [&](int t, auto wr_c, auto ld_c) {
        constexpr bool wr = decltype(wr_c)::value != 0, ld = decltype(ld_c)::value != 0;
        // piece Q of tile t+1 leaves its register for LDS and the same register is re-requested for tile t+2 right away:
        // every piece has a whole tile period to arrive.  In flight before the write of piece Q: pieces Q+1..15 of tile
        // t+1 and pieces 0..Q-1 of tile t+2 = 15 loads (only the former when nothing more is requested).
#define MF2(H, I, J, FA, FW)                                                                                       \
        do {                                                                                                       \
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[(H) * 4 + (I)][(J)]) : "v"((FW).v[(I)]), "v"((FA).v[(J)]));         \
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[(H) * 4 + (I)][(J) + 1]) : "v"((FW).v[(I)]), "v"((FA).v[(J) + 1])); \
        } while (0)
#define XFER(Q)                                                          \
        do {                                                             \
            if constexpr (wr) {                                          \
                if constexpr (ld) vm_wait<15>(); else vm_wait<15 - (Q)>(); \
                stage_write<(Q)>(wr_addr, sr);                           \
            }                                                            \
            if constexpr (ld) load_piece(sr, t + 2, (Q));                \
        } while (0)
        // ---- P0: k-step 0, W rows 0..63 (operands read at the previous seam)
        wait_aw<0>(fa0, fw0);
        MF2(0, 0, 0, fa0, fw0);
        ds_read128<4 * 2048>(fw1.v[0], cW0);
        ds_read128<5 * 2048>(fw1.v[1], cW0);
        MF2(0, 0, 2, fa0, fw0);
        ds_read128<6 * 2048>(fw1.v[2], cW0);
        ds_read128<7 * 2048>(fw1.v[3], cW0);
        MF2(0, 0, 4, fa0, fw0);
        XFER(0);
        MF2(0, 0, 6, fa0, fw0);
        MF2(0, 1, 0, fa0, fw0);
        XFER(1);
        MF2(0, 1, 2, fa0, fw0);
        MF2(0, 1, 4, fa0, fw0);
        MF2(0, 1, 6, fa0, fw0);
        XFER(2);
        MF2(0, 2, 0, fa0, fw0);
        MF2(0, 2, 2, fa0, fw0);
        MF2(0, 2, 4, fa0, fw0);
        XFER(3);
        MF2(0, 2, 6, fa0, fw0);
        MF2(0, 3, 0, fa0, fw0);
        MF2(0, 3, 2, fa0, fw0);
        XFER(4);
        MF2(0, 3, 4, fa0, fw0);
        MF2(0, 3, 6, fa0, fw0);
        // ---- P1: k-step 0, W rows 64..127
        if constexpr (wr) wait_w<5>(fw1); else wait_w<0>(fw1);
        MF2(1, 0, 0, fa0, fw1);
        ds_read128<0 * 2048>(fa1.v[0], cA1);
        ds_read128<1 * 2048>(fa1.v[1], cA1);
        MF2(1, 0, 2, fa0, fw1);
        ds_read128<2 * 2048>(fa1.v[2], cA1);
        ds_read128<3 * 2048>(fa1.v[3], cA1);
        MF2(1, 0, 4, fa0, fw1);
        ds_read128<4 * 2048>(fa1.v[4], cA1);
        ds_read128<5 * 2048>(fa1.v[5], cA1);
        MF2(1, 0, 6, fa0, fw1);
        ds_read128<6 * 2048>(fa1.v[6], cA1);
        ds_read128<7 * 2048>(fa1.v[7], cA1);
        MF2(1, 1, 0, fa0, fw1);
        ds_read128<0 * 2048>(fw0.v[0], cW1);
        ds_read128<1 * 2048>(fw0.v[1], cW1);
        MF2(1, 1, 2, fa0, fw1);
        ds_read128<2 * 2048>(fw0.v[2], cW1);
        ds_read128<3 * 2048>(fw0.v[3], cW1);
        MF2(1, 1, 4, fa0, fw1);
        XFER(5);
        MF2(1, 1, 6, fa0, fw1);
        MF2(1, 2, 0, fa0, fw1);
        XFER(6);
        MF2(1, 2, 2, fa0, fw1);
        MF2(1, 2, 4, fa0, fw1);
        XFER(7);
        MF2(1, 2, 6, fa0, fw1);
        MF2(1, 3, 0, fa0, fw1);
        XFER(8);
        MF2(1, 3, 2, fa0, fw1);
        MF2(1, 3, 4, fa0, fw1);
        XFER(9);
        MF2(1, 3, 6, fa0, fw1);
        // ---- P2: k-step 1, W rows 0..63
        if constexpr (wr) wait_aw<5>(fa1, fw0); else wait_aw<0>(fa1, fw0);
        MF2(0, 0, 0, fa1, fw0);
        ds_read128<4 * 2048>(fw1.v[0], cW1);
        ds_read128<5 * 2048>(fw1.v[1], cW1);
        MF2(0, 0, 2, fa1, fw0);
        ds_read128<6 * 2048>(fw1.v[2], cW1);
        ds_read128<7 * 2048>(fw1.v[3], cW1);
        MF2(0, 0, 4, fa1, fw0);
        XFER(10);
        MF2(0, 0, 6, fa1, fw0);
        MF2(0, 1, 0, fa1, fw0);
        XFER(11);
        MF2(0, 1, 2, fa1, fw0);
        MF2(0, 1, 4, fa1, fw0);
        XFER(12);
        MF2(0, 1, 6, fa1, fw0);
        MF2(0, 2, 0, fa1, fw0);
        MF2(0, 2, 2, fa1, fw0);
        XFER(13);
        MF2(0, 2, 4, fa1, fw0);
        MF2(0, 2, 6, fa1, fw0);
        XFER(14);
        MF2(0, 3, 0, fa1, fw0);
        MF2(0, 3, 2, fa1, fw0);
        XFER(15);
        MF2(0, 3, 4, fa1, fw0);
        MF2(0, 3, 6, fa1, fw0);
        // ---- P3: k-step 1, W rows 64..127; the seam sits after its first 8 MFMAs
        if constexpr (wr) wait_w<6>(fw1); else wait_w<0>(fw1);
        MF2(1, 0, 0, fa1, fw1);
        MF2(1, 0, 2, fa1, fw1);
        MF2(1, 0, 4, fa1, fw1);
        MF2(1, 0, 6, fa1, fw1);
        // seam: tile t+1 is in LDS for everybody, stage t is free for everybody
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cA0 ^= STAGE; cA1 ^= STAGE; cW0 ^= STAGE; cW1 ^= STAGE; wr_addr ^= STAGE;
        MF2(1, 1, 0, fa1, fw1);
        if constexpr (wr) ds_read128<0 * 2048>(fa0.v[0], cA0);
        if constexpr (wr) ds_read128<1 * 2048>(fa0.v[1], cA0);
        MF2(1, 1, 2, fa1, fw1);
        if constexpr (wr) ds_read128<2 * 2048>(fa0.v[2], cA0);
        if constexpr (wr) ds_read128<3 * 2048>(fa0.v[3], cA0);
        MF2(1, 1, 4, fa1, fw1);
        if constexpr (wr) ds_read128<4 * 2048>(fa0.v[4], cA0);
        if constexpr (wr) ds_read128<5 * 2048>(fa0.v[5], cA0);
        MF2(1, 1, 6, fa1, fw1);
        if constexpr (wr) ds_read128<6 * 2048>(fa0.v[6], cA0);
        if constexpr (wr) ds_read128<7 * 2048>(fa0.v[7], cA0);
        MF2(1, 2, 0, fa1, fw1);
        if constexpr (wr) ds_read128<0 * 2048>(fw0.v[0], cW0);
        if constexpr (wr) ds_read128<1 * 2048>(fw0.v[1], cW0);
        MF2(1, 2, 2, fa1, fw1);
        if constexpr (wr) ds_read128<2 * 2048>(fw0.v[2], cW0);
        if constexpr (wr) ds_read128<3 * 2048>(fw0.v[3], cW0);
        MF2(1, 2, 4, fa1, fw1);
        MF2(1, 2, 6, fa1, fw1);
        MF2(1, 3, 0, fa1, fw1);
        MF2(1, 3, 2, fa1, fw1);
        MF2(1, 3, 4, fa1, fw1);
        MF2(1, 3, 6, fa1, fw1);
#undef XFER
#undef MF2
    };
    int t = 0;
    for (; t + 2 < nk; ++t) tile(t, IntTag<1>{}, IntTag<1>{});       // steady state: no branches inside a tile
    if (t + 1 < nk) { tile(t, IntTag<1>{}, IntTag<0>{}); ++t; }
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

int bya_launch_gemm256w4(const void* args, int batch, hipStream_t s) {
    const GemmArgs& a = *static_cast<const GemmArgs*>(args);
    const int tiles_m = (a.M + 255) / 256, tiles_n = (a.N + 255) / 256;
    dim3 grid(tiles_m * tiles_n, 1, batch);
    const size_t lds = 2 * 512 * BK * 2;
    static std::atomic<unsigned long long> attr_done{0};
    if (bya_allow_big_lds(reinterpret_cast<const void*>(gemm256w4_kernel), (int)lds, attr_done) != BYA_OK) return BYA_ERR_LAUNCH;
    BYA_LAUNCH(gemm256w4_kernel, grid, dim3(256), lds, s, a);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}
