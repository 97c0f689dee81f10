// Persistent 256 x 256 x 128 e4m3 GEMM: ONE wave per SIMD, hand-placed K-loop -- the fp8 sibling of gemm_v4.hip.
//
// Same operands and epilogue as gemm_fp8_kernel.h (C = epi(sa[m] sw[n] sum_k A8[m,k] W8[n,k])), same skeleton as gemm_v4.hip:
// 4 waves, 128 x 128 per wave, 256 accumulator registers in AGPRs, a 2-stage 128 KiB LDS ring filled by LDS-DMA two K-tiles
// ahead, persistent workgroups walking XCD-contiguous ranges of the group-M tile order, the next output tile's first two
// K-tiles requested while the current one drains, 16-byte epilogue accesses through the permuted W staging.  A K-tile is
// 128 BYTES of a row, as there -- the LDS image, the swizzle, the DMA pieces and every address are the bf16 kernel's; a
// row holds 128 e4m3 values instead of 64 bf16.
//
// What differs is the matrix instruction: v_mfma_f32_16x16x128_f8f6f4 (e4m3 operands, 32 bytes of one row per lane, 32 cycles:
// twice the bf16 rate per clock) consumes a whole 128-byte K-tile in ONE instruction per 16 x 16 block -- 64 MFMAs per
// K-tile and wave, each reading a W fragment and an A fragment of 8 VGPRs.  (Its 32 bytes per lane are the bf16 kernel's two
// 16-byte fragments of k-steps 0 and 1, i.e. bytes 16 g .. and 64 + 16 g ..: not the instruction's natural k order, but A
// and W are permuted alike, so the products pair up.)  With every fragment needed for the whole K-tile there is no second
// k-step to hide the fragment reads behind; the order of the MFMAs does it instead:
//
//     phase 0:  W blocks 0..3  x  A blocks 0..7      phase 1:  W blocks 4..7  x  A blocks 0..7      (A-major inside a phase)
//
//   * W(0..3) retire at the end of phase 0 and are re-read for K-tile t + 1 at the start of phase 1 (1024 cycles ahead);
//   * A(j) retires after its four MFMAs of phase 1 and is re-read right there (needed a full phase later);
//   * W(4..7) retire with the last MFMAs of phase 1 and are re-read at the START of K-tile t + 1 (needed in its phase 1).
//   128 fragment VGPRs, one K-tile body, no double buffering.  Two barriers per K-tile: B1 behind the W(4..7) reads (the
//   last reads of stage t & 1, one pair behind each of the first MFMAs; B1 sits behind MFMA 9, when they have long returned:
//   the stage is free, the LDS-DMA of K-tile t + 2 goes there, one 1-KiB piece behind every third MFMA -- a vector-memory
//   instruction costs its wave ~17 cycles of matrix time, 16 of them in a row stalled all four waves), B2 at the phase boundary behind s_waitcnt vmcnt(16) (K-tile t + 1 has landed for every wave; only the 16
//   pieces just requested may still be in flight).
//
// K-tile variants: A first (accumulators start from the instruction's inline zero), B steady, C last-but-one (the DMA
// pieces fetch the NEXT output tile's K-tile 0), D last (next tile's K-tile 1; no fragment re-reads -- the next tile reads
// its first fragments behind its own barrier).  K >= 512.  No K-split of the last partial round (the bf16 kernel's slab
// exchange is not carried over).
//
// Inline-asm MFMAs are invisible to hipcc (gemm_v4.hip explains): every fragment is kept allocated to the end of the K-tile,
// the epilogue starts behind explicit s_nops, and this unit is compiled WITHOUT -amdgpu-mfma-vgpr-form (accumulators in AGPRs).
#include "gemm_common.h"

#ifndef BYA_F8_PLACE
#define BYA_F8_PLACE 9      // placement of the 16 LDS-DMA pieces inside a K-tile (tools/gen_gemm_fp8_schedule.py holds the tables)
#endif
#ifndef BYA_F8_ABLATE
#define BYA_F8_ABLATE 0     // timing-only ablations (tools/): 1 = no LDS-DMA in the K-loop, 2 = no K-loop barriers, 4 = no fragment re-reads
#endif

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int BK8 = 128;          // e4m3 elements (= bytes) per K-tile

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

template <int OFF>
__device__ __forceinline__ void ds_read16(i32x4& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}

struct Tile8 { int z, m0, n0; bool valid; };

// Wide epilogue of one wave (gemm_v4.hip's, plus the two scale vectors).  The lane (fr = lane & 15, fq = lane >> 4) holds,
// for row block j and accumulator register e, the EIGHT consecutive columns  n8 = n_wave + (4 e + fq) * 8 + i,  i = 0..7
// in acc[i][j][e], of row  m = m_wave + 16 j + fr.
template <int ACT, int JB>
__device__ __forceinline__ void epilogue_wide8(const GemmArgs& p, const float* __restrict__ sa, const float* __restrict__ sw,
                                               int z, int m_wave, int n_wave, int fr, int fq, const f32x4 (&acc)[8][8]) {
    const bool has_res = p.res != nullptr, has_gate = p.gate0 != nullptr, has_bias = p.bias != nullptr;
    const bool has_rs = p.bias_rowscale != nullptr;
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((has_res ? p.res : p.C) + (long long)z * p.res_bs), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.C + (long long)z * p.c_bs), 0, 0x7fffffff, 0x00020000);
    const char* g0base = reinterpret_cast<const char*>(p.gate0 + (long long)z * p.gate_bs);
    const char* g1base = reinterpret_cast<const char*>(p.gate1 + (long long)z * p.gate_bs);
    u32x4 bv[4], g0[4], g1[4];
    f32x4 wv[4][2];                                              // the eight channel scales of each column group
    uint32_t ncb[4], colb[4];
    bool nok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int n8 = n_wave + (4 * e + fq) * 8;
        nok[e] = n8 < p.N;                                       // N % 8 == 0 on this kernel's shapes (checked by the launcher)
        ncb[e] = nok[e] ? (uint32_t)n8 * 2u : 0u;
        colb[e] = (uint32_t)n8 * 2u;
        if (p.n_split > 0) colb[e] = ((uint32_t)(n8 / p.n_split) * (uint32_t)p.c_split_stride + (uint32_t)(n8 % p.n_split)) * 2u;
        bv[e] = has_bias ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.bias) + ncb[e]) : u32x4{0u, 0u, 0u, 0u};
        wv[e][0] = *reinterpret_cast<const f32x4*>(sw + (nok[e] ? n8 : 0));
        wv[e][1] = *reinterpret_cast<const f32x4*>(sw + (nok[e] ? n8 : 0) + 4);
        if (has_gate) {
            g0[e] = *reinterpret_cast<const u32x4*>(g0base + ncb[e]);
            g1[e] = *reinterpret_cast<const u32x4*>(g1base + ncb[e]);
        }
    }
#pragma unroll
    for (int jb = 0; jb < 8; jb += JB) {
        u32x4 rv[JB][4];
        float rs[JB], ra[JB];
        bool mok[JB];
        uint32_t roff[JB], coff[JB];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
            const int m = m_wave + 16 * (jb + jj) + fr;
            mok[jj] = m < p.M;
            const uint32_t mc = mok[jj] ? (uint32_t)m : 0u;
            rs[jj] = has_rs ? p.bias_rowscale[(long long)z * p.M + mc] : 1.0f;
            ra[jj] = sa[(long long)z * p.M + mc];
            roff[jj] = mc * (uint32_t)(p.ldres * 2);
            coff[jj] = mc * (uint32_t)(p.ldc * 2);
            if (has_res) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    rv[jj][e] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        rsR, (mok[jj] && nok[e]) ? roff[jj] + ncb[e] : 0xffffffffu, 0, 0));
            }
        }
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
            const int j = jb + jj;
            const int m = m_wave + 16 * j + fr;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float b8[8], v[8], a0[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) a0[i] = acc[i][j][e] * (ra[jj] * wv[e][i >> 2][i & 3]);      // row x channel scale
                unpack8(bv[e], b8);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = p.alpha * apply_act<ACT>(fmaf(rs[jj], b8[i], a0[i]), p.leaky);
                if (has_gate) {
                    float g8[8];
                    unpack8(m < p.gate_split ? g0[e] : g1[e], g8);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= g8[i];
                }
                if (has_res) {
                    float r8[8];
                    unpack8(rv[jj][e], r8);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] += r8[i];
                }
                __builtin_amdgcn_raw_buffer_store_b128(pack8(v), rsC, (mok[jj] && nok[e]) ? coff[jj] + colb[e] : 0xffffffffu, 0, 0);
            }
        }
    }
}

__global__ __launch_bounds__(256, 1) void gemm256p_fp8_kernel(GemmArgs p, const float* __restrict__ sa,
                                                             const float* __restrict__ sw, int tiles_m, int tiles_n, int batch,
                                                             int GM) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 256, STAGE = (BM + BN) * BK8, TILE_A = BM * BK8;
    static_assert(STAGE == 65536, "stage flip uses one address bit");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.K / BK8;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;

    // ---- this workgroup's output tiles: XCD x (= blockIdx % 8 under round-robin dispatch; speed only) owns a contiguous
    // range of the tile order, its workgroups take every (gridDim / 8)-th tile of it, round after round
    const int per_z = tiles_m * tiles_n, total = per_z * batch;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int cq = total >> 3, cr = total & 7;
    const int base = (xcd < cr) ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
    const int end = base + cq + (xcd < cr ? 1 : 0);
    auto coord = [&](int seq) {
        Tile8 c;
        const int id = base + slot + seq * slots;
        c.valid = id < end;
        const int idz = c.valid ? id : base;
        c.z = idz / per_z;
        const int idt = idz - c.z * per_z;
        const int per_group = GM * tiles_n;          // group-M order: GM row tiles sweep a column tile before moving on
        const int group = idt / per_group, first_m = group * GM;
        const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
        const int in_g = idt - group * per_group;
        c.m0 = (first_m + in_g % gsz) * BM;
        c.n0 = (in_g / gsz) * BN;
        return c;
    };
    int seq = 0;
    Tile8 cur = coord(seq);
    if (!cur.valid) return;

    // fragment read addresses (XOR swizzle on (row >> 1) & 7; row blocks are 16 rows = 2048 bytes apart): the 16-byte chunks
    // fq and 4 + fq of a row -- bytes 16 fq .. and 64 + 16 fq .. -- are the low and the high half of a lane's 32 operand bytes
    const int a_row = wm * 128 + fr, w_row = wn * 128 + fr;
    const int a_sw = (a_row >> 1) & 7, w_sw = (w_row >> 1) & 7;
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
    // (c*: the stage of the current K-tile, n*: the other one)
    uint32_t cAl = lds0 + a_row * 128 + ((fq ^ a_sw) << 4), cAh = lds0 + a_row * 128 + (((4 + fq) ^ a_sw) << 4);
    uint32_t cWl = lds0 + TILE_A + w_row * 128 + ((fq ^ w_sw) << 4);
    uint32_t cWh = lds0 + TILE_A + w_row * 128 + (((4 + fq) ^ w_sw) << 4);
    uint32_t fill = __builtin_amdgcn_readfirstlane(lds0 + wave * 64 * 128);     // this wave's first A piece, current stage

    // staging (gemm_v4.hip): wave w moves LDS slot rows [64w, 64w + 64) of the A tile and of the W tile, 8 one-KiB pieces each.
    // A slot rows are tile rows; W slot row s = 128 h + 16 i + r holds tile column 128 h + ((r & 3) * 4 + (r >> 2)) * 8 + i.
    uint32_t voA[8], voW[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int rl = wave * 64 + q * 8 + (lane >> 3);
        const int r = rl & 15, i = (rl >> 4) & 7;
        const int wcol = (rl & 128) + (((r & 3) << 2) | (r >> 2)) * 8 + i;
        const int chunk16 = ((lane & 7) ^ ((rl >> 1) & 7)) * 16;
        voA[q] = (uint32_t)rl * (uint32_t)p.lda + chunk16;
        voW[q] = (uint32_t)wcol * (uint32_t)p.ldw + chunk16;
    }
    const uint8_t* const A8 = reinterpret_cast<const uint8_t*>(p.A);
    const uint8_t* const W8 = reinterpret_cast<const uint8_t*>(p.W);
    auto a_rsrc = [&](const Tile8& c) {
        const long long left = (long long)(p.M - 1 - c.m0) * p.lda + p.K;
        return raw_rsrc(A8 + (long long)c.z * p.a_bs + (long long)c.m0 * p.lda, c.valid && left > 0 ? (uint32_t)left : 0u);
    };
    auto w_rsrc = [&](const Tile8& c) {
        const long long left = (long long)(p.N - 1 - c.n0) * p.ldw + p.K;
        return raw_rsrc(W8 + (long long)c.n0 * p.ldw, c.valid && left > 0 ? (uint32_t)left : 0u);
    };
    i32x4 rsA = a_rsrc(cur), rsW = w_rsrc(cur);

#define DMA_A(Q, BASE, VO, RS, SOFF) dma_piece<(Q) * 1024>(BASE, VO[Q], RS, SOFF)
#define DMA_W(Q, BASE, VO, RS, SOFF) dma_piece<TILE_A + (Q) * 1024>(BASE, VO[Q], RS, SOFF)
#define ALL8(M, ...) M(0, __VA_ARGS__); M(1, __VA_ARGS__); M(2, __VA_ARGS__); M(3, __VA_ARGS__); \
                     M(4, __VA_ARGS__); M(5, __VA_ARGS__); M(6, __VA_ARGS__); M(7, __VA_ARGS__)
    // ---- prologue of the FIRST tile only: K-tiles 0 and 1
    ALL8(DMA_A, fill, voA, rsA, 0u);
    ALL8(DMA_W, fill, voW, rsW, 0u);
    ALL8(DMA_A, fill ^ STAGE, voA, rsA, (uint32_t)BK8);
    ALL8(DMA_W, fill ^ STAGE, voW, rsW, (uint32_t)BK8);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");

    f32x4 acc[8][8];
    i32x4 al[8], ah[8], wl[8], wh[8];            // low / high 16 bytes of the A (row block j) and W (column block i) fragments

    for (;;) {
        // ---- K-tile 0 of this output tile has landed for this wave (prologue wait / the wait in front of the previous
        // epilogue); make that true for everybody, then fetch its A fragments and W(0..3)
        asm volatile("s_barrier" ::: "memory");
#define RAF(J, LO, HI) do { ds_read16<(J) * 2048>(al[J], LO); ds_read16<(J) * 2048>(ah[J], HI); } while (0)
#define RWF(I, LO, HI) do { ds_read16<(I) * 2048>(wl[I], LO); ds_read16<(I) * 2048>(wh[I], HI); } while (0)
        RAF(0, cAl, cAh); RAF(1, cAl, cAh); RAF(2, cAl, cAh); RAF(3, cAl, cAh);
        RAF(4, cAl, cAh); RAF(5, cAl, cAh); RAF(6, cAl, cAh); RAF(7, cAl, cAh);
        RWF(0, cWl, cWh); RWF(1, cWl, cWh); RWF(2, cWl, cWh); RWF(3, cWl, cWh);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        const Tile8 nxt = coord(seq + 1);
        const i32x4 rsAn = a_rsrc(nxt), rsWn = w_rsrc(nxt);

        // One K-tile, variant V (see the top); t = its index inside the output tile.
        auto ktile = [&](int t, auto v_c) {
            constexpr char V = decltype(v_c)::value;
            const uint32_t soff = (uint32_t)((t + 2) * BK8);
            const uint32_t nAl = cAl ^ STAGE, nAh = cAh ^ STAGE, nWl = cWl ^ STAGE, nWh = cWh ^ STAGE;
#define OPW(I) __builtin_shufflevector(wl[I], wh[I], 0, 1, 2, 3, 4, 5, 6, 7)
#define OPA(J) __builtin_shufflevector(al[J], ah[J], 0, 1, 2, 3, 4, 5, 6, 7)
#define MF8(I, J) do {                                                                                                   \
                if constexpr (V == 'A')                                                                                    \
                    asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, 0" : "=a"(acc[I][J]) : "v"(OPW(I)), "v"(OPA(J)));  \
                else                                                                                                       \
                    asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+a"(acc[I][J]) : "v"(OPW(I)), "v"(OPA(J))); \
            } while (0)
            // one LDS-DMA piece: K-tile t + 2 of this tile, or the next tile's first two
#define PIECE(Q, IS_W) do {                                                                                              \
                if constexpr ((BYA_F8_ABLATE & 1) != 0) {}                                                                 \
                else if constexpr (V == 'C') { if (IS_W) DMA_W(Q, fill, voW, rsWn, 0u); else DMA_A(Q, fill, voA, rsAn, 0u); }              \
                else if constexpr (V == 'D') { if (IS_W) DMA_W(Q, fill, voW, rsWn, (uint32_t)BK8); else DMA_A(Q, fill, voA, rsAn, (uint32_t)BK8); } \
                else { if (IS_W) DMA_W(Q, fill, voW, rsW, soff); else DMA_A(Q, fill, voA, rsA, soff); }                               \
            } while (0)
            // B1: this K-tile's stage is free (W(4..7), the last reads of it, have returned for every wave)
#define B1() do { if constexpr (BYA_F8_ABLATE & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    \
                  else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } while (0)
            // B2: K-tile t + 1 has landed for everybody (N = the pieces of K-tile t + 2 requested so far in this K-tile)
#define B2(N) do { if constexpr (V != 'D') { if constexpr (BYA_F8_ABLATE & 2) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); \
                   else asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_barrier" ::: "memory"); } } while (0)
#define REREAD_W() do { if constexpr (V != 'D' && !(BYA_F8_ABLATE & 4)) { RWF(0, nWl, nWh); RWF(1, nWl, nWh); RWF(2, nWl, nWh); RWF(3, nWl, nWh); } } while (0)
#define REREAD_A(J) do { if constexpr (V != 'D' && !(BYA_F8_ABLATE & 4)) RAF(J, nAl, nAh); } while (0)
            // GENERATED-BEGIN (tools/gen_gemm_fp8_schedule.py)
#if BYA_F8_PLACE == 0
            RWF(4, cWl, cWh); RWF(5, cWl, cWh); RWF(6, cWl, cWh); RWF(7, cWl, cWh);
            MF8(0, 0);
            MF8(1, 0);
            MF8(2, 0);
            MF8(3, 0); B1();
            MF8(0, 1); PIECE(0, false);
            MF8(1, 1); PIECE(1, false);
            MF8(2, 1); PIECE(2, false);
            MF8(3, 1); PIECE(3, false);
            MF8(0, 2); PIECE(4, false);
            MF8(1, 2); PIECE(5, false);
            MF8(2, 2); PIECE(6, false);
            MF8(3, 2); PIECE(7, false);
            MF8(0, 3); PIECE(0, true);
            MF8(1, 3); PIECE(1, true);
            MF8(2, 3); PIECE(2, true);
            MF8(3, 3); PIECE(3, true);
            MF8(0, 4); PIECE(4, true);
            MF8(1, 4); PIECE(5, true);
            MF8(2, 4); PIECE(6, true);
            MF8(3, 4); PIECE(7, true);
            MF8(0, 5);
            MF8(1, 5);
            MF8(2, 5);
            MF8(3, 5);
            MF8(0, 6);
            MF8(1, 6);
            MF8(2, 6);
            MF8(3, 6);
            MF8(0, 7);
            MF8(1, 7);
            MF8(2, 7); B2(16);
            MF8(3, 7); REREAD_W();
            MF8(4, 0);
            MF8(5, 0);
            MF8(6, 0);
            MF8(7, 0); REREAD_A(0);
            MF8(4, 1);
            MF8(5, 1);
            MF8(6, 1);
            MF8(7, 1); REREAD_A(1);
            MF8(4, 2);
            MF8(5, 2);
            MF8(6, 2);
            MF8(7, 2); REREAD_A(2);
            MF8(4, 3);
            MF8(5, 3);
            MF8(6, 3);
            MF8(7, 3); REREAD_A(3);
            MF8(4, 4);
            MF8(5, 4);
            MF8(6, 4);
            MF8(7, 4); REREAD_A(4);
            MF8(4, 5);
            MF8(5, 5);
            MF8(6, 5);
            MF8(7, 5); REREAD_A(5);
            MF8(4, 6);
            MF8(5, 6);
            MF8(6, 6);
            MF8(7, 6); REREAD_A(6);
            MF8(4, 7);
            MF8(5, 7);
            MF8(6, 7);
            MF8(7, 7); REREAD_A(7);
#elif BYA_F8_PLACE == 3
            RWF(4, cWl, cWh); RWF(5, cWl, cWh); RWF(6, cWl, cWh); RWF(7, cWl, cWh);
            MF8(0, 0);
            MF8(1, 0);
            MF8(2, 0);
            MF8(3, 0); B1(); PIECE(0, false);
            MF8(0, 1);
            MF8(1, 1);
            MF8(2, 1);
            MF8(3, 1); PIECE(1, false);
            MF8(0, 2);
            MF8(1, 2);
            MF8(2, 2);
            MF8(3, 2); PIECE(2, false);
            MF8(0, 3);
            MF8(1, 3);
            MF8(2, 3);
            MF8(3, 3); PIECE(3, false);
            MF8(0, 4);
            MF8(1, 4);
            MF8(2, 4);
            MF8(3, 4); PIECE(4, false);
            MF8(0, 5);
            MF8(1, 5);
            MF8(2, 5);
            MF8(3, 5); PIECE(5, false);
            MF8(0, 6);
            MF8(1, 6);
            MF8(2, 6);
            MF8(3, 6); PIECE(6, false);
            MF8(0, 7);
            MF8(1, 7);
            MF8(2, 7); B2(7);
            MF8(3, 7); PIECE(7, false); REREAD_W();
            MF8(4, 0);
            MF8(5, 0);
            MF8(6, 0);
            MF8(7, 0); PIECE(0, true); REREAD_A(0);
            MF8(4, 1);
            MF8(5, 1);
            MF8(6, 1);
            MF8(7, 1); PIECE(1, true); REREAD_A(1);
            MF8(4, 2);
            MF8(5, 2);
            MF8(6, 2);
            MF8(7, 2); PIECE(2, true); REREAD_A(2);
            MF8(4, 3);
            MF8(5, 3);
            MF8(6, 3);
            MF8(7, 3); PIECE(3, true); REREAD_A(3);
            MF8(4, 4);
            MF8(5, 4);
            MF8(6, 4);
            MF8(7, 4); PIECE(4, true); REREAD_A(4);
            MF8(4, 5);
            MF8(5, 5);
            MF8(6, 5);
            MF8(7, 5); PIECE(5, true); REREAD_A(5);
            MF8(4, 6);
            MF8(5, 6);
            MF8(6, 6);
            MF8(7, 6); PIECE(6, true); REREAD_A(6);
            MF8(4, 7);
            MF8(5, 7);
            MF8(6, 7);
            MF8(7, 7); PIECE(7, true); REREAD_A(7);
#elif BYA_F8_PLACE == 4
            RWF(4, cWl, cWh); RWF(5, cWl, cWh); RWF(6, cWl, cWh); RWF(7, cWl, cWh);
            MF8(0, 0);
            MF8(1, 0);
            MF8(2, 0);
            MF8(3, 0); B1();
            MF8(0, 1); PIECE(0, false);
            MF8(1, 1);
            MF8(2, 1);
            MF8(3, 1); PIECE(1, false);
            MF8(0, 2);
            MF8(1, 2);
            MF8(2, 2); PIECE(2, false);
            MF8(3, 2);
            MF8(0, 3);
            MF8(1, 3); PIECE(3, false);
            MF8(2, 3);
            MF8(3, 3);
            MF8(0, 4); PIECE(4, false);
            MF8(1, 4);
            MF8(2, 4);
            MF8(3, 4); PIECE(5, false);
            MF8(0, 5);
            MF8(1, 5);
            MF8(2, 5); PIECE(6, false);
            MF8(3, 5);
            MF8(0, 6);
            MF8(1, 6); PIECE(7, false);
            MF8(2, 6);
            MF8(3, 6);
            MF8(0, 7); PIECE(0, true);
            MF8(1, 7);
            MF8(2, 7); B2(9);
            MF8(3, 7); REREAD_W();
            MF8(4, 0);
            MF8(5, 0); PIECE(1, true);
            MF8(6, 0);
            MF8(7, 0); REREAD_A(0);
            MF8(4, 1); PIECE(2, true);
            MF8(5, 1);
            MF8(6, 1);
            MF8(7, 1); PIECE(3, true); REREAD_A(1);
            MF8(4, 2);
            MF8(5, 2);
            MF8(6, 2); PIECE(4, true);
            MF8(7, 2); REREAD_A(2);
            MF8(4, 3);
            MF8(5, 3); PIECE(5, true);
            MF8(6, 3);
            MF8(7, 3); REREAD_A(3);
            MF8(4, 4); PIECE(6, true);
            MF8(5, 4);
            MF8(6, 4);
            MF8(7, 4); PIECE(7, true); REREAD_A(4);
            MF8(4, 5);
            MF8(5, 5);
            MF8(6, 5);
            MF8(7, 5); REREAD_A(5);
            MF8(4, 6);
            MF8(5, 6);
            MF8(6, 6);
            MF8(7, 6); REREAD_A(6);
            MF8(4, 7);
            MF8(5, 7);
            MF8(6, 7);
            MF8(7, 7); REREAD_A(7);
#elif BYA_F8_PLACE == 7
            RWF(4, cWl, cWh); RWF(5, cWl, cWh); RWF(6, cWl, cWh); RWF(7, cWl, cWh);
            MF8(0, 0);
            MF8(1, 0);
            MF8(2, 0);
            MF8(3, 0);
            MF8(0, 1);
            MF8(1, 1); B1();
            MF8(2, 1); PIECE(0, false);
            MF8(3, 1);
            MF8(0, 2);
            MF8(1, 2); PIECE(1, false);
            MF8(2, 2);
            MF8(3, 2);
            MF8(0, 3); PIECE(2, false);
            MF8(1, 3);
            MF8(2, 3);
            MF8(3, 3); PIECE(3, false);
            MF8(0, 4);
            MF8(1, 4);
            MF8(2, 4); PIECE(4, false);
            MF8(3, 4);
            MF8(0, 5);
            MF8(1, 5); PIECE(5, false);
            MF8(2, 5);
            MF8(3, 5);
            MF8(0, 6); PIECE(6, false);
            MF8(1, 6);
            MF8(2, 6);
            MF8(3, 6); PIECE(7, false);
            MF8(0, 7);
            MF8(1, 7);
            MF8(2, 7); PIECE(0, true); B2(9);
            MF8(3, 7); REREAD_W();
            MF8(4, 0);
            MF8(5, 0); PIECE(1, true);
            MF8(6, 0);
            MF8(7, 0); REREAD_A(0);
            MF8(4, 1); PIECE(2, true);
            MF8(5, 1);
            MF8(6, 1);
            MF8(7, 1); PIECE(3, true); REREAD_A(1);
            MF8(4, 2);
            MF8(5, 2);
            MF8(6, 2); PIECE(4, true);
            MF8(7, 2); REREAD_A(2);
            MF8(4, 3);
            MF8(5, 3); PIECE(5, true);
            MF8(6, 3);
            MF8(7, 3); REREAD_A(3);
            MF8(4, 4); PIECE(6, true);
            MF8(5, 4);
            MF8(6, 4);
            MF8(7, 4); PIECE(7, true); REREAD_A(4);
            MF8(4, 5);
            MF8(5, 5);
            MF8(6, 5);
            MF8(7, 5); REREAD_A(5);
            MF8(4, 6);
            MF8(5, 6);
            MF8(6, 6);
            MF8(7, 6); REREAD_A(6);
            MF8(4, 7);
            MF8(5, 7);
            MF8(6, 7);
            MF8(7, 7); REREAD_A(7);
#elif BYA_F8_PLACE == 8
            RWF(4, cWl, cWh);
            MF8(0, 0); RWF(5, cWl, cWh);
            MF8(1, 0); RWF(6, cWl, cWh);
            MF8(2, 0); RWF(7, cWl, cWh);
            MF8(3, 0);
            MF8(0, 1);
            MF8(1, 1);
            MF8(2, 1);
            MF8(3, 1); B1();
            MF8(0, 2); PIECE(0, false);
            MF8(1, 2);
            MF8(2, 2);
            MF8(3, 2); PIECE(1, false);
            MF8(0, 3);
            MF8(1, 3);
            MF8(2, 3); PIECE(2, false);
            MF8(3, 3);
            MF8(0, 4);
            MF8(1, 4); PIECE(3, false);
            MF8(2, 4);
            MF8(3, 4);
            MF8(0, 5); PIECE(4, false);
            MF8(1, 5);
            MF8(2, 5);
            MF8(3, 5); PIECE(5, false);
            MF8(0, 6);
            MF8(1, 6);
            MF8(2, 6); PIECE(6, false);
            MF8(3, 6);
            MF8(0, 7);
            MF8(1, 7); PIECE(7, false);
            MF8(2, 7); B2(8);
            MF8(3, 7); REREAD_W();
            MF8(4, 0);
            MF8(5, 0); PIECE(0, true);
            MF8(6, 0);
            MF8(7, 0); REREAD_A(0);
            MF8(4, 1); PIECE(1, true);
            MF8(5, 1);
            MF8(6, 1);
            MF8(7, 1); PIECE(2, true); REREAD_A(1);
            MF8(4, 2);
            MF8(5, 2);
            MF8(6, 2); PIECE(3, true);
            MF8(7, 2); REREAD_A(2);
            MF8(4, 3);
            MF8(5, 3); PIECE(4, true);
            MF8(6, 3);
            MF8(7, 3); REREAD_A(3);
            MF8(4, 4); PIECE(5, true);
            MF8(5, 4);
            MF8(6, 4);
            MF8(7, 4); PIECE(6, true); REREAD_A(4);
            MF8(4, 5);
            MF8(5, 5);
            MF8(6, 5); PIECE(7, true);
            MF8(7, 5); REREAD_A(5);
            MF8(4, 6);
            MF8(5, 6);
            MF8(6, 6);
            MF8(7, 6); REREAD_A(6);
            MF8(4, 7);
            MF8(5, 7);
            MF8(6, 7);
            MF8(7, 7); REREAD_A(7);
#elif BYA_F8_PLACE == 9
            RWF(4, cWl, cWh);
            MF8(0, 0); RWF(5, cWl, cWh);
            MF8(1, 0); RWF(6, cWl, cWh);
            MF8(2, 0); RWF(7, cWl, cWh);
            MF8(3, 0);
            MF8(0, 1);
            MF8(1, 1);
            MF8(2, 1);
            MF8(3, 1);
            MF8(0, 2);
            MF8(1, 2); B1();
            MF8(2, 2); PIECE(0, false);
            MF8(3, 2);
            MF8(0, 3);
            MF8(1, 3); PIECE(1, false);
            MF8(2, 3);
            MF8(3, 3);
            MF8(0, 4); PIECE(2, false);
            MF8(1, 4);
            MF8(2, 4);
            MF8(3, 4); PIECE(3, false);
            MF8(0, 5);
            MF8(1, 5);
            MF8(2, 5); PIECE(4, false);
            MF8(3, 5);
            MF8(0, 6);
            MF8(1, 6); PIECE(5, false);
            MF8(2, 6);
            MF8(3, 6);
            MF8(0, 7); PIECE(6, false);
            MF8(1, 7);
            MF8(2, 7); B2(7);
            MF8(3, 7); REREAD_W();
            MF8(4, 0); PIECE(7, false);
            MF8(5, 0);
            MF8(6, 0);
            MF8(7, 0); PIECE(0, true); REREAD_A(0);
            MF8(4, 1);
            MF8(5, 1);
            MF8(6, 1); PIECE(1, true);
            MF8(7, 1); REREAD_A(1);
            MF8(4, 2);
            MF8(5, 2); PIECE(2, true);
            MF8(6, 2);
            MF8(7, 2); REREAD_A(2);
            MF8(4, 3); PIECE(3, true);
            MF8(5, 3);
            MF8(6, 3);
            MF8(7, 3); PIECE(4, true); REREAD_A(3);
            MF8(4, 4);
            MF8(5, 4);
            MF8(6, 4); PIECE(5, true);
            MF8(7, 4); REREAD_A(4);
            MF8(4, 5);
            MF8(5, 5); PIECE(6, true);
            MF8(6, 5);
            MF8(7, 5); REREAD_A(5);
            MF8(4, 6); PIECE(7, true);
            MF8(5, 6);
            MF8(6, 6);
            MF8(7, 6); REREAD_A(6);
            MF8(4, 7);
            MF8(5, 7);
            MF8(6, 7);
            MF8(7, 7); REREAD_A(7);
#elif BYA_F8_PLACE == 10
            RWF(4, cWl, cWh);
            MF8(0, 0); RWF(5, cWl, cWh);
            MF8(1, 0); RWF(6, cWl, cWh);
            MF8(2, 0); RWF(7, cWl, cWh);
            MF8(3, 0);
            MF8(0, 1);
            MF8(1, 1);
            MF8(2, 1);
            MF8(3, 1);
            MF8(0, 2);
            MF8(1, 2);
            MF8(2, 2);
            MF8(3, 2); B1();
            MF8(0, 3); PIECE(0, false);
            MF8(1, 3);
            MF8(2, 3);
            MF8(3, 3); PIECE(1, false);
            MF8(0, 4);
            MF8(1, 4);
            MF8(2, 4); PIECE(2, false);
            MF8(3, 4);
            MF8(0, 5);
            MF8(1, 5); PIECE(3, false);
            MF8(2, 5);
            MF8(3, 5);
            MF8(0, 6); PIECE(4, false);
            MF8(1, 6);
            MF8(2, 6);
            MF8(3, 6); PIECE(5, false);
            MF8(0, 7);
            MF8(1, 7);
            MF8(2, 7); PIECE(6, false); B2(7);
            MF8(3, 7); REREAD_W();
            MF8(4, 0); PIECE(7, false);
            MF8(5, 0);
            MF8(6, 0);
            MF8(7, 0); PIECE(0, true); REREAD_A(0);
            MF8(4, 1);
            MF8(5, 1);
            MF8(6, 1); PIECE(1, true);
            MF8(7, 1); REREAD_A(1);
            MF8(4, 2);
            MF8(5, 2); PIECE(2, true);
            MF8(6, 2);
            MF8(7, 2); REREAD_A(2);
            MF8(4, 3); PIECE(3, true);
            MF8(5, 3);
            MF8(6, 3);
            MF8(7, 3); PIECE(4, true); REREAD_A(3);
            MF8(4, 4);
            MF8(5, 4);
            MF8(6, 4); PIECE(5, true);
            MF8(7, 4); REREAD_A(4);
            MF8(4, 5);
            MF8(5, 5); PIECE(6, true);
            MF8(6, 5);
            MF8(7, 5); REREAD_A(5);
            MF8(4, 6); PIECE(7, true);
            MF8(5, 6);
            MF8(6, 6);
            MF8(7, 6); REREAD_A(6);
            MF8(4, 7);
            MF8(5, 7);
            MF8(6, 7);
            MF8(7, 7); REREAD_A(7);
#else
#error "BYA_F8_PLACE: unknown placement"
#endif
            // GENERATED-END
#undef B1
#undef B2
#undef REREAD_W
#undef REREAD_A
            // the next K-tile starts with A(0) and W(0..3): everything but A(7)'s two reads (LDS returns in order; they are
            // covered by that K-tile's wait in front of B1)
            if constexpr (V != 'D') asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            cAl ^= STAGE; cAh ^= STAGE; cWl ^= STAGE; cWh ^= STAGE; fill ^= STAGE;
#define KEEP8(F) asm volatile("" :: "v"(F[0]), "v"(F[1]), "v"(F[2]), "v"(F[3]), "v"(F[4]), "v"(F[5]), "v"(F[6]), "v"(F[7]))
            KEEP8(al); KEEP8(ah); KEEP8(wl); KEEP8(wh);
#undef KEEP8
#undef PIECE
#undef MF8
#undef OPA
#undef OPW
        };
        ktile(0, IntTag<'A'>{});
        for (int t = 1; t + 2 < nk; ++t) ktile(t, IntTag<'B'>{});
        ktile(nk - 2, IntTag<'C'>{});
        ktile(nk - 1, IntTag<'D'>{});
#undef RAF
#undef RWF
        // K-tile 0 of the next output tile (16 pieces, requested during variant C) has landed once all but the 16 younger
        // pieces of its K-tile 1 have; the MFMAs are inline asm, so pad their last results before the epilogue reads them
        asm volatile("s_waitcnt vmcnt(16)\n\ts_nop 15\n\ts_nop 15" ::: "memory");

        auto run = [&](auto act_tag) {
            epilogue_wide8<decltype(act_tag)::value, 2>(p, sa, sw, cur.z, cur.m0 + wm * 128, cur.n0 + wn * 128, fr, fq, acc);
        };
        dispatch_act_big(p.act, run);

        if (!nxt.valid) break;
        ++seq;
        cur = nxt;
        rsA = rsAn;
        rsW = rsWn;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the (empty-descriptor) prefetch pieces of the tile after the last
}

}  // namespace

// Is the persistent kernel applicable?  (16-byte epilogue accesses aligned, at least four K-tiles, 16-byte operand rows)
bool bya_gemm256p_fp8_eligible(const void* args) {
    const GemmArgs& a = *static_cast<const GemmArgs*>(args);
    return a.K % BK8 == 0 && a.K >= 4 * BK8 && a.N % 8 == 0 && a.n_split % 8 == 0 && a.ldc % 8 == 0 && (!a.res || a.ldres % 8 == 0) &&
        a.lda % 16 == 0 && a.ldw % 16 == 0 &&
        !(((uintptr_t)a.C | (uintptr_t)a.res | (uintptr_t)a.bias | (uintptr_t)a.gate0 | (uintptr_t)a.gate1) & 15) &&
        a.c_bs % 8 == 0 && a.res_bs % 8 == 0 && a.gate_bs % 8 == 0 && a.c_split_stride % 8 == 0 &&
        (long long)a.M * a.lda < (1LL << 32) && (long long)a.N * a.ldw < (1LL << 32);
}

int bya_launch_gemm256p_fp8(const void* args, const float* sa, const float* sw, int batch, int gm, hipStream_t s) {
    const GemmArgs& a = *static_cast<const GemmArgs*>(args);
    const int tiles_m = (a.M + 255) / 256, tiles_n = (a.N + 255) / 256;
    const long long total = (long long)tiles_m * tiles_n * batch;
    const int blocks = (int)(total < 256 ? (total + 7) / 8 * 8 : 256);
    const size_t lds = 2 * 512 * BK8;
    static std::atomic<unsigned long long> attr_done{0};
    if (bya_allow_big_lds(reinterpret_cast<const void*>(gemm256p_fp8_kernel), (int)lds, attr_done) != BYA_OK) return BYA_ERR_LAUNCH;
    BYA_LAUNCH(gemm256p_fp8_kernel, dim3(blocks), dim3(256), lds, s, a, sa, sw, tiles_m, tiles_n, batch, gm < 1 ? 1 : gm);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}
