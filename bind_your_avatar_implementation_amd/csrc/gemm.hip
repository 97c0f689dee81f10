// bf16 "NT" GEMM for every Linear on the denoise-step path:  C[m,n] = epi( sum_k A[m,k] * W[n,k] )
// A = activations [M,K] row-major, W = torch Linear weight [N,K] row-major (both K-contiguous, which is
// exactly the MFMA operand order: one ds_read_b128 per fragment, no transposes anywhere).
//
// Replaces the torch.nn.Linear dispatches of the reference (rows D1,D3,D4,P1,R1,R2,A1,F1,L1,L2 of
// SURVEY.md section 8a; e.g. models/transformer.py:241-260 via diffusers Attention/FeedForward,
// models/router.py:253-254,275,381-383, models/audio_model.py:253-256).
//
// Structure (v1): 128x128x64 block tile, 4 waves (2x2) each 64x64 via 4x4 v_mfma_f32_16x16x32_bf16,
// global_load_lds_dwordx4 staging into a 2-deep LDS ring, XOR-swizzled on the SOURCE address so the
// lane-linear LDS image is bank-conflict-free for the ds_read_b128 fragment reads, XCD-aware + group-M
// block order for L2 reuse.  The MFMA is issued "swapped" (W rows as the A operand) so each lane ends
// up with 4 consecutive output columns of one output row -> 8-byte epilogue accesses.
// Epilogue: + bias -> activation -> * gate[row-type] -> + residual -> bf16.
#include "gemm_common.h"
#include <stdlib.h>
#include <string.h>
#include "options.h"

namespace {


template <int ROWS, int NWAVES>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ src, int ld, int row0, int row_max, int k0,
                                           char* lds_tile, int wave, int lane) {
    // ROWS x 64 bf16 tile, 8 rows (1 KiB) per wave-instruction, ROWS/(8*NWAVES) instructions per wave.
    constexpr int PER_WAVE = ROWS / NWAVES;
#pragma unroll
    for (int q = 0; q < PER_WAVE / 8; ++q) {
        const int rbase = wave * PER_WAVE + q * 8;
        const int rl = rbase + (lane >> 3);
        const int chunk = (lane & 7) ^ ((rl >> 1) & 7);
        int gr = row0 + rl;
        gr = gr < row_max ? gr : row_max;
        const bf16_t* g = src + (long long)gr * ld + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(g), LDS_PTR(lds_tile + rbase * 128), 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 lds_frag(const char* tile, int row, int chunk) {
    const int off = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    return *reinterpret_cast<const bf16x8*>(tile + off);
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NWAVES = WAVES_M * WAVES_N;
    constexpr int TILE_A = BM * BK * 2, TILE_W = BN * BK * 2, STAGE = TILE_A + TILE_W;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MI = WM / 16, NI = WN / 16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int id = xcd_remap(blockIdx.x, nwg);
    // group-M order: 8 row-tiles sweep one column-tile before moving on (A and W both reused from L2)
    constexpr int GM = 8;
    const int per_group = GM * tiles_n;
    const int group = id / per_group, first_m = group * GM;
    const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int in_g = id - group * per_group;
    const int tm = first_m + in_g % gsz, tn = in_g / gsz;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.z;

    const bf16_t* A = p.A + (long long)z * p.a_bs;
    const int nk = p.K / BK;

    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * STAGE;
        stage_tile<BM, NWAVES>(A, p.lda, m0, p.M - 1, kt * BK, base, wave, lane);
        stage_tile<BN, NWAVES>(p.W, p.ldw, n0, p.N - 1, kt * BK, base + TILE_A, wave, lane);
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int fr = lane & 15, fq = lane >> 4;

    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* ta = smem + (kt & 1) * STAGE;
        const char* tw = ta + TILE_A;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[MI], fw[NI];
#pragma unroll
            for (int j = 0; j < MI; ++j) fa[j] = lds_frag(ta, wm * WM + j * 16 + fr, ks * 4 + fq);
#pragma unroll
            for (int i = 0; i < NI; ++i) fw[i] = lds_frag(tw, wn * WN + i * 16 + fr, ks * 4 + fq);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: lane holds C[m][n4..n4+3], m = m0 + wm*WM + j*16 + fr, n4 = n0 + wn*WN + i*16 + fq*4
    // (one fully unrolled, switch-free copy per activation: a runtime-indexed accumulator array would go to scratch)
    auto run = [&](auto act_tag) {
        epilogue_block<decltype(act_tag)::value, NI, MI, (NI * MI > 16 ? 1 : NI)>(p, z, m0 + wm * WM + fr, n0 + wn * WN + fq * 4, acc);
    };
    dispatch_act(p.act, run);
}

// ================================================================================================================
// Pipelined 256x256x64 kernel (8 waves = 2(M) x 4(N), wave tile 128x64, 2-stage 128 KiB LDS ring).
//
// Per K-tile a wave runs eight 8-MFMA phases: for each 32-wide k-step the four 64x32 quadrants of its 128x64 tile
// in snake order (A0,B0) (A0,B1) (A1,B1) (A1,B0), so every fragment is read from LDS exactly once per tile and
// each phase needs at most ONE new operand, whose ds_reads are issued BEFORE the MFMAs of the previous phase:
// the matrix pipe never waits on LDS inside a tile and only 48 fragment registers are live.
// There is ONE raw s_barrier per K-tile, between the last two phases: by then every fragment the wave still
// needs from the current stage is in registers, so the barrier is both "tile t+1 has landed" (each wave drains
// its own LDS-DMA with vmcnt(0) first) and "stage t is free".  Right after it the wave reads the first fragments
// of tile t+1 and starts the LDS-DMA of tile t+2 into the stage it just stopped reading (2 of its 8
// global_load_lds per phase over the next four phases): no MFMA bubble at the tile seam, and every DMA gets
// most of a K-tile period to land.  No __syncthreads() in the loop: its implied vmcnt(0) would drain the DMA it is
// supposed to overlap.
// Tried and measured slower (8192^3: 1220 vs 1305 TFLOP/s): a strict ping-pong schedule (the two waves of a SIMD
// alternating between a 32-MFMA cluster and a fragment-read + LDS-DMA cluster, one s_barrier per cluster) -- the
// memory cluster is longer than the MFMA cluster, so serialising them loses what free-running waves overlap.
struct FragA { bf16x8 v[4]; };   // 64 rows x 32 k
struct FragB { bf16x8 v[2]; };   // 32 cols x 32 k

// Fragment reads are inline asm so that (a) hipcc's conservative "LDS-DMA in flight -> s_waitcnt vmcnt(0) before the
// next LDS read" never triggers and (b) the lgkmcnt waits can be COUNTED by hand: LDS ops complete in order, so
// "wait until all but the N youngest reads are back" leaves the prefetch of the next phase in flight.
// Every wait names the registers it makes valid as "+v" operands: the MFMAs that consume them then depend on the
// wait statement and cannot be scheduled above it (a bare asm s_waitcnt does not order register-only MFMAs).
template <int HALF>
__device__ __forceinline__ void load_frag_a(FragA& f, uint32_t addr) {
    ds_read128<(HALF * 64 + 0) * 128>(f.v[0], addr);
    ds_read128<(HALF * 64 + 16) * 128>(f.v[1], addr);
    ds_read128<(HALF * 64 + 32) * 128>(f.v[2], addr);
    ds_read128<(HALF * 64 + 48) * 128>(f.v[3], addr);
}
template <int HALF>
__device__ __forceinline__ void load_frag_b(FragB& f, uint32_t addr) {
    ds_read128<(HALF * 32 + 0) * 128>(f.v[0], addr);
    ds_read128<(HALF * 32 + 16) * 128>(f.v[1], addr);
}
template <int N>
__device__ __forceinline__ void wait_frags(FragA& a, FragB& b) {
    asm volatile("s_waitcnt lgkmcnt(%6)"
                 : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(a.v[3]), "+v"(b.v[0]), "+v"(b.v[1])
                 : "i"(N));
}

template <int HA, int HB>
__device__ __forceinline__ void mfma_quadrant(f32x4 (&acc)[4][8], const FragA& a, const FragB& b) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
            acc[HB * 2 + ni][HA * 4 + mi] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(b.v[ni], a.v[mi], acc[HB * 2 + ni][HA * 4 + mi], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
}

__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs p) {
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
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;

    // lane-constant parts of the fragment addresses (the XOR swizzle depends on (row >> 1) & 7 only, and every
    // tile/half offset is a multiple of 16 rows)
    const int a_row = wm * 128 + fr, w_row = wn * 64 + fr;
    const int a_sw = (a_row >> 1) & 7, w_sw = (w_row >> 1) & 7;
    const int a_off0 = ((fq ^ a_sw) << 4), a_off1 = (((4 + fq) ^ a_sw) << 4);
    const int w_off0 = ((fq ^ w_sw) << 4), w_off1 = (((4 + fq) ^ w_sw) << 4);
    const int a_lane = a_row * 128, w_lane = w_row * 128;

    // global -> LDS staging: 8 one-KiB pieces per wave per K-tile (q = 0..3: A rows wave*32 + 8q, q = 4..7: W rows)
    // through buffer_load ... lds: the per-lane byte offsets are loop-invariant (8 VGPRs), the K-tile advances in the
    // scalar offset, and rows past M / N fall outside the descriptor's range -> the hardware writes zeros (no clamps).
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)A, 0, (int)(((long long)(p.M - 1) * p.lda + p.K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.W, 0, (int)(((long long)(p.N - 1) * p.ldw + p.K) * 2), 0x00020000);
    uint32_t voff[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int rl = wave * 32 + (q & 3) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((rl >> 1) & 7);
        voff[q] = (q < 4) ? (uint32_t)(m0 + rl) * (uint32_t)(p.lda * 2) + chunk * 16
                          : (uint32_t)(n0 + rl) * (uint32_t)(p.ldw * 2) + chunk * 16;
    }
    auto piece = [&](int t, int q) {
        char* dst = smem + (t & 1) * STAGE + (q >= 4 ? TILE_A : 0) + (wave * 32 + (q & 3) * 8) * 128;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(q >= 4 ? rsW : rsA, LDS_PTR(dst), 16, voff[q], t * (BK * 2), 0, 0);
    };

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: tile 0 completely, tile 1's first two pieces
    piece(0, 0); piece(0, 1); piece(0, 2); piece(0, 3); piece(0, 4); piece(0, 5); piece(0, 6); piece(0, 7);
    if (nk > 1) { piece(1, 0); piece(1, 1); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // LDS byte addresses of this lane's fragment rows for the CURRENT stage, k-step 0 / 1 (tile and half offsets are
    // instruction immediates); stage 1 lives STAGE = 64 KiB above stage 0, so the seam just flips one address bit.
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
    uint32_t cA0 = lds0 + a_lane + a_off0, cA1 = lds0 + a_lane + a_off1;
    uint32_t cW0 = lds0 + TILE_A + w_lane + w_off0, cW1 = lds0 + TILE_A + w_lane + w_off1;
    static_assert(STAGE == 65536, "stage flip uses one address bit");
    FragA a0, a1;
    FragB bx, by;
    load_frag_a<0>(a0, cA0);
    load_frag_b<0>(bx, cW0);

    for (int t = 0; t < nk; ++t) {
        const bool more = t + 1 < nk;
        // ---- k-step 0
        load_frag_b<1>(by, cW0);                                // B1k0 (2 reads)
        if (more) { piece(t + 1, 2); piece(t + 1, 3); }
        __builtin_amdgcn_sched_barrier(0);
        wait_frags<2>(a0, bx);
        mfma_quadrant<0, 0>(acc, a0, bx);                       // P0 (A0k0, B0k0)
        __builtin_amdgcn_sched_barrier(0);
        load_frag_a<1>(a1, cA0);                                // A1k0 (4 reads)
        if (more) { piece(t + 1, 4); piece(t + 1, 5); }
        __builtin_amdgcn_sched_barrier(0);
        wait_frags<4>(a0, by);
        mfma_quadrant<0, 1>(acc, a0, by);                       // P1 (A0k0, B1k0)
        __builtin_amdgcn_sched_barrier(0);
        load_frag_a<0>(a0, cA1);                                // A0k1 (4 reads)
        if (more) { piece(t + 1, 6); piece(t + 1, 7); }
        __builtin_amdgcn_sched_barrier(0);
        wait_frags<4>(a1, by);
        mfma_quadrant<1, 1>(acc, a1, by);                       // P2 (A1k0, B1k0)
        __builtin_amdgcn_sched_barrier(0);
        load_frag_b<0>(by, cW1);                                // B0k1 (2 reads)
        __builtin_amdgcn_sched_barrier(0);
        mfma_quadrant<1, 0>(acc, a1, bx);                       // P3 (A1k0, B0k0): operands already waited for
        __builtin_amdgcn_sched_barrier(0);
        // ---- k-step 1
        load_frag_b<1>(bx, cW1);                                // B1k1 (2 reads)
        __builtin_amdgcn_sched_barrier(0);
        wait_frags<2>(a0, by);
        mfma_quadrant<0, 0>(acc, a0, by);                       // P4 (A0k1, B0k1)
        __builtin_amdgcn_sched_barrier(0);
        load_frag_a<1>(a1, cA1);                                // A1k1 (4 reads)
        __builtin_amdgcn_sched_barrier(0);
        wait_frags<4>(a0, bx);
        mfma_quadrant<0, 1>(acc, a0, bx);                       // P5 (A0k1, B1k1)
        __builtin_amdgcn_sched_barrier(0);
        wait_frags<0>(a1, bx);
        mfma_quadrant<1, 1>(acc, a1, bx);                       // P6 (A1k1, B1k1)
        __builtin_amdgcn_sched_barrier(0);
        // ---- seam: tile t+1 landed for everybody, stage t free for everybody (P7's operands are in registers)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cA0 ^= STAGE; cA1 ^= STAGE; cW0 ^= STAGE; cW1 ^= STAGE;
        if (more) {
            load_frag_a<0>(a0, cA0);                            // A0k0 of tile t+1 (4 reads)
            load_frag_b<0>(bx, cW0);                            // B0k0 of tile t+1 (2 reads)
            if (t + 2 < nk) { piece(t + 2, 0); piece(t + 2, 1); }
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_quadrant<1, 0>(acc, a1, by);                       // P7 (A1k1, B0k1): operands already waited for
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- epilogue (same lane map as the small-tile kernel): acc[i][j] -> C[m][n4..n4+3]
    auto run = [&](auto act_tag) {
        epilogue_block<decltype(act_tag)::value, 4, 8, 2>(p, z, m0 + wm * 128 + fr, n0 + wn * 64 + fq * 4, acc);
    };
    dispatch_act_big(p.act, run);
}

// the persistent kernel's 16-byte epilogue accesses must be aligned and K must have at least three K-tiles
inline bool v4_eligible(const GemmArgs& a) {
    return a.K >= 3 * BK && a.N % 8 == 0 && a.n_split % 8 == 0 && a.ldc % 8 == 0 && (!a.res || a.ldres % 8 == 0) &&
        !(((uintptr_t)a.C | (uintptr_t)a.res | (uintptr_t)a.bias | (uintptr_t)a.gate0 | (uintptr_t)a.gate1) & 15) &&
        a.c_bs % 8 == 0 && a.res_bs % 8 == 0 && a.gate_bs % 8 == 0 && a.c_split_stride % 8 == 0;
}

// ... and the 128 x 256 persistent kernel (gemm_v5.hip) needs four K-tiles for its three-stage ring to run across tiles
inline bool p128_eligible(const GemmArgs& a) { return v4_eligible(a) && a.K >= 4 * BK && a.conv_cpg_log2 < 0 && bya_opt(BYA_OPT_GEMM_VARIANT) == 0; }

int launch256(const GemmArgs& a, int batch, hipStream_t s) {
    const int tiles_m = (a.M + 255) / 256, tiles_n = (a.N + 255) / 256;
    dim3 grid(tiles_m * tiles_n, 1, batch);
    const size_t lds = 2 * 512 * BK * 2;
    static std::atomic<unsigned long long> attr_done{0};
    if (bya_allow_big_lds(reinterpret_cast<const void*>(gemm256_kernel), (int)lds, attr_done) != BYA_OK) return BYA_ERR_LAUNCH;
    // Kernel choice for the 256x256 tile shapes.  Default: the persistent one-wave-per-SIMD kernel (gemm_v4.hip) whenever
    // its 16-byte epilogue accesses are aligned and K has at least three K-tiles, else the 8-wave kernel below.
    // BYA_GEMM_VARIANT (read per call so one process can A/B them, tools/gemm_probe.py): "w8" = this file's 8-wave
    // kernel (the fallback), anything else = gemm_v4.hip.
    const bool v4_ok = v4_eligible(a);
    if (v4_ok && bya_opt(BYA_OPT_GEMM_VARIANT) != 1) return bya_launch_gemm256p(&a, batch, s);
    BYA_LAUNCH(gemm256_kernel, grid, dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch(const GemmArgs& a, int batch, hipStream_t s) {
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n, 1, batch);
    const size_t lds = 2 * (BM + BN) * BK * 2;
    static std::atomic<unsigned long long> attr_done{0};
    if (bya_allow_big_lds(reinterpret_cast<const void*>(gemm_bf16_kernel<BM, BN, WAVES_M, WAVES_N>), (int)lds, attr_done) != BYA_OK)
        return BYA_ERR_LAUNCH;
    BYA_LAUNCH((gemm_bf16_kernel<BM, BN, WAVES_M, WAVES_N>), grid, dim3(64 * WAVES_M * WAVES_N), lds, s, a);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

// 128 x 256 tiles or 256 x 256 (gemm_v4.hip)?  In units of one 256 x 256 tile's time on a CU: the rounds of the 256-row grid
// against rounds of half tiles at the 128-row kernel's relative speed per unit of tile area -- measured on full grids
// (tools/gemm_p128_probe.py): 0.85 with loader waves (gemm_v6.hip: every plain Linear), 0.80 without (gemm_v5.hip: the q|k|v
// projection with the norm epilogue, which does not fit 128 arch registers); the 128-row tile moves 1.5x the bytes per FLOP.
constexpr double REL_128S = 0.85, REL_128P = 0.80;
inline double rounds_128(int M, int N, int batch, double rel) {
    const long long t = (long long)((M + 127) / 128) * ((N + 255) / 256) * batch;
    return 0.5 * (double)((t + 255) / 256) / rel;
}
// ... the 256-row grid's cost when its last, partial round is cut along K (gemm_v4.hip): an XCD's R left-over tiles in
// p = min(32 / R, K-tiles / shortest range) K-ranges each, plus the slab exchange -- 0.3 of a tile's time measured where it is
// a large part of the launch (2222 x 3072 x 12288: 108 tiles in two ranges each run at 0.87 of an unsplit tile's time)
inline double rounds_256_split(int M, int N, int K, int batch) {
    const long long tiles = (long long)((M + 255) / 256) * ((N + 255) / 256) * batch, full = tiles / 256, rem = tiles % 256;
    if (rem == 0) return (double)full;
    const int R = (int)((rem + 7) / 8);
    int parts = 32 / R;
    const int by_k = (K / BK) / bya_gemm_split_min_ktiles();
    if (parts > by_k) parts = by_k;
    return (double)full + (parts > 1 ? 1.0 / parts + 0.3 : 1.0);
}
// Rows [0, m0) on 256 x 256 tiles and rows [m0, M) on 128 x 256 tiles as a second launch: m0 = the rows of the full rounds of
// 256-row tiles (2222 x 9216: seven row tiles = 252 tiles, then 430 rows = 144 half tiles; all-128-row would be three rounds of
// half tiles, all-256-row two whole rounds).  0 = no such split; its cost in *cost.
inline int hybrid_rows(int M, int N, int batch, double rel, double* cost) {
    const long long tn = (N + 255) / 256, tiles = (long long)((M + 255) / 256) * tn, full = tiles / 256;
    if (batch != 1 || full < 1 || tiles % 256 == 0) return 0;
    const int main_tm = (int)(full * 256 / tn), m0 = main_tm * 256;
    if (m0 <= 0 || m0 >= M) return 0;
    *cost = (double)(((long long)main_tm * tn + 255) / 256) + rounds_128(M - m0, N, 1, rel) + 0.05;        // (+ the second launch)
    return m0;
}
// The q|k|v projection with the norm epilogue (never cut along K).  0: 256-row tiles, 1: 128-row tiles, 2: both (rows split at *m0)
inline int plan_rows_qkn(int M, int N, int batch, int* m0) {
    const int forced = bya_opt(BYA_OPT_GEMM_TILE);
    if (forced >= 0) return forced == 5 || forced == 6 ? 1 : 0;
    const long long t256 = (long long)((M + 255) / 256) * ((N + 255) / 256) * batch;
    const double r256 = (double)((t256 + 255) / 256), r128 = rounds_128(M, N, batch, REL_128P);
    double best = r256, hc = 0.0;
    int plan = 0;
    if (r128 < 0.97 * best) { best = r128; plan = 1; }
    const int hm = hybrid_rows(M, N, batch, REL_128P, &hc);
    if (hm > 0 && hc < (plan == 0 ? r256 - 0.15 : 0.97 * best)) { *m0 = hm; plan = 2; }
    return plan;
}

// Tile choice: fewest "CU rounds" (wave quantisation on 256 CUs) weighted by the tile's relative efficiency.
inline int pick_tile(int M, int N, int K, int batch, int forced, int act, bool splitk, bool p128_ok) {
    if (forced >= 0) return (forced >= 4 && !act_on_big_tiles(act)) ? 1 : forced;
    if (N <= 64) return 0;
    auto rounds = [&](int bm, int bn, int per_cu) {
        const long long blocks = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * batch;
        const long long slots = 256LL * per_cu;
        return (double)((blocks + slots - 1) / slots) * (bm * bn) * (double)per_cu;   // rounds x work per CU per round
    };
    const double t128 = rounds(128, 128, 2) / 0.80;     // measured relative speeds of the three structures
    const double t256x128 = rounds(256, 128, 1) / 0.95;
    // with the split-K workspace the persistent kernel's last round costs its fill fraction (plus the slab exchange)
    const double frac_rounds = (double)((M + 255) / 256) * ((N + 255) / 256) * batch / 256.0;
    const double t256 = splitk ? (frac_rounds + 0.12) * 65536.0 : rounds(256, 256, 1) / 1.00;
    (void)t256x128;
    // short K loops and few rows: the pipelined kernel's prologue / epilogue dominate -- unless the grid is many rounds deep, where
    // the persistent kernel's cross-tile prefetch has no prologue to pay (the audio K/V projection of all 42 layers in one launch,
    // 832 x 258048 x 768 = 15.75 rounds: 390 us against 494 on the 128 x 128 kernel; r6)
#ifdef BYA_GEMM_NO_DEEP_GRID                 // (A/B build)
    const bool deep_grid = false;
#else
    const bool deep_grid = frac_rounds >= 4.0 && M >= 512 && K >= 512;
#endif
    if (((M < 1024 || K < 1024) && !deep_grid) || N < 512 || !act_on_big_tiles(act)) return 1;
    // the pipelined 256x256 kernel is ~1.2x the 128x128 one per unit of tile area when its grid fills the CUs
    if (p128_ok) {
        // 6: the 128 x 256 persistent kernel with loader waves (gemm_v6.hip), against what a cut last round really costs
        const double t256s = (splitk ? rounds_256_split(M, N, K, batch) * 65536.0 : t256) / 1.2;
        const double best = t256s <= t128 ? t256s : t128, r128 = rounds_128(M, N, batch, REL_128S);
        if (r128 * 65536.0 / 1.2 < 0.97 * best) {
            double hc = 0.0;          // (4: dispatch_gemm's row split sends the rows behind the full rounds to the 128-row tile)
            if (!splitk && hybrid_rows(M, N, batch, REL_128S, &hc) > 0 && hc < 0.97 * r128 && t256 / 1.2 <= t128) return 4;
            return 6;
        }
    }
    return (t256 / 1.2 <= t128) ? 4 : 1;
}

}  // namespace

namespace {
// split-K workspaces of the persistent kernel (gemm_v4.hip), caller-owned, one per DEVICE: a launch uses the workspace
// registered for the device that is current when it is enqueued (a slab or counter on another GPU would be a memory fault)
constexpr int MAX_DEVICES = 64;
std::atomic<void*> g_gemm_ws[MAX_DEVICES];
inline int current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_DEVICES ? dev : -1;
}
}  // namespace

extern "C" int bya_set_gemm_workspace(void* ws, int64_t bytes) {
    if (ws && (bytes < (int64_t)GEMM_WS_BYTES || ((uintptr_t)ws & 255))) return BYA_ERR_SHAPE;
    const int dev = current_device();
    if (dev < 0) return BYA_ERR_UNSUPPORTED;
    g_gemm_ws[dev].store(ws);
    return BYA_OK;
}

extern "C" int bya_gemm_workspace_status(int32_t* timeouts, hipStream_t stream) {
    if (!timeouts) return BYA_ERR_SHAPE;
    *timeouts = 0;
    const int dev = current_device();
    char* const ws = dev < 0 ? nullptr : static_cast<char*>(g_gemm_ws[dev].load());
    if (!ws) return BYA_OK;                       // no workspace, no split-K, nothing that could have timed out
    unsigned word = 0;
    if (hipMemcpyAsync(&word, ws + 1023 * 4, 4, hipMemcpyDeviceToHost, stream) != hipSuccess) return BYA_ERR_LAUNCH;
    if (hipStreamSynchronize(stream) != hipSuccess) return BYA_ERR_LAUNCH;
    *timeouts = (int32_t)word;
    return BYA_OK;
}

extern "C" int bya_gemm_workspace_bytes(int64_t* bytes) {
    if (!bytes) return BYA_ERR_SHAPE;
    *bytes = (int64_t)GEMM_WS_BYTES;
    return BYA_OK;
}

namespace {
int dispatch_gemm(const GemmArgs& a, int nbatch, hipStream_t stream);
}  // namespace

extern "C" int bya_gemm_bf16(const void* A, const void* W, const void* bias, void* C, const void* res,
                             const void* gate0, const void* gate1, const bya_gemm_desc* d, hipStream_t stream) {
    if (!A || !W || !C || !d) return BYA_ERR_SHAPE;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0) return BYA_ERR_SHAPE;
    if (d->K % BK != 0 || d->N % 4 != 0) return BYA_ERR_SHAPE;
    if (d->lda % 8 || d->ldw % 8 || d->ldc % 4 || (res && d->ldres % 4)) return BYA_ERR_ALIGN;
    if (((uintptr_t)A | (uintptr_t)W) & 15) return BYA_ERR_ALIGN;
    if (((uintptr_t)C | (uintptr_t)res | (uintptr_t)bias | (uintptr_t)gate0 | (uintptr_t)gate1) & 7) return BYA_ERR_ALIGN;
    if (d->act < 0 || d->act > 6) return BYA_ERR_UNSUPPORTED;
    GemmArgs a;
    a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.bias = (const bf16_t*)bias; a.C = (bf16_t*)C;
    a.res = (const bf16_t*)res; a.gate0 = (const bf16_t*)gate0; a.gate1 = (const bf16_t*)(gate1 ? gate1 : gate0);
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.lda = d->lda; a.ldw = d->ldw; a.ldc = d->ldc; a.ldres = d->ldres;
    a.a_bs = d->a_batch_stride; a.c_bs = d->c_batch_stride; a.res_bs = d->res_batch_stride;
    a.gate_bs = d->gate_batch_stride; a.gate_split = d->gate_split; a.act = d->act; a.leaky = 0.01f;
    a.n_split = d->n_split; a.c_split_stride = d->c_split_stride;
    a.bias_rowscale = d->bias_rowscale; a.alpha = d->alpha == 0.0f ? 1.0f : d->alpha;
    const int dev = current_device();
    char* const ws = dev < 0 ? nullptr : static_cast<char*>(g_gemm_ws[dev].load());
    a.ws_counters = reinterpret_cast<unsigned*>(ws);
    a.ws_slabs = ws ? reinterpret_cast<float*>(ws + GEMM_WS_COUNTER_BYTES) : nullptr;
    if (d->n_split < 0 || (d->n_split > 0 && (d->n_split % 4 || d->c_split_stride % 4 || res))) return BYA_ERR_SHAPE;
    return gemm_row_chunks(a, d->batch, 2, [&](const GemmArgs& piece, int batch, long long) {
        return dispatch_gemm(piece, batch, stream);
    });
}

// The packed q|k|v projection with the q/k LayerNorm(64) + RoPE (+ the k pre-scale) in its epilogue: one launch, q and k
// written once (include/bya.h).  Only the persistent one-wave-per-SIMD kernel has that epilogue: anything it does not take
// is BYA_ERR_UNSUPPORTED and the caller keeps bya_gemm_bf16 + bya_qknorm_rope.
extern "C" int bya_gemm_qkv_norm_rope(const void* A, const void* W, const void* bias, void* C, const bya_gemm_desc* d,
                                      const bya_qknorm_desc* n, hipStream_t stream) {
    if (!A || !W || !C || !d || !n) return BYA_ERR_SHAPE;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0 || d->K % BK != 0) return BYA_ERR_SHAPE;
    if (!n->qw || !n->qb || !n->kw || !n->kb || n->width <= 0 || n->text_rows < 0) return BYA_ERR_SHAPE;
    if (n->text_rows < d->M && (!n->cos || !n->sin)) return BYA_ERR_SHAPE;
    // N = 3 width: the packed q | k | v projection; N = 2 width (r6): q | k alone -- the sharded step computes v in a launch of
    // its own first and pushes it to the peers underneath this one
    if ((d->N != 3 * n->width && d->N != 2 * n->width) || n->width % 128 != 0 || d->n_split <= 0 || d->act != 0 || d->bias_rowscale || (d->alpha != 0.0f && d->alpha != 1.0f))
        return BYA_ERR_UNSUPPORTED;
    if (d->lda % 8 || d->ldw % 8) return BYA_ERR_ALIGN;
    if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)bias | (uintptr_t)n->qw | (uintptr_t)n->qb | (uintptr_t)n->kw |
         (uintptr_t)n->kb | (uintptr_t)n->cos | (uintptr_t)n->sin) & 15) return BYA_ERR_ALIGN;
    GemmArgs a;
    a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.bias = (const bf16_t*)bias; a.C = (bf16_t*)C;
    a.res = nullptr; a.gate0 = nullptr; a.gate1 = nullptr;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.lda = d->lda; a.ldw = d->ldw; a.ldc = d->ldc; a.ldres = 0;
    a.a_bs = d->a_batch_stride; a.c_bs = d->c_batch_stride; a.res_bs = 0; a.gate_bs = 0; a.gate_split = 0; a.act = 0; a.leaky = 0.01f;
    a.n_split = d->n_split; a.c_split_stride = d->c_split_stride;
    a.bias_rowscale = nullptr; a.alpha = 1.0f;
    a.ws_counters = nullptr; a.ws_slabs = nullptr;                 // (the QKN instance never splits a tile)
    a.qkn_w[0] = (const bf16_t*)n->qw; a.qkn_b[0] = (const bf16_t*)n->qb; a.qkn_w[1] = (const bf16_t*)n->kw; a.qkn_b[1] = (const bf16_t*)n->kb;
    a.qkn_cos = n->cos; a.qkn_sin = n->sin; a.qkn_text_rows = n->text_rows; a.qkn_width = n->width;
    a.qkn_eps = n->eps; a.qkn_kscale = n->k_scale == 0.0f ? 1.0f : n->k_scale;
    if (!v4_eligible(a) || !gemm_rows_reachable(a, a.M)) return BYA_ERR_UNSUPPORTED;
    int m0 = 0;
    const int plan = p128_eligible(a) ? plan_rows_qkn(a.M, a.N, d->batch, &m0) : 0;
    if (plan == 1) return bya_launch_gemm128p_qkn(&a, d->batch, stream);
    if (plan == 2) {                                             // rows [0, m0): 256-row tiles; the rest: 128-row tiles
        GemmArgs lo = a, hi = a;
        lo.M = m0;
        hi.M = a.M - m0;
        hi.A += (long long)m0 * a.lda;
        hi.C += (long long)m0 * a.ldc;
        // the rotary table's row of token m is m - text_rows: the second launch's token 0 is token m0
        hi.qkn_text_rows = a.qkn_text_rows > m0 ? a.qkn_text_rows - m0 : 0;
        if (a.qkn_cos && m0 > a.qkn_text_rows) {
            hi.qkn_cos = a.qkn_cos + (long long)(m0 - a.qkn_text_rows) * 64;
            hi.qkn_sin = a.qkn_sin + (long long)(m0 - a.qkn_text_rows) * 64;
        }
        const int rc = bya_launch_gemm256p_qkn(&lo, 1, stream);
        return rc != BYA_OK ? rc : bya_launch_gemm128p_qkn(&hi, 1, stream);
    }
    return bya_launch_gemm256p_qkn(&a, d->batch, stream);
}

namespace {
// ------------------------------------------------------------------------------------------------------------------
// Skinny Linears (bya_gemm_skinny_bf16): at most 64 rows (batch elements stacked when they fit) against a weight of up to 8192
// rows -- the step-invariant conditioning (32 face tokens x 2 identities
// into the perceiver / router keys, 12-49 audio windows through the projector whose conv1 weight alone is 2.4 GB, the
// LocalFacialExtractor's 37-row latents): ~80 launches per step that stream WEIGHTS.  On the tiled kernels such a launch has
// N / 128 workgroups (4 for the audio projector's first layer: 47 MB through four CUs, 0.58 ms) and a tile that is mostly
// padding.  Here: one workgroup per 16 output columns (and batch element), its 16 waves split K, every wave streams its
// slice of the 16 weight rows as MFMA A operands (v_mfma_f32_16x16x32_bf16; the <= 4 row tiles of X are the B operands, from
// L2) and leaves a partial [M x 16] in LDS; the partials are added in wave order (deterministic) and the epilogue (bias,
// activation, alpha, residual) runs on one element per thread.
template <int MT>
__global__ __launch_bounds__(1024) void gemm_skinny_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);               // [16 waves][MT * 16 rows][16 columns]
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // rows: the batch elements are stacked when they fit one workgroup together (fold > 0: row r = (z, m) = (r / M, r % M)),
    // so the weight is streamed once for all of them; otherwise blockIdx.y is the batch element
    const int n0 = blockIdx.x * 16, zb = blockIdx.y, fold = p.gate_split, rows = fold > 0 ? fold * p.M : p.M;
    const int ksteps = p.K / 32;
    const int k_lo = (int)((long long)ksteps * wave / 16), k_hi = (int)((long long)ksteps * (wave + 1) / 16);
    const bf16_t* wrow = p.W + (long long)(n0 + c) * p.ldw + 8 * g;
    const bf16_t* xrow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int r = mt * 16 + c;
        r = r < rows ? r : rows - 1;
        const int z = fold > 0 ? r / p.M : zb, m = fold > 0 ? r % p.M : r;
        xrow[mt] = p.A + (long long)z * p.a_bs + (long long)m * p.lda + 8 * g;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    int ks = k_lo;
    for (; ks + 4 <= k_hi; ks += 4) {                           // four weight loads in flight per lane
        bf16x8 wf[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) wf[u] = *reinterpret_cast<const bf16x8*>(wrow + (ks + u) * 32);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xrow[mt] + (ks + u) * 32);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], xf, acc[mt], 0, 0, 0);
            }
    }
    for (; ks < k_hi; ++ks) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wrow + ks * 32);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xrow[mt] + ks * 32);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc[mt], 0, 0, 0);
        }
    }
    // lane (c, g) holds row mt * 16 + c, columns n0 + 4 g .. + 3
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        *reinterpret_cast<f32x4*>(red + ((wave * MT * 16 + mt * 16 + c) * 16 + 4 * g)) = acc[mt];
    __syncthreads();
    for (int t = tid; t < MT * 256; t += 1024) {               // one output element per thread and pass
        const int r = t >> 4, n = t & 15;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += red[(w * MT * 16 + r) * 16 + n];
        if (r < rows) {
            const int z = fold > 0 ? r / p.M : zb, m = fold > 0 ? r % p.M : r;
            if (p.bias) v += bf2f(p.bias[n0 + n]);
            switch (p.act) {
                case 1: v = gelu_tanh(v); break;
                case 2: v = gelu_erf(v); break;
                case 3: v = v > 0.f ? v : 0.f; break;
                case 4: v = silu(v); break;
                case 5: v = v > 0.f ? v : v * p.leaky; break;
                case 6: v = gelu_tanh_ieee(v); break;
                default: break;
            }
            v *= p.alpha;
            if (p.res) v += bf2f(p.res[(long long)z * p.res_bs + (long long)m * p.ldres + n0 + n]);
            p.C[(long long)z * p.c_bs + (long long)m * p.ldc + n0 + n] = f2bf(v);
        }
    }
}

inline bool skinny_eligible(const GemmArgs& a) {
    // (wide outputs -- the audio projector's conv1, N = 24576 -- fill the chip on the tiled kernel and stream better there:
    // 0.75 ms against 0.83-1.2 ms here for its 2.4 GB weight)
    return a.M <= 64 && a.N <= 8192 && a.K >= 256 && a.K % 32 == 0 && a.N % 16 == 0 && a.lda % 8 == 0 && a.ldw % 8 == 0 && a.a_bs % 8 == 0 &&
        !a.gate0 && !a.bias_rowscale && a.n_split == 0 && a.conv_cpg_log2 < 0 && !(((uintptr_t)a.A | (uintptr_t)a.W) & 15);
}

template <int MT>
int launch_skinny_mt(const GemmArgs& a, dim3 grid, hipStream_t stream) {
    const size_t lds = (size_t)16 * MT * 16 * 16 * 4;
    static std::atomic<unsigned long long> big{0};
    if (lds > 64 * 1024 && bya_allow_big_lds(reinterpret_cast<const void*>(gemm_skinny_kernel<MT>), 160 * 1024, big) != BYA_OK)
        return BYA_ERR_LAUNCH;
    BYA_LAUNCH(gemm_skinny_kernel<MT>, grid, dim3(1024), lds, stream, a);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

int launch_skinny(const GemmArgs& a0, int nbatch, hipStream_t stream) {
    GemmArgs a = a0;
    // stacking pays while the row tiles stay <= 4 (more X fragments per weight fragment cost more than a second weight pass)
    const bool fold = nbatch > 1 && (long long)nbatch * a.M <= 64;           // (gate_split is free here: no gates on this path)
    a.gate_split = fold ? nbatch : 0;
    const int rows = fold ? nbatch * a.M : a.M, mt = (rows + 15) / 16;
    const dim3 grid((unsigned)(a.N / 16), (unsigned)(fold ? 1 : nbatch));
    switch (mt) {
        case 1: return launch_skinny_mt<1>(a, grid, stream);
        case 2: return launch_skinny_mt<2>(a, grid, stream);
        case 3: return launch_skinny_mt<3>(a, grid, stream);
        case 4: return launch_skinny_mt<4>(a, grid, stream);
        case 5: return launch_skinny_mt<5>(a, grid, stream);
        case 6: return launch_skinny_mt<6>(a, grid, stream);
        case 7: return launch_skinny_mt<7>(a, grid, stream);
        default: return launch_skinny_mt<8>(a, grid, stream);
    }
}

int dispatch_gemm(const GemmArgs& a, int nbatch, hipStream_t stream) {
    char* const ws = reinterpret_cast<char*>(a.ws_counters);
    struct { int M, N, K, batch, act; } dd{a.M, a.N, a.K, nbatch, a.act};
    const auto* d = &dd;
    const int forced = bya_opt(BYA_OPT_GEMM_TILE);              // tuning / test option
    const bool splitk = ws && bya_opt(BYA_OPT_GEMM_SPLITK) != 0 && a.K / BK >= 2 * bya_gemm_split_min_ktiles() && v4_eligible(a) &&
        bya_opt(BYA_OPT_GEMM_VARIANT) != 1;
    switch (pick_tile(d->M, d->N, d->K, d->batch, forced, d->act, splitk, p128_eligible(a))) {
        case 0: return launch<128, 64, 2, 2>(a, d->batch, stream);
        case 1: return launch<128, 128, 2, 2>(a, d->batch, stream);
        case 2: return launch<256, 128, 4, 2>(a, d->batch, stream);
        case 3: return launch<256, 256, 2, 4>(a, d->batch, stream);
        case 5: if (p128_eligible(a)) return bya_launch_gemm128p(&a, d->batch, stream); break;     // (forced only; else: the 256 x 256 path below)
        case 6: if (p128_eligible(a)) return bya_launch_gemm128s(&a, d->batch, stream); break;
        default: break;
    }
    // Pipelined 256x256 tiles, one workgroup per CU.  When the last round of tiles would leave most CUs idle
    // (N = 3072 at 17776 rows: 840 tiles = 3.28 rounds), the rows of the complete rounds go to the pipelined kernel
    // and the remaining rows to the 128x128 kernel, whose many small tiles fill all CUs at once.
    const int tn = (a.N + 255) / 256, tm = (a.M + 255) / 256;
    const long long tiles = (long long)tm * tn;
    const long long full = tiles / 256;
    // (with a split-K workspace the persistent kernel cuts that last round along K itself: no row split)
    if (!splitk && d->batch == 1 && full >= 1 && tiles % 256 != 0) {
        const int main_tm = (int)(full * 256 / tn);
        const int m0 = main_tm * 256;
        if (m0 > 0 && m0 < a.M) {
            const long long tail_blocks = (long long)((a.M - m0 + 127) / 128) * ((a.N + 127) / 128);
            // in pipelined-kernel rounds.  0.9 since round 5 (0.6 before): same-process sweeps (profiles/history/r5_d_gemm_sweep_*.json)
            // have the unsplit persistent kernel ahead of the row split wherever 0.6 chose it -- 2222 x 9216 x 3072: 1078 vs 960
            // TFLOP/s, 4444 x 9216: 1334 vs 1217, 17776 x 3072 x 3072: 1248 vs 1230 -- so the split now needs a clear win
            double tail_cost = 0.9 * (double)((tail_blocks + 511) / 512);
            // (r6) ... or to the 128 x 256 persistent kernel: one round of half tiles (17776 x 3072: the 1392 rows behind three
            // full rounds are 132 of them) plus the second launch
#ifdef BYA_GEMM_NO_P128_TAIL            // (the A/B build of tools/gemm_tail_probe.py)
            const bool tail_128p = false;
#else
            const bool tail_128p = p128_eligible(a) && act_on_big_tiles(a.act) && rounds_128(a.M - m0, a.N, 1, REL_128S) + 0.05 < tail_cost;
#endif
            if (tail_128p) tail_cost = rounds_128(a.M - m0, a.N, 1, REL_128S) + 0.05;
            const double main_cost = (double)(((long long)main_tm * tn + 255) / 256);
            if (main_cost + tail_cost < (double)((tiles + 255) / 256) - 0.15) {
                GemmArgs lo = a, hi = a;
                lo.M = m0;
                hi.M = a.M - m0;
                hi.A += (long long)m0 * a.lda;
                hi.C += (long long)m0 * a.ldc;
                if (hi.res) hi.res += (long long)m0 * a.ldres;
                if (hi.bias_rowscale) hi.bias_rowscale += m0;
                hi.gate_split = a.gate_split > m0 ? a.gate_split - m0 : 0;
                const int rc = launch256(lo, 1, stream);
                if (rc != BYA_OK) return rc;
                return tail_128p ? bya_launch_gemm128s(&hi, 1, stream) : launch<128, 128, 2, 2>(hi, 1, stream);
            }
        }
    }
    return launch256(a, d->batch, stream);
}
}  // namespace


// The weight-streaming kernel is the CALLER's choice: its summation order differs from the tiled kernels', and which of the
// two bya_gemm_bf16 would pick must not depend on how many rows a launch happens to have (a rank's shard of the token stream
// has to round exactly like the whole).  The engine asks for it where the row count is a property of the model -- the
// step-invariant conditioning.
extern "C" int bya_gemm_skinny_bf16(const void* A, const void* W, const void* bias, void* C, const void* res,
                                    const bya_gemm_desc* d, hipStream_t stream) {
    if (!A || !W || !C || !d) return BYA_ERR_SHAPE;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0) return BYA_ERR_SHAPE;
    if (d->act < 0 || d->act > 6) return BYA_ERR_UNSUPPORTED;
    GemmArgs a;
    a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.bias = (const bf16_t*)bias; a.C = (bf16_t*)C;
    a.res = (const bf16_t*)res; a.gate0 = nullptr; a.gate1 = nullptr;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.lda = d->lda; a.ldw = d->ldw; a.ldc = d->ldc; a.ldres = d->ldres;
    a.a_bs = d->a_batch_stride; a.c_bs = d->c_batch_stride; a.res_bs = d->res_batch_stride;
    a.gate_bs = 0; a.gate_split = 0; a.act = d->act; a.leaky = 0.01f;
    a.n_split = d->n_split; a.c_split_stride = d->c_split_stride;
    a.bias_rowscale = d->bias_rowscale; a.alpha = d->alpha == 0.0f ? 1.0f : d->alpha;
    a.ws_counters = nullptr; a.ws_slabs = nullptr;
    if (!skinny_eligible(a)) return BYA_ERR_UNSUPPORTED;
    return launch_skinny(a, d->batch, stream);
}
