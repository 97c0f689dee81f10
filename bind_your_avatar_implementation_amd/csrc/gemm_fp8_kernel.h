// Kernel template of the e4m3 GEMM (see gemm_fp8.hip for the description; two waves per SIMD, MFMA results in arch VGPRs).
#pragma once
#include "gemm_common.h"
#include <stdlib.h>

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int BK8 = 128;          // fp8 elements (= bytes) per K-tile
constexpr int FP8_GROUP_M = 8;    // row-tiles per group of the tile order (2 / 4 / 8 measured level: tools/fp8_gemm_zeros_probe.py)
constexpr float FP8_MAX = 448.0f; // largest finite e4m3fn

template <int ROWS, int NWAVES>
__device__ __forceinline__ void stage_tile8(const uint8_t* __restrict__ src, int ld, int row0, int row_max, int k0,
                                            char* lds_tile, int wave, int lane) {
    // ROWS x 128 bytes, 8 rows (1 KiB) per wave-instruction; 16-byte chunk c of row r lands at chunk c ^ ((r >> 1) & 7)
    constexpr int PER_WAVE = ROWS / NWAVES;
#pragma unroll
    for (int q = 0; q < PER_WAVE / 8; ++q) {
        const int rbase = wave * PER_WAVE + q * 8;
        const int rl = rbase + (lane >> 3);
        const int chunk = (lane & 7) ^ ((rl >> 1) & 7);
        int gr = row0 + rl;
        gr = gr < row_max ? gr : row_max;
        const uint8_t* g = src + (long long)gr * ld + k0 + chunk * 16;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(g), LDS_PTR(lds_tile + rbase * 128), 16, 0, 0);
    }
}

// the 32 bytes k = 32 g .. 32 g + 31 of one row (A and W use the same lane -> k map, so the products pair up)
__device__ __forceinline__ i32x8 lds_frag8(const char* tile, int row, int g) {
    const int sw = (row >> 1) & 7;
    const i32x4 lo = *reinterpret_cast<const i32x4*>(tile + row * 128 + (((2 * g) ^ sw) << 4));
    const i32x4 hi = *reinterpret_cast<const i32x4*>(tile + row * 128 + (((2 * g + 1) ^ sw) << 4));
    return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_fp8_kernel(GemmArgs p, const float* __restrict__ sa,
                                                                          const float* __restrict__ sw, int GM) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NWAVES = WAVES_M * WAVES_N;
    constexpr int TILE_A = BM * BK8, TILE_W = BN * BK8, STAGE = TILE_A + TILE_W;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MI = WM / 16, NI = WN / 16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    const int id = xcd_remap(blockIdx.x, nwg);
    // group-M order: GM row-tiles sweep one column-tile before moving on (W panels are re-read once per group of rows)
    const int per_group = GM * tiles_n;
    const int group = id / per_group, first_m = group * GM;
    const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int in_g = id - group * per_group;
    const int tm = first_m + in_g % gsz, tn = in_g / gsz;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.z;

    const uint8_t* A = reinterpret_cast<const uint8_t*>(p.A) + (long long)z * p.a_bs;
    const uint8_t* W = reinterpret_cast<const uint8_t*>(p.W);
    const int nk = p.K / BK8;

    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * STAGE;
        stage_tile8<BM, NWAVES>(A, p.lda, m0, p.M - 1, kt * BK8, base, wave, lane);
        stage_tile8<BN, NWAVES>(W, p.ldw, n0, p.N - 1, kt * BK8, base + TILE_A, wave, lane);
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int fr = lane & 15, fq = lane >> 4;
    const int unit = 0x7f7f7f7f;                            // E8M0 block scales: 2^(127 - 127) in every byte

    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* ta = smem + (kt & 1) * STAGE;
        const char* tw = ta + TILE_A;
        i32x8 fa[MI], fw[NI];
#pragma unroll
        for (int j = 0; j < MI; ++j) fa[j] = lds_frag8(ta, wm * WM + j * 16 + fr, fq);
#pragma unroll
        for (int i = 0; i < NI; ++i) fw[i] = lds_frag8(tw, wn * WN + i * 16 + fr, fq);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw[i], fa[j], acc[i][j], 0, 0, 0, unit, 0, unit);
    }

    // ---- row and channel scales, then the common epilogue.  Lane holds C[m][n4 .. n4+3], m = m_base + 16 j,
    // n4 = n_base + 16 i  (W fragment = A operand: the 16 x 16 result is transposed, as in every GEMM kernel here)
    const int m_base = m0 + wm * WM + fr, n_base = n0 + wn * WN + fq * 4;
    float ra[MI];
#pragma unroll
    for (int j = 0; j < MI; ++j) {
        const int m = m_base + 16 * j;
        ra[j] = sa[(long long)z * p.M + (m < p.M ? m : p.M - 1)];
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n4 = n_base + 16 * i;
        const f32x4 rw = n4 < p.N ? *reinterpret_cast<const f32x4*>(sw + n4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] *= ra[j] * rw[e];
    }
    auto run = [&](auto act_tag) {
        epilogue_block<decltype(act_tag)::value, NI, MI, (NI * MI > 16 ? 1 : NI)>(p, z, m_base, n_base, acc);
    };
    dispatch_act_big(p.act, run);
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_fp8(const GemmArgs& a, const float* sa, const float* sw, int batch, hipStream_t s) {
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n, 1, batch);
    const size_t lds = 2 * (BM + BN) * BK8;
    static std::atomic<unsigned long long> attr_done{0};
    if (bya_allow_big_lds(reinterpret_cast<const void*>(gemm_fp8_kernel<BM, BN, WAVES_M, WAVES_N>), (int)lds, attr_done) != BYA_OK)
        return BYA_ERR_LAUNCH;
    const int gm = FP8_GROUP_M;
    BYA_LAUNCH((gemm_fp8_kernel<BM, BN, WAVES_M, WAVES_N>), grid, dim3(64 * WAVES_M * WAVES_N), lds, s, a, sa, sw, gm);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

}  // namespace

