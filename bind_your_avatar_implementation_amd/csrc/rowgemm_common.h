// Shared pieces of the router's K = 512 row-stationary kernels (rowgemm.hip: one Linear per launch; rowchain.hip: chains of
// them that keep the intermediate activation in registers).  Everything that decides the BITS of a result lives here once:
// the W-chunk ring's layout, the order in which a lane's fragments meet the matrix core, the GELU.  The kernels of the two
// files agree bit for bit because they share it (tests/test_kernels_gpu.py, "fused ... is bit-identical to the pair").
#pragma once
#include "bya_common.h"

namespace rowk {

constexpr int RK = 512;                   // K of every router Linear
constexpr int CH = 64;                    // output columns per W chunk (one 64-KiB LDS ring stage)
constexpr int NJ = CH / 16;               // 16-column W fragments per chunk
constexpr int LPC = CH / 4;               // columns a lane owns per chunk in the attention kernels' q / k layout (16 g + 4 j + e)
constexpr int STAGE_BYTES = CH * RK * 2;  // 64 KiB

// Which output column of a chunk sits in MFMA tile j, tile row i = 4 g + e (LDS row (i, j) of a stage), and the first of the
// 8 columns lane group g ends up with in its u-th 16-byte piece: the four lane groups of a token hold 8 g .. + 7 of the
// chunk's half u -- 64 CONTIGUOUS bytes per row and store instruction.  The same numbers make the piece an operand
// fragment of the NEXT K = 512 product: columns 64 c + 32 u + 8 g .. + 7 are k-step 2 c + u, lane group g.
__device__ __forceinline__ constexpr uint32_t wrow_of(int i, int j) {
    return (uint32_t)((CH / 2) * (j >> 1) + 8 * (i >> 2) + 4 * (j & 1) + (i & 3));
}
__device__ __forceinline__ constexpr uint32_t lane_col(uint32_t g, int u) { return (uint32_t)(CH / 2) * (uint32_t)u + 8u * g; }
// the attention kernels' q / k chunks: lane group g ends up with 16 consecutive features 16 g + 4 j + e of the head
__device__ __forceinline__ constexpr uint32_t wrow_lpc(int i, int j) { return (uint32_t)(LPC * (i >> 2) + 4 * j + (i & 3)); }

template <int OFF>
__device__ __forceinline__ void lds_read_w(bf16x8& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_read_f(f32x4& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N>
__device__ __forceinline__ void lgkm_wait(bf16x8 (&w)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "i"(N));
}
template <int N>
__device__ __forceinline__ void lgkm_wait(bf16x8 (&w)[2]) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(w[0]), "+v"(w[1]) : "i"(N));
}
// the W fragments (column blocks j) of k-step 4*KH + kl; kl selects the lane-constant address
template <int KH, int NJ_>
__device__ __forceinline__ void read_quad(bf16x8 (&wf)[NJ_], uint32_t addr) {
    lds_read_w<0 * 16384 + KH * 256>(wf[0], addr);
    lds_read_w<1 * 16384 + KH * 256>(wf[1], addr);
    if constexpr (NJ_ == 4) {
        lds_read_w<2 * 16384 + KH * 256>(wf[2], addr);
        lds_read_w<3 * 16384 + KH * 256>(wf[3], addr);
    }
}
template <int NJ_>
__device__ __forceinline__ void read_kstep(bf16x8 (&wf)[NJ_], const uint32_t (&wa)[4], int ks) {
    switch (ks >> 2) {            // ks is a constant after unrolling: the switch folds away
        case 0: read_quad<0>(wf, wa[ks & 3]); break;
        case 1: read_quad<1>(wf, wa[ks & 3]); break;
        case 2: read_quad<2>(wf, wa[ks & 3]); break;
        default: read_quad<3>(wf, wa[ks & 3]); break;
    }
}

// A lane constant the compiler may not hoist out of the chunk loop: hoisted address registers do not fit beside the
// X-fragment registers, get spilled, and every scratch reload comes with an s_waitcnt vmcnt(0) -- in front of each LDS-DMA
// instruction that serialised eight memory round trips per chunk.  Recomputing an address costs one or two VALU ops.
__device__ __forceinline__ uint32_t lane_now() {      // the lane id, recomputed where it is used (volatile: never hoisted)
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// GELU(erf) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 output step): one rcp,
// one exp2 and six FMAs instead of libm's two-regime erff -- the MLP epilogue evaluates it 18 M times per launch.
__device__ __forceinline__ float gelu_erf_f(float v) {
    const float x = v * 0.70710678118654752f, ax = fabsf(x);
    const float tt = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float poly = fmaf(1.061405429f, tt, -1.453152027f);
    poly = fmaf(poly, tt, 1.421413741f);
    poly = fmaf(poly, tt, -0.284496736f);
    poly = fmaf(poly, tt, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    const float erf_abs = fmaf(-poly * tt, e, 1.0f);
    return 0.5f * v * (1.0f + copysignf(erf_abs, x));
}

// Row statistics of a 16-row tile held as MFMA operand fragments (xf[ks]: lane (g, t) = row t, k = 32 ks + 8 g .. + 7), on the
// matrix core: sum(x) = ones . x^T, sum(x^2) = diag(x . x^T) -- fp32 accumulation of exact bf16 products.  -> mean and
// 1 / sqrt(var + eps) of row t = lane & 15, the same value in the four lanes of a row.
__device__ __forceinline__ void tile_row_stats(const bf16x8 (&xf)[16], uint32_t to, float eps, float& mean, float& rstd) {
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    f32x4 sm = {0.f, 0.f, 0.f, 0.f}, gr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, xf[ks], sm, 0, 0, 0);
        gr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[ks], xf[ks], gr, 0, 0, 0);
    }
    // lane (g, t) holds <x_{4g+e}, x_t>; the diagonal of token t sits in lane (t>>2, t), register t&3
    const int e = to & 3;
    const float d = e == 0 ? gr[0] : e == 1 ? gr[1] : e == 2 ? gr[2] : gr[3];
    const float sq = __shfl(d, (int)(to + 16 * (to >> 2)));
    const float mu = sm[0] * (1.0f / RK);
    const float var = fmaxf(sq * (1.0f / RK) - mu * mu, 0.0f);
    mean = mu;
    rstd = rsqrtf(var + eps);
}

}  // namespace rowk
