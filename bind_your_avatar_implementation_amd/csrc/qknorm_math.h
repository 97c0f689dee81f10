// Arithmetic of the per-head q/k LayerNorm(64) + interleaved-pair RoPE (diffusers CogVideoXAttnProcessor2_0 +
// apply_rotary_emb; reference models/transformer.py:204-208), shared by its two homes so that they agree BIT FOR BIT:
//   * qknorm_rope_kernel (norm.hip): in place on the stored q / k, 8 lanes x 8 values per head row;
//   * the QKV projection's epilogue (gemm_v4.hip, QKN instance): on the accumulators, before q / k are stored at all.
// A head row is 64 values as eight groups g of 8 consecutive ones.  Every operation whose rounding depends on its order is
// spelled out here: sums go 8 values in sequence, then the tree (g ^ 1), (g ^ 2), (g ^ 4); products that feed an addition
// are explicit fmaf; nothing is left to -ffp-contract.
#pragma once
#include "bya_common.h"

__device__ __forceinline__ float qkn_sum8(const float (&v)[8]) {
    float s = v[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) s += v[e];
    return s;
}
// v[e] -= mean; returns sum of squares of the centred values (sequential, fused multiply-adds)
__device__ __forceinline__ float qkn_centre_sq8(float (&v)[8], float mean) {
    float sq = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        v[e] -= mean;
        sq = __builtin_fmaf(v[e], v[e], sq);
    }
    return sq;
}
// centred values -> (x_hat * w + b), rotated by (cos, sin) where `rope` (a per-lane flag: branch-free, the rotation is
// computed for every lane and selected), times k_scale when it is not 1
__device__ __forceinline__ void qkn_finish8(float (&v)[8], float rstd, const float (&w)[8], const float (&b)[8], bool rope,
                                            const float (&cc)[8], const float (&ss)[8], float k_scale) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e] * rstd, w[e], b[e]);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {   // pair (2i, 2i+1): rot = (-x[2i+1], x[2i])
        o[e] = __builtin_fmaf(v[e], cc[e], -(v[e + 1] * ss[e]));
        o[e + 1] = __builtin_fmaf(v[e + 1], cc[e + 1], v[e] * ss[e + 1]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = rope ? o[e] : v[e];
    if (k_scale != 1.0f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= k_scale;
    }
}
