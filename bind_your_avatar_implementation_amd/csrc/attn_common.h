// Shared by the attention translation units (attn.hip: compiler-scheduled kernels, MFMA results in arch VGPRs;
// attn_w4.hip: the hand-placed one-wave-per-SIMD joint-attention kernel, accumulators in AGPRs).
#pragma once
#include "bya_common.h"
#include "../../include/bya.h"

namespace {

struct AttnArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o;
    int heads, nb1, nb2, Sq, Skv, nqt;
    long long q_s1, q_s2, q_row, k_s1, k_s2, k_row, v_s1, v_s2, v_row, o_s1, o_s2, o_row;
    float scale_log2;  // scale * log2(e)
    int prescaled;     // scores already in exp2 units (scale folded into k by the producer)
    float score_bound; // > 0: |score| <= bound guaranteed by the caller -> static-offset softmax (no running maximum)
    float* sk_part;    // stream-K exchange slots / flags of the joint-attention kernel (attn_w4.hip; set by its launcher)
    unsigned* sk_flags;
    int o_wide;        // o and its strides are 16-byte aligned: the epilogue stores 16 bytes per lane (store_o_tile)
    // data-dependent score bound (bya_attn_desc.bound_dev; attn_w4.hip): squared norms [slots][2][bound_heads], this launch's
    // bh at column bound_bh0 + bh; heads whose bound exceeds bound_limit are left to the running-maximum kernel, which
    // runs with only_flagged = fallback and skips every other head
    const float* bound_dev; int bound_slots, bound_heads, bound_bh0; float bound_limit;
    int* fallback;
    const int* only_flagged;
};

constexpr int KV_TILE = 64;

// XOR applied to the 16-byte chunk index of LDS row `row` (both on the staging source and on the reads):
//  K (ds_read_b128, 32 rows x one chunk per half-wave): 128-B rows -> (row>>1)&7, 256-B rows -> row&15
//  V (ds_read_b64_tr_b16, 4 rows x 64 B per half-wave):  128-B rows -> ((row>>1)&1)<<2, 256-B rows -> (row&3)<<2
// One 32-column tile of an output row block, O^T accumulator layout: lane (r, hf) holds row r's columns 8 gq + 4 hf ..+3
// in acc[4 gq ..+3].  As it stands that is four 8-byte stores per lane, each instruction touching 32 rows with 16 bytes.
// wide: the half-waves trade pieces first (v_permlane32_swap: lanes 32-63 of the first operand <-> lanes 0-31 of the
// second), after which a lower lane holds columns 16 k ..+7 and its partner 16 k + 8 ..+7 -- two 16-byte stores per
// lane, 32 contiguous bytes per row and instruction: half the store instructions and requests, same bytes.  The short
// attention launches (router: 21 key tiles per workgroup) are store-issue-bound in their tail.
__device__ __forceinline__ void store_o_tile(bf16_t* orow_d, const f32x16& acc, float inv, int hf, bool valid, bool wide) {
    uint32_t w[4][2];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        w[gq][0] = pack2bf(acc[gq * 4 + 0] * inv, acc[gq * 4 + 1] * inv);
        w[gq][1] = pack2bf(acc[gq * 4 + 2] * inv, acc[gq * 4 + 3] * inv);
    }
    if (wide) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(w[2 * k][0], w[2 * k + 1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(w[2 * k][1], w[2 * k + 1][1], false, false);
            u32x4 o;
            o[0] = s0[0]; o[1] = s1[0]; o[2] = s0[1]; o[3] = s1[1];
            if (valid) *reinterpret_cast<u32x4*>(orow_d + 16 * k + 8 * hf) = o;
        }
    } else if (valid) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            u32x2 o;
            o[0] = w[gq][0]; o[1] = w[gq][1];
            *reinterpret_cast<u32x2*>(orow_d + gq * 8 + hf * 4) = o;
        }
    }
}

template <int D> __device__ __forceinline__ int kswz(int row) { return D == 64 ? ((row >> 1) & 7) : (row & 15); }
template <int D> __device__ __forceinline__ int vswz(int row) { return D == 64 ? (((row >> 1) & 1) << 2) : ((row & 3) << 2); }

}  // namespace

// defined in attn_w4.hip: joint attention, head_dim 64, scores pre-scaled and bounded (one wave per SIMD, 512 query rows
// per workgroup); called from bya_attn_fwd
int bya_launch_attn_w4(const void* args, hipStream_t stream);
