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
};

constexpr int KV_TILE = 64;

// XOR applied to the 16-byte chunk index of LDS row `row` (both on the staging source and on the reads):
//  K (ds_read_b128, 32 rows x one chunk per half-wave): 128-B rows -> (row>>1)&7, 256-B rows -> row&15
//  V (ds_read_b64_tr_b16, 4 rows x 64 B per half-wave):  128-B rows -> ((row>>1)&1)<<2, 256-B rows -> (row&3)<<2
template <int D> __device__ __forceinline__ int kswz(int row) { return D == 64 ? ((row >> 1) & 7) : (row & 15); }
template <int D> __device__ __forceinline__ int vswz(int row) { return D == 64 ? (((row >> 1) & 1) << 2) : ((row & 3) << 2); }

}  // namespace

// defined in attn_w4.hip: joint attention, head_dim 64, scores pre-scaled and bounded (one wave per SIMD, 512 query rows
// per workgroup); called from bya_attn_fwd
int bya_launch_attn_w4(const void* args, hipStream_t stream);
