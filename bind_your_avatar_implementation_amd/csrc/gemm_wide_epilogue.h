// The 16-byte ("wide") epilogues and the LDS-DMA helpers shared by the persistent one-wave-per-SIMD GEMM kernels:
// gemm_v4.hip (256 x 256 tiles, a wave owns 128 x 128: NJ = 8 row blocks) and gemm_v5.hip (128 x 256 tiles, a wave owns
// 64 x 128: NJ = 4).  Both stage the W rows of a wave's 128-column span in the permuted order that makes a lane's eight
// accumulator tiles i = 0..7 hold EIGHT CONSECUTIVE output columns (see the top of gemm_v4.hip).
#pragma once
#include "gemm_common.h"
#include "qknorm_math.h"

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

// Wide epilogue of one wave.  The lane (fr = lane & 15, fq = lane >> 4) holds, for row block j and accumulator register e,
// the EIGHT consecutive columns  n8 = n_wave + (4 e + fq) * 8 + i,  i = 0..7  in acc[i][j][e]  (W rows are staged in the
// permuted order described at the top), of row  m = m_wave + 16 j + fr.
// Split tiles (see the top).  A slab holds a tile's partial sums in the order this epilogue walks the accumulators:
// unit (wave, j, e, half) = the lane's values i = 4 half .. 4 half + 3 of row block j, register e: 16 bytes per lane at
// ((wave * 64 + (j * 4 + e) * 2 + half) * 64 + lane) * 16, so that one store instruction writes eight whole 128-byte lines.
// raw_out != null: this workgroup is a WRITER -- its accumulators go to that slab with write-through (sc1) stores and
// nothing else happens.  Same call site as the ordinary epilogue and through one VALU multiply: a second kind of consumer
// of the asm-owned accumulators (a plain store of them) made hipcc put the store's data tuples into AGPRs too and evict
// accumulators to scratch right behind their last MFMA, inside the K-loop -- 250 registers of scratch traffic per tile.
template <int ACT, int JB, bool SPLIT, bool CONV = false, int NJ = 8>
__device__ __forceinline__ void epilogue_wide(const GemmArgs& p, int z, int m_wave, int n_wave, int fr, int fq,
                                              const f32x4 (&acc)[8][NJ], int wave, int lane, float* raw_out = nullptr) {
    const bool has_res = p.res != nullptr, has_gate = p.gate0 != nullptr, has_bias = p.bias != nullptr;
    const bool has_rs = p.bias_rowscale != nullptr;
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((has_res ? p.res : p.C) + (long long)z * p.res_bs), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.C + (long long)z * p.c_bs), 0, 0x7fffffff, 0x00020000);
    const char* g0base = reinterpret_cast<const char*>(p.gate0 + (long long)z * p.gate_bs);
    const char* g1base = reinterpret_cast<const char*>(p.gate1 + (long long)z * p.gate_bs);
    u32x4 bv[4], g0[4], g1[4];
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
        if (has_gate) {
            g0[e] = *reinterpret_cast<const u32x4*>(g0base + ncb[e]);
            g1[e] = *reinterpret_cast<const u32x4*>(g1base + ncb[e]);
        }
    }
#pragma unroll
    for (int jb = 0; jb < NJ; jb += JB) {
        u32x4 rv[JB][4];
        float rs[JB];
        bool mok[JB];
        uint32_t roff[JB], coff[JB];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
            const int m = m_wave + 16 * (jb + jj) + fr;
            mok[jj] = m < p.M;
            uint32_t mc = mok[jj] ? (uint32_t)m : 0u;
            if constexpr (CONV) {
                // row m = padded pixel (t, h, w) of the output grid [To, Hp, Wp]: kept if it is a real pixel, and then stored
                // (and its residual read) at row (t H + h) W + w of the unpadded output
                const uint32_t plane = (uint32_t)(p.conv_Hp * p.conv_Wp);
                const uint32_t t = mc / plane, rem = mc - t * plane;
                const uint32_t h = rem / (uint32_t)p.conv_Wp, w = rem - h * (uint32_t)p.conv_Wp;
                mok[jj] = mok[jj] && h < (uint32_t)p.conv_H && w < (uint32_t)p.conv_W && t < (uint32_t)p.conv_To;
                mc = mok[jj] ? (t * (uint32_t)p.conv_H + h) * (uint32_t)p.conv_W + w : 0u;
            }
            rs[jj] = has_rs ? p.bias_rowscale[(long long)z * p.M + mc] : 1.0f;
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
                for (int i = 0; i < 8; ++i) a0[i] = acc[i][j][e];
                if (SPLIT && raw_out) {
                    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)raw_out, 0, (int)GEMM_WS_SLAB_BYTES, 0x00020000);
                    const uint32_t so = (uint32_t)(((wave * 64 + (j * 4 + e) * 2) * 64 + lane) * 16);
                    // the values pass through one VALU multiply by an opaque 1.0: stored as they are, hipcc put the store's
                    // data tuples into AGPRs too and evicted accumulators to scratch inside the K-loop to make room
                    float one = 1.0f;
                    asm volatile("" : "+s"(one));
                    const f32x4 lof = {a0[0] * one, a0[1] * one, a0[2] * one, a0[3] * one};
                    const f32x4 hif = {a0[4] * one, a0[5] * one, a0[6] * one, a0[7] * one};
                    const u32x4 lo = __builtin_bit_cast(u32x4, lof), hi = __builtin_bit_cast(u32x4, hif);
                    __builtin_amdgcn_raw_buffer_store_b128(lo, rsS, so, 0, 16 /* sc1 */);
                    __builtin_amdgcn_raw_buffer_store_b128(hi, rsS, so + 1024, 0, 16 /* sc1 */);
                    continue;
                }
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

// The packed q|k|v projection's epilogue with the per-head q/k LayerNorm(64) + RoPE inside (QKN instance; reference
// models/transformer.py:204-208 = diffusers CogVideoXAttnProcessor2_0: norm_q / norm_k, apply_rotary_emb on the video rows).
// Round 4 wrote q, k and launched bya_qknorm_rope on them: 473 MB of traffic per layer against 218 MB if q and k are written
// once.  In the wide epilogue's layout a lane (fr, fq) holds, for row block j, the eight consecutive columns
// n_wave + (4 e + fq) * 8 + i: the wave's 128 columns are two heads, head hh = registers e = 2 hh, 2 hh + 1, and a head row is
// spread over the FOUR lanes fq = 0 .. 3 of one fr -- group g = 4 (e & 1) + fq of qknorm_math.h's eight.  The tree (g ^ 1),
// (g ^ 2), (g ^ 4) is therefore lane ^ 16, lane ^ 32, then the lane's own two registers: same operands, same order, same bits
// as the stand-alone kernel.  The projection is rounded to bf16 first (the value the two-launch path stored and read back).
// v tiles (n_wave >= 2 width) take the plain bias epilogue.
template <int NJ = 8>
__device__ __forceinline__ void epilogue_qkn(const GemmArgs& p, int z, int m_wave, int n_wave, int fr, int fq, const f32x4 (&acc)[8][NJ]) {
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C + (long long)z * p.c_bs), 0, 0x7fffffff, 0x00020000);
    const int tsel = n_wave / p.qkn_width;                       // 0 = q, 1 = k, 2 = v (a tile never straddles: width % 128 == 0)
    const bool has_bias = p.bias != nullptr;
    u32x4 bv[4];
    uint32_t colb[4];
    bool nok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int n8 = n_wave + (4 * e + fq) * 8;
        nok[e] = n8 < p.N;
        const uint32_t ncb = nok[e] ? (uint32_t)n8 * 2u : 0u;
        colb[e] = ((uint32_t)(n8 / p.n_split) * (uint32_t)p.c_split_stride + (uint32_t)(n8 % p.n_split)) * 2u;
        bv[e] = has_bias ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.bias) + ncb) : u32x4{0u, 0u, 0u, 0u};
    }
    // LayerNorm parameters of this lane's columns: e and e + 2 sit at the same place of their heads
    float wv[2][8], bb[2][8];
    const int tn = tsel < 2 ? tsel : 0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int hc = (4 * e + fq) * 8;
        unpack8(*reinterpret_cast<const u32x4*>(p.qkn_w[tn] + hc), wv[e]);
        unpack8(*reinterpret_cast<const u32x4*>(p.qkn_b[tn] + hc), bb[e]);
    }
    const float ks = tsel == 1 ? p.qkn_kscale : 1.0f;
    const long long trows = (long long)p.M - p.qkn_text_rows;
    const int tbytes = trows > 0 && p.qkn_cos ? (int)(trows * 256 > 0x7fffffffLL ? 0x7fffffffLL : trows * 256) : 0;
    const __amdgpu_buffer_rsrc_t rsCos = __builtin_amdgcn_make_buffer_rsrc((void*)p.qkn_cos, 0, tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsSin = __builtin_amdgcn_make_buffer_rsrc((void*)p.qkn_sin, 0, tbytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int m = m_wave + 16 * j + fr;
        const bool mok = m < p.M;
        const uint32_t coff = (mok ? (uint32_t)m : 0u) * (uint32_t)(p.ldc * 2);
        const bool rope = tsel < 2 && m >= p.qkn_text_rows && mok;
        // rotary-table row of this token, through buffer descriptors: no branch (other rows read zeros from an out-of-range
        // offset and do not use them), so hipcc can keep the next row block's loads in flight under this one's arithmetic
        float cc[2][8], ss[2][8];
        {
            const uint32_t t0 = rope ? (uint32_t)(m - p.qkn_text_rows) * 256u : 0xffffffffu;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const uint32_t off = rope ? t0 + (uint32_t)((4 * e + fq) * 32) : 0xffffffffu;
                const f32x4 c0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsCos, off, 0, 0));
                const f32x4 c1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsCos, rope ? off + 16u : off, 0, 0));
                const f32x4 s0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsSin, off, 0, 0));
                const f32x4 s1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsSin, rope ? off + 16u : off, 0, 0));
#pragma unroll
                for (int i = 0; i < 4; ++i) { cc[e][i] = c0[i]; cc[e][4 + i] = c1[i]; ss[e][i] = s0[i]; ss[e][4 + i] = s1[i]; }
            }
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            float v[2][8];
#pragma unroll
            for (int el = 0; el < 2; ++el) {
                const int e = 2 * hh + el;
                float b8[8];
                unpack8(bv[e], b8);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[el][i] = acc[i][j][e] + b8[i];
            }
            if (tsel < 2) {
                // the projection as the two-launch path stored it: one rounding to bf16
#pragma unroll
                for (int el = 0; el < 2; ++el) {
                    const u32x4 r = pack8(v[el]);
                    unpack8(r, v[el]);
                }
                // lane ^ 16 and lane ^ 32 partners by v_permlane16_swap / v_permlane32_swap (one VALU instruction each; a
                // __shfl_xor is a ds_bpermute round trip): swapping a value with itself leaves (own, partner's) in the two results
                auto add16 = [](float x) {
                    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
                    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
                };
                auto add32 = [](float x) {
                    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
                    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
                };
                float s0 = add32(add16(qkn_sum8(v[0]))), s1 = add32(add16(qkn_sum8(v[1])));
                const float mean = (s0 + s1) * (1.0f / 64);
                float q0 = add32(add16(qkn_centre_sq8(v[0], mean))), q1 = add32(add16(qkn_centre_sq8(v[1], mean)));
                const float rstd = rsqrtf((q0 + q1) * (1.0f / 64) + p.qkn_eps);
                qkn_finish8(v[0], rstd, wv[0], bb[0], rope, cc[0], ss[0], ks);
                qkn_finish8(v[1], rstd, wv[1], bb[1], rope, cc[1], ss[1], ks);
            }
#pragma unroll
            for (int el = 0; el < 2; ++el) {
                const int e = 2 * hh + el;
                __builtin_amdgcn_raw_buffer_store_b128(pack8(v[el]), rsC, (mok && nok[e]) ? coff + colb[e] : 0xffffffffu, 0, 0);
            }
        }
    }
}

}  // namespace
