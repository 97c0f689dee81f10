// HBM-bound normalisation kernels: LayerNorm (+ affine, + AdaLN modulation) and the per-head
// q/k LayerNorm(64) + RoPE.  One wave per row, 16-byte (8 x bf16) or 8-byte (4 x bf16) lane accesses,
// the whole row held in registers (single HBM read), fp32 statistics, wave-shuffle reductions.
//
// Replaces: CogVideoXLayerNormZero (models/transformer.py:233,251), nn.LayerNorm norm_final + AdaLayerNorm
// norm_out (:944,:948), PerceiverCrossAttention.norm1/norm2 (models/router.py:247-248), MultiIPRouter
// norm_q/norm_k (:380,:382), SpatialTemporalAttentionBlock.norm1-4 (:475-491), AudioAwareModel norm_q
// (models/audio_model.py:249), diffusers Attention.norm_q/norm_k + apply_rotary_emb (transformer.py:204-208).
#include "bya_common.h"
#include "qknorm_math.h"
#include "../../include/bya.h"
#include <stdlib.h>
#include "options.h"

namespace {

struct LnArgs {
    const bf16_t* x; bf16_t* y; const bf16_t* w; const bf16_t* b;
    const bf16_t* shift0; const bf16_t* scale0; const bf16_t* shift1; const bf16_t* scale1;
    long long rows_per_batch, ldx, ldy, x_bs, y_bs, mod_bs, split;
    int batch;
    float eps;
    // fp8 output (bya_layernorm_fp8): y is then a byte matrix (ldy / y_bs in bytes) and q_scale[z * rows_per_batch + row]
    // receives the row's scale
    float* q_scale;
};

template <int VEC>
__device__ __forceinline__ void load_vec(const bf16_t* p, float* f) {
    if constexpr (VEC == 8) {
        unpack8(*reinterpret_cast<const u32x4*>(p), f);
    } else {
        const u32x2 raw = *reinterpret_cast<const u32x2*>(p);
        f[0] = bflo(raw[0]); f[1] = bfhi(raw[0]); f[2] = bflo(raw[1]); f[3] = bfhi(raw[1]);
    }
}

// D = 64 * VEC * NV ; each lane owns NV vectors of VEC contiguous elements, vector v at column (v*64 + lane)*VEC
template <int VEC, int NV, bool Q8 = false>
__global__ __launch_bounds__(256) void layernorm_kernel(LnArgs p) {
    constexpr int D = 64 * VEC * NV;
    const int lane = threadIdx.x & 63;
    const long long row_lin = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long total = p.rows_per_batch * p.batch;
    if (row_lin >= total) return;
    const int z = (int)(row_lin / p.rows_per_batch);
    const long long row = row_lin - (long long)z * p.rows_per_batch;
    const bf16_t* x = p.x + z * p.x_bs + row * p.ldx;
    bf16_t* y = Q8 ? nullptr : p.y + z * p.y_bs + row * p.ldy;

    float v[NV][VEC];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * VEC;
        load_vec<VEC>(x + col, v[i]);
#pragma unroll
        for (int e = 0; e < VEC; ++e) sum += v[i][e];
    }
    const float mean = wave_sum(sum) * (1.0f / D);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < VEC; ++e) { const float d = v[i][e] - mean; sq += d * d; }
    const float rstd = rsqrtf(wave_sum(sq) * (1.0f / D) + p.eps);

    const bf16_t* shift = nullptr; const bf16_t* scale = nullptr;
    if (p.shift0) {
        const bool first = row < p.split;
        shift = (first ? p.shift0 : p.shift1) + z * p.mod_bs;
        scale = (first ? p.scale0 : p.scale1) + z * p.mod_bs;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * VEC;
        float o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = (v[i][e] - mean) * rstd;
        if (p.w && p.b && shift) {
            // affine + modulation as ONE fma with A = w (1 + scale), B = b (1 + scale) + shift: the arithmetic of
            // layernorm_adaln_rows_kernel below (which keeps A, B in registers), so the two kernels and the fp8 form agree bit for bit
            float wv[VEC], bv[VEC], sc[VEC], sh[VEC];
            load_vec<VEC>(p.w + col, wv);
            load_vec<VEC>(p.b + col, bv);
            load_vec<VEC>(scale + col, sc);
            load_vec<VEC>(shift + col, sh);
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = fmaf(o[e], wv[e] * (1.0f + sc[e]), fmaf(bv[e], 1.0f + sc[e], sh[e]));
        } else {
            if (p.w) {
                float wv[VEC], bv[VEC];
                load_vec<VEC>(p.w + col, wv);
                if (p.b) load_vec<VEC>(p.b + col, bv);
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] = o[e] * wv[e] + (p.b ? bv[e] : 0.f);
            }
            if (shift) {
                float sc[VEC], sh[VEC];
                load_vec<VEC>(scale + col, sc);
                load_vec<VEC>(shift + col, sh);
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] = o[e] * (1.0f + sc[e]) + sh[e];
            }
        }
        if constexpr (Q8) {
            // keep the result -- rounded to bf16 exactly as the bf16 kernel would store it -- for the row maximum
#pragma unroll
            for (int e = 0; e < VEC; e += 2) {
                const uint32_t w2 = pack2bf(o[e], o[e + 1]);
                v[i][e] = bflo(w2);
                v[i][e + 1] = bfhi(w2);
            }
        } else if constexpr (VEC == 8) {
            *reinterpret_cast<u32x4*>(y + col) = pack8(o);
        } else {
            u32x2 w2; w2[0] = pack2bf(o[0], o[1]); w2[1] = pack2bf(o[2], o[3]);
            *reinterpret_cast<u32x2*>(y + col) = w2;
        }
    }
    if constexpr (Q8) {
        // bya_quantize_rows_fp8 of the row this wave just normalised (same bytes as LayerNorm -> bf16 -> quantiser)
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < VEC; ++e) amax = fmaxf(amax, fabsf(v[i][e]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        const float inv = amax > 0.f ? (float)(448.0 / (double)amax) : 0.f;
        if (lane == 0) p.q_scale[z * p.rows_per_batch + row] = amax > 0.f ? (float)((double)amax / 448.0) : 1.0f;
        uint8_t* q = reinterpret_cast<uint8_t*>(p.y) + z * p.y_bs + row * p.ldy;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (i * 64 + lane) * VEC;
            uint32_t w[VEC / 4];
#pragma unroll
            for (int h = 0; h < VEC / 4; ++h)
                w[h] = f32_to_e4m3(v[i][4 * h] * inv) | (f32_to_e4m3(v[i][4 * h + 1] * inv) << 8) |
                       (f32_to_e4m3(v[i][4 * h + 2] * inv) << 16) | (f32_to_e4m3(v[i][4 * h + 3] * inv) << 24);
            if constexpr (VEC == 8) {
                u32x2 o2; o2[0] = w[0]; o2[1] = w[1];
                *reinterpret_cast<u32x2*>(q + col) = o2;
            } else {
                *reinterpret_cast<uint32_t*>(q + col) = w[0];
            }
        }
    }
}

// Affine / AdaLN form with the parameter vectors in REGISTERS (the DiT's norm1 / norm2 / norm_out and the audio / face norm_q:
// ~150 launches per step, 218 MB each).  layernorm_kernel above re-reads w, b, scale and shift for every row -- 24 KB of L1 / L2 traffic per 6-KB row, four
// times the row itself.  Here a wave owns a contiguous range of rows and keeps  A = w (1 + scale),  B = b (1 + scale) + shift
// in fp32 (2 x 48 registers at D = 3072); they are rebuilt when the range crosses the text / video split or a batch
// boundary (different modulation vectors).  Per element:  y = fma((x - mean) rstd, A, B)  -- ((x - mean) rstd w + b)(1 + scale)
// + shift up to fp32 reassociation, one rounding to bf16; layernorm_kernel computes the very same fma, so the kernels agree.
template <int NV, bool MOD>
__global__ __launch_bounds__(256) void layernorm_adaln_rows_kernel(LnArgs p, int rows_per_wave) {
    constexpr int VEC = 8, D = 64 * VEC * NV;
    const int lane = threadIdx.x & 63;
    const long long total = p.rows_per_batch * p.batch;
    const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    long long r0 = gw * rows_per_wave, r1 = r0 + rows_per_wave;
    if (r1 > total) r1 = total;
    float A[NV][VEC], B[NV][VEC];
    long long cur_set = -1;
    for (long long row_lin = r0; row_lin < r1; ++row_lin) {
        const int z = (int)(row_lin / p.rows_per_batch);
        const long long row = row_lin - (long long)z * p.rows_per_batch;
        const bool first = row < p.split;
        const long long set = 2 * (long long)z + (first ? 0 : 1);
        const bf16_t* x = p.x + z * p.x_bs + row * p.ldx;
        float v[NV][VEC];
#pragma unroll
        for (int i = 0; i < NV; ++i) load_vec<VEC>(x + (i * 64 + lane) * VEC, v[i]);
        if (set != cur_set) {
            cur_set = set;
            const bf16_t* shift = MOD ? (first ? p.shift0 : p.shift1) + z * p.mod_bs : nullptr;
            const bf16_t* scale = MOD ? (first ? p.scale0 : p.scale1) + z * p.mod_bs : nullptr;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int col = (i * 64 + lane) * VEC;
                float wv[VEC], bv[VEC];
                load_vec<VEC>(p.w + col, wv);
                load_vec<VEC>(p.b + col, bv);
                if constexpr (MOD) {
                    float sc[VEC], sh[VEC];
                    load_vec<VEC>(scale + col, sc);
                    load_vec<VEC>(shift + col, sh);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        A[i][e] = wv[e] * (1.0f + sc[e]);
                        B[i][e] = fmaf(bv[e], 1.0f + sc[e], sh[e]);
                    }
                } else {                                       // plain affine LayerNorm: the same fma as layernorm_kernel
#pragma unroll
                    for (int e = 0; e < VEC; ++e) { A[i][e] = wv[e]; B[i][e] = bv[e]; }
                }
            }
        }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < VEC; ++e) sum += v[i][e];
        const float mean = wave_sum(sum) * (1.0f / D);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float d = v[i][e] - mean; sq += d * d; }
        const float rstd = rsqrtf(wave_sum(sq) * (1.0f / D) + p.eps);
        bf16_t* y = p.y + z * p.y_bs + row * p.ldy;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float o[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = fmaf((v[i][e] - mean) * rstd, A[i][e], B[i][e]);
            *reinterpret_cast<u32x4*>(y + (i * 64 + lane) * VEC) = pack8(o);
        }
    }
}

template <int NV>
int launch_ln_adaln_rows(const LnArgs& a, hipStream_t s) {
    const long long total = a.rows_per_batch * a.batch;
    // about 16 waves per CU in flight (4096 waves), at least 2 rows each so the parameter set-up is amortised
    int rpw = (int)((total + 4095) / 4096);
    rpw = rpw < 2 ? 2 : rpw;
    const long long waves = (total + rpw - 1) / rpw;
    dim3 grid((unsigned)((waves + 3) / 4));
    if (a.shift0) BYA_LAUNCH((layernorm_adaln_rows_kernel<NV, true>), grid, dim3(256), 0, s, a, rpw);
    else BYA_LAUNCH((layernorm_adaln_rows_kernel<NV, false>), grid, dim3(256), 0, s, a, rpw);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

template <int VEC, int NV, bool Q8 = false>
int launch_ln(const LnArgs& a, hipStream_t s) {
    const long long total = a.rows_per_batch * a.batch;
    dim3 grid((unsigned)((total + 3) / 4));
    BYA_LAUNCH((layernorm_kernel<VEC, NV, Q8>), grid, dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

// ---- q/k per-head LayerNorm(64) + RoPE, in place.  8 lanes x 8 elements per (row, head); a wave covers
// 8 consecutive heads of one row (1 KiB contiguous per wave access).
struct QkArgs {
    bf16_t* q; bf16_t* k; const bf16_t* qw; const bf16_t* qb; const bf16_t* kw; const bf16_t* kb;
    const float* cos; const float* sin;
    int batch, S, heads, text_rows;
    long long ld, bs;
    float eps;
    float k_scale;     // multiplies the finished k (fp32, before the single rounding to bf16); 1 = off
    int only;          // 0: q and k; 1: q alone; 2: k alone (the sharded step norms q, starts its exchange, then norms k)
    float* stats;      // STATS instance: [slots][2][batch * heads] squared norms of the finished rows (include/bya.h)
    int stats_slots;
};

// (r4, measured and dropped: a row form -- one token row per wave iteration, the four parameter vectors in registers, a row's
// cos / sin piece loaded once for all heads of q and k -- bit-identical, 86-90 us against 78-86 us for this kernel;
// writing to a second buffer instead of in place -- 92-104 us against 78-90 us: the in-place stores hit lines the loads
// just brought into L2.  r2-r4 experiment arms of tools/timeslice/repro.py -- drained table loads, 32-bit index arithmetic,
// sc1 table loads -- closed and removed in round 5: the cause was the packed-fp32 arithmetic, see build.py.)
// STATS: every 8-lane group also raises its (slot, q | k, batch, head) entry of p.stats to the squared norm of the row it
// stores, taken from the ROUNDED values (what the attention kernel will read): one no-return atomic maximum per (row, head)
// on the bit pattern of a non-negative float, spread over `slots` copies of the table (blockIdx % slots) so that a table
// entry sees 1 / slots of the rows.  Only launched when the caller asks for statistics (engine: the worst-case score bound of
// a layer is too large): the plain instance is untouched.
template <bool STATS>
__global__ __launch_bounds__(256) void qknorm_rope_kernel(QkArgs p) {
    // one 8-lane group per (row, head) pair, pairs enumerated row-major over all batches: any head count works
    const int lane = threadIdx.x & 63;
    const long long pairs_per_tensor = (long long)p.batch * p.S * p.heads;
    const long long pair = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (lane >> 3);
    if (pair >= (p.only ? 1 : 2) * pairs_per_tensor) return;
    const int which = p.only ? p.only - 1 : (pair >= pairs_per_tensor);      // 0 = q, 1 = k
    long long rest = p.only ? pair : pair - which * pairs_per_tensor;
    const int head = (int)(rest % p.heads); rest /= p.heads;
    const int s = (int)(rest % p.S);
    const int z = (int)(rest / p.S);
    const int d0 = (lane & 7) * 8;
    bf16_t* base = (which ? p.k : p.q) + z * p.bs + (long long)s * p.ld + head * 64 + d0;
    const bf16_t* w = (which ? p.kw : p.qw) + d0;
    const bf16_t* b = (which ? p.kb : p.qb) + d0;

    // (arithmetic in qknorm_math.h, shared with the QKV GEMM's fused epilogue: the two agree bit for bit)
    float v[8];
    unpack8(*reinterpret_cast<const u32x4*>(base), v);
    float sum = qkn_sum8(v);
    sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64); sum += __shfl_xor(sum, 4, 64);
    const float mean = sum * (1.0f / 64);
    float sq = qkn_centre_sq8(v, mean);
    sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
    const float rstd = rsqrtf(sq * (1.0f / 64) + p.eps);
    float wv[8], bv[8];
    unpack8(*reinterpret_cast<const u32x4*>(w), wv);
    unpack8(*reinterpret_cast<const u32x4*>(b), bv);
    float cc[8] = {}, ss[8] = {};
    const bool rope = s >= p.text_rows;
    if (rope) {
        const float* c = p.cos + (long long)(s - p.text_rows) * 64 + d0;
        const float* sn = p.sin + (long long)(s - p.text_rows) * 64 + d0;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(c), c1 = *reinterpret_cast<const f32x4*>(c + 4);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sn), s1 = *reinterpret_cast<const f32x4*>(sn + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { cc[e] = c0[e]; cc[4 + e] = c1[e]; ss[e] = s0[e]; ss[4 + e] = s1[e]; }
    }
    qkn_finish8(v, rstd, wv, bv, rope, cc, ss, which ? p.k_scale : 1.0f);
    const u32x4 out = pack8(v);
    *reinterpret_cast<u32x4*>(base) = out;
    if constexpr (STATS) {
        float r8[8];
        unpack8(out, r8);
        float n2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) n2 += r8[e] * r8[e];
        n2 += __shfl_xor(n2, 1, 64); n2 += __shfl_xor(n2, 2, 64); n2 += __shfl_xor(n2, 4, 64);
        if ((lane & 7) == 0) {
            const long long nbh = (long long)p.batch * p.heads;
            unsigned* dst = reinterpret_cast<unsigned*>(p.stats) + ((long long)(blockIdx.x % (unsigned)p.stats_slots) * 2 + which) * nbh +
                            (long long)z * p.heads + head;
            (void)__hip_atomic_fetch_max(dst, __float_as_uint(n2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

extern "C" int bya_layernorm(const void* x, void* y, const void* w, const void* b, const void* shift0,
                             const void* scale0, const void* shift1, const void* scale1, int64_t rows_per_batch,
                             int32_t batch, int32_t D, int64_t ldx, int64_t ldy, int64_t x_batch_stride,
                             int64_t y_batch_stride, int64_t mod_batch_stride, int64_t split, float eps,
                             hipStream_t stream) {
    if (!x || !y || rows_per_batch <= 0 || batch <= 0) return BYA_ERR_SHAPE;
    if ((shift0 == nullptr) != (scale0 == nullptr)) return BYA_ERR_SHAPE;
    if (((uintptr_t)x | (uintptr_t)y) & 15) return BYA_ERR_ALIGN;
    if ((ldx | ldy | x_batch_stride | y_batch_stride) % 8) return BYA_ERR_ALIGN;
    LnArgs a;
    a.x = (const bf16_t*)x; a.y = (bf16_t*)y; a.w = (const bf16_t*)w; a.b = (const bf16_t*)b;
    a.shift0 = (const bf16_t*)shift0; a.scale0 = (const bf16_t*)scale0;
    a.shift1 = (const bf16_t*)(shift1 ? shift1 : shift0); a.scale1 = (const bf16_t*)(scale1 ? scale1 : scale0);
    a.rows_per_batch = rows_per_batch; a.ldx = ldx; a.ldy = ldy; a.x_bs = x_batch_stride; a.y_bs = y_batch_stride;
    a.mod_bs = mod_batch_stride; a.split = split; a.batch = batch; a.eps = eps; a.q_scale = nullptr;
    switch (D) {
        case 512: return launch_ln<8, 1>(a, stream);
        case 768: return launch_ln<4, 3>(a, stream);
        case 1024: return launch_ln<8, 2>(a, stream);
        case 2048: return launch_ln<8, 4>(a, stream);
        case 3072: {
            // (whatever the row count: a shard of the sequence must round exactly like the whole)
            if (a.w && a.b && !bya_ref_form(BYA_REF_LN_GENERIC)) return launch_ln_adaln_rows<6>(a, stream);
            return launch_ln<8, 6>(a, stream);
        }
        default: return BYA_ERR_UNSUPPORTED;
    }
}

extern "C" int bya_layernorm_fp8(const void* x, void* q, float* q_scale, const void* w, const void* b, const void* shift0,
                                 const void* scale0, const void* shift1, const void* scale1, int64_t rows_per_batch,
                                 int32_t batch, int32_t D, int64_t ldx, int64_t ldq, int64_t x_batch_stride,
                                 int64_t q_batch_stride, int64_t mod_batch_stride, int64_t split, float eps,
                                 hipStream_t stream) {
    if (!x || !q || !q_scale || rows_per_batch <= 0 || batch <= 0) return BYA_ERR_SHAPE;
    if ((shift0 == nullptr) != (scale0 == nullptr)) return BYA_ERR_SHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)q & 7)) return BYA_ERR_ALIGN;
    if ((ldx | ldq | x_batch_stride | q_batch_stride) % 8) return BYA_ERR_ALIGN;
    LnArgs a;
    a.x = (const bf16_t*)x; a.y = (bf16_t*)q; a.w = (const bf16_t*)w; a.b = (const bf16_t*)b;
    a.shift0 = (const bf16_t*)shift0; a.scale0 = (const bf16_t*)scale0;
    a.shift1 = (const bf16_t*)(shift1 ? shift1 : shift0); a.scale1 = (const bf16_t*)(scale1 ? scale1 : scale0);
    a.rows_per_batch = rows_per_batch; a.ldx = ldx; a.ldy = ldq; a.x_bs = x_batch_stride; a.y_bs = q_batch_stride;
    a.mod_bs = mod_batch_stride; a.split = split; a.batch = batch; a.eps = eps; a.q_scale = q_scale;
    if (D != 3072) return BYA_ERR_UNSUPPORTED;          // the DiT width: the only LayerNorm in front of an fp8 Linear
    return launch_ln<8, 6, true>(a, stream);
}

extern "C" int bya_qknorm_rope(void* q, void* k, const void* qw, const void* qb, const void* kw, const void* kb,
                               const float* cos, const float* sin, int32_t batch, int32_t S, int32_t heads,
                               int64_t ld, int64_t batch_stride, int32_t text_rows, float eps, float k_scale,
                               float* stats, int32_t stats_slots, hipStream_t stream) {
    if ((!q && !k) || !qw || !qb || !kw || !kb || batch <= 0 || S <= 0 || heads <= 0) return BYA_ERR_SHAPE;
    if (text_rows < S && (!cos || !sin)) return BYA_ERR_SHAPE;
    if (stats && (stats_slots < 1 || stats_slots > 64)) return BYA_ERR_SHAPE;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)cos | (uintptr_t)sin | (uintptr_t)qw | (uintptr_t)kw |
         (uintptr_t)qb | (uintptr_t)kb) & 15) return BYA_ERR_ALIGN;
    if ((ld | batch_stride) % 8 || ((uintptr_t)stats & 3)) return BYA_ERR_ALIGN;
    QkArgs a;
    a.q = (bf16_t*)q; a.k = (bf16_t*)k; a.qw = (const bf16_t*)qw; a.qb = (const bf16_t*)qb;
    a.kw = (const bf16_t*)kw; a.kb = (const bf16_t*)kb; a.cos = cos; a.sin = sin;
    a.batch = batch; a.S = S; a.heads = heads; a.text_rows = text_rows; a.ld = ld; a.bs = batch_stride; a.eps = eps; a.k_scale = k_scale == 0.0f ? 1.0f : k_scale;
    a.only = !k ? 1 : !q ? 2 : 0;                                             // one tensor alone (the other pointer is NULL)
    a.stats = stats; a.stats_slots = stats_slots;
    const long long total = ((long long)batch * S * heads * (a.only ? 1 : 2) + 7) / 8;      // waves: 8 (row, head) pairs each
    dim3 grid((unsigned)((total + 3) / 4));
    if (stats) BYA_LAUNCH(qknorm_rope_kernel<true>, grid, dim3(256), 0, stream, a);
    else BYA_LAUNCH(qknorm_rope_kernel<false>, grid, dim3(256), 0, stream, a);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}
