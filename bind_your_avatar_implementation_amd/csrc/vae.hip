// Video VAE either side of the denoise loop: the data-movement and normalisation kernels
// of AutoencoderKLCogVideoX's decoder / encoder (diffusers, called at models/pipeline_bindyouravatar.py:406-424 and
// :461-466).  Activations are CHANNELS-LAST bf16 [T, H, W, C]: a 1x1x1 convolution is a plain GEMM on them, a causal
// 3x3x3 (or per-frame 3x3) convolution is  patches[rows, taps * C] x W[C_out, taps * C]^T  on bya_gemm_bf16 with the bias
// and the residual in its epilogue; this file gathers the patches, computes the GroupNorm statistics and applies
// GroupNorm (+ the decoder's spatial modulation) + SiLU in one pass.  All HBM-bound; no MFMA here (built without packed
// fp32, like the other VALU units).
//
// bya_vae_patches: rows = output positions (t, h, w) of the frames [t0, t0 + nt) of a chunk; column (kt, kh, kw, c).
//   * time is CAUSAL: tap kt of output frame t reads frame t + kt - (KT - 1) of the chunk; negative indices fall into
//     the cache of the previous chunk (its last KT - 1 frames) or, for the first chunk, onto frame 0 (diffusers
//     CogVideoXCausalConv3d, pad_mode "first");
//   * space: tap (kh, kw) of output (h, w) reads (h * stride + kh - pad, w * stride + kw - pad) of the SOURCE GRID, zero
//     outside (pad = 1: the 3x3x3 and up-sampler convolutions; pad = 0, stride 2: the down-sampler, whose (0, 1, 0, 1)
//     zero pad is the out-of-range case at the far edge);
//   * the source grid may be a nearest-neighbour up-sampling of the stored tensor (up = 1: h >> 1, w >> 1; time by
//     tmode: 0 same frame, 1 every frame doubled, 2 first frame single and the rest doubled) -- CogVideoXUpsample3D's
//     F.interpolate is never materialised.
// bya_vae_groupnorm_stats: per (group) sum and sum of squares over a chunk's rows -> [groups][2] fp32, in a fixed order.
// bya_vae_norm_act: y = act( (x - mean_g) rstd_g gamma_c + beta_c ) with, for the decoder,
//   ... * Y[z(row)][c] + B[z(row)][c]  (CogVideoXSpatialNorm3D: conv_y / conv_b of the latent are 1x1x1, hence commute
//   with its nearest resize: they are evaluated at latent resolution by a GEMM and indexed here).
//
// bya_vae_conv3d (round 3): the 3 x 3 x 3 stride-1 convolutions of the resnet blocks (C = 128 / 256 / 512: > 95 % of the
//   decoder's FLOPs) WITHOUT a patch matrix -- an implicit GEMM on the persistent MFMA kernel (gemm_v4.hip, CONV instance).
//   The producer (bya_vae_norm_act, out_pad = 1) writes the conv input zero-padded, [To + 2, H + 2, W + 2, C]: two context
//   frames in front (the conv cache or the first frame again), one pixel of zeros around every frame.  With the output grid
//   padded the same way, tap (dt, dh, dw) of output pixel m is input pixel m + (dt Hp + dh) Wp + dw: every K-tile of the GEMM
//   is the SAME 256 rows of the pixel matrix at a scalar offset, which is exactly what the kernel's LDS-DMA staging takes
//   (per-lane row offsets + one scalar per K-tile).  Rows that are padding are computed and dropped by the epilogue (0.7 %
//   of the rows at 480 x 720, 5 % at 60 x 90).  The patch path wrote and re-read 27 x the activation (117 GB per convolution
//   at the top level: HBM-bound at ~500 TFLOP/s by construction); this one reads it from the L2.
#include "bya_common.h"
#include "gemm_common.h"
#include "../../include/bya.h"

int bya_launch_conv256p(const void* args, hipStream_t s);      // gemm_v4.hip

namespace {

struct PatchArgs {
    const bf16_t* x;        // stored tensor [Ts, Hs, Ws, C]
    const bf16_t* cache;    // [KT - 1, Hs, Ws, C] of the previous chunk or null
    bf16_t* out;            // [nt * Ho * Wo, Kpad]
    int Ts, Hs, Ws, C, KT, stride, pad, up, tmode, Ho, Wo, t0, nt, Kpad;
};

__device__ __forceinline__ int src_frame(int tau, int tmode) {       // frame of the up-sampled grid -> stored frame
    return tmode == 0 ? tau : tmode == 1 ? (tau >> 1) : (tau == 0 ? 0 : 1 + ((tau - 1) >> 1));
}

// one thread = one 16-byte piece (8 channels) of one (row, tap); C % 8 == 0
__global__ __launch_bounds__(256) void vae_patches_kernel(PatchArgs p) {
    const int cpt = p.C >> 3;                              // pieces per tap
    const int taps = p.KT * 9;
    const int ppr = p.Kpad >> 3;                           // pieces per row (incl. zero padding behind the last tap)
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)p.nt * p.Ho * p.Wo * ppr;
    if (idx >= total) return;
    const int piece = (int)(idx % ppr);
    long long row = idx / ppr;
    const int w = (int)(row % p.Wo); row /= p.Wo;
    const int h = (int)(row % p.Ho);
    const int t = (int)(row / p.Ho) + p.t0;
    u32x4 v = {0u, 0u, 0u, 0u};
    const int tap = piece / cpt, c8 = piece - tap * cpt;
    if (tap < taps) {
        const int kt = tap / 9, kh = (tap % 9) / 3, kw = tap % 3;
        const int hs = h * p.stride + kh - p.pad, ws = w * p.stride + kw - p.pad;     // on the (maybe up-sampled) grid
        const int Hg = p.up ? p.Hs * 2 : p.Hs, Wg = p.up ? p.Ws * 2 : p.Ws;
        if (hs >= 0 && hs < Hg && ws >= 0 && ws < Wg) {
            const int hh = p.up ? hs >> 1 : hs, ww = p.up ? ws >> 1 : ws;
            int tau = t + kt - (p.KT - 1);
            const bf16_t* base;
            if (tau >= 0) {
                base = p.x + (((long long)src_frame(tau, p.tmode) * p.Hs + hh) * p.Ws + ww) * p.C;
            } else if (p.cache) {
                base = p.cache + (((long long)(tau + p.KT - 1) * p.Hs + hh) * p.Ws + ww) * p.C;
            } else {
                base = p.x + ((long long)hh * p.Ws + ww) * p.C;                   // first chunk: its first frame again
            }
            v = *reinterpret_cast<const u32x4*>(base + c8 * 8);
        }
    }
    *reinterpret_cast<u32x4*>(p.out + (idx << 3)) = v;
}

// C < 8 (the encoder's RGB input): one thread per (row, tap), scalar copies
__global__ __launch_bounds__(256) void vae_patches_small_kernel(PatchArgs p) {
    const int taps = p.KT * 9;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long rows = (long long)p.nt * p.Ho * p.Wo;
    const int slots = (p.Kpad + p.C - 1) / p.C;           // taps + zero slots up to Kpad
    if (idx >= rows * slots) return;
    const int tap = (int)(idx % slots);
    long long row = idx / slots;
    bf16_t* dst = p.out + row * p.Kpad + (long long)tap * p.C;
    const int n = (tap * p.C + p.C <= p.Kpad) ? p.C : p.Kpad - tap * p.C;
    const int w = (int)(row % p.Wo); row /= p.Wo;
    const int h = (int)(row % p.Ho);
    const int t = (int)(row / p.Ho) + p.t0;
    const bf16_t* base = nullptr;
    if (tap < taps) {
        const int kt = tap / 9, kh = (tap % 9) / 3, kw = tap % 3;
        const int hs = h * p.stride + kh - p.pad, ws = w * p.stride + kw - p.pad;
        if (hs >= 0 && hs < p.Hs && ws >= 0 && ws < p.Ws) {
            const int tau = t + kt - (p.KT - 1);
            if (tau >= 0) base = p.x + (((long long)tau * p.Hs + hs) * p.Ws + ws) * p.C;
            else if (p.cache) base = p.cache + (((long long)(tau + p.KT - 1) * p.Hs + hs) * p.Ws + ws) * p.C;
            else base = p.x + ((long long)hs * p.Ws + ws) * p.C;
        }
    }
    for (int c = 0; c < n; ++c) dst[c] = base ? base[c] : (bf16_t)0;
}

// ---- GroupNorm statistics, DETERMINISTIC (no atomics: the same chunk gives the same bits on every run, which the causal
// chunk cache of the decoder makes observable -- frames of one chunk must not change when later latents do).
// Pass 1: a block sums 512 rows; thread t owns the 8 channels (t % cpr) * 8 .. of its rows (256 % cpr == 0: C / 8 is a power of
// two <= 64), accumulates them in registers in row order, then the block adds the threads of a channel in thread order and the
// channels of a group in channel order -> partial[block][group][2].  Pass 2: one block per (group, moment) adds the partials in
// a fixed order (strided per thread, then a shared-memory tree).
constexpr int GN_ROWS = 512;

__global__ __launch_bounds__(256) void vae_gn_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ partial, long long rows,
                                                             int C, int groups) {
    __shared__ float sh[2][256][8];
    __shared__ float chs[2][512];
    const int cpr = C >> 3, tid = threadIdx.x;
    const int c8 = tid % cpr, rstep = 256 / cpr;
    const long long r0 = (long long)blockIdx.x * GN_ROWS;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    for (int rr = tid / cpr; rr < GN_ROWS; rr += rstep) {
        const long long r = r0 + rr;
        if (r >= rows) break;
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + r * C + c8 * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += v[e]; q[e] = fmaf(v[e], v[e], q[e]); }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { sh[0][tid][e] = s[e]; sh[1][tid][e] = q[e]; }
    __syncthreads();
    for (int ch = tid; ch < C; ch += 256) {                 // channel sums: the threads of a channel in thread order
        const int cc = ch >> 3, e = ch & 7;
        float a = 0.f, b = 0.f;
        for (int t = cc; t < 256; t += cpr) { a += sh[0][t][e]; b += sh[1][t][e]; }
        chs[0][ch] = a; chs[1][ch] = b;
    }
    __syncthreads();
    const int cg = C / groups;
    for (int g = tid; g < groups; g += 256) {
        float a = 0.f, b = 0.f;
        for (int c = g * cg; c < (g + 1) * cg; ++c) { a += chs[0][c]; b += chs[1][c]; }
        partial[((long long)blockIdx.x * groups + g) * 2] = a;
        partial[((long long)blockIdx.x * groups + g) * 2 + 1] = b;
    }
}

__global__ __launch_bounds__(256) void vae_gn_final_kernel(const float* __restrict__ partial, float* __restrict__ sums, int nblocks,
                                                           int groups) {
    __shared__ float red[256];
    const int gm = blockIdx.x, tid = threadIdx.x;            // gm = group * 2 + moment
    float a = 0.f;
    for (int b = tid; b < nblocks; b += 256) a += partial[(long long)b * groups * 2 + gm];
    red[tid] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) sums[gm] = red[0];
}

struct NormArgs {
    const bf16_t* x; bf16_t* y; const float* sums; const bf16_t* gamma; const bf16_t* beta;
    const bf16_t* zy; const bf16_t* zb;      // [Tz * hz * wz, C] each (decoder) or null
    long long rows;
    int C, groups, act;
    float count, eps;                        // elements per group of the chunk
    int T, H, W, Tz, hz, wz, shift, tmode;   // row -> (t, h, w) -> latent position (h >> shift, w >> shift, frame by tmode)
    int ldz;                                 // row stride of zy / zb (they may be the two halves of one GEMM output)
    int out_pad;                             // 1: y is the zero-padded conv input [T + 2, H + 2, W + 2, C] (bya_vae_conv3d)
    int cpr_shift;                           // log2(C / 8) when that is a power of two, else -1
};

// One thread = one 16-byte piece (8 channels) of one row.  32-bit index arithmetic (a chunk has < 2^31 pieces: checked by the
// launcher), the channel-piece count a power of two: the 64-bit divisions of the first version cost more than its HBM bytes.
__global__ __launch_bounds__(256) void vae_norm_act_kernel(NormArgs p) {
    const uint32_t cpr = (uint32_t)p.C >> 3;
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= (uint32_t)p.rows * cpr) return;
    const uint32_t r = p.cpr_shift >= 0 ? idx >> p.cpr_shift : idx / cpr;
    const int c0 = (int)(idx - r * cpr) * 8;
    float v[8], gm[8], bt[8];
    unpack8(*reinterpret_cast<const u32x4*>(p.x + (long long)r * p.C + c0), v);
    unpack8(*reinterpret_cast<const u32x4*>(p.gamma + c0), gm);
    unpack8(*reinterpret_cast<const u32x4*>(p.beta + c0), bt);
    const int cg = p.C / p.groups;
    float zy[8], zb[8];
    uint32_t t = 0, h = 0, w = 0;
    if (p.zy || p.out_pad) {
        const uint32_t rr = r / (uint32_t)p.W;
        w = r - rr * (uint32_t)p.W;
        t = rr / (uint32_t)p.H;
        h = rr - t * (uint32_t)p.H;
    }
    if (p.zy) {
        // latent frame of frame t: tmode 0: same count; 1: t * Tz / T (even resize); 2: first frame apart
        int tz;
        if (p.tmode == 0) tz = (int)t;
        else if (p.tmode == 1) tz = (int)((t * (uint32_t)p.Tz) / (uint32_t)p.T);
        else tz = t == 0 ? 0 : 1 + (int)(((t - 1) * (uint32_t)(p.Tz - 1)) / (uint32_t)(p.T - 1));
        const long long zr = ((long long)tz * p.hz + (h >> p.shift)) * p.wz + (w >> p.shift);
        unpack8(*reinterpret_cast<const u32x4*>(p.zy + zr * p.ldz + c0), zy);
        unpack8(*reinterpret_cast<const u32x4*>(p.zb + zr * p.ldz + c0), zb);
    }
    // the group of these 8 channels (a piece never straddles two groups when the group size is a multiple of 8)
    float o[8];
    if ((cg & 7) == 0) {
        const int g = c0 / cg;
        const float mean = p.sums[2 * g] / p.count;
        const float var = fmaxf(p.sums[2 * g + 1] / p.count - mean * mean, 0.f);
        const float rstd = rsqrtf(var + p.eps);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = (v[e] - mean) * rstd * gm[e] + bt[e];
            if (p.zy) y = y * zy[e] + zb[e];
            if (p.act == 1) y = y / (1.0f + __expf(-y));      // SiLU
            o[e] = y;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (c0 + e) / cg;
            const float mean = p.sums[2 * g] / p.count;
            const float var = fmaxf(p.sums[2 * g + 1] / p.count - mean * mean, 0.f);
            const float rstd = rsqrtf(var + p.eps);
            float y = (v[e] - mean) * rstd * gm[e] + bt[e];
            if (p.zy) y = y * zy[e] + zb[e];
            if (p.act == 1) y = y / (1.0f + __expf(-y));      // SiLU
            o[e] = y;
        }
    }
    long long ro = r;
    if (p.out_pad) ro = ((long long)(t + 2) * (p.H + 2) + h + 1) * (p.W + 2) + w + 1;
    *reinterpret_cast<u32x4*>(p.y + ro * p.C + c0) = pack8(o);
}

// Nearest-neighbour up-sampling (space x 2; time by tmode, see bya_vae_patches) of x [T, H, W, C] INTO the interior of the
// zero-padded conv input [To, 2 H + 2, 2 W + 2, C] of the up-sampler's per-frame 3 x 3 convolution (bya_vae_conv3d, KT = 1):
// one thread per 16-byte piece of an output pixel.  4 x (8 x) the stored tensor instead of 9 x that as a patch matrix.
__global__ __launch_bounds__(256) void vae_upsample_pad_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int To, int H,
                                                               int W, int C, int tmode) {
    const int cpr = C >> 3, W2 = 2 * W, H2 = 2 * H;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)To * H2 * W2 * cpr) return;
    const int c8 = (int)(idx % cpr);
    long long r = idx / cpr;
    const int w = (int)(r % W2); r /= W2;
    const int h = (int)(r % H2);
    const int t = (int)(r / H2);
    const u32x4 v = *reinterpret_cast<const u32x4*>(x + (((long long)src_frame(t, tmode) * H + (h >> 1)) * W + (w >> 1)) * C + c8 * 8);
    *reinterpret_cast<u32x4*>(y + (((long long)t * (H2 + 2) + h + 1) * (W2 + 2) + w + 1) * C + c8 * 8) = v;
}

}  // namespace

extern "C" int bya_vae_patches(const void* x, const void* cache, void* out, int32_t Ts, int32_t Hs, int32_t Ws, int32_t C,
                               int32_t KT, int32_t stride, int32_t pad, int32_t up, int32_t tmode, int32_t Ho, int32_t Wo,
                               int32_t t0, int32_t nt, int32_t Kpad, hipStream_t stream) {
    if (!x || !out || Ts <= 0 || Hs <= 0 || Ws <= 0 || C <= 0 || nt <= 0 || Ho <= 0 || Wo <= 0) return BYA_ERR_SHAPE;
    if ((KT != 1 && KT != 3) || (stride != 1 && stride != 2) || (pad != 0 && pad != 1) || tmode < 0 || tmode > 2) return BYA_ERR_SHAPE;
    if (Kpad < KT * 9 * C || Kpad % 8) return BYA_ERR_SHAPE;
    if (up && stride != 1) return BYA_ERR_UNSUPPORTED;
    PatchArgs p{(const bf16_t*)x, (const bf16_t*)cache, (bf16_t*)out, Ts, Hs, Ws, C, KT, stride, pad, up, tmode, Ho, Wo, t0, nt, Kpad};
    const long long rows = (long long)nt * Ho * Wo;
    if (C % 8 == 0) {
        if (((uintptr_t)x | (uintptr_t)cache | (uintptr_t)out) & 15) return BYA_ERR_ALIGN;
        const long long total = rows * (Kpad >> 3);
        BYA_LAUNCH(vae_patches_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
    } else {
        if (up) return BYA_ERR_UNSUPPORTED;
        const long long total = rows * ((Kpad + C - 1) / C);
        BYA_LAUNCH(vae_patches_small_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
    }
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

extern "C" int bya_vae_groupnorm_stats(const void* x, float* sums, float* partial, int64_t rows, int32_t C, int32_t groups,
                                       hipStream_t stream) {
    if (!x || !sums || !partial || rows <= 0 || C <= 0 || groups <= 0 || C % groups || C % 8) return BYA_ERR_SHAPE;
    const int cpr = C / 8;
    if (C > 512 || (cpr & (cpr - 1)) || cpr > 64) return BYA_ERR_UNSUPPORTED;      // C / 8 a power of two <= 64
    if ((uintptr_t)x & 15) return BYA_ERR_ALIGN;
    const long long blocks = (rows + GN_ROWS - 1) / GN_ROWS;
    BYA_LAUNCH(vae_gn_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16_t*)x, partial, (long long)rows, C, groups);
    BYA_LAUNCH(vae_gn_final_kernel, dim3((unsigned)(groups * 2)), dim3(256), 0, stream, (const float*)partial, sums, (int)blocks, groups);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

extern "C" int bya_vae_norm_act(const void* x, void* y, const float* sums, const void* gamma, const void* beta, const void* zy,
                                const void* zb, int64_t rows, int32_t C, int32_t groups, int32_t act, float eps, int32_t T,
                                int32_t H, int32_t W, int32_t Tz, int32_t hz, int32_t wz, int32_t tmode, int64_t ldz,
                                int32_t out_pad, hipStream_t stream) {
    if (!x || !y || !sums || !gamma || !beta || rows <= 0 || C <= 0 || groups <= 0 || C % groups || C % 8) return BYA_ERR_SHAPE;
    if ((zy == nullptr) != (zb == nullptr)) return BYA_ERR_SHAPE;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)zy | (uintptr_t)zb) & 15) return BYA_ERR_ALIGN;
    NormArgs p{};
    p.x = (const bf16_t*)x; p.y = (bf16_t*)y; p.sums = sums; p.gamma = (const bf16_t*)gamma; p.beta = (const bf16_t*)beta;
    p.zy = (const bf16_t*)zy; p.zb = (const bf16_t*)zb; p.rows = rows; p.C = C; p.groups = groups; p.act = act;
    p.count = (float)((double)rows * (C / groups)); p.eps = eps;
    p.T = T; p.H = H; p.W = W; p.Tz = Tz; p.hz = hz; p.wz = wz; p.tmode = tmode; p.ldz = (int)ldz;
    p.out_pad = out_pad ? 1 : 0;
    if (out_pad && (T <= 0 || H <= 0 || W <= 0 || (long long)T * H * W != rows)) return BYA_ERR_SHAPE;
    if (zy) {
        if ((long long)T * H * W != rows || hz <= 0 || wz <= 0 || H % hz || W % wz || H / hz != W / wz || ldz % 8) return BYA_ERR_SHAPE;
        int sh = 0;
        while ((hz << sh) < H) ++sh;
        if ((hz << sh) != H) return BYA_ERR_SHAPE;
        p.shift = sh;
        if (tmode == 0 && Tz != T) return BYA_ERR_SHAPE;
        if (tmode == 2 && (T < 2 || Tz < 2)) return BYA_ERR_SHAPE;
    }
    const long long total = rows * (C >> 3);
    if (total >= (1LL << 31) - 256) return BYA_ERR_SHAPE;                      // 32-bit piece index (a chunk of frames, not a clip)
    p.cpr_shift = -1;
    for (int sft = 0; sft < 16; ++sft)
        if ((C >> 3) == (1 << sft)) p.cpr_shift = sft;
    BYA_LAUNCH(vae_norm_act_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

extern "C" int bya_vae_upsample_pad(const void* x, void* ypad, int32_t T, int32_t H, int32_t W, int32_t C, int32_t tmode,
                                    hipStream_t stream) {
    if (!x || !ypad || T <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || tmode < 0 || tmode > 2) return BYA_ERR_SHAPE;
    if (((uintptr_t)x | (uintptr_t)ypad) & 15) return BYA_ERR_ALIGN;
    const int To = tmode == 0 ? T : tmode == 1 ? 2 * T : 2 * T - 1;
    const long long total = (long long)To * 4 * H * W * (C >> 3);
    BYA_LAUNCH(vae_upsample_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)ypad,
               To, H, W, C, tmode);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

extern "C" int bya_vae_conv3d(const void* xpad, const void* w, const void* bias, const void* res, void* out, int32_t To,
                              int32_t H, int32_t W, int32_t C, int32_t Cout, int32_t KT, int64_t ldw, int64_t ldc,
                              int64_t ldres, hipStream_t stream) {
    if (!xpad || !w || !out || To <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (KT != 1 && KT != 3)) return BYA_ERR_SHAPE;
    if (C != 128 && C != 256 && C != 512) return BYA_ERR_UNSUPPORTED;           // whole 64-channel groups, a power of two of them
    if (Cout % 8 || ldc < Cout || ldc % 8 || (res && (ldres < Cout || ldres % 8)) || ldw < 9LL * KT * C || ldw % 8) return BYA_ERR_ALIGN;
    if (((uintptr_t)xpad | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)res | (uintptr_t)out) & 15) return BYA_ERR_ALIGN;
    const long long Hp = H + 2, Wp = W + 2, M = (long long)To * Hp * Wp;
    // 32-bit reach of the epilogue's offsets and of a tile's LDS-DMA offsets (two frames + two rows ahead of its rows)
    if (M >= (1LL << 31) || (long long)To * H * W * ldc * 2 >= (1LL << 31) || (res && (long long)To * H * W * ldres * 2 >= (1LL << 31)) ||
        ((2 * Hp + 2) * Wp + 258) * C * 2 >= (1LL << 31))
        return BYA_ERR_SHAPE;
    GemmArgs a{};
    a.A = (const bf16_t*)xpad; a.W = (const bf16_t*)w; a.bias = (const bf16_t*)bias; a.C = (bf16_t*)out; a.res = (const bf16_t*)res;
    a.gate0 = nullptr; a.gate1 = nullptr;
    a.M = (int)M; a.N = Cout; a.K = 9 * KT * C;
    a.lda = C; a.ldw = (int)ldw; a.ldc = (int)ldc; a.ldres = (int)(res ? ldres : ldc);
    a.a_bs = 0; a.c_bs = 0; a.res_bs = 0; a.gate_bs = 0; a.gate_split = 0; a.act = BYA_ACT_NONE; a.leaky = 0.01f;
    a.n_split = 0; a.c_split_stride = 0; a.bias_rowscale = nullptr; a.alpha = 1.0f;
    a.ws_slabs = nullptr; a.ws_counters = nullptr;
    a.conv_cpg_log2 = C == 128 ? 1 : C == 256 ? 2 : 3;
    a.conv_Hp = (int)Hp; a.conv_Wp = (int)Wp; a.conv_H = H; a.conv_W = W; a.conv_To = To;
    a.conv_a_bytes = (long long)(To + KT - 1) * Hp * Wp * C * 2;
    return bya_launch_conv256p(&a, stream);
}
