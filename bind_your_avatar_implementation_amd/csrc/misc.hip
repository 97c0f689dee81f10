// Small HBM-bound / index-only kernels of the denoise step: small-M linears (timestep embedding and all
// AdaLN modulation vectors), sinusoidal timestep features, the router's masked combines (G1/G2), the
// forcing max-over-frames, patchify / unpatchify and a generic activation(+add).
#include "bya_common.h"
#include "../../include/bya.h"
#include "options.h"
#include "routing_weights.h"

namespace {

// ------------------------------------------------------------------------------------------------
// out[m,n] = act( sum_k f(x[m,k]) * W[n,k] + bias[n] ), M <= 8.  One wave per output column n; the K axis is
// split over the 64 lanes in 16-byte pieces (W is streamed exactly once: HBM-bound).
template <int MAXM>
__global__ __launch_bounds__(256) void linear_small_m_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W,
                                                             const bf16_t* __restrict__ bias, bf16_t* __restrict__ out,
                                                             int M, int N, int K, int silu_in, int act_out) {
    const int lane = threadIdx.x & 63;
    const long long n = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) acc[m] = 0.f;
    const bf16_t* wrow = W + n * K;
    for (int k0 = lane * 8; k0 < K; k0 += 512) {
        float wv[8];
        unpack8(*reinterpret_cast<const u32x4*>(wrow + k0), wv);
#pragma unroll
        for (int m = 0; m < MAXM; ++m) {
            if (m < M) {
                float xv[8];
                unpack8(*reinterpret_cast<const u32x4*>(x + (long long)m * K + k0), xv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float xe = xv[e];
                    if (silu_in) xe = bf2f(f2bf(silu(xe)));  // SiLU output is a bf16 tensor in the reference
                    acc[m] += xe * wv[e];
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
        if (m < M) {
            float v = wave_sum(acc[m]);
            if (lane == 0) {
                if (bias) v += bf2f(bias[n]);
                if (act_out == BYA_ACT_SILU) v = silu(v);
                out[(long long)m * N + n] = f2bf(v);
            }
        }
    }
}

__global__ void timestep_features_kernel(const int64_t* __restrict__ t, bf16_t* __restrict__ out, int batch, int dim,
                                         int flip, float shift) {
    const int half = dim / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= batch * half) return;
    const int b = idx / half, j = idx % half;
    const float expo = -logf(10000.0f) * (float)j / ((float)half - shift);
    const float ang = (float)t[b] * expf(expo);
    const float s = sinf(ang), c = cosf(ang);
    bf16_t* o = out + (long long)b * dim;
    if (flip) { o[j] = f2bf(c); o[half + j] = f2bf(s); }
    else { o[j] = f2bf(s); o[half + j] = f2bf(c); }
}

// ------------------------------------------------------------------------------------------------
// Masked combine (G1 face / G2 audio).  One thread per 8 channels of one token.
struct CombArgs {
    bf16_t* x; const bf16_t* feat; const bf16_t* r; const bf16_t* af;
    int mode, batch, n_id, D;
    long long N, x_row, x_bs, r_bs;
    float alpha;
};

__device__ __forceinline__ void routing_weights(const CombArgs& p, const bf16_t* r, int b, float (&w)[4]) {
    routing_weights_of(p.mode, p.n_id, p.af ? p.af + b * p.n_id * p.n_id : nullptr, r, w);
}

__global__ __launch_bounds__(256) void masked_combine_kernel(CombArgs p) {
    const int vec_per_row = p.D / 8;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)p.batch * p.N * vec_per_row;
    if (gid >= total) return;
    const int c8 = (int)(gid % vec_per_row);
    const long long bn = gid / vec_per_row;
    const long long n = bn % p.N;
    const int b = (int)(bn / p.N);
    const bf16_t* r = p.r + b * p.r_bs + n * p.n_id;
    float w[4];
    routing_weights(p, r, b, w);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < p.n_id) {
            float f[8];
            unpack8(*reinterpret_cast<const u32x4*>(p.feat + (((long long)b * p.n_id + i) * p.N + n) * p.D + c8 * 8), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(w[i], f[e], acc[e]);
        }
    }
    bf16_t* xp = p.x + b * p.x_bs + n * p.x_row + c8 * 8;
    float xv[8];
    unpack8(*reinterpret_cast<const u32x4*>(xp), xv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float t = bf2f(f2bf(acc[e]));                 // bmm output is bf16
        if (p.alpha != 1.0f) t = bf2f(f2bf(p.alpha * t));
        xv[e] += t;
    }
    *reinterpret_cast<u32x4*>(xp) = pack8(xv);
}


// ------------------------------------------------------------------------------------------------
// Routed sum BEFORE the output projection (linearity of to_out):
//   sum_id w[n,id] * (o[id,n,:] @ W^T + b)  ==  (sum_id w[n,id] * o[id,n,:]) @ W^T + (sum_id w[n,id]) * b
// so the engine mixes the two per-identity attention outputs first and runs ONE half-size GEMM whose epilogue adds
// rowscale[n] * bias and the residual.  Weights are derived exactly as in masked_combine (mode 0 face / 1 audio).
__global__ __launch_bounds__(256) void routed_mix_kernel(CombArgs p, bf16_t* __restrict__ z, float* __restrict__ wsum) {
    const int vec_per_row = p.D / 8;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)p.batch * p.N * vec_per_row;
    if (gid >= total) return;
    const int c8 = (int)(gid % vec_per_row);
    const long long bn = gid / vec_per_row;
    const long long n = bn % p.N;
    const int b = (int)(bn / p.N);
    const bf16_t* r = p.r + b * p.r_bs + n * p.n_id;
    float w[4];
    routing_weights(p, r, b, w);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    float ws = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < p.n_id) {
            float f[8];
            unpack8(*reinterpret_cast<const u32x4*>(p.feat + (((long long)b * p.n_id + i) * p.N + n) * p.D + c8 * 8), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(w[i], f[e], acc[e]);
            ws += w[i];
        }
    }
    *reinterpret_cast<u32x4*>(z + ((long long)b * p.N + n) * p.D + c8 * 8) = pack8(acc);
    if (c8 == 0 && wsum) wsum[(long long)b * p.N + n] = ws;
}

__global__ void forcing_max_kernel(const bf16_t* __restrict__ f, bf16_t* __restrict__ out, int frames,
                                   long long per_frame, int n_id) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (r, id)
    const long long inner = per_frame * n_id;
    if (idx >= inner) return;
    float m = bf2f(f[idx]);
    bf16_t mb = f[idx];
    for (int t = 1; t < frames; ++t) {
        const bf16_t vb = f[t * inner + idx];
        const float v = bf2f(vb);
        if (v > m) { m = v; mb = vb; }
    }
    for (int t = 0; t < frames; ++t) out[t * inner + idx] = mb;
}

// ------------------------------------------------------------------------------------------------
__global__ void patchify_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ cols, int batch, int frames, int C,
                                int H, int W) {
    const int Ht = H / 2, Wt = W / 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, t, c, h, w), w fastest
    const long long total = (long long)batch * frames * C * Ht * Wt;
    if (idx >= total) return;
    const int w = (int)(idx % Wt); long long rest = idx / Wt;
    const int h = (int)(rest % Ht); rest /= Ht;
    const int c = (int)(rest % C); rest /= C;
    const int t = (int)(rest % frames);
    const int b = (int)(rest / frames);
    const bf16_t* src = x + ((((long long)b * frames + t) * C + c) * H + 2 * h) * W + 2 * w;
    const uint32_t top = *reinterpret_cast<const uint32_t*>(src);
    const uint32_t bot = *reinterpret_cast<const uint32_t*>(src + W);
    const long long n = ((long long)t * Ht + h) * Wt + w;
    u32x2 o; o[0] = top; o[1] = bot;
    *reinterpret_cast<u32x2*>(cols + (((long long)b * frames * Ht * Wt + n) * C + c) * 4) = o;
}

__global__ void unpatchify_kernel(const bf16_t* __restrict__ y, bf16_t* __restrict__ out, int batch, int frames, int C,
                                  int H, int W) {
    const int Ht = H / 2, Wt = W / 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, t, c, h, w), w fastest
    const long long total = (long long)batch * frames * C * Ht * Wt;
    if (idx >= total) return;
    const int w = (int)(idx % Wt); long long rest = idx / Wt;
    const int h = (int)(rest % Ht); rest /= Ht;
    const int c = (int)(rest % C); rest /= C;
    const int t = (int)(rest % frames);
    const int b = (int)(rest / frames);
    const long long n = ((long long)t * Ht + h) * Wt + w;
    const u32x2 v = *reinterpret_cast<const u32x2*>(y + (((long long)b * frames * Ht * Wt + n) * C + c) * 4);
    bf16_t* dst = out + ((((long long)b * frames + t) * C + c) * H + 2 * h) * W + 2 * w;
    *reinterpret_cast<uint32_t*>(dst) = v[0];
    *reinterpret_cast<uint32_t*>(dst + W) = v[1];
}

__global__ __launch_bounds__(256) void act_add_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ r,
                                                      bf16_t* __restrict__ y, long long nvec, int act) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + i * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = v[e];
            switch (act) {
                case BYA_ACT_GELU_TANH: t = gelu_tanh(t); break;
                case BYA_ACT_GELU_ERF: t = gelu_erf(t); break;
                case BYA_ACT_RELU: t = t > 0.f ? t : 0.f; break;
                case BYA_ACT_SILU: t = silu(t); break;
                case BYA_ACT_LEAKY_RELU: t = t > 0.f ? t : 0.01f * t; break;
                default: break;
            }
            v[e] = t;
        }
        if (r) {
            float rv[8];
            unpack8(*reinterpret_cast<const u32x4*>(r + i * 8), rv);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e])) + rv[e];
        }
        *reinterpret_cast<u32x4*>(y + i * 8) = pack8(v);
    }
}

inline int ok() { const hipError_t e = hipGetLastError(); return e == hipSuccess ? BYA_OK : -(1000 + (int)e); }


// ---- classifier-free-guidance combine + scheduler step, fused -------------------------
// One pass over the latents instead of ~10 elementwise launches.  The fp32 / bf16 rounding points are the ones torch's
// type promotion produces for the expressions of the reference loop (models/pipeline_bindyouravatar.py:924-948) and of
// diffusers' CogVideoX DDIM / DPM schedulers: a 0-dim coefficient times the bf16 sample (or noise) is a bf16 product,
// everything touching the fp32 prediction stays fp32.  FP contraction is off so no product is fused into an FMA.
#pragma clang fp contract(off)
__global__ void __launch_bounds__(256)
cfg_sched_kernel(const bf16_t* __restrict__ pred, long long pred_stride, int n_pred, const bf16_t* __restrict__ x,
                 const float* __restrict__ old_x0, const bf16_t* __restrict__ noise, bf16_t* __restrict__ prev,
                 float* __restrict__ x0_out, long long n, bya_sched_coef c) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = bf2f(pred[i]);
        if (n_pred == 2) {
            const float cond = bf2f(pred[pred_stride + i]);
            const float diff = cond - v;
            const float scaled = c.guidance * diff;
            v = v + scaled;
        }
        const float xs = bf2f(x[i]);
        const float t1 = bf2f(f2bf(c.sqrt_alpha * xs));
        const float t2 = c.sqrt_beta * v;
        const float x0 = t1 - t2;
        float d = x0;
        if (old_x0) {
            const float cur = c.k_cur * x0;
            const float old = c.k_old * old_x0[i];
            d = cur - old;
        }
        const float t3 = bf2f(f2bf(c.k_sample * xs));
        const float t4 = c.k_denoised * d;
        float r = t3 - t4;
        if (noise) {
            const float tn = bf2f(f2bf(c.k_noise * bf2f(noise[i])));
            r = r + tn;
        }
        prev[i] = f2bf(r);
        if (x0_out) x0_out[i] = x0;
    }
}


// ---- tracking masks -> routing_logits_forcing (reference util/utils.py:481-514, 871-936) ---------------------------
// One thread per latent token.  For every identity: trilinear sample (align_corners = false) of the binary mask
// video at the token's centre, in fp32 with torch's index / weight construction and evaluation order
// (outer T, then H, innermost W; out = a * w0 + b * w1 per level), threshold 0.5; later identities overwrite earlier
// ones; the token's row is one-hot (or all zero for background).
struct Lin1 { int i0, i1; float w0, w1; };
__device__ __forceinline__ Lin1 lin_index(int dst, int in_size, int out_size) {
    Lin1 r;
    if (in_size == out_size) { r.i0 = r.i1 = dst; r.w0 = 1.0f; r.w1 = 0.0f; return r; }
    const float scale = (float)in_size / (float)out_size;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.0f) src = 0.0f;
    r.i0 = (int)src;
    r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
    float l1 = src - (float)r.i0;
    l1 = fminf(fmaxf(l1, 0.0f), 1.0f);
    r.w1 = l1;
    r.w0 = 1.0f - l1;
    return r;
}

__global__ void __launch_bounds__(256)
masks_to_logits_kernel(const uint8_t* __restrict__ masks, bf16_t* __restrict__ out, int n_id, int Ti, int Hi, int Wi,
                       int To, int Ho, int Wo) {
    const long long n = (long long)To * Ho * Wo;
    const long long tok = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tok >= n) return;
    const int w = (int)(tok % Wo), h = (int)((tok / Wo) % Ho), t = (int)(tok / ((long long)Wo * Ho));
    const Lin1 lt = lin_index(t, Ti, To), lh = lin_index(h, Hi, Ho), lw = lin_index(w, Wi, Wo);
    int label = -1;
    for (int id = 0; id < n_id; ++id) {
        const uint8_t* m = masks + (long long)id * Ti * Hi * Wi;
        auto px = [&](int tt, int hh, int ww) -> float { return m[((long long)tt * Hi + hh) * Wi + ww] > 0 ? 1.0f : 0.0f; };
        auto row = [&](int tt, int hh) -> float {
            float o = px(tt, hh, lw.i0) * lw.w0;
            o += px(tt, hh, lw.i1) * lw.w1;
            return o;
        };
        auto plane = [&](int tt) -> float {
            float o = row(tt, lh.i0) * lh.w0;
            o += row(tt, lh.i1) * lh.w1;
            return o;
        };
        float v = plane(lt.i0) * lt.w0;
        v += plane(lt.i1) * lt.w1;
        if (v > 0.5f) label = id;
    }
    for (int id = 0; id < n_id; ++id) out[tok * n_id + id] = (id == label) ? (bf16_t)0x3F80 : (bf16_t)0;
}

}  // namespace

extern "C" int bya_abi_version(void) { return 2; }

// ---- process-wide options (include/bya.h, csrc/options.h): defaults, ranges, setter ----------------------------------
std::atomic<int32_t> g_bya_options[BYA_OPT_COUNT] = {
    {0},      // BYA_OPT_GEMM_SPLITK
    {0},      // BYA_OPT_GEMM_SPLITK_MIN
    {-1},     // BYA_OPT_GEMM_TILE
    {0},      // BYA_OPT_GEMM_VARIANT
    {1},      // BYA_OPT_ATTN_STREAMK
    {0},      // BYA_OPT_FP8_KERNEL
    {0},      // BYA_OPT_P2P_GROUPS
    {0},      // BYA_OPT_REFERENCE_FORMS
};

extern "C" int bya_set_option(int32_t key, int32_t value) {
    static const int32_t lo[BYA_OPT_COUNT] = {0, 0, -1, 0, 0, 0, 0, 0};
    static const int32_t hi[BYA_OPT_COUNT] = {2, 1 << 20, 6, 2, 1, 1, 1024, 31};
    if (key < 0 || key >= BYA_OPT_COUNT || value < lo[key] || value > hi[key]) return BYA_ERR_SHAPE;
    if (key == BYA_OPT_P2P_GROUPS && value != 0 && value < 16) return BYA_ERR_SHAPE;
    g_bya_options[key].store(value, std::memory_order_relaxed);
    return BYA_OK;
}

extern "C" int bya_get_option(int32_t key, int32_t* value) {
    if (key < 0 || key >= BYA_OPT_COUNT || !value) return BYA_ERR_SHAPE;
    *value = g_bya_options[key].load(std::memory_order_relaxed);
    return BYA_OK;
}

extern "C" int bya_linear_small_m(const void* x, const void* W, const void* bias, void* out, int32_t M, int32_t N,
                                  int32_t K, int32_t silu_in, int32_t act_out, hipStream_t stream) {
    if (!x || !W || !out || M <= 0 || M > 8 || N <= 0 || K <= 0) return BYA_ERR_SHAPE;
    if (K % 8) return BYA_ERR_SHAPE;
    if (((uintptr_t)x | (uintptr_t)W) & 15) return BYA_ERR_ALIGN;
    if (act_out != BYA_ACT_NONE && act_out != BYA_ACT_SILU) return BYA_ERR_UNSUPPORTED;
    dim3 grid((unsigned)((N + 3) / 4));
    if (M <= 2)
        BYA_LAUNCH((linear_small_m_kernel<2>), grid, dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)W,
                           (const bf16_t*)bias, (bf16_t*)out, M, N, K, silu_in, act_out);
    else
        BYA_LAUNCH((linear_small_m_kernel<8>), grid, dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)W,
                           (const bf16_t*)bias, (bf16_t*)out, M, N, K, silu_in, act_out);
    return ok();
}

extern "C" int bya_timestep_features(const int64_t* timesteps, void* out, int32_t batch, int32_t dim,
                                     int32_t flip_sin_to_cos, float freq_shift, hipStream_t stream) {
    if (!timesteps || !out || batch <= 0 || dim <= 0 || dim % 2) return BYA_ERR_SHAPE;
    const int total = batch * (dim / 2);
    BYA_LAUNCH(timestep_features_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, timesteps,
                       (bf16_t*)out, batch, dim, flip_sin_to_cos, freq_shift);
    return ok();
}

extern "C" int bya_masked_combine(void* x, const void* feat, const void* r, const void* af, int32_t mode, float alpha,
                                  int32_t batch, int32_t n_id, int64_t N, int32_t D, int64_t x_row,
                                  int64_t x_batch_stride, int64_t r_batch_stride, hipStream_t stream) {
    if (!x || !feat || !r || batch <= 0 || N <= 0 || D <= 0) return BYA_ERR_SHAPE;
    if (mode != 0 && mode != 1) return BYA_ERR_UNSUPPORTED;
    if (n_id < 1 || n_id > 4 || (mode == 1 && !af)) return BYA_ERR_UNSUPPORTED;
    if (D % 8 || x_row % 8 || x_batch_stride % 8) return BYA_ERR_ALIGN;
    if (((uintptr_t)x | (uintptr_t)feat) & 15) return BYA_ERR_ALIGN;
    CombArgs a;
    a.x = (bf16_t*)x; a.feat = (const bf16_t*)feat; a.r = (const bf16_t*)r; a.af = (const bf16_t*)af;
    a.mode = mode; a.batch = batch; a.n_id = n_id; a.D = D; a.N = N; a.x_row = x_row; a.x_bs = x_batch_stride;
    a.r_bs = r_batch_stride; a.alpha = alpha;
    const long long total = (long long)batch * N * (D / 8);
    BYA_LAUNCH(masked_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    return ok();
}

extern "C" int bya_routed_mix(const void* feat, const void* r, const void* af, void* z, float* wsum, int32_t mode,
                              int32_t batch, int32_t n_id, int64_t N, int32_t D, int64_t r_batch_stride,
                              hipStream_t stream) {
    if (!feat || !r || !z || batch <= 0 || N <= 0 || D <= 0) return BYA_ERR_SHAPE;
    if (mode != 0 && mode != 1) return BYA_ERR_UNSUPPORTED;
    if (n_id < 1 || n_id > 4 || (mode == 1 && !af)) return BYA_ERR_UNSUPPORTED;
    if (D % 8) return BYA_ERR_ALIGN;
    if (((uintptr_t)z | (uintptr_t)feat) & 15) return BYA_ERR_ALIGN;
    CombArgs a;
    a.x = nullptr; a.feat = (const bf16_t*)feat; a.r = (const bf16_t*)r; a.af = (const bf16_t*)af;
    a.mode = mode; a.batch = batch; a.n_id = n_id; a.D = D; a.N = N; a.x_row = 0; a.x_bs = 0;
    a.r_bs = r_batch_stride; a.alpha = 1.0f;
    const long long total = (long long)batch * N * (D / 8);
    BYA_LAUNCH(routed_mix_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, (bf16_t*)z, wsum);
    return ok();
}

extern "C" int bya_forcing_max_over_frames(const void* forcing, void* out, int32_t frames, int64_t per_frame,
                                           int32_t n_id, hipStream_t stream) {
    if (!forcing || !out || frames <= 0 || per_frame <= 0 || n_id <= 0) return BYA_ERR_SHAPE;
    const long long inner = per_frame * n_id;
    BYA_LAUNCH(forcing_max_kernel, dim3((unsigned)((inner + 255) / 256)), dim3(256), 0, stream,
                       (const bf16_t*)forcing, (bf16_t*)out, frames, (long long)per_frame, n_id);
    return ok();
}

extern "C" int bya_patchify(const void* x, void* cols, int32_t batch, int32_t frames, int32_t channels, int32_t H,
                            int32_t W, hipStream_t stream) {
    if (!x || !cols || batch <= 0 || frames <= 0 || channels <= 0 || H <= 0 || W <= 0) return BYA_ERR_SHAPE;
    if (H % 2 || W % 2) return BYA_ERR_SHAPE;
    if (((uintptr_t)x & 3) || ((uintptr_t)cols & 7)) return BYA_ERR_ALIGN;
    const long long total = (long long)batch * frames * channels * (H / 2) * (W / 2);
    BYA_LAUNCH(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)x,
                       (bf16_t*)cols, batch, frames, channels, H, W);
    return ok();
}

extern "C" int bya_unpatchify(const void* y, void* out, int32_t batch, int32_t frames, int32_t channels, int32_t H,
                              int32_t W, hipStream_t stream) {
    if (!y || !out || batch <= 0 || frames <= 0 || channels <= 0 || H <= 0 || W <= 0) return BYA_ERR_SHAPE;
    if (H % 2 || W % 2) return BYA_ERR_SHAPE;
    if (((uintptr_t)out & 3) || ((uintptr_t)y & 7)) return BYA_ERR_ALIGN;
    const long long total = (long long)batch * frames * channels * (H / 2) * (W / 2);
    BYA_LAUNCH(unpatchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       (const bf16_t*)y, (bf16_t*)out, batch, frames, channels, H, W);
    return ok();
}

extern "C" int bya_act_add(const void* x, const void* r, void* y, int64_t n, int32_t act, hipStream_t stream) {
    if (!x || !y || n <= 0 || n % 8) return BYA_ERR_SHAPE;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)r) & 15) return BYA_ERR_ALIGN;
    if (act < 0 || act > 5) return BYA_ERR_UNSUPPORTED;
    const long long nvec = n / 8;
    long long blocks = (nvec + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    BYA_LAUNCH(act_add_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16_t*)x,
                       (const bf16_t*)r, (bf16_t*)y, nvec, act);
    return ok();
}

extern "C" int bya_cfg_scheduler_step(const void* pred, int32_t n_pred, int64_t pred_stride, const void* sample,
                                      const float* old_x0, const void* noise, void* prev_sample, float* x0_out,
                                      int64_t n, const bya_sched_coef* coef, hipStream_t stream) {
    if (!pred || !sample || !prev_sample || !coef || n <= 0) return BYA_ERR_SHAPE;
    if (n_pred != 1 && n_pred != 2) return BYA_ERR_SHAPE;
    if (n_pred == 2 && pred_stride < n) return BYA_ERR_SHAPE;
    long long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    BYA_LAUNCH(cfg_sched_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16_t*)pred,
               (long long)pred_stride, n_pred, (const bf16_t*)sample, old_x0, (const bf16_t*)noise,
               (bf16_t*)prev_sample, x0_out, (long long)n, *coef);
    return ok();
}

extern "C" int bya_masks_to_routing_logits(const void* masks, void* logits, int32_t n_id, int32_t in_frames,
                                           int32_t in_h, int32_t in_w, int32_t frames, int32_t h, int32_t w,
                                           hipStream_t stream) {
    if (!masks || !logits || n_id <= 0 || in_frames <= 0 || in_h <= 0 || in_w <= 0 || frames <= 0 || h <= 0 || w <= 0)
        return BYA_ERR_SHAPE;
    const long long n = (long long)frames * h * w;
    BYA_LAUNCH(masks_to_logits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const uint8_t*)masks,
               (bf16_t*)logits, n_id, in_frames, in_h, in_w, frames, h, w);
    return ok();
}
