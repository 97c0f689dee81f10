// fp8 (OCP e4m3) "NT" GEMM on the block-scaled CDNA4 matrix instruction, and the row quantiser that feeds it:
//     C[m,n] = epi( sa[m] * sw[n] * sum_k A8[m,k] * W8[n,k] ),   A8 = fp8(A / sa),  W8 = fp8(W / sw)
// BASELINE configs[4] ("fp8 weights on CDNA4 fp8 MFMA"): the four big Linears of a DiT block
// (attn1.to_q|k|v, attn1.to_out, ff.net.0.proj, ff.net.2 -- models/transformer.py:241-260 via diffusers Attention /
// FeedForward) with per-output-channel weight scales fixed at load time and per-row activation scales taken on the fly.
// The reference has no fp8 path; parity is against the CPU restatement run on the same
// fake-quantised operands (tests/test_fp8_gpu.py).
//
// v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands runs at twice the bf16 rate per clock (MI355X_MICROARCH.md, matrix
// cores): 128 k per instruction, 32 consecutive bytes of one row per lane, both block scales fixed at 2^0 -- the row
// and channel scales are fp32 and are applied to the accumulators in the epilogue.  One K-tile is 128 fp8 = 128-byte LDS
// rows, i.e. exactly the staging / XOR-swizzle image of the bf16 128 x 128 kernel (gemm.hip) at twice the FLOPs per byte.
// Same epilogue as every other GEMM here (bias, activation, gate, residual, q|k|v split): gemm_common.h.
#include "gemm_fp8_kernel.h"
#include "options.h"

bool bya_gemm256p_fp8_eligible(const void* args);                                                       // gemm_fp8_v4.hip
int bya_launch_gemm256p_fp8(const void* args, const float* sa, const float* sw, int batch, int gm, hipStream_t s);

namespace {

// ---- row quantiser: one workgroup per row.  scale[m] = max|x[m,:]| / 448 (1 for an all-zero row),
// q[m,k] = e4m3( x[m,k] * (448 / max|x[m,:]|) ), round-to-nearest-even (f32_to_e4m3, bya_common.h).
constexpr int QT = 256, QV = 6;                 // threads per row, 8-element vectors per thread: K <= 12288

__global__ __launch_bounds__(QT) void quant_rows_fp8_kernel(const bf16_t* __restrict__ x, uint8_t* __restrict__ q,
                                                            float* __restrict__ scale, int K, long long ldx, long long ldq) {
    __shared__ float red[QT / 64];
    const int row = blockIdx.x, tid = threadIdx.x;
    const bf16_t* xr = x + (long long)row * ldx;
    const int nv = K / 8;
    u32x4 v[QV];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < QV; ++i) {
        const int c = tid + i * QT;
        v[i] = c < nv ? *reinterpret_cast<const u32x4*>(xr + c * 8) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int w = 0; w < 4; ++w) amax = fmaxf(amax, fmaxf(fabsf(bflo(v[i][w])), fabsf(bfhi(v[i][w]))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    // 448 / amax correctly rounded to fp32 (NOT 448 * (1 / amax): the two differ in the last bit for one row in three, and
    // that moves exact ties such as 8.3125 * (448 / 12.25) = 304 to the other e4m3 neighbour -- 0.13 % of the bytes)
    const float inv = amax > 0.f ? (float)((double)FP8_MAX / (double)amax) : 0.f;
    if (tid == 0) scale[row] = amax > 0.f ? (float)((double)amax / (double)FP8_MAX) : 1.0f;
    uint8_t* qr = q + (long long)row * ldq;
#pragma unroll
    for (int i = 0; i < QV; ++i) {
        const int c = tid + i * QT;
        if (c < nv) {
            u32x2 o;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t w0 = v[i][2 * h], w1 = v[i][2 * h + 1];
                o[h] = f32_to_e4m3(bflo(w0) * inv) | (f32_to_e4m3(bfhi(w0) * inv) << 8) |
                       (f32_to_e4m3(bflo(w1) * inv) << 16) | (f32_to_e4m3(bfhi(w1) * inv) << 24);
            }
            *reinterpret_cast<u32x2*>(qr + c * 8) = o;
        }
    }
}

}  // namespace

extern "C" int bya_quantize_rows_fp8(const void* x, void* q, float* scale, int32_t M, int32_t K, int64_t ldx, int64_t ldq,
                                     hipStream_t stream) {
    if (!x || !q || !scale || M <= 0 || K <= 0) return BYA_ERR_SHAPE;
    if (K % 8 || K > 8 * QT * QV) return BYA_ERR_SHAPE;
    if (ldx < K || ldq < K || ldx % 8 || ldq % 8) return BYA_ERR_ALIGN;
    if (((uintptr_t)x & 15) || ((uintptr_t)q & 7)) return BYA_ERR_ALIGN;
    BYA_LAUNCH(quant_rows_fp8_kernel, dim3(M), dim3(QT), 0, stream, (const bf16_t*)x, (uint8_t*)q, scale, K, (long long)ldx,
               (long long)ldq);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

extern "C" int bya_gemm_fp8(const void* A8, const float* a_scale, const void* W8, const float* w_scale, const void* bias,
                            void* C, const void* res, const void* gate0, const void* gate1, const bya_gemm_desc* d,
                            hipStream_t stream) {
    if (!A8 || !W8 || !a_scale || !w_scale || !C || !d) return BYA_ERR_SHAPE;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0) return BYA_ERR_SHAPE;
    if (d->K % BK8 != 0 || d->N % 4 != 0) return BYA_ERR_SHAPE;
    if (d->lda % 16 || d->ldw % 16 || d->ldc % 4 || (res && d->ldres % 4)) return BYA_ERR_ALIGN;
    if (((uintptr_t)A8 | (uintptr_t)W8 | (uintptr_t)w_scale) & 15) return BYA_ERR_ALIGN;
    if (((uintptr_t)C | (uintptr_t)res | (uintptr_t)bias | (uintptr_t)gate0 | (uintptr_t)gate1) & 7) return BYA_ERR_ALIGN;
    if (!act_on_big_tiles(d->act)) return BYA_ERR_UNSUPPORTED;            // none / GELU(tanh): the DiT Linears
    if (d->n_split < 0 || (d->n_split > 0 && (d->n_split % 4 || d->c_split_stride % 4 || res))) return BYA_ERR_SHAPE;
    GemmArgs a;
    a.A = (const bf16_t*)A8; a.W = (const bf16_t*)W8; a.bias = (const bf16_t*)bias; a.C = (bf16_t*)C;
    a.res = (const bf16_t*)res; a.gate0 = (const bf16_t*)gate0; a.gate1 = (const bf16_t*)(gate1 ? gate1 : gate0);
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.lda = d->lda; a.ldw = d->ldw; a.ldc = d->ldc; a.ldres = d->ldres;
    a.a_bs = d->a_batch_stride; a.c_bs = d->c_batch_stride; a.res_bs = d->res_batch_stride;
    a.gate_bs = d->gate_batch_stride; a.gate_split = d->gate_split; a.act = d->act; a.leaky = 0.01f;
    a.n_split = d->n_split; a.c_split_stride = d->c_split_stride;
    a.bias_rowscale = d->bias_rowscale; a.alpha = d->alpha == 0.0f ? 1.0f : d->alpha;
    a.ws_counters = nullptr; a.ws_slabs = nullptr;
    // Two 128 x 128 workgroups per CU (4 waves, 64 KiB LDS ring each) cover each other's barrier and LDS-DMA waits; a
    // 256 x 256 form (8 waves, 128 KiB ring, one workgroup per CU, half the L2 -> LDS bytes per FLOP) measured 4-16 %
    // slower on the four DiT shapes with this simple two-barrier loop, a one-wave-per-SIMD instantiation 8-20 % slower under
    // hipcc's schedule (profiles/history/r2_fp8_probe.txt; both removed from the tree in round 3).
    // ... until the loop was placed by hand: gemm_fp8_v4.hip (one wave per SIMD, persistent) takes every launch that is big
    // enough to fill its 256 x 256 tiles; BYA_FP8_KERNEL=128 (read per call: A/B runs) keeps everything on the kernel above.
    const long long tiles256 = (long long)((d->M + 255) / 256) * ((d->N + 255) / 256) * d->batch;
    const bool big = !bya_opt(BYA_OPT_FP8_KERNEL) && tiles256 >= 200;        // (about a round of its 256 workgroups, or more)
    const int gm = 4;                                                 // row-tiles per group of the persistent kernel's tile order
    return gemm_row_chunks(a, d->batch, 1, [&](const GemmArgs& piece, int batch, long long row0) {
        if (big && bya_gemm256p_fp8_eligible(&piece))
            return bya_launch_gemm256p_fp8(&piece, a_scale + row0, w_scale, batch, gm, stream);
        return launch_fp8<128, 128, 2, 2>(piece, a_scale + row0, w_scale, batch, stream);
    });
}
