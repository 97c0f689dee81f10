// Embedding-Router kernels (models/router.py:364-411, 468-493):
//  * router_scores  -- per-head q.k^T of the re-projected perceiver q/k, fused with LayerNorm(512) and the
//                      3-D positional table add (router.py:385-399)
//  * router_head    -- Linear(512->1) + sigmoid, written in the [N, n_id] routing-logit layout (router.py:408-411)
//  * attn_tiny      -- the temporal (13-long) and multi-ID (2-long) self-attentions of the
//                      SpatialTemporalAttentionBlock (router.py:480-488); sequences are far too short for MFMA
//                      tiles, so each wave keeps one (sequence, head) in registers with the head dim on the lanes.
#include "bya_common.h"
#include "../../include/bya.h"
#include <stdlib.h>
#include "options.h"

namespace {

constexpr int R_HEADS = 16, R_TOK = 32, R_HD = 128, R_QK = R_HEADS * R_HD, R_FEAT = R_HEADS * R_TOK;

// One wave = 16 video tokens of one identity.  S^T tile (16 face tokens x 16 video tokens) per (head, half):
// A = kr rows (face token), B = qr rows (video token), v_mfma_f32_16x16x32_bf16, K = 128 = 4 steps.
// Lane (c = lane&15, g = lane>>4) ends up with score(video token c, face token 16j + 4g + e, head h) in
// acc[h][j][e]; output feature index = face_token * 16 + h  (router.py:389-390 permute + reshape).
__global__ __launch_bounds__(256) void router_scores_kernel(const bf16_t* __restrict__ qr, const bf16_t* __restrict__ kr,
                                                            const bf16_t* __restrict__ ln_w, const bf16_t* __restrict__ ln_b,
                                                            const bf16_t* __restrict__ pos, bf16_t* __restrict__ out,
                                                            int n_id, long long N, float eps) {
    const int lane = threadIdx.x & 63;
    const long long tiles = (N + 15) / 16;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= tiles * n_id) return;
    const int id = (int)(wid / tiles);
    const long long n0 = (wid % tiles) * 16;
    const int c = lane & 15, g = lane >> 4;
    long long n = n0 + c;
    const bool valid = n < N;
    n = valid ? n : N - 1;
    const bf16_t* qrow = qr + n * R_QK + g * 8;
    const bf16_t* krow0 = kr + ((long long)id * R_TOK + c) * R_QK + g * 8;
    const bf16_t* krow1 = krow0 + 16 * R_QK;

    f32x4 acc[R_HEADS][2];
#pragma unroll
    for (int h = 0; h < R_HEADS; ++h) {
        acc[h][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[h][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int off = h * R_HD + ks * 32;
            const bf16x8 qf = *reinterpret_cast<const bf16x8*>(qrow + off);
            const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(krow0 + off);
            const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(krow1 + off);
            acc[h][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf, acc[h][0], 0, 0, 0);
            acc[h][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf, acc[h][1], 0, 0, 0);
        }
    }
    // LayerNorm over the 512 features of video token c: 128 in this lane, the rest in lanes c+16, c+32, c+48
    float sum = 0.f;
#pragma unroll
    for (int h = 0; h < R_HEADS; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += acc[h][j][e];
    sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / R_FEAT);
    float sq = 0.f;
#pragma unroll
    for (int h = 0; h < R_HEADS; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = acc[h][j][e] - mean; sq += d * d; }
    sq += __shfl_xor(sq, 16, 64); sq += __shfl_xor(sq, 32, 64);
    const float rstd = rsqrtf(sq * (1.0f / R_FEAT) + eps);
    if (!valid) return;
    bf16_t* orow = out + ((long long)id * N + n) * R_FEAT;
    const bf16_t* prow = pos + n * R_FEAT;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int tok = 16 * j + 4 * g + e;
            const int f0 = tok * R_HEADS;       // 16 consecutive features (h = 0..15) = 32 bytes
            float wv[16], bv[16], pv[16], o[16];
            unpack8(*reinterpret_cast<const u32x4*>(ln_w + f0), wv);
            unpack8(*reinterpret_cast<const u32x4*>(ln_w + f0 + 8), wv + 8);
            unpack8(*reinterpret_cast<const u32x4*>(ln_b + f0), bv);
            unpack8(*reinterpret_cast<const u32x4*>(ln_b + f0 + 8), bv + 8);
            unpack8(*reinterpret_cast<const u32x4*>(prow + f0), pv);
            unpack8(*reinterpret_cast<const u32x4*>(prow + f0 + 8), pv + 8);
#pragma unroll
            for (int h = 0; h < R_HEADS; ++h)
                o[h] = bf2f(f2bf((acc[h][j][e] - mean) * rstd * wv[h] + bv[h])) + pv[h];   // LN output bf16, then + pos
            *reinterpret_cast<u32x4*>(orow + f0) = pack8(o);
            *reinterpret_cast<u32x4*>(orow + f0 + 8) = pack8(o + 8);
        }
}

// router_scores with the identity's keys in LDS.  The kernel above is one wave round of DEPENDENT loads: every wave fetches its
// 64 q fragments and, again, all 128 key fragments (128 KB from L2) with ~30 registers left for loads in flight beside the 128
// accumulators -- 98 us for 4.6 GFLOP and ~220 MB.  Here a workgroup of 8 waves owns ONE identity: its 32 re-projected keys
// (32 x 4 KB = 128 KB) are staged once by LDS-DMA (source-side XOR swizzle of the 16-byte chunk index with row & 15: the
// sixteen rows of a fragment read hit sixteen different chunks), and the waves walk the identity's 16-token tiles on their
// own: per tile 64 global loads (q only; nothing else competes for the vector-memory queue), 128 ds_read_b128, 128 MFMAs,
// the same LayerNorm + positional epilogue.  Same MFMA order per accumulator -> bit-identical.
__global__ __launch_bounds__(512) void router_scores_lds_kernel(const bf16_t* __restrict__ qr, const bf16_t* __restrict__ kr,
                                                                const bf16_t* __restrict__ ln_w, const bf16_t* __restrict__ ln_b,
                                                                const bf16_t* __restrict__ pos, bf16_t* __restrict__ out,
                                                                int n_id, long long N, float eps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int id = blockIdx.x % n_id, wi = blockIdx.x / n_id, per_id = gridDim.x / n_id;
    if (wi >= per_id) return;                                  // (grid not a multiple of n_id: the last few workgroups idle)
    constexpr int ROWB = R_QK * 2;                             // 4096 bytes per key row
    {   // rows 4 wave .. 4 wave + 3, four 1-KiB pieces each; LDS chunk p of row r holds source chunk p ^ (r & 15)
        const char* ksrc = reinterpret_cast<const char*>(kr + (long long)id * R_TOK * R_QK);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wave * 4 + rr, pchunk = 64 * i + lane;
                const char* g = ksrc + (long long)r * ROWB + ((pchunk ^ (r & 15)) << 4);
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(g), LDS_PTR(smem + r * ROWB + i * 1024), 16, 0, 0);
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                           // the keys are in place; no barrier below

    const int c = lane & 15, g = lane >> 4;
    const long long tiles = (N + 15) / 16;
    const uint32_t kbase = (uint32_t)(uintptr_t)LDS_PTR(smem) + c * ROWB;
    for (long long t = (long long)wi * 8 + wave; t < tiles; t += (long long)per_id * 8) {
        long long n = t * 16 + c;
        const bool valid = n < N;
        n = valid ? n : N - 1;
        const bf16_t* qrow = qr + n * R_QK + g * 8;
        f32x4 acc[R_HEADS][2];
        // q fragments two heads ahead (8 loads = 8 KB in flight per wave; four heads ahead measured the same); the scheduling fences keep hipcc from hoisting all 64
        // loads and 128 LDS reads of the unrolled tile to its top (1.8 KB of scratch per lane when it is left alone)
        bf16x8 qb[3][4];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qb[hh][ks] = *reinterpret_cast<const bf16x8*>(qrow + hh * R_HD + ks * 32);
#pragma unroll
        for (int h = 0; h < R_HEADS; ++h) {
            if (h + 2 < R_HEADS) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    qb[(h + 2) % 3][ks] = *reinterpret_cast<const bf16x8*>(qrow + (h + 2) * R_HD + ks * 32);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[h][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[h][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const uint32_t a = kbase + (uint32_t)(((16 * h + 4 * ks + g) ^ c) << 4);
                const bf16x8 k0 = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(a);
                const bf16x8 k1 = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(a + 16 * ROWB);
                acc[h][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qb[h % 3][ks], acc[h][0], 0, 0, 0);
                acc[h][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qb[h % 3][ks], acc[h][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        float sum = 0.f;
#pragma unroll
        for (int h = 0; h < R_HEADS; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) sum += acc[h][j][e];
        sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / R_FEAT);
        float sq = 0.f;
#pragma unroll
        for (int h = 0; h < R_HEADS; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = acc[h][j][e] - mean; sq += d * d; }
        sq += __shfl_xor(sq, 16, 64); sq += __shfl_xor(sq, 32, 64);
        const float rstd = rsqrtf(sq * (1.0f / R_FEAT) + eps);
        if (!valid) continue;
        bf16_t* orow = out + ((long long)id * N + n) * R_FEAT;
        const bf16_t* prow = pos + n * R_FEAT;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int tok = 16 * j + 4 * g + e;
                const int f0 = tok * R_HEADS;
                float wv[16], bv[16], pv[16], o[16];
                unpack8(*reinterpret_cast<const u32x4*>(ln_w + f0), wv);
                unpack8(*reinterpret_cast<const u32x4*>(ln_w + f0 + 8), wv + 8);
                unpack8(*reinterpret_cast<const u32x4*>(ln_b + f0), bv);
                unpack8(*reinterpret_cast<const u32x4*>(ln_b + f0 + 8), bv + 8);
                unpack8(*reinterpret_cast<const u32x4*>(prow + f0), pv);
                unpack8(*reinterpret_cast<const u32x4*>(prow + f0 + 8), pv + 8);
#pragma unroll
                for (int h = 0; h < R_HEADS; ++h)
                    o[h] = bf2f(f2bf((acc[h][j][e] - mean) * rstd * wv[h] + bv[h])) + pv[h];
                *reinterpret_cast<u32x4*>(orow + f0) = pack8(o);
                *reinterpret_cast<u32x4*>(orow + f0 + 8) = pack8(o + 8);
            }
    }
}

// r[n, id] = sigmoid(x[id, n, :] . w + b); D = 512: one wave per row, 8 elements per lane.
__global__ __launch_bounds__(256) void router_head_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          const bf16_t* __restrict__ b, bf16_t* __restrict__ r, int n_id,
                                                          long long N, int D) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N * n_id) return;
    const int id = (int)(row / N);
    const long long n = row % N;
    float acc = 0.f;
    for (int k0 = lane * 8; k0 < D; k0 += 512) {
        float xv[8], wv[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + row * D + k0), xv);
        unpack8(*reinterpret_cast<const u32x4*>(w + k0), wv);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += xv[e] * wv[e];
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        const float z = bf2f(f2bf(acc + bf2f(b[0])));   // Linear output is bf16 before the sigmoid
        r[n * n_id + id] = f2bf(1.0f / (1.0f + __expf(-z)));
    }
}

// Tiny self-attention: one wave per (sequence, head); lane = head-dim index (64).
template <int MAXL>
__global__ __launch_bounds__(256) void attn_tiny_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                        const bf16_t* __restrict__ v, bf16_t* __restrict__ o, int L,
                                                        int heads, long long n_outer, long long n_inner,
                                                        long long outer_stride, long long seq_stride, long long ld_qkv,
                                                        long long ld_o, float scale) {
    const int lane = threadIdx.x & 63;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long total = n_outer * n_inner * heads;
    if (wid >= total) return;
    const int h = (int)(wid % heads);
    const long long gseq = wid / heads;
    const long long row0 = (gseq / n_inner) * outer_stride + (gseq % n_inner);
    float qv[MAXL], kv[MAXL], vv[MAXL];
#pragma unroll
    for (int e = 0; e < MAXL; ++e) {
        if (e < L) {
            const long long row = row0 + e * seq_stride;
            qv[e] = bf2f(q[row * ld_qkv + h * 64 + lane]) * scale;
            kv[e] = bf2f(k[row * ld_qkv + h * 64 + lane]);
            vv[e] = bf2f(v[row * ld_qkv + h * 64 + lane]);
        }
    }
#pragma unroll
    for (int i = 0; i < MAXL; ++i) {
        if (i < L) {
            float s[MAXL];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < MAXL; ++j) {
                if (j < L) { s[j] = wave_sum(qv[i] * kv[j]); mx = fmaxf(mx, s[j]); }
            }
            float den = 0.f, acc = 0.f;
#pragma unroll
            for (int j = 0; j < MAXL; ++j) {
                if (j < L) { const float pj = __expf(s[j] - mx); den += pj; acc += pj * vv[j]; }
            }
            o[(row0 + i * seq_stride) * ld_o + h * 64 + lane] = f2bf(acc / den);
        }
    }
}


// Fast path of the tiny attention for 8 heads x 64 dims per wave: lane = (head, 8-element chunk), so one wave reads
// whole 1-KiB q/k/v rows (fully coalesced), the per-(i,j) dot product is 4 x v_dot2c_f32_bf16 + a 3-step DPP
// reduction inside each 8-lane head group, and P.V stays lane-local.  Exact sequence length L is a template
// parameter (K and V of the whole sequence live in registers as packed bf16).
__device__ __forceinline__ float reduce8(float v) {
    int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, t);
    t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, t);
    t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true);      // row_half_mirror
    return v + __builtin_bit_cast(float, t);
}

template <int L>
__global__ __launch_bounds__(256) void attn_tiny8_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                         const bf16_t* __restrict__ v, bf16_t* __restrict__ o,
                                                         int hgroups, long long n_outer, long long n_inner,
                                                         long long outer_stride, long long seq_stride, long long ld_qkv,
                                                         long long ld_o, float scale) {
    const int lane = threadIdx.x & 63;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= n_outer * n_inner * hgroups) return;
    const int hgrp = (int)(wid % hgroups);
    const long long gseq = wid / hgroups;
    const long long row0 = (gseq / n_inner) * outer_stride + (gseq % n_inner);
    const int col = hgrp * 512 + lane * 8;
    constexpr bool Q_IN_REGS = L <= 16;         // L = 25 (97-frame clips): K and V alone are 200 registers
    uint32_t kq[L][4], vq[L][4], qq[Q_IN_REGS ? L : 1][4];      // plain scalars: bit_cast of an ext_vector element miscompiles
#pragma unroll
    for (int e = 0; e < L; ++e) {
        const long long off = (row0 + e * seq_stride) * ld_qkv + col;
        const u32x4 b = *reinterpret_cast<const u32x4*>(k + off);
        const u32x4 c = *reinterpret_cast<const u32x4*>(v + off);
#pragma unroll
        for (int w = 0; w < 4; ++w) { kq[e][w] = b[w]; vq[e][w] = c[w]; }
        if (Q_IN_REGS) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(q + off);
#pragma unroll
            for (int w = 0; w < 4; ++w) qq[e][w] = a[w];
        }
    }
#pragma unroll
    for (int i = 0; i < L; ++i) {
        if (!Q_IN_REGS) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(q + (row0 + i * seq_stride) * ld_qkv + col);
#pragma unroll
            for (int w = 0; w < 4; ++w) qq[0][w] = a[w];
        }
        float s[L];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < L; ++j) {
            float d = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w)
            {
                const uint32_t qa = qq[Q_IN_REGS ? i : 0][w], kb = kq[j][w];
                d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, qa), __builtin_bit_cast(bf16x2, kb), d,
                                                    false);
            }
            s[j] = reduce8(d) * scale;
            mx = fmaxf(mx, s[j]);
        }
        float den = 0.f, acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const float pj = __expf(s[j] - mx);
            den += pj;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                acc[2 * w] = fmaf(pj, bflo(vq[j][w]), acc[2 * w]);
                acc[2 * w + 1] = fmaf(pj, bfhi(vq[j][w]), acc[2 * w + 1]);
            }
        }
        const float inv = 1.0f / den;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] *= inv;
        *reinterpret_cast<u32x4*>(o + (row0 + i * seq_stride) * ld_o + col) = pack8(acc);
    }
}

inline int ok() { const hipError_t e = hipGetLastError(); return e == hipSuccess ? BYA_OK : -(1000 + (int)e); }

}  // namespace

extern "C" int bya_router_scores(const void* qr, const void* kr, const void* ln_w, const void* ln_b,
                                 const void* pos_emb, void* out, int32_t n_id, int64_t N, int32_t heads,
                                 int32_t face_tokens, float eps, hipStream_t stream) {
    if (!qr || !kr || !ln_w || !ln_b || !pos_emb || !out || n_id <= 0 || N <= 0) return BYA_ERR_SHAPE;
    if (heads != R_HEADS || face_tokens != R_TOK) return BYA_ERR_UNSUPPORTED;
    if (((uintptr_t)qr | (uintptr_t)kr | (uintptr_t)ln_w | (uintptr_t)ln_b | (uintptr_t)pos_emb | (uintptr_t)out) & 15)
        return BYA_ERR_ALIGN;
    if (N >= 4096 && n_id <= 256 && !bya_ref_form(BYA_REF_ROUTER_SCORES_WAVE)) {
        static std::atomic<unsigned long long> big{0};
        if (bya_allow_big_lds(reinterpret_cast<const void*>(router_scores_lds_kernel), 160 * 1024, big) != BYA_OK) return BYA_ERR_LAUNCH;
        const int grid = 256 / n_id * n_id;                      // one workgroup per CU, whole identities
        BYA_LAUNCH(router_scores_lds_kernel, dim3((unsigned)grid), dim3(512), (size_t)R_TOK * R_QK * 2, stream,
                   (const bf16_t*)qr, (const bf16_t*)kr, (const bf16_t*)ln_w, (const bf16_t*)ln_b,
                   (const bf16_t*)pos_emb, (bf16_t*)out, n_id, (long long)N, eps);
        return ok();
    }
    const long long waves = ((N + 15) / 16) * n_id;
    BYA_LAUNCH(router_scores_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream,
                       (const bf16_t*)qr, (const bf16_t*)kr, (const bf16_t*)ln_w, (const bf16_t*)ln_b,
                       (const bf16_t*)pos_emb, (bf16_t*)out, n_id, (long long)N, eps);
    return ok();
}

extern "C" int bya_router_head(const void* x, const void* w, const void* b, void* r, int32_t n_id, int64_t N,
                               int32_t D, hipStream_t stream) {
    if (!x || !w || !b || !r || n_id <= 0 || N <= 0 || D <= 0 || D % 512) return BYA_ERR_SHAPE;
    if (((uintptr_t)x | (uintptr_t)w) & 15) return BYA_ERR_ALIGN;
    const long long rows = (long long)N * n_id;
    BYA_LAUNCH(router_head_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (const bf16_t*)x,
                       (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)r, n_id, (long long)N, D);
    return ok();
}

extern "C" int bya_attn_tiny(const void* q, const void* k, const void* v, void* o, int32_t L, int32_t heads,
                             int64_t n_outer, int64_t n_inner, int64_t outer_stride, int64_t seq_stride,
                             int64_t ld_qkv, int64_t ld_o, float scale, hipStream_t stream) {
    if (!q || !k || !v || !o || L <= 0 || L > 32 || heads <= 0 || n_outer <= 0 || n_inner <= 0) return BYA_ERR_SHAPE;
#define TINY_ARGS (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o
#define TINY_TAIL (long long)n_outer, (long long)n_inner, (long long)outer_stride, (long long)seq_stride, \
                  (long long)ld_qkv, (long long)ld_o, scale
    // fast path: the sequence lengths of the router (frames per clip: 13 at 49 frames, 25 at 97; identities: 2, 3)
    if (heads % 8 == 0 && (L == 2 || L == 3 || L == 13 || L == 25) && ld_qkv % 8 == 0 && ld_o % 8 == 0 &&
        !(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15)) {
        const int hgroups = heads / 8;
        const long long w8 = (long long)n_outer * n_inner * hgroups;
        dim3 g8((unsigned)((w8 + 3) / 4));
        if (L == 2) BYA_LAUNCH((attn_tiny8_kernel<2>), g8, dim3(256), 0, stream, TINY_ARGS, hgroups, TINY_TAIL);
        else if (L == 3) BYA_LAUNCH((attn_tiny8_kernel<3>), g8, dim3(256), 0, stream, TINY_ARGS, hgroups, TINY_TAIL);
        else if (L == 13) BYA_LAUNCH((attn_tiny8_kernel<13>), g8, dim3(256), 0, stream, TINY_ARGS, hgroups, TINY_TAIL);
        else BYA_LAUNCH((attn_tiny8_kernel<25>), g8, dim3(256), 0, stream, TINY_ARGS, hgroups, TINY_TAIL);
        return ok();
    }
    const long long waves = (long long)n_outer * n_inner * heads;
    dim3 grid((unsigned)((waves + 3) / 4));
    if (L <= 2) BYA_LAUNCH((attn_tiny_kernel<2>), grid, dim3(256), 0, stream, TINY_ARGS, L, heads, TINY_TAIL);
    else if (L <= 4) BYA_LAUNCH((attn_tiny_kernel<4>), grid, dim3(256), 0, stream, TINY_ARGS, L, heads, TINY_TAIL);
    else if (L <= 16) BYA_LAUNCH((attn_tiny_kernel<16>), grid, dim3(256), 0, stream, TINY_ARGS, L, heads, TINY_TAIL);
    else BYA_LAUNCH((attn_tiny_kernel<32>), grid, dim3(256), 0, stream, TINY_ARGS, L, heads, TINY_TAIL);
#undef TINY_ARGS
#undef TINY_TAIL
    return ok();
}
