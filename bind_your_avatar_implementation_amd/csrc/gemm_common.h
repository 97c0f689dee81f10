// Shared by the GEMM translation units (gemm.hip, gemm_w4.hip): argument block, fused epilogue, LDS read helper.
#pragma once
#include "bya_common.h"
#include "../../include/bya.h"

namespace {

struct GemmArgs {
    const bf16_t* A; const bf16_t* W; const bf16_t* bias; bf16_t* C; const bf16_t* res;
    const bf16_t* gate0; const bf16_t* gate1;
    int M, N, K;
    int lda, ldw, ldc, ldres;
    long long a_bs, c_bs, res_bs, gate_bs;
    int gate_split;
    int act;
    float leaky;
    int n_split;            // > 0: output column n goes to C + (n / n_split) * c_split_stride, column n % n_split
    long long c_split_stride;
    const float* bias_rowscale;   // optional fp32 [batch*M]: bias is multiplied by bias_rowscale[z*M + m]
    float alpha;                  // scales (acc + bias) after the activation (local_face_scale)
};

constexpr int BK = 64;  // bf16 elements per K tile = 128-byte LDS rows

template <int ACT>
__device__ __forceinline__ float apply_act(float v, float leaky) {
    if constexpr (ACT == 1) return gelu_tanh(v);
    else if constexpr (ACT == 2) return gelu_erf(v);
    else if constexpr (ACT == 3) return v > 0.f ? v : 0.f;
    else if constexpr (ACT == 4) return silu(v);
    else if constexpr (ACT == 5) return v > 0.f ? v : v * leaky;
    else return v;
}


// Epilogue for 4 consecutive output columns n4..n4+3 of row m: + bias -> act -> * gate[row type] -> + residual -> bf16
template <int V> struct IntTag { static constexpr int value = V; };
template <typename F>
__device__ __forceinline__ void dispatch_act(int act, F&& f) {
    switch (act) {
        case 1: f(IntTag<1>{}); break;
        case 2: f(IntTag<2>{}); break;
        case 3: f(IntTag<3>{}); break;
        case 4: f(IntTag<4>{}); break;
        case 5: f(IntTag<5>{}); break;
        default: f(IntTag<0>{}); break;
    }
}

template <int ACT>
__device__ __forceinline__ void epilogue4(const GemmArgs& p, int z, int m, int n4, const f32x4 acc, const float (&b4)[4]) {
    float v[4];
    const float bs = p.bias_rowscale ? p.bias_rowscale[(long long)z * p.M + m] : 1.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = p.alpha * apply_act<ACT>(fmaf(bs, b4[e], acc[e]), p.leaky);
    if (p.gate0) {
        const bf16_t* g = (m < p.gate_split ? p.gate0 : p.gate1) + (long long)z * p.gate_bs + n4;
        const u32x2 gv = *reinterpret_cast<const u32x2*>(g);
        v[0] *= bflo(gv[0]); v[1] *= bfhi(gv[0]); v[2] *= bflo(gv[1]); v[3] *= bfhi(gv[1]);
    }
    long long col = n4;
    if (p.n_split > 0) col = (long long)(n4 / p.n_split) * p.c_split_stride + (n4 % p.n_split);
    if (p.res) {
        const u32x2 rv = *reinterpret_cast<const u32x2*>(p.res + (long long)z * p.res_bs + (long long)m * p.ldres + n4);
        v[0] += bflo(rv[0]); v[1] += bfhi(rv[0]); v[2] += bflo(rv[1]); v[3] += bfhi(rv[1]);
    }
    u32x2 o;
    o[0] = pack2bf(v[0], v[1]);
    o[1] = pack2bf(v[2], v[3]);
    *reinterpret_cast<u32x2*>(p.C + (long long)z * p.c_bs + (long long)m * p.ldc + col) = o;
}

template <int OFF>
__device__ __forceinline__ void ds_read128(bf16x8& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
}  // namespace

// defined in gemm_w4.hip (compiled with its own register-allocation flags), called from bya_gemm_bf16
int bya_launch_gemm256w4(const void* args, int batch, hipStream_t stream);
