// Shared by the GEMM translation units (gemm.hip, gemm_v4.hip, gemm_fp8.hip): argument block, fused epilogue, LDS read helper.
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
    // split-K workspace of the persistent kernel (gemm_v4.hip; bya_set_gemm_workspace): 256 slabs of 256 x 256 fp32 behind
    // 1024 counters; null = the last, partial round of tiles is not split
    float* ws_slabs;
    unsigned* ws_counters;
    int gm = 4;                   // persistent kernel: group-M width of the tile order (gemm_v4.hip; chosen per shape by its launchers)
    // Implicit-GEMM 3 x 3 x 3 convolution on the persistent kernel (bya_vae_conv3d; gemm_v4.hip, CONV instance): A is the
    // zero-padded channels-last input [To + 2, Hp, Wp, C] seen as a matrix of pixels x C (lda = C), row m of the product is
    // the padded pixel m of the OUTPUT grid [To, Hp, Wp] (rows with h >= H or w >= W are computed and dropped), K-tile t is
    // the 64-channel group t % cpg of tap t / cpg = (dt, dh, dw): the same rows, ((dt Hp + dh) Wp + dw) pixels further on.
    // q/k LayerNorm(64) + RoPE in the packed q|k|v projection's epilogue (bya_gemm_qkv_norm_rope; gemm_v4.hip, QKN instance):
    // columns [0, qkn_width) are q, [qkn_width, 2 qkn_width) k, the rest v (untouched); per 64-column head row the arithmetic
    // of qknorm_math.h on the bf16-ROUNDED projection (what the two-launch path reads back), rows >= qkn_text_rows rotated
    const bf16_t* qkn_w[2] = {nullptr, nullptr};
    const bf16_t* qkn_b[2] = {nullptr, nullptr};
    const float* qkn_cos = nullptr; const float* qkn_sin = nullptr;
    int qkn_text_rows = 0, qkn_width = 0;
    float qkn_eps = 0.f, qkn_kscale = 1.f;
    int conv_cpg_log2 = -1;            // log2(C / 64); < 0: plain GEMM
    int conv_Hp = 0, conv_Wp = 0, conv_H = 0, conv_W = 0, conv_To = 0;
    long long conv_a_bytes = 0;        // bytes of the padded input from A on (reads past it return zeros)
};

constexpr size_t GEMM_WS_COUNTER_BYTES = 4096, GEMM_WS_SLAB_BYTES = 256 * 256 * 4, GEMM_WS_SLABS = 256;
constexpr size_t GEMM_WS_BYTES = GEMM_WS_COUNTER_BYTES + GEMM_WS_SLABS * GEMM_WS_SLAB_BYTES;

constexpr int BK = 64;  // bf16 elements per K tile = 128-byte LDS rows

template <int V> struct IntTag { static constexpr int value = V; };

// The pipelined 256x256 kernels instantiate their (fully unrolled, 128-accumulator) epilogue for these activations only --
// none, GELU(tanh) of the DiT MLP and its A/B twin; every other activation sits on small GEMMs that run on the 128x128
// kernel (pick_tile), so seven copies of a 3000-line epilogue are not carried around (they also cost registers: the
// seven-way merge made hipcc copy half the accumulators).
__host__ __device__ constexpr bool act_on_big_tiles(int act) { return act == 0 || act == 1 || act == 6; }
template <typename F>
__device__ __forceinline__ void dispatch_act_big(int act, F&& f) {
    switch (act) {
        case 1: f(IntTag<1>{}); break;
        case 6: f(IntTag<6>{}); break;
        default: f(IntTag<0>{}); break;
    }
}

template <int ACT>
__device__ __forceinline__ float apply_act(float v, float leaky) {
    if constexpr (ACT == 1) return gelu_tanh(v);
    else if constexpr (ACT == 2) return gelu_erf(v);
    else if constexpr (ACT == 3) return v > 0.f ? v : 0.f;
    else if constexpr (ACT == 4) return silu(v);
    else if constexpr (ACT == 5) return v > 0.f ? v : v * leaky;
    else if constexpr (ACT == 6) return gelu_tanh_ieee(v);
    else return v;
}


// Epilogue for 4 consecutive output columns n4..n4+3 of row m: + bias -> act -> * gate[row type] -> + residual -> bf16
template <typename F>
__device__ __forceinline__ void dispatch_act(int act, F&& f) {
    switch (act) {
        case 1: f(IntTag<1>{}); break;
        case 2: f(IntTag<2>{}); break;
        case 3: f(IntTag<3>{}); break;
        case 4: f(IntTag<4>{}); break;
        case 5: f(IntTag<5>{}); break;
        case 6: f(IntTag<6>{}); break;
        default: f(IntTag<0>{}); break;
    }
}

// Fused epilogue of one wave: the lane holds C[m_j][n_i .. n_i + 3] for m_j = m_base + 16 j, n_i = n_base + 16 i
// (the transposed 16x16x32 accumulator layout of every GEMM kernel here), acc[i][j] = the four columns.
//   v = alpha * act(acc + rowscale[m] * bias[n]);  v *= gate[row type][n];  v += res[m][n];  C[m][n] = bf16(v)
// Everything the epilogue READS (bias, both gate vectors, the residual rows, the per-row bias scale) is requested up front
// in one burst, with clamped addresses instead of branches, and only then consumed: a load inside the per-tile bounds
// branch made hipcc wait for it -- and, since vmcnt counts stores too, for the previous tile's store -- once per
// 16x16 tile, 32 to 64 serialized memory round trips per wave (measured: the gate + residual epilogue cost 32 % of the
// attention-output GEMM, 780 vs 1152 TFLOP/s).  The residual may alias C (x += ...): a lane only ever reads the
// elements it later writes.
template <int ACT, int NI, int NJ, int IB = NI>
__device__ __forceinline__ void epilogue_block(const GemmArgs& p, int z, int m_base, int n_base, const f32x4 (&acc)[NI][NJ]) {
    // IB = column groups per burst: the burst's residual values occupy IB * NJ * 2 registers (the 8-wave kernel has only
    // 256 registers per lane with 128 of them accumulators, so it bursts one column group at a time).
    // Residual loads and C stores go through buffer descriptors: 32-bit per-lane byte offsets (one register per row, no
    // 64-bit pointer per access) and the hardware range check instead of branches -- an element outside M x N gets the
    // offset 0xffffffff, whose load returns zero and whose store is dropped.  The descriptors reach 2 GiB from their base:
    // the host side (gemm_row_chunks below) cuts a launch whose C or residual rows span more than that into row chunks.
    static_assert(NI % IB == 0, "burst size must divide the tile");
    const bool has_res = p.res != nullptr, has_gate = p.gate0 != nullptr, has_bias = p.bias != nullptr;
    const bool has_rs = p.bias_rowscale != nullptr;
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((has_res ? p.res : p.C) + (long long)z * p.res_bs), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.C + (long long)z * p.c_bs), 0, 0x7fffffff, 0x00020000);
    const char* g0base = reinterpret_cast<const char*>(p.gate0 + (long long)z * p.gate_bs);
    const char* g1base = reinterpret_cast<const char*>(p.gate1 + (long long)z * p.gate_bs);
    float rs[NJ];
    bool mok[NJ];
    uint32_t roff[NJ], coff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int m = m_base + 16 * j;
        mok[j] = m < p.M;
        const uint32_t mc = mok[j] ? (uint32_t)m : 0u;
        rs[j] = has_rs ? p.bias_rowscale[(long long)z * p.M + mc] : 1.0f;
        roff[j] = mok[j] ? mc * (uint32_t)(p.ldres * 2) : 0xffffffffu;
        coff[j] = mok[j] ? mc * (uint32_t)(p.ldc * 2) : 0xffffffffu;
    }
#pragma unroll
    for (int ib = 0; ib < NI; ib += IB) {
        u32x2 bv[IB], g0[IB], g1[IB], rv[IB][NJ];
        bool nok[IB];
        uint32_t ncb[IB];
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) {
            const int n4 = n_base + 16 * (ib + ii);
            nok[ii] = n4 < p.N;
            ncb[ii] = nok[ii] ? (uint32_t)n4 * 2u : 0u;
            bv[ii] = has_bias ? *reinterpret_cast<const u32x2*>(reinterpret_cast<const char*>(p.bias) + ncb[ii]) : u32x2{0u, 0u};
            if (has_gate) {
                g0[ii] = *reinterpret_cast<const u32x2*>(g0base + ncb[ii]);
                g1[ii] = *reinterpret_cast<const u32x2*>(g1base + ncb[ii]);
            }
            if (has_res) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)       // (0xffffffff + a column offset wraps: clamp invalid rows explicitly)
                    rv[ii][j] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(
                        rsR, mok[j] ? roff[j] + ncb[ii] : 0xffffffffu, 0, 0));
            }
        }
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) {
            const int n4 = n_base + 16 * (ib + ii);
            uint32_t colb = (uint32_t)n4 * 2u;
            if (p.n_split > 0) colb = ((uint32_t)(n4 / p.n_split) * (uint32_t)p.c_split_stride + (uint32_t)(n4 % p.n_split)) * 2u;
            const float b4[4] = {bflo(bv[ii][0]), bfhi(bv[ii][0]), bflo(bv[ii][1]), bfhi(bv[ii][1])};
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int m = m_base + 16 * j;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = p.alpha * apply_act<ACT>(fmaf(rs[j], b4[e], acc[ib + ii][j][e]), p.leaky);
                if (has_gate) {
                    const u32x2 gv = m < p.gate_split ? g0[ii] : g1[ii];
                    v[0] *= bflo(gv[0]); v[1] *= bfhi(gv[0]); v[2] *= bflo(gv[1]); v[3] *= bfhi(gv[1]);
                }
                if (has_res) {
                    v[0] += bflo(rv[ii][j][0]); v[1] += bfhi(rv[ii][j][0]);
                    v[2] += bflo(rv[ii][j][1]); v[3] += bfhi(rv[ii][j][1]);
                }
                u32x2 o;
                o[0] = pack2bf(v[0], v[1]);
                o[1] = pack2bf(v[2], v[3]);
                __builtin_amdgcn_raw_buffer_store_b64(o, rsC, (mok[j] && nok[ii]) ? coff[j] + colb : 0xffffffffu, 0, 0);
            }
        }
    }
}

template <int OFF>
__device__ __forceinline__ void ds_read128(bf16x8& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}

// ---- host side: the 2 GiB reach of the epilogue's buffer descriptors ---------------------------------------------------
// Every epilogue here addresses C and the residual as  base(z) + 32-bit byte offset  under num_records = 0x7fffffff: an
// element whose byte offset reaches 2^31 - 1 would be silently dropped on store and read as zero (and offsets past 4 GiB
// would wrap onto valid rows).  97 frames at 720 x 1280 are 90226 joint rows: the MLP's [90226, 12288] bf16 activation is
// 2.2 GB.  Such a launch is cut into row chunks (and batch entries) that each stay inside the reach; `run(args, batch,
// row0)` launches one piece, row0 = index of its first row in per-row side arrays ([batch * M] fp32 scales).
inline long long gemm_span_bytes(const GemmArgs& a, int M, bool res) {
    const long long ld = res ? a.ldres : a.ldc;
    const long long col = (!res && a.n_split > 0) ? (long long)(a.N / a.n_split - 1) * a.c_split_stride + a.n_split - 1
                                                  : (long long)a.N - 1;
    return ((long long)(M - 1) * ld + col) * 2 + 16;          // + the widest access of any epilogue
}

inline bool gemm_rows_reachable(const GemmArgs& a, int M) {
    constexpr long long REACH = 0x7fffffffLL;
    return gemm_span_bytes(a, M, false) < REACH && (!a.res || gemm_span_bytes(a, M, true) < REACH);
}

template <typename F>
int gemm_row_chunks(const GemmArgs& a, int batch, int a_elem_bytes, F&& run) {
    if (gemm_rows_reachable(a, a.M)) return run(a, batch, 0LL);
    int chunk = a.M;
    while (chunk > 256 && !gemm_rows_reachable(a, chunk)) chunk = ((chunk / 2 + 255) / 256) * 256;
    if (!gemm_rows_reachable(a, chunk)) return BYA_ERR_UNSUPPORTED;        // a single 256-row block out of reach: strides too large
    for (int z = 0; z < batch; ++z)
        for (int m0 = 0; m0 < a.M; m0 += chunk) {
            GemmArgs s = a;
            s.M = a.M - m0 < chunk ? a.M - m0 : chunk;
            s.A = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(a.A) +
                                                  ((long long)z * a.a_bs + (long long)m0 * a.lda) * a_elem_bytes);
            s.C = a.C + (long long)z * a.c_bs + (long long)m0 * a.ldc;
            if (a.res) s.res = a.res + (long long)z * a.res_bs + (long long)m0 * a.ldres;
            if (a.gate0) s.gate0 = a.gate0 + (long long)z * a.gate_bs;
            if (a.gate1) s.gate1 = a.gate1 + (long long)z * a.gate_bs;      // a null gate1 stays null (one gate for every row)
            if (a.bias_rowscale) s.bias_rowscale = a.bias_rowscale + (long long)z * a.M + m0;
            s.gate_split = a.gate_split > m0 ? a.gate_split - m0 : 0;
            const int rc = run(s, 1, (long long)z * a.M + m0);
            if (rc != BYA_OK) return rc;
        }
    return BYA_OK;
}
// Persistent kernels: group-M width of the tile order, in 256-row tiles, by shape: a long K sweep wants few row tiles per group
// (the W panel of a column tile is re-read by fewer row tiles, but each stays in flight longer), wide outputs want more
// (same-box sweep: tools/gemm_gm_probe.py, profiles/r6_k_gemm_gm_probe.json)
inline int gemm_group_m(const GemmArgs& a) { return a.K >= 8192 ? 2 : (a.N >= 8192 ? 8 : 4); }

}  // namespace

// defined in gemm_v4.hip (compiled with its own register-allocation flags: accumulators in AGPRs), called from
// bya_gemm_bf16: persistent one-wave-per-SIMD kernel with cross-tile prefetch and 16-byte epilogue accesses
int bya_launch_gemm256p(const void* args, int batch, hipStream_t stream);
int bya_launch_gemm256p_qkn(const void* args, int batch, hipStream_t stream);    // ... its QKN instance (bya_gemm_qkv_norm_rope)
int bya_gemm_split_min_ktiles();     // K-tiles per K-range below which the persistent kernel does not split a tile
// defined in gemm_v5.hip: the persistent kernel with 128 x 256 tiles (three-stage ring) for row counts that leave the 256 x 256
// grid half empty; callers have checked v4_eligible() and K >= 4 K-tiles
int bya_launch_gemm128p(const void* args, int batch, hipStream_t stream);
int bya_launch_gemm128p_qkn(const void* args, int batch, hipStream_t stream);
// defined in gemm_v6.hip: the same tile with loader waves (the compute waves issue no vector-memory instruction in the K-loop)
int bya_launch_gemm128s(const void* args, int batch, hipStream_t stream);
