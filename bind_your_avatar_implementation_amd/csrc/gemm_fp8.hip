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
#include "gemm_common.h"
#include <stdlib.h>

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int BK8 = 128;          // fp8 elements (= bytes) per K-tile
constexpr float FP8_MAX = 448.0f; // largest finite e4m3fn

template <int ROWS, int NWAVES>
__device__ __forceinline__ void stage_tile8(const uint8_t* __restrict__ src, int ld, int row0, int row_max, int k0,
                                            char* lds_tile, int wave, int lane) {
    // ROWS x 128 bytes, 8 rows (1 KiB) per wave-instruction; 16-byte chunk c of row r lands at chunk c ^ ((r >> 1) & 7)
    constexpr int PER_WAVE = ROWS / NWAVES;
#pragma unroll
    for (int q = 0; q < PER_WAVE / 8; ++q) {
        const int rbase = wave * PER_WAVE + q * 8;
        const int rl = rbase + (lane >> 3);
        const int chunk = (lane & 7) ^ ((rl >> 1) & 7);
        int gr = row0 + rl;
        gr = gr < row_max ? gr : row_max;
        const uint8_t* g = src + (long long)gr * ld + k0 + chunk * 16;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(g), LDS_PTR(lds_tile + rbase * 128), 16, 0, 0);
    }
}

// the 32 bytes k = 32 g .. 32 g + 31 of one row (A and W use the same lane -> k map, so the products pair up)
__device__ __forceinline__ i32x8 lds_frag8(const char* tile, int row, int g) {
    const int sw = (row >> 1) & 7;
    const i32x4 lo = *reinterpret_cast<const i32x4*>(tile + row * 128 + (((2 * g) ^ sw) << 4));
    const i32x4 hi = *reinterpret_cast<const i32x4*>(tile + row * 128 + (((2 * g + 1) ^ sw) << 4));
    return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_fp8_kernel(GemmArgs p, const float* __restrict__ sa,
                                                                          const float* __restrict__ sw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NWAVES = WAVES_M * WAVES_N;
    constexpr int TILE_A = BM * BK8, TILE_W = BN * BK8, STAGE = TILE_A + TILE_W;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MI = WM / 16, NI = WN / 16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    const int id = xcd_remap(blockIdx.x, nwg);
    constexpr int GM = 8;                                   // group-M order, as in gemm_bf16_kernel
    const int per_group = GM * tiles_n;
    const int group = id / per_group, first_m = group * GM;
    const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int in_g = id - group * per_group;
    const int tm = first_m + in_g % gsz, tn = in_g / gsz;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.z;

    const uint8_t* A = reinterpret_cast<const uint8_t*>(p.A) + (long long)z * p.a_bs;
    const uint8_t* W = reinterpret_cast<const uint8_t*>(p.W);
    const int nk = p.K / BK8;

    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * STAGE;
        stage_tile8<BM, NWAVES>(A, p.lda, m0, p.M - 1, kt * BK8, base, wave, lane);
        stage_tile8<BN, NWAVES>(W, p.ldw, n0, p.N - 1, kt * BK8, base + TILE_A, wave, lane);
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int fr = lane & 15, fq = lane >> 4;
    const int unit = 0x7f7f7f7f;                            // E8M0 block scales: 2^(127 - 127) in every byte

    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* ta = smem + (kt & 1) * STAGE;
        const char* tw = ta + TILE_A;
        i32x8 fa[MI], fw[NI];
#pragma unroll
        for (int j = 0; j < MI; ++j) fa[j] = lds_frag8(ta, wm * WM + j * 16 + fr, fq);
#pragma unroll
        for (int i = 0; i < NI; ++i) fw[i] = lds_frag8(tw, wn * WN + i * 16 + fr, fq);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw[i], fa[j], acc[i][j], 0, 0, 0, unit, 0, unit);
    }

    // ---- row and channel scales, then the common epilogue.  Lane holds C[m][n4 .. n4+3], m = m_base + 16 j,
    // n4 = n_base + 16 i  (W fragment = A operand: the 16 x 16 result is transposed, as in every GEMM kernel here)
    const int m_base = m0 + wm * WM + fr, n_base = n0 + wn * WN + fq * 4;
    float ra[MI];
#pragma unroll
    for (int j = 0; j < MI; ++j) {
        const int m = m_base + 16 * j;
        ra[j] = sa[(long long)z * p.M + (m < p.M ? m : p.M - 1)];
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n4 = n_base + 16 * i;
        const f32x4 rw = n4 < p.N ? *reinterpret_cast<const f32x4*>(sw + n4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] *= ra[j] * rw[e];
    }
    auto run = [&](auto act_tag) {
        epilogue_block<decltype(act_tag)::value, NI, MI, (NI * MI > 16 ? 1 : NI)>(p, z, m_base, n_base, acc);
    };
    dispatch_act_big(p.act, run);
}

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

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_fp8(const GemmArgs& a, const float* sa, const float* sw, int batch, hipStream_t s) {
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n, 1, batch);
    const size_t lds = 2 * (BM + BN) * BK8;
    static std::atomic<unsigned long long> attr_done{0};
    if (bya_allow_big_lds(reinterpret_cast<const void*>(gemm_fp8_kernel<BM, BN, WAVES_M, WAVES_N>), (int)lds, attr_done) != BYA_OK)
        return BYA_ERR_LAUNCH;
    BYA_LAUNCH((gemm_fp8_kernel<BM, BN, WAVES_M, WAVES_N>), grid, dim3(64 * WAVES_M * WAVES_N), lds, s, a, sa, sw);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
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
    // slower on the four DiT shapes with this simple two-barrier loop (profiles/r2_fp8_probe.txt).  BYA_FP8_TILE=256 forces it.
    const char* tile_env = getenv("BYA_FP8_TILE");
    if (tile_env && atoi(tile_env) == 256) return launch_fp8<256, 256, 2, 4>(a, a_scale, w_scale, d->batch, stream);
    return launch_fp8<128, 128, 2, 2>(a, a_scale, w_scale, d->batch, stream);
}
