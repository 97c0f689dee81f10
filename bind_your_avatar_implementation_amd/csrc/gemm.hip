// bf16 "NT" GEMM for every Linear on the denoise-step path:  C[m,n] = epi( sum_k A[m,k] * W[n,k] )
// A = activations [M,K] row-major, W = torch Linear weight [N,K] row-major (both K-contiguous, which is
// exactly the MFMA operand order: one ds_read_b128 per fragment, no transposes anywhere).
//
// Replaces the torch.nn.Linear dispatches of the reference (rows D1,D3,D4,P1,R1,R2,A1,F1,L1,L2 of
// SURVEY.md section 8a; e.g. models/transformer.py:241-260 via diffusers Attention/FeedForward,
// models/router.py:253-254,275,381-383, models/audio_model.py:253-256).
//
// Structure (v1): 128x128x64 block tile, 4 waves (2x2) each 64x64 via 4x4 v_mfma_f32_16x16x32_bf16,
// global_load_lds_dwordx4 staging into a 2-deep LDS ring, XOR-swizzled on the SOURCE address so the
// lane-linear LDS image is bank-conflict-free for the ds_read_b128 fragment reads, XCD-aware + group-M
// block order for L2 reuse.  The MFMA is issued "swapped" (W rows as the A operand) so each lane ends
// up with 4 consecutive output columns of one output row -> 8-byte epilogue accesses.
// Epilogue: + bias -> activation -> * gate[row-type] -> + residual -> bf16.
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
};

constexpr int BK = 64;  // bf16 elements per K tile = 128-byte LDS rows

template <int ROWS>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ src, int ld, int row0, int row_max, int k0,
                                           char* lds_tile, int wave, int lane) {
    // ROWS x 64 bf16 tile, 8 rows (1 KiB) per wave-instruction, ROWS/32 instructions per wave.
    constexpr int PER_WAVE = ROWS / 4;
#pragma unroll
    for (int q = 0; q < PER_WAVE / 8; ++q) {
        const int rbase = wave * PER_WAVE + q * 8;
        const int rl = rbase + (lane >> 3);
        const int chunk = (lane & 7) ^ ((rl >> 1) & 7);
        int gr = row0 + rl;
        gr = gr < row_max ? gr : row_max;
        const bf16_t* g = src + (long long)gr * ld + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(g), LDS_PTR(lds_tile + rbase * 128), 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 lds_frag(const char* tile, int row, int chunk) {
    const int off = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    return *reinterpret_cast<const bf16x8*>(tile + off);
}

__device__ __forceinline__ float apply_act(float v, int act, float leaky) {
    switch (act) {
        case 1: return gelu_tanh(v);
        case 2: return gelu_erf(v);
        case 3: return v > 0.f ? v : 0.f;
        case 4: return silu(v);
        case 5: return v > 0.f ? v : v * leaky;
        default: return v;
    }
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE_A = BM * BK * 2, TILE_W = BN * BK * 2, STAGE = TILE_A + TILE_W;
    constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 16, NI = WN / 16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int id = xcd_remap(blockIdx.x, nwg);
    // group-M order: 8 row-tiles sweep one column-tile before moving on (A and W both reused from L2)
    constexpr int GM = 8;
    const int per_group = GM * tiles_n;
    const int group = id / per_group, first_m = group * GM;
    const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int in_g = id - group * per_group;
    const int tm = first_m + in_g % gsz, tn = in_g / gsz;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.z;

    const bf16_t* A = p.A + (long long)z * p.a_bs;
    const int nk = p.K / BK;

    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * STAGE;
        stage_tile<BM>(A, p.lda, m0, p.M - 1, kt * BK, base, wave, lane);
        stage_tile<BN>(p.W, p.ldw, n0, p.N - 1, kt * BK, base + TILE_A, wave, lane);
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;

    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* ta = smem + (kt & 1) * STAGE;
        const char* tw = ta + TILE_A;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[MI], fw[NI];
#pragma unroll
            for (int j = 0; j < MI; ++j) fa[j] = lds_frag(ta, wm * WM + j * 16 + fr, ks * 4 + fq);
#pragma unroll
            for (int i = 0; i < NI; ++i) fw[i] = lds_frag(tw, wn * WN + i * 16 + fr, ks * 4 + fq);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: lane holds C[m][n4..n4+3], m = m0 + wm*WM + j*16 + fr, n4 = n0 + wn*WN + i*16 + fq*4
    bf16_t* C = p.C + (long long)z * p.c_bs;
    const bf16_t* R = p.res ? p.res + (long long)z * p.res_bs : nullptr;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n4 = n0 + wn * WN + i * 16 + fq * 4;
        if (n4 >= p.N) continue;
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
            const u32x2 bv = *reinterpret_cast<const u32x2*>(p.bias + n4);
            b4[0] = bflo(bv[0]); b4[1] = bfhi(bv[0]); b4[2] = bflo(bv[1]); b4[3] = bfhi(bv[1]);
        }
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const int m = m0 + wm * WM + j * 16 + fr;
            if (m >= p.M) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act(acc[i][j][e] + b4[e], p.act, p.leaky);
            if (p.gate0) {
                const bf16_t* g = (m < p.gate_split ? p.gate0 : p.gate1) + (long long)z * p.gate_bs + n4;
                const u32x2 gv = *reinterpret_cast<const u32x2*>(g);
                v[0] *= bflo(gv[0]); v[1] *= bfhi(gv[0]); v[2] *= bflo(gv[1]); v[3] *= bfhi(gv[1]);
            }
            if (R) {
                const u32x2 rv = *reinterpret_cast<const u32x2*>(R + (long long)m * p.ldres + n4);
                v[0] += bflo(rv[0]); v[1] += bfhi(rv[0]); v[2] += bflo(rv[1]); v[3] += bfhi(rv[1]);
            }
            u32x2 o;
            o[0] = pack2bf(v[0], v[1]);
            o[1] = pack2bf(v[2], v[3]);
            *reinterpret_cast<u32x2*>(C + (long long)m * p.ldc + n4) = o;
        }
    }
}

template <int BM, int BN>
int launch(const GemmArgs& a, int batch, hipStream_t s) {
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n, 1, batch);
    const size_t lds = 2 * (BM + BN) * BK * 2;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<BM, BN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return BYA_ERR_LAUNCH;
        attr_set = true;
    }
    BYA_LAUNCH((gemm_bf16_kernel<BM, BN>), grid, dim3(256), lds, s, a);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

}  // namespace

extern "C" int bya_gemm_bf16(const void* A, const void* W, const void* bias, void* C, const void* res,
                             const void* gate0, const void* gate1, const bya_gemm_desc* d, hipStream_t stream) {
    if (!A || !W || !C || !d) return BYA_ERR_SHAPE;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0) return BYA_ERR_SHAPE;
    if (d->K % BK != 0 || d->N % 4 != 0) return BYA_ERR_SHAPE;
    if (d->lda % 8 || d->ldw % 8 || d->ldc % 4 || (res && d->ldres % 4)) return BYA_ERR_ALIGN;
    if (((uintptr_t)A | (uintptr_t)W) & 15) return BYA_ERR_ALIGN;
    if (((uintptr_t)C | (uintptr_t)res | (uintptr_t)bias | (uintptr_t)gate0 | (uintptr_t)gate1) & 7) return BYA_ERR_ALIGN;
    if (d->act < 0 || d->act > 5) return BYA_ERR_UNSUPPORTED;
    GemmArgs a;
    a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.bias = (const bf16_t*)bias; a.C = (bf16_t*)C;
    a.res = (const bf16_t*)res; a.gate0 = (const bf16_t*)gate0; a.gate1 = (const bf16_t*)(gate1 ? gate1 : gate0);
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.lda = d->lda; a.ldw = d->ldw; a.ldc = d->ldc; a.ldres = d->ldres;
    a.a_bs = d->a_batch_stride; a.c_bs = d->c_batch_stride; a.res_bs = d->res_batch_stride;
    a.gate_bs = d->gate_batch_stride; a.gate_split = d->gate_split; a.act = d->act; a.leaky = 0.01f;
    if (d->N <= 64) return launch<128, 64>(a, d->batch, stream);
    return launch<128, 128>(a, d->batch, stream);
}
