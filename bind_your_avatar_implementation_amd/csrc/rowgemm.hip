// Row-stationary GEMM for the Embedding Router's 512-wide Linears (reference models/router.py:468-493, the twelve
// projections + MLP of every SpatialTemporalAttentionBlock):   C[M,N] = epi( LN?(X)[M,512] . W[N,512]^T ).
//
// K = 512 is only eight 64-wide K-tiles: a tiled GEMM spends its time on pipeline fill/drain and load latency, not on
// the matrix core (measured 390-545 TFLOP/s).  Here a wave keeps its 32 rows of X -- all 512 K -- in 128 VGPRs as MFMA
// operands (loaded once, straight from global memory in fragment layout, no LDS), and streams 32-column chunks of W
// (32 KiB each, one 1-KiB W row per LDS-DMA instruction) through a 2-stage LDS ring shared by the block's 4 waves.
// The product is computed transposed (W fragment = A operand, X fragment = B operand), so a lane ends up with 8
// consecutive output columns of one token: 16-byte stores, no LDS transpose.  The epilogue of chunk c-1 (residual
// loaded one chunk earlier) runs while the LDS-DMA of chunk c+1 is in flight; every manual vmcnt(0) therefore only
// waits for operations that were issued a whole chunk period before.
//
// Fused LayerNorm (the nn.LayerNorm in front of every q|k|v projection and of the MLP): gamma is folded into the
// weights and beta into the bias at pack time,
//     LN(x) . W^T + b  =  rstd * ( x . Wg^T  -  mean * s )  +  c,     Wg = W * gamma,  s = rowsum(Wg),  c = W . beta + b,
// so the GEMM runs on the RAW rows and the epilogue applies the two row statistics.  The statistics come from the
// matrix core as well: sum(x) = ones . x^T and sum(x^2) = diag(x . x^T), 64 extra MFMAs per wave instead of ~3000
// VALU cycles (fp32 accumulation of exact bf16 products).
#include "bya_common.h"
#include "../../include/bya.h"

namespace {

struct RowGemmArgs {
    const bf16_t* X; const bf16_t* W; const float* colsum; const float* cvec; const bf16_t* res; bf16_t* C;
    int M, N, ldx, ldc, ldres, nsplit;
    float eps;
};

constexpr int RK = 512;                   // K
constexpr int CH = 32;                    // output columns per chunk
constexpr int STAGE_BYTES = CH * RK * 2;  // 32 KiB

template <int OFF>
__device__ __forceinline__ void lds_read_w(bf16x8& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_read_f(f32x4& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N>
__device__ __forceinline__ void lgkm_wait(bf16x8& a, bf16x8& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "i"(N));
}
// both W fragments (column blocks j = 0, 1) of k-step 4*KH + kl; kl selects the lane-constant address
template <int STAGE, int KH>
__device__ __forceinline__ void read_pair(bf16x8 (&wf)[2], uint32_t addr) {
    lds_read_w<STAGE * STAGE_BYTES + KH * 256>(wf[0], addr);
    lds_read_w<STAGE * STAGE_BYTES + 16 * 1024 + KH * 256>(wf[1], addr);
}
template <int STAGE>
__device__ __forceinline__ void read_kstep(bf16x8 (&wf)[2], const uint32_t (&wa)[4], int ks) {
    switch (ks >> 2) {            // ks is a constant after unrolling: the switch folds away
        case 0: read_pair<STAGE, 0>(wf, wa[ks & 3]); break;
        case 1: read_pair<STAGE, 1>(wf, wa[ks & 3]); break;
        case 2: read_pair<STAGE, 2>(wf, wa[ks & 3]); break;
        default: read_pair<STAGE, 3>(wf, wa[ks & 3]); break;
    }
}

__device__ __forceinline__ float gelu_erf_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }

// One 32-column chunk: 16 k-steps x (2 W fragments) x (2 row blocks) = 64 MFMAs; the fragment reads of k-step ks+1
// are in flight while the MFMAs of k-step ks run.
template <int STAGE>
__device__ __forceinline__ void chunk_mfma(f32x4 (&acc)[2][2], const bf16x8 (&xf)[2][16], const uint32_t (&wa)[4]) {
    bf16x8 wf[2][2];
    read_kstep<STAGE>(wf[0], wa, 0);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < 16) {
            read_kstep<STAGE>(wf[cur ^ 1], wa, ks + 1);
            lgkm_wait<2>(wf[cur][0], wf[cur][1]);
        } else {
            lgkm_wait<0>(wf[cur][0], wf[cur][1]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
                acc[rb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cur][j], xf[rb][ks], acc[rb][j], 0, 0, 0);
    }
}

template <bool LN, bool RES, int ACT>
__global__ __launch_bounds__(256, 2) void rowgemm512_kernel(RowGemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = lane & 15, g = lane >> 4;

    // block -> (row block, column group); the column groups of one row block sit on ONE XCD (ids b, b+8, ...)
    const int nrb = (p.M + 127) / 128;
    const int xcd = blockIdx.x & 7, in_x = blockIdx.x >> 3;
    const int cg = in_x % p.nsplit, rb_id = (in_x / p.nsplit) * 8 + xcd;
    if (rb_id >= nrb) return;
    const int ncols = p.N / p.nsplit, nb0 = cg * ncols, nchunks = ncols / CH;
    const int r0 = rb_id * 128 + wave * 32;

    // LDS: [colsum: ncols f32][cvec: ncols f32][ring: 2 x 32 KiB]
    float* s_lds = reinterpret_cast<float*>(smem);
    float* c_lds = s_lds + ncols;
    char* ring = smem + ncols * 8;
    for (int i = tid; i < ncols; i += 256) {
        s_lds[i] = LN ? p.colsum[nb0 + i] : 0.0f;
        c_lds[i] = p.cvec[nb0 + i];
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.X, 0, (int)(((long long)(p.M - 1) * p.ldx + RK) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, p.N * RK * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.C, 0, (int)(((long long)(p.M - 1) * p.ldc + p.N) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(RES ? p.res : p.C), 0, (int)(((long long)(p.M - 1) * (RES ? p.ldres : p.ldc) + p.N) * 2), 0x00020000);

    // ---- W chunk staging: LDS row R = j*16 + i holds output column  chunk0 + 8*(i>>2) + 4*j + (i&3), its 64 16-byte
    // pieces XOR-swizzled with i so that the 16 rows read by one fragment instruction hit 16 different bank groups.
    uint32_t wvo[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int R = wave * 8 + q, i = R & 15, j = R >> 4;
        wvo[q] = (uint32_t)(8 * (i >> 2) + 4 * j + (i & 3)) * (RK * 2) + (uint32_t)((lane ^ i) << 4);
    }
    auto stage_chunk = [&](int c) {
        char* dst = ring + (c & 1) * STAGE_BYTES + wave * 8 * 1024;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(dst + q * 1024), 16, wvo[q],
                                                     (nb0 + c * CH) * (RK * 2), 0, 0);
    };

    // ---- X fragments: rows r0 + rb*16 + t, k = ks*32 + g*8 .. +8 (rows >= M read as zeros through the descriptor)
    bf16x8 xf[2][16];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const uint32_t vo = (uint32_t)(r0 + rb * 16 + t) * (uint32_t)(p.ldx * 2) + g * 16;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
            xf[rb][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsX, vo + ks * 64, 0, 0));
    }
    stage_chunk(0);

    // residual of a chunk: 8 consecutive columns of the lane's token per row block
    auto load_res = [&](int c, u32x4 (&rv)[2]) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
            rv[rb] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rsR, (uint32_t)(r0 + rb * 16 + t) * (uint32_t)(p.ldres * 2) + (uint32_t)(nb0 + c * CH + 8 * g) * 2, 0, 0));
    };

    // ---- row statistics on the matrix core
    float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
    if (LN) {
        bf16x8 ones;
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            f32x4 sm = {0.f, 0.f, 0.f, 0.f}, gr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, xf[rb][ks], sm, 0, 0, 0);
                gr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[rb][ks], xf[rb][ks], gr, 0, 0, 0);
            }
            // lane (g, t) holds <x_{4g+e}, x_t>; the diagonal of token t sits in lane (t>>2, t), register t&3
            const int e = t & 3;
            const float d = e == 0 ? gr[0] : e == 1 ? gr[1] : e == 2 ? gr[2] : gr[3];
            const float sq = __shfl(d, t + 16 * (t >> 2));
            const float mu = sm[0] * (1.0f / RK);
            const float var = fmaxf(sq * (1.0f / RK) - mu * mu, 0.0f);
            mean[rb] = mu;
            rstd[rb] = rsqrtf(var + p.eps);
        }
    }

    // lane-constant fragment read addresses: LDS row t of a column block, 16-byte piece (4*ks + g) ^ t
    uint32_t wa[4];
    const uint32_t ring_base = (uint32_t)(uintptr_t)LDS_PTR(ring);
#pragma unroll
    for (int m = 0; m < 4; ++m)
        wa[m] = ring_base + t * 1024 + (((g ^ (t & 3)) | ((m ^ (t >> 2)) << 2)) << 4);
    const uint32_t sc_addr = (uint32_t)(uintptr_t)LDS_PTR(smem) + g * 32;      // 8 floats per lane group per chunk

    auto epilogue = [&](int c, const f32x4 (&acc)[2][2], const u32x4 (&rv)[2]) {
        const int ncol = nb0 + c * CH + 8 * g;
        f32x4 s0, s1, c0, c1;
        const uint32_t a = sc_addr + c * (CH * 4), ac = a + ncols * 4;
        lds_read_f<0>(s0, a);
        lds_read_f<16>(s1, a);
        lds_read_f<0>(c0, ac);
        lds_read_f<16>(c1, ac);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s0), "+v"(s1), "+v"(c0), "+v"(c1));
        const float sv[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
        const float cv[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = acc[rb][0][e];
                v[4 + e] = acc[rb][1][e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float y = LN ? rstd[rb] * (v[e] - mean[rb] * sv[e]) + cv[e] : v[e] + cv[e];
                if (ACT == BYA_ACT_GELU_ERF) y = gelu_erf_f(y);
                v[e] = y;
            }
            if (RES) {
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    v[2 * h] += bflo(rv[rb][h]);
                    v[2 * h + 1] += bfhi(rv[rb][h]);
                }
            }
            u32x4 o;
#pragma unroll
            for (int h = 0; h < 4; ++h) o[h] = pack2bf(v[2 * h], v[2 * h + 1]);
            __builtin_amdgcn_raw_buffer_store_b128(o, rsC, (uint32_t)(r0 + rb * 16 + t) * (uint32_t)(p.ldc * 2) +
                                                               (uint32_t)ncol * 2, 0, 0);
        }
    };

    // ---- chunk loop, unrolled by two (LDS stage and accumulator set alternate)
    f32x4 accA[2][2], accB[2][2];
    u32x4 resA[2] = {}, resB[2] = {};
    auto zero = [](f32x4 (&a)[2][2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) a[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    for (int c = 0; c < nchunks; c += 2) {
        // -- even chunk c: stage 0, accumulators A; finishes odd chunk c-1 (accumulators B)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (RES) load_res(c, resA);
        stage_chunk(c + 1);                      // nchunks is even: chunk c+1 always exists
        if (c > 0) epilogue(c - 1, accB, resB);
        zero(accA);
        chunk_mfma<0>(accA, xf, wa);
        // -- odd chunk c+1: stage 1, accumulators B; finishes chunk c
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (RES) load_res(c + 1, resB);
        if (c + 2 < nchunks) stage_chunk(c + 2);
        epilogue(c, accA, resA);
        zero(accB);
        chunk_mfma<1>(accB, xf, wa);
    }
    epilogue(nchunks - 1, accB, resB);
}

template <bool LN, bool RES, int ACT>
int launch_rowgemm(const RowGemmArgs& a, hipStream_t s) {
    const int nrb = (a.M + 127) / 128;
    const int blocks = ((nrb + 7) / 8) * 8 * a.nsplit;
    const size_t lds = (size_t)(a.N / a.nsplit) * 8 + 2 * STAGE_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm512_kernel<LN, RES, ACT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
            return BYA_ERR_LAUNCH;
        attr_set = true;
    }
    BYA_LAUNCH((rowgemm512_kernel<LN, RES, ACT>), dim3(blocks), dim3(256), lds, s, a);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

}  // namespace

extern "C" int bya_rowgemm512(const void* X, const void* W, const float* colsum, const float* cvec, const void* res,
                              void* C, int32_t M, int32_t N, int32_t ldx, int32_t ldc, int32_t ldres, int32_t ln,
                              float eps, int32_t act, int32_t nsplit, hipStream_t stream) {
    if (!X || !W || !cvec || !C || M <= 0 || N <= 0) return BYA_ERR_SHAPE;
    if (ln && !colsum) return BYA_ERR_SHAPE;
    if (nsplit <= 0) {
        // enough blocks to give every CU several, as long as a block keeps >= 4 chunks to amortise its X rows
        nsplit = 1;
        const int nrb = (M + 127) / 128;
        while (nrb * nsplit < 1024 && N % (nsplit * 2 * 64) == 0 && N / (nsplit * 2) >= 128) nsplit *= 2;
    }
    if (N % (nsplit * 64) != 0 || N / nsplit > 2048) return BYA_ERR_SHAPE;
    if (ldx < RK || ldc < N || (res && ldres < N) || ldx % 8 || ldc % 8 || (res && ldres % 8)) return BYA_ERR_ALIGN;
    if (((uintptr_t)X | (uintptr_t)W | (uintptr_t)C | (uintptr_t)res) & 15) return BYA_ERR_ALIGN;
    if ((long long)M * ldx * 2 >= (1LL << 31) || (long long)M * ldc * 2 >= (1LL << 31)) return BYA_ERR_SHAPE;
    if (act != BYA_ACT_NONE && act != BYA_ACT_GELU_ERF) return BYA_ERR_UNSUPPORTED;
    RowGemmArgs a;
    a.X = (const bf16_t*)X; a.W = (const bf16_t*)W; a.colsum = colsum; a.cvec = cvec; a.res = (const bf16_t*)res;
    a.C = (bf16_t*)C; a.M = M; a.N = N; a.ldx = ldx; a.ldc = ldc; a.ldres = res ? ldres : ldc; a.nsplit = nsplit;
    a.eps = eps;
    const bool gelu = act == BYA_ACT_GELU_ERF;
    if (ln) {
        if (res) return gelu ? launch_rowgemm<true, true, BYA_ACT_GELU_ERF>(a, stream)
                             : launch_rowgemm<true, true, BYA_ACT_NONE>(a, stream);
        return gelu ? launch_rowgemm<true, false, BYA_ACT_GELU_ERF>(a, stream)
                    : launch_rowgemm<true, false, BYA_ACT_NONE>(a, stream);
    }
    if (res) return gelu ? launch_rowgemm<false, true, BYA_ACT_GELU_ERF>(a, stream)
                         : launch_rowgemm<false, true, BYA_ACT_NONE>(a, stream);
    return gelu ? launch_rowgemm<false, false, BYA_ACT_GELU_ERF>(a, stream)
                : launch_rowgemm<false, false, BYA_ACT_NONE>(a, stream);
}
