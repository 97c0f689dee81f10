// Chains of the Embedding Router's K = 512 Linears in ONE launch, the activation between them in registers
// (reference models/router.py:476-491, SpatialTemporalAttentionBlock.forward):
//
//   router_mlp_chain_kernel        x += mlp[2]( GELU( mlp[0]( norm4(x) ) ) )                          (router.py:491)
//   router_attn_chain_kernel       x += to_out( attention over small row groups( norm(x) ) )          (router.py:476-487, the
//                                  temporal and the multi-ID sub-block: groups of 13 frames / of the identities of a token)
//
// Until round 6 each was two launches (bya_rowgemm512 ln + GELU -> bya_rowgemm512 + residual; bya_router_group_attn ->
// bya_rowgemm512 + residual) with a [rows, 512] tensor written and read back in between (36 MB each way at 35100 rows)
// and, what costs more at this size, two launches' worth of fixed work: 18 GFLOP Linears take 35-44 us each, of which
// the matrix core is busy for ~10.
//
// Why a chain fits: the row GEMMs compute the product TRANSPOSED (W fragment = A operand, the rows' fragment = B operand),
// and with the ring's row order (rowk::wrow_of) lane (g, t) ends a chunk with columns 64 c + 32 u + 8 g .. + 7 of token t --
// which IS the operand fragment of k-step 2 c + u of the next K = 512 product.  The rounded output of one Linear is the
// next one's input fragment, register for register; the rows' own fragments are the residual.  Both stay in registers:
// 64 + 64 VGPRs for a 16-row tile.
//
// Why 16 rows per wave and not 32 like rowgemm.hip: two 128-register fragment sets do not fit two waves per SIMD.  With
// one tile per wave every W fragment read from LDS feeds ONE MFMA (16 cycles) instead of two, so the chunk loop runs at the
// LDS array's rate (8 waves x 64 KiB per chunk = 2048 cycles at 256 B/clk) rather than the matrix core's -- about the
// same number, and the price of the fusion.  A workgroup = 8 waves = 128 rows streams the whole chain's weights (16 / 32
// chunks of 64 KiB through the two-stage ring of rowgemm.hip) once per PASS.
//
// Passes: 35100 rows are 2194 tiles, 8.57 per CU -- one pass of 8 tiles per workgroup leaves 146 tiles.  A second round of
// full workgroups would run on 19 CUs for as long as the first ran on 256.  Instead the launch is 256 persistent
// workgroups and the remainder is dealt over ALL of them (one tile per workgroup here): waves without a tile skip the
// fragment reads and MFMAs but keep staging W, so the short pass runs at the rate the weights arrive (L2 -> LDS,
// ~1 us per chunk) instead of the LDS-read rate of a full one.  ``tiles_pass0`` (1..8, 0 = 8) is the first pass's
// tile count per workgroup, a tuning knob of the entry points.
//
// Bit-exactness: every product is accumulated over the 16 k-steps in the same order by the same MFMA as in
// rowgemm512_kernel / rowgemm512q_kernel / rowattn512_kernel, the epilogues evaluate the same expressions on the same
// rounded values (rowgemm_common.h), so a chain equals its two launches BIT FOR BIT -- which is what lets the engine
// use the chain on one GPU and the pair on a rank's smaller shard without breaking "a shard rounds like the whole".
#include "rowgemm_common.h"
#include "../../include/bya.h"

// Timing-only ablations (tools/rowchain_ablate.py builds side copies of the library; NEVER defined in the product build;
// results are meaningless): 1 = no fragment reads / MFMAs / epilogues (what the W stream and its barriers cost alone),
// 2 = no W staging after the first chunk, 4 = GELU -> identity, 8 = no W fragment reads (MFMAs on whatever the registers
// hold), 16 = no workgroup barriers
#ifndef BYA_ROWCHAIN_ABLATE
#define BYA_ROWCHAIN_ABLATE 0
#endif

#if BYA_ROWCHAIN_ABLATE & 16
#define ROWK_BARRIER() ((void)0)
#else
#define ROWK_BARRIER() __builtin_amdgcn_s_barrier()
#endif
#if BYA_ROWCHAIN_ABLATE & 1
#define ROWK_COMPUTE(active) false
#else
#define ROWK_COMPUTE(active) (active)
#endif

namespace {

using namespace rowk;

constexpr int CW = 8;                     // waves per workgroup: two per SIMD, ONE 16-row tile each
constexpr int SRW = CH / CW;              // W rows of a chunk every wave stages
constexpr int CONST_MLP = 3 * 512 * 4;                        // colsum1 | cvec1 | cvec2
constexpr int CONST_ATTN = (2 * 1536 + 512) * 4 + CW * 32 * 4;  // colsum | cvec (q|k|v) | cvec_o | per-wave (mean[16], rstd[16])

// one 64-row W chunk into a ring stage: LDS row R = 16 j + i holds W row row0 + perm(i, j), its 64 16-byte pieces
// XOR-swizzled with i (rowgemm.hip, stage_chunk)
template <bool NATURAL>
__device__ __forceinline__ void stage_rows(const __amdgpu_buffer_rsrc_t rsW, char* stage, int wave, int row0) {
    char* dst = stage + wave * SRW * 1024;
    const uint32_t l16 = lane_now() << 4;
#if BYA_ROWCHAIN_ABLATE & 2
    if (stage != nullptr) return;
#endif
#pragma unroll
    for (int r = 0; r < SRW; ++r) {
        const int R = wave * SRW + r, i = R & 15, j = R >> 4;        // wave-uniform
        const uint32_t wr = NATURAL ? wrow_of(i, j) : wrow_lpc(i, j);
        const uint32_t vo = (l16 ^ (uint32_t)(i << 4)) + wr * (RK * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(dst + r * 1024), 16, vo, row0 * (RK * 2), 0, 0);
    }
}

template <int N>
__device__ __forceinline__ void wait_newer(bf16x8 (&w)[NJ]) { lgkm_wait<N>(w); }

// one chunk's 64 MFMAs of a wave.  SWAP = false: acc[j] = W_j . X^T (lane (g, t): W rows 4g+e of block j, token t),
// SWAP = true: acc[j] = X . W_j^T (lane (g, i): tokens 4g+e, W row i of block j).  Fragment reads AH k-steps ahead.
template <bool SWAP, int AH>
__device__ __forceinline__ void chunk_mfma(f32x4 (&acc)[NJ], const bf16x8 (&xf)[16], const uint32_t (&wa)[4]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 wf[AH + 1][NJ];
#if BYA_ROWCHAIN_ABLATE & 8
#pragma unroll
    for (int a = 0; a <= AH; ++a)
#pragma unroll
        for (int j = 0; j < NJ; ++j) { u32x4 t = {wa[0] + a, wa[1] + j, 0x3f803f80u, 0x3f803f80u}; asm volatile("" : "+v"(t)); wf[a][j] = __builtin_bit_cast(bf16x8, t); }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const int cur = ks % (AH + 1);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            acc[j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[ks], wf[cur][j], acc[j], 0, 0, 0)
                          : __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cur][j], xf[ks], acc[j], 0, 0, 0);
    }
    return;
#endif
#pragma unroll
    for (int a = 0; a < AH; ++a) read_kstep(wf[a], wa, a);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const int cur = ks % (AH + 1);
        if (ks + AH < 16) {
            read_kstep(wf[(ks + AH) % (AH + 1)], wa, ks + AH);
            wait_newer<AH * NJ>(wf[cur]);
        } else if (AH == 2 && ks + 1 < 16) {
            wait_newer<NJ>(wf[cur]);
        } else {
            wait_newer<0>(wf[cur]);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            acc[j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[ks], wf[cur][j], acc[j], 0, 0, 0)
                          : __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cur][j], xf[ks], acc[j], 0, 0, 0);
    }
}

// fragment pair 2 c, 2 c + 1 of a 16-fragment register array, c a wave-uniform run-time number: eight selects on constant
// indices (the arrays stay in registers; a computed index would send them to scratch, and a switch makes hipcc copy and
// spill whole fragments around its branches)
template <typename T>
__device__ __forceinline__ void set2(T (&A)[16], int c, const T v0, const T v1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool hit = c == i;
        A[2 * i] = hit ? v0 : A[2 * i];
        A[2 * i + 1] = hit ? v1 : A[2 * i + 1];
    }
}
template <typename T>
__device__ __forceinline__ void get2(const T (&A)[16], int c, T& v0, T& v1) {
    v0 = A[14];
    v1 = A[15];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const bool hit = c == i;
        v0 = hit ? A[2 * i] : v0;
        v1 = hit ? A[2 * i + 1] : v1;
    }
}

// the pass schedule both kernels share: pass 0 = tp0 tiles per workgroup, every later pass the remainder dealt over all
// workgroups (at most CW each)
struct Passes {              // (32-bit: a launch has fewer than 2^31 / 1024 rows)
    int base, tiles, tpw, grid, wg;
    __device__ __forceinline__ Passes(int tiles_, int tp0, int grid_, int wg_) : base(0), tiles(tiles_), tpw(tp0), grid(grid_), wg(wg_) {}
    __device__ __forceinline__ int tile_of(int wave) const { return base + wg * tpw + wave; }
    __device__ __forceinline__ bool active(int wave) const { return wave < tpw && tile_of(wave) < tiles; }
    // -> does THIS workgroup have a tile in the next pass?
    __device__ __forceinline__ bool peek_next(int& nbase, int& ntpw) const {
        nbase = base + grid * tpw;
        ntpw = 0;
        if (nbase >= tiles) return false;
        const int per = (tiles - nbase + grid - 1) / grid;
        ntpw = per < CW ? per : CW;
        return nbase + wg * ntpw < tiles;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
struct MlpChainArgs {
    const bf16_t* X; bf16_t* C; const bf16_t* W1; const float* colsum1; const float* cvec1; const bf16_t* W2; const float* cvec2;
    int M, ldx, ldc, tiles, tp0;
    float eps;
};

__global__ __launch_bounds__(64 * CW, 2) void router_mlp_chain_kernel(MlpChainArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* s1 = reinterpret_cast<float*>(smem);
    float* c1 = s1 + 512;
    float* c2 = c1 + 512;
    char* ring = smem + CONST_MLP;
    for (int i = tid; i < 512; i += 64 * CW) {
        s1[i] = p.colsum1[i];
        c1[i] = p.cvec1[i];
        c2[i] = p.cvec2[i];
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.X, 0, (int)(((long long)(p.M - 1) * p.ldx + RK) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.C, 0, (int)(((long long)(p.M - 1) * p.ldc + RK) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W1, 0, 512 * RK * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W2, 0, 512 * RK * 2, 0x00020000);
    const uint32_t ring_base = (uint32_t)(uintptr_t)LDS_PTR(ring);
    const uint32_t smem_base = (uint32_t)(uintptr_t)LDS_PTR(smem);

    Passes ps(p.tiles, p.tp0, (int)gridDim.x, (int)blockIdx.x);
    int stg = 0;
    stage_rows<true>(rsW1, ring, wave, 0);
    for (;;) {
        const bool active = ps.active(wave);                           // wave-uniform
        int nbase, ntpw;
        const bool more = ps.peek_next(nbase, ntpw);                   // workgroup-uniform
        bf16x8 xf[16], hf[16];
        float mean = 0.f, rstd = 1.f;
        uint32_t crow = 0;                                             // byte offset of this lane's token row in C
        if (active) {
            // X fragments: row 16 tile + t, k = 32 ks + 8 g .. + 7 (rows >= M read as zeros through the descriptor)
            const uint32_t ln = lane_now(), to = ln & 15u, go = ln >> 4;
            const uint32_t row = (uint32_t)ps.tile_of(wave) * 16u + to;
            const uint32_t vo = row * (uint32_t)(p.ldx * 2) + go * 16;
            crow = row * (uint32_t)(p.ldc * 2);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
                xf[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsX, vo + ks * 64, 0, 0));
            tile_row_stats(xf, to, p.eps, mean, rstd);
        }
        // the fragments (and the first chunk's W) have landed; tell hipcc's wait-count bookkeeping (see rowgemm512_kernel)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(xf[ks]));

        // ---------------------------------------------------------------- h = GELU( LN(x) . W1^T + b1 ), rounded to bf16
        for (int cc = 0; cc < 8; ++cc) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this chunk's LDS-DMA (nothing younger in this phase)
            ROWK_BARRIER();                              // ... of every wave; the other stage is free
            if (cc < 7) stage_rows<true>(rsW1, ring + (stg ^ 1) * STAGE_BYTES, wave, (cc + 1) * CH);
            else stage_rows<true>(rsW2, ring + (stg ^ 1) * STAGE_BYTES, wave, 0);
            if (ROWK_COMPUTE(active)) {
                const uint32_t ln = lane_now(), to = ln & 15u, go = ln >> 4;
                uint32_t wa[4];
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    wa[m] = ring_base + stg * STAGE_BYTES + to * 1024 + (((go ^ (to & 3)) | ((m ^ (to >> 2)) << 2)) << 4);
                f32x4 acc[NJ];
                chunk_mfma<false, 2>(acc, xf, wa);
                bf16x8 hh[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x4 s0, s1v, c0, c1v;
                    const uint32_t a = smem_base + ((uint32_t)(cc * CH) + lane_col(go, u)) * 4, ac = a + 512u * 4;
                    lds_read_f<0>(s0, a);
                    lds_read_f<16>(s1v, a);
                    lds_read_f<0>(c0, ac);
                    lds_read_f<16>(c1v, ac);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s0), "+v"(s1v), "+v"(c0), "+v"(c1v));
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float a0 = acc[2 * u + (e >> 2)][e & 3];
                        const float sv = (e >> 2) ? s1v[e & 3] : s0[e & 3];
                        const float cv = (e >> 2) ? c1v[e & 3] : c0[e & 3];
                        #if BYA_ROWCHAIN_ABLATE & 4
                        v[e] = fmaf(rstd, fmaf(-mean, sv, a0), cv);
#else
                        v[e] = gelu_erf_f(fmaf(rstd, fmaf(-mean, sv, a0), cv));
#endif        // (rowgemm512_kernel<true, ., GELU_ERF>)
                    }
                    hh[u] = __builtin_bit_cast(bf16x8, pack8(v));
                }
                set2(hf, cc, hh[0], hh[1]);
            }
            stg ^= 1;
        }
        // ---------------------------------------------------------------- x += h . W2^T + b2
        for (int cc = 0; cc < 8; ++cc) {
            // only the previous chunk's two output stores are younger than this chunk's LDS-DMA (vmcnt retires in order)
            if (ROWK_COMPUTE(active) && cc > 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ROWK_BARRIER();
            if (cc < 7) stage_rows<true>(rsW2, ring + (stg ^ 1) * STAGE_BYTES, wave, (cc + 1) * CH);
            else if (more) stage_rows<true>(rsW1, ring + (stg ^ 1) * STAGE_BYTES, wave, 0);
            if (ROWK_COMPUTE(active)) {
                const uint32_t ln = lane_now(), to = ln & 15u, go = ln >> 4;
                uint32_t wa[4];
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    wa[m] = ring_base + stg * STAGE_BYTES + to * 1024 + (((go ^ (to & 3)) | ((m ^ (to >> 2)) << 2)) << 4);
                f32x4 acc[NJ];
                chunk_mfma<false, 2>(acc, hf, wa);
                bf16x8 rr[2];
                get2(xf, cc, rr[0], rr[1]);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x4 c0, c1v;
                    const uint32_t ac = smem_base + (1024u + (uint32_t)(cc * CH) + lane_col(go, u)) * 4;
                    lds_read_f<0>(c0, ac);
                    lds_read_f<16>(c1v, ac);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0), "+v"(c1v));
                    const u32x4 rv = __builtin_bit_cast(u32x4, rr[u]);
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float o = acc[2 * u + (e >> 2)][e & 3] + ((e >> 2) ? c1v[e & 3] : c0[e & 3]);   // (rowgemm512q_kernel<., true, NONE>)
                        o += (e & 1) ? bfhi(rv[e >> 1]) : bflo(rv[e >> 1]);
                        v[e] = o;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(pack8(v), rsC, crow + ((uint32_t)(cc * CH) + lane_col(go, u)) * 2, 0, 0);
                }
            }
            stg ^= 1;
        }
        if (!more) break;
        ps.base = nbase;
        ps.tpw = ntpw;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
struct AttnChainArgs {
    const bf16_t* X; bf16_t* C; const bf16_t* Wqkv; const float* colsum; const float* cvec; const bf16_t* Wo; const float* cvec_o;
    int M, ldx, ldc, L, P, G, tp0;
    int n_groups, n_inner, outer_stride, seq_stride, tiles;
    float eps, scale_log2;
};

// Tiles, cells, masks and the attention itself as in rowattn512_kernel<false> (rowgemm.hip) with ONE tile per wave; the v
// chunk is staged in the row GEMMs' order instead of the q / k order, so that O^T leaves the matrix core as the two
// operand fragments 2 head, 2 head + 1 of the out-projection.
__global__ __launch_bounds__(64 * CW, 2) void router_attn_chain_kernel(AttnChainArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* s_lds = reinterpret_cast<float*>(smem);
    float* c_lds = s_lds + 1536;
    float* co_lds = c_lds + 1536;
    float* st_lds = co_lds + 512 + wave * 32;                        // this wave's statistics
    char* ring = smem + CONST_ATTN;
    for (int i = tid; i < 1536; i += 64 * CW) {
        s_lds[i] = p.colsum[i];
        c_lds[i] = p.cvec[i];
    }
    co_lds[tid] = p.cvec_o[tid];                                     // 512 threads
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.X, 0, (int)(((long long)(p.M - 1) * p.ldx + RK) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.C, 0, (int)(((long long)(p.M - 1) * p.ldc + RK) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wqkv, 0, 1536 * RK * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsWo = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wo, 0, 512 * RK * 2, 0x00020000);
    const uint32_t ring_base = (uint32_t)(uintptr_t)LDS_PTR(ring);
    const uint32_t smem_base = (uint32_t)(uintptr_t)LDS_PTR(smem);
    const uint32_t st_base = (uint32_t)(uintptr_t)LDS_PTR(st_lds);

    Passes ps(p.tiles, p.tp0, (int)gridDim.x, (int)blockIdx.x);
    int stg = 0;
    stage_rows<false>(rsW, ring, wave, 0);                           // q chunk of head 0
    for (;;) {
        const bool active = ps.active(wave);
        int nbase, ntpw;
        const bool more = ps.peek_next(nbase, ntpw);
        bf16x8 xf[16], of[16];
        float mean = 0.f, rstd = 1.f;
        uint32_t rowi = (uint32_t)p.M;    // the row of this lane's token slot (M = none: loads read zeros, stores are dropped)
        uint32_t kmask = 0;               // bit e: key slot 4g+e belongs to the group of this lane's query
        if (active) {
            const uint32_t ln = lane_now(), to = ln & 15u, go = ln >> 4;
            const uint32_t P = (uint32_t)p.P, gq = to / P, mq = to - gq * P;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t kk = 4 * go + e, gk = kk / P, mk = kk - gk * P;
                kmask |= (gk == gq && mk < (uint32_t)p.L ? 1u : 0u) << e;
            }
            const uint32_t grp = (uint32_t)(ps.tile_of(wave) * p.G) + gq;
            const bool ok = mq < (uint32_t)p.L && grp < (uint32_t)p.n_groups;
            const uint32_t go_ = grp / (uint32_t)p.n_inner;
            const uint32_t row = go_ * (uint32_t)p.outer_stride + (grp - go_ * (uint32_t)p.n_inner) + mq * (uint32_t)p.seq_stride;
            rowi = ok ? row : (uint32_t)p.M;
            const uint32_t vo = rowi * (uint32_t)(p.ldx * 2) + go * 16;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
                xf[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsX, vo + ks * 64, 0, 0));
            tile_row_stats(xf, to, p.eps, mean, rstd);
            if (go == 0) {                        // tokens 4g+e of the v chunk's layout read them from here
                st_lds[to] = mean;
                st_lds[16 + to] = rstd;
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(xf[ks]));

        for (int head = 0; head < 8; ++head) {
            u32x4 qf[2], kf[2];
            u32x2 pf = {0u, 0u};
            float invl = 0.f;
            // ---------------------------------------------------------------- q and k: transposed product + LN epilogue
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ROWK_BARRIER();
                if (part == 0) stage_rows<false>(rsW, ring + (stg ^ 1) * STAGE_BYTES, wave, 512 + head * 64);      // k
                else stage_rows<true>(rsW, ring + (stg ^ 1) * STAGE_BYTES, wave, 1024 + head * 64);                // v
                if (ROWK_COMPUTE(active)) {
                    const uint32_t ln = lane_now(), to = ln & 15u, go = ln >> 4;
                    uint32_t wa[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        wa[m] = ring_base + stg * STAGE_BYTES + to * 1024 + (((go ^ (to & 3)) | ((m ^ (to >> 2)) << 2)) << 4);
                    f32x4 acc[NJ];
                    chunk_mfma<false, 1>(acc, xf, wa);
                    const uint32_t sc_base = smem_base + (uint32_t)(part * 512 + head * 64 + LPC * go) * 4;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        f32x4 s0, s1v, c0, c1v;
                        const uint32_t a = sc_base + (uint32_t)(8 * kb) * 4, ac = a + 1536u * 4;
                        lds_read_f<0>(s0, a);
                        lds_read_f<16>(s1v, a);
                        lds_read_f<0>(c0, ac);
                        lds_read_f<16>(c1v, ac);
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s0), "+v"(s1v), "+v"(c0), "+v"(c1v));
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float a0 = acc[2 * kb + (e >> 2)][e & 3];
                            const float sv = (e >> 2) ? s1v[e & 3] : s0[e & 3];
                            const float cv = (e >> 2) ? c1v[e & 3] : c0[e & 3];
                            v[e] = fmaf(rstd, fmaf(-mean, sv, a0), cv);
                        }
                        if (part == 0) qf[kb] = pack8(v); else kf[kb] = pack8(v);
                    }
                }
                stg ^= 1;
            }
            // ---------------------------------------------------------------- S^T = K . Q^T, masked softmax over the keys
            if (ROWK_COMPUTE(active)) {
                f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
                    st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf[kb]),
                                                                 __builtin_bit_cast(bf16x8, qf[kb]), st, 0, 0, 0);
                float sv[4], mx = -INFINITY;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sv[e] = (kmask >> e) & 1u ? st[e] : -INFINITY;
                    mx = fmaxf(mx, sv[e]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                float l = 0.f, pe[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pe[e] = __builtin_amdgcn_exp2f((sv[e] - mx) * p.scale_log2);    // masked keys: exp2(-inf) = 0
                    l += pe[e];
                }
                pf[0] = pack2bf(pe[0], pe[1]);
                pf[1] = pack2bf(pe[2], pe[3]);
                l += __shfl_xor(l, 16);
                l += __shfl_xor(l, 32);
                invl = __builtin_amdgcn_rcpf(l);
            }
            // ---------------------------------------------------------------- v: straight product, O^T = V^T . P^T
            {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ROWK_BARRIER();
                if (head < 7) stage_rows<false>(rsW, ring + (stg ^ 1) * STAGE_BYTES, wave, (head + 1) * 64);       // next q
                else stage_rows<true>(rsWo, ring + (stg ^ 1) * STAGE_BYTES, wave, 0);                              // to_out
                if (ROWK_COMPUTE(active)) {
                    const uint32_t ln = lane_now(), to = ln & 15u, go = ln >> 4;
                    uint32_t wa[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        wa[m] = ring_base + stg * STAGE_BYTES + to * 1024 + (((go ^ (to & 3)) | ((m ^ (to >> 2)) << 2)) << 4);
                    f32x4 acc[NJ];
                    chunk_mfma<true, 1>(acc, xf, wa);
                    // lane (g, i): tokens 4g+e, W row i of block j = v feature wrow_of(i, j) of the head
                    f32x4 m4, r4;
                    lds_read_f<0>(m4, st_base + (uint32_t)(4 * go) * 4);
                    lds_read_f<64>(r4, st_base + (uint32_t)(4 * go) * 4);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(m4), "+v"(r4));
                    float ov[16];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const uint32_t col = (uint32_t)(1024 + head * 64) + (uint32_t)(32 * (j >> 1) + 4 * (j & 1)) + 8 * (to >> 2) + (to & 3);
                        const float sj = s_lds[col], cj = c_lds[col];
                        float vv[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) vv[e] = fmaf(r4[e], fmaf(-m4[e], sj, acc[j][e]), cj);
                        u32x2 vt;
                        vt[0] = pack2bf(vv[0], vv[1]);
                        vt[1] = pack2bf(vv[2], vv[3]);
                        f32x4 o = {0.f, 0.f, 0.f, 0.f};
                        o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, vt), __builtin_bit_cast(s16x4, pf), o, 0, 0, 0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) ov[4 * j + e] = o[e] * invl;
                    }
                    // lane (g, t): token t, features 64 head + 32 u + 8 g .. + 7 = k-step 2 head + u of the out-projection
                    const bf16x8 o0 = __builtin_bit_cast(bf16x8, pack8(ov)), o1 = __builtin_bit_cast(bf16x8, pack8(ov + 8));
                    set2(of, head, o0, o1);
                }
                stg ^= 1;
            }
        }
        // ---------------------------------------------------------------- x += O . Wo^T + bo
        for (int cc = 0; cc < 8; ++cc) {
            if (ROWK_COMPUTE(active) && cc > 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ROWK_BARRIER();
            if (cc < 7) stage_rows<true>(rsWo, ring + (stg ^ 1) * STAGE_BYTES, wave, (cc + 1) * CH);
            else if (more) stage_rows<false>(rsW, ring + (stg ^ 1) * STAGE_BYTES, wave, 0);
            if (ROWK_COMPUTE(active)) {
                const uint32_t ln = lane_now(), to = ln & 15u, go = ln >> 4;
                uint32_t wa[4];
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    wa[m] = ring_base + stg * STAGE_BYTES + to * 1024 + (((go ^ (to & 3)) | ((m ^ (to >> 2)) << 2)) << 4);
                f32x4 acc[NJ];
                chunk_mfma<false, 1>(acc, of, wa);
                bf16x8 rr[2];
                get2(xf, cc, rr[0], rr[1]);
                const uint32_t crow = rowi * (uint32_t)(p.ldc * 2);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x4 c0, c1v;
                    const uint32_t ac = smem_base + (3072u + (uint32_t)(cc * CH) + lane_col(go, u)) * 4;
                    lds_read_f<0>(c0, ac);
                    lds_read_f<16>(c1v, ac);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0), "+v"(c1v));
                    const u32x4 rv = __builtin_bit_cast(u32x4, rr[u]);
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float o = acc[2 * u + (e >> 2)][e & 3] + ((e >> 2) ? c1v[e & 3] : c0[e & 3]);
                        o += (e & 1) ? bfhi(rv[e >> 1]) : bflo(rv[e >> 1]);
                        v[e] = o;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(pack8(v), rsC, crow + ((uint32_t)(cc * CH) + lane_col(go, u)) * 2, 0, 0);
                }
            }
            stg ^= 1;
        }
        if (!more) break;
        ps.base = nbase;
        ps.tpw = ntpw;
    }
}

// first-pass tiles per workgroup when the caller does not say: fill the waves of 256 workgroups, never more than one tile each
int auto_tp0(int tiles) {
    const int t = (tiles + 255) / 256;
    return t < 1 ? 1 : t > CW ? CW : t;
}

int grid_for(int tiles, int tp0) {
    const int g = (tiles + tp0 - 1) / tp0;
    return (int)(g < 256 ? g : 256);
}

}  // namespace

extern "C" int bya_router_mlp_fused(const void* X, const void* W1, const float* colsum1, const float* cvec1, const void* W2,
                                    const float* cvec2, void* C, int32_t M, int32_t ldx, int32_t ldc, float eps,
                                    int32_t tiles_pass0, hipStream_t stream) {
    if (!X || !W1 || !colsum1 || !cvec1 || !W2 || !cvec2 || !C || M <= 0) return BYA_ERR_SHAPE;
    if (ldx < RK || ldc < RK || ldx % 8 || ldc % 8) return BYA_ERR_ALIGN;
    if (((uintptr_t)X | (uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)C) & 15) return BYA_ERR_ALIGN;
    if (((long long)M + 16) * ldx * 2 >= (1LL << 31) || ((long long)M + 16) * ldc * 2 >= (1LL << 31)) return BYA_ERR_SHAPE;
    if (tiles_pass0 < 0 || tiles_pass0 > CW) return BYA_ERR_SHAPE;
    MlpChainArgs a;
    a.X = (const bf16_t*)X; a.C = (bf16_t*)C; a.W1 = (const bf16_t*)W1; a.colsum1 = colsum1; a.cvec1 = cvec1;
    a.W2 = (const bf16_t*)W2; a.cvec2 = cvec2; a.M = M; a.ldx = ldx; a.ldc = ldc; a.eps = eps;
    a.tiles = (M + 15) / 16;
    a.tp0 = tiles_pass0 ? tiles_pass0 : auto_tp0(a.tiles);
    static std::atomic<unsigned long long> attr_done{0};
    if (bya_allow_big_lds(reinterpret_cast<const void*>(router_mlp_chain_kernel), 160 * 1024, attr_done) != BYA_OK) return BYA_ERR_LAUNCH;
    BYA_LAUNCH(router_mlp_chain_kernel, dim3(grid_for(a.tiles, a.tp0)), dim3(64 * CW), (size_t)CONST_MLP + 2 * STAGE_BYTES, stream, a);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

extern "C" int bya_router_group_attn_out(const void* X, const void* Wqkv, const float* colsum, const float* cvec, const void* Wo,
                                         const float* cvec_o, void* C, int32_t M, int32_t ldx, int32_t ldc, int32_t L,
                                         int64_t n_outer, int64_t n_inner, int64_t outer_stride, int64_t seq_stride, float eps,
                                         float scale, int32_t tiles_pass0, hipStream_t stream) {
    if (!X || !Wqkv || !colsum || !cvec || !Wo || !cvec_o || !C || M <= 0 || n_outer <= 0 || n_inner <= 0) return BYA_ERR_SHAPE;
    if (L < 1) return BYA_ERR_SHAPE;
    if (L > 16) return BYA_ERR_UNSUPPORTED;          // a group must fit the ONE 16-row tile of a wave (longer: the unfused pair)
    if (outer_stride < 0 || seq_stride < 0) return BYA_ERR_SHAPE;
    if ((n_outer - 1) * outer_stride + (n_inner - 1) + (int64_t)(L - 1) * seq_stride >= M) return BYA_ERR_SHAPE;
    if (ldx < RK || ldc < RK || ldx % 8 || ldc % 8) return BYA_ERR_ALIGN;
    if (((uintptr_t)X | (uintptr_t)Wqkv | (uintptr_t)Wo | (uintptr_t)C) & 15) return BYA_ERR_ALIGN;
    if (((long long)M + 1) * ldx * 2 >= (1LL << 31) || ((long long)M + 1) * ldc * 2 >= (1LL << 31)) return BYA_ERR_SHAPE;
    if (tiles_pass0 < 0 || tiles_pass0 > CW) return BYA_ERR_SHAPE;
    AttnChainArgs a;
    a.X = (const bf16_t*)X; a.C = (bf16_t*)C; a.Wqkv = (const bf16_t*)Wqkv; a.colsum = colsum; a.cvec = cvec;
    a.Wo = (const bf16_t*)Wo; a.cvec_o = cvec_o; a.M = M; a.ldx = ldx; a.ldc = ldc; a.L = L;
    a.P = L <= 1 ? 1 : L <= 2 ? 2 : L <= 4 ? 4 : L <= 8 ? 8 : 16;
    a.G = 16 / a.P;
    if (n_outer * n_inner > M || n_inner > M || outer_stride > M || seq_stride > M) return BYA_ERR_SHAPE;    // (32-bit from here)
    a.n_groups = (int)(n_outer * n_inner); a.n_inner = (int)n_inner; a.outer_stride = (int)outer_stride; a.seq_stride = (int)seq_stride;
    a.tiles = (a.n_groups + a.G - 1) / a.G;
    a.tp0 = tiles_pass0 ? tiles_pass0 : auto_tp0(a.tiles);
    a.eps = eps; a.scale_log2 = scale * 1.4426950408889634f;
    static std::atomic<unsigned long long> attr_done{0};
    if (bya_allow_big_lds(reinterpret_cast<const void*>(router_attn_chain_kernel), 160 * 1024, attr_done) != BYA_OK) return BYA_ERR_LAUNCH;
    BYA_LAUNCH(router_attn_chain_kernel, dim3(grid_for(a.tiles, a.tp0)), dim3(64 * CW), (size_t)CONST_ATTN + 2 * STAGE_BYTES, stream, a);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}
