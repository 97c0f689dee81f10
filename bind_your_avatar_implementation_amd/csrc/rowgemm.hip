// Row-stationary GEMM for the Embedding Router's 512-wide Linears (reference models/router.py:468-493, the twelve
// projections + MLP of every SpatialTemporalAttentionBlock):   C[M,N] = epi( LN?(X)[M,512] . W[N,512]^T ).
//
// K = 512 is only eight 64-wide K-tiles: a tiled GEMM spends its time on pipeline fill/drain and load latency, not on
// the matrix core (measured 390-545 TFLOP/s).  Here a wave keeps its 32 rows of X -- all 512 K -- in 128 VGPRs as MFMA
// operands (loaded straight from global memory in fragment layout, no LDS), and streams 64-column chunks of W
// (64 KiB each, one 1-KiB W row per LDS-DMA instruction) through a 2-stage LDS ring shared by the block's 8 waves:
// 8 LDS-DMA instructions and one barrier per 128 MFMAs of a wave.
//
// What bounds it (round-2 ablations, profiles/history/r2_rowgemm_ablation.txt): every wave reads the whole W chunk from LDS for its
// own 32 rows, so a chunk costs 8 waves x 64 KiB = 512 KiB of ds_read_b128 = 4096 LDS cycles at 128 B/clk -- exactly the
// 4096 matrix-core cycles of its 1024 MFMAs: the loop is LDS-read-bound at ~60 % of that rate even with the X loads and
// output stores compiled out (1180 of 2000 TFLOP/s).  The two build-time alternatives kept below were measured and lost:
// BYA_ROWGEMM_CH=32 (two independent 4-wave workgroups per CU: same LDS bytes, -7 % on N = 1536, +2 % on N = 512) and
// BYA_ROWGEMM_HB=4 (64 rows per wave, X in 256 registers, one wave per SIMD: half the LDS bytes, but hipcc's schedule of
// the single wave loses 17 %; it would need a hand-placed loop like gemm_v4's).  Round 3 re-tried that form with 32 x 32 x 16
// MFMAs (half the MFMA issue slots), X in AGPRs and the previous chunk's epilogue written between the MFMAs of the next
// chunk, still compiler-scheduled: bit-identical results, 13-19 % SLOWER on all four launch shapes
// (profiles/history/r3_rowgemm_w4_compiler_scheduled_probe.json; hipcc triplicated the chunk loop and put 223 s_waitcnt and 117 s_nop
// into it) -- removed again.  A second attempt placed the chunk's 128 MFMAs and 64 fragment reads by hand (eight inline-asm
// groups of 16 MFMAs with counted lgkmcnt waits, the previous chunk's epilogue in eight compiler-scheduled pieces between
// them; clean code: no scratch in the loop, no compiler waits beyond the epilogue's LDS reads): bit-identical again and STILL
// 12-27 % slower (profiles/history/r3_rowgemm_w4_hand_placed_probe.json: 83 vs 68 us on N = 1536).  A single wave issues in order,
// so the epilogue's VALU work between two MFMA groups runs with the matrix pipe idle, where the 8-wave kernel's second wave
// per SIMD fills it -- and, the larger term, neither form overlaps the X-fragment load of a row block (all 256 workgroups
// request theirs at the same moment: 64 MB at launch, ~15 us of a 68 us launch) with MFMAs.  The kernel is bound by that and
// by HBM bytes (N = 512 launches: 108 MB algorithmic in 37 us), not by LDS reads alone; see DESIGN.md section 9.
// (r6 correction, MI355X_MICROARCH.md section LDS: ds_read_b128 moves 256 B/clk/CU, not 128 -- the chunk loop of this kernel is
// at half the LDS array's rate, not at it.  Ablations of the 16-row-per-wave chain kernels, where it IS at that rate, and of the
// W-stationary kernel below: profiles/r6_final_rowchain_ablate.json, r6_final_rowgemm_q_ablate.json -- W staging, fragment reads,
// barriers and the GELU cost 7-9 us each of a 53 us pass and add up instead of overlapping, because the two waves of a SIMD run
// the same phase at the same time; the N = 512 kernel's memory side alone takes 28 of its 35 us: 108 MB at 3.9 TB/s.)
// The product is computed transposed (W fragment = A operand, X fragment = B operand), so a lane ends up with 16
// consecutive output columns of one token: 16-byte stores, no LDS transpose.
//
// Fused LayerNorm (the nn.LayerNorm in front of every q|k|v projection and of the MLP): gamma is folded into the
// weights and beta into the bias at pack time,
//     LN(x) . W^T + b  =  rstd * ( x . Wg^T  -  mean * s )  +  c,     Wg = W * gamma,  s = rowsum(Wg),  c = W . beta + b,
// so the GEMM runs on the RAW rows and the epilogue applies the two row statistics.  The statistics come from the
// matrix core as well: sum(x) = ones . x^T and sum(x^2) = diag(x . x^T), 64 extra MFMAs per wave instead of ~3000
// VALU cycles (fp32 accumulation of exact bf16 products).
#include "rowgemm_common.h"
#include "../../include/bya.h"
#include "options.h"

#ifndef BYA_ROWGEMM_ABLATE
#define BYA_ROWGEMM_ABLATE 0     // timing-only ablations (tools/rowgemm_q_ablate.py): 1 = no X loads, 2 = no output stores; W-stationary kernel
                                 // also: 4 = no residual loads, 8 = no W preload, 16 = no MFMAs, 32 = no W fragment reads
#endif

namespace {

using namespace rowk;      // RK, CH, NJ, LPC, STAGE_BYTES, the ring's layout and fragment readers, gelu_erf_f (rowgemm_common.h)

struct RowGemmArgs {
    const bf16_t* X; const bf16_t* W; const float* colsum; const float* cvec; const bf16_t* res; bf16_t* C;
    int M, N, ldx, ldc, ldres;
    float eps;
};

// (The build-time alternatives BYA_ROWGEMM_CH = 32 and BYA_ROWGEMM_HB = 4 of rounds 2-3 -- measured, lost, see the header
// comment -- are gone from the source since round 6; the constants below are what they selected between.)
constexpr int HB = 2;                     // 16-row halves per wave: 32 rows of X in 128 registers
constexpr int NW = CH / (4 * HB);         // waves per workgroup (16*HB rows each)
constexpr int SR = CH / NW;               // W rows of a chunk every wave stages
constexpr int RB = 16 * HB * NW;          // rows per workgroup pass
constexpr int WG_PER_CU = 1;
constexpr int AHEAD = 1;                  // k-steps of W fragments in flight (a k-step is 2*NJ MFMAs)

// Persistent blocks of 8 waves (256 rows).  The work list is every (row block, 64-column chunk) pair in row-block-major
// order, cut into gridDim.x equal contiguous ranges: every CU gets the same number of chunks (+-1) no matter how M and N
// divide, and a range touches at most two or three row blocks, so the X fragments (and the row statistics) are
// reloaded only there.  The W-chunk LDS-DMA pipeline runs straight through a row-block change.
template <bool LN, bool RES, int ACT>
__global__ __launch_bounds__(64 * NW, 2) void rowgemm512_kernel(RowGemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nrb = (p.M + RB - 1) / RB, ncc = p.N / CH;
    const long long total = (long long)nrb * ncc;
    // work range of this workgroup.  Neighbouring ranges share a row block (a range is about half a row block at N = 1536):
    // give consecutive ranges to the SAME XCD (blockIdx % 8 under round-robin dispatch) so that the second reader of a row
    // block's X finds it in that XCD's L2
    const int rid = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    const int q0 = (int)(total * rid / gridDim.x), q1 = (int)(total * (rid + 1) / gridDim.x);
    if (q0 >= q1) return;

    // LDS: [colsum: N f32][cvec: N f32][ring: 2 x 64 KiB]
    float* s_lds = reinterpret_cast<float*>(smem);
    float* c_lds = s_lds + p.N;
    char* ring = smem + p.N * 8;
    for (int i = tid; i < p.N; i += 64 * NW) {
        s_lds[i] = LN ? p.colsum[i] : 0.0f;
        c_lds[i] = p.cvec[i];
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.X, 0, (int)(((long long)(p.M - 1) * p.ldx + RK) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, p.N * RK * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.C, 0, (int)(((long long)(p.M - 1) * p.ldc + p.N) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(RES ? p.res : p.C), 0, (int)(((long long)(p.M - 1) * (RES ? p.ldres : p.ldc) + p.N) * 2), 0x00020000);

    // ---- W chunk staging (8 one-KiB rows per wave): LDS row R = j*16 + i holds output column
    // chunk0 + 16*(i>>2) + 4*j + (i&3), so that lane group g ends up with the 16 consecutive columns chunk0 + 16g ..;
    // its 64 16-byte pieces are XOR-swizzled with i: the 16 rows read by one fragment instruction hit 16 bank groups.
    auto stage_chunk = [&](int q, int stg) {
        char* dst = ring + stg * STAGE_BYTES + wave * SR * 1024;
        const int col0 = (q % ncc) * CH;
        const uint32_t l16 = lane_now() << 4;
#pragma unroll
        for (int r = 0; r < SR; ++r) {
            const int R = wave * SR + r, i = R & 15, j = R >> 4;        // wave-uniform
            const uint32_t vo = (l16 ^ (uint32_t)(i << 4)) + wrow_of(i, j) * (RK * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(dst + r * 1024), 16, vo, col0 * (RK * 2), 0, 0);
        }
    };

    // lane-constant fragment read addresses: LDS row t of a column block, 16-byte piece (4*ks + g) ^ t
    const uint32_t ring_base = (uint32_t)(uintptr_t)LDS_PTR(ring);
    const uint32_t smem_base = (uint32_t)(uintptr_t)LDS_PTR(smem);

    // Outer loop: the row blocks this workgroup's range touches (two or three); inner loop: their chunks.  The X fragments
    // are read-only inside the inner loop, which has no row-block branch -- with the branch inside, hipcc shuffled and
    // spilled fragments around the MFMA section (every scratch reload drags an s_waitcnt vmcnt(0) with it).
    bf16x8 xf[HB][16];
    stage_chunk(q0, 0);
    int q = q0, it = 0;
    while (q < q1) {
        const int rb = q / ncc;
        const int qe = (rb + 1) * ncc < q1 ? (rb + 1) * ncc : q1;
        const int r0 = rb * RB + wave * (16 * HB);
        float mean[HB], rstd[HB];
#pragma unroll
        for (int h = 0; h < HB; ++h) mean[h] = 0.f, rstd[h] = 1.f;
        {
            // X fragments: rows r0 + h*16 + t, k = ks*32 + g*8 .. +8 (rows >= M read as zeros through the descriptor)
            const uint32_t ln = lane_now(), to = ln & 15u, go = ln >> 4;
#pragma unroll
            for (int h = 0; h < HB; ++h) {
                const uint32_t vo = ((uint32_t)(r0 + h * 16) + to) * (uint32_t)(p.ldx * 2) + go * 16;
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
#if BYA_ROWGEMM_ABLATE & 1
                    { u32x4 t = {vo + ks, vo, ln, 0x3f803f80u}; asm volatile("" : "+v"(t)); xf[h][ks] = __builtin_bit_cast(bf16x8, t); }
#else
                    xf[h][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsX, vo + ks * 64, 0, 0));
#endif
            }
            if (LN) {
                // row statistics on the matrix core: sum(x) = ones . x^T, sum(x^2) = diag(x . x^T)
                bf16x8 ones;
#pragma unroll
                for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#pragma unroll
                for (int h = 0; h < HB; ++h) {
                    f32x4 sm = {0.f, 0.f, 0.f, 0.f}, gr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 16; ++ks) {
                        sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, xf[h][ks], sm, 0, 0, 0);
                        gr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[h][ks], xf[h][ks], gr, 0, 0, 0);
                    }
                    // lane (g, t) holds <x_{4g+e}, x_t>; the diagonal of token t sits in lane (t>>2, t), register t&3
                    const int e = to & 3;
                    const float d = e == 0 ? gr[0] : e == 1 ? gr[1] : e == 2 ? gr[2] : gr[3];
                    const float sq = __shfl(d, (int)(to + 16 * (to >> 2)));
                    const float mu = sm[0] * (1.0f / RK);
                    const float var = fmaxf(sq * (1.0f / RK) - mu * mu, 0.0f);
                    mean[h] = mu;
                    rstd[h] = rsqrtf(var + p.eps);
                }
            }
        }
        // the X fragments (and with them the first chunk's W, requested earlier) have landed.  Passing every fragment
        // through an empty asm makes that visible to hipcc's wait-count bookkeeping: without it the chunk loop carried a
        // descending ladder of s_waitcnt vmcnt(19 .. 3) for "X loads that may still be in flight", which made the NEXT
        // chunk's LDS-DMA land in the middle of this chunk's MFMAs instead of by the next barrier.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int h = 0; h < HB; ++h)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(xf[h][ks]));

        for (; q < qe; ++q, ++it) {
            const int cc = q - rb * ncc, stg = it & 1;
            const uint32_t ln = lane_now(), to = ln & 15u, go = ln >> 4;           // for the addresses of this chunk
            // chunk q has landed (requested one chunk period ago).  vmcnt counts stores too and retires in order: the only
            // operations younger than chunk q's LDS-DMA are the previous chunk's NJ output stores, so a counted wait lets
            // them drain under this chunk's MFMAs (vmcnt(0) here exposed one store round trip per chunk).
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NJ * HB / 2) : "memory");
            __builtin_amdgcn_s_barrier();                      // ... for every wave; stage stg^1 is free
            u32x4 rv[HB][NJ / 2];
            if (RES) {
#pragma unroll
                for (int h = 0; h < HB; ++h)
#pragma unroll
                    for (int u = 0; u < NJ / 2; ++u)
                        rv[h][u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                            rsR, ((uint32_t)(r0 + h * 16) + to) * (uint32_t)(p.ldres * 2) +
                                     ((uint32_t)(cc * CH) + lane_col(go, u)) * 2, 0, 0));
            }
            if (q + 1 < q1) stage_chunk(q + 1, stg ^ 1);

            // ---- 16 k-steps x (4 W fragments) x (2 row halves) = 128 MFMAs; fragment reads one k-step ahead
            uint32_t wa[4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
                wa[m] = ring_base + stg * STAGE_BYTES + to * 1024 + (((go ^ (to & 3)) | ((m ^ (to >> 2)) << 2)) << 4);
            f32x4 acc[HB][NJ];
#pragma unroll
            for (int h = 0; h < HB; ++h)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[h][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            bf16x8 wf[AHEAD + 1][NJ];
#pragma unroll
            for (int a = 0; a < AHEAD; ++a) read_kstep(wf[a], wa, a);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const int cur = ks % (AHEAD + 1);
                if (ks + AHEAD < 16) {
                    read_kstep(wf[(ks + AHEAD) % (AHEAD + 1)], wa, ks + AHEAD);
                    lgkm_wait<AHEAD * NJ>(wf[cur]);
                } else if (ks + 1 < 16 && AHEAD == 2) {
                    lgkm_wait<NJ>(wf[cur]);
                } else {
                    lgkm_wait<0>(wf[cur]);
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int h = 0; h < HB; ++h)
                        acc[h][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cur][j], xf[h][ks], acc[h][j], 0, 0, 0);
            }

            // ---- epilogue: lane (g, t) holds token t's columns chunk0 + lane_col(g, u) + 4 (j & 1) + e, j = 2 u, 2 u + 1
#pragma unroll
            for (int u = 0; u < NJ / 2; ++u) {                   // 8 columns at a time: j = 2u, 2u+1
                f32x4 s0, s1, c0, c1;
                const uint32_t a = smem_base + ((uint32_t)(cc * CH) + lane_col(go, u)) * 4, ac = a + (uint32_t)p.N * 4;
                lds_read_f<0>(s0, a);
                lds_read_f<16>(s1, a);
                lds_read_f<0>(c0, ac);
                lds_read_f<16>(c1, ac);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s0), "+v"(s1), "+v"(c0), "+v"(c1));
#pragma unroll
                for (int h = 0; h < HB; ++h) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float a0 = acc[h][2 * u + (e >> 2)][e & 3];
                        const float sv = (e >> 2) ? s1[e & 3] : s0[e & 3];
                        const float cv = (e >> 2) ? c1[e & 3] : c0[e & 3];
                        float o = LN ? fmaf(rstd[h], fmaf(-mean[h], sv, a0), cv) : a0 + cv;     // pinned contraction
                        if (ACT == BYA_ACT_GELU_ERF) o = gelu_erf_f(o);
                        if (RES) o += (e & 1) ? bfhi(rv[h][u][e >> 1]) : bflo(rv[h][u][e >> 1]);
                        v[e] = o;
                    }
                    u32x4 ov;
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) ov[w2] = pack2bf(v[2 * w2], v[2 * w2 + 1]);
#if BYA_ROWGEMM_ABLATE & 2
                    if (ov[0] == 0x12345678u && ov[3] == 0x9abcdef0u)
#endif
                    __builtin_amdgcn_raw_buffer_store_b128(ov, rsC, ((uint32_t)(r0 + h * 16) + to) * (uint32_t)(p.ldc * 2) +
                                                                        ((uint32_t)(cc * CH) + lane_col(go, u)) * 2, 0, 0);
                }
            }
        }
    }
}

template <bool LN, bool RES, int ACT>
int launch_rowgemm(const RowGemmArgs& a, hipStream_t s) {
    const long long total = (long long)((a.M + RB - 1) / RB) * (a.N / CH);
    const int blocks = (int)(total < 256 * WG_PER_CU ? total : 256 * WG_PER_CU);
    const size_t lds = (size_t)a.N * 8 + 2 * STAGE_BYTES;
    static std::atomic<unsigned long long> attr_done{0};
    if (bya_allow_big_lds(reinterpret_cast<const void*>(rowgemm512_kernel<LN, RES, ACT>), 160 * 1024, attr_done) != BYA_OK)
        return BYA_ERR_LAUNCH;
    BYA_LAUNCH((rowgemm512_kernel<LN, RES, ACT>), dim3(blocks), dim3(64 * NW), lds, s, a);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}


// =====================================================================================================================
// N = 512 form of the row GEMM (the router's out-projections, mlp[0], mlp[2]: 420 launches per step): W-STATIONARY.
// With 8 column chunks per row block the chunk-balanced kernel above gives a CU 4.3 chunks of ~1.6 row blocks: it loads
// 410 KB of X to use 40 % of it, every chunk costs a workgroup barrier, and the X load of the next row block cannot
// overlap anything (X fills the registers).  Here a workgroup keeps ONE QUARTER of W (128 output columns x 512 K = the two
// 64-KiB ring stages, loaded once) in LDS and streams rows past it: the four workgroups of a row group (same XCD: the rows
// they share come from its L2) cover the 512 columns, 64 row groups cover M.  No barrier after the first; a wave walks its
// 16-row tiles two at a time, and X travels through an 8-k-step register ring that is refilled 8 k-steps ahead -- across
// tile pairs too -- so HBM streams steadily under the MFMAs.  (No LayerNorm-folding instance: mlp[0] keeps the
// chunk-balanced kernel; accumulating row statistics from streamed fragments does not fit two waves per SIMD.)
template <bool LN, bool RES, int ACT>
__global__ __launch_bounds__(64 * NW, 2) void rowgemm512q_kernel(RowGemmArgs p) {
    static_assert(!LN, "the W-stationary form has no LayerNorm-folding instance");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per_xcd = gridDim.x >> 5;       // row groups per XCD
    const int quarter = idx & 3, rgrp = xcd * per_xcd + (idx >> 2), ngroups = gridDim.x >> 2;
    const int col_base = quarter * 128;

    // LDS: [colsum: 128 f32][cvec: 128 f32][W quarter: 2 x 64 KiB in the ring's layout]
    float* s_lds = reinterpret_cast<float*>(smem);
    float* c_lds = s_lds + 128;
    char* ring = smem + 1024;
    if (tid < 128) {
        s_lds[tid] = LN ? p.colsum[col_base + tid] : 0.0f;
        c_lds[tid] = p.cvec[col_base + tid];
    }
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.X, 0, (int)(((long long)(p.M - 1) * p.ldx + RK) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, p.N * RK * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.C, 0, (int)(((long long)(p.M - 1) * p.ldc + p.N) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(RES ? p.res : p.C), 0, (int)(((long long)(p.M - 1) * (RES ? p.ldres : p.ldc) + p.N) * 2), 0x00020000);
    if (!(BYA_ROWGEMM_ABLATE & 8)) {   // the W quarter: stage st holds columns col_base + 64 st .. + 63 in the ring's row order (see stage_chunk above)
        const uint32_t l16 = (uint32_t)lane << 4;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int r = 0; r < SR; ++r) {
                const int R = wave * SR + r, i = R & 15, j = R >> 4;
                const uint32_t vo = (l16 ^ (uint32_t)(i << 4)) + wrow_of(i, j) * (RK * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(ring + st * STAGE_BYTES + (wave * SR + r) * 1024), 16, vo,
                                                         (col_base + 64 * st) * (RK * 2), 0, 0);
            }
    }
    // this wave's 16-row tiles: a contiguous eighth of the row group's contiguous share of all tiles
    const int tiles_total = (p.M + 15) / 16;
    const int g0 = (int)((long long)tiles_total * rgrp / ngroups), g1 = (int)((long long)tiles_total * (rgrp + 1) / ngroups);
    const int tb = g0 + (int)((long long)(g1 - g0) * wave / NW), te = g0 + (int)((long long)(g1 - g0) * (wave + 1) / NW);
    const int npair = (te - tb + 1) / 2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // W and the constants are in place; no barrier after this one
    if (npair == 0) return;

    const uint32_t to = (uint32_t)lane & 15u, go = (uint32_t)lane >> 4;
    const uint32_t ring_base = (uint32_t)(uintptr_t)LDS_PTR(ring);
    const uint32_t smem_base = (uint32_t)(uintptr_t)LDS_PTR(smem);
    uint32_t wa[2][4];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int m = 0; m < 4; ++m)
            wa[st][m] = ring_base + st * STAGE_BYTES + to * 1024 + (((go ^ (to & 3)) | ((m ^ (to >> 2)) << 2)) << 4);

    constexpr int D = 8;                                   // k-steps of X in flight = ring slots
    bf16x8 xr[D][HB];
    // byte offset of row (tile, slot to) in a matrix of row stride ld; tiles past this wave's range: outside every descriptor
    auto row_off = [&](int pair, int h, int ld) -> uint32_t {
        const int tile = tb + 2 * pair + h;
        return (pair < npair && tile < te) ? ((uint32_t)tile * 16u + to) * (uint32_t)(ld * 2) : 0x7ffff000u;
    };
    // (hand-counted s_waitcnt around asm loads was tried: the register copies hipcc makes at the pair loop's back edge read
    // slots whose loads are still in flight.  With builtin loads hipcc drains the ring at the head of every pair; measured,
    // that costs nothing -- the kernel is bound by its LDS reads of W, one fragment per two MFMAs.)
#if BYA_ROWGEMM_ABLATE & 1
#define XLOAD(DST, VOFF, OFF) { u32x4 t_ = {(VOFF) + (OFF), (VOFF), 0x3f803f80u, 0x3f803f80u}; asm volatile("" : "+v"(t_)); DST = __builtin_bit_cast(bf16x8, t_); }
#else
#define XLOAD(DST, VOFF, OFF) DST = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsX, (VOFF) + (OFF), 0, 0))
#endif
    uint32_t xo[2][HB];                                    // X row offsets of the pair being computed / the next one
#pragma unroll
    for (int h = 0; h < HB; ++h) {
        xo[0][h] = row_off(0, h, p.ldx) + go * 16;
        xo[1][h] = row_off(1, h, p.ldx) + go * 16;
    }
#pragma unroll
    for (int ks = 0; ks < D; ++ks) {
        XLOAD(xr[ks][0], xo[0][0], ks * 64);
        XLOAD(xr[ks][1], xo[0][1], ks * 64);
    }

    for (int pair = 0; pair < npair; ++pair) {
        f32x4 acc[2][HB][NJ];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int h = 0; h < HB; ++h)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[st][h][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 rv[2][HB][NJ / 2];
        uint32_t ro[HB], co[HB];
#pragma unroll
        for (int h = 0; h < HB; ++h) {
            ro[h] = row_off(pair, h, RES ? p.ldres : p.ldc) + ((uint32_t)col_base + lane_col(go, 0)) * 2;
            co[h] = row_off(pair, h, p.ldc) + ((uint32_t)col_base + lane_col(go, 0)) * 2;
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int slot = ks % D;
            bf16x8 wf[2][NJ];
#if BYA_ROWGEMM_ABLATE & 32
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int j = 0; j < NJ; ++j) { u32x4 t_ = {wa[st][0] + j, wa[st][1], 0x3f803f80u, 0x3f803f80u}; asm volatile("" : "+v"(t_)); wf[st][j] = __builtin_bit_cast(bf16x8, t_); }
            const bf16x8 x0 = xr[slot][0], x1 = xr[slot][1];
#else
#pragma unroll
            for (int st = 0; st < 2; ++st) read_kstep(wf[st], wa[st], ks);
            const bf16x8 x0 = xr[slot][0], x1 = xr[slot][1];
            lgkm_wait<0>(wf[0]);
            lgkm_wait<0>(wf[1]);
#endif
#if BYA_ROWGEMM_ABLATE & 16
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    asm volatile("" : "+v"(acc[st][0][j]) : "v"(wf[st][j]), "v"(x0));
                    asm volatile("" : "+v"(acc[st][1][j]) : "v"(wf[st][j]), "v"(x1));
                }
#else
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[st][0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[st][j], x0, acc[st][0][j], 0, 0, 0);
                    acc[st][1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[st][j], x1, acc[st][1][j], 0, 0, 0);
                }
#endif
            // refill the slot 8 k-steps ahead: the second half of a pair prefetches the first half of the next pair
            if (ks + D < 16) {
                XLOAD(xr[slot][0], xo[0][0], (ks + D) * 64);
                XLOAD(xr[slot][1], xo[0][1], (ks + D) * 64);
            } else {
                XLOAD(xr[slot][0], xo[1][0], (ks + D - 16) * 64);
                XLOAD(xr[slot][1], xo[1][1], (ks + D - 16) * 64);
            }
            if (RES && ks == 7 && !(BYA_ROWGEMM_ABLATE & 4)) {                          // the residual of this pair, under the second half of its MFMAs
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int h = 0; h < HB; ++h)
#pragma unroll
                        for (int u = 0; u < NJ / 2; ++u)
                            rv[st][h][u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                rsR, ro[h] + (64 * st + (CH / 2) * u) * 2, 0, 0));
            }
        }
        // the next pair's offsets move up; the pair after it is looked up (outside the descriptor past the end)
#pragma unroll
        for (int h = 0; h < HB; ++h) {
            xo[0][h] = xo[1][h];
            xo[1][h] = row_off(pair + 2, h, p.ldx) + go * 16;
        }
        // ---- epilogue: lane (g, t) holds token t's columns col_base + 64 st + lane_col(g, u) + 4 (j & 1) + e
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int u = 0; u < NJ / 2; ++u) {
                f32x4 c0, c1;
                const uint32_t ac = smem_base + ((uint32_t)(128 + 64 * st) + lane_col(go, u)) * 4;
                lds_read_f<0>(c0, ac);
                lds_read_f<16>(c1, ac);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0), "+v"(c1));
#pragma unroll
                for (int h = 0; h < HB; ++h) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float o = acc[st][h][2 * u + (e >> 2)][e & 3] + ((e >> 2) ? c1[e & 3] : c0[e & 3]);
                        if (ACT == BYA_ACT_GELU_ERF) o = gelu_erf_f(o);
                        if (RES) o += (e & 1) ? bfhi(rv[st][h][u][e >> 1]) : bflo(rv[st][h][u][e >> 1]);
                        v[e] = o;
                    }
                    u32x4 ov;
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) ov[w2] = pack2bf(v[2 * w2], v[2 * w2 + 1]);
#if BYA_ROWGEMM_ABLATE & 2
                    if (ov[0] == 0x12345678u && ov[3] == 0x9abcdef0u)
#endif
                    __builtin_amdgcn_raw_buffer_store_b128(ov, rsC, co[h] + (64 * st + (CH / 2) * u) * 2, 0, 0);
                }
            }
    }
#undef XLOAD
}

template <bool LN, bool RES, int ACT>
int launch_rowgemm_q(const RowGemmArgs& a, hipStream_t s) {
    const size_t lds = 1024 + 2 * STAGE_BYTES;
    static std::atomic<unsigned long long> attr_done{0};
    if (bya_allow_big_lds(reinterpret_cast<const void*>(rowgemm512q_kernel<LN, RES, ACT>), 160 * 1024, attr_done) != BYA_OK)
        return BYA_ERR_LAUNCH;
    BYA_LAUNCH((rowgemm512q_kernel<LN, RES, ACT>), dim3(256), dim3(64 * NW), lds, s, a);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

// =====================================================================================================================
// Fused  LayerNorm -> q|k|v projection -> grouped tiny attention  for the router's temporal and multi-ID sub-blocks
// (reference models/router.py:476-487: norm2/3 -> diffusers Attention(8 heads x 64) over the 13 frames of a location /
// over the identities of a token).  Unfused, every such sub-block wrote the [R, 1536] q|k|v tensor (108 MB) and read it
// back in a separate 13-key / 2-key attention launch; here it never leaves the CU:
//
//   * rows are GATHERED so that every 16-row MFMA tile holds whole attention groups (one (id, location) with its 13 frames
//     + 3 unused slots; 8 tokens x 2 ids; 4 tokens x 3 ids in 4-slot cells).  The row-stationary loader reads each row straight
//     from global memory, so the gather costs nothing; unused slots read zeros through the buffer descriptor and their
//     stores fall outside it.
//   * the work unit is (row block, head): three 64-column chunks of the packed weight, q_h | k_h | v_h, through the same
//     LDS ring as rowgemm512_kernel.  q and k use the transposed product (W fragment = A operand), which leaves lane
//     (g, t) with 16 features of token t -- exactly an A/B operand fragment of the 16 x 16 x 32 MFMA (the feature order
//     inside the dot product does not matter as long as q and k share it), so  S^T = K . Q^T  of a tile is two MFMAs.
//   * lane (g, t) then holds S^T[keys 4g .. 4g+3, query t]: keys of other groups are masked, the softmax reduces over 4
//     registers and, with two cross-lane steps, over g; the rounded P^T IS the B operand of the 16 x 16 x 16 MFMA.
//   * v uses the straight product (X fragment = A operand): lane (g, i) gets V[tokens 4g .. 4g+3, feature i] = the A
//     operand V^T of that MFMA.  O^T = V^T . P^T lands as lane (g, t) = 16 consecutive features of token t (the W-row
//     staging permutation of the ring is the same as rowgemm512's), scaled by 1 / l and stored with two 16-byte stores.
// Six small MFMAs and ~60 VALU instructions per tile and head replace the 144 MB round trip.  q, k, v and P are rounded
// to bf16 where the unfused path stored them (P: like every flash kernel here).
struct RowAttnArgs {
    const bf16_t* X; const bf16_t* W; const float* colsum; const float* cvec; bf16_t* O;
    int M, ldx, ldo, L, P, G;     // group size, slots per group (power of two >= L), groups per 16-row tile
    long long n_groups, n_inner, outer_stride, seq_stride;
    float eps, scale_log2;
};

constexpr int RA_N = 1536, RA_HEADS = 8;
constexpr int RA_CONST_BYTES = RA_N * 8 + NW * HB * 32 * 4;      // colsum | cvec | per-wave (mean[16], rstd[16]) per tile

// one chunk's 128 MFMAs of a wave; SWAP = false: acc[h][j] = W_j . X_h^T (lane (g, t): W rows 4g+e of block j, token t),
// SWAP = true: acc[h][j] = X_h . W_j^T (lane (g, i): tokens 4g+e, W row i of block j)
template <bool SWAP>
__device__ __forceinline__ void rowattn_chunk(f32x4 (&acc)[HB][NJ], const bf16x8 (&xf)[HB][16], const uint32_t (&wa)[4]) {
#pragma unroll
    for (int h = 0; h < HB; ++h)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[h][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 wf[AHEAD + 1][NJ];
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) read_kstep(wf[a], wa, a);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const int cur = ks % (AHEAD + 1);
        if (ks + AHEAD < 16) {
            read_kstep(wf[(ks + AHEAD) % (AHEAD + 1)], wa, ks + AHEAD);
            lgkm_wait<AHEAD * NJ>(wf[cur]);
        } else {
            lgkm_wait<0>(wf[cur]);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int h = 0; h < HB; ++h)
                acc[h][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[h][ks], wf[cur][j], acc[h][j], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cur][j], xf[h][ks], acc[h][j], 0, 0, 0);
    }
}

// WIDE = false: groups of L <= 16 rows in cells of P = 1, 2, 4, 8 or 16 slots of one tile.  WIDE = true: 16 < L <= 32 (the 25
// latent frames of a 97-frame clip): a group is the two tiles of ONE wave, member m in slot m % 16 of tile m / 16; S^T is the
// 2 x 2 grid of (key tile, query tile) products, the softmax runs over both key tiles, O^T sums two P.V products.
template <bool WIDE>
__global__ __launch_bounds__(64 * NW, 2) void rowattn512_kernel(RowAttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const long long tiles = WIDE ? 2 * p.n_groups : (p.n_groups + p.G - 1) / p.G;
    const int nrb = (int)((tiles + NW * HB - 1) / (NW * HB));
    const int total = nrb * RA_HEADS;                                   // (row block, head) units, row-block-major
    const int rid = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    const int u0 = (int)((long long)total * rid / gridDim.x), u1 = (int)((long long)total * (rid + 1) / gridDim.x);
    if (u0 >= u1) return;

    float* s_lds = reinterpret_cast<float*>(smem);
    float* c_lds = s_lds + RA_N;
    float* st_lds = c_lds + RA_N + wave * (HB * 32);                    // this wave's statistics
    char* ring = smem + RA_CONST_BYTES;
    for (int i = tid; i < RA_N; i += 64 * NW) {
        s_lds[i] = p.colsum[i];
        c_lds[i] = p.cvec[i];
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.X, 0, (int)(((long long)(p.M - 1) * p.ldx + RK) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, RA_N * RK * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.O, 0, (int)(((long long)(p.M - 1) * p.ldo + RK) * 2), 0x00020000);

    // chunk counter qq = 3 * unit + part (0 = q, 1 = k, 2 = v): W rows part * 512 + head * 64 ..
    auto stage_chunk = [&](int qq, int stg) {
        char* dst = ring + stg * STAGE_BYTES + wave * SR * 1024;
        const int u = qq / 3, part = qq - 3 * u;
        const int col0 = part * 512 + (u & (RA_HEADS - 1)) * 64;
        const uint32_t l16 = lane_now() << 4;
#pragma unroll
        for (int r = 0; r < SR; ++r) {
            const int R = wave * SR + r, i = R & 15, j = R >> 4;
            const uint32_t vo = (l16 ^ (uint32_t)(i << 4)) + (uint32_t)(LPC * (i >> 2) + 4 * j + (i & 3)) * (RK * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(dst + r * 1024), 16, vo, col0 * (RK * 2), 0, 0);
        }
    };
    const uint32_t ring_base = (uint32_t)(uintptr_t)LDS_PTR(ring);
    const uint32_t smem_base = (uint32_t)(uintptr_t)LDS_PTR(smem);
    const uint32_t st_base = (uint32_t)(uintptr_t)LDS_PTR(st_lds);

    bf16x8 xf[HB][16];
    const int q_end = 3 * u1;
    stage_chunk(3 * u0, 0);
    int u = u0, stg = 0;
    while (u < u1) {
        const int rb = u >> 3;
        const int ue = (rb + 1) * RA_HEADS < u1 ? (rb + 1) * RA_HEADS : u1;
        float mean[HB], rstd[HB];
        uint32_t rowi[HB];            // the row of this lane's token slot (M = none: loads read zeros, stores are dropped)
        uint32_t kmask;               // bit e (+ 4 hk when WIDE): key slot 4g+e (of key tile hk) belongs to the group of this lane's query
        {
            const uint32_t ln = lane_now(), to = ln & 15u, go = ln >> 4;
            // slot s of a tile = member s % P of the tile's group s / P (P = the power of two >= L, G = 16 / P groups): a
            // group never straddles a 4-slot lane group unless it fills whole ones, so WHERE in a tile a group sits does
            // not change the order in which the matrix core and the row-sum reduction add its terms (the other slots
            // contribute exact zeros) -- results do not depend on how the caller's rows are partitioned (a rank's shard
            // of the router rows vs the whole clip).
            const uint32_t P = WIDE ? 16u : (uint32_t)p.P, gq = WIDE ? 0u : to / P, mq = to - gq * P;
            kmask = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t kk = 4 * go + e, gk = kk / P, mk = kk - gk * P;
                if (WIDE) {
                    kmask |= (kk < (uint32_t)p.L ? 1u : 0u) << e;                 // key tile 0: members 0 .. 15
                    kmask |= (16u + kk < (uint32_t)p.L ? 1u : 0u) << (4 + e);     // key tile 1: members 16 .. 31
                } else {
                    kmask |= (gk == gq && mk < (uint32_t)p.L ? 1u : 0u) << e;
                }
            }
#pragma unroll
            for (int h = 0; h < HB; ++h) {
                const long long tile = (long long)rb * (NW * HB) + wave * HB + h;
                const long long grp = WIDE ? tile / 2 : tile * p.G + gq;
                const uint32_t member = WIDE ? 16u * (uint32_t)h + to : mq;
                const bool ok = member < (uint32_t)p.L && grp < p.n_groups;
                const long long row = (grp / p.n_inner) * p.outer_stride + (grp % p.n_inner) + (long long)member * p.seq_stride;
                rowi[h] = ok ? (uint32_t)row : (uint32_t)p.M;
                const uint32_t vo = rowi[h] * (uint32_t)(p.ldx * 2) + go * 16;
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
                    xf[h][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsX, vo + ks * 64, 0, 0));
            }
            bf16x8 ones;
#pragma unroll
            for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#pragma unroll
            for (int h = 0; h < HB; ++h) {
                f32x4 sm = {0.f, 0.f, 0.f, 0.f}, gr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, xf[h][ks], sm, 0, 0, 0);
                    gr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[h][ks], xf[h][ks], gr, 0, 0, 0);
                }
                const int e = to & 3;
                const float d = e == 0 ? gr[0] : e == 1 ? gr[1] : e == 2 ? gr[2] : gr[3];
                const float sq = __shfl(d, (int)(to + 16 * (to >> 2)));
                const float mu = sm[0] * (1.0f / RK);
                const float var = fmaxf(sq * (1.0f / RK) - mu * mu, 0.0f);
                mean[h] = mu;
                rstd[h] = rsqrtf(var + p.eps);
                if (go == 0) {                        // tokens 4g+e of the v chunk's layout read them from here
                    st_lds[h * 32 + to] = mean[h];
                    st_lds[h * 32 + 16 + to] = rstd[h];
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int h = 0; h < HB; ++h)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(xf[h][ks]));

        for (; u < ue; ++u) {
            const int head = u & (RA_HEADS - 1);
            u32x4 qf[HB][2], kf[HB][2];
            u32x2 pf[HB][WIDE ? HB : 1];
            float invl[HB];
            // ---------------------------------------------------------------- q and k: transposed product + LN epilogue
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const uint32_t ln = lane_now(), to = ln & 15u, go = ln >> 4;
                // only the previous unit's 2 * HB output stores are younger than this chunk's LDS-DMA (see rowgemm512_kernel)
                if (part == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * HB) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (3 * u + part + 1 < q_end) stage_chunk(3 * u + part + 1, stg ^ 1);
                uint32_t wa[4];
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    wa[m] = ring_base + stg * STAGE_BYTES + to * 1024 + (((go ^ (to & 3)) | ((m ^ (to >> 2)) << 2)) << 4);
                f32x4 acc[HB][NJ];
                rowattn_chunk<false>(acc, xf, wa);
                const uint32_t sc_base = smem_base + (uint32_t)(part * 512 + head * 64 + LPC * go) * 4;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    f32x4 s0, s1, c0, c1;
                    const uint32_t a = sc_base + (uint32_t)(8 * kb) * 4, ac = a + (uint32_t)RA_N * 4;
                    lds_read_f<0>(s0, a);
                    lds_read_f<16>(s1, a);
                    lds_read_f<0>(c0, ac);
                    lds_read_f<16>(c1, ac);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s0), "+v"(s1), "+v"(c0), "+v"(c1));
#pragma unroll
                    for (int h = 0; h < HB; ++h) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float a0 = acc[h][2 * kb + (e >> 2)][e & 3];
                            const float sv = (e >> 2) ? s1[e & 3] : s0[e & 3];
                            const float cv = (e >> 2) ? c1[e & 3] : c0[e & 3];
                            v[e] = fmaf(rstd[h], fmaf(-mean[h], sv, a0), cv);
                        }
                        u32x4 w;
#pragma unroll
                        for (int w2 = 0; w2 < 4; ++w2) w[w2] = pack2bf(v[2 * w2], v[2 * w2 + 1]);
                        if (part == 0) qf[h][kb] = w; else kf[h][kb] = w;
                    }
                }
                stg ^= 1;
            }
            // ---------------------------------------------------------------- S^T = K . Q^T, masked softmax over the keys
            constexpr int NKT = WIDE ? HB : 1;                  // key tiles a query attends to
#pragma unroll
            for (int h = 0; h < HB; ++h) {                      // query tile
                float sv[NKT][4];
                float mx = -INFINITY;
#pragma unroll
                for (int hk = 0; hk < NKT; ++hk) {
                    const int kt = WIDE ? hk : h;               // key tile: the other tile of the wave too when WIDE
                    f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
                        st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf[kt][kb]),
                                                                     __builtin_bit_cast(bf16x8, qf[h][kb]), st, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        sv[hk][e] = (kmask >> (4 * hk + e)) & 1u ? st[e] : -INFINITY;
                        mx = fmaxf(mx, sv[hk][e]);
                    }
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                float l = 0.f;
#pragma unroll
                for (int hk = 0; hk < NKT; ++hk) {
                    float pe[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pe[e] = __builtin_amdgcn_exp2f((sv[hk][e] - mx) * p.scale_log2);    // masked keys: exp2(-inf) = 0
                        l += pe[e];
                    }
                    pf[h][hk][0] = pack2bf(pe[0], pe[1]);
                    pf[h][hk][1] = pack2bf(pe[2], pe[3]);
                }
                l += __shfl_xor(l, 16);
                l += __shfl_xor(l, 32);
                invl[h] = __builtin_amdgcn_rcpf(l);
            }
            // ---------------------------------------------------------------- v: straight product, O^T = V^T . P^T, store
            {
                const uint32_t ln = lane_now(), to = ln & 15u, go = ln >> 4;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (3 * u + 3 < q_end) stage_chunk(3 * u + 3, stg ^ 1);
                uint32_t wa[4];
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    wa[m] = ring_base + stg * STAGE_BYTES + to * 1024 + (((go ^ (to & 3)) | ((m ^ (to >> 2)) << 2)) << 4);
                f32x4 acc[HB][NJ];
                rowattn_chunk<true>(acc, xf, wa);
                // lane (g, i): tokens 4g+e, output column 16 (i >> 2) + 4 j + (i & 3) of the head (the ring's staging order)
                const uint32_t col = (uint32_t)(1024 + head * 64) + LPC * (to >> 2) + (to & 3);
                float sj[NJ], cj[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    sj[j] = s_lds[col + 4 * j];
                    cj[j] = c_lds[col + 4 * j];
                }
                // V^T fragments (A operands: lane (g, i) = V[tokens 4g+e of the tile, column of W row i]).  WIDE: of both tiles
                // first, every query tile needs both; else tile by tile (fewer live registers)
                auto make_vt = [&](int h, u32x2 (&vt)[NJ]) {
                    f32x4 m4, r4;
                    lds_read_f<0>(m4, st_base + (uint32_t)(h * 32 + 4 * go) * 4);
                    lds_read_f<64>(r4, st_base + (uint32_t)(h * 32 + 4 * go) * 4);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(m4), "+v"(r4));
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        float vv[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) vv[e] = fmaf(r4[e], fmaf(-m4[e], sj[j], acc[h][j][e]), cj[j]);
                        vt[j][0] = pack2bf(vv[0], vv[1]);
                        vt[j][1] = pack2bf(vv[2], vv[3]);
                    }
                };
                u32x2 vt[WIDE ? HB : 1][NJ];
                if constexpr (WIDE) {
#pragma unroll
                    for (int h = 0; h < HB; ++h) make_vt(h, vt[h]);
                }
#pragma unroll
                for (int h = 0; h < HB; ++h) {                  // query tile
                    if constexpr (!WIDE) make_vt(h, vt[0]);
                    float ov[16];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int hk = 0; hk < NKT; ++hk)
                            o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, vt[WIDE ? hk : 0][j]),
                                                                          __builtin_bit_cast(s16x4, pf[h][hk]), o, 0, 0, 0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) ov[4 * j + e] = o[e] * invl[h];
                    }
                    // lane (g, t): token t, columns head * 64 + 16 g + 4 j + e
                    const uint32_t off = rowi[h] * (uint32_t)(p.ldo * 2) + ((uint32_t)(head * 64) + LPC * go) * 2;
                    __builtin_amdgcn_raw_buffer_store_b128(pack8(ov), rsO, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(pack8(ov + 8), rsO, off + 16, 0, 0);
                }
                stg ^= 1;
            }
        }
    }
}

}  // namespace

extern "C" int bya_rowgemm512(const void* X, const void* W, const float* colsum, const float* cvec, const void* res,
                              void* C, int32_t M, int32_t N, int32_t ldx, int32_t ldc, int32_t ldres, int32_t ln,
                              float eps, int32_t act, int32_t nsplit, hipStream_t stream) {
    (void)nsplit;      // kept in the ABI as a tuning hint; the persistent schedule balances by itself
    if (!X || !W || !cvec || !C || M <= 0 || N <= 0) return BYA_ERR_SHAPE;
    if (ln && !colsum) return BYA_ERR_SHAPE;
    if (N % CH != 0 || N > 4096) return BYA_ERR_SHAPE;             // colsum + cvec + ring must fit 160 KiB of LDS
    if (ldx < RK || ldc < N || (res && ldres < N) || ldx % 8 || ldc % 8 || (res && ldres % 8)) return BYA_ERR_ALIGN;
    if (((uintptr_t)X | (uintptr_t)W | (uintptr_t)C | (uintptr_t)res) & 15) return BYA_ERR_ALIGN;
    if ((long long)M * ldx * 2 >= (1LL << 31) || (long long)M * ldc * 2 >= (1LL << 31)) return BYA_ERR_SHAPE;
    if (act != BYA_ACT_NONE && act != BYA_ACT_GELU_ERF) return BYA_ERR_UNSUPPORTED;
    RowGemmArgs a;
    a.X = (const bf16_t*)X; a.W = (const bf16_t*)W; a.colsum = colsum; a.cvec = cvec; a.res = (const bf16_t*)res;
    a.C = (bf16_t*)C; a.M = M; a.N = N; a.ldx = ldx; a.ldc = ldc; a.ldres = res ? ldres : ldc;
    a.eps = eps;
    const bool gelu = act == BYA_ACT_GELU_ERF;
    // N = 512: the W-stationary, barrier-free form (one W quarter per workgroup).  The rows M * ld must stay below 2 GiB like
    // everywhere here; tiles outside a wave's range are addressed outside the descriptors.
    // measured (profiles/history/r4_v_rowgemm_q_probe.json): -4 % at 35100 rows, -22 % at 17550, -47 % at 8788, -29 % at 4394 and
    // 2194, level at 70200 -- above that the chunk-balanced kernel's finer work units win back what its barriers cost
    if (N == 512 && !ln && M >= 2048 && M <= 65536 && !bya_ref_form(BYA_REF_ROWGEMM_CHUNKED)) {
        // (the LayerNorm-folding instance -- mlp[0] -- keeps the chunk-balanced kernel: accumulating the row statistics from
        // the streamed fragments needs ~40 more registers than two waves per SIMD leave, and spills)
        if (res) return gelu ? launch_rowgemm_q<false, true, BYA_ACT_GELU_ERF>(a, stream) : launch_rowgemm_q<false, true, BYA_ACT_NONE>(a, stream);
        return gelu ? launch_rowgemm_q<false, false, BYA_ACT_GELU_ERF>(a, stream) : launch_rowgemm_q<false, false, BYA_ACT_NONE>(a, stream);
    }
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

extern "C" int bya_router_group_attn(const void* X, const void* Wqkv, const float* colsum, const float* cvec, void* O,
                                     int32_t M, int32_t ldx, int32_t ldo, int32_t L, int64_t n_outer, int64_t n_inner,
                                     int64_t outer_stride, int64_t seq_stride, float eps, float scale, hipStream_t stream) {
    if (!X || !Wqkv || !colsum || !cvec || !O || M <= 0 || n_outer <= 0 || n_inner <= 0) return BYA_ERR_SHAPE;
    if (L < 1 || L > 32) return BYA_ERR_UNSUPPORTED;                  // a group must fit the two 16-row MFMA tiles of a wave
    if (outer_stride < 0 || seq_stride < 0) return BYA_ERR_SHAPE;
    if ((n_outer - 1) * outer_stride + (n_inner - 1) + (int64_t)(L - 1) * seq_stride >= M) return BYA_ERR_SHAPE;
    if (ldx < RK || ldo < RK || ldx % 8 || ldo % 8) return BYA_ERR_ALIGN;
    if (((uintptr_t)X | (uintptr_t)Wqkv | (uintptr_t)O) & 15) return BYA_ERR_ALIGN;
    if (((long long)M + 1) * ldx * 2 >= (1LL << 31) || ((long long)M + 1) * ldo * 2 >= (1LL << 31)) return BYA_ERR_SHAPE;
    RowAttnArgs a;
    a.X = (const bf16_t*)X; a.W = (const bf16_t*)Wqkv; a.colsum = colsum; a.cvec = cvec; a.O = (bf16_t*)O;
    a.M = M; a.ldx = ldx; a.ldo = ldo; a.L = L;
    const bool wide = L > 16;
    a.P = L <= 1 ? 1 : L <= 2 ? 2 : L <= 4 ? 4 : L <= 8 ? 8 : 16;
    a.G = 16 / a.P;
    a.n_groups = n_outer * n_inner; a.n_inner = n_inner; a.outer_stride = outer_stride; a.seq_stride = seq_stride;
    a.eps = eps; a.scale_log2 = scale * 1.4426950408889634f;
    const long long tiles = wide ? 2 * a.n_groups : (a.n_groups + a.G - 1) / a.G;
    const long long total = ((tiles + NW * HB - 1) / (NW * HB)) * RA_HEADS;
    const int blocks = (int)(total < 256 ? total : 256);
    const size_t lds = (size_t)RA_CONST_BYTES + 2 * STAGE_BYTES;
    static std::atomic<unsigned long long> attr_done{0};
    static std::atomic<unsigned long long> attr_done_w{0};
    if (wide) {
        if (bya_allow_big_lds(reinterpret_cast<const void*>(rowattn512_kernel<true>), 160 * 1024, attr_done_w) != BYA_OK) return BYA_ERR_LAUNCH;
        BYA_LAUNCH(rowattn512_kernel<true>, dim3(blocks), dim3(64 * NW), lds, stream, a);
    } else {
        if (bya_allow_big_lds(reinterpret_cast<const void*>(rowattn512_kernel<false>), 160 * 1024, attr_done) != BYA_OK) return BYA_ERR_LAUNCH;
        BYA_LAUNCH(rowattn512_kernel<false>, dim3(blocks), dim3(64 * NW), lds, stream, a);
    }
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}
