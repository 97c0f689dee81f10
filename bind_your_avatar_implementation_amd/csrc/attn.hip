// Flash-attention forward for gfx950, head_dim 64 / 128, no mask, arbitrary Sq / Skv (tails masked).
//
// Replaces F.scaled_dot_product_attention inside diffusers' CogVideoXAttnProcessor2_0 (joint 17776-token
// self-attention, models/transformer.py:208,241), the default SDPA processor of the router's spatial
// attention (models/router.py:476) and of the audio cross-attention (models/audio_model.py:253), and the
// hand-rolled softmax attention of PerceiverCrossAttention / PerceiverAttention (models/router.py:264-270,
// 68-71; their q*s, k*s with s = d^-1/4 is scale = d^-1/2 on the product).
//
// Structure (v1): block = 4 waves x 32 query rows = 128 rows of one (batch, head); K/V tiles of 64 keys
// staged by global_load_lds_dwordx4 into a 2-deep LDS ring (source-side XOR swizzles: K for the
// ds_read_b128 A-operand reads, V for the ds_read_b64_tr_b16 transposed reads).
// QK^T is issued "swapped" (S^T = K.Q^T, v_mfma_f32_32x32x16_bf16) so a query row lives on one lane
// (+ its partner lane+32): the online softmax is an in-register reduction plus one cross-half exchange,
// and the S^T accumulator is directly the B operand of O^T += V^T.P^T (no LDS round trip for P).
// Blocks of one (batch, head) are dealt to one XCD so its K/V stream is served by that XCD's L2.
#include "attn_common.h"
#include "routing_weights.h"
#include <stdlib.h>
#include "options.h"

namespace {

// Ablation build (tools/attn_ablate.py; NEVER defined in the product build): a bit mask of work to leave out of the hot
// loop, results become meaningless, only the time is read.  1: v_exp -> one FMA, 2: K fragments read from LDS once per
// block instead of per tile, 4: V fragments likewise, 8: no K/V staging after the first tile, 16: no row-sum adds.
#ifndef BYA_ATTN_ABLATE
#define BYA_ATTN_ABLATE 0
#endif
// (The build knobs BYA_ATTN_KPREFETCH / OCC / RING of rounds 2-4 -- all K fragment reads of a tile up front: +-1 % on the static-bound
// kernel, -30 % on the running-maximum one (profiles/history/r2_attn_ablation.json); a third K/V stage: no gain -- are gone since
// round 6; 2 is the ring depth they left.)
constexpr int ATTN_RING = 2;     // K/V stages in LDS
constexpr int Q_PER_WAVE = 32;
constexpr int Q_PER_BLOCK = 128;

// Stage a [64 keys][D] bf16 tile (rows of D*2 bytes) into LDS, lane-linear image, swizzled source (kswz / vswz:
// attn_common.h).
template <int D, bool IS_V>
__device__ __forceinline__ void stage_kv(const bf16_t* __restrict__ src, long long row_stride, int kv0, int kv_max,
                                         char* lds_tile, int wave, int lane) {
    constexpr int ROW_BYTES = D * 2;
    constexpr int ROWS_PER_INSTR = 1024 / ROW_BYTES;        // 8 (D=64) or 4 (D=128)
    constexpr int CHUNKS = ROW_BYTES / 16;                   // 8 or 16
    constexpr int INSTR_PER_WAVE = (KV_TILE / ROWS_PER_INSTR) / 4;
#pragma unroll
    for (int q = 0; q < INSTR_PER_WAVE; ++q) {
        const int rbase = (wave * INSTR_PER_WAVE + q) * ROWS_PER_INSTR;
        const int rl = rbase + lane / CHUNKS;
        const int slot = lane % CHUNKS;
        const int chunk = slot ^ (IS_V ? vswz<D>(rl) : kswz<D>(rl));
        int gr = kv0 + rl;
        gr = gr < kv_max ? gr : kv_max;
        const bf16_t* g = src + (long long)gr * row_stride + chunk * 8;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(g), LDS_PTR(lds_tile + rbase * ROW_BYTES), 16, 0, 0);
    }
}

// ds_read_b64_tr_b16 through inline asm: the builtin makes hipcc drain the in-flight LDS-DMA of the NEXT tile
// (s_waitcnt vmcnt(0)) in the middle of the loop; an asm read is invisible to that bookkeeping.  Waits for these
// reads are counted by hand (lgkmcnt), always followed by sched_barrier(0) so no MFMA is hoisted above the wait.
template <int OFF>
__device__ __forceinline__ s16x4 lds_tr_read(uint32_t addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}

template <int OFF>
__device__ __forceinline__ bf16x8 lds_read128(uint32_t addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}

template <int N>
__device__ __forceinline__ void lgkm_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int D>
struct VFrag {
    s16x4 lo[D / 32], hi[D / 32];
};

template <int D, int KS>
__device__ __forceinline__ void v_issue(VFrag<D>& f, const uint32_t (&vbase)[D / 32]) {
    constexpr int RB = D * 2;
    if ((BYA_ATTN_ABLATE & 4) && KS >= 2) return;          // ablation: half of the V reads (k-steps 2, 3 reuse 0, 1)
#pragma unroll
    for (int d = 0; d < D / 32; ++d) {
        f.lo[d] = lds_tr_read<KS * 16 * RB>(vbase[d]);
        f.hi[d] = lds_tr_read<KS * 16 * RB + 8 * RB>(vbase[d]);
    }
}

template <int D>
__device__ __forceinline__ void pv_mfma(const VFrag<D>& f, const bf16x8& pf, f32x16 (&oacc)[D / 32]) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll
    for (int d = 0; d < D / 32; ++d) {
        const s16x8 both = {f.lo[d][0], f.lo[d][1], f.lo[d][2], f.lo[d][3], f.hi[d][0], f.hi[d][1], f.hi[d][2], f.hi[d][3]};
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, both), pf, oacc[d], 0, 0, 0);
    }
}

constexpr float RESCALE_THR = 6.0f;   // skip the O rescale while the running max grows by < 2^6 (P stays < 64)

// One 64-key tile: S^T = K.Q^T, online softmax, O^T += V^T.P^T.
// TAIL: only the first kv_valid keys of the tile exist (scores of the rest are forced to -inf -> P = 0).
//
// VALU diet (the kernel is VALU-issue-bound at head_dim 64: rocprofv3 SQ_ACTIVE_INST_VALU was 1.7x the MFMA busy
// time): q is pre-scaled by scale*log2(e) when it is loaded, and the running maximum is subtracted BY THE MATRIX PIPE:
// each score chain starts with one extra MFMA  ones[32 x 16] . mfrag[16 x 32]  whose B operand carries -m_run of the
// lane's query row (split into two bf16 terms, k = 0 and 1), so the accumulator already holds  s*c - m_run  and the
// common path is ONE v_exp_f32 per score (no per-element fma/sub; +2 MFMAs per tile, which the matrix pipe has room
// for).  m_run moves only when some row's maximum grows by more than 2^RESCALE_THR (wave-uniform branch); the first
// tile always sets it.  m_run is kept equal to the two-term bf16 value actually subtracted, so the algebra is exact.
// PRESCALED = true: the caller already folded scale*log2(e) into k (bya_qknorm_rope's k_scale applies it in fp32
// BEFORE k is rounded to bf16, so there is no extra rounding): scores are in exp2 units, one v_exp per score.
// false: scores and m_run stay in raw units and c = scale*log2(e) is applied in fp32 right before the exp
// (one extra v_mul per score).  Things tried and measured slower on MI355X (kept out of the tree): a 3-stage ring with
// S(t+1) issued before softmax(t) at 2 waves/SIMD (884 vs 910 TFLOP/s), two 32-row query blocks per wave sharing the
// K/V fragments (884 vs 933), pre-scaling q in the kernel (faster, but the second rounding of q costs 1e-3 accuracy).
// BOUNDED (with PRESCALED): the caller guarantees |score| <= B in exp2 units (the engine derives B from the q/k
// LayerNorm weights: ||LN(x) * gamma + beta|| <= 8 max|gamma| + ||beta||, RoPE is a rotation) and passes it in ``c``.
// Softmax is shift-invariant and with B <= 48 nothing can overflow or vanish without a shift at all: every
// P = exp2(s) lies in [2^-48, 2^48] (bf16 and fp32 share the 8-bit exponent; a row sum stays below 2^48 * Skv), so the
// whole maximum machinery -- 16 v_max3, the lane swap, the ballot, the rescale path and the two extra MFMAs --
// disappears from the tile and the score chains start from the MFMA's inline zero.
template <int D, bool TAIL, bool PRESCALED, bool BOUNDED = false>
__device__ __forceinline__ void attn_tile(const char* kt, const uint32_t (&vbase)[D / 32], const bf16x8 (&qf)[D / 16],
                                          f32x16 (&oacc)[D / 32], const bf16x8& ones, bf16x8& mfrag, float& m_run,
                                          float& l_run, bool first, int kv_valid, int r, int hf, float c) {
    const float thr = PRESCALED ? RESCALE_THR : RESCALE_THR / c;
    constexpr int ROW_BYTES = D * 2, DSTEPS = D / 16, DT = D / 32;
    f32x16 sacc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[u][i] = 0.f;
        if (!BOUNDED) sacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, mfrag, sacc[u], 0, 0, 0);  // = -m_run everywhere
        const int krow = u * 32 + r;
#pragma unroll
        for (int s = 0; s < DSTEPS; ++s) {
            const int chunk = 2 * s + hf;
            const int off = krow * ROW_BYTES + ((chunk ^ kswz<D>(krow)) << 4);
            const bf16x8 kf = (BYA_ATTN_ABLATE & 2) ? qf[(s + 1 + u) % DSTEPS] : *reinterpret_cast<const bf16x8*>(kt + off);   // (+ u: no CSE of the two chains)
            sacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc[u], 0, 0, 0);
        }
    }
    // first V fragments can fly while the softmax runs
    VFrag<D> fa, fb;
    v_issue<D, 0>(fa, vbase);

    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (TAIL) {
                const int kv = u * 32 + (i & 3) + 8 * (i >> 2) + 4 * hf;
                if (kv >= kv_valid) sacc[u][i] = -INFINITY;
            }
            if (!BOUNDED) mx = fmaxf(mx, sacc[u][i]);
        }
    if (!BOUNDED) {
    {   // partner lane (lane ^ 32) holds the other 32 keys of this query row: one v_permlane32_swap, no LDS
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));        // row maximum RELATIVE to m_run
    }
    if (first || __builtin_amdgcn_ballot_w64(mx > thr) != 0) {      // wave-uniform slow path (rare)
        float delta = first ? mx : fmaxf(mx, 0.f);
        // new reference = a value exactly representable as hi + lo (two bf16 terms); delta is what really changes
        const float m_want = m_run + delta;
        const float m_hi = bf2f(f2bf(m_want));
        const float m_lo = bf2f(f2bf(m_want - m_hi));
        const float m_new = m_hi + m_lo;
        delta = m_new - m_run;
        if (!first) {
            const float alpha = __builtin_amdgcn_exp2f(PRESCALED ? -delta : -delta * c);
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[d][i] *= alpha;
        }
        m_run = m_new;
        if (hf == 0) {                               // lanes 0..31 carry k = 0..7 of the B operand
            mfrag[0] = (__bf16)(-m_hi);
            mfrag[1] = (__bf16)(-m_lo);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[u][i] -= delta;    // this tile's scores were taken against the old m_run
    }
    }
    float psum = 0.f;
    bf16x8 pf[4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sv = PRESCALED ? sacc[u][tt * 8 + e] : sacc[u][tt * 8 + e] * c;
                const float pv = (BYA_ATTN_ABLATE & 1) ? fmaf(sv, 0.015625f, 1.0f) : __builtin_amdgcn_exp2f(sv);
                if (!(BYA_ATTN_ABLATE & 16)) psum += pv;
                pf[u * 2 + tt][e] = (__bf16)pv;
            }
    l_run += psum;

    // O^T += V^T . P^T, k-step ks covers tile keys 16ks + 8(e>>2) + 4hf + (e&3); reads run one step ahead.
    v_issue<D, 1>(fb, vbase);
    lgkm_wait<2 * DT>();
    pv_mfma<D>(fa, pf[0], oacc);
    v_issue<D, 2>(fa, vbase);
    lgkm_wait<2 * DT>();
    pv_mfma<D>(fb, pf[1], oacc);
    v_issue<D, 3>(fb, vbase);
    lgkm_wait<2 * DT>();
    pv_mfma<D>(fa, pf[2], oacc);
    lgkm_wait<0>();
    pv_mfma<D>(fb, pf[3], oacc);
}

template <int D, bool PRESCALED, bool BOUNDED = false>
__device__ __forceinline__ void attn_fwd_body(const AttnArgs& p, char* smem) {
    constexpr int ROW_BYTES = D * 2;
    constexpr int TILE_BYTES = KV_TILE * ROW_BYTES;
    constexpr int DSTEPS = D / 16;   // k-steps of the QK^T product
    constexpr int DT = D / 32;       // 32-row tiles of O^T
    constexpr int RING = ATTN_RING;
    // LDS ring: stage b holds K at smem + b*2*TILE_BYTES and V right behind it

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hf = lane >> 5;

    // block -> (bh, q-tile): blocks with equal (blockIdx % 8) share an XCD; give each XCD whole (batch, head)s.
    // When the (batch, head) count is not a multiple of 8 -- e.g. 6 heads per rank under head-parallel sharding, 26
    // (id, frame) pairs x 8 heads is fine -- every XCD takes a CONTIGUOUS eighth of the (head, q-tile) order instead:
    // it then works on one or two heads at a time (their K/V fit its L2) and every CU stays busy.  (Plain block order
    // spread every head over all eight L2s: 930 instead of 1150 TFLOP/s at 6 heads x 17776 tokens.)
    const int nbh = p.nb1 * p.nb2 * p.heads;
    int bh, qt;
    if (nbh % 8 == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        bh = (j / p.nqt) * 8 + xcd;
        qt = j % p.nqt;
    } else {
        const int total = nbh * p.nqt, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int cq = total >> 3, cr = total & 7;
        const int base = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
        if (j >= cq + (xcd < cr ? 1 : 0)) return;
        bh = (base + j) / p.nqt;
        qt = (base + j) % p.nqt;
    }
    if (bh >= nbh) return;
    if (p.only_flagged && !p.only_flagged[bh]) return;       // second pass of the device-bound form: flagged heads only
    const int head = bh % p.heads;
    const int b12 = bh / p.heads;
    const int b1 = b12 / p.nb2, b2 = b12 % p.nb2;

    const bf16_t* Q = p.q + b1 * p.q_s1 + b2 * p.q_s2 + (long long)head * D;
    const bf16_t* K = p.k + b1 * p.k_s1 + b2 * p.k_s2 + (long long)head * D;
    const bf16_t* V = p.v + b1 * p.v_s1 + b2 * p.v_s2 + (long long)head * D;
    bf16_t* O = p.o + b1 * p.o_s1 + b2 * p.o_s2 + (long long)head * D;

    // ---- Q fragments (B operand of S^T = K.Q^T): lane (r,hf) holds Q[q0+r][16s + 8hf .. +7]
    const int q0 = qt * Q_PER_BLOCK + wave * Q_PER_WAVE;
    int qrow = q0 + r;
    const bool q_valid = qrow < p.Sq;
    qrow = q_valid ? qrow : p.Sq - 1;
    bf16x8 qf[DSTEPS];
#pragma unroll
    for (int s = 0; s < DSTEPS; ++s) {
        const u32x4 raw = *reinterpret_cast<const u32x4*>(Q + (long long)qrow * p.q_row + s * 16 + hf * 8);
        qf[s] = __builtin_bit_cast(bf16x8, raw);
    }

    f32x16 oacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
    bf16x8 ones, mfrag;                  // A = 1 at k = 0, 1 (lanes 0..31), B = (-m_hi, -m_lo) of the lane's query row
#pragma unroll
    for (int e = 0; e < 8; ++e) { ones[e] = (__bf16)((hf == 0 && e < 2) ? 1.0f : 0.0f); mfrag[e] = (__bf16)0.0f; }
    float m_run = 0.f, l_run = 0.f;

    // per-lane LDS byte offsets of the transposed V reads inside a V tile (everything but the k-step is
    // lane-constant; the k-step and the lo/hi half go into the instruction's immediate offset)
    const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    uint32_t voff[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) {
        const int row = 4 * hf + tq;
        const int chunk = 4 * d + 2 * (g & 1) + (tp >> 1);
        voff[d] = row * ROW_BYTES + ((chunk ^ vswz<D>(row)) << 4) + (tp & 1) * 8;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);

    // LDS-DMA staging through buffer_load ... lds: loop-invariant per-lane offsets, the tile advances in the scalar
    // offset, keys past Skv are outside the descriptor -> zeros land in LDS (their scores are masked in the tail tile).
    constexpr int ROWS_PER_INSTR = 1024 / ROW_BYTES, CHUNKS = ROW_BYTES / 16, NPIECE = 16 / ROWS_PER_INSTR;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(
        (void*)K, 0, (int)(((long long)(p.Skv - 1) * p.k_row + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(
        (void*)V, 0, (int)(((long long)(p.Skv - 1) * p.v_row + D) * 2), 0x00020000);
    uint32_t kvo[NPIECE], vvo[NPIECE];
#pragma unroll
    for (int q = 0; q < NPIECE; ++q) {
        const int row = wave * 16 + q * ROWS_PER_INSTR + lane / CHUNKS, slot = lane % CHUNKS;
        kvo[q] = (uint32_t)row * (uint32_t)(p.k_row * 2) + ((slot ^ kswz<D>(row)) << 4);
        vvo[q] = (uint32_t)row * (uint32_t)(p.v_row * 2) + ((slot ^ vswz<D>(row)) << 4);
    }
    const int k_tile_stride = KV_TILE * (int)p.k_row * 2, v_tile_stride = KV_TILE * (int)p.v_row * 2;
    auto stage = [&](int t) {
        char* st = smem + (RING == 2 ? (t & 1) : (t % 3)) * 2 * TILE_BYTES + wave * 16 * ROW_BYTES;
#pragma unroll
        for (int q = 0; q < NPIECE; ++q) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, LDS_PTR(st + q * 1024), 16, kvo[q], t * k_tile_stride, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, LDS_PTR(st + TILE_BYTES + q * 1024), 16, vvo[q],
                                                     t * v_tile_stride, 0, 0);
        }
    };

    const int ntiles = (p.Skv + KV_TILE - 1) / KV_TILE;
    const int nfull = p.Skv / KV_TILE;
    stage(0);
    if (RING == 3 && ntiles > 1) stage(1);
    int slot = 0;                                         // ring slot of tile t

    for (int t = 0; t < nfull; ++t) {                     // full tiles: no masking code in the hot loop
        if constexpr (RING == 3) {
            // tile t has landed once all but the 2 NPIECE younger requests (tile t + 1) are done; raw barrier: a
            // __syncthreads() would drain the LDS-DMA in flight.  Tile t + 2 goes where tile t - 1 was: everybody is past it.
            if (t + 1 < ntiles) { if constexpr (NPIECE == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            if (t + 2 < ntiles) stage(t + 2);
        } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < ntiles && (!(BYA_ATTN_ABLATE & 8) || t == 0)) stage(t + 1);
        }
        const char* kt = smem + slot * 2 * TILE_BYTES;
        uint32_t vbase[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) vbase[d] = lds0 + slot * 2 * TILE_BYTES + TILE_BYTES + voff[d];
        slot = (slot + 1 == RING) ? 0 : slot + 1;
        int tz = t;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+s"(tz));                       // opaque: keeps hipcc from peeling the first iteration
#endif
        attn_tile<D, false, PRESCALED, BOUNDED>(kt, vbase, qf, oacc, ones, mfrag, m_run, l_run, tz == 0, KV_TILE, r, hf,
                                                BOUNDED ? p.score_bound : p.scale_log2);
    }
    if (nfull < ntiles) {                                 // ragged last tile
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* kt = smem + slot * 2 * TILE_BYTES;
        uint32_t vbase[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) vbase[d] = lds0 + slot * 2 * TILE_BYTES + TILE_BYTES + voff[d];
        attn_tile<D, true, PRESCALED, BOUNDED>(kt, vbase, qf, oacc, ones, mfrag, m_run, l_run, nfull == 0,
                                               p.Skv - nfull * KV_TILE, r, hf, BOUNDED ? p.score_bound : p.scale_log2);
    }

    // ---- epilogue: O[q][d] = O^T[d][q] / l ; lane (r,hf) holds d = 32dt + (i&3) + 8(i>>2) + 4hf
    const auto lsw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    const float l_tot = __uint_as_float(lsw[0]) + __uint_as_float(lsw[1]);
    const float inv = 1.0f / l_tot;
    bf16_t* orow = O + (long long)(q_valid ? q0 + r : 0) * p.o_row;
#pragma unroll
    for (int d = 0; d < DT; ++d) store_o_tile(orow + d * 32, oacc[d], inv, hf, q_valid, p.o_wide != 0);
}

// ------------------------------------------------------------------------------------------------------------------
// Cross-attention onto a handful of keys per identity, with the router's masked combine in its epilogue
// (bya_attn_kv_mix): the audio cross-attention (32 keys per latent frame and identity, 48 heads of 64;
// models/audio_model.py:247-258) and the face Perceiver cross-attention (32 face tokens per identity, 16 heads of 128;
// models/router.py:255-270).  Both are followed in the reference by  hidden += sum_id w[n, id] * to_out(o[id, n, :])
// (models/transformer.py:821-832, 895-936); to_out is linear, so the engine mixes first and projects once --
// and the mix needs nothing but the attention outputs of ONE token.  Unfused that was: attention writes o for every
// identity (2 x 108 MB), bya_routed_mix reads them back and writes z.  Here a workgroup holds its 128 query rows (shared
// by all identities), runs the one-tile attention once per identity on the K / V of that identity and accumulates
//     z = sum_id w[n, id] * (O_id / l_id)      in fp32, rounded to bf16 once,
// so o never reaches HBM: q is read once, z written once.  The per-identity attention is attn_tile<TAIL> of the generic
// kernel above (single tile, running maximum, keys past Skv masked), the weights are routing_weights.h.
struct MixArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* z; const bf16_t* r; const bf16_t* af; float* wsum;
    int heads, n_id, n_grp, Sq, Skv, nqt, mode;
    long long q_grp, q_row, k_id, k_grp, k_row, v_id, v_grp, v_row, z_grp, z_row;
    float scale_log2;
};

template <int D>
__device__ __forceinline__ void attn_mix_body(const MixArgs& p, char* smem) {
    constexpr int ROW_BYTES = D * 2, TILE_BYTES = KV_TILE * ROW_BYTES, DSTEPS = D / 16, DT = D / 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hf = lane >> 5;
    int bid = blockIdx.x;
    const int qt = bid % p.nqt; bid /= p.nqt;
    const int head = bid % p.heads;
    const int grp = bid / p.heads;
    const bf16_t* Q = p.q + grp * p.q_grp + (long long)head * D;
    bf16_t* Z = p.z + grp * p.z_grp + (long long)head * D;
    const int q0 = qt * Q_PER_BLOCK + wave * Q_PER_WAVE;
    int qrow = q0 + r;
    const bool q_valid = qrow < p.Sq;
    qrow = q_valid ? qrow : p.Sq - 1;
    bf16x8 qf[DSTEPS];
#pragma unroll
    for (int s = 0; s < DSTEPS; ++s)
        qf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Q + (long long)qrow * p.q_row + s * 16 + hf * 8));
    float w[4];
    routing_weights_of(p.mode, p.n_id, p.af, p.r + ((long long)grp * p.Sq + qrow) * p.n_id, w);
    if (p.wsum && head == 0 && hf == 0 && q_valid) {
        float ws = 0.f;
        for (int i = 0; i < p.n_id; ++i) ws += w[i];
        p.wsum[(long long)grp * p.Sq + qrow] = ws;
    }

    const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    uint32_t voff[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) {
        const int row = 4 * hf + tq;
        const int chunk = 4 * d + 2 * (g & 1) + (tp >> 1);
        voff[d] = row * ROW_BYTES + ((chunk ^ vswz<D>(row)) << 4) + (tp & 1) * 8;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)((hf == 0 && e < 2) ? 1.0f : 0.0f);

    f32x16 zacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) zacc[d][i] = 0.f;
    for (int id = 0; id < p.n_id; ++id) {
        char* st = smem + (id & 1) * 2 * TILE_BYTES;           // two stages
        auto stage_id = [&](int j) {
            char* dst = smem + (j & 1) * 2 * TILE_BYTES;
            stage_kv<D, false>(p.k + j * p.k_id + grp * p.k_grp + (long long)head * D, p.k_row, 0, p.Skv - 1, dst, wave, lane);
            stage_kv<D, true>(p.v + j * p.v_id + grp * p.v_grp + (long long)head * D, p.v_row, 0, p.Skv - 1, dst + TILE_BYTES, wave, lane);
        };
        // the K/V of the first TWO identities are requested together (one memory round trip and one rendezvous for the
        // common two-identity case: the kernel is latency-bound, 13 rounds of small workgroups); identity id + 1 >= 3
        // lands in the stage identity id - 1 has left while id computes
        if (id == 0) {
            stage_id(0);
            if (p.n_id > 1) stage_id(1);
        }
        if (id != 1) {                                          // identity 1 landed together with identity 0
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (id >= 1 && id + 1 < p.n_id) {
            if (id == 1) __syncthreads();                       // every wave is done with identity 0's stage
            stage_id(id + 1);
        }
        f32x16 oacc[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
        bf16x8 mfrag;
#pragma unroll
        for (int e = 0; e < 8; ++e) mfrag[e] = (__bf16)0.0f;
        float m_run = 0.f, l_run = 0.f;
        uint32_t vbase[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) vbase[d] = lds0 + (id & 1) * 2 * TILE_BYTES + TILE_BYTES + voff[d];
        attn_tile<D, true, false>(st, vbase, qf, oacc, ones, mfrag, m_run, l_run, true, p.Skv, r, hf, p.scale_log2);
        const auto lsw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        const float sc = w[id] / (__uint_as_float(lsw[0]) + __uint_as_float(lsw[1]));
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) zacc[d][i] = fmaf(sc, oacc[d][i], zacc[d][i]);
    }
    if (q_valid) {
        bf16_t* zrow = Z + (long long)(q0 + r) * p.z_row;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                u32x2 o;
                o[0] = pack2bf(zacc[d][gq * 4 + 0], zacc[d][gq * 4 + 1]);
                o[1] = pack2bf(zacc[d][gq * 4 + 2], zacc[d][gq * 4 + 3]);
                *reinterpret_cast<u32x2*>(zrow + d * 32 + gq * 8 + hf * 4) = o;
            }
    }
}

// ---- the <= 32-key form (both callers: 32 audio context tokens per frame, 32 face tokens per identity) --------------
// attn_mix_body runs ONE 128-row query tile per workgroup: stage K / V (64-row tiles, half of them padding), rendezvous,
// attention on a 64-key tile with the upper half masked, store -- 6864 workgroups per audio launch, each a serial chain
// of memory round trips with a barrier in the middle, twice the MFMAs and exps the 32 keys need: 87 us for 218 MB.
// Here a workgroup owns one (group, head) and a CHUNK of its query rows: the K / V of every identity are staged once
// (32 rows each) and stay in LDS, after the one rendezvous every wave walks its own 32-row tiles (tile w, w + 4, ... of the
// chunk) with no further barrier: per tile 4 K-fragment reads, D / 16 MFMAs for S^T, one softmax over 32 keys (two lanes per
// query), D / 16 MFMAs for O^T, per identity.  blockIdx runs over the heads fastest, so the workgroups in flight
// together read the 128-byte head segments of the SAME rows (whole 6-KiB rows over a short time, not one segment per
// row spread over the launch).  Arithmetic per element is that of attn_tile<TAIL> on a first tile (the two-term bf16
// maximum, the order of the row sum, k-steps 0 and 1 of P.V) -- so results are BIT-IDENTICAL to attn_mix_body and, for
// one-hot masks, to bya_attn_fwd's rows (tests/test_kernels_gpu.py).
template <int D>
__device__ __forceinline__ void stage_kv32(const bf16_t* __restrict__ src, long long row_stride, int kv_max, char* lds_tile,
                                           int wave, int lane, bool is_v) {
    constexpr int ROW_BYTES = D * 2, ROWS_PER_INSTR = 1024 / ROW_BYTES, CHUNKS = ROW_BYTES / 16;
    for (int q = wave; q < 32 / ROWS_PER_INSTR; q += 4) {
        const int rbase = q * ROWS_PER_INSTR;
        const int rl = rbase + lane / CHUNKS;
        const int slot = lane % CHUNKS;
        const int chunk = slot ^ (is_v ? vswz<D>(rl) : kswz<D>(rl));
        const int gr = rl < kv_max ? rl : kv_max;
        const bf16_t* g = src + (long long)gr * row_stride + chunk * 8;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(g), LDS_PTR(lds_tile + rbase * ROW_BYTES), 16, 0, 0);
    }
}

template <int D>
__device__ __forceinline__ void attn_mix32_body(const MixArgs& p, char* smem) {
    constexpr int ROW_BYTES = D * 2, KV_BYTES = 32 * ROW_BYTES, DSTEPS = D / 16, DT = D / 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hf = lane >> 5;
    int bid = blockIdx.x;
    const int head = bid % p.heads; bid /= p.heads;
    const int chunk = bid % p.nqt;                               // nqt = row chunks per (group, head) in this form
    const int grp = bid / p.nqt;
    const int n32 = (p.Sq + 31) >> 5;
    const int t0 = (int)((long long)n32 * chunk / p.nqt), t1 = (int)((long long)n32 * (chunk + 1) / p.nqt);
    const bf16_t* Q = p.q + grp * p.q_grp + (long long)head * D;
    bf16_t* Z = p.z + grp * p.z_grp + (long long)head * D;

    for (int id = 0; id < p.n_id; ++id) {
        stage_kv32<D>(p.k + id * p.k_id + grp * p.k_grp + (long long)head * D, p.k_row, p.Skv - 1, smem + id * 2 * KV_BYTES,
                      wave, lane, false);
        stage_kv32<D>(p.v + id * p.v_id + grp * p.v_grp + (long long)head * D, p.v_row, p.Skv - 1,
                      smem + id * 2 * KV_BYTES + KV_BYTES, wave, lane, true);
    }
    const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    uint32_t voff[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) {
        const int row = 4 * hf + tq;
        const int ch = 4 * d + 2 * (g & 1) + (tp >> 1);
        voff[d] = row * ROW_BYTES + ((ch ^ vswz<D>(row)) << 4) + (tp & 1) * 8;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
    uint32_t koff[DSTEPS];
#pragma unroll
    for (int s = 0; s < DSTEPS; ++s) koff[s] = r * ROW_BYTES + (((2 * s + hf) ^ kswz<D>(r)) << 4);
    const float c = p.scale_log2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                             // K / V of every identity are in LDS; no barrier below

    // (measured and dropped: q as whole head segments by LDS-DMA into the wave's patch + ds_read_b128 fragments -- audio
    // level, face 42 -> 49 us; q requested one tile ahead in registers -- +35 registers, one wave per SIMD less, level)
    for (int t = t0 + wave; t < t1; t += 4) {
        int qrow = t * 32 + r;
        const bool q_valid = qrow < p.Sq;
        qrow = q_valid ? qrow : p.Sq - 1;
        bf16x8 qf[DSTEPS];
#pragma unroll
        for (int s = 0; s < DSTEPS; ++s)
            qf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Q + (long long)qrow * p.q_row + s * 16 + hf * 8));
        float w[4];
        routing_weights_of(p.mode, p.n_id, p.af, p.r + ((long long)grp * p.Sq + qrow) * p.n_id, w);
        if (p.wsum && head == 0 && hf == 0 && q_valid) {
            float ws = 0.f;
            for (int i = 0; i < p.n_id; ++i) ws += w[i];
            p.wsum[(long long)grp * p.Sq + qrow] = ws;
        }
        char* zb = smem + p.n_id * 2 * KV_BYTES + wave * KV_BYTES;
        constexpr int CHUNKS = ROW_BYTES / 16;
        f32x16 zacc[DT];
        for (int id = 0; id < p.n_id; ++id) {
            const uint32_t kb = lds0 + id * 2 * KV_BYTES, vb = kb + KV_BYTES;
            // S^T = K . Q^T (32 keys x 32 queries): lane (q = r, hf) gets keys (i & 3) + 8 (i >> 2) + 4 hf
            bf16x8 kf[DSTEPS];
#pragma unroll
            for (int s = 0; s < DSTEPS; ++s) kf[s] = lds_read128<0>(kb + koff[s]);
            f32x16 sacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
            lgkm_wait<0>();
#pragma unroll
            for (int s = 0; s < DSTEPS; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s], qf[s], sacc, 0, 0, 0);
            // the V fragments fly under the softmax
            VFrag<D> fa, fb;
            uint32_t vbase[DT];
#pragma unroll
            for (int d = 0; d < DT; ++d) vbase[d] = vb + voff[d];
            v_issue<D, 0>(fa, vbase);
            v_issue<D, 1>(fb, vbase);
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kv = (i & 3) + 8 * (i >> 2) + 4 * hf;
                if (kv >= p.Skv) sacc[i] = -INFINITY;
                mx = fmaxf(mx, sacc[i]);
            }
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            // the reference point attn_tile takes on a first tile: the maximum as a two-term bf16 value
            const float m_hi = bf2f(f2bf(mx));
            const float m_lo = bf2f(f2bf(mx - m_hi));
            const float m_new = m_hi + m_lo;
            float psum = 0.f;
            bf16x8 pf[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = __builtin_amdgcn_exp2f((sacc[tt * 8 + e] - m_new) * c);
                    psum += pv;
                    pf[tt][e] = (__bf16)pv;
                }
            const auto lsw = __builtin_amdgcn_permlane32_swap(__float_as_uint(psum), __float_as_uint(psum), false, false);
            const float sc = w[id] / (__uint_as_float(lsw[0]) + __uint_as_float(lsw[1]));
            f32x16 oacc[DT];
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
            lgkm_wait<2 * DT>();
            pv_mfma<D>(fa, pf[0], oacc);
            lgkm_wait<0>();
            pv_mfma<D>(fb, pf[1], oacc);
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) zacc[d][i] = id == 0 ? fmaf(sc, oacc[d][i], 0.f) : fmaf(sc, oacc[d][i], zacc[d][i]);
        }
        // ---- z through this wave's LDS patch, stored as WHOLE head segments: a lane holds 8-byte pieces of its row (d = 32 dt
        // + 8 gq + 4 hf ..+3); stored as they stand that is 8 DT instructions each touching 32 rows with 16 bytes -- sixteen
        // partial-line requests per 128-byte segment.  Re-read row-major, 16 bytes per lane: every request a full line.
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                u32x2 o;
                o[0] = pack2bf(zacc[d][gq * 4 + 0], zacc[d][gq * 4 + 1]);
                o[1] = pack2bf(zacc[d][gq * 4 + 2], zacc[d][gq * 4 + 3]);
                *reinterpret_cast<u32x2*>(zb + r * ROW_BYTES + (((4 * d + gq) ^ (r & 7)) << 4) + 8 * hf) = o;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < 32 * CHUNKS / 64; ++j) {
            const int idx = j * 64 + lane, row = idx / CHUNKS, ch = idx % CHUNKS;
            const u32x4 o = *reinterpret_cast<const u32x4*>(zb + row * ROW_BYTES + ((ch ^ (row & 7)) << 4));
            if (t * 32 + row < p.Sq) *reinterpret_cast<u32x4*>(Z + (long long)(t * 32 + row) * p.z_row + ch * 8) = o;
        }
        __builtin_amdgcn_wave_barrier();                         // the patch is rewritten by the next tile
    }
}

__global__ __launch_bounds__(256, 2) void attn_kv_mix32_kernel_d64(MixArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_mix32_body<64>(p, smem);
}
__global__ __launch_bounds__(256) void attn_kv_mix32_kernel_d128(MixArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_mix32_body<128>(p, smem);
}

__global__ __launch_bounds__(256, 2) void attn_kv_mix_kernel_d64(MixArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_mix_body<64>(p, smem);
}
__global__ __launch_bounds__(256) void attn_kv_mix_kernel_d128(MixArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_mix_body<128>(p, smem);
}

// ------------------------------------------------------------------------------------------------------------------
// (rounds 2-4 had a two-query-blocks-per-wave static-bound kernel here, attn_fwd_kernel_d64_bounded2 -- the joint attention
// until the hand-placed one-wave-per-SIMD kernel of attn_w4.hip replaced it, then its BYA_ATTN_W4=0 A/B arm; retired in round 5
// together with the one-block bounded instance: every bounded launch now runs attn_w4.hip.  git history has the code.)

// (plain __global__ wrappers: hipcc's host pass did not emit the launch stub of the templated kernel once its body
// used the buffer-resource builtins)
__global__ __launch_bounds__(256, 4) void attn_fwd_kernel_d64(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_fwd_body<64, false>(p, smem);
}
__global__ __launch_bounds__(256, 4) void attn_fwd_kernel_d64_prescaled(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_fwd_body<64, true>(p, smem);
}
__global__ __launch_bounds__(256) void attn_fwd_kernel_d128(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_fwd_body<128, false>(p, smem);
}

inline float static_bound_limit() { return BYA_ATTN_BOUND_LIMIT; }

template <int D>
int launch_attn(const AttnArgs& a, hipStream_t s) {
    const int nbh = a.nb1 * a.nb2 * a.heads;
    dim3 grid((nbh * a.nqt + 7) / 8 * 8);          // whole groups of 8 (one block per XCD); surplus blocks exit at once
    const size_t lds = (size_t)ATTN_RING * 2 * KV_TILE * D * 2;
    if (D == 64 && a.prescaled && a.bound_dev) {
        // data-dependent bound: the static kernel serves every head whose bound is usable and flags the others, the
        // running-maximum kernel right behind it serves exactly those (its other workgroups exit at once)
        const int rc = bya_launch_attn_w4(&a, s);
        if (rc != BYA_OK) return rc;
        AttnArgs b = a;
        b.bound_dev = nullptr; b.score_bound = 0.f; b.only_flagged = a.fallback; b.fallback = nullptr;
        BYA_LAUNCH(attn_fwd_kernel_d64_prescaled, grid, dim3(256), lds, s, b);
        return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
    }
    if (D == 64 && a.prescaled && a.score_bound > 0.f) return bya_launch_attn_w4(&a, s);
    if (D == 64 && a.prescaled) BYA_LAUNCH(attn_fwd_kernel_d64_prescaled, grid, dim3(256), lds, s, a);
    else if (D == 64) BYA_LAUNCH(attn_fwd_kernel_d64, grid, dim3(256), lds, s, a);
    else BYA_LAUNCH(attn_fwd_kernel_d128, grid, dim3(256), lds, s, a);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

}  // namespace

// Which kernel a descriptor selects (reported to the host so tests and bench.py can say which softmax variant ran).
extern "C" int bya_attn_variant(const bya_attn_desc* d) {
    if (!d) return BYA_ERR_SHAPE;
    if (d->head_dim == 128) return d->scores_prescaled ? BYA_ERR_UNSUPPORTED : BYA_ATTN_D128;
    if (d->head_dim != 64) return BYA_ERR_UNSUPPORTED;
    if (!d->scores_prescaled) return BYA_ATTN_D64_RUNNING_MAX;
    if (d->bound_dev && d->fallback_flags) return BYA_ATTN_D64_DEVICE_BOUND_W4;
    // a usable static bound keeps every P = exp2(s) and every row sum a normal number: |s| <= 90; otherwise the
    // running-maximum kernel runs
    if (!(d->score_bound > 0.f && d->score_bound <= static_bound_limit())) return BYA_ATTN_D64_PRESCALED;
    return BYA_ATTN_D64_STATIC_BOUND_W4;
}

extern "C" int bya_attn_fwd(const void* q, const void* k, const void* v, void* o, const bya_attn_desc* d,
                            hipStream_t stream) {
    if (!q || !k || !v || !o || !d) return BYA_ERR_SHAPE;
    if (d->head_dim != 64 && d->head_dim != 128) return BYA_ERR_UNSUPPORTED;
    if (d->heads <= 0 || d->nb1 <= 0 || d->nb2 <= 0 || d->Sq <= 0 || d->Skv <= 0) return BYA_ERR_SHAPE;
    if ((d->q_row | d->k_row | d->v_row | d->q_s1 | d->q_s2 | d->k_s1 | d->k_s2 | d->v_s1 | d->v_s2) % 8) return BYA_ERR_ALIGN;
    if ((d->o_row | d->o_s1 | d->o_s2) % 4) return BYA_ERR_ALIGN;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) return BYA_ERR_ALIGN;
    if ((uintptr_t)o & 7) return BYA_ERR_ALIGN;
    AttnArgs a;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)o;
    a.heads = d->heads; a.nb1 = d->nb1; a.nb2 = d->nb2; a.Sq = d->Sq; a.Skv = d->Skv;
    a.nqt = (d->Sq + Q_PER_BLOCK - 1) / Q_PER_BLOCK;
    a.q_s1 = d->q_s1; a.q_s2 = d->q_s2; a.q_row = d->q_row;
    a.k_s1 = d->k_s1; a.k_s2 = d->k_s2; a.k_row = d->k_row;
    a.v_s1 = d->v_s1; a.v_s2 = d->v_s2; a.v_row = d->v_row;
    a.o_s1 = d->o_s1; a.o_s2 = d->o_s2; a.o_row = d->o_row;
    a.o_wide = !((uintptr_t)o & 15) && (d->o_s1 | d->o_s2 | d->o_row) % 8 == 0 && !bya_ref_form(BYA_REF_ATTN_NARROW_STORE);
    a.scale_log2 = d->scale * 1.4426950408889634f;
    a.prescaled = d->scores_prescaled;
    if (a.prescaled && d->head_dim != 64) return BYA_ERR_UNSUPPORTED;
    a.score_bound = (a.prescaled && d->score_bound > 0.f && d->score_bound <= static_bound_limit()) ? d->score_bound : 0.f;
    a.bound_dev = nullptr; a.bound_slots = 0; a.bound_heads = 0; a.bound_bh0 = 0; a.bound_limit = BYA_ATTN_BOUND_LIMIT;
    a.fallback = nullptr; a.only_flagged = nullptr;
    if (d->bound_dev && a.prescaled) {
        if (!d->fallback_flags || d->bound_slots < 1 || d->bound_slots > 64 || d->bound_bh0 < 0 ||
            d->bound_bh0 + d->nb1 * d->nb2 * d->heads > d->bound_heads) return BYA_ERR_SHAPE;
        if (((uintptr_t)d->bound_dev | (uintptr_t)d->fallback_flags) & 3) return BYA_ERR_ALIGN;
        a.bound_dev = d->bound_dev; a.bound_slots = d->bound_slots; a.bound_heads = d->bound_heads; a.bound_bh0 = d->bound_bh0;
        a.fallback = d->fallback_flags;
    }
    return d->head_dim == 64 ? launch_attn<64>(a, stream) : launch_attn<128>(a, stream);
}

extern "C" int bya_attn_kv_mix(const void* q, const void* k, const void* v, const void* r, const void* af, void* z, float* wsum,
                               const bya_attn_mix_desc* d, hipStream_t stream) {
    if (!q || !k || !v || !r || !z || !d) return BYA_ERR_SHAPE;
    if (d->head_dim != 64 && d->head_dim != 128) return BYA_ERR_UNSUPPORTED;
    if (d->heads <= 0 || d->n_id < 1 || d->n_id > 4 || d->n_grp <= 0 || d->Sq <= 0 || d->Skv <= 0 || d->Skv > KV_TILE) return BYA_ERR_SHAPE;
    if (af && d->n_id < 2) return BYA_ERR_SHAPE;      // the audio mix (af . r) needs >= 2 streams; routing_weights_of has no 1-stream audio case
    if ((d->q_grp | d->q_row | d->k_id | d->k_grp | d->k_row | d->v_id | d->v_grp | d->v_row) % 8) return BYA_ERR_ALIGN;
    if ((d->z_grp | d->z_row) % 4) return BYA_ERR_ALIGN;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) return BYA_ERR_ALIGN;
    if ((uintptr_t)z & 7) return BYA_ERR_ALIGN;
    MixArgs a;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.z = (bf16_t*)z;
    a.r = (const bf16_t*)r; a.af = (const bf16_t*)af; a.wsum = wsum;
    a.heads = d->heads; a.n_id = d->n_id; a.n_grp = d->n_grp; a.Sq = d->Sq; a.Skv = d->Skv;
    a.nqt = (d->Sq + Q_PER_BLOCK - 1) / Q_PER_BLOCK; a.mode = af ? 1 : 0;
    a.q_grp = d->q_grp; a.q_row = d->q_row; a.k_id = d->k_id; a.k_grp = d->k_grp; a.k_row = d->k_row;
    a.v_id = d->v_id; a.v_grp = d->v_grp; a.v_row = d->v_row; a.z_grp = d->z_grp; a.z_row = d->z_row;
    a.scale_log2 = d->scale * 1.4426950408889634f;
    if (d->Skv <= 32 && !bya_ref_form(BYA_REF_KV_MIX_GENERIC) && !((uintptr_t)z & 15) && (d->z_grp | d->z_row) % 8 == 0) {
        // row chunks per (group, head): about three 32-row tiles per wave, and at least ~4 workgroups per CU in total
        const int n32 = (d->Sq + 31) / 32;
        int nqc = (n32 + 11) / 12;
        a.nqt = nqc < 1 ? 1 : nqc;
        const dim3 grid32((unsigned)((long long)a.nqt * a.heads * a.n_grp));
        const size_t lds32 = (size_t)(a.n_id * 2 + 4) * 32 * d->head_dim * 2;    // K, V per identity + a z patch per wave
        if (lds32 > 64 * 1024) {                                 // four identities at head_dim 128: 96 KiB
            static std::atomic<unsigned long long> big64{0}, big128{0};
            const int rc = d->head_dim == 64
                ? bya_allow_big_lds(reinterpret_cast<const void*>(attn_kv_mix32_kernel_d64), 160 * 1024, big64)
                : bya_allow_big_lds(reinterpret_cast<const void*>(attn_kv_mix32_kernel_d128), 160 * 1024, big128);
            if (rc != BYA_OK) return rc;
        }
        if (d->head_dim == 64) BYA_LAUNCH(attn_kv_mix32_kernel_d64, grid32, dim3(256), lds32, stream, a);
        else BYA_LAUNCH(attn_kv_mix32_kernel_d128, grid32, dim3(256), lds32, stream, a);
        return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
    }
    const dim3 grid((unsigned)((long long)a.nqt * a.heads * a.n_grp));
    const size_t lds = (size_t)2 * 2 * KV_TILE * d->head_dim * 2;
    if (d->head_dim == 64) BYA_LAUNCH(attn_kv_mix_kernel_d64, grid, dim3(256), lds, stream, a);
    else BYA_LAUNCH(attn_kv_mix_kernel_d128, grid, dim3(256), lds, stream, a);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}
