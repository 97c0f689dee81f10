// Joint self-attention of the DiT blocks (17776 x 17776 tokens, 48 heads of 64; replaces F.scaled_dot_product_attention
// inside diffusers' CogVideoXAttnProcessor2_0, models/transformer.py:200-209,241-245) as ONE WAVE PER SIMD with a
// hand-placed instruction stream -- the form DESIGN.md section 9.1 named as the next step after the two-block kernel.
//
// Same arithmetic as attn_fwd_kernel_d64_bounded2 (attn.hip): scores arrive in exp2 units (scale * log2 e folded into k
// by bya_qknorm_rope) and are bounded by the caller (|s| <= B <= 48), so softmax needs no running maximum: P = exp2(s),
// row sums in fp32, P rounded to bf16 for the P.V product, O / l at the end.  Swapped QK^T (S^T = K.Q^T, a query row
// lives on one lane pair), the S accumulator IS the B operand of O^T += V^T.P^T, V by ds_read_b64_tr_b16.
//
// What is different is who schedules it.  A workgroup = 4 waves = 512 query rows of one (batch, head), one workgroup
// per CU (512 registers per lane): a wave owns FOUR 32-row query blocks, every K and V fragment it reads from LDS feeds
// all four (a quarter of the LDS reads per FLOP of the one-block kernel, half the K/V staging per FLOP of the two-block
// one).  The 64 MFMAs of a 64-key tile run as four periods { QK_b ; PV_(b-1) } with the softmax of block b in the MFMA
// gaps behind QK_b (tools/gen_attn_w4_schedule.py holds the placement table and describes it) -- every instruction of
// the hot loop is a volatile asm statement in program order, hipcc only allocates registers.  At head_dim 64 the loop is
// bound by the vector issue port (2 v_exp + 2 v_add + 1 v_cvt_pk per MFMA gap = 28 cycles + the MFMA's own 8 against the
// matrix pipe's 32): 36.5 cycles per MFMA is the floor of ANY d = 64 softmax on this ISA; the point of placing the
// stream by hand is that nothing else -- LDS latency, the tile rendezvous, DMA issue -- is exposed on top of it.
//
// K/V tiles: LDS-DMA (buffer_load ... lds) three tiles ahead into a 3-stage ring (48 KiB), one s_barrier per tile at
// the head of PV_2: by then every wave has read tile t's V (under QK_1) and tile t+1 must have landed.
// Rows past Skv in the last tile: their K and V rows land as zeros (hardware range check); the last tile of a piece runs the
// 'M' copy of the stream, which sets their scores to -inf between its statements (P = 0: they are in no row sum).  (Until
// round 5 they stayed at score 0 -> P = 1 and their count was subtracted from the row sum at the end: exact arithmetic, but a
// row whose real scores all lay below ~-10 lost its sum in that subtraction -- a constant negative offset of a head, e.g.
// from the q/k-LayerNorm biases, is within the +-90 the softmax promises to handle.)
// Built WITHOUT -amdgpu-mfma-vgpr-form (O accumulators and the Q fragments live in AGPRs).
#include "attn_common.h"
#include <stdlib.h>
#include "options.h"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
template <char C> struct IntTagC { static constexpr char value = C; };

__device__ __forceinline__ i32x4 raw_rsrc(const void* base, uint32_t bytes) {
    const unsigned long long b = (unsigned long long)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffu));
    r.y = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

// one 1-KiB LDS-DMA piece: 64 lanes x 16 bytes from per-lane global offsets to LDS [m0 .. m0 + 1024)
__device__ __forceinline__ void dma_piece(uint32_t lds_dst, uint32_t voff, const i32x4& rsrc, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

constexpr int QB = 4;                                   // 32-row query blocks per wave
constexpr int ROWS_PER_WG = 4 * QB * 32;                // 512
constexpr int RB = 128, TILE_BYTES = KV_TILE * RB, STAGE_BYTES = 2 * TILE_BYTES, NST = 3;

// Stream-K form (SK = true, bya_launch_attn_w4 when a workspace is registered): 1680 (head, q-tile) items on 256 CUs are
// 6.56 rounds that cost 7.  Here the grid is 256 persistent workgroups, 32 per XCD, and an XCD owns whole heads as before.
//   * full rounds: workgroup i of the XCD takes items i, i + 32, i + 64, ... of the XCD's (head, q-tile) list -- at any
//     moment the 32 workgroups sit on 32 consecutive q-tiles of (mostly) one head and stream the same K/V through the
//     XCD's L2, exactly like the dispatcher's order of the one-workgroup-per-item launch (contiguous ranges per
//     workgroup puts every CU at its own head position).
//   * the remaining rem < 32 items are cut along the keys AT THE SAME KEY TILE a = nt * rem / 32: workgroups 0 .. rem-1 of the
//     XCD ("mains") take key tiles [0, a) of one item each -- in step, like a full round -- and the other 32 - rem
//     ("helpers") share the suffixes [a, nt) of those items evenly, item after item (the suffix K/V of a head is < 1 MB and
//     stays in the L2 while a helper walks the q-tiles of that head).  Cutting the (item, key tile) list into 32
//     contiguous ranges instead puts every workgroup at a different key position; both forms measure the same (+1 % at
//     48 heads, +9 % at 24), so the cleaner one stays.  With the static-bound softmax partial results are additive (no running maximum toum to
//     reconcile): a helper writes the un-normalised O^T accumulators and row sums of each of its pieces to that piece's
//     workspace slot (pieces are numbered in step order: helper index + item index, < 64 per XCD) and raises the slot's
//     flag; the main adds the slots of its item to its registers, normalises and stores.  Helpers never wait, so there
//     is no circular wait.  A grid that does not fill ONE round (a rank's 6 heads of an 8-GPU step: 210 items on 256 CUs:
//     every item a leftover item) runs correctly this way too but 7 % SLOWER than one workgroup per item, so the launcher
//     declines it (see bya_launch_attn_w4).
struct SkItem { int bh, qt, tb, nt, nt_all, role, local, slot; };   // role: 0 whole item, 1 suffix piece (writer), 2 prefix (merger)

constexpr int SK_SLOT_FLOATS = 4 * QB * 2 * 16 * 64 + 4 * QB * 64;        // O^T register image + per-lane row sums
constexpr int SK_FLAG_BYTES = 4096;                                       // 1024 words: [slot] flags, [1023] time-outs

template <bool SK>
__global__ __launch_bounds__(256, 1) void attn_joint_w4_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hf = lane >> 5;

    const int nbh = p.nb1 * p.nb2 * p.heads;
    const int nt_all = (p.Skv + KV_TILE - 1) / KV_TILE;
    // ---- the work of this workgroup
    const int sk_xcd = blockIdx.x & 7, sk_idx = blockIdx.x >> 3, sk_ncu = gridDim.x >> 3;
    int sk_round = 0, sk_rfull = 0, sk_rem = 0, sk_a = 0, sk_suf = 0, sk_s = 0, sk_end = 0;   // SK: rounds, leftover split, helper steps
    bool sk_main_done = false;
    SkItem it;
    int sk_base = 0;                                               // nbh % 8 != 0: first item of this XCD in (head, q-tile) order
    if (SK) {
        // items of this XCD: whole (batch, head)s when their count divides by 8, else a contiguous eighth of the
        // (head, q-tile) order -- the same two rules as the one-workgroup-per-item launch below
        int ipx = (nbh >> 3) * p.nqt;
        if (nbh % 8 != 0) {
            const int total = nbh * p.nqt, cq = total >> 3, cr = total & 7;
            sk_base = sk_xcd < cr ? sk_xcd * (cq + 1) : cr * (cq + 1) + (sk_xcd - cr) * cq;
            ipx = cq + (sk_xcd < cr ? 1 : 0);
        }
        sk_rfull = ipx / sk_ncu;
        sk_rem = ipx - sk_rfull * sk_ncu;                          // leftover items: mains 0 .. rem-1, helpers rem .. ncu-1
        sk_a = sk_rem ? (int)(((long long)nt_all * sk_rem + sk_ncu - 1) / sk_ncu) : nt_all;      // mains' key tiles [0, a)
        sk_suf = nt_all - sk_a;                                    // suffix tiles per leftover item
        if (sk_idx >= sk_rem && sk_rem && sk_suf) {                // helper: its share of the rem * suf suffix steps
            const long long S = (long long)sk_rem * sk_suf;
            const int nh = sk_ncu - sk_rem, h = sk_idx - sk_rem;
            sk_s = (int)(S * h / nh);
            sk_end = (int)(S * (h + 1) / nh);
        }
    } else {
        // block -> (bh, q-tile): blocks with equal (blockIdx % 8) share an XCD; whole (batch, head)s per XCD when the count
        // divides by 8, else a contiguous eighth of the (head, q-tile) order (same rule as attn.hip)
        if (nbh % 8 == 0) {
            const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
            it.bh = (j / p.nqt) * 8 + xcd;
            it.qt = j % p.nqt;
        } else {
            const int total = nbh * p.nqt, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
            const int cq = total >> 3, cr = total & 7;
            const int base = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
            if (j >= cq + (xcd < cr ? 1 : 0)) return;
            it.bh = (base + j) / p.nqt;
            it.qt = (base + j) % p.nqt;
        }
        if (it.bh >= nbh) return;
        it.tb = 0; it.nt = nt_all; it.nt_all = nt_all; it.role = 0; it.local = 0;
    }

    // ---- K/V staging: wave w moves rows [16 w, 16 w + 16) of the K tile and of the V tile, two 1-KiB pieces each
    i32x4 rsK, rsV;
    uint32_t dvo[4];                                       // per-lane source offsets of pieces K0 K1 V0 V1
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = wave * 16 + q * 8 + lane / 8, slot = lane % 8;
        dvo[q] = (uint32_t)row * (uint32_t)(p.k_row * 2) + ((slot ^ kswz<D>(row)) << 4);
        dvo[2 + q] = (uint32_t)row * (uint32_t)(p.v_row * 2) + ((slot ^ vswz<D>(row)) << 4);
    }
    const uint32_t k_tile_stride = KV_TILE * (uint32_t)p.k_row * 2, v_tile_stride = KV_TILE * (uint32_t)p.v_row * 2;
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
    const uint32_t lds0s = __builtin_amdgcn_readfirstlane(lds0) + wave * 16 * RB;       // this wave's first K piece, stage 0
    auto stage_tile = [&](int t, int st) {                 // tiles past the end are outside the descriptors: zeros land
        const uint32_t dst = lds0s + st * STAGE_BYTES;
        dma_piece(dst, dvo[0], rsK, t * k_tile_stride);
        dma_piece(dst + 1024, dvo[1], rsK, t * k_tile_stride);
        dma_piece(dst + TILE_BYTES, dvo[2], rsV, t * v_tile_stride);
        dma_piece(dst + TILE_BYTES + 1024, dvo[3], rsV, t * v_tile_stride);
    };

    // ---- state.  Q fragments (B operand of S^T = K.Q^T): lane (r, hf) holds Q[q0 + 32 b + r][16 s + 8 hf .. + 7].  S and P
    // are double-buffered by block parity; the loop's first period finishes "block 3 of tile -1": S[1][1] = -inf
    // (exp2 -> 0), P[1] = 0 and V = 0 make that a no-op.  (Re-)initialised per item below.
    bf16x8 qf[QB][4];
    bool q_valid[QB];
    f32x16 oacc[QB][2], sacc[2][2];
    u32x4 pf[2][4];
    u32x2 vh[2][4][2];
    bf16x8 kf[2][4];
    float psum[QB][2];
    int keys_last = 64;                                    // keys that exist in the tile the 'M' stream runs on

    // per-lane LDS offsets inside a stage: K fragment (u, s) at kofs[s] + u * 32 rows; V fragment (d, ks, h) at
    // vofs[d] + (16 ks + 8 h) rows (transposed read: 4 rows x 64 B per half-wave, attn.hip)
    uint32_t kofs[4], vofs[2];
    {
        const int sw = kswz<D>(r);
#pragma unroll
        for (int s = 0; s < 4; ++s) kofs[s] = lds0 + r * RB + (((2 * s + hf) ^ sw) << 4);
        const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int row = 4 * hf + tq;
            const int chunk = 4 * d + 2 * (g & 1) + (tp >> 1);
            vofs[d] = lds0 + TILE_BYTES + row * RB + ((chunk ^ vswz<D>(row)) << 4) + (tp & 1) * 8;
        }
    }

#define LGKM(N) asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory")
#define VMC(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define BAR() asm volatile("s_barrier" ::: "memory")
    // K fragments of the tile in stage `kst`; V fragments of the tile in stage `vst`
#define RK(U, S) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[U][S]) : "v"(kaddr[S]), "i"((U) * 32 * RB))
    // scores of keys that do not exist (this lane's key of register v of a 32-key half: (v & 3) + 8 (v >> 2) + 4 hf) -> -inf
#define MASKPAD(ACC, BASE)                                                                     \
    if ((BASE) + 32 > keys_last)                         /* (wave-uniform: this half holds such keys at all) */ \
        _Pragma("unroll") for (int v_ = 0; v_ < 16; ++v_)                                      \
            if ((BASE) + (v_ & 3) + 8 * (v_ >> 2) + 4 * hf >= keys_last) ACC[v_] = -INFINITY
    // One tile ('L'), the last tile of the piece ('M': 'L' + MASKPAD) or the tail behind the last tile ('T').  kaddr: K fragment addresses of tile t + 1, vaddr: V
    // fragment addresses of tile t, dma_dst / soffK / soffV: where tile t + 3 goes and comes from.
    auto body = [&](auto v_c, const uint32_t (&kaddr)[4], const uint32_t (&vaddr)[2], uint32_t dma_dst, uint32_t soffK,
                    uint32_t soffV) {
        constexpr char VAR = decltype(v_c)::value;
        // GENERATED-BEGIN (tools/gen_attn_w4_schedule.py)
        if constexpr (VAR == 'L') {
            // period 0: QK_0 | PV_3
            { float t1_0; float t1_1; float t1_2; float t1_3; float t1_4; float t1_5; float t1_6; float t1_7; float t1_8; float t1_9; float t1_10; float t1_11; float t1_12; float t1_13; float t1_14; float t1_15; uint32_t w2_0; uint32_t w2_1; uint32_t w2_2; uint32_t w2_3; uint32_t w3_0; uint32_t w3_1; uint32_t w3_2; uint32_t w3_3; asm volatile("v_mfma_f32_32x32x16_bf16 %0, %28, %29, 0\n\tv_exp_f32 %1, %30\n\tv_exp_f32 %2, %31\n\tv_mfma_f32_32x32x16_bf16 %0, %32, %33, %0\n\tv_exp_f32 %3, %34\n\tv_add_f32 %5, %5, %1\n\tv_exp_f32 %4, %35\n\tv_add_f32 %6, %6, %2\n\tv_cvt_pk_bf16_f32 %7, %1, %2\n\tv_mfma_f32_32x32x16_bf16 %0, %36, %37, %0\n\tv_exp_f32 %8, %38\n\tv_add_f32 %5, %5, %3\n\tv_exp_f32 %9, %39\n\tv_add_f32 %6, %6, %4\n\tv_cvt_pk_bf16_f32 %10, %3, %4\n\tv_mfma_f32_32x32x16_bf16 %0, %40, %41, %0\n\tv_exp_f32 %11, %42\n\tv_add_f32 %5, %5, %8\n\tv_exp_f32 %12, %43\n\tv_add_f32 %6, %6, %9\n\tv_cvt_pk_bf16_f32 %13, %8, %9\n\tv_mfma_f32_32x32x16_bf16 %14, %44, %29, 0\n\tv_exp_f32 %15, %45\n\tv_add_f32 %5, %5, %11\n\tv_exp_f32 %16, %46\n\tv_add_f32 %6, %6, %12\n\tv_cvt_pk_bf16_f32 %17, %11, %12\n\tv_mfma_f32_32x32x16_bf16 %14, %47, %33, %14\n\tv_exp_f32 %18, %48\n\tv_add_f32 %5, %5, %15\n\tv_exp_f32 %19, %49\n\tv_add_f32 %6, %6, %16\n\tv_cvt_pk_bf16_f32 %20, %15, %16\n\tv_mfma_f32_32x32x16_bf16 %14, %50, %37, %14\n\tv_exp_f32 %21, %51\n\tv_add_f32 %5, %5, %18\n\tv_exp_f32 %22, %52\n\tv_add_f32 %6, %6, %19\n\tv_cvt_pk_bf16_f32 %23, %18, %19\n\tv_mfma_f32_32x32x16_bf16 %14, %53, %41, %14\n\tv_exp_f32 %24, %54\n\tv_add_f32 %5, %5, %21\n\tv_exp_f32 %25, %55\n\tv_add_f32 %6, %6, %22\n\tv_cvt_pk_bf16_f32 %26, %21, %22\n\tv_add_f32 %5, %5, %24\n\tv_add_f32 %6, %6, %25\n\tv_cvt_pk_bf16_f32 %27, %24, %25" : "=&v"(sacc[0][0]), "=&v"(t1_0), "=&v"(t1_1), "=&v"(t1_2), "=&v"(t1_3), "+v"(psum[3][0]), "+v"(psum[3][1]), "=&v"(w2_0), "=&v"(t1_4), "=&v"(t1_5), "=&v"(w2_1), "=&v"(t1_6), "=&v"(t1_7), "=&v"(w2_2), "=&v"(sacc[0][1]), "=&v"(t1_8), "=&v"(t1_9), "=&v"(w2_3), "=&v"(t1_10), "=&v"(t1_11), "=&v"(w3_0), "=&v"(t1_12), "=&v"(t1_13), "=&v"(w3_1), "=&v"(t1_14), "=&v"(t1_15), "=&v"(w3_2), "=&v"(w3_3) : "v"(kf[0][0]), "a"(qf[0][0]), "v"(sacc[1][1][0]), "v"(sacc[1][1][1]), "v"(kf[0][1]), "a"(qf[0][1]), "v"(sacc[1][1][2]), "v"(sacc[1][1][3]), "v"(kf[0][2]), "a"(qf[0][2]), "v"(sacc[1][1][4]), "v"(sacc[1][1][5]), "v"(kf[0][3]), "a"(qf[0][3]), "v"(sacc[1][1][6]), "v"(sacc[1][1][7]), "v"(kf[1][0]), "v"(sacc[1][1][8]), "v"(sacc[1][1][9]), "v"(kf[1][1]), "v"(sacc[1][1][10]), "v"(sacc[1][1][11]), "v"(kf[1][2]), "v"(sacc[1][1][12]), "v"(sacc[1][1][13]), "v"(kf[1][3]), "v"(sacc[1][1][14]), "v"(sacc[1][1][15]) : "memory"); pf[1][2][0] = w2_0; pf[1][2][1] = w2_1; pf[1][2][2] = w2_2; pf[1][2][3] = w2_3; pf[1][3][0] = w3_0; pf[1][3][1] = w3_1; pf[1][3][2] = w3_2; pf[1][3][3] = w3_3; }
            { const u32x4 vv0_0 = {vh[0][0][0][0], vh[0][0][0][1], vh[0][0][1][0], vh[0][0][1][1]}; const u32x4 vv1_0 = {vh[1][0][0][0], vh[1][0][0][1], vh[1][0][1][0], vh[1][0][1][1]}; const u32x4 vv0_1 = {vh[0][1][0][0], vh[0][1][0][1], vh[0][1][1][0], vh[0][1][1][1]}; const u32x4 vv1_1 = {vh[1][1][0][0], vh[1][1][0][1], vh[1][1][1][0], vh[1][1][1][1]}; const u32x4 vv0_2 = {vh[0][2][0][0], vh[0][2][0][1], vh[0][2][1][0], vh[0][2][1][1]}; const u32x4 vv1_2 = {vh[1][2][0][0], vh[1][2][0][1], vh[1][2][1][0], vh[1][2][1][1]}; const u32x4 vv0_3 = {vh[0][3][0][0], vh[0][3][0][1], vh[0][3][1][0], vh[0][3][1][1]}; const u32x4 vv1_3 = {vh[1][3][0][0], vh[1][3][0][1], vh[1][3][1][0], vh[1][3][1][1]}; float t0_0; float t0_1; float t0_2; float t0_3; float t0_4; float t0_5; float t0_6; float t0_7; float t0_8; float t0_9; float t0_10; float t0_11; float t0_12; float t0_13; float t0_14; float t0_15; uint32_t w0_0; uint32_t w0_1; uint32_t w0_2; uint32_t w0_3; uint32_t w1_0; uint32_t w1_1; uint32_t w1_2; uint32_t w1_3; asm volatile("v_mfma_f32_32x32x16_bf16 %0, %28, %29, %0\n\tv_exp_f32 %1, %30\n\tv_exp_f32 %2, %31\n\tv_mfma_f32_32x32x16_bf16 %3, %32, %29, %3\n\tv_exp_f32 %4, %33\n\tv_add_f32 %6, %6, %1\n\tv_exp_f32 %5, %34\n\tv_add_f32 %7, %7, %2\n\tv_cvt_pk_bf16_f32 %8, %1, %2\n\tv_mfma_f32_32x32x16_bf16 %0, %35, %36, %0\n\tv_exp_f32 %9, %37\n\tv_add_f32 %6, %6, %4\n\tv_exp_f32 %10, %38\n\tv_add_f32 %7, %7, %5\n\tv_cvt_pk_bf16_f32 %11, %4, %5\n\tv_mfma_f32_32x32x16_bf16 %3, %39, %36, %3\n\tv_exp_f32 %12, %40\n\tv_add_f32 %6, %6, %9\n\tv_exp_f32 %13, %41\n\tv_add_f32 %7, %7, %10\n\tv_cvt_pk_bf16_f32 %14, %9, %10\n\tv_mfma_f32_32x32x16_bf16 %0, %42, %43, %0\n\tv_exp_f32 %15, %44\n\tv_add_f32 %6, %6, %12\n\tv_exp_f32 %16, %45\n\tv_add_f32 %7, %7, %13\n\tv_cvt_pk_bf16_f32 %17, %12, %13\n\tv_mfma_f32_32x32x16_bf16 %3, %46, %43, %3\n\tv_exp_f32 %18, %47\n\tv_add_f32 %6, %6, %15\n\tv_exp_f32 %19, %48\n\tv_add_f32 %7, %7, %16\n\tv_cvt_pk_bf16_f32 %20, %15, %16\n\tv_mfma_f32_32x32x16_bf16 %0, %49, %50, %0\n\tv_exp_f32 %21, %51\n\tv_add_f32 %6, %6, %18\n\tv_exp_f32 %22, %52\n\tv_add_f32 %7, %7, %19\n\tv_cvt_pk_bf16_f32 %23, %18, %19\n\tv_mfma_f32_32x32x16_bf16 %3, %53, %50, %3\n\tv_exp_f32 %24, %54\n\tv_add_f32 %6, %6, %21\n\tv_exp_f32 %25, %55\n\tv_add_f32 %7, %7, %22\n\tv_cvt_pk_bf16_f32 %26, %21, %22\n\tv_add_f32 %6, %6, %24\n\tv_add_f32 %7, %7, %25\n\tv_cvt_pk_bf16_f32 %27, %24, %25" : "+a"(oacc[3][0]), "=&v"(t0_0), "=&v"(t0_1), "+a"(oacc[3][1]), "=&v"(t0_2), "=&v"(t0_3), "+v"(psum[0][0]), "+v"(psum[0][1]), "=&v"(w0_0), "=&v"(t0_4), "=&v"(t0_5), "=&v"(w0_1), "=&v"(t0_6), "=&v"(t0_7), "=&v"(w0_2), "=&v"(t0_8), "=&v"(t0_9), "=&v"(w0_3), "=&v"(t0_10), "=&v"(t0_11), "=&v"(w1_0), "=&v"(t0_12), "=&v"(t0_13), "=&v"(w1_1), "=&v"(t0_14), "=&v"(t0_15), "=&v"(w1_2), "=&v"(w1_3) : "v"(vv0_0), "v"(pf[1][0]), "v"(sacc[0][0][0]), "v"(sacc[0][0][1]), "v"(vv1_0), "v"(sacc[0][0][2]), "v"(sacc[0][0][3]), "v"(vv0_1), "v"(pf[1][1]), "v"(sacc[0][0][4]), "v"(sacc[0][0][5]), "v"(vv1_1), "v"(sacc[0][0][6]), "v"(sacc[0][0][7]), "v"(vv0_2), "v"(pf[1][2]), "v"(sacc[0][0][8]), "v"(sacc[0][0][9]), "v"(vv1_2), "v"(sacc[0][0][10]), "v"(sacc[0][0][11]), "v"(vv0_3), "v"(pf[1][3]), "v"(sacc[0][0][12]), "v"(sacc[0][0][13]), "v"(vv1_3), "v"(sacc[0][0][14]), "v"(sacc[0][0][15]) : "memory"); pf[0][0][0] = w0_0; pf[0][0][1] = w0_1; pf[0][0][2] = w0_2; pf[0][0][3] = w0_3; pf[0][1][0] = w1_0; pf[0][1][1] = w1_1; pf[0][1][2] = w1_2; pf[0][1][3] = w1_3; }
            // period 1: QK_1 | PV_0
            { float t1_0; float t1_1; float t1_2; float t1_3; float t1_4; float t1_5; float t1_6; float t1_7; float t1_8; float t1_9; float t1_10; float t1_11; float t1_12; float t1_13; float t1_14; float t1_15; uint32_t w2_0; uint32_t w2_1; uint32_t w2_2; uint32_t w2_3; uint32_t w3_0; uint32_t w3_1; uint32_t w3_2; uint32_t w3_3; asm volatile("v_mfma_f32_32x32x16_bf16 %0, %44, %45, 0\n\tds_read_b64_tr_b16 %1, %46 offset:0\n\tds_read_b64_tr_b16 %2, %46 offset:1024\n\tds_read_b64_tr_b16 %3, %47 offset:0\n\tv_exp_f32 %4, %48\n\tv_exp_f32 %5, %49\n\tv_mfma_f32_32x32x16_bf16 %0, %50, %51, %0\n\tds_read_b64_tr_b16 %6, %47 offset:1024\n\tds_read_b64_tr_b16 %7, %46 offset:2048\n\tds_read_b64_tr_b16 %8, %46 offset:3072\n\tv_exp_f32 %9, %52\n\tv_add_f32 %11, %11, %4\n\tv_exp_f32 %10, %53\n\tv_add_f32 %12, %12, %5\n\tv_cvt_pk_bf16_f32 %13, %4, %5\n\tv_mfma_f32_32x32x16_bf16 %0, %54, %55, %0\n\tds_read_b64_tr_b16 %14, %47 offset:2048\n\tds_read_b64_tr_b16 %15, %47 offset:3072\n\tds_read_b64_tr_b16 %16, %46 offset:4096\n\tv_exp_f32 %17, %56\n\tv_add_f32 %11, %11, %9\n\tv_exp_f32 %18, %57\n\tv_add_f32 %12, %12, %10\n\tv_cvt_pk_bf16_f32 %19, %9, %10\n\tv_mfma_f32_32x32x16_bf16 %0, %58, %59, %0\n\tds_read_b64_tr_b16 %20, %46 offset:5120\n\tds_read_b64_tr_b16 %21, %47 offset:4096\n\tds_read_b64_tr_b16 %22, %47 offset:5120\n\tv_exp_f32 %23, %60\n\tv_add_f32 %11, %11, %17\n\tv_exp_f32 %24, %61\n\tv_add_f32 %12, %12, %18\n\tv_cvt_pk_bf16_f32 %25, %17, %18\n\tv_mfma_f32_32x32x16_bf16 %26, %62, %45, 0\n\tds_read_b64_tr_b16 %27, %46 offset:6144\n\tds_read_b64_tr_b16 %28, %46 offset:7168\n\tv_exp_f32 %29, %63\n\tv_add_f32 %11, %11, %23\n\tv_exp_f32 %30, %64\n\tv_add_f32 %12, %12, %24\n\tv_cvt_pk_bf16_f32 %31, %23, %24\n\tv_mfma_f32_32x32x16_bf16 %26, %65, %51, %26\n\tds_read_b64_tr_b16 %32, %47 offset:6144\n\tds_read_b64_tr_b16 %33, %47 offset:7168\n\tv_exp_f32 %34, %66\n\tv_add_f32 %11, %11, %29\n\tv_exp_f32 %35, %67\n\tv_add_f32 %12, %12, %30\n\tv_cvt_pk_bf16_f32 %36, %29, %30\n\tv_mfma_f32_32x32x16_bf16 %26, %68, %55, %26\n\tv_exp_f32 %37, %69\n\tv_add_f32 %11, %11, %34\n\tv_exp_f32 %38, %70\n\tv_add_f32 %12, %12, %35\n\tv_cvt_pk_bf16_f32 %39, %34, %35\n\tv_mfma_f32_32x32x16_bf16 %26, %71, %59, %26\n\tv_exp_f32 %40, %72\n\tv_add_f32 %11, %11, %37\n\tv_exp_f32 %41, %73\n\tv_add_f32 %12, %12, %38\n\tv_cvt_pk_bf16_f32 %42, %37, %38\n\tv_add_f32 %11, %11, %40\n\tv_add_f32 %12, %12, %41\n\tv_cvt_pk_bf16_f32 %43, %40, %41\n\ts_waitcnt lgkmcnt(0)" : "=&v"(sacc[1][0]), "=&v"(vh[0][0][0]), "=&v"(vh[0][0][1]), "=&v"(vh[1][0][0]), "=&v"(t1_0), "=&v"(t1_1), "=&v"(vh[1][0][1]), "=&v"(vh[0][1][0]), "=&v"(vh[0][1][1]), "=&v"(t1_2), "=&v"(t1_3), "+v"(psum[0][0]), "+v"(psum[0][1]), "=&v"(w2_0), "=&v"(vh[1][1][0]), "=&v"(vh[1][1][1]), "=&v"(vh[0][2][0]), "=&v"(t1_4), "=&v"(t1_5), "=&v"(w2_1), "=&v"(vh[0][2][1]), "=&v"(vh[1][2][0]), "=&v"(vh[1][2][1]), "=&v"(t1_6), "=&v"(t1_7), "=&v"(w2_2), "=&v"(sacc[1][1]), "=&v"(vh[0][3][0]), "=&v"(vh[0][3][1]), "=&v"(t1_8), "=&v"(t1_9), "=&v"(w2_3), "=&v"(vh[1][3][0]), "=&v"(vh[1][3][1]), "=&v"(t1_10), "=&v"(t1_11), "=&v"(w3_0), "=&v"(t1_12), "=&v"(t1_13), "=&v"(w3_1), "=&v"(t1_14), "=&v"(t1_15), "=&v"(w3_2), "=&v"(w3_3) : "v"(kf[0][0]), "a"(qf[1][0]), "v"(vaddr[0]), "v"(vaddr[1]), "v"(sacc[0][1][0]), "v"(sacc[0][1][1]), "v"(kf[0][1]), "a"(qf[1][1]), "v"(sacc[0][1][2]), "v"(sacc[0][1][3]), "v"(kf[0][2]), "a"(qf[1][2]), "v"(sacc[0][1][4]), "v"(sacc[0][1][5]), "v"(kf[0][3]), "a"(qf[1][3]), "v"(sacc[0][1][6]), "v"(sacc[0][1][7]), "v"(kf[1][0]), "v"(sacc[0][1][8]), "v"(sacc[0][1][9]), "v"(kf[1][1]), "v"(sacc[0][1][10]), "v"(sacc[0][1][11]), "v"(kf[1][2]), "v"(sacc[0][1][12]), "v"(sacc[0][1][13]), "v"(kf[1][3]), "v"(sacc[0][1][14]), "v"(sacc[0][1][15]) : "memory"); pf[0][2][0] = w2_0; pf[0][2][1] = w2_1; pf[0][2][2] = w2_2; pf[0][2][3] = w2_3; pf[0][3][0] = w3_0; pf[0][3][1] = w3_1; pf[0][3][2] = w3_2; pf[0][3][3] = w3_3; }
            { const u32x4 vv0_0 = {vh[0][0][0][0], vh[0][0][0][1], vh[0][0][1][0], vh[0][0][1][1]}; const u32x4 vv1_0 = {vh[1][0][0][0], vh[1][0][0][1], vh[1][0][1][0], vh[1][0][1][1]}; const u32x4 vv0_1 = {vh[0][1][0][0], vh[0][1][0][1], vh[0][1][1][0], vh[0][1][1][1]}; const u32x4 vv1_1 = {vh[1][1][0][0], vh[1][1][0][1], vh[1][1][1][0], vh[1][1][1][1]}; const u32x4 vv0_2 = {vh[0][2][0][0], vh[0][2][0][1], vh[0][2][1][0], vh[0][2][1][1]}; const u32x4 vv1_2 = {vh[1][2][0][0], vh[1][2][0][1], vh[1][2][1][0], vh[1][2][1][1]}; const u32x4 vv0_3 = {vh[0][3][0][0], vh[0][3][0][1], vh[0][3][1][0], vh[0][3][1][1]}; const u32x4 vv1_3 = {vh[1][3][0][0], vh[1][3][0][1], vh[1][3][1][0], vh[1][3][1][1]}; float t0_0; float t0_1; float t0_2; float t0_3; float t0_4; float t0_5; float t0_6; float t0_7; float t0_8; float t0_9; float t0_10; float t0_11; float t0_12; float t0_13; float t0_14; float t0_15; uint32_t w0_0; uint32_t w0_1; uint32_t w0_2; uint32_t w0_3; uint32_t w1_0; uint32_t w1_1; uint32_t w1_2; uint32_t w1_3; asm volatile("v_mfma_f32_32x32x16_bf16 %0, %28, %29, %0\n\tv_exp_f32 %1, %30\n\tv_exp_f32 %2, %31\n\tv_mfma_f32_32x32x16_bf16 %3, %32, %29, %3\n\tv_exp_f32 %4, %33\n\tv_add_f32 %6, %6, %1\n\tv_exp_f32 %5, %34\n\tv_add_f32 %7, %7, %2\n\tv_cvt_pk_bf16_f32 %8, %1, %2\n\tv_mfma_f32_32x32x16_bf16 %0, %35, %36, %0\n\tv_exp_f32 %9, %37\n\tv_add_f32 %6, %6, %4\n\tv_exp_f32 %10, %38\n\tv_add_f32 %7, %7, %5\n\tv_cvt_pk_bf16_f32 %11, %4, %5\n\tv_mfma_f32_32x32x16_bf16 %3, %39, %36, %3\n\tv_exp_f32 %12, %40\n\tv_add_f32 %6, %6, %9\n\tv_exp_f32 %13, %41\n\tv_add_f32 %7, %7, %10\n\tv_cvt_pk_bf16_f32 %14, %9, %10\n\tv_mfma_f32_32x32x16_bf16 %0, %42, %43, %0\n\tv_exp_f32 %15, %44\n\tv_add_f32 %6, %6, %12\n\tv_exp_f32 %16, %45\n\tv_add_f32 %7, %7, %13\n\tv_cvt_pk_bf16_f32 %17, %12, %13\n\tv_mfma_f32_32x32x16_bf16 %3, %46, %43, %3\n\tv_exp_f32 %18, %47\n\tv_add_f32 %6, %6, %15\n\tv_exp_f32 %19, %48\n\tv_add_f32 %7, %7, %16\n\tv_cvt_pk_bf16_f32 %20, %15, %16\n\tv_mfma_f32_32x32x16_bf16 %0, %49, %50, %0\n\tv_exp_f32 %21, %51\n\tv_add_f32 %6, %6, %18\n\tv_exp_f32 %22, %52\n\tv_add_f32 %7, %7, %19\n\tv_cvt_pk_bf16_f32 %23, %18, %19\n\tv_mfma_f32_32x32x16_bf16 %3, %53, %50, %3\n\tv_exp_f32 %24, %54\n\tv_add_f32 %6, %6, %21\n\tv_exp_f32 %25, %55\n\tv_add_f32 %7, %7, %22\n\tv_cvt_pk_bf16_f32 %26, %21, %22\n\tv_add_f32 %6, %6, %24\n\tv_add_f32 %7, %7, %25\n\tv_cvt_pk_bf16_f32 %27, %24, %25" : "+a"(oacc[0][0]), "=&v"(t0_0), "=&v"(t0_1), "+a"(oacc[0][1]), "=&v"(t0_2), "=&v"(t0_3), "+v"(psum[1][0]), "+v"(psum[1][1]), "=&v"(w0_0), "=&v"(t0_4), "=&v"(t0_5), "=&v"(w0_1), "=&v"(t0_6), "=&v"(t0_7), "=&v"(w0_2), "=&v"(t0_8), "=&v"(t0_9), "=&v"(w0_3), "=&v"(t0_10), "=&v"(t0_11), "=&v"(w1_0), "=&v"(t0_12), "=&v"(t0_13), "=&v"(w1_1), "=&v"(t0_14), "=&v"(t0_15), "=&v"(w1_2), "=&v"(w1_3) : "v"(vv0_0), "v"(pf[0][0]), "v"(sacc[1][0][0]), "v"(sacc[1][0][1]), "v"(vv1_0), "v"(sacc[1][0][2]), "v"(sacc[1][0][3]), "v"(vv0_1), "v"(pf[0][1]), "v"(sacc[1][0][4]), "v"(sacc[1][0][5]), "v"(vv1_1), "v"(sacc[1][0][6]), "v"(sacc[1][0][7]), "v"(vv0_2), "v"(pf[0][2]), "v"(sacc[1][0][8]), "v"(sacc[1][0][9]), "v"(vv1_2), "v"(sacc[1][0][10]), "v"(sacc[1][0][11]), "v"(vv0_3), "v"(pf[0][3]), "v"(sacc[1][0][12]), "v"(sacc[1][0][13]), "v"(vv1_3), "v"(sacc[1][0][14]), "v"(sacc[1][0][15]) : "memory"); pf[1][0][0] = w0_0; pf[1][0][1] = w0_1; pf[1][0][2] = w0_2; pf[1][0][3] = w0_3; pf[1][1][0] = w1_0; pf[1][1][1] = w1_1; pf[1][1][2] = w1_2; pf[1][1][3] = w1_3; }
            // period 2: QK_2 | PV_1
            { float t1_0; float t1_1; float t1_2; float t1_3; float t1_4; float t1_5; float t1_6; float t1_7; float t1_8; float t1_9; float t1_10; float t1_11; float t1_12; float t1_13; float t1_14; float t1_15; uint32_t w2_0; uint32_t w2_1; uint32_t w2_2; uint32_t w2_3; uint32_t w3_0; uint32_t w3_1; uint32_t w3_2; uint32_t w3_3; asm volatile("v_mfma_f32_32x32x16_bf16 %0, %28, %29, 0\n\tv_exp_f32 %1, %30\n\tv_exp_f32 %2, %31\n\tv_mfma_f32_32x32x16_bf16 %0, %32, %33, %0\n\tv_exp_f32 %3, %34\n\tv_add_f32 %5, %5, %1\n\tv_exp_f32 %4, %35\n\tv_add_f32 %6, %6, %2\n\tv_cvt_pk_bf16_f32 %7, %1, %2\n\tv_mfma_f32_32x32x16_bf16 %0, %36, %37, %0\n\tv_exp_f32 %8, %38\n\tv_add_f32 %5, %5, %3\n\tv_exp_f32 %9, %39\n\tv_add_f32 %6, %6, %4\n\tv_cvt_pk_bf16_f32 %10, %3, %4\n\tv_mfma_f32_32x32x16_bf16 %0, %40, %41, %0\n\tv_exp_f32 %11, %42\n\tv_add_f32 %5, %5, %8\n\tv_exp_f32 %12, %43\n\tv_add_f32 %6, %6, %9\n\tv_cvt_pk_bf16_f32 %13, %8, %9\n\tv_mfma_f32_32x32x16_bf16 %14, %44, %29, 0\n\tv_exp_f32 %15, %45\n\tv_add_f32 %5, %5, %11\n\tv_exp_f32 %16, %46\n\tv_add_f32 %6, %6, %12\n\tv_cvt_pk_bf16_f32 %17, %11, %12\n\tv_mfma_f32_32x32x16_bf16 %14, %47, %33, %14\n\tv_exp_f32 %18, %48\n\tv_add_f32 %5, %5, %15\n\tv_exp_f32 %19, %49\n\tv_add_f32 %6, %6, %16\n\tv_cvt_pk_bf16_f32 %20, %15, %16\n\tv_mfma_f32_32x32x16_bf16 %14, %50, %37, %14\n\tv_exp_f32 %21, %51\n\tv_add_f32 %5, %5, %18\n\tv_exp_f32 %22, %52\n\tv_add_f32 %6, %6, %19\n\tv_cvt_pk_bf16_f32 %23, %18, %19\n\tv_mfma_f32_32x32x16_bf16 %14, %53, %41, %14\n\tv_exp_f32 %24, %54\n\tv_add_f32 %5, %5, %21\n\tv_exp_f32 %25, %55\n\tv_add_f32 %6, %6, %22\n\tv_cvt_pk_bf16_f32 %26, %21, %22\n\tv_add_f32 %5, %5, %24\n\tv_add_f32 %6, %6, %25\n\tv_cvt_pk_bf16_f32 %27, %24, %25" : "=&v"(sacc[0][0]), "=&v"(t1_0), "=&v"(t1_1), "=&v"(t1_2), "=&v"(t1_3), "+v"(psum[1][0]), "+v"(psum[1][1]), "=&v"(w2_0), "=&v"(t1_4), "=&v"(t1_5), "=&v"(w2_1), "=&v"(t1_6), "=&v"(t1_7), "=&v"(w2_2), "=&v"(sacc[0][1]), "=&v"(t1_8), "=&v"(t1_9), "=&v"(w2_3), "=&v"(t1_10), "=&v"(t1_11), "=&v"(w3_0), "=&v"(t1_12), "=&v"(t1_13), "=&v"(w3_1), "=&v"(t1_14), "=&v"(t1_15), "=&v"(w3_2), "=&v"(w3_3) : "v"(kf[0][0]), "a"(qf[2][0]), "v"(sacc[1][1][0]), "v"(sacc[1][1][1]), "v"(kf[0][1]), "a"(qf[2][1]), "v"(sacc[1][1][2]), "v"(sacc[1][1][3]), "v"(kf[0][2]), "a"(qf[2][2]), "v"(sacc[1][1][4]), "v"(sacc[1][1][5]), "v"(kf[0][3]), "a"(qf[2][3]), "v"(sacc[1][1][6]), "v"(sacc[1][1][7]), "v"(kf[1][0]), "v"(sacc[1][1][8]), "v"(sacc[1][1][9]), "v"(kf[1][1]), "v"(sacc[1][1][10]), "v"(sacc[1][1][11]), "v"(kf[1][2]), "v"(sacc[1][1][12]), "v"(sacc[1][1][13]), "v"(kf[1][3]), "v"(sacc[1][1][14]), "v"(sacc[1][1][15]) : "memory"); pf[1][2][0] = w2_0; pf[1][2][1] = w2_1; pf[1][2][2] = w2_2; pf[1][2][3] = w2_3; pf[1][3][0] = w3_0; pf[1][3][1] = w3_1; pf[1][3][2] = w3_2; pf[1][3][3] = w3_3; }
            { const u32x4 vv0_0 = {vh[0][0][0][0], vh[0][0][0][1], vh[0][0][1][0], vh[0][0][1][1]}; const u32x4 vv1_0 = {vh[1][0][0][0], vh[1][0][0][1], vh[1][0][1][0], vh[1][0][1][1]}; const u32x4 vv0_1 = {vh[0][1][0][0], vh[0][1][0][1], vh[0][1][1][0], vh[0][1][1][1]}; const u32x4 vv1_1 = {vh[1][1][0][0], vh[1][1][0][1], vh[1][1][1][0], vh[1][1][1][1]}; const u32x4 vv0_2 = {vh[0][2][0][0], vh[0][2][0][1], vh[0][2][1][0], vh[0][2][1][1]}; const u32x4 vv1_2 = {vh[1][2][0][0], vh[1][2][0][1], vh[1][2][1][0], vh[1][2][1][1]}; const u32x4 vv0_3 = {vh[0][3][0][0], vh[0][3][0][1], vh[0][3][1][0], vh[0][3][1][1]}; const u32x4 vv1_3 = {vh[1][3][0][0], vh[1][3][0][1], vh[1][3][1][0], vh[1][3][1][1]}; float t0_0; float t0_1; float t0_2; float t0_3; float t0_4; float t0_5; float t0_6; float t0_7; float t0_8; float t0_9; float t0_10; float t0_11; float t0_12; float t0_13; float t0_14; float t0_15; uint32_t w0_0; uint32_t w0_1; uint32_t w0_2; uint32_t w0_3; uint32_t w1_0; uint32_t w1_1; uint32_t w1_2; uint32_t w1_3; asm volatile("v_mfma_f32_32x32x16_bf16 %0, %28, %29, %0\n\tv_exp_f32 %1, %30\n\tv_exp_f32 %2, %31\n\tv_mfma_f32_32x32x16_bf16 %3, %32, %29, %3\n\tv_exp_f32 %4, %33\n\tv_add_f32 %6, %6, %1\n\tv_exp_f32 %5, %34\n\tv_add_f32 %7, %7, %2\n\tv_cvt_pk_bf16_f32 %8, %1, %2\n\tv_mfma_f32_32x32x16_bf16 %0, %35, %36, %0\n\tv_exp_f32 %9, %37\n\tv_add_f32 %6, %6, %4\n\tv_exp_f32 %10, %38\n\tv_add_f32 %7, %7, %5\n\tv_cvt_pk_bf16_f32 %11, %4, %5\n\tv_mfma_f32_32x32x16_bf16 %3, %39, %36, %3\n\tv_exp_f32 %12, %40\n\tv_add_f32 %6, %6, %9\n\tv_exp_f32 %13, %41\n\tv_add_f32 %7, %7, %10\n\tv_cvt_pk_bf16_f32 %14, %9, %10\n\tv_mfma_f32_32x32x16_bf16 %0, %42, %43, %0\n\tv_exp_f32 %15, %44\n\tv_add_f32 %6, %6, %12\n\tv_exp_f32 %16, %45\n\tv_add_f32 %7, %7, %13\n\tv_cvt_pk_bf16_f32 %17, %12, %13\n\tv_mfma_f32_32x32x16_bf16 %3, %46, %43, %3\n\tv_exp_f32 %18, %47\n\tv_add_f32 %6, %6, %15\n\tv_exp_f32 %19, %48\n\tv_add_f32 %7, %7, %16\n\tv_cvt_pk_bf16_f32 %20, %15, %16\n\tv_mfma_f32_32x32x16_bf16 %0, %49, %50, %0\n\tv_exp_f32 %21, %51\n\tv_add_f32 %6, %6, %18\n\tv_exp_f32 %22, %52\n\tv_add_f32 %7, %7, %19\n\tv_cvt_pk_bf16_f32 %23, %18, %19\n\tv_mfma_f32_32x32x16_bf16 %3, %53, %50, %3\n\tv_exp_f32 %24, %54\n\tv_add_f32 %6, %6, %21\n\tv_exp_f32 %25, %55\n\tv_add_f32 %7, %7, %22\n\tv_cvt_pk_bf16_f32 %26, %21, %22\n\tv_add_f32 %6, %6, %24\n\tv_add_f32 %7, %7, %25\n\tv_cvt_pk_bf16_f32 %27, %24, %25" : "+a"(oacc[1][0]), "=&v"(t0_0), "=&v"(t0_1), "+a"(oacc[1][1]), "=&v"(t0_2), "=&v"(t0_3), "+v"(psum[2][0]), "+v"(psum[2][1]), "=&v"(w0_0), "=&v"(t0_4), "=&v"(t0_5), "=&v"(w0_1), "=&v"(t0_6), "=&v"(t0_7), "=&v"(w0_2), "=&v"(t0_8), "=&v"(t0_9), "=&v"(w0_3), "=&v"(t0_10), "=&v"(t0_11), "=&v"(w1_0), "=&v"(t0_12), "=&v"(t0_13), "=&v"(w1_1), "=&v"(t0_14), "=&v"(t0_15), "=&v"(w1_2), "=&v"(w1_3) : "v"(vv0_0), "v"(pf[1][0]), "v"(sacc[0][0][0]), "v"(sacc[0][0][1]), "v"(vv1_0), "v"(sacc[0][0][2]), "v"(sacc[0][0][3]), "v"(vv0_1), "v"(pf[1][1]), "v"(sacc[0][0][4]), "v"(sacc[0][0][5]), "v"(vv1_1), "v"(sacc[0][0][6]), "v"(sacc[0][0][7]), "v"(vv0_2), "v"(pf[1][2]), "v"(sacc[0][0][8]), "v"(sacc[0][0][9]), "v"(vv1_2), "v"(sacc[0][0][10]), "v"(sacc[0][0][11]), "v"(vv0_3), "v"(pf[1][3]), "v"(sacc[0][0][12]), "v"(sacc[0][0][13]), "v"(vv1_3), "v"(sacc[0][0][14]), "v"(sacc[0][0][15]) : "memory"); pf[0][0][0] = w0_0; pf[0][0][1] = w0_1; pf[0][0][2] = w0_2; pf[0][0][3] = w0_3; pf[0][1][0] = w1_0; pf[0][1][1] = w1_1; pf[0][1][2] = w1_2; pf[0][1][3] = w1_3; }
            // period 3: QK_3 | PV_2
            { float t1_0; float t1_1; float t1_2; float t1_3; float t1_4; float t1_5; float t1_6; float t1_7; float t1_8; float t1_9; float t1_10; float t1_11; float t1_12; float t1_13; float t1_14; float t1_15; uint32_t w2_0; uint32_t w2_1; uint32_t w2_2; uint32_t w2_3; uint32_t w3_0; uint32_t w3_1; uint32_t w3_2; uint32_t w3_3; asm volatile("v_mfma_f32_32x32x16_bf16 %0, %28, %29, 0\n\tv_exp_f32 %1, %30\n\tv_exp_f32 %2, %31\n\tv_mfma_f32_32x32x16_bf16 %0, %32, %33, %0\n\tv_exp_f32 %3, %34\n\tv_add_f32 %5, %5, %1\n\tv_exp_f32 %4, %35\n\tv_add_f32 %6, %6, %2\n\tv_cvt_pk_bf16_f32 %7, %1, %2\n\tv_mfma_f32_32x32x16_bf16 %0, %36, %37, %0\n\tv_exp_f32 %8, %38\n\tv_add_f32 %5, %5, %3\n\tv_exp_f32 %9, %39\n\tv_add_f32 %6, %6, %4\n\tv_cvt_pk_bf16_f32 %10, %3, %4\n\tv_mfma_f32_32x32x16_bf16 %0, %40, %41, %0\n\tv_exp_f32 %11, %42\n\tv_add_f32 %5, %5, %8\n\tv_exp_f32 %12, %43\n\tv_add_f32 %6, %6, %9\n\tv_cvt_pk_bf16_f32 %13, %8, %9\n\tv_mfma_f32_32x32x16_bf16 %14, %44, %29, 0\n\tv_exp_f32 %15, %45\n\tv_add_f32 %5, %5, %11\n\tv_exp_f32 %16, %46\n\tv_add_f32 %6, %6, %12\n\tv_cvt_pk_bf16_f32 %17, %11, %12\n\tv_mfma_f32_32x32x16_bf16 %14, %47, %33, %14\n\tv_exp_f32 %18, %48\n\tv_add_f32 %5, %5, %15\n\tv_exp_f32 %19, %49\n\tv_add_f32 %6, %6, %16\n\tv_cvt_pk_bf16_f32 %20, %15, %16\n\tv_mfma_f32_32x32x16_bf16 %14, %50, %37, %14\n\tv_exp_f32 %21, %51\n\tv_add_f32 %5, %5, %18\n\tv_exp_f32 %22, %52\n\tv_add_f32 %6, %6, %19\n\tv_cvt_pk_bf16_f32 %23, %18, %19\n\tv_mfma_f32_32x32x16_bf16 %14, %53, %41, %14\n\tv_exp_f32 %24, %54\n\tv_add_f32 %5, %5, %21\n\tv_exp_f32 %25, %55\n\tv_add_f32 %6, %6, %22\n\tv_cvt_pk_bf16_f32 %26, %21, %22\n\tv_add_f32 %5, %5, %24\n\tv_add_f32 %6, %6, %25\n\tv_cvt_pk_bf16_f32 %27, %24, %25" : "=&v"(sacc[1][0]), "=&v"(t1_0), "=&v"(t1_1), "=&v"(t1_2), "=&v"(t1_3), "+v"(psum[2][0]), "+v"(psum[2][1]), "=&v"(w2_0), "=&v"(t1_4), "=&v"(t1_5), "=&v"(w2_1), "=&v"(t1_6), "=&v"(t1_7), "=&v"(w2_2), "=&v"(sacc[1][1]), "=&v"(t1_8), "=&v"(t1_9), "=&v"(w2_3), "=&v"(t1_10), "=&v"(t1_11), "=&v"(w3_0), "=&v"(t1_12), "=&v"(t1_13), "=&v"(w3_1), "=&v"(t1_14), "=&v"(t1_15), "=&v"(w3_2), "=&v"(w3_3) : "v"(kf[0][0]), "a"(qf[3][0]), "v"(sacc[0][1][0]), "v"(sacc[0][1][1]), "v"(kf[0][1]), "a"(qf[3][1]), "v"(sacc[0][1][2]), "v"(sacc[0][1][3]), "v"(kf[0][2]), "a"(qf[3][2]), "v"(sacc[0][1][4]), "v"(sacc[0][1][5]), "v"(kf[0][3]), "a"(qf[3][3]), "v"(sacc[0][1][6]), "v"(sacc[0][1][7]), "v"(kf[1][0]), "v"(sacc[0][1][8]), "v"(sacc[0][1][9]), "v"(kf[1][1]), "v"(sacc[0][1][10]), "v"(sacc[0][1][11]), "v"(kf[1][2]), "v"(sacc[0][1][12]), "v"(sacc[0][1][13]), "v"(kf[1][3]), "v"(sacc[0][1][14]), "v"(sacc[0][1][15]) : "memory"); pf[0][2][0] = w2_0; pf[0][2][1] = w2_1; pf[0][2][2] = w2_2; pf[0][2][3] = w2_3; pf[0][3][0] = w3_0; pf[0][3][1] = w3_1; pf[0][3][2] = w3_2; pf[0][3][3] = w3_3; }
            { const u32x4 vv0_0 = {vh[0][0][0][0], vh[0][0][0][1], vh[0][0][1][0], vh[0][0][1][1]}; const u32x4 vv1_0 = {vh[1][0][0][0], vh[1][0][0][1], vh[1][0][1][0], vh[1][0][1][1]}; const u32x4 vv0_1 = {vh[0][1][0][0], vh[0][1][0][1], vh[0][1][1][0], vh[0][1][1][1]}; const u32x4 vv1_1 = {vh[1][1][0][0], vh[1][1][0][1], vh[1][1][1][0], vh[1][1][1][1]}; const u32x4 vv0_2 = {vh[0][2][0][0], vh[0][2][0][1], vh[0][2][1][0], vh[0][2][1][1]}; const u32x4 vv1_2 = {vh[1][2][0][0], vh[1][2][0][1], vh[1][2][1][0], vh[1][2][1][1]}; const u32x4 vv0_3 = {vh[0][3][0][0], vh[0][3][0][1], vh[0][3][1][0], vh[0][3][1][1]}; const u32x4 vv1_3 = {vh[1][3][0][0], vh[1][3][0][1], vh[1][3][1][0], vh[1][3][1][1]}; const uint32_t dma_dst0 = dma_dst + 0 + 0 * TILE_BYTES; const uint32_t dma_dst1 = dma_dst + 1024 + 0 * TILE_BYTES; const uint32_t dma_dst2 = dma_dst + 0 + 1 * TILE_BYTES; const uint32_t dma_dst3 = dma_dst + 1024 + 1 * TILE_BYTES; float t0_0; float t0_1; float t0_2; float t0_3; float t0_4; float t0_5; float t0_6; float t0_7; float t0_8; float t0_9; float t0_10; float t0_11; float t0_12; float t0_13; float t0_14; float t0_15; uint32_t w0_0; uint32_t w0_1; uint32_t w0_2; uint32_t w0_3; uint32_t w1_0; uint32_t w1_1; uint32_t w1_2; uint32_t w1_3; asm volatile("v_mfma_f32_32x32x16_bf16 %0, %36, %37, %0\n\ts_waitcnt vmcnt(4)\n\ts_barrier\n\tds_read_b128 %1, %38 offset:0\n\tds_read_b128 %2, %39 offset:0\n\tv_exp_f32 %3, %40\n\tv_exp_f32 %4, %41\n\tv_mfma_f32_32x32x16_bf16 %5, %42, %37, %5\n\tds_read_b128 %6, %43 offset:0\n\tds_read_b128 %7, %44 offset:0\n\tv_exp_f32 %8, %45\n\tv_add_f32 %10, %10, %3\n\tv_exp_f32 %9, %46\n\tv_add_f32 %11, %11, %4\n\tv_cvt_pk_bf16_f32 %12, %3, %4\n\tv_mfma_f32_32x32x16_bf16 %0, %47, %48, %0\n\tds_read_b128 %13, %38 offset:4096\n\tds_read_b128 %14, %39 offset:4096\n\tv_exp_f32 %15, %49\n\tv_add_f32 %10, %10, %8\n\tv_exp_f32 %16, %50\n\tv_add_f32 %11, %11, %9\n\tv_cvt_pk_bf16_f32 %17, %8, %9\n\tv_mfma_f32_32x32x16_bf16 %5, %51, %48, %5\n\tds_read_b128 %18, %43 offset:4096\n\tds_read_b128 %19, %44 offset:4096\n\tv_exp_f32 %20, %52\n\tv_add_f32 %10, %10, %15\n\tv_exp_f32 %21, %53\n\tv_add_f32 %11, %11, %16\n\tv_cvt_pk_bf16_f32 %22, %15, %16\n\tv_mfma_f32_32x32x16_bf16 %0, %54, %55, %0\n\ts_mov_b32 m0, %56\n\ts_nop 0\n\tbuffer_load_dwordx4 %57, %58, %59 offen lds\n\tv_exp_f32 %23, %60\n\tv_add_f32 %10, %10, %20\n\tv_exp_f32 %24, %61\n\tv_add_f32 %11, %11, %21\n\tv_cvt_pk_bf16_f32 %25, %20, %21\n\tv_mfma_f32_32x32x16_bf16 %5, %62, %55, %5\n\ts_mov_b32 m0, %63\n\ts_nop 0\n\tbuffer_load_dwordx4 %64, %58, %59 offen lds\n\tv_exp_f32 %26, %65\n\tv_add_f32 %10, %10, %23\n\tv_exp_f32 %27, %66\n\tv_add_f32 %11, %11, %24\n\tv_cvt_pk_bf16_f32 %28, %23, %24\n\tv_mfma_f32_32x32x16_bf16 %0, %67, %68, %0\n\ts_mov_b32 m0, %69\n\ts_nop 0\n\tbuffer_load_dwordx4 %70, %71, %72 offen lds\n\tv_exp_f32 %29, %73\n\tv_add_f32 %10, %10, %26\n\tv_exp_f32 %30, %74\n\tv_add_f32 %11, %11, %27\n\tv_cvt_pk_bf16_f32 %31, %26, %27\n\tv_mfma_f32_32x32x16_bf16 %5, %75, %68, %5\n\ts_mov_b32 m0, %76\n\ts_nop 0\n\tbuffer_load_dwordx4 %77, %71, %72 offen lds\n\tv_exp_f32 %32, %78\n\tv_add_f32 %10, %10, %29\n\tv_exp_f32 %33, %79\n\tv_add_f32 %11, %11, %30\n\tv_cvt_pk_bf16_f32 %34, %29, %30\n\tv_add_f32 %10, %10, %32\n\tv_add_f32 %11, %11, %33\n\tv_cvt_pk_bf16_f32 %35, %32, %33\n\ts_waitcnt lgkmcnt(0)" : "+a"(oacc[2][0]), "=&v"(kf[0][0]), "=&v"(kf[0][1]), "=&v"(t0_0), "=&v"(t0_1), "+a"(oacc[2][1]), "=&v"(kf[0][2]), "=&v"(kf[0][3]), "=&v"(t0_2), "=&v"(t0_3), "+v"(psum[3][0]), "+v"(psum[3][1]), "=&v"(w0_0), "=&v"(kf[1][0]), "=&v"(kf[1][1]), "=&v"(t0_4), "=&v"(t0_5), "=&v"(w0_1), "=&v"(kf[1][2]), "=&v"(kf[1][3]), "=&v"(t0_6), "=&v"(t0_7), "=&v"(w0_2), "=&v"(t0_8), "=&v"(t0_9), "=&v"(w0_3), "=&v"(t0_10), "=&v"(t0_11), "=&v"(w1_0), "=&v"(t0_12), "=&v"(t0_13), "=&v"(w1_1), "=&v"(t0_14), "=&v"(t0_15), "=&v"(w1_2), "=&v"(w1_3) : "v"(vv0_0), "v"(pf[0][0]), "v"(kaddr[0]), "v"(kaddr[1]), "v"(sacc[1][0][0]), "v"(sacc[1][0][1]), "v"(vv1_0), "v"(kaddr[2]), "v"(kaddr[3]), "v"(sacc[1][0][2]), "v"(sacc[1][0][3]), "v"(vv0_1), "v"(pf[0][1]), "v"(sacc[1][0][4]), "v"(sacc[1][0][5]), "v"(vv1_1), "v"(sacc[1][0][6]), "v"(sacc[1][0][7]), "v"(vv0_2), "v"(pf[0][2]), "s"(dma_dst0), "v"(dvo[0]), "s"(rsK), "s"(soffK), "v"(sacc[1][0][8]), "v"(sacc[1][0][9]), "v"(vv1_2), "s"(dma_dst1), "v"(dvo[1]), "v"(sacc[1][0][10]), "v"(sacc[1][0][11]), "v"(vv0_3), "v"(pf[0][3]), "s"(dma_dst2), "v"(dvo[2]), "s"(rsV), "s"(soffV), "v"(sacc[1][0][12]), "v"(sacc[1][0][13]), "v"(vv1_3), "s"(dma_dst3), "v"(dvo[3]), "v"(sacc[1][0][14]), "v"(sacc[1][0][15]) : "memory"); pf[1][0][0] = w0_0; pf[1][0][1] = w0_1; pf[1][0][2] = w0_2; pf[1][0][3] = w0_3; pf[1][1][0] = w1_0; pf[1][1][1] = w1_1; pf[1][1][2] = w1_2; pf[1][1][3] = w1_3; }
        } else if constexpr (VAR == 'M') {
            // period 0: QK_0 | PV_3
            { float t1_0; float t1_1; float t1_2; float t1_3; float t1_4; float t1_5; float t1_6; float t1_7; float t1_8; float t1_9; float t1_10; float t1_11; float t1_12; float t1_13; float t1_14; float t1_15; uint32_t w2_0; uint32_t w2_1; uint32_t w2_2; uint32_t w2_3; uint32_t w3_0; uint32_t w3_1; uint32_t w3_2; uint32_t w3_3; asm volatile("v_mfma_f32_32x32x16_bf16 %0, %28, %29, 0\n\tv_exp_f32 %1, %30\n\tv_exp_f32 %2, %31\n\tv_mfma_f32_32x32x16_bf16 %0, %32, %33, %0\n\tv_exp_f32 %3, %34\n\tv_add_f32 %5, %5, %1\n\tv_exp_f32 %4, %35\n\tv_add_f32 %6, %6, %2\n\tv_cvt_pk_bf16_f32 %7, %1, %2\n\tv_mfma_f32_32x32x16_bf16 %0, %36, %37, %0\n\tv_exp_f32 %8, %38\n\tv_add_f32 %5, %5, %3\n\tv_exp_f32 %9, %39\n\tv_add_f32 %6, %6, %4\n\tv_cvt_pk_bf16_f32 %10, %3, %4\n\tv_mfma_f32_32x32x16_bf16 %0, %40, %41, %0\n\tv_exp_f32 %11, %42\n\tv_add_f32 %5, %5, %8\n\tv_exp_f32 %12, %43\n\tv_add_f32 %6, %6, %9\n\tv_cvt_pk_bf16_f32 %13, %8, %9\n\tv_mfma_f32_32x32x16_bf16 %14, %44, %29, 0\n\tv_exp_f32 %15, %45\n\tv_add_f32 %5, %5, %11\n\tv_exp_f32 %16, %46\n\tv_add_f32 %6, %6, %12\n\tv_cvt_pk_bf16_f32 %17, %11, %12\n\tv_mfma_f32_32x32x16_bf16 %14, %47, %33, %14\n\tv_exp_f32 %18, %48\n\tv_add_f32 %5, %5, %15\n\tv_exp_f32 %19, %49\n\tv_add_f32 %6, %6, %16\n\tv_cvt_pk_bf16_f32 %20, %15, %16\n\tv_mfma_f32_32x32x16_bf16 %14, %50, %37, %14\n\tv_exp_f32 %21, %51\n\tv_add_f32 %5, %5, %18\n\tv_exp_f32 %22, %52\n\tv_add_f32 %6, %6, %19\n\tv_cvt_pk_bf16_f32 %23, %18, %19\n\tv_mfma_f32_32x32x16_bf16 %14, %53, %41, %14\n\tv_exp_f32 %24, %54\n\tv_add_f32 %5, %5, %21\n\tv_exp_f32 %25, %55\n\tv_add_f32 %6, %6, %22\n\tv_cvt_pk_bf16_f32 %26, %21, %22\n\tv_add_f32 %5, %5, %24\n\tv_add_f32 %6, %6, %25\n\tv_cvt_pk_bf16_f32 %27, %24, %25" : "=&v"(sacc[0][0]), "=&v"(t1_0), "=&v"(t1_1), "=&v"(t1_2), "=&v"(t1_3), "+v"(psum[3][0]), "+v"(psum[3][1]), "=&v"(w2_0), "=&v"(t1_4), "=&v"(t1_5), "=&v"(w2_1), "=&v"(t1_6), "=&v"(t1_7), "=&v"(w2_2), "=&v"(sacc[0][1]), "=&v"(t1_8), "=&v"(t1_9), "=&v"(w2_3), "=&v"(t1_10), "=&v"(t1_11), "=&v"(w3_0), "=&v"(t1_12), "=&v"(t1_13), "=&v"(w3_1), "=&v"(t1_14), "=&v"(t1_15), "=&v"(w3_2), "=&v"(w3_3) : "v"(kf[0][0]), "a"(qf[0][0]), "v"(sacc[1][1][0]), "v"(sacc[1][1][1]), "v"(kf[0][1]), "a"(qf[0][1]), "v"(sacc[1][1][2]), "v"(sacc[1][1][3]), "v"(kf[0][2]), "a"(qf[0][2]), "v"(sacc[1][1][4]), "v"(sacc[1][1][5]), "v"(kf[0][3]), "a"(qf[0][3]), "v"(sacc[1][1][6]), "v"(sacc[1][1][7]), "v"(kf[1][0]), "v"(sacc[1][1][8]), "v"(sacc[1][1][9]), "v"(kf[1][1]), "v"(sacc[1][1][10]), "v"(sacc[1][1][11]), "v"(kf[1][2]), "v"(sacc[1][1][12]), "v"(sacc[1][1][13]), "v"(kf[1][3]), "v"(sacc[1][1][14]), "v"(sacc[1][1][15]) : "memory"); pf[1][2][0] = w2_0; pf[1][2][1] = w2_1; pf[1][2][2] = w2_2; pf[1][2][3] = w2_3; pf[1][3][0] = w3_0; pf[1][3][1] = w3_1; pf[1][3][2] = w3_2; pf[1][3][3] = w3_3; }
            MASKPAD(sacc[0][0], 0);
            { const u32x4 vv0_0 = {vh[0][0][0][0], vh[0][0][0][1], vh[0][0][1][0], vh[0][0][1][1]}; const u32x4 vv1_0 = {vh[1][0][0][0], vh[1][0][0][1], vh[1][0][1][0], vh[1][0][1][1]}; const u32x4 vv0_1 = {vh[0][1][0][0], vh[0][1][0][1], vh[0][1][1][0], vh[0][1][1][1]}; const u32x4 vv1_1 = {vh[1][1][0][0], vh[1][1][0][1], vh[1][1][1][0], vh[1][1][1][1]}; const u32x4 vv0_2 = {vh[0][2][0][0], vh[0][2][0][1], vh[0][2][1][0], vh[0][2][1][1]}; const u32x4 vv1_2 = {vh[1][2][0][0], vh[1][2][0][1], vh[1][2][1][0], vh[1][2][1][1]}; const u32x4 vv0_3 = {vh[0][3][0][0], vh[0][3][0][1], vh[0][3][1][0], vh[0][3][1][1]}; const u32x4 vv1_3 = {vh[1][3][0][0], vh[1][3][0][1], vh[1][3][1][0], vh[1][3][1][1]}; float t0_0; float t0_1; float t0_2; float t0_3; float t0_4; float t0_5; float t0_6; float t0_7; float t0_8; float t0_9; float t0_10; float t0_11; float t0_12; float t0_13; float t0_14; float t0_15; uint32_t w0_0; uint32_t w0_1; uint32_t w0_2; uint32_t w0_3; uint32_t w1_0; uint32_t w1_1; uint32_t w1_2; uint32_t w1_3; asm volatile("v_mfma_f32_32x32x16_bf16 %0, %28, %29, %0\n\tv_exp_f32 %1, %30\n\tv_exp_f32 %2, %31\n\tv_mfma_f32_32x32x16_bf16 %3, %32, %29, %3\n\tv_exp_f32 %4, %33\n\tv_add_f32 %6, %6, %1\n\tv_exp_f32 %5, %34\n\tv_add_f32 %7, %7, %2\n\tv_cvt_pk_bf16_f32 %8, %1, %2\n\tv_mfma_f32_32x32x16_bf16 %0, %35, %36, %0\n\tv_exp_f32 %9, %37\n\tv_add_f32 %6, %6, %4\n\tv_exp_f32 %10, %38\n\tv_add_f32 %7, %7, %5\n\tv_cvt_pk_bf16_f32 %11, %4, %5\n\tv_mfma_f32_32x32x16_bf16 %3, %39, %36, %3\n\tv_exp_f32 %12, %40\n\tv_add_f32 %6, %6, %9\n\tv_exp_f32 %13, %41\n\tv_add_f32 %7, %7, %10\n\tv_cvt_pk_bf16_f32 %14, %9, %10\n\tv_mfma_f32_32x32x16_bf16 %0, %42, %43, %0\n\tv_exp_f32 %15, %44\n\tv_add_f32 %6, %6, %12\n\tv_exp_f32 %16, %45\n\tv_add_f32 %7, %7, %13\n\tv_cvt_pk_bf16_f32 %17, %12, %13\n\tv_mfma_f32_32x32x16_bf16 %3, %46, %43, %3\n\tv_exp_f32 %18, %47\n\tv_add_f32 %6, %6, %15\n\tv_exp_f32 %19, %48\n\tv_add_f32 %7, %7, %16\n\tv_cvt_pk_bf16_f32 %20, %15, %16\n\tv_mfma_f32_32x32x16_bf16 %0, %49, %50, %0\n\tv_exp_f32 %21, %51\n\tv_add_f32 %6, %6, %18\n\tv_exp_f32 %22, %52\n\tv_add_f32 %7, %7, %19\n\tv_cvt_pk_bf16_f32 %23, %18, %19\n\tv_mfma_f32_32x32x16_bf16 %3, %53, %50, %3\n\tv_exp_f32 %24, %54\n\tv_add_f32 %6, %6, %21\n\tv_exp_f32 %25, %55\n\tv_add_f32 %7, %7, %22\n\tv_cvt_pk_bf16_f32 %26, %21, %22\n\tv_add_f32 %6, %6, %24\n\tv_add_f32 %7, %7, %25\n\tv_cvt_pk_bf16_f32 %27, %24, %25" : "+a"(oacc[3][0]), "=&v"(t0_0), "=&v"(t0_1), "+a"(oacc[3][1]), "=&v"(t0_2), "=&v"(t0_3), "+v"(psum[0][0]), "+v"(psum[0][1]), "=&v"(w0_0), "=&v"(t0_4), "=&v"(t0_5), "=&v"(w0_1), "=&v"(t0_6), "=&v"(t0_7), "=&v"(w0_2), "=&v"(t0_8), "=&v"(t0_9), "=&v"(w0_3), "=&v"(t0_10), "=&v"(t0_11), "=&v"(w1_0), "=&v"(t0_12), "=&v"(t0_13), "=&v"(w1_1), "=&v"(t0_14), "=&v"(t0_15), "=&v"(w1_2), "=&v"(w1_3) : "v"(vv0_0), "v"(pf[1][0]), "v"(sacc[0][0][0]), "v"(sacc[0][0][1]), "v"(vv1_0), "v"(sacc[0][0][2]), "v"(sacc[0][0][3]), "v"(vv0_1), "v"(pf[1][1]), "v"(sacc[0][0][4]), "v"(sacc[0][0][5]), "v"(vv1_1), "v"(sacc[0][0][6]), "v"(sacc[0][0][7]), "v"(vv0_2), "v"(pf[1][2]), "v"(sacc[0][0][8]), "v"(sacc[0][0][9]), "v"(vv1_2), "v"(sacc[0][0][10]), "v"(sacc[0][0][11]), "v"(vv0_3), "v"(pf[1][3]), "v"(sacc[0][0][12]), "v"(sacc[0][0][13]), "v"(vv1_3), "v"(sacc[0][0][14]), "v"(sacc[0][0][15]) : "memory"); pf[0][0][0] = w0_0; pf[0][0][1] = w0_1; pf[0][0][2] = w0_2; pf[0][0][3] = w0_3; pf[0][1][0] = w1_0; pf[0][1][1] = w1_1; pf[0][1][2] = w1_2; pf[0][1][3] = w1_3; }
            MASKPAD(sacc[0][1], 32);
            // period 1: QK_1 | PV_0
            { float t1_0; float t1_1; float t1_2; float t1_3; float t1_4; float t1_5; float t1_6; float t1_7; float t1_8; float t1_9; float t1_10; float t1_11; float t1_12; float t1_13; float t1_14; float t1_15; uint32_t w2_0; uint32_t w2_1; uint32_t w2_2; uint32_t w2_3; uint32_t w3_0; uint32_t w3_1; uint32_t w3_2; uint32_t w3_3; asm volatile("v_mfma_f32_32x32x16_bf16 %0, %44, %45, 0\n\tds_read_b64_tr_b16 %1, %46 offset:0\n\tds_read_b64_tr_b16 %2, %46 offset:1024\n\tds_read_b64_tr_b16 %3, %47 offset:0\n\tv_exp_f32 %4, %48\n\tv_exp_f32 %5, %49\n\tv_mfma_f32_32x32x16_bf16 %0, %50, %51, %0\n\tds_read_b64_tr_b16 %6, %47 offset:1024\n\tds_read_b64_tr_b16 %7, %46 offset:2048\n\tds_read_b64_tr_b16 %8, %46 offset:3072\n\tv_exp_f32 %9, %52\n\tv_add_f32 %11, %11, %4\n\tv_exp_f32 %10, %53\n\tv_add_f32 %12, %12, %5\n\tv_cvt_pk_bf16_f32 %13, %4, %5\n\tv_mfma_f32_32x32x16_bf16 %0, %54, %55, %0\n\tds_read_b64_tr_b16 %14, %47 offset:2048\n\tds_read_b64_tr_b16 %15, %47 offset:3072\n\tds_read_b64_tr_b16 %16, %46 offset:4096\n\tv_exp_f32 %17, %56\n\tv_add_f32 %11, %11, %9\n\tv_exp_f32 %18, %57\n\tv_add_f32 %12, %12, %10\n\tv_cvt_pk_bf16_f32 %19, %9, %10\n\tv_mfma_f32_32x32x16_bf16 %0, %58, %59, %0\n\tds_read_b64_tr_b16 %20, %46 offset:5120\n\tds_read_b64_tr_b16 %21, %47 offset:4096\n\tds_read_b64_tr_b16 %22, %47 offset:5120\n\tv_exp_f32 %23, %60\n\tv_add_f32 %11, %11, %17\n\tv_exp_f32 %24, %61\n\tv_add_f32 %12, %12, %18\n\tv_cvt_pk_bf16_f32 %25, %17, %18\n\tv_mfma_f32_32x32x16_bf16 %26, %62, %45, 0\n\tds_read_b64_tr_b16 %27, %46 offset:6144\n\tds_read_b64_tr_b16 %28, %46 offset:7168\n\tv_exp_f32 %29, %63\n\tv_add_f32 %11, %11, %23\n\tv_exp_f32 %30, %64\n\tv_add_f32 %12, %12, %24\n\tv_cvt_pk_bf16_f32 %31, %23, %24\n\tv_mfma_f32_32x32x16_bf16 %26, %65, %51, %26\n\tds_read_b64_tr_b16 %32, %47 offset:6144\n\tds_read_b64_tr_b16 %33, %47 offset:7168\n\tv_exp_f32 %34, %66\n\tv_add_f32 %11, %11, %29\n\tv_exp_f32 %35, %67\n\tv_add_f32 %12, %12, %30\n\tv_cvt_pk_bf16_f32 %36, %29, %30\n\tv_mfma_f32_32x32x16_bf16 %26, %68, %55, %26\n\tv_exp_f32 %37, %69\n\tv_add_f32 %11, %11, %34\n\tv_exp_f32 %38, %70\n\tv_add_f32 %12, %12, %35\n\tv_cvt_pk_bf16_f32 %39, %34, %35\n\tv_mfma_f32_32x32x16_bf16 %26, %71, %59, %26\n\tv_exp_f32 %40, %72\n\tv_add_f32 %11, %11, %37\n\tv_exp_f32 %41, %73\n\tv_add_f32 %12, %12, %38\n\tv_cvt_pk_bf16_f32 %42, %37, %38\n\tv_add_f32 %11, %11, %40\n\tv_add_f32 %12, %12, %41\n\tv_cvt_pk_bf16_f32 %43, %40, %41\n\ts_waitcnt lgkmcnt(0)" : "=&v"(sacc[1][0]), "=&v"(vh[0][0][0]), "=&v"(vh[0][0][1]), "=&v"(vh[1][0][0]), "=&v"(t1_0), "=&v"(t1_1), "=&v"(vh[1][0][1]), "=&v"(vh[0][1][0]), "=&v"(vh[0][1][1]), "=&v"(t1_2), "=&v"(t1_3), "+v"(psum[0][0]), "+v"(psum[0][1]), "=&v"(w2_0), "=&v"(vh[1][1][0]), "=&v"(vh[1][1][1]), "=&v"(vh[0][2][0]), "=&v"(t1_4), "=&v"(t1_5), "=&v"(w2_1), "=&v"(vh[0][2][1]), "=&v"(vh[1][2][0]), "=&v"(vh[1][2][1]), "=&v"(t1_6), "=&v"(t1_7), "=&v"(w2_2), "=&v"(sacc[1][1]), "=&v"(vh[0][3][0]), "=&v"(vh[0][3][1]), "=&v"(t1_8), "=&v"(t1_9), "=&v"(w2_3), "=&v"(vh[1][3][0]), "=&v"(vh[1][3][1]), "=&v"(t1_10), "=&v"(t1_11), "=&v"(w3_0), "=&v"(t1_12), "=&v"(t1_13), "=&v"(w3_1), "=&v"(t1_14), "=&v"(t1_15), "=&v"(w3_2), "=&v"(w3_3) : "v"(kf[0][0]), "a"(qf[1][0]), "v"(vaddr[0]), "v"(vaddr[1]), "v"(sacc[0][1][0]), "v"(sacc[0][1][1]), "v"(kf[0][1]), "a"(qf[1][1]), "v"(sacc[0][1][2]), "v"(sacc[0][1][3]), "v"(kf[0][2]), "a"(qf[1][2]), "v"(sacc[0][1][4]), "v"(sacc[0][1][5]), "v"(kf[0][3]), "a"(qf[1][3]), "v"(sacc[0][1][6]), "v"(sacc[0][1][7]), "v"(kf[1][0]), "v"(sacc[0][1][8]), "v"(sacc[0][1][9]), "v"(kf[1][1]), "v"(sacc[0][1][10]), "v"(sacc[0][1][11]), "v"(kf[1][2]), "v"(sacc[0][1][12]), "v"(sacc[0][1][13]), "v"(kf[1][3]), "v"(sacc[0][1][14]), "v"(sacc[0][1][15]) : "memory"); pf[0][2][0] = w2_0; pf[0][2][1] = w2_1; pf[0][2][2] = w2_2; pf[0][2][3] = w2_3; pf[0][3][0] = w3_0; pf[0][3][1] = w3_1; pf[0][3][2] = w3_2; pf[0][3][3] = w3_3; }
            MASKPAD(sacc[1][0], 0);
            { const u32x4 vv0_0 = {vh[0][0][0][0], vh[0][0][0][1], vh[0][0][1][0], vh[0][0][1][1]}; const u32x4 vv1_0 = {vh[1][0][0][0], vh[1][0][0][1], vh[1][0][1][0], vh[1][0][1][1]}; const u32x4 vv0_1 = {vh[0][1][0][0], vh[0][1][0][1], vh[0][1][1][0], vh[0][1][1][1]}; const u32x4 vv1_1 = {vh[1][1][0][0], vh[1][1][0][1], vh[1][1][1][0], vh[1][1][1][1]}; const u32x4 vv0_2 = {vh[0][2][0][0], vh[0][2][0][1], vh[0][2][1][0], vh[0][2][1][1]}; const u32x4 vv1_2 = {vh[1][2][0][0], vh[1][2][0][1], vh[1][2][1][0], vh[1][2][1][1]}; const u32x4 vv0_3 = {vh[0][3][0][0], vh[0][3][0][1], vh[0][3][1][0], vh[0][3][1][1]}; const u32x4 vv1_3 = {vh[1][3][0][0], vh[1][3][0][1], vh[1][3][1][0], vh[1][3][1][1]}; float t0_0; float t0_1; float t0_2; float t0_3; float t0_4; float t0_5; float t0_6; float t0_7; float t0_8; float t0_9; float t0_10; float t0_11; float t0_12; float t0_13; float t0_14; float t0_15; uint32_t w0_0; uint32_t w0_1; uint32_t w0_2; uint32_t w0_3; uint32_t w1_0; uint32_t w1_1; uint32_t w1_2; uint32_t w1_3; asm volatile("v_mfma_f32_32x32x16_bf16 %0, %28, %29, %0\n\tv_exp_f32 %1, %30\n\tv_exp_f32 %2, %31\n\tv_mfma_f32_32x32x16_bf16 %3, %32, %29, %3\n\tv_exp_f32 %4, %33\n\tv_add_f32 %6, %6, %1\n\tv_exp_f32 %5, %34\n\tv_add_f32 %7, %7, %2\n\tv_cvt_pk_bf16_f32 %8, %1, %2\n\tv_mfma_f32_32x32x16_bf16 %0, %35, %36, %0\n\tv_exp_f32 %9, %37\n\tv_add_f32 %6, %6, %4\n\tv_exp_f32 %10, %38\n\tv_add_f32 %7, %7, %5\n\tv_cvt_pk_bf16_f32 %11, %4, %5\n\tv_mfma_f32_32x32x16_bf16 %3, %39, %36, %3\n\tv_exp_f32 %12, %40\n\tv_add_f32 %6, %6, %9\n\tv_exp_f32 %13, %41\n\tv_add_f32 %7, %7, %10\n\tv_cvt_pk_bf16_f32 %14, %9, %10\n\tv_mfma_f32_32x32x16_bf16 %0, %42, %43, %0\n\tv_exp_f32 %15, %44\n\tv_add_f32 %6, %6, %12\n\tv_exp_f32 %16, %45\n\tv_add_f32 %7, %7, %13\n\tv_cvt_pk_bf16_f32 %17, %12, %13\n\tv_mfma_f32_32x32x16_bf16 %3, %46, %43, %3\n\tv_exp_f32 %18, %47\n\tv_add_f32 %6, %6, %15\n\tv_exp_f32 %19, %48\n\tv_add_f32 %7, %7, %16\n\tv_cvt_pk_bf16_f32 %20, %15, %16\n\tv_mfma_f32_32x32x16_bf16 %0, %49, %50, %0\n\tv_exp_f32 %21, %51\n\tv_add_f32 %6, %6, %18\n\tv_exp_f32 %22, %52\n\tv_add_f32 %7, %7, %19\n\tv_cvt_pk_bf16_f32 %23, %18, %19\n\tv_mfma_f32_32x32x16_bf16 %3, %53, %50, %3\n\tv_exp_f32 %24, %54\n\tv_add_f32 %6, %6, %21\n\tv_exp_f32 %25, %55\n\tv_add_f32 %7, %7, %22\n\tv_cvt_pk_bf16_f32 %26, %21, %22\n\tv_add_f32 %6, %6, %24\n\tv_add_f32 %7, %7, %25\n\tv_cvt_pk_bf16_f32 %27, %24, %25" : "+a"(oacc[0][0]), "=&v"(t0_0), "=&v"(t0_1), "+a"(oacc[0][1]), "=&v"(t0_2), "=&v"(t0_3), "+v"(psum[1][0]), "+v"(psum[1][1]), "=&v"(w0_0), "=&v"(t0_4), "=&v"(t0_5), "=&v"(w0_1), "=&v"(t0_6), "=&v"(t0_7), "=&v"(w0_2), "=&v"(t0_8), "=&v"(t0_9), "=&v"(w0_3), "=&v"(t0_10), "=&v"(t0_11), "=&v"(w1_0), "=&v"(t0_12), "=&v"(t0_13), "=&v"(w1_1), "=&v"(t0_14), "=&v"(t0_15), "=&v"(w1_2), "=&v"(w1_3) : "v"(vv0_0), "v"(pf[0][0]), "v"(sacc[1][0][0]), "v"(sacc[1][0][1]), "v"(vv1_0), "v"(sacc[1][0][2]), "v"(sacc[1][0][3]), "v"(vv0_1), "v"(pf[0][1]), "v"(sacc[1][0][4]), "v"(sacc[1][0][5]), "v"(vv1_1), "v"(sacc[1][0][6]), "v"(sacc[1][0][7]), "v"(vv0_2), "v"(pf[0][2]), "v"(sacc[1][0][8]), "v"(sacc[1][0][9]), "v"(vv1_2), "v"(sacc[1][0][10]), "v"(sacc[1][0][11]), "v"(vv0_3), "v"(pf[0][3]), "v"(sacc[1][0][12]), "v"(sacc[1][0][13]), "v"(vv1_3), "v"(sacc[1][0][14]), "v"(sacc[1][0][15]) : "memory"); pf[1][0][0] = w0_0; pf[1][0][1] = w0_1; pf[1][0][2] = w0_2; pf[1][0][3] = w0_3; pf[1][1][0] = w1_0; pf[1][1][1] = w1_1; pf[1][1][2] = w1_2; pf[1][1][3] = w1_3; }
            MASKPAD(sacc[1][1], 32);
            // period 2: QK_2 | PV_1
            { float t1_0; float t1_1; float t1_2; float t1_3; float t1_4; float t1_5; float t1_6; float t1_7; float t1_8; float t1_9; float t1_10; float t1_11; float t1_12; float t1_13; float t1_14; float t1_15; uint32_t w2_0; uint32_t w2_1; uint32_t w2_2; uint32_t w2_3; uint32_t w3_0; uint32_t w3_1; uint32_t w3_2; uint32_t w3_3; asm volatile("v_mfma_f32_32x32x16_bf16 %0, %28, %29, 0\n\tv_exp_f32 %1, %30\n\tv_exp_f32 %2, %31\n\tv_mfma_f32_32x32x16_bf16 %0, %32, %33, %0\n\tv_exp_f32 %3, %34\n\tv_add_f32 %5, %5, %1\n\tv_exp_f32 %4, %35\n\tv_add_f32 %6, %6, %2\n\tv_cvt_pk_bf16_f32 %7, %1, %2\n\tv_mfma_f32_32x32x16_bf16 %0, %36, %37, %0\n\tv_exp_f32 %8, %38\n\tv_add_f32 %5, %5, %3\n\tv_exp_f32 %9, %39\n\tv_add_f32 %6, %6, %4\n\tv_cvt_pk_bf16_f32 %10, %3, %4\n\tv_mfma_f32_32x32x16_bf16 %0, %40, %41, %0\n\tv_exp_f32 %11, %42\n\tv_add_f32 %5, %5, %8\n\tv_exp_f32 %12, %43\n\tv_add_f32 %6, %6, %9\n\tv_cvt_pk_bf16_f32 %13, %8, %9\n\tv_mfma_f32_32x32x16_bf16 %14, %44, %29, 0\n\tv_exp_f32 %15, %45\n\tv_add_f32 %5, %5, %11\n\tv_exp_f32 %16, %46\n\tv_add_f32 %6, %6, %12\n\tv_cvt_pk_bf16_f32 %17, %11, %12\n\tv_mfma_f32_32x32x16_bf16 %14, %47, %33, %14\n\tv_exp_f32 %18, %48\n\tv_add_f32 %5, %5, %15\n\tv_exp_f32 %19, %49\n\tv_add_f32 %6, %6, %16\n\tv_cvt_pk_bf16_f32 %20, %15, %16\n\tv_mfma_f32_32x32x16_bf16 %14, %50, %37, %14\n\tv_exp_f32 %21, %51\n\tv_add_f32 %5, %5, %18\n\tv_exp_f32 %22, %52\n\tv_add_f32 %6, %6, %19\n\tv_cvt_pk_bf16_f32 %23, %18, %19\n\tv_mfma_f32_32x32x16_bf16 %14, %53, %41, %14\n\tv_exp_f32 %24, %54\n\tv_add_f32 %5, %5, %21\n\tv_exp_f32 %25, %55\n\tv_add_f32 %6, %6, %22\n\tv_cvt_pk_bf16_f32 %26, %21, %22\n\tv_add_f32 %5, %5, %24\n\tv_add_f32 %6, %6, %25\n\tv_cvt_pk_bf16_f32 %27, %24, %25" : "=&v"(sacc[0][0]), "=&v"(t1_0), "=&v"(t1_1), "=&v"(t1_2), "=&v"(t1_3), "+v"(psum[1][0]), "+v"(psum[1][1]), "=&v"(w2_0), "=&v"(t1_4), "=&v"(t1_5), "=&v"(w2_1), "=&v"(t1_6), "=&v"(t1_7), "=&v"(w2_2), "=&v"(sacc[0][1]), "=&v"(t1_8), "=&v"(t1_9), "=&v"(w2_3), "=&v"(t1_10), "=&v"(t1_11), "=&v"(w3_0), "=&v"(t1_12), "=&v"(t1_13), "=&v"(w3_1), "=&v"(t1_14), "=&v"(t1_15), "=&v"(w3_2), "=&v"(w3_3) : "v"(kf[0][0]), "a"(qf[2][0]), "v"(sacc[1][1][0]), "v"(sacc[1][1][1]), "v"(kf[0][1]), "a"(qf[2][1]), "v"(sacc[1][1][2]), "v"(sacc[1][1][3]), "v"(kf[0][2]), "a"(qf[2][2]), "v"(sacc[1][1][4]), "v"(sacc[1][1][5]), "v"(kf[0][3]), "a"(qf[2][3]), "v"(sacc[1][1][6]), "v"(sacc[1][1][7]), "v"(kf[1][0]), "v"(sacc[1][1][8]), "v"(sacc[1][1][9]), "v"(kf[1][1]), "v"(sacc[1][1][10]), "v"(sacc[1][1][11]), "v"(kf[1][2]), "v"(sacc[1][1][12]), "v"(sacc[1][1][13]), "v"(kf[1][3]), "v"(sacc[1][1][14]), "v"(sacc[1][1][15]) : "memory"); pf[1][2][0] = w2_0; pf[1][2][1] = w2_1; pf[1][2][2] = w2_2; pf[1][2][3] = w2_3; pf[1][3][0] = w3_0; pf[1][3][1] = w3_1; pf[1][3][2] = w3_2; pf[1][3][3] = w3_3; }
            MASKPAD(sacc[0][0], 0);
            { const u32x4 vv0_0 = {vh[0][0][0][0], vh[0][0][0][1], vh[0][0][1][0], vh[0][0][1][1]}; const u32x4 vv1_0 = {vh[1][0][0][0], vh[1][0][0][1], vh[1][0][1][0], vh[1][0][1][1]}; const u32x4 vv0_1 = {vh[0][1][0][0], vh[0][1][0][1], vh[0][1][1][0], vh[0][1][1][1]}; const u32x4 vv1_1 = {vh[1][1][0][0], vh[1][1][0][1], vh[1][1][1][0], vh[1][1][1][1]}; const u32x4 vv0_2 = {vh[0][2][0][0], vh[0][2][0][1], vh[0][2][1][0], vh[0][2][1][1]}; const u32x4 vv1_2 = {vh[1][2][0][0], vh[1][2][0][1], vh[1][2][1][0], vh[1][2][1][1]}; const u32x4 vv0_3 = {vh[0][3][0][0], vh[0][3][0][1], vh[0][3][1][0], vh[0][3][1][1]}; const u32x4 vv1_3 = {vh[1][3][0][0], vh[1][3][0][1], vh[1][3][1][0], vh[1][3][1][1]}; float t0_0; float t0_1; float t0_2; float t0_3; float t0_4; float t0_5; float t0_6; float t0_7; float t0_8; float t0_9; float t0_10; float t0_11; float t0_12; float t0_13; float t0_14; float t0_15; uint32_t w0_0; uint32_t w0_1; uint32_t w0_2; uint32_t w0_3; uint32_t w1_0; uint32_t w1_1; uint32_t w1_2; uint32_t w1_3; asm volatile("v_mfma_f32_32x32x16_bf16 %0, %28, %29, %0\n\tv_exp_f32 %1, %30\n\tv_exp_f32 %2, %31\n\tv_mfma_f32_32x32x16_bf16 %3, %32, %29, %3\n\tv_exp_f32 %4, %33\n\tv_add_f32 %6, %6, %1\n\tv_exp_f32 %5, %34\n\tv_add_f32 %7, %7, %2\n\tv_cvt_pk_bf16_f32 %8, %1, %2\n\tv_mfma_f32_32x32x16_bf16 %0, %35, %36, %0\n\tv_exp_f32 %9, %37\n\tv_add_f32 %6, %6, %4\n\tv_exp_f32 %10, %38\n\tv_add_f32 %7, %7, %5\n\tv_cvt_pk_bf16_f32 %11, %4, %5\n\tv_mfma_f32_32x32x16_bf16 %3, %39, %36, %3\n\tv_exp_f32 %12, %40\n\tv_add_f32 %6, %6, %9\n\tv_exp_f32 %13, %41\n\tv_add_f32 %7, %7, %10\n\tv_cvt_pk_bf16_f32 %14, %9, %10\n\tv_mfma_f32_32x32x16_bf16 %0, %42, %43, %0\n\tv_exp_f32 %15, %44\n\tv_add_f32 %6, %6, %12\n\tv_exp_f32 %16, %45\n\tv_add_f32 %7, %7, %13\n\tv_cvt_pk_bf16_f32 %17, %12, %13\n\tv_mfma_f32_32x32x16_bf16 %3, %46, %43, %3\n\tv_exp_f32 %18, %47\n\tv_add_f32 %6, %6, %15\n\tv_exp_f32 %19, %48\n\tv_add_f32 %7, %7, %16\n\tv_cvt_pk_bf16_f32 %20, %15, %16\n\tv_mfma_f32_32x32x16_bf16 %0, %49, %50, %0\n\tv_exp_f32 %21, %51\n\tv_add_f32 %6, %6, %18\n\tv_exp_f32 %22, %52\n\tv_add_f32 %7, %7, %19\n\tv_cvt_pk_bf16_f32 %23, %18, %19\n\tv_mfma_f32_32x32x16_bf16 %3, %53, %50, %3\n\tv_exp_f32 %24, %54\n\tv_add_f32 %6, %6, %21\n\tv_exp_f32 %25, %55\n\tv_add_f32 %7, %7, %22\n\tv_cvt_pk_bf16_f32 %26, %21, %22\n\tv_add_f32 %6, %6, %24\n\tv_add_f32 %7, %7, %25\n\tv_cvt_pk_bf16_f32 %27, %24, %25" : "+a"(oacc[1][0]), "=&v"(t0_0), "=&v"(t0_1), "+a"(oacc[1][1]), "=&v"(t0_2), "=&v"(t0_3), "+v"(psum[2][0]), "+v"(psum[2][1]), "=&v"(w0_0), "=&v"(t0_4), "=&v"(t0_5), "=&v"(w0_1), "=&v"(t0_6), "=&v"(t0_7), "=&v"(w0_2), "=&v"(t0_8), "=&v"(t0_9), "=&v"(w0_3), "=&v"(t0_10), "=&v"(t0_11), "=&v"(w1_0), "=&v"(t0_12), "=&v"(t0_13), "=&v"(w1_1), "=&v"(t0_14), "=&v"(t0_15), "=&v"(w1_2), "=&v"(w1_3) : "v"(vv0_0), "v"(pf[1][0]), "v"(sacc[0][0][0]), "v"(sacc[0][0][1]), "v"(vv1_0), "v"(sacc[0][0][2]), "v"(sacc[0][0][3]), "v"(vv0_1), "v"(pf[1][1]), "v"(sacc[0][0][4]), "v"(sacc[0][0][5]), "v"(vv1_1), "v"(sacc[0][0][6]), "v"(sacc[0][0][7]), "v"(vv0_2), "v"(pf[1][2]), "v"(sacc[0][0][8]), "v"(sacc[0][0][9]), "v"(vv1_2), "v"(sacc[0][0][10]), "v"(sacc[0][0][11]), "v"(vv0_3), "v"(pf[1][3]), "v"(sacc[0][0][12]), "v"(sacc[0][0][13]), "v"(vv1_3), "v"(sacc[0][0][14]), "v"(sacc[0][0][15]) : "memory"); pf[0][0][0] = w0_0; pf[0][0][1] = w0_1; pf[0][0][2] = w0_2; pf[0][0][3] = w0_3; pf[0][1][0] = w1_0; pf[0][1][1] = w1_1; pf[0][1][2] = w1_2; pf[0][1][3] = w1_3; }
            MASKPAD(sacc[0][1], 32);
            // period 3: QK_3 | PV_2
            { float t1_0; float t1_1; float t1_2; float t1_3; float t1_4; float t1_5; float t1_6; float t1_7; float t1_8; float t1_9; float t1_10; float t1_11; float t1_12; float t1_13; float t1_14; float t1_15; uint32_t w2_0; uint32_t w2_1; uint32_t w2_2; uint32_t w2_3; uint32_t w3_0; uint32_t w3_1; uint32_t w3_2; uint32_t w3_3; asm volatile("v_mfma_f32_32x32x16_bf16 %0, %28, %29, 0\n\tv_exp_f32 %1, %30\n\tv_exp_f32 %2, %31\n\tv_mfma_f32_32x32x16_bf16 %0, %32, %33, %0\n\tv_exp_f32 %3, %34\n\tv_add_f32 %5, %5, %1\n\tv_exp_f32 %4, %35\n\tv_add_f32 %6, %6, %2\n\tv_cvt_pk_bf16_f32 %7, %1, %2\n\tv_mfma_f32_32x32x16_bf16 %0, %36, %37, %0\n\tv_exp_f32 %8, %38\n\tv_add_f32 %5, %5, %3\n\tv_exp_f32 %9, %39\n\tv_add_f32 %6, %6, %4\n\tv_cvt_pk_bf16_f32 %10, %3, %4\n\tv_mfma_f32_32x32x16_bf16 %0, %40, %41, %0\n\tv_exp_f32 %11, %42\n\tv_add_f32 %5, %5, %8\n\tv_exp_f32 %12, %43\n\tv_add_f32 %6, %6, %9\n\tv_cvt_pk_bf16_f32 %13, %8, %9\n\tv_mfma_f32_32x32x16_bf16 %14, %44, %29, 0\n\tv_exp_f32 %15, %45\n\tv_add_f32 %5, %5, %11\n\tv_exp_f32 %16, %46\n\tv_add_f32 %6, %6, %12\n\tv_cvt_pk_bf16_f32 %17, %11, %12\n\tv_mfma_f32_32x32x16_bf16 %14, %47, %33, %14\n\tv_exp_f32 %18, %48\n\tv_add_f32 %5, %5, %15\n\tv_exp_f32 %19, %49\n\tv_add_f32 %6, %6, %16\n\tv_cvt_pk_bf16_f32 %20, %15, %16\n\tv_mfma_f32_32x32x16_bf16 %14, %50, %37, %14\n\tv_exp_f32 %21, %51\n\tv_add_f32 %5, %5, %18\n\tv_exp_f32 %22, %52\n\tv_add_f32 %6, %6, %19\n\tv_cvt_pk_bf16_f32 %23, %18, %19\n\tv_mfma_f32_32x32x16_bf16 %14, %53, %41, %14\n\tv_exp_f32 %24, %54\n\tv_add_f32 %5, %5, %21\n\tv_exp_f32 %25, %55\n\tv_add_f32 %6, %6, %22\n\tv_cvt_pk_bf16_f32 %26, %21, %22\n\tv_add_f32 %5, %5, %24\n\tv_add_f32 %6, %6, %25\n\tv_cvt_pk_bf16_f32 %27, %24, %25" : "=&v"(sacc[1][0]), "=&v"(t1_0), "=&v"(t1_1), "=&v"(t1_2), "=&v"(t1_3), "+v"(psum[2][0]), "+v"(psum[2][1]), "=&v"(w2_0), "=&v"(t1_4), "=&v"(t1_5), "=&v"(w2_1), "=&v"(t1_6), "=&v"(t1_7), "=&v"(w2_2), "=&v"(sacc[1][1]), "=&v"(t1_8), "=&v"(t1_9), "=&v"(w2_3), "=&v"(t1_10), "=&v"(t1_11), "=&v"(w3_0), "=&v"(t1_12), "=&v"(t1_13), "=&v"(w3_1), "=&v"(t1_14), "=&v"(t1_15), "=&v"(w3_2), "=&v"(w3_3) : "v"(kf[0][0]), "a"(qf[3][0]), "v"(sacc[0][1][0]), "v"(sacc[0][1][1]), "v"(kf[0][1]), "a"(qf[3][1]), "v"(sacc[0][1][2]), "v"(sacc[0][1][3]), "v"(kf[0][2]), "a"(qf[3][2]), "v"(sacc[0][1][4]), "v"(sacc[0][1][5]), "v"(kf[0][3]), "a"(qf[3][3]), "v"(sacc[0][1][6]), "v"(sacc[0][1][7]), "v"(kf[1][0]), "v"(sacc[0][1][8]), "v"(sacc[0][1][9]), "v"(kf[1][1]), "v"(sacc[0][1][10]), "v"(sacc[0][1][11]), "v"(kf[1][2]), "v"(sacc[0][1][12]), "v"(sacc[0][1][13]), "v"(kf[1][3]), "v"(sacc[0][1][14]), "v"(sacc[0][1][15]) : "memory"); pf[0][2][0] = w2_0; pf[0][2][1] = w2_1; pf[0][2][2] = w2_2; pf[0][2][3] = w2_3; pf[0][3][0] = w3_0; pf[0][3][1] = w3_1; pf[0][3][2] = w3_2; pf[0][3][3] = w3_3; }
            MASKPAD(sacc[1][0], 0);
            { const u32x4 vv0_0 = {vh[0][0][0][0], vh[0][0][0][1], vh[0][0][1][0], vh[0][0][1][1]}; const u32x4 vv1_0 = {vh[1][0][0][0], vh[1][0][0][1], vh[1][0][1][0], vh[1][0][1][1]}; const u32x4 vv0_1 = {vh[0][1][0][0], vh[0][1][0][1], vh[0][1][1][0], vh[0][1][1][1]}; const u32x4 vv1_1 = {vh[1][1][0][0], vh[1][1][0][1], vh[1][1][1][0], vh[1][1][1][1]}; const u32x4 vv0_2 = {vh[0][2][0][0], vh[0][2][0][1], vh[0][2][1][0], vh[0][2][1][1]}; const u32x4 vv1_2 = {vh[1][2][0][0], vh[1][2][0][1], vh[1][2][1][0], vh[1][2][1][1]}; const u32x4 vv0_3 = {vh[0][3][0][0], vh[0][3][0][1], vh[0][3][1][0], vh[0][3][1][1]}; const u32x4 vv1_3 = {vh[1][3][0][0], vh[1][3][0][1], vh[1][3][1][0], vh[1][3][1][1]}; const uint32_t dma_dst0 = dma_dst + 0 + 0 * TILE_BYTES; const uint32_t dma_dst1 = dma_dst + 1024 + 0 * TILE_BYTES; const uint32_t dma_dst2 = dma_dst + 0 + 1 * TILE_BYTES; const uint32_t dma_dst3 = dma_dst + 1024 + 1 * TILE_BYTES; float t0_0; float t0_1; float t0_2; float t0_3; float t0_4; float t0_5; float t0_6; float t0_7; float t0_8; float t0_9; float t0_10; float t0_11; float t0_12; float t0_13; float t0_14; float t0_15; uint32_t w0_0; uint32_t w0_1; uint32_t w0_2; uint32_t w0_3; uint32_t w1_0; uint32_t w1_1; uint32_t w1_2; uint32_t w1_3; asm volatile("v_mfma_f32_32x32x16_bf16 %0, %36, %37, %0\n\ts_waitcnt vmcnt(4)\n\ts_barrier\n\tds_read_b128 %1, %38 offset:0\n\tds_read_b128 %2, %39 offset:0\n\tv_exp_f32 %3, %40\n\tv_exp_f32 %4, %41\n\tv_mfma_f32_32x32x16_bf16 %5, %42, %37, %5\n\tds_read_b128 %6, %43 offset:0\n\tds_read_b128 %7, %44 offset:0\n\tv_exp_f32 %8, %45\n\tv_add_f32 %10, %10, %3\n\tv_exp_f32 %9, %46\n\tv_add_f32 %11, %11, %4\n\tv_cvt_pk_bf16_f32 %12, %3, %4\n\tv_mfma_f32_32x32x16_bf16 %0, %47, %48, %0\n\tds_read_b128 %13, %38 offset:4096\n\tds_read_b128 %14, %39 offset:4096\n\tv_exp_f32 %15, %49\n\tv_add_f32 %10, %10, %8\n\tv_exp_f32 %16, %50\n\tv_add_f32 %11, %11, %9\n\tv_cvt_pk_bf16_f32 %17, %8, %9\n\tv_mfma_f32_32x32x16_bf16 %5, %51, %48, %5\n\tds_read_b128 %18, %43 offset:4096\n\tds_read_b128 %19, %44 offset:4096\n\tv_exp_f32 %20, %52\n\tv_add_f32 %10, %10, %15\n\tv_exp_f32 %21, %53\n\tv_add_f32 %11, %11, %16\n\tv_cvt_pk_bf16_f32 %22, %15, %16\n\tv_mfma_f32_32x32x16_bf16 %0, %54, %55, %0\n\ts_mov_b32 m0, %56\n\ts_nop 0\n\tbuffer_load_dwordx4 %57, %58, %59 offen lds\n\tv_exp_f32 %23, %60\n\tv_add_f32 %10, %10, %20\n\tv_exp_f32 %24, %61\n\tv_add_f32 %11, %11, %21\n\tv_cvt_pk_bf16_f32 %25, %20, %21\n\tv_mfma_f32_32x32x16_bf16 %5, %62, %55, %5\n\ts_mov_b32 m0, %63\n\ts_nop 0\n\tbuffer_load_dwordx4 %64, %58, %59 offen lds\n\tv_exp_f32 %26, %65\n\tv_add_f32 %10, %10, %23\n\tv_exp_f32 %27, %66\n\tv_add_f32 %11, %11, %24\n\tv_cvt_pk_bf16_f32 %28, %23, %24\n\tv_mfma_f32_32x32x16_bf16 %0, %67, %68, %0\n\ts_mov_b32 m0, %69\n\ts_nop 0\n\tbuffer_load_dwordx4 %70, %71, %72 offen lds\n\tv_exp_f32 %29, %73\n\tv_add_f32 %10, %10, %26\n\tv_exp_f32 %30, %74\n\tv_add_f32 %11, %11, %27\n\tv_cvt_pk_bf16_f32 %31, %26, %27\n\tv_mfma_f32_32x32x16_bf16 %5, %75, %68, %5\n\ts_mov_b32 m0, %76\n\ts_nop 0\n\tbuffer_load_dwordx4 %77, %71, %72 offen lds\n\tv_exp_f32 %32, %78\n\tv_add_f32 %10, %10, %29\n\tv_exp_f32 %33, %79\n\tv_add_f32 %11, %11, %30\n\tv_cvt_pk_bf16_f32 %34, %29, %30\n\tv_add_f32 %10, %10, %32\n\tv_add_f32 %11, %11, %33\n\tv_cvt_pk_bf16_f32 %35, %32, %33\n\ts_waitcnt lgkmcnt(0)" : "+a"(oacc[2][0]), "=&v"(kf[0][0]), "=&v"(kf[0][1]), "=&v"(t0_0), "=&v"(t0_1), "+a"(oacc[2][1]), "=&v"(kf[0][2]), "=&v"(kf[0][3]), "=&v"(t0_2), "=&v"(t0_3), "+v"(psum[3][0]), "+v"(psum[3][1]), "=&v"(w0_0), "=&v"(kf[1][0]), "=&v"(kf[1][1]), "=&v"(t0_4), "=&v"(t0_5), "=&v"(w0_1), "=&v"(kf[1][2]), "=&v"(kf[1][3]), "=&v"(t0_6), "=&v"(t0_7), "=&v"(w0_2), "=&v"(t0_8), "=&v"(t0_9), "=&v"(w0_3), "=&v"(t0_10), "=&v"(t0_11), "=&v"(w1_0), "=&v"(t0_12), "=&v"(t0_13), "=&v"(w1_1), "=&v"(t0_14), "=&v"(t0_15), "=&v"(w1_2), "=&v"(w1_3) : "v"(vv0_0), "v"(pf[0][0]), "v"(kaddr[0]), "v"(kaddr[1]), "v"(sacc[1][0][0]), "v"(sacc[1][0][1]), "v"(vv1_0), "v"(kaddr[2]), "v"(kaddr[3]), "v"(sacc[1][0][2]), "v"(sacc[1][0][3]), "v"(vv0_1), "v"(pf[0][1]), "v"(sacc[1][0][4]), "v"(sacc[1][0][5]), "v"(vv1_1), "v"(sacc[1][0][6]), "v"(sacc[1][0][7]), "v"(vv0_2), "v"(pf[0][2]), "s"(dma_dst0), "v"(dvo[0]), "s"(rsK), "s"(soffK), "v"(sacc[1][0][8]), "v"(sacc[1][0][9]), "v"(vv1_2), "s"(dma_dst1), "v"(dvo[1]), "v"(sacc[1][0][10]), "v"(sacc[1][0][11]), "v"(vv0_3), "v"(pf[0][3]), "s"(dma_dst2), "v"(dvo[2]), "s"(rsV), "s"(soffV), "v"(sacc[1][0][12]), "v"(sacc[1][0][13]), "v"(vv1_3), "s"(dma_dst3), "v"(dvo[3]), "v"(sacc[1][0][14]), "v"(sacc[1][0][15]) : "memory"); pf[1][0][0] = w0_0; pf[1][0][1] = w0_1; pf[1][0][2] = w0_2; pf[1][0][3] = w0_3; pf[1][1][0] = w1_0; pf[1][1][1] = w1_1; pf[1][1][2] = w1_2; pf[1][1][3] = w1_3; }
            MASKPAD(sacc[1][1], 32);
        } else if constexpr (VAR == 'T') {
            // period 4: rest of block 3's softmax | PV_3
            { float t1_0; float t1_1; float t1_2; float t1_3; float t1_4; float t1_5; float t1_6; float t1_7; float t1_8; float t1_9; float t1_10; float t1_11; float t1_12; float t1_13; float t1_14; float t1_15; uint32_t w2_0; uint32_t w2_1; uint32_t w2_2; uint32_t w2_3; uint32_t w3_0; uint32_t w3_1; uint32_t w3_2; uint32_t w3_3; asm volatile("v_exp_f32 %0, %26\n\tv_exp_f32 %1, %27\n\tv_exp_f32 %2, %28\n\tv_add_f32 %4, %4, %0\n\tv_exp_f32 %3, %29\n\tv_add_f32 %5, %5, %1\n\tv_cvt_pk_bf16_f32 %6, %0, %1\n\tv_exp_f32 %7, %30\n\tv_add_f32 %4, %4, %2\n\tv_exp_f32 %8, %31\n\tv_add_f32 %5, %5, %3\n\tv_cvt_pk_bf16_f32 %9, %2, %3\n\tv_exp_f32 %10, %32\n\tv_add_f32 %4, %4, %7\n\tv_exp_f32 %11, %33\n\tv_add_f32 %5, %5, %8\n\tv_cvt_pk_bf16_f32 %12, %7, %8\n\tv_exp_f32 %13, %34\n\tv_add_f32 %4, %4, %10\n\tv_exp_f32 %14, %35\n\tv_add_f32 %5, %5, %11\n\tv_cvt_pk_bf16_f32 %15, %10, %11\n\tv_exp_f32 %16, %36\n\tv_add_f32 %4, %4, %13\n\tv_exp_f32 %17, %37\n\tv_add_f32 %5, %5, %14\n\tv_cvt_pk_bf16_f32 %18, %13, %14\n\tv_exp_f32 %19, %38\n\tv_add_f32 %4, %4, %16\n\tv_exp_f32 %20, %39\n\tv_add_f32 %5, %5, %17\n\tv_cvt_pk_bf16_f32 %21, %16, %17\n\tv_exp_f32 %22, %40\n\tv_add_f32 %4, %4, %19\n\tv_exp_f32 %23, %41\n\tv_add_f32 %5, %5, %20\n\tv_cvt_pk_bf16_f32 %24, %19, %20\n\tv_add_f32 %4, %4, %22\n\tv_add_f32 %5, %5, %23\n\tv_cvt_pk_bf16_f32 %25, %22, %23" : "=&v"(t1_0), "=&v"(t1_1), "=&v"(t1_2), "=&v"(t1_3), "+v"(psum[3][0]), "+v"(psum[3][1]), "=&v"(w2_0), "=&v"(t1_4), "=&v"(t1_5), "=&v"(w2_1), "=&v"(t1_6), "=&v"(t1_7), "=&v"(w2_2), "=&v"(t1_8), "=&v"(t1_9), "=&v"(w2_3), "=&v"(t1_10), "=&v"(t1_11), "=&v"(w3_0), "=&v"(t1_12), "=&v"(t1_13), "=&v"(w3_1), "=&v"(t1_14), "=&v"(t1_15), "=&v"(w3_2), "=&v"(w3_3) : "v"(sacc[1][1][0]), "v"(sacc[1][1][1]), "v"(sacc[1][1][2]), "v"(sacc[1][1][3]), "v"(sacc[1][1][4]), "v"(sacc[1][1][5]), "v"(sacc[1][1][6]), "v"(sacc[1][1][7]), "v"(sacc[1][1][8]), "v"(sacc[1][1][9]), "v"(sacc[1][1][10]), "v"(sacc[1][1][11]), "v"(sacc[1][1][12]), "v"(sacc[1][1][13]), "v"(sacc[1][1][14]), "v"(sacc[1][1][15]) : "memory"); pf[1][2][0] = w2_0; pf[1][2][1] = w2_1; pf[1][2][2] = w2_2; pf[1][2][3] = w2_3; pf[1][3][0] = w3_0; pf[1][3][1] = w3_1; pf[1][3][2] = w3_2; pf[1][3][3] = w3_3; }
            { const u32x4 vv0_0 = {vh[0][0][0][0], vh[0][0][0][1], vh[0][0][1][0], vh[0][0][1][1]}; const u32x4 vv1_0 = {vh[1][0][0][0], vh[1][0][0][1], vh[1][0][1][0], vh[1][0][1][1]}; const u32x4 vv0_1 = {vh[0][1][0][0], vh[0][1][0][1], vh[0][1][1][0], vh[0][1][1][1]}; const u32x4 vv1_1 = {vh[1][1][0][0], vh[1][1][0][1], vh[1][1][1][0], vh[1][1][1][1]}; const u32x4 vv0_2 = {vh[0][2][0][0], vh[0][2][0][1], vh[0][2][1][0], vh[0][2][1][1]}; const u32x4 vv1_2 = {vh[1][2][0][0], vh[1][2][0][1], vh[1][2][1][0], vh[1][2][1][1]}; const u32x4 vv0_3 = {vh[0][3][0][0], vh[0][3][0][1], vh[0][3][1][0], vh[0][3][1][1]}; const u32x4 vv1_3 = {vh[1][3][0][0], vh[1][3][0][1], vh[1][3][1][0], vh[1][3][1][1]}; asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %3, %1\n\tv_mfma_f32_32x32x16_bf16 %0, %5, %6, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %7, %6, %1\n\tv_mfma_f32_32x32x16_bf16 %0, %8, %9, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %10, %9, %1\n\tv_mfma_f32_32x32x16_bf16 %0, %11, %12, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %13, %12, %1" : "+a"(oacc[3][0]), "+a"(oacc[3][1]) : "v"(vv0_0), "v"(pf[1][0]), "v"(vv1_0), "v"(vv0_1), "v"(pf[1][1]), "v"(vv1_1), "v"(vv0_2), "v"(pf[1][2]), "v"(vv1_2), "v"(vv0_3), "v"(pf[1][3]), "v"(vv1_3) : "memory"); }
        }
        // GENERATED-END
    };

    bool first_item = true;
    for (;;) {
        if (SK) {
            int item;
            if (sk_round < sk_rfull) {
                item = sk_round * sk_ncu + sk_idx;
                ++sk_round;
                it.tb = 0; it.nt = nt_all; it.role = 0; it.local = 0;
            } else if (sk_idx < sk_rem) {                       // main: key tiles [0, a) of leftover item sk_idx
                if (sk_main_done) break;
                sk_main_done = true;
                it.local = sk_idx;
                item = sk_rfull * sk_ncu + it.local;
                it.tb = 0; it.nt = sk_a; it.role = sk_suf ? 2 : 0; it.slot = 0;
            } else {                                            // helper: the next piece of its suffix steps
                if (sk_s >= sk_end) break;
                it.local = sk_s / sk_suf;
                item = sk_rfull * sk_ncu + it.local;
                const int off = sk_s - it.local * sk_suf, left = sk_end - sk_s;
                it.tb = sk_a + off;
                it.nt = sk_suf - off < left ? sk_suf - off : left;
                it.role = 1;
                it.slot = sk_xcd * (2 * sk_ncu) + (sk_idx - sk_rem) + it.local;      // pieces in step order: helper h + item index
                sk_s += it.nt;
            }
            it.nt_all = nt_all;
            if (nbh % 8 == 0) {
                it.bh = (item / p.nqt) * 8 + sk_xcd;
                it.qt = item % p.nqt;
            } else {
                it.bh = (sk_base + item) / p.nqt;
                it.qt = (sk_base + item) % p.nqt;
            }
            // every wave is done with the previous item's K/V stages before the next item's tiles land in them
            if (!first_item) asm volatile("s_barrier" ::: "memory");
            first_item = false;
        }
        if (p.bound_dev) {
            // data-dependent bound of this (batch, head): |q . k| <= max ||q|| max ||k|| (squared norms from bya_qknorm_rope,
            // one partial table per slot).  Every workgroup of the head takes the same decision and writes the same flag.
            const float* tq = p.bound_dev + (long long)lane * 2 * p.bound_heads + p.bound_bh0 + it.bh;
            float q2 = lane < p.bound_slots ? tq[0] : 0.f, k2 = lane < p.bound_slots ? tq[p.bound_heads] : 0.f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                q2 = fmaxf(q2, __shfl_xor(q2, o, 64));
                k2 = fmaxf(k2, __shfl_xor(k2, o, 64));
            }
            const bool skip = !(sqrtf(q2) * sqrtf(k2) <= p.bound_limit);         // (NaN statistics: leave the head to the fallback)
            if (tid == 0) p.fallback[it.bh] = skip ? 1 : 0;
            if (skip) {
                if (SK) continue;
                return;
            }
        }
        const int head = it.bh % p.heads, b12 = it.bh / p.heads;
        const int b1 = b12 / p.nb2, b2 = b12 % p.nb2;
        const bf16_t* Qp = p.q + b1 * p.q_s1 + b2 * p.q_s2 + (long long)head * D;
        const bf16_t* Kp = p.k + b1 * p.k_s1 + b2 * p.k_s2 + (long long)head * D + (long long)it.tb * KV_TILE * p.k_row;
        const bf16_t* Vp = p.v + b1 * p.v_s1 + b2 * p.v_s2 + (long long)head * D + (long long)it.tb * KV_TILE * p.v_row;
        bf16_t* Op = p.o + b1 * p.o_s1 + b2 * p.o_s2 + (long long)head * D;
        const int skv_left = p.Skv - it.tb * KV_TILE;             // keys from this piece's first tile to the end of K / V
        rsK = raw_rsrc(Kp, (uint32_t)(((long long)(skv_left - 1) * p.k_row + D) * 2));
        rsV = raw_rsrc(Vp, (uint32_t)(((long long)(skv_left - 1) * p.v_row + D) * 2));
        const int ntiles = it.nt;
        stage_tile(0, 0);
        stage_tile(1, 1);
        stage_tile(2, 2);

        const int q0 = it.qt * ROWS_PER_WG + wave * (QB * 32);
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            int qrow = q0 + b * 32 + r;
            q_valid[b] = qrow < p.Sq;
            qrow = q_valid[b] ? qrow : p.Sq - 1;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                qf[b][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Qp + (long long)qrow * p.q_row + s * 16 + hf * 8));
        }
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            psum[b][0] = psum[b][1] = 0.f;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[b][d][i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) { sacc[0][0][i] = 0.f; sacc[0][1][i] = 0.f; sacc[1][0][i] = 0.f; sacc[1][1][i] = -INFINITY; }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                pf[x][ks] = u32x4{0u, 0u, 0u, 0u};
                vh[x][ks][0] = u32x2{0u, 0u};
                vh[x][ks][1] = u32x2{0u, 0u};
            }

        // The Q loads are hipcc's own: make it wait for them HERE (an empty asm that takes every fragment as an AGPR operand),
        // or it parks one s_waitcnt vmcnt(N) in front of each fragment's first MFMA inside the loop -- down to vmcnt(0), which
        // would drain the K/V prefetch every tile.  (That drains tiles 0..2 as well: the prologue's wait below is then free.)
#pragma unroll
        for (int b = 0; b < QB; ++b)
#pragma unroll
            for (int s = 0; s < 4; ++s) asm volatile("" : "+a"(qf[b][s]));
        // tile 0 has landed once all but the 8 younger pieces (tiles 1, 2) have; then its K fragments
        VMC(8);
        BAR();
        {
            uint32_t kaddr[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) kaddr[s] = kofs[s];
            RK(0, 0); RK(0, 1); RK(0, 2); RK(0, 3); RK(1, 0); RK(1, 1); RK(1, 2); RK(1, 3);
            LGKM(0);
        }
        int st = 0;                                            // ring stage of tile t (= of tile t + 3)
        for (int t = 0; t < ntiles - 1; ++t) {
            const int st1 = st == NST - 1 ? 0 : st + 1;
            uint32_t kaddr[4], vaddr[2];
#pragma unroll
            for (int s = 0; s < 4; ++s) kaddr[s] = kofs[s] + st1 * STAGE_BYTES;
#pragma unroll
            for (int d = 0; d < 2; ++d) vaddr[d] = vofs[d] + st * STAGE_BYTES;
            body(IntTagC<'L'>{}, kaddr, vaddr, lds0s + st * STAGE_BYTES, (uint32_t)(t + 3) * k_tile_stride,
                 (uint32_t)(t + 3) * v_tile_stride);
            st = st1;
        }
        {
            // the piece's last tile: keys past Skv (only the piece that holds the sequence's last tile has any) masked
            const int t = ntiles - 1, st1 = st == NST - 1 ? 0 : st + 1;
            keys_last = skv_left - t * KV_TILE;               // >= 64: every key of the tile exists
            uint32_t kaddr[4], vaddr[2];
#pragma unroll
            for (int s = 0; s < 4; ++s) kaddr[s] = kofs[s] + st1 * STAGE_BYTES;
#pragma unroll
            for (int d = 0; d < 2; ++d) vaddr[d] = vofs[d] + st * STAGE_BYTES;
            body(IntTagC<'M'>{}, kaddr, vaddr, lds0s + st * STAGE_BYTES, (uint32_t)(t + 3) * k_tile_stride,
                 (uint32_t)(t + 3) * v_tile_stride);
        }
        {
            const uint32_t none4[4] = {0u, 0u, 0u, 0u}, none2[2] = {0u, 0u};
            body(IntTagC<'T'>{}, none4, none2, 0u, 0u, 0u);
        }
        // the MFMAs are inline asm: pad their last results before compiler code reads them; drain the tail prefetches
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)" ::: "memory");

        if (SK && it.role == 1) {
            // ---- a suffix piece of a leftover item: hand the un-normalised accumulators and row sums to the item's main
            const int my_slot = it.slot;
            float* const slot = p.sk_part + (size_t)my_slot * SK_SLOT_FLOATS;
#pragma unroll
            for (int b = 0; b < QB; ++b) {
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        f32x4 w = {oacc[b][d][gq * 4 + 0], oacc[b][d][gq * 4 + 1], oacc[b][d][gq * 4 + 2], oacc[b][d][gq * 4 + 3]};
                        *reinterpret_cast<f32x4*>(slot + ((size_t)(((wave * QB + b) * 2 + d) * 4 + gq) * 64 + lane) * 4) = w;
                    }
                slot[4 * QB * 2 * 16 * 64 + (wave * QB + b) * 64 + lane] = psum[b][0] + psum[b][1];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            if (tid == 0) __hip_atomic_store(p.sk_flags + my_slot, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        if (SK && it.role == 2) {
            // ---- prefix [0, a) of a leftover item: the helpers of this XCD hold its suffix
            // the suffix [local * suf, (local + 1) * suf) of the helpers' step list is held by every helper whose range
            // [S h / nh, S (h + 1) / nh) overlaps it; a piece's slot is (helper + item index): its ordinal in step order
            const long long S = (long long)sk_rem * sk_suf;
            const int nh = sk_ncu - sk_rem, lo = it.local * sk_suf, hi = lo + sk_suf;
            for (int hb = 0; hb < nh; ++hb) {
                const int b0 = (int)(S * hb / nh), b1 = (int)(S * (hb + 1) / nh);
                if (b1 <= lo || b0 >= hi || b0 == b1) continue;
                const int cs = sk_xcd * (2 * sk_ncu) + hb + it.local;
                if (tid == 0) {
                    int spins = 0;
                    while (__hip_atomic_load(p.sk_flags + cs, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 1u) {
                        __builtin_amdgcn_s_sleep(8);
                        if (++spins > (1 << 22)) {     // ~1 s: never hang the GPU; the event is counted (bya_attn_workspace_status)
                            __hip_atomic_fetch_add(p.sk_flags + 1023, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                }
                asm volatile("s_barrier" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                const float* const slot = p.sk_part + (size_t)cs * SK_SLOT_FLOATS;
#pragma unroll
                for (int b = 0; b < QB; ++b) {
#pragma unroll
                    for (int d = 0; d < 2; ++d)
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const f32x4 w = *reinterpret_cast<const f32x4*>(slot + ((size_t)(((wave * QB + b) * 2 + d) * 4 + gq) * 64 + lane) * 4);
                            oacc[b][d][gq * 4 + 0] += w[0]; oacc[b][d][gq * 4 + 1] += w[1];
                            oacc[b][d][gq * 4 + 2] += w[2]; oacc[b][d][gq * 4 + 3] += w[3];
                        }
                    psum[b][0] += slot[4 * QB * 2 * 16 * 64 + (wave * QB + b) * 64 + lane];
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // one block at a time: 128 loads in flight would spill
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");          // every wave has read the slot: the flag goes back to 0
                if (tid == 0) __hip_atomic_store(p.sk_flags + cs, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }

        // ---- epilogue: O[q][d] = O^T[d][q] / l ; lane (r, hf) holds d = 32 dd + (i & 3) + 8 (i >> 2) + 4 hf
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            const float l_half = psum[b][0] + psum[b][1];
            const auto lsw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_half), __float_as_uint(l_half), false, false);
            const float inv = 1.0f / (__uint_as_float(lsw[0]) + __uint_as_float(lsw[1]));
            bf16_t* orow = Op + (long long)(q_valid[b] ? q0 + b * 32 + r : 0) * p.o_row;
#pragma unroll
            for (int d = 0; d < 2; ++d) store_o_tile(orow + d * 32, oacc[b][d], inv, hf, q_valid[b], p.o_wide != 0);
        }
        if (!SK) break;
    }
}

}  // namespace

namespace {
// stream-K exchange workspace, caller-owned, one per DEVICE (same rules as the GEMM's split-K workspace, gemm.hip)
constexpr int SK_MAX_DEVICES = 64, SK_GRID = 256, SK_SLOTS = 2 * SK_GRID;  // slot = xcd * 64 + helper + leftover item
constexpr long long SK_WS_BYTES = SK_FLAG_BYTES + (long long)SK_SLOTS * SK_SLOT_FLOATS * 4;
std::atomic<void*> g_attn_ws[SK_MAX_DEVICES];
inline int sk_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < SK_MAX_DEVICES ? dev : -1;
}
}  // namespace

extern "C" int bya_set_attn_workspace(void* ws, int64_t bytes) {
    if (ws && (bytes < (int64_t)SK_WS_BYTES || ((uintptr_t)ws & 255))) return BYA_ERR_SHAPE;
    const int dev = sk_device();
    if (dev < 0) return BYA_ERR_UNSUPPORTED;
    g_attn_ws[dev].store(ws);
    return BYA_OK;
}

extern "C" int bya_attn_workspace_bytes(int64_t* bytes) {
    if (!bytes) return BYA_ERR_SHAPE;
    *bytes = (int64_t)SK_WS_BYTES;
    return BYA_OK;
}

extern "C" int bya_attn_workspace_status(int32_t* timeouts, hipStream_t stream) {
    if (!timeouts) return BYA_ERR_SHAPE;
    *timeouts = 0;
    const int dev = sk_device();
    char* const ws = dev < 0 ? nullptr : static_cast<char*>(g_attn_ws[dev].load());
    if (!ws) return BYA_OK;
    unsigned word = 0;
    if (hipMemcpyAsync(&word, ws + 1023 * 4, 4, hipMemcpyDeviceToHost, stream) != hipSuccess) return BYA_ERR_LAUNCH;
    if (hipStreamSynchronize(stream) != hipSuccess) return BYA_ERR_LAUNCH;
    *timeouts = (int32_t)word;
    return BYA_OK;
}

int bya_launch_attn_w4(const void* args, hipStream_t s) {
    AttnArgs a = *static_cast<const AttnArgs*>(args);
    a.nqt = (a.Sq + ROWS_PER_WG - 1) / ROWS_PER_WG;
    const int nbh = a.nb1 * a.nb2 * a.heads;
    const int dev = sk_device();
    char* const ws = dev < 0 ? nullptr : static_cast<char*>(g_attn_ws[dev].load());
    const bool sk_on = bya_opt(BYA_OPT_ATTN_STREAMK) != 0;   // default on when a workspace exists
    const long long nt_all = (a.Skv + KV_TILE - 1) / KV_TILE;
    const long long items = (long long)nbh * a.nqt;
    // stream-K pays when an XCD's items make at least one whole round of its 32 CUs plus a partial one.  Measured
    // (profiles/history/r4_r_attn_streamk_probe.json): +1 % at 48 heads x 17776 (6.56 rounds), +8.5 % at a 2-rank shard's 24 heads
    // (3.28 rounds), +2.7 % at 47026 tokens -- a fraction of what the round counts promise, and a grid that does not fill ONE
    // round (6 heads of an 8-rank shard: 210 items on 256 CUs) LOSES 7 % although every workgroup then has 0.81 items
    // of work.  Two forms of the cut (contiguous step ranges; mains + helpers at one key tile) measure the same, so it is
    // not L2 locality: the kernel is power-limited, the CUs a partial round leaves idle give their
    // power budget to the busy ones as clock, and filling them buys little; below one round the hand-offs cost more.
    const long long ipx = items / 8;                            // per XCD, +- one item when 8 does not divide
    const long long rem = ipx % (SK_GRID / 8);
    const bool sk = ws && sk_on && ipx >= SK_GRID / 8 && (items % SK_GRID != 0) && rem * nt_all >= 8 * (SK_GRID / 8) &&
                    nt_all >= 16 && items * nt_all < (1LL << 31);
    if (sk) {
        a.sk_flags = reinterpret_cast<unsigned*>(ws);
        a.sk_part = reinterpret_cast<float*>(ws + SK_FLAG_BYTES);
        BYA_LAUNCH(attn_joint_w4_kernel<true>, dim3(SK_GRID), dim3(256), (size_t)NST * STAGE_BYTES, s, a);
    } else {
        a.sk_flags = nullptr;
        a.sk_part = nullptr;
        dim3 grid((nbh * a.nqt + 7) / 8 * 8);
        BYA_LAUNCH(attn_joint_w4_kernel<false>, grid, dim3(256), (size_t)NST * STAGE_BYTES, s, a);
    }
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}
