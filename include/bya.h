/*
 * bya.h -- C ABI of libbya_hip.so: the hand-written gfx950 (MI355X / CDNA4) kernels under the
 * Bind-Your-Avatar per-denoise-step hot path.
 *
 * The reference (Yubo-Shankui/Bind-Your-Avatar-Implementation) is pure Python: its hot path is
 * `BindyouravatarTransformer3DModel.forward` (models/transformer.py:615-964), called once per step by
 * `BindyouravatarPipeline.__call__` (models/pipeline_bindyouravatar.py:910-923).  It has no FFI of its
 * own; every op is a PyTorch dispatch.  This header is therefore the boundary a maintainer binds
 * (ctypes stub in INTEGRATION.md) and each entry point cites the reference dispatches it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM); tensors are row-major, bf16 (raw uint16) unless noted;
 *     fp32 for RoPE tables, fp32/int64 where stated.
 *   - functions only enqueue work on `stream`; they never allocate, never synchronise, never throw.
 *   - return value: 0 = ok, <0 = error (BYA_ERR_*): nothing was launched.
 *   - "rows" of the joint sequence are [text rows (Tt=226) | video rows (N = T*Ht*Wt)], S = Tt + N.
 */
#ifndef BYA_H
#define BYA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the declarations below are its whole dynamic symbol table. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

enum { BYA_ACT_NONE = 0, BYA_ACT_GELU_TANH = 1, BYA_ACT_GELU_ERF = 2, BYA_ACT_RELU = 3, BYA_ACT_SILU = 4,
       BYA_ACT_LEAKY_RELU = 5,
       BYA_ACT_GELU_TANH_IEEE = 6 /* bya_gemm_bf16 only: GELU(tanh) through expf + IEEE division (A/B reference) */ };

/* library / build identification: returns the ABI version (bumped when a signature changes). */
int bya_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Process-wide options (round 6).  The entry points below are functions of their arguments and of THIS table only: the
 * library never reads the environment.  The host sets what it wants once, after loading the library (the Python module
 * maps its BYA_* environment variables onto these keys at import, ops.apply_env_options); a launch reads the table when
 * it is prepared, so a change applies to the launches enqueued after it.  bya_set_option returns BYA_ERR_SHAPE for an
 * unknown key or a value outside the key's range; bya_get_option writes the current value.
 * --------------------------------------------------------------------------------------------- */
enum bya_option {
    BYA_OPT_GEMM_SPLITK = 0,      /* 0 (default since the end of round 6): bya_gemm_bf16 never cuts a tile along K -- the rows
                                     behind the last full round of 256-row tiles run on 128-row tiles instead, as fast and without
                                     a hand-off (a shard's rows round like the whole's); 1: the tiles of a partial last round are
                                     split along K when a workspace is registered; 2: every split tile in two */
    BYA_OPT_GEMM_SPLITK_MIN = 1,  /* shortest K range (in 64-wide K tiles) a split may produce; 0 = the built-in default */
    BYA_OPT_GEMM_TILE = 2,        /* -1 (default): tile shape by the cost model; 0..6: force one (tests: every shape through
                                     every kernel) */
    BYA_OPT_GEMM_VARIANT = 3,     /* 0 (default): the one-wave-per-SIMD kernels (256 x 256 tiles, 128 x 256 where that fills the CUs
                                     better) where eligible; 1: the 8-wave kernel; 2: 256 x 256 only (A/B of the 128-row tile) */
    BYA_OPT_ATTN_STREAMK = 4,     /* 1 (default): the joint attention cuts the items of a partial last round between
                                     workgroups when a workspace is registered; 0: one workgroup per item */
    BYA_OPT_FP8_KERNEL = 5,       /* 0 (default): 256 x 256 fp8 kernel where a launch fills it; 1: always the 128 x 128 one */
    BYA_OPT_P2P_GROUPS = 6,       /* workgroups of a push / exchange launch (16..1024); 0 = the built-in default */
    BYA_OPT_REFERENCE_FORMS = 7,  /* bit mask, tests only: take the OLDER kernel form of an A/B the docs call closed; every
                                     form pair is bit-identical, which is what the tests that set these bits assert */
    BYA_OPT_COUNT = 8
};
enum { BYA_REF_ROWGEMM_CHUNKED = 1,    /* N = 512 row GEMM: chunk-balanced kernel instead of the W-stationary one */
       BYA_REF_KV_MIX_GENERIC = 2,     /* bya_attn_kv_mix: the generic kernel instead of the <= 32-key one */
       BYA_REF_LN_GENERIC = 4,         /* bya_layernorm at 3072 columns: one row per wave instead of parameters in registers */
       BYA_REF_ROUTER_SCORES_WAVE = 8, /* bya_router_scores: keys per wave instead of keys in LDS */
       BYA_REF_ATTN_NARROW_STORE = 16  /* attention epilogues: 8-byte instead of 16-byte stores */ };
int bya_set_option(int32_t key, int32_t value);
int bya_get_option(int32_t key, int32_t* value);

/* Board calibration (diagnostic; csrc/calib.hip): 256 workgroups x 4 waves run `iters` sweeps of 64 v_mfma_f32_16x16x32_bf16 each
 * (a 128 x 128 x 32 wave tile: 2^20 FLOP per wave and sweep, BYA_CALIBRATION_FLOP_PER_ITER per launch and iteration) on the
 * operand fragments at `operands` (>= BYA_CALIBRATION_OPERAND_BYTES of bf16: gaussian for the rate the board sustains on real
 * data, zeros for its full clock).  The caller times the launch (HIP events) -- TFLOP/s = iters * BYA_CALIBRATION_FLOP_PER_ITER /
 * seconds / 1e12.  `sink`: 4 bytes of device memory (never written in practice).  No reference counterpart. */
#define BYA_CALIBRATION_OPERAND_BYTES (256LL * 256 * 16 * 16)
#define BYA_CALIBRATION_FLOP_PER_ITER (256.0 * 4 * 2.0 * 128 * 128 * 32)
int bya_mfma_calibration(const void* operands, int64_t operand_bytes, void* sink, int32_t iters, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * GEMM:  C[z][m,n] = res[z][m,n] + gate[z][row-type(m)][n] * alpha * act( sum_k A[z][m,k] * W[n,k] + rowscale[m]*bias[n] )
 * Replaces every nn.Linear / 2x2-stride Conv2d-as-GEMM on the path: attn1.to_q/k/v/to_out,
 * ff.net.0.proj/ff.net.2 (models/transformer.py:241-260), perceiver to_q/to_out (models/router.py:253,275),
 * router to_q/to_k and the SpatialTemporalAttentionBlock projections + mlp (models/router.py:381-383,
 * 468-491), audio attn to_q/to_out (models/audio_model.py:253-256), patch_embed.proj/text_proj,
 * proj_out (models/transformer.py:690,949), and the step-invariant LocalFacialExtractor /
 * AudioProjModel linears (models/router.py:157-193, models/audio_model.py:78-114).
 * bias/res/gate0 may be NULL.  gate1==NULL means gate0 for all rows; rows m < gate_split use gate0.
 * Requirements: K % 64 == 0, N % 4 == 0, lda/ldw % 8 == 0, A/W 16-byte aligned.
 * --------------------------------------------------------------------------------------------- */
typedef struct bya_gemm_desc {
    int32_t M, N, K;          /* per batch entry */
    int32_t batch;            /* grid.z; A/C/res/gate advance by the *_batch_stride (elements) */
    int32_t lda, ldw, ldc, ldres;
    int64_t a_batch_stride, c_batch_stride, res_batch_stride, gate_batch_stride;
    int32_t gate_split;
    int32_t act;              /* BYA_ACT_* applied to (acc + bias) */
    int32_t n_split;          /* > 0: column n is written to C + (n / n_split) * c_split_stride, column n % n_split
                                 (one launch for the packed q|k|v projection writing three separate tensors) */
    int64_t c_split_stride;
    const float* bias_rowscale; /* optional fp32 [batch*M]: bias[n] is multiplied by bias_rowscale[z*M + m] */
    float alpha;              /* scales act(acc + bias); 0 is read as 1 */
} bya_gemm_desc;

int bya_gemm_bf16(const void* A, const void* W, const void* bias, void* C, const void* res,
                  const void* gate0, const void* gate1, const bya_gemm_desc* desc, hipStream_t stream);

/* The packed q|k|v projection of a DiT block WITH the per-head q/k LayerNorm(64, eps, affine) + interleaved-pair RoPE of
 * bya_qknorm_rope in its epilogue (reference models/transformer.py:200-209,241-245 = diffusers Attention.to_q/k/v followed by
 * CogVideoXAttnProcessor2_0's norm_q / norm_k / apply_rotary_emb): columns [0, width) of the product are q, [width, 2 width) k,
 * [2 width, 3 width) v (plain bias epilogue); desc->n_split / c_split_stride place them as for bya_gemm_bf16.  q and k are
 * normalised from the bf16-ROUNDED projection and with the arithmetic of csrc/qknorm_math.h, so the result equals
 * bya_gemm_bf16 followed by bya_qknorm_rope BIT FOR BIT -- q and k are just written once instead of written, read and written
 * again.  rows >= text_rows are rotated with cos / sin [rows - text_rows, 64] fp32 (row index = row of the launch, per batch
 * entry); k_scale as for bya_qknorm_rope.  N = 3 width (q | k | v) or, since round 6, N = 2 width (q | k alone: the sharded step
 * computes v first and pushes it to the peers underneath this launch).  Constraints: no activation / residual / gates, width % 128 == 0, the persistent
 * kernel's alignment rules; BYA_ERR_UNSUPPORTED otherwise (the caller then issues the two launches). */
typedef struct bya_qknorm_desc {
    const void* qw; const void* qb; const void* kw; const void* kb;   /* bf16 [64] each */
    const float* cos; const float* sin;
    int32_t text_rows;
    int32_t width;                                                      /* heads * 64 */
    float eps, k_scale;
} bya_qknorm_desc;
int bya_gemm_qkv_norm_rope(const void* A, const void* W, const void* bias, void* C, const bya_gemm_desc* desc,
                           const bya_qknorm_desc* norm, hipStream_t stream);

/* The same product for SKINNY launches -- at most 64 rows, N <= 8192, N % 16 == 0, K % 32 == 0, K >= 256, no gates, no
 * bias_rowscale, no n_split (anything else: BYA_ERR_UNSUPPORTED) -- on a weight-streaming kernel: one workgroup per 16
 * output columns (batch elements stacked as rows while they fit 64 together), 16 waves split K, partial sums added in a
 * fixed order.  For Linears whose few rows meet a wide weight (the step-invariant conditioning: nn.Linear call sites of
 * models/router.py:216-262 (Perceiver to_kv / LocalFacialExtractor) and models/audio_model.py:78-114 (AudioProjModel)): the
 * tiled kernels give such a launch N / 128 workgroups.  It is a separate entry point because its fp32 summation order
 * differs from bya_gemm_bf16's: which kernel runs is the caller's decision, never a function of the row count. */
int bya_gemm_skinny_bf16(const void* A, const void* W, const void* bias, void* C, const void* res,
                         const bya_gemm_desc* desc, hipStream_t stream);

/* Optional split-K workspace of the persistent GEMM kernel (device memory owned by the caller, 256-byte aligned, at
 * least the size bya_gemm_workspace_bytes reports, ZERO-FILLED once): with it, the last partial round of 256 x 256 output tiles
 * of a bya_gemm_bf16 launch is cut along K over the idle CUs (partial sums and completion counters live here).  One
 * workspace per DEVICE (registered for, and used by launches enqueued under, the current device); launches that use it
 * must be ordered on one stream.  NULL unregisters.  Results do not depend on it beyond fp32 summation order.  (No
 * reference counterpart: scheduling detail of the Linear layers.) */
int bya_set_gemm_workspace(void* ws, int64_t bytes);
int bya_gemm_workspace_bytes(int64_t* bytes);
/* Health of the current device's workspace: *timeouts = number of split tiles whose finisher gave up waiting (~1 s) for a
 * partial sum and finished without it -- 0 on a healthy run; anything else means outputs of that launch were wrong.
 * Completion counters carry the launch epoch, so a late writer of such a launch cannot corrupt later launches.  The ONE
 * entry point that synchronises (it copies a word back over `stream`): call it at step or run end, not per launch. */
int bya_gemm_workspace_status(int32_t* timeouts, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * fp8 weights (BASELINE configs[4]; no reference counterpart: the reference runs bf16/fp16 only).
 * OCP e4m3fn bytes, symmetric per-row scales:  x[m,k] ~= scale[m] * fp8(q[m,k]).
 *
 * bya_quantize_rows_fp8:  scale[m] = max_k |x[m,k]| / 448 (1 for an all-zero row),
 *                         q[m,k]  = e4m3( x[m,k] * inv ),  inv = fp32(448 / max_k |x[m,k]|) correctly rounded, the product
 *                         in fp32, one round-to-nearest-even to e4m3 (torch.float8_e4m3fn's converter, byte for byte).
 *   Used once per weight at load time (rows = output channels) and on the fly for the activations of each fp8 GEMM.
 *   x bf16 [M, K] (row stride ldx), q uint8 [M, K] (row stride ldq), scale fp32 [M].  K % 8 == 0, K <= 12288.
 *
 * bya_gemm_fp8:  C[z][m,n] = res + gate * alpha * act( a_scale[z*M+m] * w_scale[n] * sum_k A8[z][m,k] * W8[n,k] + rowscale*bias[n] )
 *   on v_mfma_scale_f32_16x16x128_f8f6f4 (fp32 accumulation); same descriptor and epilogue as bya_gemm_bf16, with lda / ldw /
 *   a_batch_stride counted in fp8 elements (bytes).  Replaces attn1.to_q|k|v, attn1.to_out, ff.net.0.proj and ff.net.2 of a
 *   CogVideoXBlock (models/transformer.py:241-260) when the engine is built with fp8 weights.
 *   Requirements: K % 128 == 0, N % 4 == 0, lda/ldw % 16 == 0, A8/W8/w_scale 16-byte aligned; act in {NONE, GELU_TANH(_IEEE)}.
 * --------------------------------------------------------------------------------------------- */
int bya_quantize_rows_fp8(const void* x, void* q, float* scale, int32_t M, int32_t K, int64_t ldx, int64_t ldq,
                          hipStream_t stream);
int bya_gemm_fp8(const void* A8, const float* a_scale, const void* W8, const float* w_scale, const void* bias, void* C,
                 const void* res, const void* gate0, const void* gate1, const bya_gemm_desc* desc, hipStream_t stream);
/* bya_layernorm_fp8: bya_layernorm (same arguments, same arithmetic) followed by bya_quantize_rows_fp8 of each output row,
 * in one pass: the normalised + modulated row is rounded to bf16 exactly as bya_layernorm would store it, then quantised --
 * byte for byte what the two launches produce, without the bf16 round trip.  q uint8 [batch][rows][D] (row stride ldq,
 * batch stride q_batch_stride, bytes), q_scale fp32 [batch * rows_per_batch].  D = 3072 only (the AdaLN LayerNorms of
 * CogVideoXBlock, models/transformer.py:233,251, in front of the q|k|v and MLP Linears). */
int bya_layernorm_fp8(const void* x, void* q, float* q_scale, const void* w, const void* b, const void* shift0,
                      const void* scale0, const void* shift1, const void* scale1, int64_t rows_per_batch, int32_t batch,
                      int32_t D, int64_t ldx, int64_t ldq, int64_t x_batch_stride, int64_t q_batch_stride,
                      int64_t mod_batch_stride, int64_t split, float eps, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Small-M linear (M <= 8 rows):  out[m,n] = sum_k f(x[m,k]) * W[n,k] + bias[n],  f = identity or SiLU.
 * HBM-bound weight stream.  Replaces TimestepEmbedding.linear_1/2, CogVideoXLayerNormZero.linear
 * (models/transformer.py:198,212 -- all 2*num_layers of them in ONE launch over packed weights) and
 * AdaLayerNorm.linear (models/transformer.py:420-426).  act_out applied after bias.
 * --------------------------------------------------------------------------------------------- */
int bya_linear_small_m(const void* x, const void* W, const void* bias, void* out, int32_t M, int32_t N,
                       int32_t K, int32_t silu_in, int32_t act_out, hipStream_t stream);

/* Sinusoidal timestep features (diffusers Timesteps, flip_sin_to_cos, shift 0) in fp32 then rounded
 * to bf16: out[b, 0:dim/2] = cos(t*w_k), out[b, dim/2:] = sin(t*w_k)  (models/transformer.py:679-685). */
int bya_timestep_features(const int64_t* timesteps, void* out, int32_t batch, int32_t dim,
                          int32_t flip_sin_to_cos, float freq_shift, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm over the last dim D (fp32 statistics), optional affine (w,b), optional AdaLN modulation
 *   y = (LN(x)*w + b) * (1 + scale[z][type(row)]) + shift[z][type(row)]
 * rows r < split use (shift0,scale0) else (shift1,scale1); mod pointers NULL = plain LayerNorm.
 * Replaces CogVideoXLayerNormZero / AdaLayerNorm / nn.LayerNorm call sites
 * (models/transformer.py:233,251,944,948; models/router.py:247-248,380-393,475-491;
 * models/audio_model.py:249).  D in {512, 768, 1024, 2048, 3072}; rows_per_batch rows per z.
 * Affine + modulation are evaluated as ONE fma, y = fma(LN(x), w (1 + scale), b (1 + scale) + shift), in every kernel
 * behind this entry point (at D = 3072 with w and b a wave keeps those two vectors in registers and walks several rows),
 * so a result does not depend on how the rows are cut into launches.
 * --------------------------------------------------------------------------------------------- */
int bya_layernorm(const void* x, void* y, const void* w, const void* b, const void* shift0, const void* scale0,
                  const void* shift1, const void* scale1, int64_t rows_per_batch, int32_t batch, int32_t D,
                  int64_t ldx, int64_t ldy, int64_t x_batch_stride, int64_t y_batch_stride,
                  int64_t mod_batch_stride, int64_t split, float eps, hipStream_t stream);

/* Per-head LayerNorm(64, eps, affine) on q and k followed by interleaved-pair RoPE on rows >= text_rows
 * (diffusers CogVideoXAttnProcessor2_0 + apply_rotary_emb; models/transformer.py:204-208).  In place.
 * q,k: [batch, S, heads*64] with row stride ld; cos,sin: fp32 [S - text_rows, 64].
 * k_scale (0 or 1 = off): the finished k is multiplied by it in fp32 before its single rounding to bf16 -- the engine
 * folds softmax_scale*log2(e) into k here so that bya_attn_fwd can run with scores_prescaled = 1.
 * q or k (not both) may be NULL: only the other tensor is processed (the sharded step norms q, starts q's exchange on
 * the RCCL stream and norms k underneath it).
 * stats (may be NULL): fp32 [stats_slots][2][batch * heads], zero-filled by the caller before the launch.  The kernel
 * raises (atomic maximum on the bit pattern) entry [blockIdx % stats_slots][0 = q, 1 = k][z * heads + head] to the squared
 * Euclidean norm of every finished, bf16-rounded head row it writes: max over the slots = max ||q||^2 / max ||k||^2 per
 * (batch, head) -- the data-dependent score bound  |q . k| <= max||q|| max||k||  that bya_attn_fwd reads from device
 * memory (bya_attn_desc.bound_dev) when the worst case over the LayerNorm's parameters is too large to be useful. */
int bya_qknorm_rope(void* q, void* k, const void* qw, const void* qb, const void* kw, const void* kb,
                    const float* cos, const float* sin, int32_t batch, int32_t S, int32_t heads, int64_t ld,
                    int64_t batch_stride, int32_t text_rows, float eps, float k_scale, float* stats, int32_t stats_slots,
                    hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Flash attention forward, head_dim 64 or 128, no mask, fp32 online softmax, bf16 P.V on MFMA.
 *   O[b1,b2,i,h,:] = softmax_j( scale * Q[b1,b2,i,h,:].K[b1,b2,j,h,:] ) V[b1,b2,j,h,:]
 * Two-level batch (b1 < nb1, b2 < nb2) with independent element strides so one kernel serves: the joint
 * text+video self-attention (F.scaled_dot_product_attention inside CogVideoXAttnProcessor2_0,
 * models/transformer.py:208), the router's spatial attention (models/router.py:476), the perceiver
 * face cross-attention (models/router.py:264-270; q shared by both ids) and the per-frame audio
 * cross-attention (models/audio_model.py:253-256).
 * --------------------------------------------------------------------------------------------- */
typedef struct bya_attn_desc {
    int32_t head_dim;       /* 64 or 128 */
    int32_t heads;
    int32_t nb1, nb2;
    int32_t Sq, Skv;
    int64_t q_s1, q_s2, q_row;   /* element strides: batch level 1, level 2, sequence row */
    int64_t k_s1, k_s2, k_row;
    int64_t v_s1, v_s2, v_row;
    int64_t o_s1, o_s2, o_row;
    float scale;
    int32_t scores_prescaled;   /* 1: q.k is already scale*log2(e)*(q.k) (folded into k upstream); scale is ignored;
                                   head_dim 64 only.  Saves one multiply per score in the VALU-bound softmax. */
    float score_bound;          /* > 0 (with scores_prescaled): the caller GUARANTEES |score| <= score_bound in exp2
                                   units for every (q, k) pair, e.g. because q and k come out of a LayerNorm with known
                                   weights (||q|| <= 8 max|gamma| + ||beta|| at head_dim 64; RoPE is a rotation).  The
                                   softmax then uses no running maximum at all (P = exp2(s): mathematically identical,
                                   shift invariance; every P and every row sum stays a normal fp32 / bf16 number while
                                   |s| <= 90) -- bounds above BYA_ATTN_BOUND_LIMIT are ignored.  0 = unknown. */
    /* Data-dependent bound (head_dim 64, scores_prescaled; NULL = off): squared norms written by bya_qknorm_rope's `stats`,
       fp32 [bound_slots][2][bound_heads]; this launch's (batch, head) index bh reads column bound_bh0 + bh.  Per (batch,
       head) the kernel takes B = sqrt(max_slots q2 * max_slots k2); where B <= BYA_ATTN_BOUND_LIMIT the static kernel runs,
       elsewhere it leaves the head alone, sets fallback_flags[bh] (int32 [nb1 * nb2 * heads], written for EVERY bh, no reset
       needed) and the running-maximum kernel, launched right behind it, computes exactly the flagged heads. */
    const float* bound_dev;
    int32_t bound_slots, bound_heads, bound_bh0;
    int32_t* fallback_flags;
} bya_attn_desc;
#define BYA_ATTN_BOUND_LIMIT 90.0f

int bya_attn_fwd(const void* q, const void* k, const void* v, void* o, const bya_attn_desc* desc,
                 hipStream_t stream);

/* Which softmax variant bya_attn_fwd runs for a descriptor (host-side query, launches nothing): the joint attention
 * silently falls back from the static-bound kernel to the running-maximum kernel when score_bound is above the limit, and callers
 * (tests, bench.py) need to say which one they measured.  Negative = the descriptor is rejected. */
#define BYA_ATTN_D64_RUNNING_MAX 0   /* online softmax, scores scaled in the kernel */
#define BYA_ATTN_D64_PRESCALED 1     /* online softmax, scores already in exp2 units */
#define BYA_ATTN_D64_STATIC_BOUND 2  /* (retired in round 5: the two-block static-bound kernel; never returned) */
#define BYA_ATTN_D128 3
#define BYA_ATTN_D64_STATIC_BOUND_W4 4  /* P = exp2(s), |s| <= score_bound <= 90, on the one-wave-per-SIMD hand-placed kernel (csrc/attn_w4.hip) */
#define BYA_ATTN_D64_DEVICE_BOUND_W4 5  /* the same kernel under the data-dependent bound (bound_dev), per-head running-maximum fallback */
int bya_attn_variant(const bya_attn_desc* desc);

/* Optional stream-K workspace of the static-bound joint-attention kernel (BYA_ATTN_D64_STATIC_BOUND_W4): device memory
 * owned by the caller, 256-byte aligned, at least bya_attn_workspace_bytes (69 MB), ZERO-FILLED once; one per DEVICE, used
 * by launches enqueued under the current device, which must be ordered on one stream.  NULL unregisters.  With it, a
 * bya_attn_fwd launch whose (batch x head, 512-row q-tile) items do not fill whole rounds of 256 CUs (49 x 480 x 720:
 * 1680 items = 6.56 rounds that cost 7; a rank's 6 heads of an 8-GPU step: 210 items on 256 CUs) runs as 256 persistent
 * workgroups: whole rounds item by item, then the leftover items cut at ONE key tile -- "mains" take the key prefix of an
 * item each, the remaining workgroups share the key suffixes -- and an item is completed by adding the fp32 partial (O, l)
 * of its suffix pieces, exchanged through this workspace, to the prefix (partials of the static-bound softmax are
 * additive).  Results equal the one-workgroup-per-item form up to fp32 summation order at the cut items.
 * bya_attn_workspace_status: number of hand-offs that timed out (~1 s; 0 on a healthy run), synchronises `stream`.  (No
 * reference counterpart: scheduling detail of F.scaled_dot_product_attention, models/transformer.py:200-209 via diffusers
 * CogVideoXAttnProcessor2_0.) */
int bya_set_attn_workspace(void* ws, int64_t bytes);
int bya_attn_workspace_bytes(int64_t* bytes);
int bya_attn_workspace_status(int32_t* timeouts, hipStream_t stream);

/* Cross-attention onto at most 64 keys per identity with the router's masked combine in its epilogue:
 *   z[g, n, :] = sum_id w[g n, id] * softmax(q[g, n] . K[id, g]^T * scale) V[id, g]      (fp32 mix, one rounding to bf16)
 * The audio cross-attention (models/audio_model.py:247-258: 32 audio tokens per latent frame g and identity) and the face
 * Perceiver cross-attention (models/router.py:255-270: 32 face tokens per identity), each followed in the reference by the
 * masked combine of models/transformer.py:821-832 / 895-936 AFTER the (linear) output projection: the engine mixes first
 * and projects once.  r: routing logits bf16 [n_grp * Sq, n_id]; af: NULL = face weights (w = r), else the audio-to-face
 * matrix bf16 [n_id, n_id] (w as in bya_masked_combine mode 1).  wsum (optional, fp32 [n_grp * Sq]) = sum_id w, the row scale
 * of the projection's bias.  q rows are shared by all identities.  Element strides; head h of a row starts at h * head_dim.
 * Up to 32 keys (both callers) with z 16-byte aligned: the K / V of every identity stay in LDS for a (group, head) and z is
 * stored as whole head segments; otherwise one 128-row tile per workgroup on 64-key tiles -- the same bits either way. */
typedef struct bya_attn_mix_desc {
    int32_t head_dim, heads, n_id, n_grp, Sq, Skv;
    int64_t q_grp, q_row, k_id, k_grp, k_row, v_id, v_grp, v_row, z_grp, z_row;
    float scale;
} bya_attn_mix_desc;
int bya_attn_kv_mix(const void* q, const void* k, const void* v, const void* r, const void* af, void* z, float* wsum,
                    const bya_attn_mix_desc* desc, hipStream_t stream);

/* Tiny-sequence self-attention (sequence length L <= 16, head_dim 64) used by the router's temporal
 * (L = frames) and multi-ID (L = ids) attentions (models/router.py:482,488).  Element e of sequence
 * `g` lives at row  g_outer(g)*outer_stride + e*seq_stride + g_inner(g)  of the [rows, ld] matrices,
 * with g = g_outer * n_inner + g_inner. */
int bya_attn_tiny(const void* q, const void* k, const void* v, void* o, int32_t L, int32_t heads,
                  int64_t n_outer, int64_t n_inner, int64_t outer_stride, int64_t seq_stride, int64_t ld_qkv,
                  int64_t ld_o, float scale, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Embedding-Router specific kernels (models/router.py:364-411).
 * router_scores: s[id,n,tok*heads+h] = sum_d qr[n,h*128+d] * kr[id,tok,h*128+d]; LayerNorm(512, w,b);
 *                + pos_emb[n]  ->  out[id,n,512]
 * router_head:   r[n,id] = sigmoid(x[id,n,:].w + b)  ->  [N, n_id]  (the [1,N,n_id] routing logits)
 * (router_scores keeps an identity's 32 keys in LDS from 4096 tokens on; same bits as the per-wave form below that.)
 * --------------------------------------------------------------------------------------------- */
int bya_router_scores(const void* qr, const void* kr, const void* ln_w, const void* ln_b, const void* pos_emb,
                      void* out, int32_t n_id, int64_t N, int32_t heads, int32_t face_tokens, float eps,
                      hipStream_t stream);
int bya_router_head(const void* x, const void* w, const void* b, void* r, int32_t n_id, int64_t N, int32_t D,
                    hipStream_t stream);

/* Forcing override (models/transformer.py:813-819): out[t,r,id] = max_t' forcing[t',r,id]. Bit-exact. */
int bya_forcing_max_over_frames(const void* forcing, void* out, int32_t frames, int64_t per_frame,
                                int32_t n_id, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Masked combines (models/transformer.py:821-822,831-832 and 860-863,895-900,925-926,935-936).
 *  mode 0 (face):  x[b,n,:] += alpha * sum_id r[b][n,id] * feat[b,id,n,:]
 *  mode 1 (audio): av[n,i] = bf16(sum_j af[b,i,j]*r[b][n,j]); w[n,id] = bf16(1 - av[n,1-id]);
 *                  x[b,n,:] += sum_id w[n,id] * feat[b,id,n,:]
 *                  n_id in 3..4 (the reference hard-codes two streams; build-defined): af is [batch, n_id, n_id] and
 *                  w[n,a] = prod_{b != a} bf16(1 - av[n,b]), every product rounded to bf16 (two streams: same bits)
 * r: [batch or 1, N, n_id] (r_batch_stride 0 = shared forcing mask); feat [batch, n_id, N, D]. In place.
 * --------------------------------------------------------------------------------------------- */
int bya_masked_combine(void* x, const void* feat, const void* r, const void* af, int32_t mode, float alpha,
                       int32_t batch, int32_t n_id, int64_t N, int32_t D, int64_t x_row, int64_t x_batch_stride,
                       int64_t r_batch_stride, hipStream_t stream);

/* Routed mix BEFORE the output projection (to_out is linear, so sum_id w*(o_id W^T + b) = (sum_id w*o_id) W^T +
 * (sum_id w) b): z[b,n,:] = sum_id w[b,n,id] * feat[b,id,n,:], wsum[b,n] = sum_id w (fp32, may be NULL); w derived from
 * the routing logits exactly as in bya_masked_combine (mode 0 face, mode 1 audio).  The half-size GEMM that follows
 * takes wsum as bias_rowscale and the hidden stream as residual. */
int bya_routed_mix(const void* feat, const void* r, const void* af, void* z, float* wsum, int32_t mode, int32_t batch,
                   int32_t n_id, int64_t N, int32_t D, int64_t r_batch_stride, hipStream_t stream);

/* Patchify (im2col of the 2x2/stride-2 Conv2d of CogVideoXPatchEmbed, models/transformer.py:690):
 *   cols[b, (t*Ht+h)*Wt+w, c*4+ph*2+pw] = x[b,t,c,2h+ph,2w+pw]
 * and unpatchify (models/transformer.py:955-957):
 *   out[b,t,c,2h+ph,2w+pw] = y[b,(t*Ht+h)*Wt+w, c*4+ph*2+pw].  Index-only, bit-exact. */
int bya_patchify(const void* x, void* cols, int32_t batch, int32_t frames, int32_t channels, int32_t H, int32_t W,
                 hipStream_t stream);
int bya_unpatchify(const void* y, void* out, int32_t batch, int32_t frames, int32_t channels, int32_t H, int32_t W,
                   hipStream_t stream);

/* Elementwise helpers: y = act(x) (+ r) over n bf16 elements (n % 8 == 0). */
int bya_act_add(const void* x, const void* r, void* y, int64_t n, int32_t act, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Row-stationary GEMM for K = 512 with optional fused LayerNorm, GELU(erf) and residual:
 *     C[M,N] = res + act( LN?(X)[M,512] . W[N,512]^T + bias )
 * Replaces, inside every SpatialTemporalAttentionBlock of the Embedding Router (models/router.py:468-493), the
 * pairs  nn.LayerNorm -> to_q|to_k|to_v  and  nn.LayerNorm -> mlp[0] -> GELU  (ln = 1) and the to_out / mlp[2]
 * projections with their residual adds (ln = 0, res = X of the block).
 * ln = 1: W must be the gamma-folded weight  Wg[n,k] = W[n,k] * gamma[k]  (bf16),  colsum[n] = sum_k Wg[n,k]  (fp32),
 *         cvec[n] = sum_k W[n,k] * beta[k] + bias[n]  (fp32);  the kernel computes the row mean / rstd itself and
 *         applies  rstd * (x . Wg^T - mean * colsum) + cvec.
 * ln = 0: W as is, cvec = bias (fp32), colsum ignored.
 * X: bf16 rows of 512 with row stride ldx; C / res: bf16 with row strides ldc / ldres (C may alias res).
 * N % (64 * nsplit) == 0; nsplit <= 0 lets the library choose how many column groups share a row block.
 * act: BYA_ACT_NONE or BYA_ACT_GELU_ERF.
 * --------------------------------------------------------------------------------------------- */
int bya_rowgemm512(const void* X, const void* W, const float* colsum, const float* cvec, const void* res, void* C,
                   int32_t M, int32_t N, int32_t ldx, int32_t ldc, int32_t ldres, int32_t ln, float eps, int32_t act,
                   int32_t nsplit, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused  LayerNorm(512) -> to_q | to_k | to_v (8 heads x 64) -> softmax(q k^T / 8) v  over SMALL groups of rows: the
 * temporal attention (the 13 frames of one (identity, location)) and the multi-ID attention (the identities of one
 * token) of SpatialTemporalAttentionBlock.  Replaces models/router.py:476-478 / :482-484 up to (not including) the
 * to_out projection: diffusers Attention + the default SDPA processor on [B', L, 512] with L = 13 / n_id, i.e. the
 * sequence  bya_rowgemm512(ln = 1, N = 1536)  ->  bya_attn_tiny  without the [M, 1536] q|k|v tensor in between.
 * X [M, 512] bf16 rows (stride ldx), O [M, 512] bf16 attention output in the SAME row order (stride ldo; may not alias X).
 * Wqkv [1536, 512] bf16: rows 0..511 = gamma-folded to_q, 512..1023 = to_k, 1024..1535 = to_v; colsum / cvec [1536] fp32
 * exactly as for bya_rowgemm512 with ln = 1.  Groups as for bya_attn_tiny: group (o, i), o < n_outer, i < n_inner,
 * consists of the L rows  o * outer_stride + i + e * seq_stride,  e < L;  1 <= L <= 32: groups of up to 16 rows share a
 * 16-row MFMA tile in power-of-two cells, groups of 17 .. 32 rows (25 latent frames of a 97-frame clip) take the two
 * tiles of one wave; longer sequences return BYA_ERR_UNSUPPORTED (the unfused pair handles them).
 * scale = 1 / sqrt(64).  q, k, v are rounded to bf16 where the unfused path stored them, P to bf16 before P.V.
 * --------------------------------------------------------------------------------------------- */
int bya_router_group_attn(const void* X, const void* Wqkv, const float* colsum, const float* cvec, void* O,
                          int32_t M, int32_t ldx, int32_t ldo, int32_t L, int64_t n_outer, int64_t n_inner,
                          int64_t outer_stride, int64_t seq_stride, float eps, float scale, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Router chains (round 6; csrc/rowchain.hip): two Linears of a SpatialTemporalAttentionBlock sub-block in ONE launch, the
 * [M, 512] activation between them in registers, updating the router rows in place.
 *
 * bya_router_mlp_fused:  C = X + mlp[2]( GELU_erf( mlp[0]( LayerNorm(X) ) ) )  -- models/router.py:491 with norm4 -- i.e.
 *   bya_rowgemm512(ln = 1, act = GELU_ERF) -> bya_rowgemm512(res = X)  without the hidden tensor.  W1 [512, 512] gamma-
 *   folded with colsum1 / cvec1 as for bya_rowgemm512 ln = 1; W2 [512, 512] with cvec2 = its bias.  C may be X (every row
 *   is read and written by one wave); the router's hidden width must equal 512 (mlp_ratio 1, models/router.py:318).
 * bya_router_group_attn_out:  C = X + to_out( attention over row groups( LayerNorm(X) ) )  -- models/router.py:482-483 /
 *   :486-487 -- i.e.  bya_router_group_attn -> bya_rowgemm512(res = X)  without the attention-output tensor.  Wqkv, colsum,
 *   cvec, the group geometry (L, n_outer, n_inner, outer_stride, seq_stride) and scale exactly as for
 *   bya_router_group_attn; Wo [512, 512] with cvec_o = its bias.  1 <= L <= 16 (a group is held by ONE 16-row MFMA tile;
 *   longer groups: BYA_ERR_UNSUPPORTED, the caller keeps the two launches).  C may be X.  Rows outside every group are
 *   neither read nor written.
 * Both give, BIT FOR BIT, what their two launches give (same MFMA order over K, same epilogue expressions on the same
 * rounded values).  tiles_pass0: 16-row tiles per workgroup in the first pass over the rows (1..8; 0 = 8); the remainder
 * is dealt over all workgroups in a short second pass.  A tuning hint: results do not depend on it.
 * --------------------------------------------------------------------------------------------- */
int bya_router_mlp_fused(const void* X, const void* W1, const float* colsum1, const float* cvec1, const void* W2,
                         const float* cvec2, void* C, int32_t M, int32_t ldx, int32_t ldc, float eps, int32_t tiles_pass0,
                         hipStream_t stream);
int bya_router_group_attn_out(const void* X, const void* Wqkv, const float* colsum, const float* cvec, const void* Wo,
                              const float* cvec_o, void* C, int32_t M, int32_t ldx, int32_t ldc, int32_t L, int64_t n_outer,
                              int64_t n_inner, int64_t outer_stride, int64_t seq_stride, float eps, float scale,
                              int32_t tiles_pass0, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Classifier-free-guidance combine + scheduler step in one pass over the latents.
 * Replaces models/pipeline_bindyouravatar.py:924-948: noise = u + g*(c - u) in fp32 on the bf16 prediction
 * (n_pred = 2: [uncond, cond], second sample at pred + pred_stride; n_pred = 1: no guidance), then the
 * v-prediction step of diffusers' CogVideoXDDIMScheduler / CogVideoXDPMScheduler:
 *   x0   = bf16(sqrt_alpha * x) - sqrt_beta * noise_pred
 *   d    = old_x0 ? k_cur * x0 - k_old * old_x0 : x0                (DPM second-order correction)
 *   prev = bf16( (bf16(k_sample * x) - k_denoised * d) [+ bf16(k_noise * noise)] )
 * DDIM: k_sample = a_t, k_denoised = -b_t, old_x0 = noise = NULL.  x0_out (fp32, optional) feeds the next DPM step.
 * The bf16() roundings are where torch's type promotion rounds (0-dim coefficient x bf16 tensor).
 * --------------------------------------------------------------------------------------------- */
typedef struct bya_sched_coef {
    float guidance, sqrt_alpha, sqrt_beta, k_sample, k_denoised, k_noise, k_cur, k_old;
} bya_sched_coef;

int bya_cfg_scheduler_step(const void* pred, int32_t n_pred, int64_t pred_stride, const void* sample,
                           const float* old_x0, const void* noise, void* prev_sample, float* x0_out,
                           int64_t n, const bya_sched_coef* coef, hipStream_t stream);

/* Tracking masks -> routing_logits_forcing (stage 2 of the reference inference; util/utils.py:481-514 resize_mask,
 * :871-936 process_masks_to_routing_logits parts 2-3).  masks: uint8 [n_id, in_frames, in_h, in_w], > 0 = foreground;
 * logits: bf16 [frames*h*w, n_id], one-hot per token (zero row = background), the LAST identity whose trilinearly
 * resized (align_corners = false) mask exceeds 0.5 wins.  Index / threshold work: bit-exact against the reference. */
int bya_masks_to_routing_logits(const void* masks, void* logits, int32_t n_id, int32_t in_frames, int32_t in_h,
                                int32_t in_w, int32_t frames, int32_t h, int32_t w, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Video VAE either side of the denoise loop: AutoencoderKLCogVideoX (diffusers; un-vendored
 * third-party layer) as called by models/pipeline_bindyouravatar.py:406-424 (encode of the conditioning frame) and :461-466
 * (decode of the finished latents).  Activations are channels-last bf16 [T, H, W, C]; every convolution is
 * bya_vae_patches + bya_gemm_bf16 (bias / residual in its epilogue), 1x1x1 convolutions are bya_gemm_bf16 alone.
 * --------------------------------------------------------------------------------------------- */
/* Patch matrix of a causal KT x 3 x 3 convolution (KT = 3: CogVideoXCausalConv3d; KT = 1: the per-frame 3 x 3 convolutions of
 * CogVideoXUpsample3D / CogVideoXDownsample3D) for output frames [t0, t0 + nt) of a chunk: out [nt * Ho * Wo, Kpad] bf16, column
 * (kt, kh, kw, c), zero beyond KT * 9 * C.  x: the chunk [Ts, Hs, Ws, C]; cache: the previous chunk's last KT - 1 frames
 * [KT - 1, Hs, Ws, C] or NULL (first chunk: frame 0 is repeated -- diffusers pad_mode "first").  Space: tap (kh, kw) of output
 * (h, w) reads (h * stride + kh - pad, w * stride + kw - pad), zero outside.  up = 1: the convolution runs on the
 * nearest-neighbour x2 up-sampling of x in space (never materialised) and in time by tmode: 0 = frames as stored, 1 = every
 * frame doubled, 2 = first frame single, the rest doubled (CogVideoXUpsample3D.compress_time with an odd frame count). */
int bya_vae_patches(const void* x, const void* cache, void* out, int32_t Ts, int32_t Hs, int32_t Ws, int32_t C, int32_t KT,
                    int32_t stride, int32_t pad, int32_t up, int32_t tmode, int32_t Ho, int32_t Wo, int32_t t0, int32_t nt,
                    int32_t Kpad, hipStream_t stream);
/* GroupNorm statistics of one chunk: sums [groups][2] fp32 = (sum, sum of squares) over x [rows, C], summed in a FIXED order
 * (no atomics: the same chunk gives the same bits every run).  partial: caller's scratch, ceil(rows / 512) * groups * 2 floats.
 * C <= 512 with C / 8 a power of two. */
int bya_vae_groupnorm_stats(const void* x, float* sums, float* partial, int64_t rows, int32_t C, int32_t groups,
                            hipStream_t stream);
/* y = act( GroupNorm(x; sums, gamma, beta, eps) [ * zy[z(row)] + zb[z(row)] ] ), act 0 = none, 1 = SiLU.  zy / zb (both or
 * neither): conv_y / conv_b of CogVideoXSpatialNorm3D evaluated at LATENT resolution [Tz * hz * wz rows, row stride ldz]; row
 * (t, h, w) of x [T, H, W, C] reads latent position (frame by tmode, h >> log2(H / hz), w >> log2(W / wz)): tmode 0 = same
 * frame, 1 = floor(t Tz / T), 2 = first frame apart (0 -> 0, t -> 1 + floor((t - 1)(Tz - 1) / (T - 1))): torch's nearest
 * F.interpolate, first frame and the rest resized separately when T is odd and > 1. */
/* out_pad = 1: y is the zero-padded input of bya_vae_conv3d, [T + 2, H + 2, W + 2, C] (the caller zero-fills it once and
 * writes the two context frames; this call writes pixel (t, h, w) at (t + 2, h + 1, w + 1)); rows must be T H W. */
int bya_vae_norm_act(const void* x, void* y, const float* sums, const void* gamma, const void* beta, const void* zy,
                     const void* zb, int64_t rows, int32_t C, int32_t groups, int32_t act, float eps, int32_t T, int32_t H,
                     int32_t W, int32_t Tz, int32_t hz, int32_t wz, int32_t tmode, int64_t ldz, int32_t out_pad,
                     hipStream_t stream);

/* Causal KT x 3 x 3 convolution, stride 1 (KT = 3: diffusers CogVideoXCausalConv3d inside CogVideoXResnetBlock3D and conv_out;
 * KT = 1: the per-frame 3 x 3 convolution of CogVideoXUpsample3D; reached from models/pipeline_bindyouravatar.py:461-466 /
 * :406-424), as an IMPLICIT GEMM on the persistent MFMA kernel: no patch matrix.
 * xpad: bf16 [To + KT - 1, H + 2, W + 2, C], channels-last, zero border of one pixel around every frame; for KT = 3 frames 0
 * and 1 are the causal context (last two input frames of the previous chunk, or the first frame twice); w: [Cout, ldw] bf16
 * with column ((dt 3 + dh) 3 + dw) C + c; bias [Cout] or NULL; res [To, H, W, ldres] or NULL; out [To, H, W, ldc] (may alias
 * res).  C in {128, 256, 512}; Cout, ldc, ldres multiples of 8.  out = res + bias + conv(x). */
int bya_vae_conv3d(const void* xpad, const void* w, const void* bias, const void* res, void* out, int32_t To, int32_t H,
                   int32_t W, int32_t C, int32_t Cout, int32_t KT, int64_t ldw, int64_t ldc, int64_t ldres,
                   hipStream_t stream);

/* Nearest-neighbour up-sampling of x [T, H, W, C] (space x 2; time by tmode as in bya_vae_patches: 0 = frames as they are,
 * 1 = every frame doubled, 2 = first frame single and the rest doubled) into the interior of ypad [To, 2 H + 2, 2 W + 2, C],
 * the zero-padded input of the up-sampler's convolution (bya_vae_conv3d, KT = 1); the caller zero-fills ypad once. */
int bya_vae_upsample_pad(const void* x, void* ypad, int32_t T, int32_t H, int32_t W, int32_t C, int32_t tmode,
                         hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU exchanges of the sharded step over RCCL.  ``comm`` is the caller's ncclComm_t (one
 * process per GPU); both calls only ENQUEUE on ``stream`` -- give them a stream of their own to overlap with compute.
 * The reference has no inference parallelism; these have no counterpart there.  (The Python module of this package
 * issues the same exchanges through torch.distributed's nccl backend = RCCL, because torch does not expose its
 * communicator.)  BYA_ERR_UNSUPPORTED: no RCCL in the process.
 * --------------------------------------------------------------------------------------------- */
/* Exchange A, K/V form: k_local / v_local [rows_local, row_elems] bf16 of this rank -> k_full / v_full
 * [world * rows_local, row_elems] on every rank (rank-major = global row order), one grouped call. */
int bya_allgather_kv(const void* k_local, const void* v_local, void* k_full, void* v_full, int64_t rows_local,
                     int64_t row_elems, void* comm, hipStream_t stream);
/* Exchange A head-parallel form and exchange B (router repartition frame-major <-> location-major): uneven all-to-all
 * of bf16 elements.  send_counts[p] elements go to rank p from send + sum(send_counts[:p]); recv_counts[p] elements
 * arrive from rank p at recv + sum(recv_counts[:p]).  Grouped point-to-point: every element crosses one xGMI link once. */
int bya_alltoall_router(const void* send, void* recv, const int64_t* send_counts, const int64_t* recv_counts,
                        int32_t world, void* comm, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * P2P exchange engine (csrc/comm.hip): the exchanges of the sharded step as ordinary kernels that store
 * straight into the peers' HBM over xGMI -- one launch per exchange whatever its scatter/gather list, capturable in a
 * hipGraph (RCCL collectives are neither).  No reference counterpart (the reference has no inference parallelism).
 * Set-up is the host's (bind_your_avatar_implementation_amd/p2p.py): every rank allocates its receive buffers and one
 * control block of 64 uint32 words per channel (zero-filled), trades hipIpc handles once, and builds per channel, in DEVICE
 * memory, (a) the copy table (bya_p2p_copy below): `src` local, `dst` an address inside a peer's (or its own) receive buffer
 * mapped into this process; 16-byte aligned pieces and pitches take the fast path; (b) `peer_ctrl[p]` = the
 * channel's control block ON PEER p (mapped), p < world, own rank included.
 * bya_p2p_push: copy every table entry, then publish the channel's next sequence number to word `rank` of every peer's
 * control block.  bya_p2p_wait (receiver, same channel, once per push of the peers): returns to the stream when all
 * `world` source ranks have published the receiver's next expected number; kernels enqueued behind it see the data (the
 * wait runs as 16 small workgroups, dealt over the XCDs, each ending in a system-scope acquire).  bya_p2p_exchange = push
 * and wait in ONE launch, for the exchanges nothing is overlapped with.
 * Sequence numbers live in the control block (words 32 / 33): a hipGraph replay advances them by itself.  A wait is bounded
 * by wall time: `wait_limit_ms` of the launch (<= 0: 30 s; a captured launch keeps the limit it was captured with).  One
 * that gives up counts itself in word 35 of its channel and in word 37 of `group_ctrl` -- the FIRST control block of the
 * group the channel belongs to (may be NULL: then only the channel's own word is used) -- both sticky.  A wait that finds
 * the group's word set returns at once, counted as timed out: after the first time-out of a group the rest of the step runs
 * through without polling.  bya_p2p_poison(ctrl_base, n_channels, out, n) -- enqueue it behind the last consumer of a step --
 * overwrites the bf16 tensor `out` with NaN when any of the n_channels control blocks (64 words apart) carries a time-out or a
 * non-zero word 38 (written by the host: "this step's exchange checksum differed between the ranks"), so that a result made
 * from a buffer that never arrived, or arrived stale, cannot be consumed.  The caller guarantees that a receive buffer is not
 * pushed into again before its owner has consumed it (the step's data dependencies do).
 * --------------------------------------------------------------------------------------------- */
typedef struct bya_p2p_copy {      /* a 2-D piece: `rows` rows of `row_bytes` bytes (a contiguous piece is ONE row) */
    const void* src;
    void* dst;
    int64_t row_bytes;             /* % 2 == 0 */
    int64_t rows;
    int64_t src_pitch, dst_pitch;  /* bytes from one row to the next on either side (% 2 == 0; ignored when rows == 1) */
    int64_t chunk0;                /* chunks of the entries before this one; an entry has rows * ceil(row_bytes / 64 KiB) chunks
                                      when row_bytes > 64 KiB, else ceil(rows / floor(64 KiB / row_bytes)) */
} bya_p2p_copy;

int bya_p2p_push(const bya_p2p_copy* copies_dev, int32_t n_copies, int64_t total_chunks, void* const* peer_ctrl_dev,
                 int32_t world, int32_t rank, void* ctrl, hipStream_t stream);
int bya_p2p_wait(void* ctrl, int32_t world, void* group_ctrl, int64_t wait_limit_ms, hipStream_t stream);
int bya_p2p_exchange(const bya_p2p_copy* copies_dev, int32_t n_copies, int64_t total_chunks, void* const* peer_ctrl_dev,
                     int32_t world, int32_t rank, void* ctrl, void* group_ctrl, int64_t wait_limit_ms, hipStream_t stream);
int bya_p2p_poison(const void* ctrl_base, int32_t n_channels, void* out, int64_t n_elems, hipStream_t stream);
/* Set-up only (the ONLY entry points of this library that allocate; never called inside a step): device memory a peer may
 * store into or a running kernel polls, by kind -- 0 coarse-grained (hipMalloc), 1 fine-grained, 2 uncached
 * (hipExtMallocWithFlags) -- zero-filled, and its hipIpc handle (64 bytes) for the peers.  BYA_ERR_UNSUPPORTED: the
 * platform refused (the host module then falls back, p2p.py). */
int bya_p2p_alloc(int64_t bytes, int32_t kind, void** out);
int bya_p2p_free(void* ptr);
int bya_p2p_ipc_export(void* ptr, void* handle64);
int bya_p2p_ipc_import(const void* handle64, void** out);
int bya_p2p_ipc_release(void* ptr);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* BYA_H */
