"""The per-denoise-step engine: host-side sequencing of the hand-written HIP kernels.

One ``DenoiseEngine.step`` == one reference ``BindyouravatarTransformer3DModel.forward``
(reference models/transformer.py:615-964, inference branch).  Design (DESIGN.md has the full picture):

* ONE resident activation stream ``x [B, S, D]`` (text rows first, video rows behind): the reference's
  ``torch.cat([enc, hid])`` / ``split`` round trips (diffusers CogVideoXAttnProcessor2_0, CogVideoXBlock)
  disappear; kernels pick the text/video AdaLN triplet by row index.
* All 2L+1 AdaLN modulation vectors of a step come from ONE weight-streaming launch over packed weights
  (``emb`` is layer-invariant).
* The per-sample Python loops + ``repeat(2,1,1)`` + ``empty_cache()`` of the reference
  (models/transformer.py:779-832, 870-936) become batched launches; work the reference does twice on
  identical data (perceiver ``to_q``, router ``norm_q``/``to_q`` for both ids) is done once.
* The router's head-interleaving permutes (models/router.py:375-378) are folded into the weights at pack time.
* Step-invariant tensors (face tokens, per-layer face K/V, router keys, audio context, per-layer audio K/V) are
  produced by ``_face_invariants`` / ``_audio_invariants``; they are recomputed every step like the reference
  unless ``cache_invariants`` is switched on (explicit opt-in used by the pipeline).

Nothing here falls back to torch math: torch is used for device memory, views and copies only.
"""
import contextlib
import os

import torch

from . import ops
from .parallel import RouterPartition, SeqShard


def _key(tensors):
    return tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in tensors)


# the Linears that can run on e4m3 operands: attn1 q|k|v, attn1.to_out, ff.net.0, ff.net.2 of every DiT block, and the
# perceiver / audio query projections
FP8_LINEARS = ("qkv", "out", "ff1", "ff2", "pq", "aq")
# ... and the ones that do by default: the four DiT Linears.  The two query projections feed the Embedding Router and the
# cross-attentions, whose sigmoid routing amplifies their error: in e4m3 the perceiver to_q ALONE doubles the 42-layer output
# error (3.7e-2 against the bf16 engine's 1.8e-2; any one DiT Linear: 1.9-2.0e-2) for 1 % of the step time
# (tools/fp8_error_by_linear.py, profiles/history/r3_fp8_error_by_linear.json).
FP8_DEFAULT = ("qkv", "out", "ff1", "ff2")


class DenoiseEngine:
    # Identities / audio streams per sample.  The reference forward dereferences exactly id_cond[0], id_cond[1]
    # (models/transformer.py:638-639) and repeats the video twice (:784, :881); everything else in it (router, perceiver,
    # masked bmm) already runs over a leading identity axis.  Here the count follows the inputs (2..4, BASELINE
    # configs[4] uses 3); the one place with no n-identity form in the reference, the audio weights' [1, 0] swap, is
    # generalised as w[a] = prod_{b != a} (1 - av[b]) (include/bya.h, bya_masked_combine).
    MAX_ID = 4
    n_id = None          # identity count of the LAST step (informational, for tests); never read by the engine itself

    def _count_ids(self, id_cond, audio_embeds):
        if self.m.is_train_face and id_cond is not None:
            n = len(id_cond)
        elif audio_embeds is not None and audio_embeds.ndim == 5:
            n = audio_embeds.shape[1]
        else:
            n = 2
        if not 2 <= n <= self.MAX_ID:
            raise ValueError(f"{n} identities: the engine runs 2..{self.MAX_ID} (the reference itself exactly 2)")
        return n

    def __init__(self, model):
        p = model.proj_out.weight
        if not p.is_cuda:
            raise RuntimeError("the Bind-Your-Avatar MI355X engine needs its parameters on a GPU "
                               "(there is no CPU fallback); call model.to('cuda')")
        if p.dtype != torch.bfloat16:
            raise RuntimeError(f"engine computes in bf16 (MFMA, fp32 accumulate); parameters are {p.dtype}")
        ops._hip.load()      # fail loudly right here if libbya_hip.so is missing
        ops.ensure_gemm_workspace(p.device)      # split-K slabs of the persistent GEMM kernel (one per process)
        self.m = model
        self.cfg = model.config
        self.dev = p.device
        self.D = model.inner_dim
        self.H = self.cfg.num_attention_heads
        self.L = len(model.transformer_blocks)
        self.Tt_max = self.cfg.max_text_seq_length
        self.cache_invariants = False
        # softmax scale * log2(e) of the joint attention is folded into k where k is finished in fp32 (q/k-norm + RoPE
        # kernel), so the attention kernel's scores are born in exp2 units (no per-score multiply)
        self.k_scale = (self.cfg.attention_head_dim ** -0.5) * 1.4426950408889634
        # route-then-project (exact by linearity of to_out; halves those GEMMs and drops the [2,N,D] round trip).
        # False = the reference's order of operations (project each identity, then combine).
        self.mix_before_projection = os.environ.get("BYA_MIX_BEFORE_PROJECTION", "1") != "0"
        # ... and mix inside the cross-attention kernel's epilogue (bya_attn_kv_mix): the per-identity attention outputs never
        # reach HBM.  "0" = attention, then bya_routed_mix (the round-2 path; kept for the exact-row-selection tests)
        self.fused_attn_mix = self.mix_before_projection and os.environ.get("BYA_FUSED_ATTN_MIX", "1") != "0"
        # BASELINE configs[4]: the four big Linears of every DiT block on e4m3 operands (per-channel weight scales taken at
        # pack time, per-row activation scales on the fly; fp32 accumulation, bf16 everywhere else)
        self.fp8_weights = bool(getattr(model, "_fp8_weights", False)) or os.environ.get("BYA_FP8_WEIGHTS") == "1"
        self.fuse_ln_quant = os.environ.get("BYA_FP8_FUSED_LN", "1") != "0"     # AdaLN LayerNorm writes e4m3 directly
        # the remaining A/B switches of the step, read ONCE here (round 4 looked them up in os.environ on every step / call)
        self.side_stream_conditioning = os.environ.get("BYA_INVARIANTS_SIDE_STREAM", "1") != "0"
        self.sp_allgather = os.environ.get("BYA_SP_ALLGATHER", "0") == "1"       # exchange A as a K/V all-gather (A/B)
        self.router_fused_attn = os.environ.get("BYA_ROUTER_FUSED_ATTN", "1") != "0"
        # (r6) the location-major sub-blocks as chains (csrc/rowchain.hip: attention + to_out + residual, MLP pair): "1" where
        # the measurements say a chain wins (ROUTER_CHAIN_ROWS: the 2- and 4-rank shard shapes; at the one-GPU 35100 rows the
        # remainder pass eats the gain, at an 8-rank shard 34 workgroups are too few), "0" never, "all" always.  A chain is
        # bit-identical to its two launches, so the choice by row count keeps "a rank's shard rounds like the whole clip".
        self.router_chain = os.environ.get("BYA_ROUTER_CHAIN", "1")
        # (r6) sharded step on the P2P transport: compare a checksum of the step's gathered prediction across the ranks, every
        # step (parallel.SeqShard.verify_gathered); bench.py --gpus N and the sharded tests switch it on
        self.verify_exchanges = os.environ.get("BYA_SP_VERIFY", "0") == "1"
        # (r6) sharded step, P2P transport: exchange A's v third on the side stream underneath the q | k projection
        # ("1" where a rank has at least SP_OVERLAP_V_MIN_ROWS rows, "0" never, "all" always)
        self.sp_overlap_v = os.environ.get("BYA_SP_OVERLAP_V", "1") != "0"
        self.sp_shard_mods = os.environ.get("BYA_SP_SHARD_MODS", "1") != "0"      # (r6) the AdaLN vector in column slices, one per rank
        if os.environ.get("BYA_SP_OVERLAP_V") == "all":
            self.SP_OVERLAP_V_MIN_ROWS = 0
        self.router_replicated = os.environ.get("BYA_ROUTER_REPLICATED", "0") == "1"
        # q/k-norm + RoPE inside the q|k|v projection's epilogue (bya_gemm_qkv_norm_rope; bit-identical to the two launches)
        self.qkn_epilogue = os.environ.get("BYA_QKN_EPILOGUE", "1") != "0"
        self._inv_cache = {}
        self._side = None              # side stream of the step-invariant conditioning
        self._ws_key, self._ws = None, None
        self._parts = {}
        if model.is_train_audio and not model.is_train_face:
            raise RuntimeError("audio injection needs the face router's logits (models/transformer.py:860)")
        self._pack()

    def _fingerprint(self):
        """Cheap identity of the parameter set the packed copies were made from, taken from the LIVE module tree on every
        call (a rebound ``module.weight = nn.Parameter(...)`` is a different tensor object): storage pointers catch
        replaced tensors, ``_version`` catches in-place edits (``copy_``, ``load_state_dict`` of a submodule, optimiser
        steps).  Tensors created under ``torch.inference_mode()`` have no version counter: their pointer alone counts."""
        ptrs = vers = n = 0
        for group in (self.m.parameters(), self.m.buffers()):
            for t in group:
                n += 1
                ptrs += t.data_ptr()
                if not t.is_inference():
                    vers += t._version
        return (n, ptrs, vers)

    def refresh_if_stale(self):
        """A submodule was loaded or edited in place without ``invalidate_engine()``: the packed copies (q|k|v, AdaLN,
        permuted router weights, folded LayerNorms, score bounds) and everything derived from them (cached
        conditioning, captured graphs) are stale -> rebuild.  Returns True when it did."""
        if self._fingerprint() == self._fp:
            return False
        self._pack()
        self._inv_cache = {}
        self.m._graphs = {}
        return True

    # ------------------------------------------------------------------------------------------ packing
    def _pack(self):
        m, D = self.m, self.D
        cat = torch.cat
        self._fp = self._fingerprint()
        # |q.k| * k_scale <= (8 max|gamma_q| + ||beta_q||)(8 max|gamma_k| + ||beta_k||) * k_scale for q, k out of
        # LayerNorm(64) (||x_hat|| <= 8) followed by RoPE (a rotation); 2 % slack for the bf16 roundings of q and k.
        # bya_attn_fwd runs its softmax without a running maximum when that is <= ops.ATTN_BOUND_LIMIT (include/bya.h).  A
        # layer whose WORST CASE is larger (learned gains: a few large elements of gamma are enough) does not lose the fast
        # kernel: its q/k-norm launch then records the norms the rows actually have, and the attention takes
        # max||q|| max||k|| per (batch, head) from device memory (``_attn_bound``), heads above the limit alone fall back.
        self.score_bound = []
        for blk in m.transformer_blocks:
            nq, nk = blk.attn1.norm_q, blk.attn1.norm_k
            bq = 8.0 * nq.weight.float().abs().max().item() + nq.bias.float().norm().item()
            bk = 8.0 * nk.weight.float().abs().max().item() + nk.bias.float().norm().item()
            self.score_bound.append(1.02 * bq * bk * self.k_scale if os.environ.get("BYA_ATTN_BOUNDED", "1") != "0" else 0.0)
        self.device_bound = os.environ.get("BYA_ATTN_DEVICE_BOUND", "1") != "0"
        mods_w, mods_b = [], []
        for blk in m.transformer_blocks:
            for nz in (blk.norm1, blk.norm2):
                mods_w.append(nz.linear.weight)
                mods_b.append(nz.linear.bias)
        mods_w.append(m.norm_out.linear.weight)
        mods_b.append(m.norm_out.linear.bias)
        self.mod_w = cat(mods_w).contiguous()          # [(2L*6 + 2) * D, temb]
        self.mod_b = cat(mods_b).contiguous()
        self.patch_w = m.patch_embed.proj.weight.reshape(D, -1)   # [D, C*4], k = c*4 + ph*2 + pw
        # q|k|v projection weights packed per block: one GEMM launch with N = 3D fills the CUs evenly
        # (N = D alone leaves the last of 3.3 "rounds" of 256x256 tiles three-quarters empty)
        self.qkv_w = [cat([b.attn1.to_q.weight, b.attn1.to_k.weight, b.attn1.to_v.weight]).contiguous()
                      for b in m.transformer_blocks]
        self.qkv_b = [cat([b.attn1.to_q.bias, b.attn1.to_k.bias, b.attn1.to_v.bias]).contiguous()
                      for b in m.transformer_blocks]
        self.w8 = None
        if self.fp8_weights:
            # which Linears run in e4m3 (enable_fp8_weights(linears=...) / BYA_FP8_LINEARS; "all" = every kind): the rest stays bf16
            keep = getattr(m, "_fp8_linears", None) or os.environ.get("BYA_FP8_LINEARS") or FP8_DEFAULT
            keep = set(FP8_LINEARS if keep == "all" else keep.split(",")) if isinstance(keep, str) else set(keep)
            unknown = keep - set(FP8_LINEARS)
            if unknown:
                raise ValueError(f"fp8 linears {sorted(unknown)}: expected a subset of {FP8_LINEARS}")
            blocks = m.transformer_blocks
            source = {"qkv": lambda: self.qkv_w,
                      "out": lambda: [b.attn1.to_out[0].weight for b in blocks],
                      "ff1": lambda: [b.ff.net[0].proj.weight for b in blocks],
                      "ff2": lambda: [b.ff.net[2].weight for b in blocks],
                      # ... and the two 3072-wide query projections that sit directly behind a LayerNorm of the video rows
                      "pq": lambda: [pc.to_q.weight for pc in m.perceiver_cross_attention] if m.is_train_face else None,
                      "aq": lambda: [al["attn"].to_q.weight for al in m.audio_model.layers] if m.is_train_audio else None}
            self.w8 = {}
            for k in FP8_LINEARS:
                ws = source[k]() if k in keep else None
                if ws is not None:
                    self.w8[k] = [ops.quantize_rows_fp8(w) for w in ws]
        pe = getattr(m.patch_embed, "pos_embedding", None)
        use_pe = (not self.cfg.use_rotary_positional_embeddings) or self.cfg.use_learned_positional_embeddings
        self.pos_embedding = pe[0] if (pe is not None and use_pe) else None
        if m.is_train_face:
            r = m.router
            qk = r.norm_q.weight.numel()
            hd = qk // r.heads
            j = torch.arange(qk, device=self.dev)
            perm = (j % hd) * r.heads + j // hd           # natural (h*128+d) position -> router's (d*16+h) index
            self.r_nq_w, self.r_nq_b = r.norm_q.weight[perm].contiguous(), r.norm_q.bias[perm].contiguous()
            self.r_nk_w, self.r_nk_b = r.norm_k.weight[perm].contiguous(), r.norm_k.bias[perm].contiguous()
            self.r_to_q = [l.weight[:, perm].contiguous() for l in r.to_q]
            self.r_to_k = [l.weight[:, perm].contiguous() for l in r.to_k]
            self.r_pos = r.pos_emb.reshape(-1, r.feat_dim).contiguous()
            self.r_qkv = []
            for st in r.spatial_temporal_layers:
                packed = {}
                for name in ("spatial_attn", "temporal_attn", "multi_id_attn"):
                    a = getattr(st, name)
                    packed[name] = (cat([a.to_q.weight, a.to_k.weight, a.to_v.weight]).contiguous(),
                                    cat([a.to_q.bias, a.to_k.bias, a.to_v.bias]).contiguous())
                if r.feat_dim == 512 and os.environ.get("BYA_ROUTER_ROWGEMM", "1") != "0":
                    # row-stationary K = 512 GEMMs: LayerNorm folded into the q|k|v / MLP weights (ops.pack_rowgemm512)
                    for name, ln in (("spatial_attn", st.norm1), ("temporal_attn", st.norm2), ("multi_id_attn", st.norm3)):
                        a = getattr(st, name)
                        w, b = packed[name]
                        if name == "spatial_attn":
                            # softmax scale * log2(e) folded into to_k (weights and bias, in fp32 before the one bf16 rounding
                            # of the packed weight): the 1350 x 1350 attention's scores are born in exp2 units, like the joint
                            # attention's (-9 % on that launch, profiles/history/r4_m_spatial_attn_probe.json)
                            c = (64 ** -0.5) * 1.4426950408889634
                            F_ = r.feat_dim
                            w, b = w.float().clone(), b.float().clone()
                            w[F_:2 * F_] *= c
                            b[F_:2 * F_] *= c
                        packed["rg_" + name] = (ops.pack_rowgemm512(w, b, ln.weight, ln.bias), ln.eps,
                                                ops.pack_rowgemm512(a.to_out[0].weight, a.to_out[0].bias))
                    packed["rg_mlp"] = (ops.pack_rowgemm512(st.mlp[0].weight, st.mlp[0].bias, st.norm4.weight, st.norm4.bias),
                                        st.norm4.eps, ops.pack_rowgemm512(st.mlp[2].weight, st.mlp[2].bias))
                self.r_qkv.append(packed)
            self.lfe_proj_t = m.local_facial_extractor.proj_out.t().contiguous()
        if m.is_train_audio:
            cw = m.audio_model.audio_proj_model.conv1.weight             # [C, C, 2] -> [C, 2*C], k = pos*C + i
            self.conv_w = cw.permute(0, 2, 1).reshape(cw.shape[0], -1).contiguous()
            # every layer's audio to_k | to_v (diffusers Attention, models/audio_model.py:247-256) reads the SAME 32 context
            # tokens per frame: stack the 2 x 42 weights so that ONE launch writes all 84 K / V tensors (n_split epilogue)
            # instead of 84 launches of 832 x 3072 x 768 (-1.3 ms per step; same products, same epilogue)
            att = [l["attn"] for l in m.audio_model.layers]
            self.audio_kv_w = cat([w for a in att for w in (a.to_k.weight, a.to_v.weight)]).contiguous()
            self.audio_kv_b = cat([b for a in att for b in (a.to_k.bias, a.to_v.bias)]).contiguous()

    # ------------------------------------------------------------------------------------------ helpers
    def _shard(self, rank, world, S, Tt, group):
        """The row partition object of this geometry; cached, because it owns the preallocated exchange buffers."""
        key = ("seq", rank, world, S, Tt, id(group))
        sh = self._parts.get(key)
        if sh is None:
            sh = self._parts[key] = SeqShard(rank, world, S, Tt, group, getattr(self.m, "_seq_p2p", None) if group is not None else None)
        return sh

    def _router_partition(self, sh, pairs, per_frame):
        key = ("router", sh.rank, sh.world, pairs, per_frame, id(sh.group))
        rp = self._parts.get(key)
        if rp is None:
            rp = self._parts[key] = RouterPartition(sh.rank, sh.world, pairs, per_frame, sh.group, sh.p2p)
        return rp

    def _buf(self, name, *shape):
        t = self._ws.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.empty(*shape, dtype=torch.bfloat16, device=self.dev)
            self._ws[name] = t
        return t

    def _attn_bound(self, i, n_bh, slots=64):
        """How the joint attention of layer ``i`` bounds its scores: ``(static bound, None)`` when the worst case over the
        layer's q/k-LayerNorm parameters is usable (or nothing is: 0 -> running maximum), else ``(0, stats)`` with a zeroed fp32
        table [slots, 2, n_bh] for this layer's q/k-norm launch to fill (``ops.qknorm_rope(stats=...)``)."""
        wc = self.score_bound[i]
        if wc <= ops.ATTN_BOUND_LIMIT or not self.device_bound or wc == 0.0:
            return wc, None
        key = ("qk_stats", slots, n_bh)
        st = self._ws.get(key)
        if st is None:
            st = self._ws[key] = torch.empty(slots, 2, n_bh, dtype=torch.float32, device=self.dev)
            self._ws["qk_flags"] = torch.zeros(max(n_bh, 64), dtype=torch.int32, device=self.dev)
        st.zero_()
        return 0.0, st

    def _qkv_norm_rope(self, i, at, xn, out, split, xq, cos, sin, text_rows, heads, q_view, k_view, stats):
        """Block ``i``'s packed q|k|v projection + q/k LayerNorm + RoPE (models/transformer.py:200-209, 241-245): ONE launch
        with the norm in the GEMM's epilogue where the library takes it (bf16 weights, no statistics wanted), else the
        projection and ``bya_qknorm_rope`` on its output -- the same bits either way."""
        fused = self.qkn_epilogue and stats is None and (self.w8 is None or "qkv" not in self.w8) and xq is None
        if fused and ops.gemm_qkv_norm_rope(xn, self.qkv_w[i], out, self.qkv_b[i], split, at.norm_q.weight, at.norm_q.bias,
                                            at.norm_k.weight, at.norm_k.bias, cos, sin, text_rows, eps=at.norm_q.eps,
                                            k_scale=self.k_scale):
            return
        self._dit_linear("qkv", i, xn, self.qkv_w[i], out, bias=self.qkv_b[i], split=split, quantised=xq)
        ops.qknorm_rope(q_view, k_view, at.norm_q.weight, at.norm_q.bias, at.norm_k.weight, at.norm_k.bias, cos, sin,
                        heads=heads, text_rows=text_rows, eps=at.norm_q.eps, k_scale=self.k_scale, stats=stats)

    def _dit_linear(self, which, i, a, w, out, quantised=None, **kw):
        """One of the four big Linears of DiT block ``i`` (models/transformer.py:241-260): the bf16 GEMM, or -- when the
        engine holds fp8 weights -- row-quantise the activations (unless the producer already did: ``quantised``) and run
        the e4m3 GEMM with the same epilogue."""
        if self.w8 is None or which not in self.w8:
            return ops.gemm(a, w, out, **kw)
        if quantised is None:
            quantised = ops.quantize_rows_fp8(a, *self._a8(a.shape))
        a8, sa = quantised
        w8, sw = self.w8[which][i]
        return ops.gemm_fp8(a8.view(*a.shape), sa.view(*a.shape[:-1]), w8, sw, out, **kw)

    def _ln_linear(self, which, i, x, xn, norm, w, out, **kw):
        """LayerNorm(x) -> Linear for the perceiver / audio query projections (models/router.py:246-253,
        models/audio_model.py:247-253): two launches in bf16; with fp8 weights the LayerNorm emits e4m3 directly."""
        if self.w8 is not None and which in self.w8 and self.fuse_ln_quant:
            xq = ops.layernorm_fp8(x, *self._a8(xn.shape), norm.weight, norm.bias, eps=norm.eps)
            return self._dit_linear(which, i, xn, w, out, quantised=xq, **kw)
        ops.layernorm(x, xn, norm.weight, norm.bias, eps=norm.eps)
        return self._dit_linear(which, i, xn, w, out, **kw)

    def _a8(self, shape):
        """Workspace for the e4m3 copy of one activation matrix and its row scales."""
        key = ("a8", tuple(shape))
        hold = self._ws.get(key)
        if hold is None:
            hold = self._ws[key] = (torch.empty(*shape, dtype=torch.uint8, device=self.dev),
                                    torch.empty(*shape[:-1], dtype=torch.float32, device=self.dev))
        return hold

    def _linear(self, x, lin_w, lin_b, out, act=None, res=None):
        """x: [(G,) M, K] -> out; weight-streaming kernel for tiny M, MFMA GEMM otherwise."""
        if x.dim() == 2 and x.shape[0] <= 8 and res is None and act in (None, "silu") and x.is_contiguous() \
                and out.is_contiguous():
            return ops.linear_small_m(x, lin_w, lin_b, out, act_out=act)
        return ops.gemm(x, lin_w, out, bias=lin_b, act=act, res=res)

    @staticmethod
    def _ln(x, out, ln, eps=None):
        return ops.layernorm(x, out, ln.weight, ln.bias, eps=ln.eps if eps is None else eps)

    def _bf(self, t):
        return t.to(device=self.dev, dtype=torch.bfloat16).contiguous()

    # ------------------------------------------------------------------------------------------ invariants
    def _face_invariants(self, id_cond, id_vit_hidden, B, n_id, after_layer=None):
        # the row counts in here (face tokens, ViT tokens, latents) are the model's, not a partition's: skinny Linears may
        # take the weight-streaming kernel (ops.weight_streaming)
        with ops.weight_streaming():
            return self._face_invariants_body(id_cond, id_vit_hidden, B, n_id, after_layer)

    def _face_invariants_body(self, id_cond, id_vit_hidden, B, n_id, after_layer=None):
        """LocalFacialExtractor (models/router.py:157-193) + the per-layer face K/V (router.py:247,254) and router
        keys (router.py:377-383).  Returns (kv[l] [B,n_id,32,2*inner], kr[l] [B,n_id,32,qk])."""
        m = self.m
        lfe = m.local_facial_extractor
        G, dim = n_id * B, lfe.dim
        nq, nt = lfe.num_queries, lfe.num_id_token
        idc = self._bf(torch.cat([id_cond[i] for i in range(n_id)], 0))                    # [(id,b), 1280]
        vit = [self._bf(torch.cat([id_vit_hidden[i][k] for i in range(n_id)], 0)) for k in range(5)]
        n_vit = vit[0].shape[1]
        E = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=self.dev)

        def mapper(seq, x, out_last):
            h = x
            for a, b in ((0, 1), (3, 4)):
                t = E(*h.shape[:-1], seq[a].weight.shape[0])
                self._linear(h, seq[a].weight, seq[a].bias, t)
                self._ln(t, t, seq[b])
                ops.act_add(t, t, act="leaky_relu")
                h = t
            return self._linear(h, seq[6].weight, seq[6].bias, out_last)

        x_tok = mapper(lfe.id_embedding_mapping, idc, E(G, nt * dim)).view(G, nt, dim)
        lat = E(G, nq + nt, dim)
        lat[:, :nq] = lfe.latents.to(torch.bfloat16)
        lat[:, nq:] = x_tok
        n_ctx, n_lat = nt + n_vit, nq + nt
        ctx = E(G, n_ctx, dim)
        ctx[:, :nt] = x_tok
        kvin = E(G, n_ctx + n_lat, dim)
        inner = lfe.heads * lfe.dim_head
        q, kv = E(G, n_lat, inner), E(G, n_ctx + n_lat, 2 * inner)
        ao, f = E(G, n_lat, inner), E(G, n_lat, dim)
        hid = E(G, n_lat, lfe.layers[0][1][1].weight.shape[0])
        for lvl in range(5):
            mapper(getattr(lfe, f"mapping_{lvl}"), vit[lvl], ctx[:, nt:])
            for attn, ff in lfe.layers[lvl * lfe.depth:(lvl + 1) * lfe.depth]:
                self._ln(ctx, kvin[:, :n_ctx], attn.norm1)
                self._ln(lat, kvin[:, n_ctx:], attn.norm2)
                ops.gemm(kvin[:, n_ctx:], attn.to_q.weight, q)
                ops.gemm(kvin, attn.to_kv.weight, kv)
                ops.self_attention(q, kv[:, :, :inner], kv[:, :, inner:], ao, heads=lfe.heads,
                                   head_dim=lfe.dim_head)
                ops.gemm(ao, attn.to_out.weight, lat, res=lat)
                self._ln(lat, f, ff[0])
                ops.gemm(f, ff[1].weight, hid, act="gelu_erf")
                ops.gemm(hid, ff[3].weight, lat, res=lat)
        face_all = E(G, nq, self.lfe_proj_t.shape[0])
        ops.gemm(lat[:, :nq], self.lfe_proj_t, face_all)
        face = face_all.view(n_id, B, nq, -1).transpose(0, 1).contiguous()                   # [B, n_id, 32, 2048]
        self.last_face_emb = face

        kvs, krs = [], []
        face2d = face.view(B * n_id * nq, -1)
        fn = E(*face2d.shape)
        for l, pc in enumerate(m.perceiver_cross_attention):
            self._ln(face2d, fn, pc.norm1)
            kv_l = E(B * n_id * nq, pc.to_kv.weight.shape[0])
            # one batch element per (sample, identity): the rows of a launch (32 face tokens) must not grow with the batch, or a
            # batch of two would pick another kernel than two batches of one (ops.weight_streaming takes <= 64 rows)
            ops.gemm(fn.view(B * n_id, nq, -1), pc.to_kv.weight, kv_l.view(B * n_id, nq, -1))
            inner_p = pc.to_q.weight.shape[0]
            kn = E(B * n_id * nq, inner_p)
            ops.layernorm(kv_l[:, :inner_p], kn, self.r_nk_w, self.r_nk_b, eps=m.router.norm_k.eps)
            kr_l = E(B * n_id * nq, inner_p)
            ops.gemm(kn.view(B * n_id, nq, -1), self.r_to_k[l], kr_l.view(B * n_id, nq, -1))
            kvs.append(kv_l.view(B, n_id, nq, -1))
            krs.append(kr_l.view(B, n_id, nq, -1))
            if after_layer is not None:
                after_layer(l)           # (the step marks its side stream here: routing layer l waits for exactly this much)
        return kvs, krs

    def _audio_invariants(self, audio_embeds, T, B, n_id):
        with ops.weight_streaming():
            return self._audio_invariants_body(audio_embeds, T, B, n_id)

    def _audio_invariants_body(self, audio_embeds, T, B, n_id):
        """sliding_windows + AudioProjModel (models/audio_model.py:188-193, 78-114) and the per-layer audio K/V
        (diffusers Attention.to_k/to_v on the 32 context tokens of each latent frame)."""
        am = self.m.audio_model
        ap = am.audio_proj_model
        mono = audio_embeds.dim() == 4
        if mono:
            # models/transformer.py:674-676, 874-878: one stream per sample, the second stream of every sample is the
            # "mute" embedding (models/audio_model.py:201-221: tests/input/ae_mute.pt cut to 4 T + 1 frames) -- here a
            # second row of the projector's batch, so the silent stream costs no extra launch
            mute = am.mute_audio_embeds
            if mute is None:
                mute = torch.load("tests/input/ae_mute.pt")           # the reference's own relative path and error
            mute = self._bf(mute[:T * 4 + 1].to(self.dev))
            if mute.shape != audio_embeds.shape[1:]:
                raise AssertionError(f"cur_audio_context_tokens.shape: {tuple(audio_embeds.shape[1:])} frames x blocks x "
                                     f"channels, mute_context_tokens.shape: {tuple(mute.shape)}")
            audio_embeds = torch.stack([self._bf(audio_embeds), mute.unsqueeze(0).expand_as(audio_embeds)], dim=1)
        a = self._bf(audio_embeds).contiguous()
        bs, ni, F, blk, ch = a.shape
        assert bs == B and ni == n_id
        assert 1 + (T - 1) * 4 + (am.window_size - am.window_stride) == F, \
            f"hidden_states_num_frames: {T}, window_size: {am.window_size}, window_stride: {am.window_stride}, " \
            f"audio_embeds.shape[1]: {F}"
        G, per = B * n_id, blk * ch
        nwin = F - am.window_size + 1
        E = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=self.dev)
        win = torch.as_strided(a, (G, nwin, am.window_size * per), (F * per, per, 1))      # overlapping windows, no copy
        h1, h2 = E(G, nwin, ap.proj1.weight.shape[0]), E(G, nwin, ap.proj2.weight.shape[0])
        ops.gemm(win, ap.proj1.weight, h1, bias=ap.proj1.bias, act="relu")
        ops.gemm(h1, ap.proj2.weight, h2, bias=ap.proj2.bias, act="relu")
        C = ap.proj3.weight.shape[0]
        cur = E(G, nwin, C)
        ops.gemm(h2, ap.proj3.weight, cur, bias=ap.proj3.bias)
        for _ in range(2):                        # Conv1d(k=2, s=2) over frames == GEMM over [x[2j], x[2j+1]] rows
            n = cur.shape[1]
            keep = n % 2
            n_out = (n - keep) // 2
            nxt = E(G, keep + n_out, C)
            if keep:
                nxt[:, 0] = cur[:, 0]
            if n_out > 0:
                pairs = torch.as_strided(cur, (G, n_out, 2 * C), (n * C, 2 * C, 1), cur.storage_offset() + keep * C)
                ops.gemm(pairs, self.conv_w, nxt[:, keep:], bias=ap.conv1.bias)
            cur = nxt
        assert cur.shape[1] == T
        tok = ap.context_tokens
        ctx = E(G * T * tok, ap.output_dim)
        self._ln(cur.view(G * T * tok, ap.output_dim), ctx, ap.norm)
        if mono:                                  # get_mute_audio_feat: + mute_learnable_tokens on the silent stream
            ctx.view(B, n_id, T, tok, ap.output_dim)[:, 1] += am.mute_learnable_tokens.view(1, 1, tok, ap.output_dim)
        self.last_audio_ctx = ctx.view(B, n_id, T, tok, ap.output_dim)
        rows, inner = G * T * tok, am.layers[0]["attn"].to_k.weight.shape[0]
        kv = E(2 * len(am.layers), rows, inner)                 # [k_0, v_0, k_1, v_1, ...], each [rows, inner] contiguous
        ops.gemm(ctx, self.audio_kv_w, kv[0], bias=self.audio_kv_b, split=(inner, rows * inner))
        ks = [kv[2 * l].view(B, n_id, T, tok, -1) for l in range(len(am.layers))]
        vs = [kv[2 * l + 1].view(B, n_id, T, tok, -1) for l in range(len(am.layers))]
        return ks, vs

    def _cached(self, name, tensors, fn):
        if not self.cache_invariants:
            return fn()
        key = _key(tensors)
        hit = self._inv_cache.get(name)
        if hit is not None and hit[0] == key:
            return hit[2]
        val = fn()
        self._inv_cache[name] = (key, list(tensors), val)    # keep the inputs alive so pointers cannot be recycled
        return val

    @torch.no_grad()
    def precompute(self, id_cond=None, id_vit_hidden=None, audio_embeds=None, latent_frames=13):
        """Explicit step-invariant cache: run the LocalFacialExtractor, every perceiver
        ``to_kv`` / router ``to_k`` and the AudioProjModel + per-layer audio K/V ONCE for the given conditioning and
        keep the results for every following ``step`` that is passed the same tensors (2.4 GB of ``conv1`` weight
        traffic and ~0.3 TFLOP per step disappear; results are bit-identical to recomputation because the same
        launches produce them).  ``release()`` drops the cache and returns to the reference's recompute-every-step."""
        self.cache_invariants = True
        self._inv_cache = {}
        n_id = self._count_ids(id_cond, audio_embeds)
        if self.m.is_train_face and id_cond is not None:
            B = id_cond[0].shape[0]
            flat = list(id_cond[:n_id]) + [t for i in range(n_id) for t in id_vit_hidden[i]]
            self._cached("face", flat, lambda: self._face_invariants(id_cond, id_vit_hidden, B, n_id))
        if self.m.is_train_audio and audio_embeds is not None:
            self._cached("audio", [audio_embeds],
                         lambda: self._audio_invariants(audio_embeds, latent_frames, audio_embeds.shape[0], n_id))
        return self

    def release(self):
        self.cache_invariants = False
        self._inv_cache = {}

    # ------------------------------------------------------------------------------------------ the step
    @torch.no_grad()
    def step(self, *args, **kwargs):
        """One denoise step (see ``_step``); every launch of the step goes to the stream that is current at entry."""
        self.refresh_if_stale()
        with ops.pinned_stream():
            return self._step(*args, **kwargs)

    def _step(self, hidden_states, encoder_hidden_states, timestep, image_rotary_emb, id_cond, id_vit_hidden,
              audio_embeds, af_matrix, routing_logits_forcing, taps=None):
        n_id = self.n_id = self._count_ids(id_cond, audio_embeds)
        m, cfg, D, H = self.m, self.cfg, self.D, self.H
        B, T, C, Hh, Ww = hidden_states.shape
        ht, wt = Hh // 2, Ww // 2
        per_frame, N = ht * wt, T * ht * wt
        Tt = encoder_hidden_states.shape[1]
        S = Tt + N
        sh = self._shard(getattr(m, "_seq_rank", 0), getattr(m, "_seq_world", 1), S, Tt, getattr(m, "_seq_group", None))
        if sh.active and B != 1:
            raise NotImplementedError("sequence-parallel execution shards ONE sample; split a CFG batch over rank groups")
        S_loc, Tt_loc, N_loc, v0, v1 = sh.S_loc, sh.Tt_loc, sh.N_loc, sh.v0, sh.v1
        key = (B, T, C, Hh, Ww, Tt, sh.rank, sh.world, n_id)
        if self._ws_key != key:
            self._ws_key, self._ws = key, {}
            if sh.p2p is not None:
                sh.p2p.drop_tables()       # the copy tables of the old workspace keep its tensors alive (a graph's are pinned)
        sh.begin_step()
        buf = self._buf
        hs = self._bf(hidden_states)
        enc_in = self._bf(encoder_hidden_states)
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep] * B)
        ts = timestep.to(device=self.dev, dtype=torch.int64).reshape(-1)
        if ts.numel() == 1 and B > 1:
            ts = ts.expand(B)
        ts = ts.contiguous()
        use_face = m.is_train_face
        use_audio = m.is_train_audio and audio_embeds is not None
        if use_face and m.router.frames * m.router.height * m.router.width != N:
            raise ValueError(f"router positional table is {m.router.frames}x{m.router.height}x{m.router.width} "
                             f"but the latents give {T}x{wt}x{ht} tokens")
        cos = sin = None
        if image_rotary_emb is not None:
            cos = image_rotary_emb[0].to(device=self.dev, dtype=torch.float32)[v0:v1].contiguous()
            sin = image_rotary_emb[1].to(device=self.dev, dtype=torch.float32)[v0:v1].contiguous()

        # ---- step-invariant conditioning (replicated on every rank: it is tiny)
        # (recomputed every step like the reference unless cached.  Their ~250 small launches -- 5 ms when they run alone --
        # go to a side stream: nothing needs them before the first routing layer, ~8 ms into the step, so they fill the CUs
        # the big kernels of patch embed and block 0 leave idle; the main stream joins right before block 0's face branch.)
        # (round 5: the main stream no longer joins the WHOLE conditioning before block 0's face branch.  Routing layer l waits
        # for the side stream's mark behind face layer l's K/V and keys, the first audio layer for the mark behind the audio
        # K/V launch, which is enqueued right behind face layer 0: at 8 ranks the conditioning -- replicated on every rank -- is
        # 7 ms of launches against a 1.9 ms DiT layer, and nothing needs face layer 20's keys before layer 40.)
        inv_side, face_marks, audio_mark = None, {}, [None]
        if (use_face or use_audio) and ops._TIMERS is None and self.side_stream_conditioning:
            if self._side is None:
                self._side = torch.cuda.Stream(self.dev)
            inv_side = ops.on_stream(self._side)
        with (inv_side if inv_side is not None else contextlib.nullcontext()):
            audio_res = []

            def audio_now():
                if use_audio and not audio_res:
                    audio_res.append(self._cached("audio", [audio_embeds], lambda: self._audio_invariants(audio_embeds, T, B, n_id)))
                    if inv_side is not None:
                        audio_mark[0] = inv_side.mark()

            def after_face_layer(l):
                if inv_side is not None:
                    face_marks[l] = inv_side.mark()
                if l == 0:
                    audio_now()
            if use_face:
                flat_face = list(id_cond[:n_id]) + [t for i in range(n_id) for t in id_vit_hidden[i]]
                face_kv, router_k = self._cached("face", flat_face,
                                                 lambda: self._face_invariants(id_cond, id_vit_hidden, B, n_id, after_face_layer))
            audio_now()
            if use_audio:
                audio_k, audio_v = audio_res[0]
        if use_audio:
            af = self._bf(af_matrix)
        forced = None
        if routing_logits_forcing is not None and use_face:
            f_in = self._bf(routing_logits_forcing).reshape(T, per_frame, n_id)
            forced = ops.forcing_max_over_frames(f_in, buf("forced", 1, N, n_id).view(T, per_frame, n_id), T, per_frame,
                                                 n_id).view(1, N, n_id)[:, v0:v1].contiguous()

        # ---- D0: timestep embedding and every AdaLN modulation vector of the step
        tfeat = ops.timestep_features(ts, buf("tfeat", B, D), cfg.flip_sin_to_cos, float(cfg.freq_shift))
        te = m.time_embedding
        e1 = ops.linear_small_m(tfeat, te.linear_1.weight, te.linear_1.bias, buf("e1", B, te.linear_1.weight.shape[0]),
                                act_out="silu")
        emb = ops.linear_small_m(e1, te.linear_2.weight, te.linear_2.bias, buf("emb", B, te.linear_2.weight.shape[0]))
        Nm = self.mod_w.shape[0]
        if sh.active and sh.p2p is not None and B == 1 and self.sp_shard_mods:
            # (r6) sharded step: the AdaLN vector -- 1.55 M outputs against a 1.6 GB weight, 0.6 ms of weight streaming that every
            # rank used to repeat before its first LayerNorm -- is computed in column slices, one per rank, and gathered by ONE
            # exchange (3 MB).  Same bits: the weight-streaming kernel's outputs are independent of one another.  The receive
            # buffer is re-used every step: a peer starts its next step only behind its output gather, which this rank joins
            # after its last use of the vector.
            W, r = sh.world, sh.rank
            cut = [(Nm * j // W) // 8 * 8 for j in range(W)] + [Nm]
            part = buf("mods_part", 1, cut[r + 1] - cut[r])
            ops.linear_small_m(emb, self.mod_w[cut[r]:cut[r + 1]], self.mod_b[cut[r]:cut[r + 1]], part, silu_in=True)
            mods = sh.p2p.symmetric(f"mods:{(1, Nm)}", (1, Nm), torch.bfloat16)
            sh._exchange(("mods", Nm), [(part[0], j, f"mods:{(1, Nm)}", cut[r]) for j in range(W)])
        else:
            mods = ops.linear_small_m(emb, self.mod_w, self.mod_b, buf("mods", B, Nm), silu_in=True)
        mbs = mods.stride(0)

        # ---- D1: patch embed into the joint stream x = [text | video] (this rank's rows only)
        x = buf("x", B, S_loc, D)
        xv = x[:, Tt_loc:]
        pe = self.pos_embedding
        if pe is not None and pe.shape[0] != S:
            raise ValueError("learned positional embeddings need the configured sample height/width/frames")
        if Tt_loc:
            tp = m.patch_embed.text_proj
            ops.gemm(enc_in[:, sh.r0:sh.r0 + Tt_loc], tp.weight, x[:, :Tt_loc], bias=tp.bias,
                     res=None if pe is None else pe[sh.r0:sh.r0 + Tt_loc])
        cols = ops.patchify(hs, buf("cols", B, N, C * 4))
        ops.gemm(cols[:, v0:v1], self.patch_w, xv, bias=m.patch_embed.proj.bias,
                 res=None if pe is None else pe[Tt + v0:Tt + v1])
        if taps is not None:
            taps["emb"], taps["embed"] = emb.clone(), x.clone()

        xn = buf("xn", B, S_loc, D)
        qkv = buf("qkv", 3, B, S_loc, D)
        q, k, v = qkv[0], qkv[1], qkv[2]
        ff = buf("ff", B, S_loc, 4 * D)
        head_parallel = sh.active and H % sh.world == 0 and not self.sp_allgather
        if head_parallel:
            W = sh.world
            Dl = D // W
            qkvb = buf("qkv_blocks", 3 * W, S_loc, Dl)         # column block t*W + j: tensor t (q,k,v), heads of rank j
            qh, kh, vh = buf("q_heads", S, Dl), buf("k_heads", S, Dl), buf("v_heads", S, Dl)
            oh = buf("o_heads", S, Dl)
        elif sh.active:
            k_full, v_full = buf("k_full", 1, S, D), buf("v_full", 1, S, D)
        r_logits = None
        for i, blk in enumerate(m.transformer_blocks):
            # ---- D2..D4: CogVideoXBlock (models/transformer.py:223-262)
            for half, nz in enumerate((blk.norm1, blk.norm2)):
                mo = mods[:, (2 * i + half) * 6 * D:]
                # chunk order: shift, scale, gate, enc_shift, enc_scale, enc_gate
                ln_kw = dict(eps=nz.norm.eps, shift0=mo[:, 3 * D:], scale0=mo[:, 4 * D:], shift1=mo, scale1=mo[:, D:],
                             split=Tt_loc, mod_batch_stride=mbs)
                xq = None
                if self.w8 is not None and ("qkv", "ff1")[half] in self.w8 and self.fuse_ln_quant:
                    # the AdaLN output feeds exactly one Linear (q|k|v, or the MLP's first): emit it in e4m3 directly
                    xq = ops.layernorm_fp8(x, *self._a8(xn.shape), nz.norm.weight, nz.norm.bias, **ln_kw)
                else:
                    ops.layernorm(x, xn, nz.norm.weight, nz.norm.bias, **ln_kw)
                if half == 0:
                    at = blk.attn1
                    if head_parallel:
                        # exchange A, head-parallel: the projection writes per-destination column blocks, q/k-norm +
                        # RoPE run on the local rows, then rows are traded for heads (every element moves once)
                        if sh.p2p is None:
                            self._dit_linear("qkv", i, xn[0], self.qkv_w[i], qkvb[0], bias=self.qkv_b[i], split=(Dl, S_loc * Dl),
                                             quantised=None if xq is None else (xq[0][0], xq[1][0]))
                        # v needs no norm: its exchange runs on the RCCL stream underneath q's norm + RoPE, q's exchange
                        # underneath k's; only k's is exposed (every exchange is enqueued on the communicator's stream,
                        # the compute stream waits by event right before the attention launch)
                        qk_kw = dict(heads=H // W, text_rows=Tt_loc if cos is not None else S_loc, eps=at.norm_q.eps,
                                     k_scale=self.k_scale)
                        if sh.p2p is not None:
                            # P2P transport: q/k-norm + RoPE on the local rows of both, then ONE push kernel carries all
                            # 3 x W column blocks to their places in the peers' q / k / v buffers
                            # (device bound: this rank's rows give its partial maxima for ALL heads; the tables travel with
                            # the q|k|v exchange and the attention takes the maximum over the ranks' tables for its heads)
                            sb, st = self._attn_bound(i, H, slots=max(1, 64 // W))
                            if (self.sp_overlap_v and st is None and xq is None and (self.w8 is None or "qkv" not in self.w8)
                                    and S_loc >= self.SP_OVERLAP_V_MIN_ROWS):
                                # (r6) v FIRST: v needs no norm -- its projection is a launch of its own, its column blocks travel
                                # on the side stream while the q | k projection (with the norm in its epilogue) runs; only the
                                # q | k exchange is exposed.  Same bits: every output element is summed over K in the same order
                                ops.gemm(xn[0], self.qkv_w[i][2 * D:], qkvb[2 * W], bias=self.qkv_b[i][2 * D:], split=(Dl, S_loc * Dl))
                                sh.push_v_heads(qkvb[2 * W:])
                                if not (self.qkn_epilogue and ops.gemm_qkv_norm_rope(
                                        xn[0], self.qkv_w[i][:2 * D], qkvb[0], self.qkv_b[i][:2 * D], (Dl, S_loc * Dl), at.norm_q.weight,
                                        at.norm_q.bias, at.norm_k.weight, at.norm_k.bias, cos, sin, qk_kw["text_rows"], eps=at.norm_q.eps,
                                        k_scale=self.k_scale, tensors=2)):
                                    ops.gemm(xn[0], self.qkv_w[i][:2 * D], qkvb[0], bias=self.qkv_b[i][:2 * D], split=(Dl, S_loc * Dl))
                                    ops.qknorm_rope(qkvb[:W], qkvb[W:2 * W], at.norm_q.weight, at.norm_q.bias, at.norm_k.weight,
                                                    at.norm_k.bias, cos, sin, heads=H // W, text_rows=qk_kw["text_rows"],
                                                    eps=at.norm_q.eps, k_scale=self.k_scale)
                                qh_, kh_, vh_ = sh.rows_to_heads_qk(qkvb[:2 * W])
                                st_all = None
                            else:
                                self._qkv_norm_rope(i, at, xn[0], qkvb[0], (Dl, S_loc * Dl), None if xq is None else (xq[0][0], xq[1][0]),
                                                    cos, sin, qk_kw["text_rows"], H // W, qkvb[:W], qkvb[W:2 * W], st)
                                qh_, kh_, vh_, st_all = sh.rows_to_heads_qkv(qkvb, st)
                            ops.self_attention(qh_[None], kh_[None], vh_[None], oh[None], heads=H // W, tag="joint", prescaled=True,
                                               score_bound=sb, bound=None if st is None else (st_all, sh.rank * (H // W), self._ws["qk_flags"]))
                            xo = sh.heads_to_rows(oh)              # (the symmetric receive buffer, already in [rows, heads] order)
                            self._dit_linear("out", i, xo[None], at.to_out[0].weight, x, bias=at.to_out[0].bias, res=x,
                                             gate0=mo[:, 5 * D:], gate1=mo[:, 2 * D:], gate_split=Tt_loc, gate_batch_stride=mbs)
                            continue
                        pending = [sh.rows_to_heads(qkvb[2 * W:], vh, async_op=True)]
                        ops.qknorm_rope(qkvb[:W], None, at.norm_q.weight, at.norm_q.bias, at.norm_k.weight, at.norm_k.bias,
                                        cos, sin, **qk_kw)
                        pending.append(sh.rows_to_heads(qkvb[:W], qh, async_op=True))
                        ops.qknorm_rope(None, qkvb[W:2 * W], at.norm_q.weight, at.norm_q.bias, at.norm_k.weight,
                                        at.norm_k.bias, cos, sin, **qk_kw)
                        pending.append(sh.rows_to_heads(qkvb[W:2 * W], kh, async_op=True))
                        for h in pending:
                            if h is not None:
                                h.wait()            # the compute stream waits (no host synchronisation)
                        ops.self_attention(qh[None], kh[None], vh[None], oh[None], heads=H // W, tag="joint", prescaled=True,
                                           score_bound=self.score_bound[i])
                        sh.heads_to_rows(oh, xn[0])
                        self._dit_linear("out", i, xn, at.to_out[0].weight, x, bias=at.to_out[0].bias, res=x, gate0=mo[:, 5 * D:],
                                 gate1=mo[:, 2 * D:], gate_split=Tt_loc, gate_batch_stride=mbs)
                        continue
                    # (rows of K from other ranks are not in this rank's statistics: the all-gather form keeps the worst case)
                    sb, st = self._attn_bound(i, B * H) if not sh.active else (self.score_bound[i], None)
                    self._qkv_norm_rope(i, at, xn, q, (D, B * S_loc * D), xq, cos, sin, Tt_loc if cos is not None else S_loc,
                                        H, q, k, st)
                    if sh.active:      # exchange A (fallback when heads % world != 0): all-gather K and V
                        if sh.p2p is not None:
                            sh.gather_rows_many([k[0], v[0]], [k_full[0], v_full[0]])       # one exchange, double-buffered
                        else:
                            sh.gather_rows(k[0], k_full[0])
                            sh.gather_rows(v[0], v_full[0])
                        ops.self_attention(q, k_full, v_full, xn, heads=H, tag="joint", prescaled=True,
                                           score_bound=self.score_bound[i])
                    else:
                        ops.self_attention(q, k, v, xn, heads=H, tag="joint", prescaled=True, score_bound=sb,
                                           bound=None if st is None else (st, 0, self._ws["qk_flags"]))
                    self._dit_linear("out", i, xn, at.to_out[0].weight, x, bias=at.to_out[0].bias, res=x, gate0=mo[:, 5 * D:],
                             gate1=mo[:, 2 * D:], gate_split=Tt_loc, gate_batch_stride=mbs)
                else:
                    self._dit_linear("ff1", i, xn, blk.ff.net[0].proj.weight, ff, bias=blk.ff.net[0].proj.bias, act="gelu_tanh",
                                     quantised=xq)
                    self._dit_linear("ff2", i, ff, blk.ff.net[2].weight, x, bias=blk.ff.net[2].bias, res=x, gate0=mo[:, 5 * D:],
                             gate1=mo[:, 2 * D:], gate_split=Tt_loc, gate_batch_stride=mbs)
            if taps is not None:
                taps[f"block{i}"] = x.clone()

            # ---- P1 + R1..R4 + G1: face routing (models/transformer.py:737-833)
            if use_face and i % m.cross_attn_interval == 0:
                ca = i // m.cross_attn_interval
                ops.on_stream.need(face_marks.pop(ca, None))     # the side stream has produced this layer's face K/V and keys
                pc = m.perceiver_cross_attention[ca]
                inner_p = pc.to_q.weight.shape[0]
                hd_p = inner_p // 16
                lat = buf("lat", B, N_loc, D)
                qp = self._ln_linear("pq", ca, xv, lat, pc.norm2, pc.to_q.weight, buf("qp", B, N_loc, inner_p))
                kv_l = face_kv[ca]
                ntok = kv_l.shape[2]
                fuse_face = self.fused_attn_mix and taps is None
                pout = None if fuse_face else buf("pout", B, n_id, N_loc, inner_p)

                def perceiver_attention():
                    if fuse_face:
                        return                       # runs behind the router, with the mix in its epilogue (below)
                    ops.attention(qp, kv_l, kv_l[..., inner_p:], pout, head_dim=hd_p, heads=16, nb1=B, nb2=n_id, Sq=N_loc,
                                  Skv=ntok, q_strides=(N_loc * inner_p, 0, inner_p),
                                  k_strides=(n_id * ntok * 2 * inner_p, ntok * 2 * inner_p, 2 * inner_p),
                                  v_strides=(n_id * ntok * 2 * inner_p, ntok * 2 * inner_p, 2 * inner_p),
                                  o_strides=(n_id * N_loc * inner_p, N_loc * inner_p, inner_p), scale=hd_p ** -0.5)
                if forced is None:
                    # (sharded: the perceiver attention -- independent of the router -- is enqueued while the router's first
                    # repartition exchange is in flight on the RCCL stream)
                    r_logits = self._router(qp, router_k[ca], ca, B, T, per_frame, n_id, sh, taps, perceiver_attention)
                else:
                    perceiver_attention()
                    r_logits = forced
                if fuse_face:
                    # Perceiver attention with the masked combine in its epilogue: z = sum_id r[n, id] * attention_id
                    z = buf("zmix_p", B, N_loc, inner_p)
                    kvs_p = (ntok * 2 * inner_p, 0, 2 * inner_p)
                    for b in range(B):
                        rb = r_logits[b if r_logits.shape[0] > 1 else 0]
                        ops.attn_kv_mix(qp[b], kv_l[b], kv_l[b][..., inner_p:], rb, None, z[b], head_dim=hd_p, heads=16,
                                        n_id=n_id, n_grp=1, Sq=N_loc, Skv=ntok, q_strides=(0, inner_p), k_strides=kvs_p,
                                        v_strides=kvs_p, z_strides=(0, inner_p), scale=hd_p ** -0.5)
                    ops.gemm(z, pc.to_out.weight, xv, res=xv, alpha=m.local_face_scale)
                elif self.mix_before_projection:
                    # to_out is linear and bias-free: route first, project once (half the GEMM, no feat round trip)
                    z = ops.routed_mix(pout, r_logits, None, "face", buf("zmix_p", B, N_loc, inner_p))
                    ops.gemm(z, pc.to_out.weight, xv, res=xv, alpha=m.local_face_scale)
                else:
                    feat = buf("feat", B, n_id, N_loc, D)
                    ops.gemm(pout.view(B * n_id, N_loc, inner_p), pc.to_out.weight, feat.view(B * n_id, N_loc, D))
                    if taps is not None:
                        taps[f"id_feat{ca}"] = feat[0].clone()
                    ops.masked_combine(xv, feat, r_logits, None, "face", alpha=m.local_face_scale)
                if taps is not None:
                    taps[f"face{i}"] = xv.clone()

            # ---- A1 + G2: audio injection (models/transformer.py:858-936)
            if use_audio and i % m.audio_attn_interval == 0:
                ops.on_stream.need(audio_mark[0])
                audio_mark[0] = None
                al = m.audio_model.layers[i // m.audio_attn_interval]
                at = al["attn"]
                an = buf("lat", B, N_loc, D)
                qa = self._ln_linear("aq", i // m.audio_attn_interval, xv, an, al["norm_q"], at.to_q.weight,
                                     buf("qa", B, N_loc, D), bias=at.to_q.bias)
                ka, va = audio_k[i // m.audio_attn_interval], audio_v[i // m.audio_attn_interval]
                ntok = ka.shape[3]
                fuse_audio = self.fused_attn_mix and taps is None
                kvs = (T * ntok * D, ntok * D, D)
                if fuse_audio:
                    # audio cross-attention with the masked combine in its epilogue (per sample: af differs per sample)
                    wsum = self._ws.get("wsum")
                    if wsum is None or wsum.numel() != B * N_loc:
                        wsum = self._ws["wsum"] = torch.empty(B, N_loc, dtype=torch.float32, device=self.dev)
                    z = buf("zmix_a", B, N_loc, D)
                    for b in range(B):
                        rb = r_logits[b if r_logits.shape[0] > 1 else 0]
                        if not sh.active:
                            ops.attn_kv_mix(qa[b], ka[b], va[b], rb, af[b], z[b], wsum[b], head_dim=64, heads=H, n_id=n_id,
                                            n_grp=T, Sq=per_frame, Skv=ntok, q_strides=(per_frame * D, D), k_strides=kvs,
                                            v_strides=kvs, z_strides=(per_frame * D, D), scale=64 ** -0.5)
                        else:           # shard boundaries cut frames: one launch per (partial) frame of this rank
                            for f, start, length in sh.frame_segments(per_frame):
                                ops.attn_kv_mix(qa[b, start:], ka[b, :, f], va[b, :, f], rb[start:start + length], af[b],
                                                z[b, start:], wsum[b, start:], head_dim=64, heads=H, n_id=n_id, n_grp=1,
                                                Sq=length, Skv=ntok, q_strides=(0, D), k_strides=(kvs[0], 0, D),
                                                v_strides=(kvs[0], 0, D), z_strides=(0, D), scale=64 ** -0.5)
                    ops.gemm(z, at.to_out[0].weight, xv, bias=at.to_out[0].bias, res=xv, bias_rowscale=wsum)
                    continue
                ao = buf("ao", B, n_id, N_loc, D)
                for b in range(B):      # (id, frame) batch of one sample; q rows shared by both ids
                    if not sh.active:
                        ops.attention(qa[b], ka[b], va[b], ao[b], head_dim=64, heads=H, nb1=n_id, nb2=T, Sq=per_frame,
                                      Skv=ntok, q_strides=(0, per_frame * D, D), k_strides=kvs, v_strides=kvs,
                                      o_strides=(N_loc * D, per_frame * D, D), scale=64 ** -0.5)
                    else:               # shard boundaries cut frames: one launch per (partial) frame of this rank
                        for f, start, length in sh.frame_segments(per_frame):
                            ops.attention(qa[b, start:], ka[b, :, f], va[b, :, f], ao[b, :, start:], head_dim=64,
                                          heads=H, nb1=n_id, nb2=1, Sq=length, Skv=ntok, q_strides=(0, 0, D),
                                          k_strides=kvs, v_strides=kvs, o_strides=(N_loc * D, 0, D), scale=64 ** -0.5)
                if self.mix_before_projection:
                    wsum = self._ws.get("wsum")
                    if wsum is None or wsum.numel() != B * N_loc:
                        wsum = self._ws["wsum"] = torch.empty(B, N_loc, dtype=torch.float32, device=self.dev)
                    z = ops.routed_mix(ao, r_logits, af, "audio", buf("zmix_a", B, N_loc, D), wsum)
                    ops.gemm(z, at.to_out[0].weight, xv, bias=at.to_out[0].bias, res=xv, bias_rowscale=wsum)
                else:
                    feat = buf("feat", B, n_id, N_loc, D)
                    ops.gemm(ao.view(B * n_id, N_loc, D), at.to_out[0].weight, feat.view(B * n_id, N_loc, D),
                             bias=at.to_out[0].bias)
                    ops.masked_combine(xv, feat, r_logits, af, "audio")
                if taps is not None:
                    taps[f"audio{i}"] = xv.clone()

        if inv_side is not None:
            inv_side.join()                  # (whatever of the conditioning no layer waited for: the step ends behind all of it)
        # ---- F1: final norm, AdaLN head, projection, unpatchify (models/transformer.py:938-957)
        xf = buf("lat", B, N_loc, D)
        ops.layernorm(xv, xf, m.norm_final.weight, m.norm_final.bias, eps=m.norm_final.eps)
        mo = mods[:, 2 * self.L * 6 * D:]                      # AdaLayerNorm chunk_dim=1: (shift, scale)
        xo = buf("qa", B, N_loc, D)
        ops.layernorm(xf, xo, m.norm_out.norm.weight, m.norm_out.norm.bias, eps=m.norm_out.norm.eps, shift0=mo,
                      scale0=mo[:, D:], shift1=mo, scale1=mo[:, D:], split=0, mod_batch_stride=mbs)
        co = m.proj_out.weight.shape[0]
        y = ops.gemm(xo, m.proj_out.weight, buf("y", B, N_loc, co), bias=m.proj_out.bias)
        if sh.active:                                        # every rank returns the full latent prediction
            y = sh.gather_video_rows(y, out=buf("y_full", B, N, co), in_place=True)
            if self.verify_exchanges:
                sh.verify_gathered(y)          # every rank holds the same gathered prediction -- or a receive buffer went stale
        out = torch.empty(B, T, co // 4, Hh, Ww, dtype=torch.bfloat16, device=self.dev)
        ops.unpatchify(y, out)
        if sh.p2p is not None:
            sh.p2p.poison(out)             # NaN instead of numbers if any wait of this group ever gave up (sticky)
        return out

    def _r_lnlin(self, x, tmp, ln, pk, key, w, b, out, act=None):
        """out = act(Linear(LayerNorm(x))) on router rows: one fused row GEMM when packed, else LN + GEMM."""
        rg = pk.get(key)
        if rg is not None:
            return ops.rowgemm512(x, rg[0], out, act=act, eps=rg[1])
        self._ln(x, tmp, ln)
        return ops.gemm(tmp, w, out, bias=b, act=act)

    def _r_group_attn(self, x, tmp, qkv, out, ln, pk, name, L, heads, n_outer, n_inner, outer_stride, seq_stride):
        """out = Attention over groups of L rows (temporal: the T frames of a location; multi-ID: the identities of a token)
        of LN(x), up to the out-projection.  One fused launch (ops.router_group_attn: the q|k|v tensor stays on chip) when
        the group fits the two 16-row MFMA tiles of a wave (32 rows) and the folded weights exist, else LN -> q|k|v GEMM -> attn_tiny."""
        rg = pk.get("rg_" + name)
        hd = 64
        if rg is not None and L <= 32 and heads == 8 and self.router_fused_attn:
            return ops.router_group_attn(x, rg[0], out, L, n_outer, n_inner, outer_stride, seq_stride, eps=rg[1],
                                         scale=hd ** -0.5)
        F = x.shape[1]
        self._r_lnlin(x, tmp, ln, pk, "rg_" + name, *pk[name], qkv)
        return ops.attn_tiny(qkv, qkv[:, F:], qkv[:, 2 * F:], out, L, heads, n_outer, n_inner, outer_stride, seq_stride,
                             3 * F, F, hd ** -0.5)

    # row ranges in which a chain beats its two launches (profiles/r6_a_router_chain_probe.json, same box, us: MLP 50.5 -> 39.0 at
    # 17550 rows, 37.8 -> 28.9 at 8788, 69.2 -> 69.8 at 35100, 23.8 -> 26.8 at 4394; multi-ID 64.3 -> 60.2, 47.1 -> 43.2, 93.4 ->
    # 109, 28.6 -> 41.4; temporal level or slower everywhere)
    # below this many rows per rank the two smaller projection launches cost more than the hidden link time buys (2222 rows of
    # an 8-rank step: 108 + 216 tiles of 256 x 256 on 256 CUs instead of 324)
    SP_OVERLAP_V_MIN_ROWS = 4096

    ROUTER_CHAIN_ROWS = {"mlp": (6000, 24000), "multi_id_attn": (6000, 24000), "temporal_attn": (1, 0)}

    def _chain_ok(self, rows, kind):
        lo, hi = self.ROUTER_CHAIN_ROWS[kind]
        return self.router_chain == "all" or (self.router_chain != "0" and lo <= rows <= hi)

    def _r_attn_block(self, x, tmp, qkv, a, ln, pk, name, to_out, L, heads, n_outer, n_inner, outer_stride, seq_stride):
        """x += to_out(Attention over groups of L rows of LN(x)): one sub-block of SpatialTemporalAttentionBlock (temporal /
        multi-ID, models/router.py:482-487).  One chain launch where it pays, else group attention + out-projection."""
        rg = pk.get("rg_" + name)
        if rg is not None and L <= 16 and heads == 8 and self.router_fused_attn and self._chain_ok(x.shape[0], name):
            return ops.router_group_attn_out(x, rg[0], rg[2], L, n_outer, n_inner, outer_stride, seq_stride, eps=rg[1],
                                             scale=64 ** -0.5)
        self._r_group_attn(x, tmp, qkv, a, ln, pk, name, L, heads, n_outer, n_inner, outer_stride, seq_stride)
        return self._r_linres(a, pk, "rg_" + name, to_out, x)

    def _r_mlp_block(self, x, tmp, h, st, pk):
        """x += mlp(norm4(x)) (models/router.py:491): one chain launch where it pays, else the two row GEMMs."""
        rg = pk.get("rg_mlp")
        if rg is not None and rg[0]["w"].shape[0] == 512 and self._chain_ok(x.shape[0], "mlp"):      # (hidden width 512: mlp_ratio 1)
            return ops.router_mlp_fused(x, rg[0], rg[2], eps=rg[1])
        self._r_lnlin(x, tmp, st.norm4, pk, "rg_mlp", st.mlp[0].weight, st.mlp[0].bias, h, act="gelu_erf")
        return self._r_linres(h, pk, "rg_mlp", st.mlp[2], x)

    def _r_linres(self, a, pk, key, lin, x):
        """x += Linear(a) on router rows."""
        rg = pk.get(key)
        if rg is not None:
            return ops.rowgemm512(a, rg[2], x, res=x)
        return ops.gemm(a, lin.weight, x, bias=lin.bias, res=x)

    def _router(self, qp, kr, ca, B, T, per_frame, n_id, sh, taps, overlap):
        """MultiIPRouter.forward (models/router.py:364-411) on the perceiver's q (shared by both ids) and the
        pre-projected router keys.  Returns this rank's rows of the routing logits, [B, N_loc, n_id] (sigmoid)."""
        m, buf = self.m, self._buf
        r = m.router
        N, N_loc = T * per_frame, sh.N_loc
        F = r.feat_dim
        qk = qp.shape[-1]
        if sh.active and n_id * T >= sh.world and per_frame >= sh.world and not self.router_replicated:
            return self._router_sharded(qp, kr, ca, T, per_frame, n_id, sh, taps, overlap)
        overlap()
        qn = buf("r_qn", B, N_loc, qk)
        ops.layernorm(qp, qn, self.r_nq_w, self.r_nq_b, eps=r.norm_q.eps)
        qr = ops.gemm(qn, self.r_to_q[ca], buf("r_qr", B, N_loc, qk))
        rs = buf("r_s_loc", B, n_id, N_loc, F)
        pos = self.r_pos[sh.v0:sh.v1]
        for b in range(B):
            ops.router_scores(qr[b], kr[b].contiguous(), r.norm.weight, r.norm.bias, pos, rs[b], n_id, N_loc,
                              eps=r.norm.eps)
        # exchange B: the spatial / temporal attentions mix tokens across the whole clip -> gather the 512-wide
        # router rows (36 MB) and run the four small blocks replicated
        if sh.active:
            rs = sh.gather_video_rows(rs, out=buf("r_s_full", B, n_id, N, F))
        R = B * n_id * N
        rs2, rn = rs.view(R, F), buf("r_n", R, F)
        qkv, ra, rh = buf("r_qkv", R, 3 * F), buf("r_a", R, F), buf("r_h", R, F)
        hd = 64
        heads = F // hd
        for st, pk in zip(r.spatial_temporal_layers, self.r_qkv):
            # 1. spatial: every (sample, id, frame) attends over its per_frame tokens
            self._r_lnlin(rs2, rn, st.norm1, pk, "rg_spatial_attn", *pk["spatial_attn"], qkv)
            ops.attention(qkv, qkv[:, F:], qkv[:, 2 * F:], ra, head_dim=hd, heads=heads, nb1=B * n_id * T, nb2=1,
                          Sq=per_frame, Skv=per_frame, q_strides=(per_frame * 3 * F, 0, 3 * F),
                          k_strides=(per_frame * 3 * F, 0, 3 * F), v_strides=(per_frame * 3 * F, 0, 3 * F),
                          o_strides=(per_frame * F, 0, F), scale=hd ** -0.5, prescaled="rg_spatial_attn" in pk)
            self._r_linres(ra, pk, "rg_spatial_attn", st.spatial_attn.to_out[0], rs2)
            # 2. temporal: every (sample, id, location) attends over its T frames
            self._r_attn_block(rs2, rn, qkv, ra, st.norm2, pk, "temporal_attn", st.temporal_attn.to_out[0], T, heads,
                               B * n_id, per_frame, N, per_frame)
            # 3. multi-ID: every (sample, token) attends over the ids
            self._r_attn_block(rs2, rn, qkv, ra, st.norm3, pk, "multi_id_attn", st.multi_id_attn.to_out[0], n_id, heads,
                               B, N, n_id * N, N)
            # 4. MLP (GELU erf)
            self._r_mlp_block(rs2, rn, rh, st, pk)
        logits = buf("r_logits", B, N, n_id)
        fp = r.final_proj[0]
        rs4 = rs2.view(B, n_id, N, F)
        for b in range(B):
            ops.router_head(rs4[b], fp.weight, fp.bias, logits[b], n_id, N)
        if taps is not None:
            for b in range(B):
                taps[f"router{ca}_b{b}"] = logits[b:b + 1].clone()
        return logits if not sh.active else logits[:, sh.v0:sh.v1].contiguous()

    def _router_sharded(self, qp, kr, ca, T, per_frame, n_id, sh, taps, overlap):
        """Multi-GPU form of ``_router`` (B = 1): the four SpatialTemporalAttentionBlocks run SHARDED.  Spatial attention in
        the frame-major partition (whole (id, frame) pairs per rank), everything else in the location-major partition
        (a range of within-frame locations for all frames and ids per rank), one all-to-all between them
        (parallel.RouterPartition); the sigmoid logits (70 KB) are all-gathered at the end."""
        m, buf = self.m, self._buf
        r = m.router
        N, N_loc, F = T * per_frame, sh.N_loc, r.feat_dim
        pairs = n_id * T
        rp = self._router_partition(sh, pairs, per_frame)
        qk = qp.shape[-1]
        qn = buf("r_qn", 1, N_loc, qk)
        ops.layernorm(qp, qn, self.r_nq_w, self.r_nq_b, eps=r.norm_q.eps)
        qr = ops.gemm(qn, self.r_to_q[ca], buf("r_qr", 1, N_loc, qk))
        rs_loc = buf("r_s_loc", 1, n_id, N_loc, F)
        ops.router_scores(qr[0], kr[0].contiguous(), r.norm.weight, r.norm.bias, self.r_pos[sh.v0:sh.v1], rs_loc[0],
                          n_id, N_loc, eps=r.norm.eps)
        if rp.p2p is not None:
            xa = rp.tokens_to_a(rs_loc[0], sh.v0, T)                                   # token ranges -> the owners of their pairs
        else:
            rs_full = sh.gather_video_rows(rs_loc, out=buf("r_s_full", 1, n_id, N, F))     # token ranges -> everyone (36 MB, once)
            xa = buf("rp_xa", rp.nPA, per_frame, F)
            xa.copy_(rs_full.view(pairs, per_frame, F)[rp.pa0:rp.pa1])
        RA, RB = rp.nPA * per_frame, pairs * rp.nLB
        rn_a, qkv_a, ra_a = buf("rp_rn_a", RA, F), buf("rp_qkv_a", RA, 3 * F), buf("rp_ra_a", RA, F)
        xb = rp.recv_buf("rp_xb", (pairs, rp.nLB, F), xa)          # (P2P transport: the peers store into it directly)
        rn_b, qkv_b, ra_b, rh_b = buf("rp_rn_b", RB, F), buf("rp_qkv_b", RB, 3 * F), buf("rp_ra_b", RB, F), buf("rp_rh_b", RB, F)
        hd = 64
        heads = F // hd
        nblk = len(r.spatial_temporal_layers)
        for bi, (st, pk) in enumerate(zip(r.spatial_temporal_layers, self.r_qkv)):
            # ---- frame-major: spatial attention over the per_frame tokens of each local (id, frame) pair
            xa2 = xa.view(RA, F)
            self._r_lnlin(xa2, rn_a, st.norm1, pk, "rg_spatial_attn", *pk["spatial_attn"], qkv_a)
            ops.attention(qkv_a, qkv_a[:, F:], qkv_a[:, 2 * F:], ra_a, head_dim=hd, heads=heads, nb1=rp.nPA, nb2=1,
                          Sq=per_frame, Skv=per_frame, q_strides=(per_frame * 3 * F, 0, 3 * F),
                          k_strides=(per_frame * 3 * F, 0, 3 * F), v_strides=(per_frame * 3 * F, 0, 3 * F),
                          o_strides=(per_frame * F, 0, F), scale=hd ** -0.5, prescaled="rg_spatial_attn" in pk)
            self._r_linres(ra_a, pk, "rg_spatial_attn", st.spatial_attn.to_out[0], xa2)
            # ---- location-major: temporal, multi-ID, MLP
            rp.a_to_b(xa, xb, overlap=overlap if bi == 0 else None)
            xb2 = xb.view(RB, F)
            self._r_attn_block(xb2, rn_b, qkv_b, ra_b, st.norm2, pk, "temporal_attn", st.temporal_attn.to_out[0], T, heads,
                               n_id, rp.nLB, T * rp.nLB, rp.nLB)
            self._r_attn_block(xb2, rn_b, qkv_b, ra_b, st.norm3, pk, "multi_id_attn", st.multi_id_attn.to_out[0], n_id, heads,
                               1, T * rp.nLB, 0, T * rp.nLB)
            self._r_mlp_block(xb2, rn_b, rh_b, st, pk)
            if bi + 1 < nblk:
                xa = rp.b_to_a(xb, xa)
        fp = r.final_proj[0]
        lb = buf("rp_logits_b", T * rp.nLB, n_id)
        ops.router_head(xb.view(n_id, T * rp.nLB, F), fp.weight, fp.bias, lb, n_id, T * rp.nLB)
        logits = rp.gather_b_rows(lb.view(T, rp.nLB, n_id), out=buf("rp_logits", T, per_frame, n_id)).view(1, N, n_id)
        if taps is not None:
            taps[f"router{ca}_b0"] = logits.clone()
        return logits[:, sh.v0:sh.v1].contiguous()
