"""Token-axis (sequence) sharding of one denoise step across the GPUs of a node: one process per GPU,
``torch.distributed`` with the ``nccl`` backend (= RCCL over xGMI on ROCm); ``gloo`` in the CPU tests.

The reference has no inference parallelism at all; this is new capability required by
BASELINE.json.  Partition: the joint sequence ``[text (Tt) | video (N)]`` of S = Tt + N rows is cut into ``world``
contiguous row ranges of equal size (S = 17776 divides by 2, 4 and 8) or sizes differing by one row (other geometries,
e.g. S = 47026 of BASELINE configs[3]).  Everything on the path is row-local EXCEPT:

  * joint self-attention, HEAD-PARALLEL (``SeqShard.rows_to_heads`` / ``heads_to_rows``): the packed q|k|v projection
    writes one column block per destination rank, three all-to-alls trade this rank's rows of every head for all rows
    of this rank's 48 / world heads, attention runs on whole sequences of the local heads, one all-to-all trades back.
    Every element of q, k, v and of the output crosses one xGMI link once (48 MB received per rank and layer at 8
    ranks; all-gathering K and V would replicate 191 MB onto every rank -- kept as ``gather_rows`` for head counts the
    rank count does not divide);
  * the Embedding Router's SpatialTemporalAttentionBlocks, SHARDED in two partitions (``RouterPartition``): frame-major
    (whole (id, frame) pairs per rank) for the spatial attention, location-major (a range of within-frame locations,
    all frames and ids) for the temporal / multi-ID attentions, the MLP and the head, one uneven all-to-all per change
    of partition; the 70 KB of sigmoid logits are all-gathered at the end;
  * the final unpatchify, which needs every token's 64 output channels (2 MB all-gather).

Weights are replicated (17 GB of 288 GB).  There is no reduce: no GEMM is K-split across ranks.

Transport (``BYA_SP_TRANSPORT``, default ``p2p`` on GPUs): every exchange below is ONE push kernel that stores straight
into the peers' receive buffers (``p2p.py`` / ``bya_p2p_push``, hipIpc-mapped, xGMI on a node) plus a one-wave wait kernel
on the receiver -- capturable in a hipGraph, ~5 us of launch instead of ~20 us per collective, and the packed q|k|v
projection travels as ONE exchange instead of three.  ``torch`` keeps the round-3 path: ``torch.distributed`` collectives
(RCCL on the ``nccl`` backend, host-staged on ``gloo``).

``FORCE_COLLECTIVES`` (a module attribute, set by tests/test_rccl_gpu.py): take the collective code path even with ONE rank -- a 1-rank
RCCL communicator on one GPU then executes every exchange (symbol resolution, dtypes, split lists, async handles on the
communicator's stream) although nothing moves between devices; used by ``tests/test_rccl_gpu.py``.
"""
import os
from dataclasses import dataclass

import torch
import torch.distributed as dist


FORCE_COLLECTIVES = False
COLLECTIVE_CALLS = {}          # kind -> number of device-side collectives issued (not host-staged ones); read by the tests


def _count(kind):
    COLLECTIVE_CALLS[kind] = COLLECTIVE_CALLS.get(kind, 0) + 1


@dataclass
class SeqShard:
    """Row range of this rank.  Global rows [r0, r1); text rows are global rows < Tt."""
    rank: int
    world: int
    S: int
    Tt: int
    group: object = None
    p2p: object = None               # p2p.P2PGroup: exchanges as push kernels over peer-mapped buffers (None: torch.distributed)

    def __post_init__(self):
        # contiguous row ranges whose sizes differ by at most one (17776 = 8 * 2222 divides exactly; the 720x1280
        # geometry of BASELINE config 3, S = 47026, does not)
        base, extra = divmod(self.S, self.world)
        self.sizes = [base + (1 if j < extra else 0) for j in range(self.world)]
        self.starts = [sum(self.sizes[:j]) for j in range(self.world)]
        self.even = extra == 0
        self.S_loc, self.S_max = self.sizes[self.rank], max(self.sizes)
        self.r0 = self.starts[self.rank]
        self.r1 = self.r0 + self.S_loc
        if self.world > 1 and self.Tt > self.sizes[0]:
            raise ValueError("text rows must fit inside the first rank's shard")
        self.Tt_loc = max(0, min(self.Tt, self.r1) - self.r0)          # local text rows (rank 0 only)
        self.v0 = max(self.r0, self.Tt) - self.Tt                      # first local video token (global index)
        self.v1 = self.r1 - self.Tt
        self.N_loc = self.v1 - self.v0
        self.N = self.S - self.Tt
        self._bufs = {}

    @property
    def active(self):
        """True when the step takes the sharded code path (more than one rank, or the single-rank test hook)."""
        return self.world > 1 or FORCE_COLLECTIVES

    # ---- exchange buffers: allocated once per (name, shape), reused by every layer of every step --------------------
    def buf(self, name, shape, like, zero=False):
        key = (name, tuple(shape), like.dtype, like.device)
        t = self._bufs.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(*shape, dtype=like.dtype, device=like.device)
            self._bufs[key] = t
        return t

    def recv_buf(self, name, shape, like):
        """A buffer an exchange RECEIVES into: with the P2P transport it must be a symmetric buffer the peers have mapped."""
        if self.p2p is not None:
            return self.p2p.symmetric(f"{name}:{tuple(shape)}", tuple(shape), like.dtype)
        return self.buf(name, shape, like)

    def _push(self, key, pieces, side=False):
        """One P2P exchange: pieces = [(contiguous local tensor, peer, receive-buffer name, element offset there)]."""
        ch = self.p2p.channel(("seq", key, self.S, self.Tt), pieces)
        _count("p2p_exchange")
        return ch.push(side=side)

    def _exchange(self, key, pieces):
        """push + wait as ONE launch (the exchanges nothing is overlapped with)."""
        ch = self.p2p.channel(("seq", key, self.S, self.Tt), pieces)
        _count("p2p_exchange")
        return ch.exchange()

    def staged(self, t):
        """gloo has no device collectives (single-GPU functional tests, CPU tests): stage through host memory."""
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    # ---- collectives ------------------------------------------------------------------------------------
    def _all_gather(self, out, local):
        """RCCL all-gather; with the gloo backend device tensors are staged through host memory."""
        if self.staged(local):
            host_out = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host_out, local.cpu(), group=self.group)
            out.copy_(host_out)
        else:
            _count("all_gather")
            dist.all_gather_into_tensor(out, local, group=self.group)

    def _gather_padded(self, local):
        """[C, S_loc, F] per rank (S_loc may differ by one) -> [C, S, F]: equal-size all-gather of S_max-row pads."""
        C, _, F = local.shape
        if self.even:
            pad = local.contiguous()
        else:
            pad = self.buf("gp_pad", (C, self.S_max, F), local, zero=True)
            pad[:, :self.S_loc] = local
        full = self.buf("gp_full", (self.world * C, self.S_max, F), local)
        self._all_gather(full, pad)                                       # rank-major concatenation along dim 0
        full = full.view(self.world, C, self.S_max, F)
        out = self.buf("gp_out", (C, self.S, F), local)
        if self.even:
            out.view(C, self.world, self.S_max, F).copy_(full.permute(1, 0, 2, 3))
        else:
            for j, n in enumerate(self.sizes):
                out[:, self.starts[j]:self.starts[j] + n] = full[j, :, :n]
        return out

    def gather_rows(self, local, out=None):
        """[S_loc, F] per rank -> [S, F], rank-major == global row order."""
        if not self.active:
            return local
        if self.p2p is not None:
            res = self.gather_rows_many([local], None if out is None else [out])[0]
            return res
        if self.even:
            if out is None:
                out = torch.empty(self.S, *local.shape[1:], dtype=local.dtype, device=local.device)
            self._all_gather(out, local.contiguous())
            return out
        res = self._gather_padded(local.reshape(1, self.S_loc, -1))[0].reshape(self.S, *local.shape[1:])
        if out is not None:
            out.copy_(res)
            return out
        return res.clone()                  # (res lives in a reused staging buffer)

    def begin_step(self):
        """Called by the engine at the top of a step: the parity of ``gather_rows_many``'s double buffer restarts, so that a
        replayed hipGraph (parities baked in at capture) and eager steps agree.  Safe across the step boundary: a step ends
        with the output gather, an exchange every rank joins after it has consumed its last gathered rows."""
        self._gr_parity = 0

    def gather_rows_many(self, locals_, outs=None):
        """P2P transport: several [S_loc, F] tensors of this rank (k and v of one layer) -> their [S, F] forms on every rank
        in ONE exchange.  The receive buffer alternates between two copies by call parity: a peer can only be one exchange
        ahead of this rank (it passed the wait of exchange n, for which this rank pushed AFTER copying exchange n - 1 out), so
        exchange n + 1 never lands in rows this rank is still copying out of (round 4 re-used one buffer for k, then v, then
        the next layer's k with nothing ordering the peer's push behind this rank's copy)."""
        C = len(locals_)
        F = locals_[0][0].numel()
        par = getattr(self, "_gr_parity", 0)
        self._gr_parity = par ^ 1
        name = f"gr{par}:{(C, self.S, F)}"
        full = self.p2p.symmetric(name, (C, self.S, F), locals_[0].dtype)
        srcs = [t.contiguous().view(self.S_loc, F) for t in locals_]
        self._exchange(("gr", C, F, par), [(srcs[c], j, name, (c * self.S + self.r0) * F) for j in range(self.world) for c in range(C)])
        res = []
        for c, t in enumerate(locals_):
            fc = full[c].view(self.S, *t.shape[1:])
            if outs is None:
                res.append(fc.clone())
            else:
                outs[c].copy_(fc)
                res.append(outs[c])
        return res

    def gather_video_rows(self, local_video, scratch=None, out=None, in_place=False):
        """[..., N_loc, F] per rank (rank 0 owns fewer video rows: its shard starts with the text rows)
        -> [..., N, F].  Implemented as a row all-gather of Tt_loc junk rows + the video rows.  ``out``: where the result
        goes (the engine passes a workspace tensor; without it a fresh tensor is returned); the staging buffers of the
        exchange itself are allocated once per shape and reused."""
        if not self.active:
            return local_video
        lead = local_video.shape[:-2]
        F = local_video.shape[-1]
        flat = local_video.reshape(-1, self.N_loc, F)
        C = flat.shape[0]
        if self.p2p is not None:
            # every rank stores its video rows of every leading index straight into place on every peer
            name = f"gv:{(C, self.N, F)}"
            full = self.p2p.symmetric(name, (C, self.N, F), flat.dtype)
            src = flat.contiguous()
            self._exchange("gv", [(src[c], j, name, (c * self.N + self.v0) * F) for j in range(self.world) for c in range(C)])
            if in_place:                      # the caller only READS the result before the next gather of this shape: no copy
                return full.view(*lead, self.N, F)
            if out is None:
                return full.view(*lead, self.N, F).clone()
            out.view(C, self.N, F).copy_(full)
            return out
        pad = self.buf("gv_pad", (C, self.S_loc, F), flat, zero=True) if scratch is None else scratch
        pad[:, self.Tt_loc:] = flat
        if self.Tt_loc and scratch is not None:
            pad[:, :self.Tt_loc] = 0
        full = self._gather_padded(pad)[:, self.Tt:]
        if out is None:
            out = torch.empty(*lead, self.N, F, dtype=flat.dtype, device=flat.device)
        out.view(C, self.N, F).copy_(full)
        return out

    def verify_gathered(self, full):
        """Canary for the P2P transport's weakest assumption: receive buffers are ordinary cached device
        memory a REMOTE GPU stores into, and a reader sees those stores only because the wait kernel's system-scope acquire made
        every XCD drop what it held of them.  A line that stays stale now and then is caught by no set-up test.  ``full`` is a
        tensor every rank holds in full after a gather (the engine passes the step's gathered prediction, a receive buffer that
        is re-used by every step): each rank sums its copy as 16-bit integers (exact, order-independent), the sums are traded in
        one 8-byte all-to-all, and a rank whose peers' sums differ from its own bumps word 38 of the group's first control block
        -- sticky: ``bya_p2p_poison`` turns the step's output into NaN, ``P2PGroup.check`` raises.  ~30 us per step;
        graph-capturable.  (The reductions are torch ops: diagnostics, not the step's arithmetic.)"""
        g = self.p2p
        if g is None or g.solo is not None or self.world == 1:
            return
        sums = g.symmetric("__verify__", (self.world,), torch.int64, zero=True)
        mine = self._bufs.get("verify_mine")         # (a fixed address: the exchange's copy table is built once, before any capture)
        if mine is None:
            mine = self._bufs["verify_mine"] = torch.zeros(1, dtype=torch.int64, device=full.device)
        mine.copy_(full.contiguous().view(torch.int16).sum(dtype=torch.int64))
        g.channel(("seq", "verify", self.S, self.Tt), [(mine, j, "__verify__", self.rank) for j in range(self.world)]).exchange()
        _count("p2p_exchange")
        g.ctrl[0, 38] += (sums != mine).any().to(torch.int32)

    # ---- head-parallel ("Ulysses") exchange for the joint self-attention ---------------------------------------
    # All-gathering K and V replicates 2*S*D elements onto every rank (191 MB received per rank and layer at 8 GPUs);
    # trading rows for heads moves every element of q, k, v (and of the output) exactly once: 48 MB per rank and layer.
    def _a2a(self, out, inp, out_splits=None, in_splits=None, async_op=False):
        """all_to_all_single.  ``async_op``: the exchange is enqueued on the process group's own RCCL stream (behind the
        work already on the compute stream) and a handle is returned; kernels launched afterwards on the compute stream
        overlap with it until ``handle.wait()`` makes the compute stream wait (no host synchronisation either way)."""
        if self.staged(inp):
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(host, inp.cpu(), out_splits, in_splits, group=self.group)
            out.copy_(host)
            return None
        _count("all_to_all_async" if async_op else "all_to_all")
        return dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group, async_op=async_op)

    def rows_to_heads(self, blocks, out=None, async_op=False):
        """blocks [world, S_loc, Dl]: this rank's rows, column block j = the heads owned by rank j
        -> [S, Dl]: ALL rows (global order) of this rank's heads.  Returns ``out`` (or the async handle)."""
        W, S_loc, Dl = blocks.shape
        if out is None:
            out = torch.empty(self.S, Dl, dtype=blocks.dtype, device=blocks.device)
        if self.even:
            h = self._a2a(out.view(-1), blocks.reshape(-1), async_op=async_op)
        else:
            h = self._a2a(out.view(-1), blocks.reshape(-1), [n * Dl for n in self.sizes], [S_loc * Dl] * W, async_op=async_op)
        return h if async_op else out

    def rows_to_heads_qkv(self, blocks, stats=None):
        """P2P transport: the packed projection's column blocks [3 * world, S_loc, Dl] (block t * world + j = tensor t of
        q | k | v, heads of rank j) -> ONE exchange (one launch) into the symmetric [3, S, Dl] buffer of every destination.
        ``stats`` (fp32 [slots, 2, heads], this rank's partial q/k squared-norm maxima): travels in the same exchange, into
        row block ``rank`` of every peer's [world * slots, 2, heads] table.
        Returns (q_heads, k_heads, v_heads, stats of all ranks or None), complete for every kernel enqueued behind the call."""
        W3, S_loc, Dl = blocks.shape
        W = self.world
        name = f"qkvh:{(3, self.S, Dl)}"
        full = self.p2p.symmetric(name, (3, self.S, Dl), blocks.dtype)
        pieces = [(blocks[t * W + j], j, name, (t * self.S + self.r0) * Dl) for j in range(W) for t in range(3)]
        st_all = None
        if stats is not None:
            slots, _, nh = stats.shape
            sname = f"qkstats:{(W * slots, 2, nh)}"
            st_all = self.p2p.symmetric(sname, (W * slots, 2, nh), stats.dtype)
            pieces += [(stats, j, sname, self.rank * stats.numel()) for j in range(W)]
        self._exchange(("qkvh", stats is not None), pieces)
        return full[0], full[1], full[2], st_all

    def push_v_heads(self, blocks_v):
        """(r6) Exchange A with v FIRST: the v projection's column blocks [world, S_loc, Dl] go to the v third of every
        destination's [3, S, Dl] buffer on the SIDE stream (``Channel.push(side=True)``), underneath the q | k projection the
        caller launches next; ``rows_to_heads_qk`` then exchanges q | k and waits for this push as well."""
        W, S_loc, Dl = blocks_v.shape
        name = f"qkvh:{(3, self.S, Dl)}"
        self.p2p.symmetric(name, (3, self.S, Dl), blocks_v.dtype)
        self._v_push = self._push(("vh",), [(blocks_v[j], j, name, (2 * self.S + self.r0) * Dl) for j in range(W)], side=True)

    def rows_to_heads_qk(self, blocks_qk):
        """The q | k column blocks [2 * world, S_loc, Dl] in one exchange, then the wait for the v push that has been running
        underneath the q | k projection.  Returns (q_heads, k_heads, v_heads) like ``rows_to_heads_qkv``."""
        W2, S_loc, Dl = blocks_qk.shape
        W = self.world
        name = f"qkvh:{(3, self.S, Dl)}"
        full = self.p2p.symmetric(name, (3, self.S, Dl), blocks_qk.dtype)
        self._exchange(("qkh",), [(blocks_qk[t * W + j], j, name, (t * self.S + self.r0) * Dl) for j in range(W) for t in range(2)])
        self._v_push.wait()
        self._v_push = None
        return full[0], full[1], full[2]

    def heads_to_rows(self, o_heads, out=None):
        """o_heads [S, Dl] (all rows, this rank's heads) -> [S_loc, world*Dl] (this rank's rows, all heads).  The
        exchange is enqueued on the communicator's own stream like every other one here (``async_op``) and the compute
        stream waits for it by event: nothing of the step is independent of the attention output, so there is nothing to
        put underneath it -- what the side stream buys here is that the exchange never queues behind unrelated compute."""
        S, Dl = o_heads.shape
        if self.p2p is not None:
            # (the name carries the geometry like every other symmetric buffer: one P2PGroup serves every resolution and frame
            # count the model is run at)
            # 2-D pieces: this rank's heads of destination j's rows go straight to column block `rank` of j's [S_loc_j, W * Dl]
            # buffer -- the layout the out-projection reads (round 4 received [W, S_loc, Dl] and permuted it with a copy kernel).
            # The RETURNED tensor is the symmetric buffer itself: the caller reads it in place (``out`` is not written).
            name = f"h2r:{(self.S, self.world, Dl)}"
            recv = self.p2p.symmetric(name, (self.S_loc, self.world * Dl), o_heads.dtype)     # (rank j: [sizes[j], W * Dl])
            pieces = [(o_heads[self.starts[j]:self.starts[j] + self.sizes[j]], j, name, self.rank * Dl, self.world * Dl)
                      for j in range(self.world)]
            self._exchange("h2r", pieces)
            return recv
        recv = self.buf("h2r_recv", (self.world, self.S_loc, Dl), o_heads)
        if self.even:
            h = self._a2a(recv.view(-1), o_heads.reshape(-1), async_op=True)
        else:
            h = self._a2a(recv.view(-1), o_heads.reshape(-1), [self.S_loc * Dl] * self.world, [n * Dl for n in self.sizes],
                          async_op=True)
        if h is not None:
            h.wait()
        if out is None:
            out = torch.empty(self.S_loc, self.world * Dl, dtype=o_heads.dtype, device=o_heads.device)
        out.view(self.S_loc, self.world, Dl).copy_(recv.permute(1, 0, 2))        # one strided copy, no temporaries
        return out

    def frame_segments(self, per_frame):
        """Local video rows split at frame boundaries: [(frame, local_start, length)]."""
        out, v = [], self.v0
        while v < self.v1:
            f = v // per_frame
            end = min(self.v1, (f + 1) * per_frame)
            out.append((f, v - self.v0, end - v))
            v = end
        return out


def _splits(n, parts):
    """Contiguous split of range(n) into ``parts`` pieces whose sizes differ by at most one: [(start, stop)]."""
    base, extra = divmod(n, parts)
    out, a = [], 0
    for i in range(parts):
        b = a + base + (1 if i < extra else 0)
        out.append((a, b))
        a = b
    return out


class RouterPartition:
    """Two partitions of the router feature tensor ``[n_id*T pairs, per_frame locations, F]`` and the all-to-all
    between them (exchange B of DESIGN.md section 5, sharded form).

    * partition A ("frame-major"): a rank owns whole (id, frame) pairs -> the spatial attention is local;
      local layout ``xa [nPA, per_frame, F]``.
    * partition B ("location-major"): a rank owns a range of within-frame locations for every pair -> the temporal
      and multi-ID attentions, every LayerNorm / GEMM / MLP and the sigmoid head are local; layout ``xb [pairs, nLB, F]``.

    An element moves once per exchange (all-to-all), 36 MB in total across the node instead of 36 MB *per rank* for an
    all-gather.  A -> B packs on the send side (W slice copies) and receives in place; B -> A sends in place and
    unpacks on the receive side."""

    def __init__(self, rank, world, pairs, per_frame, group=None, p2p=None):
        self.rank, self.world, self.pairs, self.per_frame, self.group = rank, world, pairs, per_frame, group
        self.p2p = p2p
        self.PA = _splits(pairs, world)
        self.LB = _splits(per_frame, world)
        self.pa0, self.pa1 = self.PA[rank]
        self.lb0, self.lb1 = self.LB[rank]
        self.nPA, self.nLB = self.pa1 - self.pa0, self.lb1 - self.lb0
        self._bufs = {}

    def buf(self, name, shape, like, zero=False):
        key = (name, tuple(shape), like.dtype, like.device)
        t = self._bufs.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(*shape, dtype=like.dtype, device=like.device)
            self._bufs[key] = t
        return t

    def recv_buf(self, name, shape, like):
        if self.p2p is not None:
            return self.p2p.symmetric(f"{name}:{(self.pairs, self.per_frame)}:{tuple(shape)}", tuple(shape), like.dtype)
        return self.buf(name, shape, like)

    def _push(self, key, pieces, side=False):
        ch = self.p2p.channel(("router", key, self.pairs, self.per_frame), pieces)
        _count("p2p_exchange")
        return ch.push(side=side)

    def _exchange(self, key, pieces):
        ch = self.p2p.channel(("router", key, self.pairs, self.per_frame), pieces)
        _count("p2p_exchange")
        return ch.exchange()

    def _name(self, base, shape):
        return f"{base}:{(self.pairs, self.per_frame)}:{tuple(shape)}"

    def _a2a(self, out, inp, out_splits, in_splits):
        """Uneven all-to-all on the communicator's own stream; returns the handle whose ``wait()`` makes the compute stream
        wait for it (None when the exchange was staged through the host, gloo)."""
        if inp.is_cuda and dist.get_backend(self.group) == "gloo":      # single-GPU functional test path
            host_out = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(host_out, inp.cpu(), out_splits, in_splits, group=self.group)
            out.copy_(host_out)
            return None
        _count("all_to_all_async")
        return dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group, async_op=True)

    def tokens_to_a(self, rs_loc, v0, T):
        """P2P transport: this rank's token rows of the router features, rs_loc [n_id, N_loc, F] (global video tokens
        v0 .. v0 + N_loc), go straight to the owners of their (id, frame) pairs: xa [nPA, per_frame, F] on every rank is filled
        by ONE exchange in which each element crosses one link once (the round-3 form all-gathered all 36 MB onto every
        rank and let it pick its pairs).  -> xa"""
        n_id, N_loc, F = rs_loc.shape
        pf = self.per_frame
        xa = self.recv_buf("rp_xa", (self.nPA, pf, F), rs_loc)
        name = self._name("rp_xa", (self.nPA, pf, F))
        src = rs_loc.contiguous()
        pieces = []
        for i in range(n_id):
            v = v0
            while v < v0 + N_loc:
                f = v // pf
                end = min(v0 + N_loc, (f + 1) * pf)
                pair = i * T + f
                owner = next(j for j, (a, b) in enumerate(self.PA) if a <= pair < b)
                pieces.append((src[i, v - v0:end - v0], owner, name, ((pair - self.PA[owner][0]) * pf + (v - f * pf)) * F))
                v = end
        self._exchange("t2a", pieces)
        return xa

    def a_to_b(self, xa, xb=None, overlap=None):
        """xa [nPA, per_frame, F] -> xb [pairs, nLB, F].  ``overlap``: a callable that enqueues independent work on the
        compute stream; it runs while the exchange is in flight on the communicator's stream."""
        F = xa.shape[-1]
        in_splits = [self.nPA * (b - a) * F for a, b in self.LB]
        if self.p2p is not None:
            # my pairs of destination j's locations go to rows [pa0, pa1) of j's xb [pairs, nLB_j, F]: one 2-D piece per
            # destination straight out of xa (rows = my pairs; a pair's locations [a, b) are dense) -- no pack copies
            xb = self.recv_buf("rp_xb", (self.pairs, self.nLB, F), xa)
            name = self._name("rp_xb", (self.pairs, self.nLB, F))
            pieces = [(xa[:, a:b], j, name, self.pa0 * (b - a) * F) for j, (a, b) in enumerate(self.LB)]     # (peer j's copy has ITS shape)
            if overlap is None:
                self._exchange("a2b", pieces)
                return xb
            h = self._push("a2b", pieces, side=True)
            overlap()
            h.wait()
            return xb
        send = self.buf("a2b_send", (sum(in_splits),), xa)
        off = 0
        for (a, b), n in zip(self.LB, in_splits):                  # pack per destination: W slice copies, no temporaries
            send[off:off + n].view(self.nPA, b - a, F).copy_(xa[:, a:b])
            off += n
        if xb is None:
            xb = torch.empty(self.pairs, self.nLB, F, dtype=xa.dtype, device=xa.device)
        out_splits = [(b - a) * self.nLB * F for a, b in self.PA]
        h = self._a2a(xb.view(-1), send, out_splits, in_splits)
        if overlap is not None:
            overlap()
        if h is not None:
            h.wait()
        return xb

    def b_to_a(self, xb, xa=None):
        """xb [pairs, nLB, F] -> xa [nPA, per_frame, F]."""
        F = xb.shape[-1]
        if self.p2p is not None:
            # destination j's pairs [a, b) of my locations go to columns [lb0, lb1) of j's xa [nPA_j, per_frame, F] -- which is
            # the symmetric "rp_xa" buffer (what ``tokens_to_a`` returned): one 2-D piece per destination, no unpack copies
            name = self._name("rp_xa", (self.nPA, self.per_frame, F))
            mine = self.recv_buf("rp_xa", (self.nPA, self.per_frame, F), xb)
            if xa is not None and xa.data_ptr() != mine.data_ptr():
                raise ValueError("b_to_a on the P2P transport writes into the symmetric rp_xa buffer: pass the tensor tokens_to_a returned")
            pieces = [(xb[a:b], j, name, self.lb0 * F, self.per_frame * F) for j, (a, b) in enumerate(self.PA)]
            self._exchange("b2a", pieces)
            return mine
        if xa is None:
            xa = torch.empty(self.nPA, self.per_frame, F, dtype=xb.dtype, device=xb.device)
        in_splits = [(b - a) * self.nLB * F for a, b in self.PA]
        out_splits = [self.nPA * (b - a) * F for a, b in self.LB]
        recv = self.buf("b2a_recv", (sum(out_splits),), xb)
        h = self._a2a(recv, xb.reshape(-1), out_splits, in_splits)
        if h is not None:
            h.wait()
        off = 0
        for a, b in self.LB:
            n = self.nPA * (b - a) * F
            xa[:, a:b] = recv[off:off + n].view(self.nPA, b - a, F)
            off += n
        return xa

    def gather_b_rows(self, yb, out=None):
        """yb [T_or_pairs, nLB, C] per rank (location-major) -> [T_or_pairs, per_frame, C] on every rank."""
        lead, C = yb.shape[0], yb.shape[-1]
        if self.p2p is not None:
            # every rank's locations straight into place on every peer: one 2-D piece per destination (rows = lead).  The
            # result is the symmetric buffer itself (read it before the next gather of this shape; ``out`` is not written).
            full = self.recv_buf("gb_out", (lead, self.per_frame, C), yb)
            name = self._name("gb_out", (lead, self.per_frame, C))
            src = yb.contiguous()
            self._exchange("gb", [(src, j, name, self.lb0 * C, self.per_frame * C) for j in range(self.world)])
            return full
        nmax = max(b - a for a, b in self.LB)
        pad = self.buf("gb_pad", (lead, nmax, C), yb, zero=True)
        pad[:, :self.nLB] = yb
        full = self.buf("gb_full", (self.world * lead, nmax, C), yb)
        if pad.is_cuda and dist.get_backend(self.group) == "gloo":
            host = torch.empty(full.shape, dtype=full.dtype)
            dist.all_gather_into_tensor(host, pad.cpu(), group=self.group)
            full.copy_(host)
        else:
            _count("all_gather")
            dist.all_gather_into_tensor(full, pad, group=self.group)
        full = full.view(self.world, lead, nmax, C)
        if out is None:
            out = torch.empty(lead, self.per_frame, C, dtype=yb.dtype, device=yb.device)
        for j, (a, b) in enumerate(self.LB):
            out[:, a:b] = full[j, :, :b - a]
        return out


TRANSPORTS = ("p2p", "p2p-fine", "torch")


def shard_sequence(model, group=None, transport=None):
    """Switch ``model`` (BindyouravatarTransformer3DModel) to sequence-parallel execution over ``group``.
    ``transport`` (BYA_SP_TRANSPORT), a ladder the ranks walk TOGETHER from the rung asked for (default: the first on GPUs,
    the last on CPUs) until one passes its set-up test on every rank:
      "p2p"       push kernels into the peers' ordinary (coarse-grained, L2-cached) receive buffers -- the fast form;
      "p2p-fine"  the same into fine-grained receive buffers (BYA_P2P_MEM=fine: coherent by construction, slower to read);
      "torch"     torch.distributed collectives (RCCL on the nccl backend).
    ``model._seq_transport`` names the rung that runs.  The set-up test is ``P2PGroup.self_test`` (repeated exchanges into a
    re-used buffer with consumer reads in between, every word checked); ``bench.py --gpus N`` additionally compares the sharded
    step with the unsharded one and moves down the ladder on a difference.  COLLECTIVE: every rank of the group calls it at the
    same point."""
    model._seq_group = group if group is not None else dist.group.WORLD
    model._seq_world = dist.get_world_size(model._seq_group)
    model._seq_rank = dist.get_rank(model._seq_group)
    transport = transport or os.environ.get("BYA_SP_TRANSPORT") or ("p2p" if torch.cuda.is_available() else "torch")
    if transport == "p2p" and os.environ.get("BYA_P2P_MEM") == "fine":
        transport = "p2p-fine"
    if transport not in TRANSPORTS:
        raise ValueError(f"unknown transport {transport!r}: expected one of {TRANSPORTS}")
    old = getattr(model, "_seq_p2p", None)
    if old is not None:
        old.close()             # (its time-out counters are sticky: a rung that was left must not fail the run at its end)
    model._seq_p2p, model._seq_transport, notes = None, "torch", []
    for rung in TRANSPORTS[TRANSPORTS.index(transport):]:
        if rung == "torch":
            break
        from .p2p import P2PGroup
        dev = next(model.parameters()).device
        p2p, err = None, None
        try:
            p2p = P2PGroup(model._seq_group, dev, mem="fine" if rung == "p2p-fine" else "coarse")
            p2p.self_test()
        except Exception as e:              # no hipIpc on this platform, peer access refused, stale words, ...
            err = repr(e)                   # (the text only: the traceback would keep the failed group alive)
        # every rank must agree: one rank on collectives and another on push kernels would deadlock.  A rank that raised
        # LOCALLY inside the set-up arrives here one object collective early: the tag shows it to everybody, the peers raise
        # out of their set-up (PhaseMismatch, caught above) and come here too, this rank asks again (p2p.gather_tagged)
        from .p2p import PhaseMismatch, gather_tagged
        for _ in range(3):
            try:
                ok = gather_tagged(model._seq_group, model._seq_world, ("ladder", rung), err is None)
                break
            except PhaseMismatch as e:
                err = err or repr(e)
        else:
            raise RuntimeError(f"sequence-parallel set-up: the ranks did not get back in step on rung {rung!r}")
        if all(ok):
            model._seq_p2p, model._seq_transport = p2p, rung
            break
        if p2p is not None:
            p2p.close()
        notes.append(f"{rung}: {err} on this rank; ranks ok: {ok}")
    if notes:
        import warnings
        warnings.warn("P2P exchange engine: " + "; ".join(notes) + f" -> running on {model._seq_transport!r}")
    model._seq_transport_notes = notes
    model.invalidate_engine()
    return model


# ---- classifier-free-guidance batch split (first row) ---------------------------------------
class CfgSplit:
    """The two halves of a CFG batch ([uncond, cond], reference models/pipeline_bindyouravatar.py:897-923) never interact
    inside ``forward``, so ranks [0, W/2) run sample 0 and ranks [W/2, W) run sample 1, each half optionally
    sequence-sharded over its own W/2 ranks.  One exchange per step: rank r and rank r + W/2 trade their
    ``[1,13,16,60,90]`` predictions (2 MB) so every rank returns the full ``[2, ...]`` batch."""

    def __init__(self, group=None):
        group = group if group is not None else dist.group.WORLD
        self.parent = group
        ranks = dist.get_process_group_ranks(group)
        W = len(ranks)
        if W % 2:
            raise ValueError("the CFG split needs an even number of ranks")
        me = dist.get_rank(group)
        self.world, self.half_size, self.half = W, W // 2, me // (W // 2)
        # every rank creates every subgroup, in the same order (torch.distributed requirement)
        halves = [dist.new_group(ranks[h * self.half_size:(h + 1) * self.half_size]) for h in range(2)]
        pairs = [dist.new_group([ranks[r], ranks[r + self.half_size]]) for r in range(self.half_size)]
        self.seq_group = halves[self.half]
        self.pair_group = pairs[me % self.half_size]

    # positions in the engine's argument tuple (hidden_states, encoder_hidden_states, timestep, image_rotary_emb,
    # id_cond, id_vit_hidden, audio_embeds, af_matrix, routing_logits_forcing) and the rank each batched tensor has
    # WITH its batch axis.  Everything else (RoPE tables, forcing logits) is shared by both samples and never sliced.
    BATCHED = {0: 5, 1: 3, 2: 1, 4: 2, 5: 3, 6: None, 7: 3}

    def _slice(self, obj, ndim, what):
        if obj is None:
            return None
        if isinstance(obj, (list, tuple)):
            return type(obj)(self._slice(o, ndim, what) for o in obj)
        if not torch.is_tensor(obj):
            return obj
        if ndim is not None and obj.dim() != ndim:
            if obj.dim() == ndim - 1 or obj.dim() == 0:
                return obj                                   # unbatched form (e.g. a scalar timestep): shared
            raise ValueError(f"CFG split: {what} has {obj.dim()} dims, expected {ndim} (batch first)")
        if obj.shape[0] == 1:
            return obj                                       # broadcast over the batch: shared
        if obj.shape[0] != 2:
            raise ValueError(f"CFG split: {what} has batch {obj.shape[0]}, expected 2 ([uncond, cond])")
        return obj[self.half:self.half + 1]

    def take(self, args):
        """This rank's sample of the engine's argument tuple: arguments are sliced BY POSITION (only the ones that
        carry a batch axis), never by a shape heuristic."""
        names = ("hidden_states", "encoder_hidden_states", "timestep", "image_rotary_emb", "id_cond",
                 "id_vit_hidden", "audio_embeds", "af_matrix", "routing_logits_forcing")
        return tuple(self._slice(a, self.BATCHED[i], names[i]) if i in self.BATCHED else a
                     for i, a in enumerate(args))

    def join(self, local):
        """[1, ...] per rank -> [2, ...] on every rank (sample order = half order)."""
        out = torch.empty(2, *local.shape[1:], dtype=local.dtype, device=local.device)
        if local.is_cuda and dist.get_backend(self.pair_group) == "gloo":
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host, local.contiguous().cpu(), group=self.pair_group)
            out.copy_(host)
        else:
            _count("all_gather")
            dist.all_gather_into_tensor(out, local.contiguous(), group=self.pair_group)
        return out


def shard_cfg(model, group=None, transport=None):
    """Run the two samples of a CFG batch on two halves of ``group``; inside a half the step is sequence-parallel
    when the half has more than one rank (2 x 2, 2 x 4; ``transport`` as for ``shard_sequence``).  Batch-1 calls fall
    through to plain execution on the half.  Calling it again (another transport) keeps the process groups."""
    if getattr(model, "_cfg", None) is None or model._cfg.parent is not (group if group is not None else dist.group.WORLD):
        model._cfg = CfgSplit(group)
    if model._cfg.half_size > 1:
        shard_sequence(model, model._cfg.seq_group, transport=transport)
    else:
        model.invalidate_engine()
    return model
