"""Torch-tensor front end of the C-ABI kernels (``include/bya.h``).

Every function enqueues hand-written HIP kernels on torch's current stream and returns ``out``.
Tensors are bf16 device tensors unless stated; 2-D ``[rows, cols]`` or 3-D ``[batch, rows, cols]`` views
with unit inner stride are accepted (no copies are made here).
"""
import ctypes
import os

import torch

from . import _hip
from ._hip import AttnDesc, AttnMixDesc, GemmDesc, check

ACT = {None: 0, "none": 0, "gelu_tanh": 1, "gelu_erf": 2, "relu": 3, "silu": 4, "leaky_relu": 5,
       "gelu_tanh_ieee": 6}        # 6: round-1 GELU form (expf + IEEE division), GEMM epilogue only, kept for A/B


_PINNED_STREAM = None


def _stream():
    """HIP stream every launch goes to: torch's current stream (looked up per call, ~8 us) unless a step pinned it."""
    if _PINNED_STREAM is not None:
        return _PINNED_STREAM
    return torch.cuda.current_stream().cuda_stream


class pinned_stream:
    """Context manager used by the engine around one step: resolve torch's current stream ONCE for the ~2000 launches
    of the step (the lookup was a third of the host-side enqueue time).  Nesting keeps the outer pin."""

    def __enter__(self):
        global _PINNED_STREAM
        self.prev = _PINNED_STREAM
        if _PINNED_STREAM is None:
            _PINNED_STREAM = torch.cuda.current_stream().cuda_stream
        return self

    def __exit__(self, *exc):
        global _PINNED_STREAM
        _PINNED_STREAM = self.prev
        return False


class on_stream:
    """Enqueue the launches inside the block on ``stream`` (a torch.cuda.Stream) instead of the step's pinned stream, behind
    everything already enqueued on the pinned stream; ``join()`` afterwards makes the pinned stream wait for them.  Used for
    work that nothing of the current stream's near future depends on (the step-invariant conditioning, which the first
    routing layer needs ~8 ms later): its small launches fill the CUs the big kernels leave idle."""

    def __init__(self, stream):
        self.stream, self.done = stream, None

    def __enter__(self):
        global _PINNED_STREAM
        self.main = torch.cuda.current_stream()
        self.prev_pin = _PINNED_STREAM
        fork = torch.cuda.Event()
        fork.record(self.main)
        self.stream.wait_event(fork)
        self.ctx = torch.cuda.stream(self.stream)
        self.ctx.__enter__()
        _PINNED_STREAM = self.stream.cuda_stream
        return self

    def __exit__(self, *exc):
        global _PINNED_STREAM
        self.done = torch.cuda.Event()
        self.done.record(self.stream)
        _PINNED_STREAM = self.prev_pin
        self.ctx.__exit__(*exc)
        return False

    def mark(self):
        """Inside the block: an event at this point of the side stream (``need(event)`` later makes the main stream wait for
        everything enqueued up to here, and no more)."""
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return ev

    @staticmethod
    def need(ev):
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def join(self):
        if self.done is not None:
            torch.cuda.current_stream().wait_event(self.done)
            self.done = None


# ---- which kernel serves a skinny Linear is the caller's decision (include/bya.h: bya_gemm_skinny_bf16) ------------
_WEIGHT_STREAMING = False


class weight_streaming:
    """``with ops.weight_streaming():`` -- Linears of at most 64 rows inside the block go to the weight-streaming kernel
    (``bya_gemm_skinny_bf16``) when they qualify.  For code whose row counts are properties of the MODEL (the step-invariant
    conditioning), never for the token stream: a shard's rows must round like the whole's.  BYA_GEMM_SKINNY=0 switches it off."""

    def __enter__(self):
        global _WEIGHT_STREAMING
        self._old, _WEIGHT_STREAMING = _WEIGHT_STREAMING, os.environ.get("BYA_GEMM_SKINNY") != "0"      # (read per block, not per GEMM)
        return self

    def __exit__(self, *exc):
        global _WEIGHT_STREAMING
        _WEIGHT_STREAMING = self._old
        return False


# ---- library options (include/bya.h bya_set_option): the C entry points read no environment ---------------------------
set_option, get_option = _hip.set_option, _hip.get_option


class options:
    """``with ops.options(gemm_splitk=0, attn_streamk=0): ...`` -- set library options (names: ``_hip.OPTIONS``;
    ``reference_forms`` takes a mask or an iterable of ``_hip.REFERENCE_FORMS`` names) for the launches ENQUEUED inside the
    block, restore the previous values on exit."""

    def __init__(self, **kw):
        forms = kw.get("reference_forms")
        if forms is not None and not isinstance(forms, int):
            kw["reference_forms"] = sum(_hip.REFERENCE_FORMS[f] for f in ([forms] if isinstance(forms, str) else forms))
        self.kw = kw

    def __enter__(self):
        self.old = {k: get_option(k) for k in self.kw}
        for k, v in self.kw.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def strict_summation():
    """No split-K tails, no stream-K attention: every output element is summed in ONE fixed order whatever the launch's
    row count -- the mode in which a rank's shard reproduces the unsharded step bit for bit."""
    return options(gemm_splitk=0, attn_streamk=0)


def board_calibration(device=None, seconds=0.3, zeros=False):
    """TFLOP/s THIS board sustains on a loop of nothing but the GEMM's MFMA (v_mfma_f32_16x16x32_bf16) with gaussian bf16
    operands in registers (include/bya.h bya_mfma_calibration): the ceiling a bench line can be read against on a pool whose
    boxes differ by a few per cent in clock under load.  Two launches of ~``seconds`` each, the second one timed (the first
    brings the board to its steady clock).  Synchronises."""
    lib = _hip.load()
    device = torch.device(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    n = (256 * 256 * 16 * 16) // 2
    g = torch.Generator(device="cpu").manual_seed(1234)
    src = torch.zeros(n, dtype=torch.bfloat16, device=device) if zeros else torch.randn(n, generator=g).to(torch.bfloat16).to(device)
    sink = torch.zeros(1, dtype=torch.float32, device=device)
    flop_per_iter = 256.0 * 4 * 2.0 * 128 * 128 * 32
    iters = max(1000, int(seconds * 2.0e15 / flop_per_iter))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.device(device):
        for timed in (False, True):
            if timed:
                s.record()
            check(lib.bya_mfma_calibration(src.data_ptr(), src.numel() * 2, sink.data_ptr(), iters, _stream()), "bya_mfma_calibration")
        e.record()
        torch.cuda.synchronize(device)
    return iters * flop_per_iter / (s.elapsed_time(e) * 1e-3) / 1e12


# ---- optional per-entry-point timers (HIP events recorded on the launch stream; used by bench.py) -------------
_TIMERS = None
_FLOPS = {}


_SHAPE_LABELS = False


def enable_kernel_timers(by_shape=False):
    """``by_shape``: GEMM launches are keyed ``bya_gemm_bf16:BxMxNxK:epilogue`` instead of by entry point alone
    (tools/gemm_breakdown.py)."""
    global _TIMERS, _FLOPS, _SHAPE_LABELS
    _TIMERS, _FLOPS, _SHAPE_LABELS = {}, {}, bool(by_shape)
    return _TIMERS


def _begin(name, flops=0.0):
    if _TIMERS is None:
        return None
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()           # torch's CURRENT stream == the stream every kernel here is launched on
    _FLOPS[name] = _FLOPS.get(name, 0.0) + flops
    return (name, s, e)


def _end(tok):
    if tok is not None:
        tok[2].record()
        _TIMERS.setdefault(tok[0], []).append((tok[1], tok[2]))


def collect_kernel_timers():
    """-> {name: [seconds per launch]}; synchronises, then disables the timers."""
    global _TIMERS
    torch.cuda.synchronize()
    out = {k: [s.elapsed_time(e) * 1e-3 for s, e in v] for k, v in (_TIMERS or {}).items()}
    _TIMERS = None
    return out


def kernel_timer_flops():
    return dict(_FLOPS)


def _p(t):
    return None if t is None else t.data_ptr()


def _mat(t, name):
    """-> (batch, rows, cols, batch_stride, row_stride) of a 2-D/3-D view with unit inner stride."""
    if t.dtype != torch.bfloat16:
        raise TypeError(f"{name}: expected bf16, got {t.dtype}")
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a device tensor (the engine has no CPU path)")
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: inner stride must be 1")
    if t.dim() == 2:
        return 1, t.shape[0], t.shape[1], 0, t.stride(0)
    if t.dim() == 3:
        return t.shape[0], t.shape[1], t.shape[2], t.stride(0), t.stride(1)
    raise ValueError(f"{name}: expected 2-D or 3-D, got {t.dim()}-D")


_GEMM_WS = {}          # device index -> workspace tensor (lives as long as the process)


def ensure_gemm_workspace(device):
    """Register the split-K workspace of the persistent GEMM kernel (bya_set_gemm_workspace) once per DEVICE: 64 MiB of
    zero-filled device memory that lives as long as the process.  Without it the kernels still run (no K split).  The
    library picks the workspace of the device that is current when a GEMM is enqueued."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ws = _GEMM_WS.get(idx)
    if ws is not None:
        return ws
    lib = _hip.load()
    n = ctypes.c_int64(0)
    check(lib.bya_gemm_workspace_bytes(ctypes.byref(n)), "bya_gemm_workspace_bytes")
    with torch.cuda.device(idx):
        ws = torch.zeros(n.value, dtype=torch.uint8, device=torch.device("cuda", idx))
        torch.cuda.synchronize(idx)
        check(lib.bya_set_gemm_workspace(ws.data_ptr(), n.value), "bya_set_gemm_workspace")
    _GEMM_WS[idx] = ws
    return ws


def gemm_workspace_status(device=None):
    """Number of split-K tiles of this device whose finisher timed out waiting for a partial sum (bya_gemm_workspace_status):
    0 on a healthy run.  Synchronises the current stream -- call it at step / run end.  ``check_gemm_workspace`` raises."""
    lib = _hip.load()
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    n = ctypes.c_int32(0)
    with torch.cuda.device(idx):
        check(lib.bya_gemm_workspace_status(ctypes.byref(n), _stream()), "bya_gemm_workspace_status")
    return n.value


_HEALED = {}           # device index -> [split-K time-outs, stream-K time-outs] already absorbed by heal_handoffs
HANDOFF_MODE = {}      # device index -> why the split forms are off on that device (absent: they are on)


def _handoff_counts(device=None):
    idx = None if device is None else torch.device(device).index
    if idx is None:                      # (None, "cuda": the current device)
        idx = torch.cuda.current_device()
    g = gemm_workspace_status(idx) if idx in _GEMM_WS else 0
    a = attn_workspace_status(idx) if idx in _ATTN_WS else 0
    seen = _HEALED.get(idx, [0, 0])
    return idx, g - seen[0], a - seen[1], g, a


def heal_handoffs(device=None):
    """Self-healing for the two kernels that hand partial sums between workgroups of ONE launch (split-K tails of
    bya_gemm_bf16, stream-K items of the joint attention).  Both count on the whole grid being resident, which holds on a
    GPU the process owns and not when something else keeps CUs busy (a second job, a profiler's helper process): the
    waiting side then gives up after a bounded spin, finishes without the missing sums and COUNTS the event.  This call
    (it synchronises: once per step) looks at the counters; if any hand-off timed out since the last call it switches both
    split forms off for the rest of the process (library options gemm_splitk = 0, attn_streamk = 0 -- one workgroup per
    tile / item, nothing to wait for), warns, records the mode in ``HANDOFF_MODE`` and returns True: THE CALLER RE-RUNS
    THE STEP, whose result is not to be trusted.  -> False on a healthy step."""
    idx, g_new, a_new, g, a = _handoff_counts(device)
    if not (g_new or a_new):
        return False
    _HEALED[idx] = [g, a]
    set_option("gemm_splitk", 0)
    set_option("attn_streamk", 0)
    HANDOFF_MODE[idx] = (f"unsplit (self-healed: {g_new} split-K and {a_new} stream-K hand-off(s) timed out -- the launch's grid was "
                         f"not co-resident, the GPU is shared; gemm_splitk = attn_streamk = 0 from here on)")
    import warnings
    warnings.warn("bind_your_avatar_implementation_amd: " + HANDOFF_MODE[idx] + "; the step is re-run")
    return True


def check_gemm_workspace(device=None):
    """Raise if a hand-off timed out that ``heal_handoffs`` has not absorbed (callers that do not re-run steps: the end of a
    clip, the end of a bench run), or if a P2P wait gave up."""
    idx, g_new, a_new, _, _ = _handoff_counts(device)
    if g_new:
        raise _hip.ByaError(f"{g_new} split-K tile(s) of bya_gemm_bf16 were finished without all their partial sums (a hand-off "
                            f"between workgroups timed out): results of this run are not to be trusted")
    if a_new:
        raise _hip.ByaError(f"{a_new} stream-K hand-off(s) of the joint attention timed out: results of this run are not to be trusted")
    # ... and no wait of a P2P exchange gave up (sharded runs; the step's output is NaN-poisoned as well, parallel / p2p.py)
    import sys
    p2p = sys.modules.get(__package__ + ".p2p")
    if p2p is not None:
        for g in list(p2p.LIVE_GROUPS):
            g.check()


_ATTN_WS = {}          # device index -> stream-K exchange workspace of the joint-attention kernel


def ensure_attn_workspace(device):
    """Register the stream-K workspace of the joint-attention kernel (bya_set_attn_workspace) once per DEVICE: 69 MB of
    zero-filled device memory that lives as long as the process.  With it, a launch whose (head, q-tile) items do not fill
    whole rounds of 256 CUs runs as 256 persistent workgroups (whole rounds, then the leftover items cut at one key tile
    between "mains" and "helpers"); without it, one workgroup per item (the last round partly idle).  Same results up to
    fp32 summation order at the cut items."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ws = _ATTN_WS.get(idx)
    if ws is not None:
        return ws
    lib = _hip.load()
    n = ctypes.c_int64(0)
    check(lib.bya_attn_workspace_bytes(ctypes.byref(n)), "bya_attn_workspace_bytes")
    with torch.cuda.device(idx):
        ws = torch.zeros(n.value, dtype=torch.uint8, device=torch.device("cuda", idx))
        torch.cuda.synchronize(idx)
        check(lib.bya_set_attn_workspace(ws.data_ptr(), n.value), "bya_set_attn_workspace")
    _ATTN_WS[idx] = ws
    return ws


def attn_workspace_status(device=None):
    """Number of stream-K hand-offs of the joint attention that timed out on this device (0 on a healthy run); synchronises."""
    lib = _hip.load()
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    n = ctypes.c_int32(0)
    with torch.cuda.device(idx):
        check(lib.bya_attn_workspace_status(ctypes.byref(n), _stream()), "bya_attn_workspace_status")
    return n.value


def gemm(a, w, out, bias=None, res=None, gate0=None, gate1=None, gate_split=0, gate_batch_stride=0, act=None,
         split=None, bias_rowscale=None, alpha=1.0):
    """out = res + gate * act(a @ w.T + bias).  a: [(B,) M, K], w: [N, K], out/res: [(B,) M, N].

    ``split=(n_split, stride)``: ``out`` is the FIRST of N/n_split equally shaped tensors ``stride`` elements apart;
    column n of the product lands in tensor n // n_split (packed q|k|v projection -> three buffers, one launch)."""
    lib = _hip.load()
    ab, M, K, a_bs, lda = _mat(a, "a")
    ob, Mo, N, c_bs, ldc = _mat(out, "out")
    if a.device.index not in _GEMM_WS:
        ensure_gemm_workspace(a.device)
    if split is not None:
        N = w.shape[0]
    if w.dim() != 2 or w.shape[1] != K or w.shape[0] != N or w.stride(1) != 1 or w.dtype != torch.bfloat16:
        raise ValueError(f"w: expected bf16 [{N}, {K}], got {tuple(w.shape)} {w.dtype}")
    if (ab, M) != (ob, Mo):
        raise ValueError("a/out row mismatch")
    d = GemmDesc()
    d.M, d.N, d.K, d.batch = M, N, K, ab
    d.lda, d.ldw, d.ldc = lda, w.stride(0), ldc
    d.a_batch_stride, d.c_batch_stride = a_bs, c_bs
    d.ldres, d.res_batch_stride = 0, 0
    if res is not None:
        rb, Mr, Nr, r_bs, ldres = _mat(res, "res")
        if (Mr, Nr) != (M, N) or rb not in (1, ab):
            raise ValueError("res shape mismatch")
        d.ldres, d.res_batch_stride = ldres, (r_bs if rb == ab else 0)
    d.gate_batch_stride, d.gate_split, d.act = gate_batch_stride, gate_split, ACT[act]
    d.n_split, d.c_split_stride = (0, 0) if split is None else split
    d.bias_rowscale, d.alpha = _p(bias_rowscale), float(alpha)
    if bias_rowscale is not None:
        assert bias_rowscale.dtype == torch.float32 and bias_rowscale.is_contiguous() and bias_rowscale.numel() == ab * M
    # per-kernel timers: Linears over fewer than 1024 rows (the step-invariant conditioning: 32 face tokens, 52 audio windows,
    # 577 ViT tokens against 2048..49152-wide weights) stream their WEIGHTS and are bound by HBM, not by the matrix cores --
    # they get their own bucket so that the MFMA roofline of bench.py is taken over the launches it applies to
    name = "bya_gemm_bf16" if M >= 1024 else "bya_gemm_bf16_small_m"
    if _SHAPE_LABELS:
        name += f":{ab}x{M}x{N}x{K}:{act or 'none'}{'+gate' if gate0 is not None else ''}{'+res' if res is not None else ''}"
    tok = _begin(name, 2.0 * ab * M * N * K)
    if (_WEIGHT_STREAMING and M <= 64 and N <= 8192 and N % 16 == 0 and K % 32 == 0 and K >= 256 and gate0 is None
            and bias_rowscale is None and split is None and a_bs % 8 == 0):
        check(lib.bya_gemm_skinny_bf16(_p(a), _p(w), _p(bias), _p(out), _p(res), ctypes.byref(d), _stream()),
              "bya_gemm_skinny_bf16")
    else:
        check(lib.bya_gemm_bf16(_p(a), _p(w), _p(bias), _p(out), _p(res), _p(gate0), _p(gate1), ctypes.byref(d),
                                _stream()), "bya_gemm_bf16")
    _end(tok)
    return out


def gemm_qkv_norm_rope(a, w, out, bias, split, qw, qb, kw, kb, cos, sin, text_rows, eps=1e-6, k_scale=1.0, tensors=3):
    """The packed q|k|v projection with the q/k LayerNorm(64) + RoPE in its epilogue (bya_gemm_qkv_norm_rope): equals
    ``gemm(..., split=split)`` followed by ``qknorm_rope`` bit for bit, in one launch.  Returns False (nothing launched) when
    the library does not take the shape -- the caller then issues the two launches."""
    lib = _hip.load()
    ab, M, K, a_bs, lda = _mat(a, "a")
    ob, Mo, _, c_bs, ldc = _mat(out, "out")
    N = w.shape[0]
    if w.dim() != 2 or w.shape[1] != K or w.stride(1) != 1 or w.dtype != torch.bfloat16 or tensors not in (2, 3) or N % tensors \
            or (ab, M) != (ob, Mo):
        raise ValueError("gemm_qkv_norm_rope: a [(B,) M, K], w [3 * width, K] (or [2 * width, K]: q | k alone, tensors=2), "
                         "out = the first of the split outputs")
    d = GemmDesc()
    d.M, d.N, d.K, d.batch = M, N, K, ab
    d.lda, d.ldw, d.ldc = lda, w.stride(0), ldc
    d.a_batch_stride, d.c_batch_stride = a_bs, c_bs
    d.n_split, d.c_split_stride = split
    d.alpha = 1.0
    n = _hip.QkNormDesc()
    n.qw, n.qb, n.kw, n.kb, n.cos, n.sin = _p(qw), _p(qb), _p(kw), _p(kb), _p(cos), _p(sin)
    n.text_rows, n.width, n.eps, n.k_scale = int(text_rows), N // tensors, float(eps), float(k_scale)
    if cos is not None:
        assert cos.dtype == torch.float32 and sin.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous()
        assert cos.shape == (M - text_rows, 64)
    tok = _begin("bya_gemm_bf16" if M >= 1024 else "bya_gemm_bf16_small_m")
    rc = lib.bya_gemm_qkv_norm_rope(_p(a), _p(w), _p(bias), _p(out), ctypes.byref(d), ctypes.byref(n), _stream())
    if rc == -4:                       # BYA_ERR_UNSUPPORTED: not this kernel's shape -- nothing was launched, nothing is counted
        return False                   # (the caller's plain GEMM counts the FLOPs; round 5 counted them here as well)
    check(rc, "bya_gemm_qkv_norm_rope")
    if tok is not None:
        _FLOPS[tok[0]] = _FLOPS.get(tok[0], 0.0) + 2.0 * ab * M * N * K
    _end(tok)
    return True


def quantize_rows_fp8(x, q=None, scale=None):
    """Per-row symmetric e4m3 quantisation of a bf16 matrix [(B,) M, K] -> (uint8 [.., M, K], fp32 scale [.., M])."""
    lib = _hip.load()
    b, M, K, bs, ldx = _mat(x, "x")
    if b > 1 and bs != M * ldx:
        raise ValueError("quantize_rows_fp8: batch entries must be evenly stacked rows")
    if q is None:
        q = torch.empty(*x.shape, dtype=torch.uint8, device=x.device)
    if scale is None:
        scale = torch.empty(*x.shape[:-1], dtype=torch.float32, device=x.device)
    assert q.dtype == torch.uint8 and q.is_contiguous() and scale.dtype == torch.float32 and scale.is_contiguous()
    assert q.numel() == b * M * K and scale.numel() == b * M
    tok = _begin("bya_quantize_rows_fp8")
    check(lib.bya_quantize_rows_fp8(_p(x), _p(q), _p(scale), b * M, K, ldx, K, _stream()), "bya_quantize_rows_fp8")
    _end(tok)
    return q, scale


def gemm_fp8(a8, a_scale, w8, w_scale, out, bias=None, res=None, gate0=None, gate1=None, gate_split=0,
             gate_batch_stride=0, act=None, split=None, alpha=1.0):
    """out = res + gate * act(a_scale * w_scale * (a8 @ w8.T) + bias) with e4m3 operands (``quantize_rows_fp8``)."""
    lib = _hip.load()
    if a8.dim() == 2:
        ab, (M, K) = 1, a8.shape
    else:
        ab, M, K = a8.shape
    ob, Mo, N, c_bs, ldc = _mat(out, "out")
    if split is not None:
        N = w8.shape[0]
    assert a8.dtype == torch.uint8 and w8.dtype == torch.uint8 and a8.is_contiguous() and w8.is_contiguous()
    assert w8.shape == (N, K) and (ab, M) == (ob, Mo)
    assert a_scale.dtype == torch.float32 and a_scale.numel() == ab * M and a_scale.is_contiguous()
    assert w_scale.dtype == torch.float32 and w_scale.numel() == N and w_scale.is_contiguous()
    d = GemmDesc()
    d.M, d.N, d.K, d.batch = M, N, K, ab
    d.lda, d.ldw, d.ldc = K, K, ldc
    d.a_batch_stride, d.c_batch_stride = M * K, c_bs
    d.ldres, d.res_batch_stride = 0, 0
    if res is not None:
        rb, Mr, Nr, r_bs, ldres = _mat(res, "res")
        if (Mr, Nr) != (M, N) or rb not in (1, ab):
            raise ValueError("res shape mismatch")
        d.ldres, d.res_batch_stride = ldres, (r_bs if rb == ab else 0)
    d.gate_batch_stride, d.gate_split, d.act = gate_batch_stride, gate_split, ACT[act]
    d.n_split, d.c_split_stride = (0, 0) if split is None else split
    d.bias_rowscale, d.alpha = None, float(alpha)
    name = "bya_gemm_fp8"
    if _SHAPE_LABELS:
        name += f":{ab}x{M}x{N}x{K}:{act or 'none'}{'+gate' if gate0 is not None else ''}{'+res' if res is not None else ''}"
    tok = _begin(name, 2.0 * ab * M * N * K)
    check(lib.bya_gemm_fp8(_p(a8), _p(a_scale), _p(w8), _p(w_scale), _p(bias), _p(out), _p(res), _p(gate0), _p(gate1),
                           ctypes.byref(d), _stream()), "bya_gemm_fp8")
    _end(tok)
    return out


def linear_small_m(x, w, bias, out, silu_in=False, act_out=None):
    """out[M<=8, N] = f(x) @ w.T + bias (weight-streaming kernel)."""
    lib = _hip.load()
    M, K = x.shape
    N = w.shape[0]
    assert x.is_contiguous() and w.is_contiguous() and out.is_contiguous() and out.shape == (M, N)
    assert x.dtype == w.dtype == out.dtype == torch.bfloat16 and w.shape[1] == K
    tok = _begin("bya_linear_small_m")
    check(lib.bya_linear_small_m(_p(x), _p(w), _p(bias), _p(out), M, N, K, int(silu_in), ACT[act_out], _stream()),
          "bya_linear_small_m")
    _end(tok)
    return out


def timestep_features(timesteps, out, flip_sin_to_cos=True, freq_shift=0.0):
    lib = _hip.load()
    assert timesteps.dtype == torch.int64 and timesteps.is_cuda and out.dtype == torch.bfloat16
    b, dim = out.shape
    tok = _begin("bya_timestep_features")
    check(lib.bya_timestep_features(_p(timesteps), _p(out), b, dim, int(flip_sin_to_cos), float(freq_shift),
                                    _stream()), "bya_timestep_features")
    _end(tok)
    return out


def layernorm(x, out, weight=None, bias=None, eps=1e-5, shift0=None, scale0=None, shift1=None, scale1=None,
              split=0, mod_batch_stride=0):
    """LayerNorm over the last dim (+ affine, + AdaLN modulation: rows < split use (shift0, scale0))."""
    lib = _hip.load()
    xb, rows, D, x_bs, ldx = _mat(x, "x")
    ob, rows_o, Do, y_bs, ldy = _mat(out, "out")
    assert (xb, rows, D) == (ob, rows_o, Do)
    tok = _begin("bya_layernorm")
    check(lib.bya_layernorm(_p(x), _p(out), _p(weight), _p(bias), _p(shift0), _p(scale0), _p(shift1), _p(scale1),
                            rows, xb, D, ldx, ldy, x_bs, y_bs, mod_batch_stride, split, float(eps), _stream()),
          "bya_layernorm")
    _end(tok)
    return out


def layernorm_fp8(x, q, q_scale, weight=None, bias=None, eps=1e-5, shift0=None, scale0=None, shift1=None, scale1=None,
                  split=0, mod_batch_stride=0):
    """``layernorm`` + ``quantize_rows_fp8`` of its output in one pass (same bytes, no bf16 round trip)."""
    lib = _hip.load()
    xb, rows, D, x_bs, ldx = _mat(x, "x")
    assert q.dtype == torch.uint8 and q.is_contiguous() and q.numel() == xb * rows * D
    assert q_scale.dtype == torch.float32 and q_scale.is_contiguous() and q_scale.numel() == xb * rows
    tok = _begin("bya_layernorm_fp8")
    check(lib.bya_layernorm_fp8(_p(x), _p(q), _p(q_scale), _p(weight), _p(bias), _p(shift0), _p(scale0), _p(shift1),
                                _p(scale1), rows, xb, D, ldx, D, x_bs, rows * D, mod_batch_stride, split, float(eps),
                                _stream()), "bya_layernorm_fp8")
    _end(tok)
    return q, q_scale


def qknorm_rope(q, k, qw, qb, kw, kb, cos, sin, heads, text_rows, eps=1e-6, k_scale=1.0, stats=None):
    """In place on q, k [B, S, heads*64]; q or k may be None (only the other one is processed).  ``stats``: fp32
    [slots, 2, B * heads], ZEROED by the caller: the kernel raises its entries to the squared norms of the rows it writes
    (max over the slots = max ||q||^2, max ||k||^2 per (batch, head): the data-dependent score bound of ``attention``)."""
    lib = _hip.load()
    b, S, _, bs, ld = _mat(q if q is not None else k, "q")
    assert q is None or k is None or _mat(k, "k") == _mat(q, "q")
    if stats is not None:
        assert stats.dtype == torch.float32 and stats.is_contiguous() and stats.dim() == 3 and stats.shape[1:] == (2, b * heads)
    if cos is not None:
        assert cos.dtype == torch.float32 and sin.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous()
        assert cos.shape == (S - text_rows, 64)
    tok = _begin("bya_qknorm_rope")
    check(lib.bya_qknorm_rope(_p(q), _p(k), _p(qw), _p(qb), _p(kw), _p(kb), _p(cos), _p(sin), b, S, heads, ld,
                              bs if b > 1 else 0, text_rows, float(eps), float(k_scale), _p(stats),
                              0 if stats is None else stats.shape[0], _stream()), "bya_qknorm_rope")
    _end(tok)


# (call tag, softmax variant) -> launches since the last reset; the variant is reported by the library itself
ATTN_VARIANT_NAMES = {0: "d64_running_max", 1: "d64_prescaled_running_max", 2: "d64_static_bound", 3: "d128_running_max",
                      4: "d64_static_bound_w4", 5: "d64_device_bound_w4"}
ATTN_BOUND_LIMIT = 90.0          # BYA_ATTN_BOUND_LIMIT (include/bya.h): |score| <= 90 keeps P = exp2(s) and its row sums normal
ATTN_VARIANTS = {}


def attention(q, k, v, out, *, head_dim, heads, nb1, nb2, Sq, Skv, q_strides, k_strides, v_strides, o_strides,
              scale, tag="other", prescaled=False, score_bound=0.0, bound=None):
    """Flash attention with explicit (level-1, level-2, row) element strides for q, k, v, out.
    ``bound`` = (stats, bh0, flags): the data-dependent score bound -- ``stats`` fp32 [slots, 2, n] as written by
    ``qknorm_rope(stats=...)``, this launch's (batch, head) index bh at column bh0 + bh, ``flags`` int32 [nb1 * nb2 * heads]
    (scratch: which heads went to the running-maximum kernel)."""
    lib = _hip.load()
    d = AttnDesc()
    d.head_dim, d.heads, d.nb1, d.nb2, d.Sq, d.Skv = head_dim, heads, nb1, nb2, Sq, Skv
    d.q_s1, d.q_s2, d.q_row = q_strides
    d.k_s1, d.k_s2, d.k_row = k_strides
    d.v_s1, d.v_s2, d.v_row = v_strides
    d.o_s1, d.o_s2, d.o_row = o_strides
    d.scale = float(scale)
    d.scores_prescaled = int(prescaled)
    d.score_bound = float(score_bound)
    for t in (q, k, v, out):
        assert t.dtype == torch.bfloat16 and t.is_cuda
    if bound is not None:
        stats, bh0, flags = bound
        assert stats.dtype == torch.float32 and stats.is_contiguous() and stats.dim() == 3 and stats.shape[1] == 2
        assert flags.dtype == torch.int32 and flags.is_contiguous() and flags.numel() >= nb1 * nb2 * heads
        d.bound_dev, d.bound_slots, d.bound_heads, d.bound_bh0 = stats.data_ptr(), stats.shape[0], stats.shape[2], int(bh0)
        d.fallback_flags = flags.data_ptr()
    if prescaled and (score_bound > 0 or bound is not None) and q.device.index not in _ATTN_WS:
        ensure_attn_workspace(q.device)
    var = ATTN_VARIANT_NAMES.get(lib.bya_attn_variant(ctypes.byref(d)), "rejected")
    ATTN_VARIANTS[tag, var] = ATTN_VARIANTS.get((tag, var), 0) + 1
    label = "bya_attn_fwd:" + tag
    if _SHAPE_LABELS:
        label += f":{nb1}x{nb2}x{heads}h_q{Sq}_kv{Skv}_d{head_dim}"
    tok = _begin(label, 4.0 * nb1 * nb2 * heads * Sq * Skv * head_dim)
    check(lib.bya_attn_fwd(_p(q), _p(k), _p(v), _p(out), ctypes.byref(d), _stream()), "bya_attn_fwd")
    _end(tok)
    return out


def self_attention(q, k, v, out, heads, head_dim=64, scale=None, tag="other", prescaled=False, score_bound=0.0, bound=None):
    """q,k,v,out: [B, S, heads*head_dim] views (row-strided ok)."""
    b, S, _, q_bs, q_ld = _mat(q, "q")
    _, Skv, _, k_bs, k_ld = _mat(k, "k")
    _, _, _, v_bs, v_ld = _mat(v, "v")
    _, _, _, o_bs, o_ld = _mat(out, "out")
    scale = head_dim ** -0.5 if scale is None else scale
    return attention(q, k, v, out, head_dim=head_dim, heads=heads, nb1=b, nb2=1, Sq=S, Skv=Skv,
                     q_strides=(q_bs, 0, q_ld), k_strides=(k_bs, 0, k_ld), v_strides=(v_bs, 0, v_ld),
                     o_strides=(o_bs, 0, o_ld), scale=scale, tag=tag, prescaled=prescaled, score_bound=score_bound, bound=bound)


def attn_kv_mix(q, k, v, r, af, z, wsum=None, *, head_dim, heads, n_id, n_grp, Sq, Skv, q_strides, k_strides, v_strides,
                z_strides, scale):
    """z[g, n] = sum_id w[g n, id] * softmax(q[g, n] . K[id, g]^T * scale) V[id, g]: cross-attention onto <= 64 keys per identity
    with the router's masked combine in its epilogue (bya_attn_kv_mix).  q_strides = (group, row), k / v_strides = (identity,
    group, row), z_strides = (group, row), in elements; r: bf16 [n_grp * Sq, n_id]; af: None (face) or bf16 [n_id, n_id]."""
    lib = _hip.load()
    d = AttnMixDesc()
    d.head_dim, d.heads, d.n_id, d.n_grp, d.Sq, d.Skv = head_dim, heads, n_id, n_grp, Sq, Skv
    d.q_grp, d.q_row = q_strides
    d.k_id, d.k_grp, d.k_row = k_strides
    d.v_id, d.v_grp, d.v_row = v_strides
    d.z_grp, d.z_row = z_strides
    d.scale = float(scale)
    for t in (q, k, v, z, r):
        assert t.dtype == torch.bfloat16 and t.is_cuda
    assert r.is_contiguous() and r.numel() == n_grp * Sq * n_id
    assert af is None or (af.is_contiguous() and af.dtype == torch.bfloat16 and af.numel() == n_id * n_id)
    assert wsum is None or (wsum.dtype == torch.float32 and wsum.is_contiguous() and wsum.numel() >= n_grp * Sq)
    tok = _begin("bya_attn_kv_mix", 4.0 * n_id * n_grp * heads * Sq * Skv * head_dim)
    check(lib.bya_attn_kv_mix(_p(q), _p(k), _p(v), _p(r), _p(af), _p(z), _p(wsum), ctypes.byref(d), _stream()), "bya_attn_kv_mix")
    _end(tok)
    return z


def attn_tiny(q, k, v, out, L, heads, n_outer, n_inner, outer_stride, seq_stride, ld_qkv, ld_o, scale):
    lib = _hip.load()
    tok = _begin("bya_attn_tiny")
    check(lib.bya_attn_tiny(_p(q), _p(k), _p(v), _p(out), L, heads, n_outer, n_inner, outer_stride, seq_stride,
                            ld_qkv, ld_o, float(scale), _stream()), "bya_attn_tiny")
    _end(tok)
    return out


def router_scores(qr, kr, ln_w, ln_b, pos_emb, out, n_id, N, eps=1e-5):
    lib = _hip.load()
    for t in (qr, kr, ln_w, ln_b, pos_emb, out):
        assert t.is_contiguous() and t.dtype == torch.bfloat16
    tok = _begin("bya_router_scores", 2.0 * n_id * N * 32 * qr.shape[-1])
    check(lib.bya_router_scores(_p(qr), _p(kr), _p(ln_w), _p(ln_b), _p(pos_emb), _p(out), n_id, N, 16, 32,
                                float(eps), _stream()), "bya_router_scores")
    _end(tok)
    return out


def router_head(x, w, b, r, n_id, N):
    lib = _hip.load()
    assert x.is_contiguous() and r.is_contiguous()
    tok = _begin("bya_router_head")
    check(lib.bya_router_head(_p(x), _p(w), _p(b), _p(r), n_id, N, x.shape[-1], _stream()), "bya_router_head")
    _end(tok)
    return r


def forcing_max_over_frames(forcing, out, frames, per_frame, n_id):
    lib = _hip.load()
    assert forcing.is_contiguous() and out.is_contiguous() and forcing.dtype == out.dtype == torch.bfloat16
    tok = _begin("bya_forcing_max_over_frames")
    check(lib.bya_forcing_max_over_frames(_p(forcing), _p(out), frames, per_frame, n_id, _stream()),
          "bya_forcing_max_over_frames")
    _end(tok)
    return out


def masked_combine(x, feat, r, af, mode, alpha=1.0):
    """x [B, N, D] view (in place) += combine(feat [B, n_id, N, D], r [B or 1, N, n_id])."""
    lib = _hip.load()
    b, N, D, x_bs, x_row = _mat(x, "x")
    assert feat.is_contiguous() and feat.shape[0] == b and feat.shape[2] == N and feat.shape[3] == D
    n_id = feat.shape[1]
    assert r.is_contiguous() and r.shape[-2:] == (N, n_id) and r.dtype == torch.bfloat16
    r_bs = 0 if r.shape[0] == 1 else N * n_id
    if af is not None:
        assert af.is_contiguous() and af.dtype == torch.bfloat16 and af.shape == (b, n_id, n_id)
    tok = _begin("bya_masked_combine")
    check(lib.bya_masked_combine(_p(x), _p(feat), _p(r), _p(af), {"face": 0, "audio": 1}[mode], float(alpha), b,
                                 n_id, N, D, x_row, x_bs, r_bs, _stream()), "bya_masked_combine")
    _end(tok)
    return x


def routed_mix(feat, r, af, mode, z, wsum=None):
    """z [B, N, D] = sum_id w[b,n,id] * feat [B, n_id, N, D]; wsum [B, N] fp32 = sum_id w (optional)."""
    lib = _hip.load()
    b, n_id, N, D = feat.shape
    assert feat.is_contiguous() and z.is_contiguous() and z.shape == (b, N, D)
    assert r.is_contiguous() and r.shape[-2:] == (N, n_id) and r.dtype == torch.bfloat16
    r_bs = 0 if r.shape[0] == 1 else N * n_id
    if wsum is not None:
        assert wsum.dtype == torch.float32 and wsum.is_contiguous() and wsum.numel() == b * N
    tok = _begin("bya_routed_mix")
    check(lib.bya_routed_mix(_p(feat), _p(r), _p(af), _p(z), _p(wsum), {"face": 0, "audio": 1}[mode], b, n_id, N, D,
                             r_bs, _stream()), "bya_routed_mix")
    _end(tok)
    return z


def patchify(x, cols):
    lib = _hip.load()
    b, t, c, h, w = x.shape
    assert x.is_contiguous() and cols.is_contiguous() and x.dtype == cols.dtype == torch.bfloat16
    tok = _begin("bya_patchify")
    check(lib.bya_patchify(_p(x), _p(cols), b, t, c, h, w, _stream()), "bya_patchify")
    _end(tok)
    return cols


def unpatchify(y, out):
    lib = _hip.load()
    b, t, c, h, w = out.shape
    assert y.is_contiguous() and out.is_contiguous() and y.dtype == out.dtype == torch.bfloat16
    tok = _begin("bya_unpatchify")
    check(lib.bya_unpatchify(_p(y), _p(out), b, t, c, h, w, _stream()), "bya_unpatchify")
    _end(tok)
    return out


def act_add(x, out, act=None, res=None):
    lib = _hip.load()
    assert x.is_contiguous() and out.is_contiguous() and (res is None or res.is_contiguous())
    tok = _begin("bya_act_add")
    check(lib.bya_act_add(_p(x), _p(res), _p(out), x.numel(), ACT[act], _stream()), "bya_act_add")
    _end(tok)
    return out


def cfg_scheduler_step(pred, sample, coef, old_x0=None, noise=None, x0_out=None, out=None):
    """Fused CFG combine + scheduler step (reference models/pipeline_bindyouravatar.py:924-948).
    pred: bf16 [1 or 2, ...] model output ([uncond, cond] when 2); sample: bf16 latents [1, ...] (or any shape with the
    element count of one prediction); coef: dict with the fields of ``bya_sched_coef``; old_x0 / x0_out: fp32;
    noise: bf16.  Returns the new bf16 latents."""
    import ctypes
    lib = _hip.load()
    n = sample.numel()
    n_pred = pred.shape[0] if pred.numel() != n else 1
    assert pred.dtype == sample.dtype == torch.bfloat16 and pred.is_contiguous() and sample.is_contiguous()
    assert n_pred in (1, 2) and pred.numel() == n_pred * n
    for t, dt in ((old_x0, torch.float32), (noise, torch.bfloat16), (x0_out, torch.float32)):
        assert t is None or (t.dtype == dt and t.is_contiguous() and t.numel() == n)
    if out is None:
        out = torch.empty_like(sample)
    c = _hip.SchedCoef(**{k: float(v) for k, v in coef.items()})
    tok = _begin("bya_cfg_scheduler_step")
    check(lib.bya_cfg_scheduler_step(_p(pred), n_pred, n, _p(sample), _p(old_x0), _p(noise), _p(out), _p(x0_out), n,
                                     ctypes.byref(c), _stream()), "bya_cfg_scheduler_step")
    _end(tok)
    return out


def masks_to_routing_logits(masks, frames=13, h=30, w=45, out=None):
    """Tracking masks uint8 [n_id, T, H, W] (> 0 = foreground) -> ``routing_logits_forcing`` bf16 [1, frames*h*w, n_id]
    (reference util/utils.py:871-936; feed it to ``forward(routing_logits_forcing=...)``)."""
    lib = _hip.load()
    assert masks.dtype == torch.uint8 and masks.is_contiguous() and masks.dim() == 4
    n_id, Ti, Hi, Wi = masks.shape
    if out is None:
        out = torch.empty(1, frames * h * w, n_id, dtype=torch.bfloat16, device=masks.device)
    tok = _begin("bya_masks_to_routing_logits")
    check(lib.bya_masks_to_routing_logits(_p(masks), _p(out), n_id, Ti, Hi, Wi, frames, h, w, _stream()),
          "bya_masks_to_routing_logits")
    _end(tok)
    return out


def pack_rowgemm512(weight, bias, ln_weight=None, ln_bias=None):
    """Pack a [N, 512] Linear (optionally preceded by LayerNorm(512)) for ``rowgemm512``: gamma folded into the bf16
    weight, its fp32 row sums, and the fp32 constant vector  W . beta + bias  (see include/bya.h)."""
    w32 = weight.float()
    b32 = bias.float() if bias is not None else torch.zeros(weight.shape[0], device=weight.device)
    if ln_weight is None:
        return dict(w=weight.to(torch.bfloat16).contiguous(), colsum=None, cvec=b32.contiguous(), ln=False)
    wg = (w32 * ln_weight.float()[None, :]).to(torch.bfloat16).contiguous()
    return dict(w=wg, colsum=wg.float().sum(1).contiguous(), cvec=(w32 @ ln_bias.float() + b32).contiguous(), ln=True)


def rowgemm512(x, pack, out, res=None, act=None, eps=1e-5, nsplit=0):
    """out = res + act( LN?(x) @ W.T + b ) for K = 512 (router projections, reference models/router.py:468-493)."""
    lib = _hip.load()
    M, K = x.shape
    N = pack["w"].shape[0]
    assert K == 512 and x.stride(1) == 1 and out.stride(1) == 1 and out.shape == (M, N)
    assert res is None or (res.shape == out.shape and res.stride(1) == 1)
    tok = _begin("bya_rowgemm512", 2.0 * M * N * K)
    check(lib.bya_rowgemm512(_p(x), _p(pack["w"]), _p(pack["colsum"]), _p(pack["cvec"]), _p(res), _p(out), M, N,
                             x.stride(0), out.stride(0), res.stride(0) if res is not None else 0, int(pack["ln"]),
                             float(eps), ACT[act], int(nsplit), _stream()), "bya_rowgemm512")
    _end(tok)
    return out


def router_group_attn(x, pack, out, L, n_outer, n_inner, outer_stride, seq_stride, eps=1e-5, scale=0.125):
    """out = softmax(q k^T * scale) v per group of L rows, (q | k | v) = LN(x) @ Wqkv.T + b, 8 heads x 64: the temporal /
    multi-ID attention of SpatialTemporalAttentionBlock up to its out-projection in ONE launch (reference
    models/router.py:476-478, :482-484; include/bya.h bya_router_group_attn).  ``pack`` = ``pack_rowgemm512`` of the
    concatenated to_q | to_k | to_v with the LayerNorm folded in; groups as for ``attn_tiny``."""
    lib = _hip.load()
    M, K = x.shape
    assert K == 512 and pack["ln"] and pack["w"].shape == (1536, 512) and out.shape == (M, 512)
    assert x.stride(1) == 1 and out.stride(1) == 1 and x.data_ptr() != out.data_ptr()
    tok = _begin("bya_router_group_attn", 2.0 * M * 1536 * K + 4.0 * M * L * 512)
    check(lib.bya_router_group_attn(_p(x), _p(pack["w"]), _p(pack["colsum"]), _p(pack["cvec"]), _p(out), M, x.stride(0),
                                    out.stride(0), int(L), int(n_outer), int(n_inner), int(outer_stride), int(seq_stride),
                                    float(eps), float(scale), _stream()), "bya_router_group_attn")
    _end(tok)
    return out


def router_mlp_fused(x, pack1, pack2, eps=1e-5, out=None, tiles_pass0=0):
    """x += mlp[2](GELU(mlp[0](LayerNorm(x)))) on router rows in ONE launch, the hidden activation in registers (reference
    models/router.py:491; include/bya.h bya_router_mlp_fused).  ``pack1`` = ``pack_rowgemm512`` of mlp[0] with norm4 folded
    in, ``pack2`` of mlp[2].  Bit-identical to ``rowgemm512(x, pack1, h, act="gelu_erf"); rowgemm512(h, pack2, x, res=x)``."""
    lib = _hip.load()
    M, K = x.shape
    out = x if out is None else out
    assert K == 512 and pack1["ln"] and not pack2["ln"] and pack1["w"].shape == (512, 512) and pack2["w"].shape == (512, 512)
    assert x.stride(1) == 1 and out.stride(1) == 1 and out.shape == x.shape
    tok = _begin("bya_router_mlp_fused", 4.0 * M * 512 * K)
    check(lib.bya_router_mlp_fused(_p(x), _p(pack1["w"]), _p(pack1["colsum"]), _p(pack1["cvec"]), _p(pack2["w"]),
                                   _p(pack2["cvec"]), _p(out), M, x.stride(0), out.stride(0), float(eps), int(tiles_pass0),
                                   _stream()), "bya_router_mlp_fused")
    _end(tok)
    return out


def router_group_attn_out(x, pack, pack_out, L, n_outer, n_inner, outer_stride, seq_stride, eps=1e-5, scale=0.125, out=None,
                          tiles_pass0=0):
    """x += to_out(attention over groups of L rows of LayerNorm(x)) in ONE launch (reference models/router.py:482-483,
    :486-487; include/bya.h bya_router_group_attn_out): ``router_group_attn`` followed by the out-projection + residual, the
    attention output in registers.  L <= 16.  Bit-identical to the two launches."""
    lib = _hip.load()
    M, K = x.shape
    out = x if out is None else out
    assert K == 512 and pack["ln"] and pack["w"].shape == (1536, 512) and not pack_out["ln"] and pack_out["w"].shape == (512, 512)
    assert x.stride(1) == 1 and out.stride(1) == 1 and out.shape == x.shape
    tok = _begin("bya_router_group_attn_out", 2.0 * M * 2048 * K + 4.0 * M * L * 512)
    check(lib.bya_router_group_attn_out(_p(x), _p(pack["w"]), _p(pack["colsum"]), _p(pack["cvec"]), _p(pack_out["w"]),
                                        _p(pack_out["cvec"]), _p(out), M, x.stride(0), out.stride(0), int(L), int(n_outer),
                                        int(n_inner), int(outer_stride), int(seq_stride), float(eps), float(scale),
                                        int(tiles_pass0), _stream()), "bya_router_group_attn_out")
    _end(tok)
    return out


# ---- video VAE (csrc/vae.hip) ----------------------------------------------------------
def vae_patches(x, cache, out, KT, stride, pad, up, tmode, Ho, Wo, t0, nt):
    """Patch matrix of a causal KT x 3 x 3 convolution over channels-last x [Ts, Hs, Ws, C] -> out [nt * Ho * Wo, Kpad]."""
    lib = _hip.load()
    Ts, Hs, Ws, C = x.shape
    assert x.is_contiguous() and out.is_contiguous() and x.dtype == out.dtype == torch.bfloat16
    assert cache is None or (cache.is_contiguous() and tuple(cache.shape) == (KT - 1, Hs, Ws, C))
    assert out.shape[0] == nt * Ho * Wo
    tok = _begin("bya_vae_patches")
    check(lib.bya_vae_patches(_p(x), _p(cache), _p(out), Ts, Hs, Ws, C, KT, stride, pad, int(up), tmode, Ho, Wo, t0, nt,
                              out.shape[1], _stream()), "bya_vae_patches")
    _end(tok)
    return out


def vae_groupnorm_stats(x2d, sums, groups, partial=None):
    lib = _hip.load()
    rows, C = x2d.shape
    assert x2d.is_contiguous() and sums.dtype == torch.float32 and sums.numel() == 2 * groups
    need = (rows + 511) // 512 * groups * 2
    if partial is None:
        partial = torch.empty(need, dtype=torch.float32, device=x2d.device)
    assert partial.dtype == torch.float32 and partial.numel() >= need and partial.is_contiguous()
    tok = _begin("bya_vae_groupnorm_stats")
    check(lib.bya_vae_groupnorm_stats(_p(x2d), _p(sums), _p(partial), rows, C, groups, _stream()), "bya_vae_groupnorm_stats")
    _end(tok)
    return sums


def vae_norm_act(x, y, sums, gamma, beta, groups, act="silu", eps=1e-6, zy=None, zb=None, latent_shape=None, tmode=1,
                 out_pad=False):
    """x: [T, H, W, C] channels-last; y: the same, or (out_pad) the zero-padded conv input [T + 2, H + 2, W + 2, C] whose
    interior frames 2.. are written; zy / zb: [Tz * hz * wz, C] views (row stride = their .stride(0))."""
    lib = _hip.load()
    T, H, W, C = x.shape
    assert x.is_contiguous() and y.is_contiguous()
    assert tuple(y.shape) == ((T + 2, H + 2, W + 2, C) if out_pad else (T, H, W, C))
    Tz, hz, wz = latent_shape if latent_shape is not None else (T, H, W)
    ldz = zy.stride(0) if zy is not None else 0
    tok = _begin("bya_vae_norm_act")
    check(lib.bya_vae_norm_act(_p(x), _p(y), _p(sums), _p(gamma), _p(beta), _p(zy), _p(zb), T * H * W, C, groups,
                               {None: 0, "none": 0, "silu": 1}[act], float(eps), T, H, W, Tz, hz, wz, tmode, ldz,
                               int(bool(out_pad)), _stream()), "bya_vae_norm_act")
    _end(tok)
    return y


def vae_conv3d(xpad, w, bias, out, res=None, KT=3):
    """Causal KT x 3 x 3 convolution as an implicit GEMM (bya_vae_conv3d): xpad [To + KT - 1, H + 2, W + 2, C] zero-padded (KT = 3:
    its two context frames in front), w [Cout, >= 9 KT C] (tap-major columns), out / res [To, H, W, Cout] (out = res + bias + conv)."""
    lib = _hip.load()
    Tp, Hp, Wp, C = xpad.shape
    To, H, W, Cout = out.shape
    assert (Tp, Hp, Wp) == (To + KT - 1, H + 2, W + 2) and xpad.is_contiguous() and w.is_contiguous() and w.shape[0] >= Cout
    assert out.stride(3) == 1
    ldc = out.stride(2)
    assert out.stride(1) == W * ldc and out.stride(0) == H * W * ldc
    ldres = 0
    if res is not None:
        ldres = res.stride(2)
        assert res.shape == out.shape and res.stride(1) == W * ldres and res.stride(0) == H * W * ldres and res.stride(3) == 1
    tok = _begin("bya_vae_conv3d", 2.0 * To * H * W * Cout * 9 * KT * C)
    check(lib.bya_vae_conv3d(_p(xpad), _p(w), _p(bias), _p(res), _p(out), To, H, W, C, Cout, KT, w.stride(0), ldc, ldres,
                             _stream()), "bya_vae_conv3d")
    _end(tok)
    return out


def vae_upsample_pad(x, ypad, tmode):
    """Nearest up-sampling of x [T, H, W, C] into the interior of the zero-padded ypad [To, 2 H + 2, 2 W + 2, C]."""
    lib = _hip.load()
    T, H, W, C = x.shape
    To = T if tmode == 0 else (2 * T if tmode == 1 else 2 * T - 1)
    assert x.is_contiguous() and ypad.is_contiguous() and tuple(ypad.shape) == (To, 2 * H + 2, 2 * W + 2, C)
    tok = _begin("bya_vae_upsample_pad")
    check(lib.bya_vae_upsample_pad(_p(x), _p(ypad), T, H, W, C, tmode, _stream()), "bya_vae_upsample_pad")
    _end(tok)
    return ypad
