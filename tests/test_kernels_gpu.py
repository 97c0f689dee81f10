"""Per-kernel parity on a real MI355X: every C-ABI entry point against an fp32 torch restatement of the
reference op it replaces, on the same seeded bf16 inputs.

Tolerance (stated per the north star, "within 1e-3 rel of reference"): the kernels write bf16, so the
comparison target is the fp32 reference result ROUNDED to bf16 (an exact kernel scores 0);
``relative Frobenius error <= 1e-3`` unless a test states otherwise.  Index-only kernels are bit-exact.
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_fro

pytestmark = pytest.mark.gpu

TOL = 1e-3
# Attention rounds the softmax probabilities P to bf16 for the P.V MFMA (exactly what the flash SDPA the
# reference dispatches to does); with few keys the per-element 2^-9 rounding of P does not average out, so the
# bound is 3e-3 (measured 2.2e-3 at 64 keys, falling with sequence length).
ATTN_TOL = 3e-3


def bf(t):
    return t.to(torch.bfloat16)


def rnd(shape, dev, seed, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return bf(torch.randn(shape, generator=g) * std).to(dev)


def check(gpu, ref32, tol=TOL, what=""):
    err = rel_fro(gpu.float(), bf(ref32).float())
    print(f"{what}: rel-Fro vs bf16(fp32 ref) = {err:.3e}")
    assert err <= tol, f"{what}: {err:.3e} > {tol}"
    assert torch.isfinite(gpu.float()).all()


@pytest.fixture(scope="module")
def ops():
    from bind_your_avatar_implementation_amd import ops
    return ops


# ----------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (300, 512, 512), (17, 64, 128),
                                   (1350, 3072, 768), (2222, 2048, 3072), (130, 12288, 3072),
                                   (520, 87552, 512)])     # few rows, short K, 1026 tiles: the persistent kernel (a deep grid)
def test_gemm_plain(ops, dev, M, N, K):
    a, w = rnd((M, K), dev, 1), rnd((N, K), dev, 2, K ** -0.5)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(a, w, out)
    check(out, a.float() @ w.float().T, what=f"gemm {M}x{N}x{K}")


@pytest.mark.parametrize("fp8", [False, True])
def test_gemm_output_past_the_2gib_reach_of_the_epilogue_descriptors(ops, dev, fp8):
    """The GEMM epilogues address C and the residual as base + 32-bit byte offset under a 2 GiB buffer descriptor
    (csrc/gemm_common.h): 97 frames at 720 x 1280 are 90226 joint rows, whose MLP activation [90226, 12288] bf16 is 2.2 GB.
    Rows whose byte offset reaches 2^31 used to be dropped on store and read as zero.  The host side now cuts such a
    launch into row chunks: (1) FF1 shape just over the limit -- the rows past 2 GiB are written and correct; (2) a
    gate + residual launch whose output is a strided view (ldc = 12288) reaching past 2 GiB, residual aliasing the
    output, text/video gate split in the first chunk."""
    M, N, K = 87381 + 1100, 12288, 256               # row 87381 is the first whose offset reaches 2^31
    a, w, b = rnd((M, K), dev, 41), rnd((N, K), dev, 42, K ** -0.5), rnd((N,), dev, 43)
    out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=dev)
    if fp8:
        a8, sa = ops.quantize_rows_fp8(a)
        w8, sw = ops.quantize_rows_fp8(w)
        run = lambda x8, sx, o, **kw: ops.gemm_fp8(x8, sx, w8, sw, o, **kw)
        deq = lambda q, sc: q.view(torch.float8_e4m3fn).float() * sc[:, None]
        af, wf = deq(a8, sa), deq(w8, sw)
        run(a8, sa, out, bias=b, act="gelu_tanh")
    else:
        af, wf = a.float(), w.float()
        ops.gemm(a, w, out, bias=b, act="gelu_tanh")
    for lo, hi in ((0, 512), (87381 - 300, 87381 + 300), (M - 500, M)):
        ref = F.gelu(af[lo:hi] @ wf.T + b.float(), approximate="tanh")
        check(out[lo:hi], ref, tol=2e-3, what=f"{'fp8 ' if fp8 else ''}gemm rows {lo}..{hi} of {M} x {N}")
    assert not bool((out[87381:] == 7.0).all(dim=1).any()), "rows past the 2 GiB reach were not written"
    # (2) in-place gated residual on the first 3072 columns of the same wide buffer
    n2 = 3072
    x0 = out[:, :n2].clone()
    w2, g0, g1 = rnd((n2, K), dev, 44, K ** -0.5), rnd((n2,), dev, 45), rnd((n2,), dev, 46)
    view = out[:, :n2]
    kw = dict(bias=b[:n2].contiguous(), res=view, gate0=g0, gate1=g1, gate_split=226)
    if fp8:
        w28, sw2 = ops.quantize_rows_fp8(w2)
        ops.gemm_fp8(a8, sa, w28, sw2, view, **kw)
        w2f = w28.view(torch.float8_e4m3fn).float() * sw2[:, None]
    else:
        ops.gemm(a, w2, view, **kw)
        w2f = w2.float()
    for lo, hi in ((0, 512), (87381 - 300, 87381 + 300), (M - 500, M)):
        y = af[lo:hi] @ w2f.T + b[:n2].float()
        rows = torch.arange(lo, hi, device=dev)[:, None]
        ref = x0[lo:hi].float() + torch.where(rows < 226, g0.float()[None], g1.float()[None]) * y
        check(out[lo:hi, :n2], ref, tol=2e-3, what=f"{'fp8 ' if fp8 else ''}gated residual rows {lo}..{hi}")
    assert ops.gemm_workspace_status() == 0


@pytest.mark.parametrize("M,N,K,epi", [
    (17776, 3072, 3072, "gate_res"),      # attention out-projection: 840 tiles = 3.28 rounds -> 9 leftover tiles per XCD x 3
    (17550, 3072, 3072, "res"),           # audio out-projection: 828 tiles, XCDs with 7 and with 8 leftover tiles (x 4)
    (17776, 3072, 12288, "gate_res"),     # FF2
    (2222, 3072, 3072, "res"),            # one rank's rows of an 8-GPU step: 108 tiles on 256 CUs, every tile split in 2
    (2222, 9216, 3072, "split3"),         # its packed q|k|v projection: one full round + leftovers
    (1024, 1024, 2048, "gelu"),           # 16 tiles: 2 per XCD, 4 K-ranges each
    (4000, 1536, 1024, "plain"),          # ragged M (4000 = 15.6 tiles), 16 K-tiles: exactly 2 ranges of 8
    (2222, 3072, 12288, "gate_res"),      # one rank's FF2
])
def test_gemm_split_k_last_round(ops, dev, M, N, K, epi, lib_options):
    """With option gemm_splitk = 1 (the default until the end of round 6; now the row plan of gemm.hip does the same job without a
    hand-off) the persistent kernel cuts the last, partial round of 256 x 256 tiles along K when the split-K workspace is
    registered (ops.gemm registers it): partial sums travel through fp32 slabs between workgroups (write-through stores,
    agent-scope counter).  Against the fp32 product with the usual bar, against the unsplit kernel (option gemm_splitk = 0:
    same products, only the fp32 summation order of a split tile differs) and 12 repeats bit-identical on a busy GPU
    (a wrong wait, a stale slab or a counter that is not reset shows as rare wrong tiles).  By default only K-ranges of
    40+ K-tiles are split (K = 12288: the exchange costs ~20 us); option gemm_splitk_min = 8 makes every shape here split."""
    lib_options(gemm_splitk=1, gemm_splitk_min=8)
    a, w, b = rnd((M, K), dev, 31), rnd((N, K), dev, 32, K ** -0.5), rnd((N,), dev, 33, 0.5)
    kw, ref = {}, a.float() @ w.float().T + b.float()
    res = rnd((M, N), dev, 34)
    gate = rnd((2, N), dev, 35)
    n_out = N
    if epi == "gelu":
        kw, ref = dict(act="gelu_tanh"), F.gelu(ref, approximate="tanh")
    elif epi == "res":
        kw, ref = dict(res=res), ref + res.float()
    elif epi == "gate_res":
        kw = dict(res=res, gate0=gate[0], gate1=gate[1], gate_split=226)
        g = torch.where((torch.arange(M, device=dev) < 226)[:, None], gate[0].float()[None], gate[1].float()[None])
        ref = ref * g + res.float()
    def run():
        if epi == "split3":
            out = torch.empty(3, M, N // 3, dtype=torch.bfloat16, device=dev)
            ops.gemm(a, w, out[0], bias=b, split=(N // 3, M * (N // 3)))
            return out.permute(1, 0, 2).reshape(M, N)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ops.gemm(a, w, out, bias=b, **kw)
        return out
    first = run()
    check(first, ref, what=f"split-K gemm {M}x{N}x{K} {epi}")
    with ops.options(gemm_splitk=0):
        unsplit = run()
    d = (first.float() - unsplit.float()).abs()
    frac = (d > 0).float().mean().item()
    print(f"split vs unsplit: {frac * 100:.2f} % of the outputs differ, max {d.max().item():.3e} (one bf16 step of the value)")
    assert rel_fro(first.float(), unsplit.float()) < 2e-3
    x, y = rnd((8192, 8192), dev, 36), rnd((8192, 8192), dev, 37)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(4):
            x @ y
    outs = [run() for _ in range(12)]
    torch.cuda.synchronize()
    assert all(torch.equal(first, o) for o in outs)
    assert torch.equal(run(), first)
    # the finisher's bounded wait never gave up (it would have counted the event in the workspace: bya_gemm_workspace_status)
    assert ops.gemm_workspace_status() == 0
    ops.check_gemm_workspace()


def test_gemm_split_k_inside_a_replayed_hip_graph(ops, dev, lib_options):
    """The split-K hand-off keeps state in the workspace (slabs, counters the finisher puts back to zero): a captured
    hipGraph that contains a split launch must replay bit-identically, on new input values too, and interleaved with eager
    launches of another split shape."""
    lib_options(gemm_splitk=1)
    M, N, K = 17776, 3072, 12288
    a, w = rnd((M, K), dev, 41), rnd((N, K), dev, 42, K ** -0.5)
    a2 = rnd((M, K), dev, 43)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(a, w, out)
    eager1 = out.clone()
    static_a = a.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.gemm(static_a, w, out)                      # warm-up on a side stream, as torch's capture recipe prescribes
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        ops.gemm(static_a, w, out)
    for _ in range(3):
        out.zero_()
        g.replay()
        assert torch.equal(out, eager1)
    small_a, small_w = rnd((2222, K), dev, 44), rnd((N, K), dev, 45, K ** -0.5)
    small_out = torch.empty(2222, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(small_a, small_w, small_out)               # another split shape in between (same counters / slabs)
    small_ref = small_out.clone()
    static_a.copy_(a2)
    g.replay()
    replay2 = out.clone()
    ops.gemm(a2, w, out)
    assert torch.equal(replay2, out)
    ops.gemm(small_a, small_w, small_out)
    assert torch.equal(small_out, small_ref)
    # no finisher gave up (a timed-out finisher still adds slabs that landed long before, so equality alone proves nothing)
    torch.cuda.synchronize()
    assert ops.gemm_workspace_status() == 0
    ops.check_gemm_workspace()


def test_gemm_several_split_launches_in_one_replayed_hip_graph(ops, dev, lib_options):
    """A hipGraph replays its launches with the launch epochs baked in at capture.  With several split launches in one
    graph the counter words, shared by all of them, carry the LAST launch's epoch when the next replay starts: the first
    launch's writers must still be able to claim them (an idle word is claimable whatever its epoch).  Before that rule
    every replay after the first stalled ~1 s per split launch in the finisher's bounded wait and raised
    bya_gemm_workspace_status.  Checks: bit-identical replays, status 0, and a wall-time bound far below one time-out."""
    import time
    lib_options(gemm_splitk=1)
    M, N, K = 17776, 3072, 12288
    a = [rnd((M, K), dev, 51 + i) for i in range(3)]
    w = rnd((N, K), dev, 55, K ** -0.5)
    outs = [torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(3)]
    eager = []
    for x, o in zip(a, outs):
        ops.gemm(x, w, o)
        eager.append(o.clone())
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for x, o in zip(a, outs):
            ops.gemm(x, w, o)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for x, o in zip(a, outs):
            ops.gemm(x, w, o)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(4):
        for o in outs:
            o.zero_()
        g.replay()
        if rep == 1:                                    # an eager split launch between two replays stamps a newer epoch too
            tmp = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            ops.gemm(a[0], w, tmp)
            assert torch.equal(tmp, eager[0])
        for o, e in zip(outs, eager):
            assert torch.equal(o, e)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert ops.gemm_workspace_status() == 0, "a split-K finisher timed out inside a replayed graph"
    ops.check_gemm_workspace()
    assert dt < 0.5, f"4 replays of 3 split launches took {dt:.2f} s: a finisher sat in its bounded wait"


@pytest.mark.parametrize("B,M,N,K,act,res", [(1, 64, 2048, 2048, None, False), (1, 64, 4096, 2048, None, False),
                                               (2, 37, 1024, 4096, None, True), (2, 49, 512, 46080, "relu", False),
                                               (2, 37, 4096, 1024, "gelu_erf", False), (2, 12, 1536, 8192, None, False),
                                               (1, 1, 512, 256, "silu", True), (2, 33, 48, 1088, "gelu_tanh", True),
                                               (2, 24, 1024, 2048, None, False), (3, 40, 256, 512, None, True), (4, 16, 8192, 512, "relu", False)])
def test_gemm_skinny_rows_weight_streaming_kernel(ops, dev, monkeypatch, B, M, N, K, act, res):
    """Linears with at most 64 rows and up to 8192 outputs inside ``ops.weight_streaming()`` (the step-invariant conditioning)
    run on bya_gemm_skinny_bf16: one workgroup
    per 16 output columns (batch elements stacked as rows when they fit together, else one workgroup per batch element),
    16 waves split K, partial sums added in wave order.  Against fp32 at the usual
    per-kernel bar, against the tiled kernel (BYA_GEMM_SKINNY=0) within the same bar, repeatable bit for bit, strided A /
    out / res, rows that are no multiple of 16, every epilogue it takes over."""
    a_w = rnd((B, M, K + 64), dev, 21)
    a = a_w[..., 32:32 + K]
    w, b = rnd((N, K), dev, 22, K ** -0.5), rnd((N,), dev, 23, 0.5)
    r_w = rnd((B, M, N + 16), dev, 24)
    r = r_w[..., 8:8 + N] if res else None
    o_w = torch.full((B, M, N + 32), 3.0, dtype=torch.bfloat16, device=dev)
    out = o_w[..., 16:16 + N]
    def run():
        o_w.fill_(3.0)
        with ops.weight_streaming():
            ops.gemm(a, w, out, bias=b, act=act, res=r)
        torch.cuda.synchronize()
        return o_w.clone()
    got = run()
    y = a.float() @ w.float().T + b.float()
    y = {None: lambda t: t, "relu": torch.relu, "silu": F.silu, "gelu_erf": F.gelu,
         "gelu_tanh": lambda t: F.gelu(t, approximate="tanh")}[act](y)
    if res:
        y = y + r.float()
    check(got[..., 16:16 + N], y, tol=2e-3, what=f"skinny gemm {B}x{M}x{N}x{K} {act} res={res}")
    assert bool((got[..., :16] == 3.0).all()) and bool((got[..., 16 + N:] == 3.0).all())
    assert torch.equal(run(), got)                                        # deterministic: fixed order of the partial sums
    monkeypatch.setenv("BYA_GEMM_SKINNY", "0")
    tiled = run()
    check(tiled[..., 16:16 + N], y, tol=2e-3, what="the tiled kernel on the same launch")
    assert rel_fro(got[..., 16:16 + N].float(), tiled[..., 16:16 + N].float()) < 4e-3
    monkeypatch.delenv("BYA_GEMM_SKINNY")
    # outside ops.weight_streaming() the tiled kernel runs whatever the row count: bya_gemm_bf16 never picks by rows
    o_w.fill_(3.0)
    ops.gemm(a, w, out, bias=b, act=act, res=r)
    torch.cuda.synchronize()
    assert torch.equal(o_w, tiled)


def test_gemm_mfma_layout_asymmetric(ops, dev):
    """A = I against an asymmetric integer-valued W catches swapped row/col maps exactly."""
    K = N = 128
    a = torch.eye(K, dtype=torch.bfloat16, device=dev)
    w = (torch.arange(N * K, device=dev).reshape(N, K) % 251 - 125).to(torch.bfloat16)
    out = torch.empty(K, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(a, w, out)
    assert torch.equal(out.float(), w.float().T)


@pytest.mark.parametrize("act", ["gelu_tanh", "gelu_erf", "relu", "silu", "leaky_relu"])
def test_gemm_bias_act(ops, dev, act):
    M, N, K = 200, 256, 256
    a, w, b = rnd((M, K), dev, 3), rnd((N, K), dev, 4, K ** -0.5), rnd((N,), dev, 5, 0.5)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(a, w, out, bias=b, act=act)
    pre = a.float() @ w.float().T + b.float()
    ref = {"gelu_tanh": lambda x: F.gelu(x, approximate="tanh"), "gelu_erf": F.gelu, "relu": F.relu,
           "silu": F.silu, "leaky_relu": F.leaky_relu}[act](pre)
    check(out, ref, what=f"gemm+{act}")


def test_gemm_gate_residual_batched(ops, dev):
    """The attn1.to_out / ff.net.2 epilogue: x + gate[row type] * (a @ W^T + b), text rows use gate0."""
    B, S, T, N, K = 2, 300, 26, 256, 128
    a = rnd((B, S, K), dev, 6)
    w, b = rnd((N, K), dev, 7, K ** -0.5), rnd((N,), dev, 8, 0.1)
    x = rnd((B, S, N), dev, 9)
    mods = rnd((B, 6, N), dev, 10, 0.5)
    out = x.clone()
    ops.gemm(a, w, out, bias=b, res=out, gate0=mods[:, 5], gate1=mods[:, 2], gate_split=T,
             gate_batch_stride=mods.stride(0))
    y = a.float() @ w.float().T + b.float()
    gate = torch.cat([mods[:, 5:6].float().expand(-1, T, -1), mods[:, 2:3].float().expand(-1, S - T, -1)], 1)
    check(out, x.float() + gate * y, what="gemm gate+res")


def test_gemm_strided_views(ops, dev):
    """Row-strided A (video rows of the joint buffer) and strided output (writing into a wider buffer)."""
    S, T, D, N = 200, 26, 128, 64
    xbuf = rnd((2, S, D), dev, 11)
    w = rnd((N, D), dev, 12, D ** -0.5)
    big = torch.zeros(2, S - T, 3 * N, dtype=torch.bfloat16, device=dev)
    ops.gemm(xbuf[:, T:], w, big[:, :, N:2 * N])
    check(big[:, :, N:2 * N], xbuf[:, T:].float() @ w.float().T, what="gemm strided")
    assert big[:, :, :N].abs().max() == 0 and big[:, :, 2 * N:].abs().max() == 0


def test_gemm_split_output(ops, dev):
    """Packed q|k|v projection: one launch, three output tensors."""
    B, S, D = 2, 300, 256
    a = rnd((B, S, D), dev, 13)
    w, b = rnd((3 * D, D), dev, 14, D ** -0.5), rnd((3 * D,), dev, 15, 0.2)
    qkv = torch.zeros(3, B, S, D, dtype=torch.bfloat16, device=dev)
    ops.gemm(a, w, qkv[0], bias=b, split=(D, B * S * D))
    ref = a.float() @ w.float().T + b.float()
    for i in range(3):
        check(qkv[i], ref[..., i * D:(i + 1) * D], what=f"gemm split {i}")


@pytest.mark.parametrize("M,N,K", [(1024, 1024, 64), (2500, 3072, 192), (4100, 768, 3072), (17776, 512, 512)])
def test_gemm_pipelined_256(ops, dev, M, N, K):
    """Shapes that select the pipelined 256x256 kernel (ragged M and N tiles, K = 1 and 3 tiles, long K)."""
    a, w, b = rnd((M, K), dev, 16), rnd((N, K), dev, 17, K ** -0.5), rnd((N,), dev, 18)
    res = rnd((M, N), dev, 19)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(a, w, out, bias=b, res=res, act="gelu_tanh")
    check(out, res.float() + F.gelu(a.float() @ w.float().T + b.float(), approximate="tanh"), what=f"gemm256 {M}x{N}x{K}")


@pytest.mark.parametrize("B,M,N,K,epi", [(1, 2222, 3072, 3072, "gate+res"), (1, 2193, 2048, 2048, "bias"), (2, 700, 520, 256, "gelu+res"),
                                         (1, 129, 264, 832, "plain"), (1, 2222, 768, 1024, "split"), (1, 130, 256, 192, "short-k")])
def test_gemm_128_row_persistent_tile_equals_the_256_row_one(ops, dev, B, M, N, K, epi):
    """The 128 x 256 persistent kernels (three-stage ring that runs across output tiles, one barrier per K-tile; gemm_v6.hip: four
    compute waves + four LOADER waves that issue the LDS-DMA stream, every plain Linear; gemm_v5.hip: four waves that do both, the
    q|k|v projection's norm epilogue) are the 256 x 256 one's twins for row counts that leave its grid half empty -- a rank's 2222
    rows of the 8-way sharded step.  Same K order, same epilogue code (gemm_wide_epilogue.h), so FORCED on a shape (option
    gemm_tile = 6 / 5) they must equal the forced 256-row kernel (gemm_tile = 4) BIT FOR BIT and sit within the fp32
    reference's tolerance: ragged M and N tiles (129 rows = a
    second, one-row tile; 264 = a column tile of eight columns), a batch, gates + residual in place, GELU, split outputs, several
    tiles per workgroup (2222 x 3072 is 216 tiles; 700 x 520 x 2 is 36), and K of three K-tiles (below the ring's minimum of four:
    the 256-row path must take it).  The library's own choice picks the 128-row tile for the first two shapes."""
    a = rnd((B, M, K), dev, 41)
    w, b = rnd((N, K), dev, 42, K ** -0.5), rnd((N,), dev, 43, 0.3)
    x = rnd((B, M, N), dev, 44)
    mods = rnd((B, 2, N), dev, 45, 0.5)
    outs = {}
    for tile in (4, 5, 6, -1):
        with ops.options(gemm_tile=tile, gemm_splitk=0):
            if epi == "gate+res":
                o = x.clone()
                ops.gemm(a, w, o, bias=b, res=o, gate0=mods[:, 0], gate1=mods[:, 1], gate_split=226, gate_batch_stride=mods.stride(0))
            elif epi == "gelu+res":
                o = torch.empty_like(x)
                ops.gemm(a, w, o, bias=b, res=x, act="gelu_tanh")
            elif epi == "split":
                o = torch.zeros(3, B, M, N // 3, dtype=torch.bfloat16, device=dev)
                ops.gemm(a, w, o[0], bias=b, split=(N // 3, B * M * (N // 3)))
            else:
                o = torch.empty_like(x)
                ops.gemm(a, w, o, bias=b if epi != "plain" else None)
        outs[tile] = o
    torch.cuda.synchronize()
    assert torch.equal(outs[5], outs[4]), int((outs[5] != outs[4]).sum())
    assert torch.equal(outs[6], outs[4]), int((outs[6] != outs[4]).sum())
    assert torch.equal(outs[-1], outs[4])
    y = a.float() @ w.float().T + (b.float() if epi != "plain" else 0.0)
    if epi == "gate+res":
        gate = torch.cat([mods[:, 0:1].float().expand(-1, 226, -1), mods[:, 1:2].float().expand(-1, M - 226, -1)], 1)
        ref = x.float() + gate * y
    elif epi == "gelu+res":
        ref = x.float() + F.gelu(y, approximate="tanh")
    elif epi == "split":
        ref = torch.stack([y[..., i * (N // 3):(i + 1) * (N // 3)] for i in range(3)])
    else:
        ref = y
    check(outs[6], ref, what=f"gemm 128-row tile {M}x{N}x{K} {epi}")


@pytest.mark.parametrize("gate_split", [226, 16500])
def test_gemm_quantisation_tail_split(ops, dev, gate_split):
    """17776 x 3072 is 3.28 rounds of 256x256 tiles: rows [0, 16384) run on the 256-row persistent kernel and the last 1392
    rows as a second launch (round 6: on the 128 x 256 kernel with loader waves; before: the 128x128 kernel).  Every
    row-indexed operand (residual, gate switch row, per-row bias scale) must land on the right rows in both parts."""
    M, N, K = 17776, 3072, 1024
    a, w, b = rnd((M, K), dev, 30), rnd((N, K), dev, 31, K ** -0.5), rnd((N,), dev, 32, 0.3)
    x = rnd((M, N), dev, 33)
    g = rnd((2, N), dev, 34, 0.5)
    rs = torch.rand(M, generator=torch.Generator().manual_seed(35)).to(dev)
    out = torch.empty_like(x)
    ops.gemm(a, w, out, bias=b, res=x, gate0=g[0], gate1=g[1], gate_split=gate_split, bias_rowscale=rs)
    y = a.float() @ w.float().T + b.float() * rs[:, None]
    gate = torch.where(torch.arange(M, device=dev)[:, None] < gate_split, g[0].float(), g[1].float())
    ref = x.float() + gate * y
    for lo, hi in ((0, 16384), (16384, M)):
        check(out[lo:hi], ref[lo:hi], what=f"gemm tail split rows {lo}:{hi} gate_split={gate_split}")


@pytest.mark.parametrize("M,N,ln,res,act,nsplit", [
    (300, 512, True, False, None, 0), (35100, 1536, True, False, None, 0), (35100, 512, False, True, None, 0),
    (4133, 512, True, False, "gelu_erf", 2), (1000, 512, False, True, None, 1), (129, 1536, True, False, None, 4),
    (2700, 512, True, True, "gelu_erf", 4)])
def test_rowgemm512(ops, dev, M, N, ln, res, act, nsplit):
    """Row-stationary K = 512 GEMM with LayerNorm folded into weights + matrix-core row statistics, GELU(erf), in-place
    residual; vs fp32 LayerNorm -> Linear.  Rows carry a common offset (|mean| up to ~20 std) to exercise the
    E[x^2] - mean^2 form of the variance (fp32: relative error ~1e-7 * (1 + mean^2/var), harmless for a residual stream
    but not meant for rows that are constant up to bf16 noise)."""
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, 512, generator=g) * (0.3 + 2.7 * torch.rand(M, 1, generator=g)) + torch.randn(M, 1, generator=g) * 2
    x = bf(x).to(dev)
    w, b = rnd((N, 512), dev, 70, 512 ** -0.5), rnd((N,), dev, 71, 0.2)
    gam, bet = bf(1 + 0.3 * torch.randn(512, generator=g)).to(dev), bf(0.2 * torch.randn(512, generator=g)).to(dev)
    pack = ops.pack_rowgemm512(w, b, gam if ln else None, bet if ln else None)
    r = rnd((M, N), dev, 72) if res else None
    out = r.clone() if res else torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.rowgemm512(x, pack, out, res=out if res else None, act=act, nsplit=nsplit)
    def chain(rounded):
        """fp32 truth, or the reference's own bf16 op chain (every op's output rounded to bf16)."""
        r_ = (lambda v: bf(v).float()) if rounded else (lambda v: v)
        h = r_(F.layer_norm(x.float(), (512,), gam.float(), bet.float(), 1e-5)) if ln else x.float()
        y = r_(h @ w.float().T + b.float())
        if act:
            y = r_(F.gelu(y))
        if res:
            y = r_(y + r.float())
        return y
    truth, ref_bf16 = chain(False), chain(True)
    e_gpu, e_ref = rel_fro(out.float(), truth), rel_fro(ref_bf16, truth)
    print(f"rowgemm512 M={M} N={N} ln={ln} res={res} act={act}: vs fp32 truth {e_gpu:.3e}; reference bf16 chain {e_ref:.3e}")
    # LayerNorm is folded (W*gamma rounded once; LN(x) itself is never rounded), so the rounding points differ from
    # the reference chain: the bar is "no further from the exact result than the reference's own bf16 chain"
    assert e_gpu <= 1.25 * e_ref + 2e-4
    assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize("N,ln,res", [(1536, True, False), (512, False, True), (512, True, False)])
def test_rowgemm512_repeatable_at_router_shape(ops, dev, N, ln, res):
    """The W-chunk ring is synchronised with COUNTED waits (the previous chunk's output stores stay in flight across the
    barrier): a wait that is one short would read a chunk before it has landed -- rare wrong tiles that depend on memory
    load.  35100 rows x the router's widths, 25 launches on a busy GPU (a large GEMM queued on a second stream), every
    result bit-identical to the first and to a launch on an idle GPU."""
    M = 35100
    x = rnd((M, 512), dev, 80)
    w, b = rnd((N, 512), dev, 81, 512 ** -0.5), rnd((N,), dev, 82, 0.2)
    gam, bet = rnd((512,), dev, 83, 0.3) + 1, rnd((512,), dev, 84, 0.2)
    pack = ops.pack_rowgemm512(w, b, gam if ln else None, bet if ln else None)
    r = rnd((M, N), dev, 85) if res else None
    def run():
        out = r.clone() if res else torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ops.rowgemm512(x, pack, out, res=out if res else None)
        return out
    first = run()
    torch.cuda.synchronize()
    a, bb = rnd((8192, 8192), dev, 86), rnd((8192, 8192), dev, 87)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(6):
            a @ bb
    outs = [run() for _ in range(25)]
    torch.cuda.synchronize()
    assert all(torch.equal(first, o) for o in outs)


@pytest.mark.parametrize("M", [2048, 2049, 2063, 4394, 8788, 17550, 35100, 52650])
@pytest.mark.parametrize("res,act", [(False, None), (True, None), (True, "gelu_erf"), (False, "gelu_erf")])
def test_rowgemm512_w_stationary_form_matches_chunk_balanced(ops, dev, monkeypatch, M, res, act):
    """N = 512 without LayerNorm takes the W-stationary kernel (one W quarter per workgroup, rows streamed) for
    2048 <= M <= 65536; the reference form "rowgemm_chunked" (ops.options) keeps the chunk-balanced kernel.  Same MFMA order over K per output element, so
    the two agree BIT FOR BIT -- ragged tile counts (waves with 0, 1, odd numbers of 16-row tiles, a last tile of 1 or 15
    rows), strided X, residual added in place."""
    xw = rnd((M, 768), dev, 90 + M % 7)
    x = xw[:, 128:640]                                       # row stride 768: the loader must use ldx, not 512
    w, b = rnd((512, 512), dev, 91, 512 ** -0.5), rnd((512,), dev, 92, 0.2)
    pack = ops.pack_rowgemm512(w, b, None, None)
    r = rnd((M, 512), dev, 93) if res else None
    def run(flag):
        out = r.clone() if res else torch.full((M, 512), 7.0, dtype=torch.bfloat16, device=dev)
        with ops.options(reference_forms=[] if flag else "rowgemm_chunked"):
            ops.rowgemm512(x, pack, out, res=out if res else None, act=act)
        torch.cuda.synchronize()
        return out
    assert torch.equal(run(True), run(False))


def _group_attn_inputs(dev, M, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, 512, generator=g) * (0.5 + 1.5 * torch.rand(M, 1, generator=g)) + torch.randn(M, 1, generator=g)
    x = bf(x).to(dev)
    w, b = rnd((1536, 512), dev, seed + 1, 3.0 * 512 ** -0.5), rnd((1536,), dev, seed + 2, 0.2)      # scores of a few units
    gam, bet = bf(1 + 0.3 * torch.randn(512, generator=g)).to(dev), bf(0.2 * torch.randn(512, generator=g)).to(dev)
    return x, w, b, gam, bet


def _group_rows(L, n_outer, n_inner, outer_stride, seq_stride, dev):
    o = torch.arange(n_outer, device=dev)[:, None, None] * outer_stride
    i = torch.arange(n_inner, device=dev)[None, :, None]
    e = torch.arange(L, device=dev)[None, None, :] * seq_stride
    return (o + i + e).reshape(-1, L)                                    # [groups, L] row indices


@pytest.mark.parametrize("name,L,n_outer,n_inner,outer_stride,seq_stride,M", [
    ("temporal 13 frames x 2 ids", 13, 2, 1350, 17550, 1350, 35100),            # one group + 3 unused slots per tile
    ("multi-ID 2 ids", 2, 1, 17550, 35100, 17550, 35100),                        # 8 groups per tile
    ("multi-ID 3 ids", 3, 1, 1111, 3333, 1111, 3333),                            # 4 groups in 4-slot cells, ragged tail
    ("temporal, one rank's location range", 13, 2, 169, 13 * 169, 169, 2 * 13 * 169),
    ("temporal 16 frames, CFG batch", 16, 4, 77, 16 * 77, 77, 4 * 16 * 77 + 5),  # full tiles, rows beyond the groups untouched
    ("single rows", 1, 1, 300, 0, 0, 300),                                        # softmax over one key = v itself
    ("temporal 25 frames x 3 ids (97-frame clip)", 25, 3, 1350, 25 * 1350, 1350, 3 * 25 * 1350),   # a group = both tiles of a wave
    ("17 rows, ragged", 17, 2, 101, 17 * 101, 101, 2 * 17 * 101 + 3),
    ("32 rows", 32, 1, 77, 0, 77, 32 * 77),
])
def test_router_group_attn_fused_vs_fp32_and_unfused(ops, dev, name, L, n_outer, n_inner, outer_stride, seq_stride, M):
    """bya_router_group_attn (LayerNorm -> q|k|v -> attention over groups of L gathered rows, q|k|v never written) against
    (a) the fp32 chain LayerNorm -> Linear -> softmax(q k^T / 8) v, with the usual bar "no further from it than the
    reference's own bf16 op chain", and (b) the unfused pair bya_rowgemm512 + bya_attn_tiny on the same packed weights:
    the same bf16 q, k, v, only P is rounded to bf16 before P.V -- a few 1e-3.  Rows outside every group are not written."""
    x, w, b, gam, bet = _group_attn_inputs(dev, M, 90 + L)
    pack = ops.pack_rowgemm512(w, b, gam, bet)
    out = torch.full((M, 512), 7.0, dtype=torch.bfloat16, device=dev)
    ops.router_group_attn(x, pack, out, L, n_outer, n_inner, outer_stride, seq_stride)
    rows = _group_rows(L, n_outer, n_inner, outer_stride, seq_stride, dev)
    def chain(rounded):
        r_ = (lambda v: bf(v).float()) if rounded else (lambda v: v)
        h = r_(F.layer_norm(x.float(), (512,), gam.float(), bet.float(), 1e-5))
        qkv = r_(h @ w.float().T + b.float())
        q, k, v = (qkv[:, i * 512:(i + 1) * 512][rows].view(-1, L, 8, 64).transpose(1, 2) for i in range(3))
        o = r_(sdpa_ref(q, k, v, 0.125))                                  # [groups, 8, L, 64]
        return o.transpose(1, 2).reshape(-1, L, 512)
    truth, ref16 = chain(False), chain(True)
    got = out[rows].float()
    e, e16 = rel_fro(got, truth), rel_fro(ref16, truth)
    qkv = torch.empty(M, 1536, dtype=torch.bfloat16, device=dev)
    ops.rowgemm512(x, pack, qkv)
    un = torch.full((M, 512), 7.0, dtype=torch.bfloat16, device=dev)
    ops.attn_tiny(qkv, qkv[:, 512:], qkv[:, 1024:], un, L, 8, n_outer, n_inner, outer_stride, seq_stride, 1536, 512, 0.125)
    d = rel_fro(got, un[rows].float())
    print(f"{name}: fused vs fp32 {e:.3e} (reference bf16 chain {e16:.3e}); fused vs unfused pair {d:.3e}")
    assert torch.isfinite(out.float()).all()
    assert e <= 1.25 * e16 + 1e-3 and d <= 4e-3
    with pytest.raises(Exception):
        ops.router_group_attn(x, pack, out, 33, 1, 1, 0, 1)            # longer than two tiles: refused, not wrong
    touched = torch.zeros(M, dtype=torch.bool, device=dev)
    touched[rows.reshape(-1)] = True
    assert bool((out[~touched] == 7.0).all()), "rows outside the groups were written"
    assert not bool((out[touched] == 7.0).all(dim=1).any()), "a group row was not written"


def test_router_group_attn_repeatable_and_masks_exact(ops, dev):
    """(1) 20 launches at the router's full size on a busy GPU are bit-identical (the chunk ring's counted waits now differ
    per chunk kind: q after 4 stores, k and v after none).  (2) Groups do not leak into each other: changing the rows of ONE
    group changes that group's outputs only, bit for bit -- the in-tile mask is exact, not approximately zero."""
    L, n_outer, n_inner, M = 13, 2, 1350, 35100
    x, w, b, gam, bet = _group_attn_inputs(dev, M, 77)
    pack = ops.pack_rowgemm512(w, b, gam, bet)
    run = lambda xx, LL=L, no=n_outer, ni=n_inner, os_=17550, ss=1350: ops.router_group_attn(
        xx, pack, torch.zeros(M, 512, dtype=torch.bfloat16, device=dev), LL, no, ni, os_, ss)
    first = run(x)
    torch.cuda.synchronize()
    a, bb = rnd((8192, 8192), dev, 86), rnd((8192, 8192), dev, 87)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(6):
            a @ bb
    outs = [run(x) for _ in range(20)]
    torch.cuda.synchronize()
    assert all(torch.equal(first, o) for o in outs)
    # multi-ID groups share tiles (8 per tile): perturb token 1234 of both identities
    base = run(x, 2, 1, 17550, 35100, 17550)
    x2 = x.clone()
    x2[[1234, 17550 + 1234]] = rnd((2, 512), dev, 5)
    pert = run(x2, 2, 1, 17550, 35100, 17550)
    same = (base == pert).all(dim=1)
    assert bool(same[:1234].all()) and bool(same[1235:17550 + 1234].all()) and bool(same[17550 + 1235:].all())
    assert not bool(same[1234]) and not bool(same[17550 + 1234])
    # (3) where a group sits inside a 16-row tile does not change its result: three identities, the whole token range against
    # the same tokens presented as two shards that start at different offsets modulo the groups-per-tile count
    N3 = 1111
    x3 = x[:3 * N3].contiguous()
    whole = ops.router_group_attn(x3, pack, torch.zeros(3 * N3, 512, dtype=torch.bfloat16, device=dev), 3, 1, N3, 3 * N3, N3)
    for lo, hi in ((0, 557), (557, N3)):
        n = hi - lo
        shard = torch.cat([x3[i * N3 + lo:i * N3 + hi] for i in range(3)]).contiguous()
        part = ops.router_group_attn(shard, pack, torch.zeros(3 * n, 512, dtype=torch.bfloat16, device=dev), 3, 1, n, 3 * n, n)
        want = torch.cat([whole[i * N3 + lo:i * N3 + hi] for i in range(3)])
        assert torch.equal(part, want), (lo, hi)


@pytest.mark.parametrize("M,tp0", [(35100, 0), (35100, 5), (4388, 0), (2049, 3), (300, 0), (15, 1), (70200, 0)])
def test_router_mlp_chain_is_bit_identical_to_its_two_launches(ops, dev, M, tp0):
    """bya_router_mlp_fused (csrc/rowchain.hip: LayerNorm -> mlp[0] -> GELU -> mlp[2] -> + x in one launch, the hidden
    activation in registers) against the two bya_rowgemm512 launches it replaces: same MFMA order over K, same epilogue
    expressions on the same rounded values => torch.equal, for one pass, a pass + remainder (35100 rows = 2194 tiles on 256
    workgroups), three passes (70200), ragged last tiles, every first-pass size.  Strided rows, in place.  And against the
    fp32 chain at the row GEMM's bar ("no further from it than the reference's own bf16 op chain")."""
    g = torch.Generator().manual_seed(M)
    xw = bf(torch.randn(M, 640, generator=g) * (0.3 + 2.7 * torch.rand(M, 1, generator=g)) + torch.randn(M, 1, generator=g) * 2).to(dev)
    x = xw[:, 64:576]                                                 # row stride 640
    w1, b1 = rnd((512, 512), dev, 70, 512 ** -0.5), rnd((512,), dev, 71, 0.2)
    w2, b2 = rnd((512, 512), dev, 72, 512 ** -0.5), rnd((512,), dev, 73, 0.2)
    gam, bet = bf(1 + 0.3 * torch.randn(512, generator=g)).to(dev), bf(0.2 * torch.randn(512, generator=g)).to(dev)
    p1, p2 = ops.pack_rowgemm512(w1, b1, gam, bet), ops.pack_rowgemm512(w2, b2)
    h = torch.empty(M, 512, dtype=torch.bfloat16, device=dev)
    two = x.clone()
    ops.rowgemm512(x, p1, h, act="gelu_erf")
    ops.rowgemm512(h, p2, two, res=two)
    keep = xw.clone()
    ops.router_mlp_fused(x, p1, p2, tiles_pass0=tp0)                  # in place, through the strided view
    assert torch.equal(x, two)
    assert torch.equal(xw[:, :64], keep[:, :64]) and torch.equal(xw[:, 576:], keep[:, 576:]), "columns outside the view were written"
    x0 = keep[:, 64:576].float()
    def chain(rounded):
        r_ = (lambda v: bf(v).float()) if rounded else (lambda v: v)
        y = r_(F.layer_norm(x0, (512,), gam.float(), bet.float(), 1e-5))
        y = r_(F.gelu(r_(y @ w1.float().T + b1.float())))
        return r_(x0 + r_(y @ w2.float().T + b2.float()))
    truth, ref16 = chain(False), chain(True)
    e, e16 = rel_fro(x.float(), truth), rel_fro(ref16, truth)
    print(f"router MLP chain M={M}: vs fp32 {e:.3e} (reference bf16 chain {e16:.3e})")
    assert e <= 1.25 * e16 + 2e-4 and torch.isfinite(x.float()).all()


@pytest.mark.parametrize("name,L,n_outer,n_inner,outer_stride,seq_stride,M,tp0", [
    ("temporal 13 frames x 2 ids", 13, 2, 1350, 17550, 1350, 35100, 0),        # 2700 tiles: 8 per workgroup + a remainder pass of 3
    ("temporal, first pass of 6", 13, 2, 1350, 17550, 1350, 35100, 6),
    ("multi-ID 2 ids", 2, 1, 17550, 35100, 17550, 35100, 0),                    # 8 groups per tile, 2194 tiles
    ("multi-ID 3 ids", 3, 1, 1111, 3333, 1111, 3333, 0),                        # 4-slot cells, ragged tail
    ("temporal, one rank's location range", 13, 2, 169, 13 * 169, 169, 2 * 13 * 169, 0),
    ("temporal 16 frames, CFG batch", 16, 4, 77, 16 * 77, 77, 4 * 16 * 77 + 5, 2),
    ("single rows", 1, 1, 300, 0, 0, 300, 0),
])
def test_router_attn_chain_is_bit_identical_to_its_two_launches(ops, dev, name, L, n_outer, n_inner, outer_stride, seq_stride, M, tp0):
    """bya_router_group_attn_out (LayerNorm -> q|k|v -> group attention -> to_out -> + x in one launch, the attention output
    in registers) against bya_router_group_attn + bya_rowgemm512(res = x): torch.equal on the rows of the groups; rows
    outside every group keep their values (the pair's out-projection would have touched them: compared on group rows only).
    Groups longer than one tile are refused."""
    x, w, b, gam, bet = _group_attn_inputs(dev, M, 190 + L)
    wo, bo = rnd((512, 512), dev, 191, 512 ** -0.5), rnd((512,), dev, 192, 0.2)
    pack, po = ops.pack_rowgemm512(w, b, gam, bet), ops.pack_rowgemm512(wo, bo)
    rows = _group_rows(L, n_outer, n_inner, outer_stride, seq_stride, dev).reshape(-1)
    a = torch.zeros(M, 512, dtype=torch.bfloat16, device=dev)
    ops.router_group_attn(x, pack, a, L, n_outer, n_inner, outer_stride, seq_stride)
    two = x.clone()
    ops.rowgemm512(a, po, two, res=two)
    one = x.clone()
    ops.router_group_attn_out(one, pack, po, L, n_outer, n_inner, outer_stride, seq_stride, tiles_pass0=tp0)
    assert torch.equal(one[rows], two[rows]), name
    touched = torch.zeros(M, dtype=torch.bool, device=dev)
    touched[rows] = True
    assert torch.equal(one[~touched], x[~touched]), "rows outside the groups were written"
    assert torch.isfinite(one.float()).all()
    with pytest.raises(Exception):
        ops.router_group_attn_out(one, pack, po, 17, 1, 1, 0, 1)


def test_router_chains_repeatable_on_a_busy_gpu(ops, dev):
    """The chains' W ring is synchronised with counted waits that differ by phase (none, then two output stores per chunk)
    and by wave (waves without a tile in the remainder pass issue no stores): 15 launches each at the router's full size on
    a busy GPU, every result bit-identical to the first."""
    M = 35100
    x, w, b, gam, bet = _group_attn_inputs(dev, M, 300)
    wo, bo = rnd((512, 512), dev, 301, 512 ** -0.5), rnd((512,), dev, 302, 0.2)
    w1, b1 = rnd((512, 512), dev, 303, 512 ** -0.5), rnd((512,), dev, 304, 0.2)
    pack, po = ops.pack_rowgemm512(w, b, gam, bet), ops.pack_rowgemm512(wo, bo)
    p1 = ops.pack_rowgemm512(w1, b1, gam, bet)
    runs = {"temporal": lambda: ops.router_group_attn_out(x.clone(), pack, po, 13, 2, 1350, 17550, 1350),
            "multi-ID": lambda: ops.router_group_attn_out(x.clone(), pack, po, 2, 1, 17550, 35100, 17550),
            "mlp": lambda: ops.router_mlp_fused(x.clone(), p1, po)}
    first = {k: f() for k, f in runs.items()}
    torch.cuda.synchronize()
    aa, bb = rnd((8192, 8192), dev, 86), rnd((8192, 8192), dev, 87)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(6):
            aa @ bb
    outs = {k: [f() for _ in range(15)] for k, f in runs.items()}
    torch.cuda.synchronize()
    for k in runs:
        assert all(torch.equal(first[k], o) for o in outs[k]), k


def test_gemm_loader_wave_kernel_repeatable_on_a_busy_gpu(ops, dev):
    """gemm_v6.hip synchronises two kinds of waves through ONE barrier per K-tile: the loaders' `vmcnt` says which pieces of the
    LDS-DMA stream have landed, the compute waves' arrival says which ring stage may be refilled, and the ring runs across a
    workgroup's output tiles.  A protocol slip (a stage refilled before its last reader, a fragment read before its piece
    landed) would show as rare wrong tiles when workgroups drift apart: 15 launches each of a one-round shape (216 tiles), of a
    shape with seven tiles per workgroup and of a long K (192 K-tiles) on a busy GPU (a large GEMM queued on a second stream),
    every result bit-identical to the first and to the 256-row kernel's."""
    shapes = [(2222, 3072, 3072), (4444, 12288, 3072), (2222, 3072, 12288)]
    data, first, want = {}, {}, {}
    for M, N, K in shapes:
        a, w, b = rnd((M, K), dev, 90), rnd((N, K), dev, 91, K ** -0.5), rnd((N,), dev, 92, 0.3)
        data[(M, N, K)] = (a, w, b, rnd((M, N), dev, 93))
    def run(key, tile):
        a, w, b, x = data[key]
        with ops.options(gemm_tile=tile, gemm_splitk=0):
            return ops.gemm(a, w, torch.empty_like(x), bias=b, res=x)
    for key in data:
        first[key], want[key] = run(key, 6), run(key, 4)
    torch.cuda.synchronize()
    aa, bb = rnd((8192, 8192), dev, 86), rnd((8192, 8192), dev, 87)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(6):
            aa @ bb
    outs = {key: [run(key, 6) for _ in range(15)] for key in data}
    torch.cuda.synchronize()
    for key in data:
        assert torch.equal(first[key], want[key]), key
        assert all(torch.equal(first[key], o) for o in outs[key]), key


@pytest.mark.parametrize("tile", [4, 6])
def test_gemm_big_tile_kernels_whole_suite(ops, dev, tile):
    """The GEMM parity tests again with a persistent kernel FORCED for all their shapes (option gemm_tile = 4: the 256 x 256
    one-wave-per-SIMD kernel of gemm_v4.hip; 6: the 128 x 256 kernel with loader waves of gemm_v6.hip): ragged M / N, K of one, two
    and three K-tiles (shorter than a kernel's software pipeline: it must hand the launch on -- gemm_v4 needs three K-tiles,
    gemm_v6 four), batches, strided views, split outputs, gate + residual epilogues, every activation (the ones without a
    big-tile epilogue fall back to the 128 x 128 kernel).  In-process since round 6 (a child process with the variables set cost
    20 s of start-up; the K-split tests force their own options).  (The "w8" arm -- the 8-wave fallback kernel forced on every
    shape -- went with the suite's time budget in round 6; the fallback still runs where the default picks it.)"""
    with ops.options(gemm_tile=tile):
        for M, N, K in [(128, 128, 64), (256, 384, 192), (300, 512, 512), (17, 64, 128), (1350, 3072, 768), (2222, 2048, 3072),
                        (130, 12288, 3072)]:
            test_gemm_plain(ops, dev, M, N, K)
        test_gemm_mfma_layout_asymmetric(ops, dev)
        for act in ("gelu_tanh", "gelu_erf", "relu", "silu", "leaky_relu"):
            test_gemm_bias_act(ops, dev, act)
        test_gemm_gate_residual_batched(ops, dev)
        test_gemm_strided_views(ops, dev)
        test_gemm_split_output(ops, dev)
        for M, N, K in [(1024, 1024, 64), (2500, 3072, 192), (4100, 768, 3072), (17776, 512, 512)]:
            test_gemm_pipelined_256(ops, dev, M, N, K)
        for gate_split in (226, 16500):
            test_gemm_quantisation_tail_split(ops, dev, gate_split)


def test_gemm_rejects_bad_shapes(ops, dev):
    from bind_your_avatar_implementation_amd._hip import ByaError
    a, w = rnd((64, 96), dev, 1), rnd((64, 96), dev, 2)       # K % 64 != 0
    with pytest.raises(ByaError):
        ops.gemm(a, w, torch.empty(64, 64, dtype=torch.bfloat16, device=dev))


# ----------------------------------------------------------------------------------------------- attention
def sdpa_ref(q, k, v, scale):
    s = torch.einsum("bhid,bhjd->bhij", q.float(), k.float()) * scale
    return torch.einsum("bhij,bhjd->bhid", torch.softmax(s, -1), v.float())


@pytest.mark.parametrize("S,H", [(64, 8), (128, 8), (200, 8), (1350, 8), (2226, 16)])
def test_attn_self_d64(ops, dev, S, H):
    B, D = 2, 64
    q, k, v = (rnd((B, S, H * D), dev, 20 + i) for i in range(3))
    out = torch.empty_like(q)
    ops.self_attention(q, k, v, out, heads=H)
    sp = lambda t: t.view(B, S, H, D).transpose(1, 2)
    ref = sdpa_ref(sp(q), sp(k), sp(v), D ** -0.5).transpose(1, 2).reshape(B, S, H * D)
    check(out, ref, tol=ATTN_TOL, what=f"attn S={S} H={H}")


def test_attn_prescaled_scores(ops, dev):
    """Engine path of the joint attention: scale*log2(e) is folded into k by the q/k-norm kernel (k_scale), the
    attention kernel then takes the scores as exp2 exponents.  Reference: softmax(ln2 * q.k') with the same k'."""
    B, S, H, D = 1, 1000, 8, 64
    q, k, v = (rnd((B, S, H * D), dev, 25 + i) for i in range(3))
    c = D ** -0.5 * 1.4426950408889634
    k2 = bf(k.float() * c)
    out = torch.empty_like(q)
    ops.self_attention(q, k2, v, out, heads=H, prescaled=True)
    sp = lambda t: t.view(B, S, H, D).transpose(1, 2)
    ref = sdpa_ref(sp(q), sp(k2), sp(v), math.log(2.0)).transpose(1, 2).reshape(B, S, H * D)
    check(out, ref, tol=ATTN_TOL, what="attn prescaled")


def test_attn_layout_exact(ops, dev):
    """One-hot softmax (a huge matching key) makes the output an exact row gather of V: pins every
    MFMA / transposed-LDS-read lane map with integer data."""
    B, S, H, D = 1, 192, 8, 64
    g = torch.Generator().manual_seed(3)
    perm = torch.randperm(S, generator=g)
    basis = torch.zeros(S, D)
    # distinct +-1 codes per key so q_i . k_j is maximal only for j = perm[i]
    code = torch.sign(torch.randn(S, D, generator=g))
    k = bf(code * 4).to(dev)
    q = bf(code[perm] * 4).to(dev)
    v = bf((torch.arange(S * D).reshape(S, D) % 127 - 63).float()).to(dev)
    qq = q[None, :, None, :].expand(B, S, H, D).reshape(B, S, H * D).contiguous()
    kk = k[None, :, None, :].expand(B, S, H, D).reshape(B, S, H * D).contiguous()
    vv = (v[None, :, None, :] + torch.arange(H, device=dev)[None, None, :, None]).to(torch.bfloat16)
    vv = vv.reshape(B, S, H * D).contiguous()
    out = torch.empty_like(qq)
    ops.self_attention(qq, kk, vv, out, heads=H, scale=1.0)
    ref = vv.view(B, S, H, D)[:, perm.to(dev)].reshape(B, S, H * D)
    assert torch.equal(out, ref)


def test_attn_online_softmax_rescale(ops, dev):
    """A key spike late in the sequence forces the running max to jump (rescale branch), rule 26."""
    B, S, H, D = 1, 512, 8, 64
    q, k, v = (rnd((B, S, H * D), dev, 30 + i) for i in range(3))
    k.view(B, S, H, D)[0, 450] = q.view(B, S, H, D)[0, 7] * 3
    out = torch.empty_like(q)
    ops.self_attention(q, k, v, out, heads=H)
    sp = lambda t: t.view(B, S, H, D).transpose(1, 2)
    ref = sdpa_ref(sp(q), sp(k), sp(v), D ** -0.5).transpose(1, 2).reshape(B, S, H * D)
    check(out, ref, tol=ATTN_TOL, what="attn spike")


@pytest.mark.parametrize("D,H", [(128, 16), (64, 48)])
def test_attn_cross_kv32_shared_q(ops, dev, D, H):
    """Perceiver / audio pattern: q shared by both ids (level-2 stride 0), 32 keys per (sample, id)."""
    B, NID, N, KV = 2, 2, 300, 32
    q = rnd((B, N, H * D), dev, 40)
    kv = rnd((B, NID, KV, 2 * H * D), dev, 41)
    out = torch.empty(B, NID, N, H * D, dtype=torch.bfloat16, device=dev)
    ops.attention(q, kv, kv[..., H * D:], out, head_dim=D, heads=H, nb1=B, nb2=NID, Sq=N, Skv=KV,
                  q_strides=(N * H * D, 0, H * D), k_strides=(NID * KV * 2 * H * D, KV * 2 * H * D, 2 * H * D),
                  v_strides=(NID * KV * 2 * H * D, KV * 2 * H * D, 2 * H * D),
                  o_strides=(NID * N * H * D, N * H * D, H * D), scale=D ** -0.5)
    qh = q.view(B, 1, N, H, D).expand(B, NID, N, H, D).permute(0, 1, 3, 2, 4).reshape(B * NID, H, N, D)
    kh = kv[..., :H * D].reshape(B * NID, KV, H, D).transpose(1, 2)
    vh = kv[..., H * D:].reshape(B * NID, KV, H, D).transpose(1, 2)
    ref = sdpa_ref(qh, kh, vh, D ** -0.5).transpose(1, 2).reshape(B, NID, N, H * D)
    check(out, ref, tol=ATTN_TOL, what=f"cross-attn D={D}")


@pytest.mark.parametrize("L,n_outer,n_inner", [(13, 2, 90), (2, 1, 500)])
def test_attn_tiny(ops, dev, L, n_outer, n_inner):
    """Router temporal (L=frames, stride = tokens per frame) and multi-ID (L=ids, stride = N) attention."""
    H, D = 8, 64
    if L == 13:
        rows, seq_stride, outer_stride = n_outer * L * n_inner, n_inner, L * n_inner
    else:
        rows, seq_stride, outer_stride = L * n_inner, n_inner, 0
    qkv = rnd((rows, 3 * H * D), dev, 50)
    out = torch.zeros(rows, H * D, dtype=torch.bfloat16, device=dev)
    ops.attn_tiny(qkv, qkv[:, H * D:], qkv[:, 2 * H * D:], out, L, H, n_outer, n_inner, outer_stride, seq_stride,
                  3 * H * D, H * D, D ** -0.5)
    idx = (torch.arange(n_outer)[:, None, None] * outer_stride + torch.arange(L)[None, None, :] * seq_stride
           + torch.arange(n_inner)[None, :, None]).reshape(-1, L).to(dev)          # [groups, L] row ids
    g = qkv[idx].float().view(-1, L, 3, H, D)
    ref = sdpa_ref(g[:, :, 0].transpose(1, 2), g[:, :, 1].transpose(1, 2), g[:, :, 2].transpose(1, 2), D ** -0.5)
    ref_rows = torch.zeros(rows, H * D, device=dev)
    ref_rows[idx.reshape(-1)] = ref.transpose(1, 2).reshape(-1, H * D)
    check(out, ref_rows, what=f"attn_tiny L={L}")


# ----------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("D", [512, 768, 1024, 2048, 3072])
def test_layernorm_affine(ops, dev, D):
    x = rnd((333, D), dev, 60, 2.0) + 0.5
    w, b = rnd((D,), dev, 61, 0.2) + 1, rnd((D,), dev, 62, 0.2)
    out = torch.empty_like(x)
    ops.layernorm(x, out, w, b, eps=1e-5)
    check(out, F.layer_norm(x.float(), (D,), w.float(), b.float(), 1e-5), what=f"LN {D}")
    ops.layernorm(x, out, None, None, eps=1e-5)
    check(out, F.layer_norm(x.float(), (D,), None, None, 1e-5), what=f"LN {D} no affine")


def test_layernorm_adaln_zero(ops, dev):
    """CogVideoXLayerNormZero: text rows and video rows take their own (shift, scale) chunk."""
    B, S, T, D = 2, 140, 26, 3072
    x = rnd((B, S, D), dev, 63)
    w, b = rnd((D,), dev, 64, 0.2) + 1, rnd((D,), dev, 65, 0.2)
    mods = rnd((B, 6 * D), dev, 66, 0.3)     # shift, scale, gate, enc_shift, enc_scale, enc_gate
    out = torch.empty_like(x)
    ops.layernorm(x, out, w, b, eps=1e-5, shift0=mods[:, 3 * D:], scale0=mods[:, 4 * D:], shift1=mods[:, 0:],
                  scale1=mods[:, D:], split=T, mod_batch_stride=mods.stride(0))
    ln = F.layer_norm(x.float(), (D,), w.float(), b.float(), 1e-5)
    m = mods.float()
    ref = torch.cat([ln[:, :T] * (1 + m[:, None, 4 * D:5 * D]) + m[:, None, 3 * D:4 * D],
                     ln[:, T:] * (1 + m[:, None, D:2 * D]) + m[:, None, 0:D]], 1)
    check(out, ref, what="AdaLN-zero")


def test_layernorm_adaln_rows_kernel(ops, dev, monkeypatch):
    """The AdaLN LayerNorm of the DiT width keeps A = w (1 + scale), B = b (1 + scale) + shift in registers and walks several
    rows per wave (reference form "ln_generic": one row per wave, parameters re-read per row).  Against fp32 at the usual bar; bit-identical
    to the one-row-per-wave kernel (which evaluates the same fma); and INDEPENDENT of how the rows
    are cut into launches -- a shard of the sequence must round exactly like the whole (ranges that cross the text / video
    split and a batch boundary, strided input and output)."""
    B, S, T, D = 2, 1237, 226, 3072
    xw, ow = rnd((B, S, D + 64), dev, 63), torch.empty(B, S, D + 64, dtype=torch.bfloat16, device=dev)
    x, out = xw[..., 32:32 + D], ow[..., 8:8 + D]
    w, b = rnd((D,), dev, 64, 0.2) + 1, rnd((D,), dev, 65, 0.2)
    mods = rnd((B, 6 * D), dev, 66, 0.3)
    kw = dict(eps=1e-5, shift0=mods[:, 3 * D:], scale0=mods[:, 4 * D:], shift1=mods[:, 0:], scale1=mods[:, D:],
              mod_batch_stride=mods.stride(0))
    ow.fill_(9.0)
    ops.layernorm(x, out, w, b, split=T, **kw)
    rows = out.clone()
    assert bool((ow[..., :8] == 9.0).all()) and bool((ow[..., 8 + D:] == 9.0).all())
    ln = F.layer_norm(x.float(), (D,), w.float(), b.float(), 1e-5)
    m = mods.float()
    ref = torch.cat([ln[:, :T] * (1 + m[:, None, 4 * D:5 * D]) + m[:, None, 3 * D:4 * D],
                     ln[:, T:] * (1 + m[:, None, D:2 * D]) + m[:, None, 0:D]], 1)
    check(rows, ref, what="AdaLN rows kernel")
    with ops.options(reference_forms="ln_generic"):
        ops.layernorm(x, out, w, b, split=T, **kw)
    assert torch.equal(out, rows)                              # both kernels evaluate the same fma
    # without modulation the rows kernel is the plain affine LayerNorm, bit for bit the one-row-per-wave kernel
    a1, a0 = torch.empty(B, S, D, dtype=torch.bfloat16, device=dev), torch.empty(B, S, D, dtype=torch.bfloat16, device=dev)
    ops.layernorm(x, a1, w, b, eps=1e-5)
    with ops.options(reference_forms="ln_generic"):
        ops.layernorm(x, a0, w, b, eps=1e-5)
    assert torch.equal(a1, a0)
    # any cut of the rows into launches gives the same bits
    for lo, hi in ((0, 100), (100, 226), (226, 227), (227, 900), (900, S)):
        part = torch.empty(B, hi - lo, D, dtype=torch.bfloat16, device=dev)
        ops.layernorm(x[:, lo:hi], part, w, b, split=max(0, min(T - lo, hi - lo)), **kw)
        assert torch.equal(part, rows[:, lo:hi]), (lo, hi)


def test_qknorm_rope(ops, dev):
    from bind_your_avatar_implementation_amd.synth import rope_table
    B, T, H = 2, 26, 48
    grid = (3, 4, 5)
    N = grid[0] * grid[1] * grid[2]
    S = T + N
    q, k = rnd((B, S, H * 64), dev, 70), rnd((B, S, H * 64), dev, 71)
    qw, qb, kw, kb = (rnd((64,), dev, 72 + i, 0.3) + (1 if i % 2 == 0 else 0) for i in range(4))
    cos, sin = (t.to(dev) for t in rope_table(grid))
    q0, k0 = q.clone(), k.clone()
    ops.qknorm_rope(q, k, qw, qb, kw, kb, cos, sin, heads=H, text_rows=T, eps=1e-6)

    def ref(x, w, b):
        xh = F.layer_norm(x.float().view(B, S, H, 64), (64,), w.float(), b.float(), 1e-6)
        v = xh[:, T:]
        xr, xi = v.reshape(B, N, H, 32, 2).unbind(-1)
        rot = torch.stack([-xi, xr], -1).flatten(3)
        v = v * cos[None, :, None, :] + rot * sin[None, :, None, :]
        return torch.cat([xh[:, :T], v], 1).reshape(B, S, H * 64)

    check(q, ref(q0, qw, qb), what="qknorm_rope q")
    check(k, ref(k0, kw, kb), what="qknorm_rope k")
    q, k = q0.clone(), k0.clone()
    ops.qknorm_rope(q, k, qw, qb, kw, kb, cos, sin, heads=H, text_rows=T, eps=1e-6, k_scale=0.18)
    check(q, ref(q0, qw, qb), what="qknorm_rope q (k_scale leaves q alone)")
    check(k, 0.18 * ref(k0, kw, kb), what="qknorm_rope k * k_scale (single rounding)")


@pytest.mark.parametrize("B,M,K,width,blocks", [(1, 1000, 512, 384, 1), (2, 826, 3072, 3072, 1), (1, 2222, 3072, 3072, 8),
                                                 (1, 300, 256, 128, 1)])
def test_qkv_projection_with_the_norm_in_its_epilogue_is_bit_identical(ops, dev, B, M, K, width, blocks):
    """bya_gemm_qkv_norm_rope = the packed q|k|v projection whose epilogue applies the per-head q/k LayerNorm(64) + RoPE + k
    pre-scale (round 4: a second launch that re-read and re-wrote q and k).  It normalises the bf16-ROUNDED projection with the
    arithmetic of csrc/qknorm_math.h in the stand-alone kernel's summation order, so it must equal bya_gemm_bf16 followed by
    bya_qknorm_rope BIT FOR BIT: plain q | k | v outputs, a batch of two, ragged row counts, and the sharded step's form
    (`blocks` column blocks per tensor = the heads of `blocks` destination ranks: [3 * blocks, M, width / blocks])."""
    from bind_your_avatar_implementation_amd.synth import rope_table
    text = 226 if M > 400 else 40
    grid = None
    n_video = M - text
    for t in range(1, 40):                                   # a (t, h, w) grid with exactly n_video tokens, or rows of a bigger one
        if n_video % t == 0:
            grid = (t, n_video // t, 1)
    cos, sin = (c.to(dev)[:n_video].contiguous() for c in rope_table(grid))
    x = rnd((B, M, K), dev, 11)
    w = rnd((3 * width, K), dev, 12, K ** -0.5)
    bias = rnd((3 * width,), dev, 13, 0.5)
    qw, qb, kw, kb = (rnd((64,), dev, 72 + i, 0.3) + (1 if i % 2 == 0 else 0) for i in range(4))
    Dl = width // blocks
    heads = width // 64
    def outputs():
        return torch.zeros(3 * blocks, B, M, Dl, dtype=torch.bfloat16, device=dev)
    # (with blocks > 1 the engine runs one sample: B == 1)
    two = outputs()
    ops.gemm(x, w, two[0], bias=bias, split=(Dl, B * M * Dl))
    if blocks == 1:
        ops.qknorm_rope(two[0], two[1], qw, qb, kw, kb, cos, sin, heads=heads, text_rows=text, eps=1e-6, k_scale=0.18)
    else:
        ops.qknorm_rope(two[:blocks, 0], two[blocks:2 * blocks, 0], qw, qb, kw, kb, cos, sin, heads=heads // blocks, text_rows=text,
                        eps=1e-6, k_scale=0.18)
    one = outputs()
    took = ops.gemm_qkv_norm_rope(x, w, one[0], bias, (Dl, B * M * Dl), qw, qb, kw, kb, cos, sin, text, eps=1e-6, k_scale=0.18)
    torch.cuda.synchronize()
    if width % 128 or K < 192:
        assert not took
        return
    assert took
    assert torch.equal(one, two), (float((one.float() - two.float()).abs().max()),
                                   int((one != two).sum()), [int((one[t] != two[t]).sum()) for t in range(3 * blocks)])
    assert float(one[:2 * blocks].float().abs().sum()) > 0
    # ... on either persistent tile: 256 x 256 (gemm_v4.hip) and 128 x 256 (gemm_v5.hip; the library's choice at 2222 rows)
    for tile in (4, 5):
        forced = outputs()
        with ops.options(gemm_tile=tile):
            assert ops.gemm_qkv_norm_rope(x, w, forced[0], bias, (Dl, B * M * Dl), qw, qb, kw, kb, cos, sin, text, eps=1e-6, k_scale=0.18)
        assert torch.equal(forced, two), (tile, int((forced != two).sum()))


def test_qknorm_rope_statistics_bound_every_row(ops, dev):
    """``stats``: the q/k-norm launch records, per (batch, head), the largest squared norm of the rows it WRITES (bf16-rounded,
    after RoPE and k_scale) in `slots` partial tables; their maximum is the data-dependent score bound the joint attention
    reads from device memory.  It must bound every row (it is a maximum of exactly these numbers), be attained, leave the
    kernel's output bits unchanged, and work for q alone / k alone (the sharded step's call forms)."""
    from bind_your_avatar_implementation_amd.synth import rope_table
    B, T, H, grid = 2, 26, 48, (3, 4, 5)
    S = T + grid[0] * grid[1] * grid[2]
    q0, k0 = rnd((B, S, H * 64), dev, 70), rnd((B, S, H * 64), dev, 71)
    qw, qb, kw, kb = (rnd((64,), dev, 72 + i, 0.3) + (1 if i % 2 == 0 else 0) for i in range(4))
    qw = qw * torch.linspace(0.5, 3.0, 64, device=dev).to(torch.bfloat16)           # uneven gains: the case the bound is for
    cos, sin = (t.to(dev) for t in rope_table(grid))
    q, k = q0.clone(), k0.clone()
    ops.qknorm_rope(q, k, qw, qb, kw, kb, cos, sin, heads=H, text_rows=T, eps=1e-6, k_scale=0.18)
    for slots in (1, 8, 64):
        st = torch.zeros(slots, 2, B * H, dtype=torch.float32, device=dev)
        q2, k2 = q0.clone(), k0.clone()
        ops.qknorm_rope(q2, k2, qw, qb, kw, kb, cos, sin, heads=H, text_rows=T, eps=1e-6, k_scale=0.18, stats=st)
        assert torch.equal(q2, q) and torch.equal(k2, k)
        got = st.amax(0)                                                           # [2, B * H]
        for which, t in enumerate((q, k)):
            n2 = t.float().view(B, S, H, 64).pow(2).sum(-1)                         # [B, S, H]
            want = n2.amax(1).reshape(B * H)
            assert torch.allclose(got[which], want, rtol=1e-5, atol=0), (slots, which)
            assert bool((n2.permute(0, 2, 1).reshape(B * H, S) <= got[which][:, None] * (1 + 1e-5)).all())
    st = torch.zeros(4, 2, B * H, dtype=torch.float32, device=dev)
    q3, k3 = q0.clone(), k0.clone()
    ops.qknorm_rope(q3, None, qw, qb, kw, kb, cos, sin, heads=H, text_rows=T, eps=1e-6, k_scale=0.18, stats=st)
    ops.qknorm_rope(None, k3, qw, qb, kw, kb, cos, sin, heads=H, text_rows=T, eps=1e-6, k_scale=0.18, stats=st)
    assert torch.equal(q3, q) and torch.equal(k3, k)
    assert torch.allclose(st.amax(0)[0], q.float().view(B, S, H, 64).pow(2).sum(-1).amax(1).reshape(-1), rtol=1e-5)
    assert torch.allclose(st.amax(0)[1], k.float().view(B, S, H, 64).pow(2).sum(-1).amax(1).reshape(-1), rtol=1e-5)


@pytest.mark.parametrize("S,H", [(777, 8), (1500, 6)])
def test_joint_attention_device_bound_with_per_head_fallback(ops, dev, S, H):
    """The data-dependent bound: heads whose max||q|| max||k|| is within the limit (90, exp2 units) run on the static-bound
    one-wave-per-SIMD kernel, the others -- flagged by that kernel -- on the running-maximum kernel launched right behind it.
    Half the heads here have large q (bound ~ 300): flags must say exactly which, and EVERY head must match the fp32
    softmax at the attention bar (3e-3)."""
    torch.manual_seed(5)
    q = torch.randn(1, S, H, 64, device=dev) * 1.2
    k = torch.randn(1, S, H, 64, device=dev) * 0.6
    v = torch.randn(1, S, H, 64, device=dev)
    big = torch.arange(H, device=dev) % 2 == 1
    q[:, :, big] *= 6.0                                                            # |q| ~ 58, |k| ~ 4.8 -> bound ~ 280 > 90
    qb, kb_, vb = (t.to(torch.bfloat16).reshape(1, S, H * 64).contiguous() for t in (q, k, v))
    st = torch.zeros(8, 2, H, dtype=torch.float32, device=dev)
    n2q = qb.float().view(S, H, 64).pow(2).sum(-1).amax(0)
    n2k = kb_.float().view(S, H, 64).pow(2).sum(-1).amax(0)
    st[3, 0], st[5, 1] = n2q, n2k                                                  # (any slot: the kernel takes the maximum)
    flags = torch.full((64,), -7, dtype=torch.int32, device=dev)
    out = torch.empty_like(qb)
    ops.ATTN_VARIANTS.clear()
    ops.self_attention(qb, kb_, vb, out, heads=H, tag="t", prescaled=True, bound=(st, 0, flags))
    torch.cuda.synchronize()
    assert ops.ATTN_VARIANTS == {("t", "d64_device_bound_w4"): 1}
    bound = n2q.sqrt() * n2k.sqrt()
    assert torch.equal(flags[:H].bool(), bound > 90.0) and bool(flags[:H].bool().any()) and not bool(flags[:H].bool().all())
    sc = torch.einsum("bqhd,bkhd->bhqk", qb.float().view(1, S, H, 64), kb_.float().view(1, S, H, 64)) * math.log(2.0)
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(sc, -1), vb.float().view(1, S, H, 64))
    for h in range(H):
        e = rel_fro(out.view(1, S, H, 64)[:, :, h], ref[:, :, h])
        assert e <= 3e-3, (h, bool(flags[h]), e)
    # the table of another launch geometry: this launch's heads at a column offset (the sharded step's form)
    st2 = torch.zeros(2, 2, H + 5, dtype=torch.float32, device=dev)
    st2[1, 0, 5:], st2[0, 1, 5:] = n2q, n2k
    out2 = torch.empty_like(qb)
    ops.self_attention(qb, kb_, vb, out2, heads=H, tag="t", prescaled=True, bound=(st2, 5, flags))
    assert torch.equal(out2, out)


# ----------------------------------------------------------------------------------------------- small linears
def test_linear_small_m_and_timestep(ops, dev):
    B, dim = 2, 3072
    t = torch.tensor([999, 500], dtype=torch.int64, device=dev)
    feat = torch.empty(B, dim, dtype=torch.bfloat16, device=dev)
    ops.timestep_features(t, feat)
    half = dim // 2
    e = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=dev) / half)
    ang = t[:, None].float() * e[None]
    ref = torch.cat([torch.cos(ang), torch.sin(ang)], -1)
    assert (feat.float() - ref).abs().max() < 1.2e-2          # bf16 rounding of values in [-1, 1] (+1 ulp of cos)
    w1, b1 = rnd((512, dim), dev, 80, dim ** -0.5), rnd((512,), dev, 81, 0.1)
    h = torch.empty(B, 512, dtype=torch.bfloat16, device=dev)
    ops.linear_small_m(feat, w1, b1, h)
    check(h, feat.float() @ w1.float().T + b1.float(), what="linear_small_m")
    w2, b2 = rnd((18432, 512), dev, 82, 512 ** -0.5), rnd((18432,), dev, 83, 0.1)
    m = torch.empty(B, 18432, dtype=torch.bfloat16, device=dev)
    ops.linear_small_m(h, w2, b2, m, silu_in=True)
    check(m, bf(F.silu(h.float())).float() @ w2.float().T + b2.float(), what="linear_small_m silu_in")


# ----------------------------------------------------------------------------------------------- router pieces
@pytest.mark.parametrize("NID,N", [(2, 150), (2, 17550), (3, 4099), (4, 6001)])
def test_router_scores(ops, dev, monkeypatch, NID, N):
    """From 4096 tokens on the identity's 32 keys are resident in LDS (one workgroup per CU, whole identities, every wave on
    its own 16-token tiles); the reference form "router_scores_wave" keeps the kernel that re-reads them per wave.  Same MFMA order:
    bit-identical, ragged last tile and 2-4 identities included."""
    qr, kr = rnd((N, 2048), dev, 90), rnd((NID, 32, 2048), dev, 91, 0.2)
    w, b = rnd((512,), dev, 92, 0.2) + 1, rnd((512,), dev, 93, 0.2)
    pos = rnd((N, 512), dev, 94)
    out = torch.empty(NID, N, 512, dtype=torch.bfloat16, device=dev)
    ops.router_scores(qr, kr, w, b, pos, out, NID, N)
    qh = qr.float().view(1, N, 16, 128).transpose(1, 2)                  # [1,16,N,128]
    kh = kr.float().view(NID, 32, 16, 128).transpose(1, 2)              # [NID,16,32,128]
    s = (qh @ kh.transpose(-2, -1)).permute(0, 2, 3, 1).reshape(NID, N, 512)
    ref = bf(F.layer_norm(s, (512,), w.float(), b.float(), 1e-5)).float() + pos.float()
    check(out, ref, tol=2e-3, what="router_scores")
    out0 = torch.empty_like(out)
    with ops.options(reference_forms="router_scores_wave"):
        ops.router_scores(qr, kr, w, b, pos, out0, NID, N)
    assert torch.equal(out, out0)


def test_router_head_and_forcing(ops, dev):
    NID, N, D = 2, 700, 512
    x = rnd((NID, N, D), dev, 95)
    w, b = rnd((D,), dev, 96, D ** -0.5), rnd((1,), dev, 97)
    r = torch.empty(N, NID, dtype=torch.bfloat16, device=dev)
    ops.router_head(x, w, b, r, NID, N)
    ref = torch.sigmoid(bf(x.float() @ w.float() + b.float()).float()).T
    check(r, ref, tol=2e-3, what="router_head")
    # forcing: exact
    T, per = 13, 54
    g = torch.Generator().manual_seed(5)
    f = (torch.rand(T, per, 2, generator=g) > 0.8).to(torch.bfloat16).to(dev)
    o = torch.empty_like(f)
    ops.forcing_max_over_frames(f, o, T, per, 2)
    assert torch.equal(o, f.max(0).values[None].expand(T, -1, -1))


# ----------------------------------------------------------------------------------------------- combines / patches
@pytest.mark.parametrize("hard", [False, True])
def test_masked_combine(ops, dev, hard):
    B, NID, N, D, T = 2, 2, 330, 3072, 26
    xbuf = rnd((B, T + N, D), dev, 100)
    feat = rnd((B, NID, N, D), dev, 101)
    if hard:
        r = (torch.rand(1, N, NID, generator=torch.Generator().manual_seed(7)) > 0.5).to(torch.bfloat16).to(dev)
    else:
        r = bf(torch.rand(B, N, NID, generator=torch.Generator().manual_seed(8))).to(dev)
    af = bf(torch.stack([torch.eye(2), 1 - torch.eye(2)])).to(dev)
    for mode, alpha in (("face", 1.0), ("face", 0.5), ("audio", 1.0)):
        x = xbuf.clone()
        ops.masked_combine(x[:, T:], feat, r, af if mode == "audio" else None, mode, alpha)
        rr = r.expand(B, -1, -1)
        if mode == "audio":
            av = bf(af.float() @ rr.float().transpose(-2, -1)).transpose(-2, -1)          # [B, N, 2]
            wgt = bf(1 - av[:, :, [1, 0]].float())
        else:
            wgt = rr
        # bf16 op-by-op restatement of models/transformer.py:821-832 / 925-936 -> must match bit for bit
        mix = bf(torch.einsum("bni,bind->bnd", wgt.float(), feat.float()))
        if alpha != 1.0:
            mix = bf(alpha * mix.float())
        ref = bf(xbuf[:, T:].float() + mix.float())
        assert torch.equal(x[:, T:], ref), f"{mode} alpha={alpha} hard={hard}"
        assert torch.equal(x[:, :T], xbuf[:, :T])


@pytest.mark.parametrize("mode", ["face", "audio"])
def test_routed_mix_then_projection_equals_project_then_combine(ops, dev, mode):
    """Route-then-project (engine default) against the reference order of operations
    (models/transformer.py:821-832 / 925-936): equal up to bf16 rounding; exact row selection on hard masks."""
    B, NID, N, Dp, D = 2, 2, 260, 256, 512
    o = rnd((B, NID, N, Dp), dev, 130)
    w_out, b_out = rnd((D, Dp), dev, 131, Dp ** -0.5), rnd((D,), dev, 132, 0.3)
    x = rnd((B, N, D), dev, 133)
    r = bf(torch.rand(B, N, NID, generator=torch.Generator().manual_seed(9))).to(dev)
    r[:, :40] = (r[:, :40] > 0.5).to(torch.bfloat16)                      # some hard-routed tokens
    af = bf(torch.stack([torch.eye(2), 1 - torch.eye(2)])).to(dev)
    z = torch.empty(B, N, Dp, dtype=torch.bfloat16, device=dev)
    wsum = torch.empty(B, N, dtype=torch.float32, device=dev)
    ops.routed_mix(o, r, af if mode == "audio" else None, mode, z, wsum)
    out = x.clone()
    ops.gemm(z, w_out, out, bias=b_out, res=out, bias_rowscale=wsum, alpha=1.0)
    if mode == "audio":
        av = bf(af.float() @ r.float().transpose(-2, -1)).transpose(-2, -1)
        wgt = bf(1 - av[:, :, [1, 0]].float()).float()
    else:
        wgt = r.float()
    assert torch.allclose(wsum, wgt.sum(-1), atol=1e-6)
    feat = o.float() @ w_out.float().T + b_out.float()                     # [B, NID, N, D]
    ref = x.float() + torch.einsum("bni,bind->bnd", wgt, feat)
    check(out, ref, tol=2e-3, what=f"routed mix {mode}")


@pytest.mark.parametrize("mode", ["face", "audio"])
@pytest.mark.parametrize("M", [300, 17550])
def test_routed_mix_projection_exact_row_selection(ops, dev, mode, M):
    """The DEFAULT engine path (``routed_mix`` -> ``bya_gemm_bf16`` with residual / row-scaled bias, engine.py G1/G2)
    on hard one-hot masks, integer-valued features, a permutation matrix as ``to_out`` and integer biases: every
    intermediate is exactly representable, so the result must equal the reference's order of operations
    (models/transformer.py:821-832 face; :895-936 audio with the ``[1,0]`` swap, ``1 - x`` and the af mixing)
    BIT FOR BIT -- a swapped identity, a wrong row or a wrong mask column cannot hide behind a tolerance.
    M = 17550 with K = N = 1024 runs the pipelined 256x256 GEMM (+ its 128x128 tail split) like the full-size step."""
    B, NID = 2, 2
    Dp, D = (256, 512) if M < 1024 else (1024, 1024)
    g = torch.Generator().manual_seed(77 + M)
    o = torch.randint(-8, 9, (B, NID, M, Dp), generator=g).to(torch.bfloat16).to(dev)
    perm = torch.randperm(Dp, generator=g)
    w_out = torch.zeros(D, Dp)
    w_out[torch.arange(D), perm[torch.arange(D) % Dp]] = 1.0             # out[:, j] = z[:, perm[j % Dp]]
    w_out = w_out.to(torch.bfloat16).to(dev)
    b_out = torch.randint(-3, 4, (D,), generator=g).to(torch.bfloat16).to(dev)
    x = torch.randint(-16, 17, (B, M, D), generator=g).to(torch.bfloat16).to(dev)
    lab = torch.randint(-1, NID, (1, M), generator=g)                     # -1 = background, else the identity
    r = torch.zeros(1, M, NID)
    for i in range(NID):
        r[0, lab[0] == i, i] = 1.0
    r = r.to(torch.bfloat16).to(dev)
    af = torch.stack([torch.eye(2), 1 - torch.eye(2)]).to(torch.bfloat16).to(dev)      # sample 0: eye, sample 1: swap
    z = torch.empty(B, M, Dp, dtype=torch.bfloat16, device=dev)
    wsum = torch.empty(B, M, dtype=torch.float32, device=dev)
    out = x.clone()
    if mode == "face":
        ops.routed_mix(o, r, None, "face", z)
        ops.gemm(z, w_out, out, res=out, alpha=1.0)
        wgt = r.float().expand(B, -1, -1)
        bias = torch.zeros(D, device=dev)
    else:
        ops.routed_mix(o, r, af, "audio", z, wsum)
        ops.gemm(z, w_out, out, bias=b_out, res=out, bias_rowscale=wsum)
        av = (af.float() @ r.float().expand(B, -1, -1).transpose(-2, -1)).transpose(-2, -1)      # [B, M, 2]
        wgt = 1 - av[:, :, [1, 0]]
        bias = b_out.float()
    # reference order: project every identity (+ bias), then combine with the mask weights, then add
    feat = o.float()[..., perm[torch.arange(D) % Dp].to(dev)] + bias                            # [B, NID, M, D]
    ref = x.float() + torch.einsum("bni,bind->bnd", wgt, feat)
    assert ref.abs().max() < 256                                                                 # exact in bf16
    assert torch.equal(out.float(), ref), f"{mode}: {(out.float() - ref).abs().max().item()}"
    # and the selection really happened: foreground rows carry exactly one identity's features
    k = int((lab[0] == 0).nonzero()[0])
    if mode == "face":
        assert torch.equal(out[0, k].float() - x[0, k].float(), feat[0, 0, k])
    else:       # sample 0 (af = eye): id 0's region hears id 0 only; sample 1 (af swapped): it hears id 1 only
        assert torch.equal(out[0, k].float() - x[0, k].float(), feat[0, 0, k])
        assert torch.equal(out[1, k].float() - x[1, k].float(), feat[1, 1, k])


def test_patchify_unpatchify_exact(ops, dev):
    B, T, C, H, W = 2, 3, 48, 12, 20
    x = rnd((B, T, C, H, W), dev, 110)
    cols = torch.empty(B, T * (H // 2) * (W // 2), C * 4, dtype=torch.bfloat16, device=dev)
    ops.patchify(x, cols)
    ref = x.view(B, T, C, H // 2, 2, W // 2, 2).permute(0, 1, 3, 5, 2, 4, 6).reshape(B, -1, C * 4)
    assert torch.equal(cols, ref)
    # conv2d(k=2, s=2) == cols @ weight.view(out, C*4)^T
    w = rnd((64, C, 2, 2), dev, 111, 0.1)
    conv = F.conv2d(x.float().view(B * T, C, H, W), w.float(), stride=2)
    conv = conv.view(B, T, 64, -1).transpose(2, 3).reshape(B, -1, 64)
    assert rel_fro(cols.float() @ w.float().view(64, -1).T, conv) < 1e-5
    # unpatchify against models/transformer.py:956-957
    CO = 16
    y = rnd((B, T * (H // 2) * (W // 2), CO * 4), dev, 112)
    out = torch.empty(B, T, CO, H, W, dtype=torch.bfloat16, device=dev)
    ops.unpatchify(y, out)
    ref = y.reshape(B, T, H // 2, W // 2, -1, 2, 2).permute(0, 1, 4, 2, 5, 3, 6).flatten(5, 6).flatten(3, 4)
    assert torch.equal(out, ref)


def test_act_add(ops, dev):
    x, r = rnd((1000, 1024), dev, 120), rnd((1000, 1024), dev, 121)
    out = torch.empty_like(x)
    ops.act_add(x, out, act="leaky_relu")
    check(out, F.leaky_relu(x.float()), what="leaky")
    ops.act_add(x, out, act="gelu_erf", res=r)
    check(out, bf(F.gelu(x.float())).float() + r.float(), what="gelu+res")


@pytest.mark.parametrize("S", [40, 200, 1350, 4133])
def test_attn_static_bound_softmax(ops, dev, S):
    """score_bound: q and k with ||q|| <= 8, ||k|| <= 8 * k_scale (what LayerNorm(64) + RoPE guarantee), so every score is
    within +-11.6 in exp2 units and the kernel may drop the running maximum (P = exp2(s)).  Must agree with the fp32 softmax
    like the running-max kernel does, including rows whose scores sit far below the bound.  The hand-placed
    one-wave-per-SIMD kernel (csrc/attn_w4.hip, 512 query rows per workgroup) is the only static-bound kernel since round 5.
    Sizes: fewer keys than one tile (40: the keys past Skv of the only tile are masked), ragged tails
    (200 = 3 x 64 + 8, 4133), fewer rows than a workgroup holds."""
    H, D = 4, 64
    g = torch.Generator().manual_seed(S)
    def unit_rows(scale_rows):
        x = torch.randn(1, S, H, D, generator=g)
        x = x / x.norm(dim=-1, keepdim=True) * 8.0 * scale_rows
        return x
    k_scale = D ** -0.5 * 1.4426950408889634
    q = bf(unit_rows(torch.rand(1, S, H, 1, generator=g) * 0.9 + 0.1)).to(dev)        # some rows far below the bound
    k32 = unit_rows(torch.ones(1, S, H, 1))
    k = bf(k32 * k_scale).to(dev)                                                    # scale folded into k (prescaled)
    v = rnd((1, S, H * D), dev, 3)
    out = torch.empty(1, S, H * D, dtype=torch.bfloat16, device=dev)
    bound = 1.02 * 64 * k_scale
    ops.self_attention(q.view(1, S, H * D), k.view(1, S, H * D), v, out, heads=H, prescaled=True, score_bound=bound)
    qf, kf, vf = (t_.float().view(1, S, H, D).transpose(1, 2) for t_ in (q, k, v))
    ref = torch.softmax(qf @ kf.transpose(-1, -2) * math.log(2.0), dim=-1) @ vf
    check(out.view(1, S, H, D).transpose(1, 2), ref, tol=ATTN_TOL, what=f"attention static bound S={S}")
    out2 = torch.empty_like(out)
    ops.self_attention(q.view(1, S, H * D), k.view(1, S, H * D), v, out2, heads=H, prescaled=True, score_bound=500.0)
    check(out2.view(1, S, H, D).transpose(1, 2), ref, tol=ATTN_TOL, what="unusable bound -> running-max kernel")


@pytest.mark.parametrize("S", [200, 1350, 17776])
def test_attn_static_bound_softmax_is_shift_tolerant(ops, dev, S):
    """Softmax does not care about a constant added to every score of a row; the static-bound kernel (P = exp2(s), no running
    maximum) must not either while the scores stay inside its bound.  q = a u + noise, k = -c u + noise puts a constant of
    about -5 ... -44 (exp2 units) under every score of a head -- what q/k-LayerNorm biases of opposite sign do.  Until round 5
    the keys past Skv of the last tile counted as P = 1 each and were subtracted from the row sum at the end: rows whose real
    sum was tiny lost it in that subtraction (16 + 1e-9 - 16).  They are masked to -inf now.  Every S here leaves a ragged last
    tile (8, 6 and 48 keys)."""
    H, D = 4, 64
    g = torch.Generator().manual_seed(7 * S + 1)
    u = torch.randn(D, generator=g)
    u = u / u.norm()
    k_scale = D ** -0.5 * 1.4426950408889634
    off = torch.tensor([3.0, 6.0, 9.0, 0.0]).view(1, 1, H, 1)                      # head 3: no offset (the ordinary case)
    q32 = off * u + torch.randn(1, S, H, D, generator=g) * 0.5
    k32 = -off * u * 3.0 + torch.randn(1, S, H, D, generator=g) * 0.5
    q, k = bf(q32).to(dev), bf(k32 * k_scale).to(dev)
    v = rnd((1, S, H * D), dev, 5)
    qf, kf, vf = (t_.float().view(1, S, H, D).transpose(1, 2) for t_ in (q, k, v))
    scores = qf @ kf.transpose(-1, -2)
    assert scores[0, 2].max().item() < -20 and scores.abs().max().item() < 85, (scores[0, 2].max().item(), scores.abs().max().item())
    ref = torch.softmax(scores * math.log(2.0), dim=-1) @ vf
    out = torch.empty(1, S, H * D, dtype=torch.bfloat16, device=dev)
    ops.ATTN_VARIANTS.clear()
    ops.self_attention(q.view(1, S, H * D), k.view(1, S, H * D), v, out, heads=H, prescaled=True, score_bound=88.0)
    assert [v for (_, v) in ops.ATTN_VARIANTS] == ["d64_static_bound_w4"], ops.ATTN_VARIANTS
    for h in range(H):
        check(out.view(1, S, H, D).transpose(1, 2)[:, h], ref[:, h], tol=ATTN_TOL, what=f"shifted scores S={S} head {h}")


@pytest.mark.parametrize("S,H", [(17776, 48), (5000, 16), (33976, 8), (17776, 12)])
def test_attn_stream_k_matches_one_workgroup_per_item(ops, dev, S, H, monkeypatch):
    """The joint attention as 256 persistent workgroups over evenly cut (item, key-tile) ranges (stream-K; workspace
    registered by ops.attention) against the one-workgroup-per-item launch of the same kernel (option attn_streamk = 0).
    Partials of the static-bound softmax are additive, so an item cut between two workgroups differs from the uncut form by
    fp32 summation order only: (1) every row outside the <= 248 cut items is bit-identical, (2) cut rows agree to bf16
    rounding, (3) both agree with the fp32 softmax on sampled heads, (4) 8 launches on a busy GPU are bit-identical (the
    suffix -> prefix hand-off: flag, fences, flag reset) and no hand-off timed out.  Shapes: BASELINE configs[1] (1680 items,
    6.56 rounds), a small one (160 items < 256 CUs: stream-K declines, both launches are the same kernel), the 97-frame
    sequence with 8 heads (536 items, ragged last q-tile and ragged last key tile) and a 4-rank shard's 12 heads (8 does not
    divide the head count: an XCD owns a contiguous eighth of the (head, q-tile) order, 52 or 53 items)."""
    D = 64
    g = torch.Generator().manual_seed(S + H)
    def unit_rows(scale_rows):
        x = torch.randn(1, S, H, D, generator=g)
        return x / x.norm(dim=-1, keepdim=True) * 8.0 * scale_rows
    k_scale = D ** -0.5 * 1.4426950408889634
    q = bf(unit_rows(torch.rand(1, S, H, 1, generator=g) * 0.9 + 0.1)).to(dev).view(1, S, H * D)
    k = bf(unit_rows(torch.ones(1, S, H, 1)) * k_scale).to(dev).view(1, S, H * D)
    v = rnd((1, S, H * D), dev, 3)
    bound = 1.02 * 64 * k_scale
    def run():
        out = torch.empty(1, S, H * D, dtype=torch.bfloat16, device=dev)
        ops.self_attention(q, k, v, out, heads=H, prescaled=True, score_bound=bound, tag="joint")
        return out
    sk = run()
    with ops.options(attn_streamk=0):
        plain = run()
    diff_rows = (sk != plain).view(S, H, D).any(dim=-1)                       # [S, H]
    n_diff_tiles = int(diff_rows.view(-1, H).float().sum().item())
    # a cut item is one (head, 512-row q-tile): at most 248 of them
    cut_items = set()
    idx = diff_rows.nonzero()
    for row, head in idx[:: max(1, len(idx) // 20000)].tolist():
        cut_items.add((head, row // 512))
    print(f"S={S} H={H}: {n_diff_tiles} (row, head) pairs differ, in {len(cut_items)} (head, q-tile) items (sampled)")
    assert len(cut_items) <= 248 or H * ((S + 511) // 512) <= 256
    e = rel_fro(sk.float(), plain.float())
    assert e < 2e-3, e
    for head in (0, H - 1):
        sl = slice(head * D, (head + 1) * D)
        rows = torch.arange(0, S, 7, device=dev)[:700]
        s_ = (q[0, rows, sl].float() @ k[0, :, sl].float().T) * math.log(2.0)
        ref = torch.softmax(s_, dim=-1) @ v[0, :, sl].float()
        check(sk[0, rows, sl], ref, tol=ATTN_TOL, what=f"stream-K attention head {head}")
    a, bb = rnd((8192, 8192), dev, 86), rnd((8192, 8192), dev, 87)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(4):
            a @ bb
    outs = [run() for _ in range(8)]
    torch.cuda.synchronize()
    assert all(torch.equal(sk, o) for o in outs)
    assert ops.attn_workspace_status() == 0


@pytest.mark.parametrize("D,H,Sq,Skv,kw", [(64, 8, 1350, 1350, {}), (64, 3, 333, 97, dict(prescaled=True)),
                                           (128, 2, 577, 32, {}), (64, 48, 1031, 1031, dict(prescaled=True, score_bound=11.8))])
@pytest.mark.parametrize("o_off", [0, 4])
def test_attn_wide_epilogue_stores_match_the_8_byte_ones(ops, dev, monkeypatch, D, H, Sq, Skv, kw, o_off):
    """The attention epilogues store 16 bytes per lane after a v_permlane32_swap exchange between the half-waves (32
    contiguous bytes per row and instruction instead of 16); the reference form "attn_narrow_store", or an output that is only 8-byte
    aligned (o_off = 4 elements), keeps the 8-byte stores.  Same values, same addresses: bit-identical, ragged last
    q tile included, and nothing written outside the heads' columns (generic d64 / d128 kernels and the joint w4 kernel)."""
    E = H * D
    q, k, v = rnd((1, Sq, E), dev, 11), rnd((1, Skv, E), dev, 12), rnd((1, Skv, E), dev, 13)
    if "score_bound" in kw:
        nrm = lambda t: (t.float() / t.float().view(1, -1, H, D).norm(dim=-1, keepdim=True).repeat_interleave(D, -1).view(t.shape) * 8)
        q, k = nrm(q).to(torch.bfloat16), (nrm(k) * (0.125 * 1.4426950408889634)).to(torch.bfloat16)
    W = E + 16
    buf = torch.empty(Sq * W + 8, dtype=torch.bfloat16, device=dev)
    def run(flag):
        buf.fill_(5.0)
        o = buf[o_off:o_off + Sq * W].view(1, Sq, W)[..., 8:8 + E]
        with ops.options(reference_forms=[] if flag else "attn_narrow_store"):
            ops.attention(q, k, v, o, head_dim=D, heads=H, nb1=1, nb2=1, Sq=Sq, Skv=Skv, q_strides=(0, 0, E), k_strides=(0, 0, E),
                          v_strides=(0, 0, E), o_strides=(0, 0, W), scale=D ** -0.5, **kw)
        torch.cuda.synchronize()
        return buf.clone()
    a, b = run(True), run(False)
    assert torch.equal(a, b)
    rows = a[o_off:o_off + Sq * W].view(Sq, W)
    assert bool((rows[:, :8] == 5.0).all()) and bool((rows[:, 8 + E:] == 5.0).all()) and not bool((rows[:, 8:8 + E] == 5.0).all())


@pytest.mark.parametrize("mode,D,H,n_id,grp,Sq", [("audio", 64, 48, 2, 13, 150), ("audio", 64, 6, 3, 4, 333),
                                                  ("face", 128, 16, 2, 1, 1000), ("face", 128, 16, 3, 1, 130)])
def test_attn_kv_mix_equals_attention_then_routed_mix(ops, dev, mode, D, H, n_id, grp, Sq):
    """bya_attn_kv_mix (cross-attention onto 32 keys per identity with the masked combine in its epilogue) against the two
    launches it replaces, bya_attn_fwd + bya_routed_mix, and against fp32 torch; with ONE-HOT routing masks the fused kernel
    must return exactly the selected identity's attention output, bit for bit with the unfused attention (the mix is
    1.0 * o + 0.0 * o': row selection is index work)."""
    g = torch.Generator().manual_seed(D + H + n_id)
    Skv, E = 32, H * D
    q = rnd((grp, Sq, E), dev, 1)
    k = rnd((n_id, grp, Skv, E), dev, 2)
    v = rnd((n_id, grp, Skv, E), dev, 3)
    N = grp * Sq
    r_soft = torch.sigmoid(torch.randn(N, n_id, generator=g)).to(torch.bfloat16).to(dev)
    lab = torch.randint(-1, n_id, (N,), generator=g)
    r_hard = torch.zeros(N, n_id)
    for i in range(n_id):
        r_hard[lab == i, i] = 1
    r_hard = r_hard.to(torch.bfloat16).to(dev)
    af = None if mode == "face" else torch.roll(torch.eye(n_id), 1, dims=1).to(torch.bfloat16).to(dev)
    scale = D ** -0.5
    # unfused: attention per identity (q shared), then the routed mix
    ao = torch.empty(1, n_id, N, E, dtype=torch.bfloat16, device=dev)
    ops.attention(q, k, v, ao[0], head_dim=D, heads=H, nb1=n_id, nb2=grp, Sq=Sq, Skv=Skv, q_strides=(0, Sq * E, E),
                  k_strides=(grp * Skv * E, Skv * E, E), v_strides=(grp * Skv * E, Skv * E, E),
                  o_strides=(N * E, Sq * E, E), scale=scale)
    for r in (r_soft, r_hard):
        z_ref, ws_ref = torch.empty(1, N, E, dtype=torch.bfloat16, device=dev), torch.empty(1, N, dtype=torch.float32, device=dev)
        ops.routed_mix(ao, r[None], None if af is None else af[None], mode, z_ref, ws_ref)
        z, ws = torch.empty(grp, Sq, E, dtype=torch.bfloat16, device=dev), torch.full((N,), -1.0, dtype=torch.float32, device=dev)
        ops.attn_kv_mix(q, k, v, r, af, z, ws, head_dim=D, heads=H, n_id=n_id, n_grp=grp, Sq=Sq, Skv=Skv,
                        q_strides=(Sq * E, E), k_strides=(grp * Skv * E, Skv * E, E), v_strides=(grp * Skv * E, Skv * E, E),
                        z_strides=(Sq * E, E), scale=scale)
        assert torch.equal(ws, ws_ref[0])
        e = rel_fro(z.view(N, E).float(), z_ref[0].float())
        print(f"attn_kv_mix {mode} d{D} n_id {n_id}: vs attention + routed_mix {e:.3e}")
        assert e < 4e-3                                        # one rounding instead of two
    if mode == "face":                                         # hard masks: exactly the selected identity's rows
        zv = z.view(N, E)
        for i in range(n_id):
            assert torch.equal(zv[(lab == i).to(dev)], ao[0, i][(lab == i).to(dev)])
        assert bool((zv[(lab == -1).to(dev)] == 0).all())
    # fp32 reference of the whole thing (soft masks)
    qf = q.float().view(grp, Sq, H, D).permute(0, 2, 1, 3)
    outs = []
    for i in range(n_id):
        kf = k[i].float().view(grp, Skv, H, D).permute(0, 2, 1, 3)
        vf = v[i].float().view(grp, Skv, H, D).permute(0, 2, 1, 3)
        outs.append((torch.softmax(qf @ kf.transpose(-1, -2) * scale, -1) @ vf).permute(0, 2, 1, 3).reshape(N, E))
    if mode == "face":
        w = r_soft.float()
    else:
        av = (r_soft.float() @ af.float().T).to(torch.bfloat16).float()
        comp = (1 - av).to(torch.bfloat16).float()
        cols = []
        for a in range(n_id):
            t_ = torch.ones(N, device=dev)
            for b_ in range(n_id):
                if b_ != a:
                    t_ = (t_ * comp[:, b_]).to(torch.bfloat16).float()
            cols.append(t_)
        w = torch.stack(cols, -1)
    ref = sum(w[:, i:i + 1] * outs[i] for i in range(n_id))
    z2 = torch.empty(grp, Sq, E, dtype=torch.bfloat16, device=dev)
    ops.attn_kv_mix(q, k, v, r_soft, af, z2, None, head_dim=D, heads=H, n_id=n_id, n_grp=grp, Sq=Sq, Skv=Skv,
                    q_strides=(Sq * E, E), k_strides=(grp * Skv * E, Skv * E, E), v_strides=(grp * Skv * E, Skv * E, E),
                    z_strides=(Sq * E, E), scale=scale)
    check(z2.view(N, E), ref, tol=ATTN_TOL, what=f"attn_kv_mix {mode} vs fp32")


@pytest.mark.parametrize("mode,D,H,n_id,grp,Sq,Skv", [("audio", 64, 48, 2, 13, 1350, 32), ("audio", 64, 6, 4, 3, 333, 32),
                                                      ("audio", 64, 5, 3, 2, 97, 17), ("face", 128, 16, 2, 1, 17550, 32),
                                                      ("face", 128, 2, 1, 1, 31, 32), ("face", 128, 3, 4, 2, 1000, 9)])
def test_attn_kv_mix_32_key_form_matches_one_tile_per_workgroup(ops, dev, monkeypatch, mode, D, H, n_id, grp, Sq, Skv):
    """Up to 32 keys per identity bya_attn_kv_mix runs the persistent form (K / V of every identity resident in LDS, every
    wave walking its own 32-row tiles, z stored as whole head segments through LDS); the reference form "kv_mix_generic" keeps the
    one-128-row-tile-per-workgroup kernel on 64-key tiles.  Same arithmetic per element -> BIT-IDENTICAL z and weight
    sums, at the step's two shapes, with ragged row counts, 1-4 identities, fewer than 32 keys, strided q / k / v / z."""
    E = H * D
    pad = 64                                                      # rows are wider than the heads they carry
    qw, zw = rnd((grp, Sq, E + pad), dev, 1), torch.empty(grp, Sq, E + pad, dtype=torch.bfloat16, device=dev)
    kw, vw = rnd((n_id, grp, Skv, E + pad), dev, 2), rnd((n_id, grp, Skv, E + pad), dev, 3)
    g = torch.Generator().manual_seed(Sq + n_id)
    r = torch.sigmoid(torch.randn(grp * Sq, n_id, generator=g)).to(torch.bfloat16).to(dev)
    af = None if mode == "face" else torch.roll(torch.eye(n_id), 1, dims=1).to(torch.bfloat16).to(dev)
    W = E + pad
    def run(flag):
        zw.fill_(3.0)
        ws = torch.full((grp * Sq,), -1.0, dtype=torch.float32, device=dev)
        with ops.options(reference_forms=[] if flag else "kv_mix_generic"):
            ops.attn_kv_mix(qw[..., pad:], kw[..., :E], vw[..., 8:], r, af, zw[..., 16:], ws, head_dim=D, heads=H, n_id=n_id, n_grp=grp,
                            Sq=Sq, Skv=Skv, q_strides=(Sq * W, W), k_strides=(grp * Skv * W, Skv * W, W),
                            v_strides=(grp * Skv * W, Skv * W, W), z_strides=(Sq * W, W), scale=D ** -0.5)
        torch.cuda.synchronize()
        return zw.clone(), ws
    z1, w1 = run(True)
    z0, w0 = run(False)
    assert torch.equal(z1, z0) and torch.equal(w1, w0)
    assert bool((z1[..., :16] == 3.0).all()) and bool((z1[..., 16 + E:] == 3.0).all())     # nothing outside the heads' columns


# ----------------------------------------------------------------------------------------------- CFG + scheduler step
@pytest.mark.parametrize("cfg", [False, True])
def test_cfg_ddim_step_bit_exact(ops, dev, cfg):
    """Fused CFG combine + DDIM step == the oracle's expression-by-expression torch restatement, bit for bit
    (every fp32 / bf16 rounding point of torch's type promotion reproduced, no FMA contraction)."""
    from oracle import scheduler as osch
    from bind_your_avatar_implementation_amd.pipeline import DDIMScheduler
    shape = (1, 3, 16, 16, 24)
    pred = rnd(((2 if cfg else 1),) + shape[1:], dev, 40)
    x = rnd(shape, dev, 41)
    s, o = DDIMScheduler(), osch.DDIM()
    s.set_timesteps(50)
    o.set_timesteps(50)
    for t in (999, 499, 19):
        got = s.step(pred, t, x, guidance=6.0 if cfg else 1.0)
        n32 = osch.cfg_combine(pred, 6.0) if cfg else pred.float()
        ref = o.step(n32, t, x).to(torch.bfloat16)
        assert got.dtype == torch.bfloat16 and torch.equal(got, ref), (t, float((got.float() - ref.float()).abs().max()))


def test_cfg_dpm_steps_bit_exact(ops, dev):
    """Three consecutive DPM-Solver++ steps (first-order, second-order with the carried x0, last step) with the
    generator-driven noise draws: latents and carried x0 equal the oracle bit for bit."""
    from oracle import scheduler as osch
    from bind_your_avatar_implementation_amd.pipeline import DPMScheduler
    shape = (1, 3, 16, 16, 24)
    s, o = DPMScheduler(), osch.DPM()
    ts = s.set_timesteps(3, dev)
    o.set_timesteps(3)
    g1, g2 = torch.Generator(device=dev).manual_seed(7), torch.Generator(device=dev).manual_seed(7)
    x = rnd(shape, dev, 50)
    xr, old, old_r = x.clone(), None, None
    for i, t in enumerate(ts):
        pred = rnd((2,) + shape[1:], dev, 60 + i)
        back = ts[i - 1] if i > 0 else None
        x, old = s.step(pred, old, t, back, x, guidance=4.0, generator=g1)
        prev_t = int(t) - 1000 // 3
        noise = torch.randn(shape, generator=g2, device=dev, dtype=torch.bfloat16)
        if old_r is not None and prev_t >= 0:
            noise = torch.randn(shape, generator=g2, device=dev, dtype=torch.bfloat16)
        pr, old_r = o.step(osch.cfg_combine(pred, 4.0), old_r, t, back, xr, noise)
        xr = pr.to(torch.bfloat16)
        assert torch.equal(x, xr), (i, float((x.float() - xr.float()).abs().max()))
        assert torch.equal(old, old_r), i


# ----------------------------------------------------------------------------------------------- mask path
def test_masks_to_routing_logits_bit_exact(ops, dev):
    """Tracking masks -> forcing logits: bit-exact against the reference's own output (golden) at 49x480x720 and against
    the oracle on ragged sizes (non-integer scales in every axis, identical sizes, single frame, 3 identities)."""
    import os
    import numpy as np
    from golden.mask_cases import synthetic_masks
    from oracle.masks import masks_to_routing_logits
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_masks_seed0.npz"))
    for sd in (0, 1):
        want = np.unpackbits(fx[f"logits.seed{sd}"], axis=0)[:17550]
        got = ops.masks_to_routing_logits(torch.from_numpy(synthetic_masks(sd)).to(dev))
        assert got.shape == (1, 17550, 2) and got.dtype == torch.bfloat16
        assert np.array_equal(got[0].float().cpu().numpy().astype(np.uint8), want), sd
    g = torch.Generator().manual_seed(3)
    for (n_id, Ti, Hi, Wi, To, Ho, Wo) in [(2, 25, 100, 147, 7, 13, 21), (3, 13, 30, 45, 13, 30, 45), (2, 1, 64, 64, 1, 8, 8),
                                           (2, 9, 31, 17, 13, 40, 23)]:
        blobs = F.interpolate(torch.rand(n_id, 1, 3, 5, 6, generator=g), size=(Ti, Hi, Wi), mode="trilinear")[:, 0]
        masks = ((blobs > 0.5) * 255).to(torch.uint8)
        ref = masks_to_routing_logits(masks, To, Ho * 2, Wo * 2)
        got = ops.masks_to_routing_logits(masks.to(dev), To, Ho, Wo)
        assert torch.equal(got.float().cpu(), ref), (n_id, Ti, Hi, Wi)
