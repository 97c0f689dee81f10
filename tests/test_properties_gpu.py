"""Size-independent properties at the FULL BASELINE sizes, where an fp32 oracle run would take minutes: round trips,
idempotence, linearity, convexity, permutation equivariance.  Every test runs at the three full geometries of BASELINE.json:
49x480x720 (configs[1]: 13x30x45 = 17550 video tokens, 17776 joint rows), 49x720x1280 (configs[3]: 13x45x80 = 46800
video tokens, 47026 joint rows -- an ODD row count with ragged 256-row tiles everywhere) and 97x480x720 with THREE
identities (configs[4]: 25x30x45 = 33750 video tokens, 33976 joint rows).  They complement the oracle /
golden comparisons of test_kernels_gpu.py and test_forward_gpu.py (small sizes, exact values)."""
import pytest
import torch

from conftest import rel_fro

pytestmark = pytest.mark.gpu

TT, D = 226, 3072


class Geom:
    def __init__(self, T, HT, WT, n_id=2):
        self.T, self.HT, self.WT, self.N, self.n_id = T, HT, WT, T * HT * WT, n_id


@pytest.fixture(scope="module", params=["49x480x720", "49x720x1280", "97x480x720x3ids"])
def G(request):
    return {"49x480x720": Geom(13, 30, 45), "49x720x1280": Geom(13, 45, 80),
            "97x480x720x3ids": Geom(25, 30, 45, n_id=3)}[request.param]


def rnd(shape, dev, seed, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * std).to(torch.bfloat16).to(dev)


@pytest.fixture(scope="module")
def ops():
    from bind_your_avatar_implementation_amd import ops
    return ops


def test_patchify_is_a_bijection_on_the_conditioned_latents(ops, dev, G):
    T, HT, WT, N = G.T, G.HT, G.WT, G.N
    """unpatchify(patchify(x)) == x bit for bit for [1, 13, 48, 2 HT, 2 WT] (both are pure index maps with the same 2x2
    patch order); and every input element appears exactly once in the patch matrix."""
    x = rnd((1, T, 48, 2 * HT, 2 * WT), dev, 1)
    cols = torch.empty(1, N, 48 * 4, dtype=torch.bfloat16, device=dev)
    ops.patchify(x, cols)
    back = torch.empty_like(x)
    ops.unpatchify(cols, back)
    assert torch.equal(back, x)
    assert torch.equal(cols.flatten().sort().values, x.flatten().sort().values)


def test_forcing_max_is_idempotent_and_monotone(ops, dev, G):
    T, HT, WT, N = G.T, G.HT, G.WT, G.N
    """max over frames broadcast back over frames: applying it twice changes nothing; the result dominates the input and
    stays inside {0, 1} for hard masks."""
    g = torch.Generator().manual_seed(2)
    f = (torch.rand(T, HT * WT, G.n_id, generator=g) > 0.93).to(torch.bfloat16).to(dev)
    a, b = torch.empty_like(f), torch.empty_like(f)
    ops.forcing_max_over_frames(f, a, T, HT * WT, G.n_id)
    ops.forcing_max_over_frames(a, b, T, HT * WT, G.n_id)
    assert torch.equal(a, b) and bool((a >= f).all()) and set(a.unique().tolist()) <= {0.0, 1.0}
    assert torch.equal(a[0], a[-1])


def test_masked_combine_hard_masks_select_rows(ops, dev, G):
    T, HT, WT, N = G.T, G.HT, G.WT, G.N
    """With 0/1 routing weights the face combine is a row selection: tokens routed to nobody keep x bit for bit, tokens
    routed to identity i receive exactly bf16(x + feat_i)."""
    x0 = rnd((1, TT + N, D), dev, 3)
    feat = rnd((1, G.n_id, N, D), dev, 4)
    g = torch.Generator().manual_seed(5)
    lab = torch.randint(-1, G.n_id, (N,), generator=g).to(dev)            # -1 background, 0 .. n_id - 1 identity
    r = torch.zeros(1, N, G.n_id, dtype=torch.bfloat16, device=dev)
    for i in range(G.n_id):
        r[0, lab == i, i] = 1
    x = x0.clone()
    ops.masked_combine(x[:, TT:], feat, r, None, "face", 1.0)
    assert torch.equal(x[:, :TT], x0[:, :TT])
    xv, x0v = x[0, TT:], x0[0, TT:]
    assert torch.equal(xv[lab == -1], x0v[lab == -1])
    for i in range(G.n_id):
        want = (x0v[lab == i].float() + feat[0, i][lab == i].float()).to(torch.bfloat16)
        assert torch.equal(xv[lab == i], want)


def test_audio_combine_swaps_speakers_with_af_matrix(ops, dev, G):
    T, HT, WT, N = G.T, G.HT, G.WT, G.N
    """G2: w = 1 - (af @ r^T)^T[:, [1, 0]].  Swapping the audio-face assignment (eye <-> 1 - eye) together with the two
    audio feature maps must give the same hidden states: the [1, 0] column swap is what makes that true."""
    x0 = rnd((1, N, D), dev, 6)
    n = G.n_id
    feat = rnd((1, n, N, D), dev, 7)
    eye = torch.eye(n, dtype=torch.bfloat16, device=dev)[None]
    a, b = x0.clone(), x0.clone()
    if n == 2:
        r = (torch.rand(1, N, 2, generator=torch.Generator().manual_seed(8)) > 0.5).to(torch.bfloat16).to(dev)
        ops.masked_combine(a, feat, r, eye, "audio", 1.0)
        ops.masked_combine(b, feat.flip(1).contiguous(), r.flip(-1).contiguous(), eye, "audio", 1.0)
        assert torch.equal(a, b)                                          # relabelling the identities changes nothing
        return
    # n identities (build-defined weights w[a] = prod_{b != a} (1 - av[b]), oracle/model.py::audio_weights): with one-hot
    # masks a token on face i hears stream i only, a background token hears every stream -- and a cyclic relabelling of
    # identities, streams and masks together changes nothing
    lab = torch.randint(-1, n, (N,), generator=torch.Generator().manual_seed(8)).to(dev)
    r = torch.zeros(1, N, n, dtype=torch.bfloat16, device=dev)
    for i in range(n):
        r[0, lab == i, i] = 1
    ops.masked_combine(a, feat, r, eye, "audio", 1.0)
    perm = [(i + 1) % n for i in range(n)]
    ops.masked_combine(b, feat[:, perm].contiguous(), r[..., perm].contiguous(), eye, "audio", 1.0)
    assert torch.equal(a[0][lab >= 0], b[0][lab >= 0])                    # one term per token: exact under relabelling
    assert rel_fro(b.float(), a.float()) < 4e-3                           # background: a 3-term sum in another order
    for i in range(n):
        want = (x0[0][lab == i].float() + feat[0, i][lab == i].float()).to(torch.bfloat16)
        assert torch.equal(a[0][lab == i], want)


def test_joint_attention_rows_are_convex_combinations(ops, dev, G):
    T, HT, WT, N = G.T, G.HT, G.WT, G.N
    """Every attention output lies inside the per-dimension [min, max] of V over the keys (softmax weights are a convex
    combination), a constant V comes back unchanged, and permuting the keys (with their values) permutes nothing."""
    H, Dh, S = 48, 64, TT + N
    g = torch.Generator().manual_seed(9)
    q = torch.randn(1, S, H, Dh, generator=g)
    k = torch.randn(1, S, H, Dh, generator=g)
    q = (q / q.norm(dim=-1, keepdim=True) * 8).to(torch.bfloat16).to(dev).view(1, S, H * Dh)
    k = (k / k.norm(dim=-1, keepdim=True) * 8 * Dh ** -0.5 * 1.4426950408889634).to(torch.bfloat16).to(dev).view(1, S, H * Dh)
    v = rnd((1, S, H * Dh), dev, 10)
    out = torch.empty_like(v)
    kw = dict(heads=H, prescaled=True, score_bound=1.02 * 64 * Dh ** -0.5 * 1.4426950408889634)
    ops.self_attention(q, k, v, out, **kw)
    lo, hi = v.float().amin(1, keepdim=True), v.float().amax(1, keepdim=True)
    of = out.float()
    tol = 2e-2 * (hi - lo)
    assert bool((of >= lo - tol).all()) and bool((of <= hi + tol).all())
    perm = torch.randperm(S, generator=g).to(dev)
    out_p = torch.empty_like(v)
    ops.self_attention(q, k[:, perm].contiguous(), v[:, perm].contiguous(), out_p, **kw)
    assert rel_fro(out_p.float(), of) < 3e-3                               # summation order changes, nothing else
    const = torch.full_like(v, 0.75)
    ops.self_attention(q, k, const, out, **kw)
    assert float((out.float() - 0.75).abs().max()) < 4e-3


def test_gemm_is_linear_in_the_activations_at_full_size(ops, dev, G):
    T, HT, WT, N = G.T, G.HT, G.WT, G.N
    """(a1 + a2) @ W^T == a1 @ W^T + a2 @ W^T up to bf16 rounding on the (226 + N) x 3072 x 3072 attention-output shape, and
    a zero input returns exactly the bias."""
    a1, a2 = rnd((TT + N, D), dev, 11), rnd((TT + N, D), dev, 12)
    w, b = rnd((D, D), dev, 13, D ** -0.5), rnd((D,), dev, 14)
    o1, o2, o12 = (torch.empty(TT + N, D, dtype=torch.bfloat16, device=dev) for _ in range(3))
    ops.gemm(a1, w, o1)
    ops.gemm(a2, w, o2)
    ops.gemm((a1.float() + a2.float()).to(torch.bfloat16), w, o12)
    assert rel_fro(o12.float(), o1.float() + o2.float()) < 6e-3
    z = torch.empty(TT + N, D, dtype=torch.bfloat16, device=dev)
    ops.gemm(torch.zeros_like(a1), w, z, bias=b)
    assert torch.equal(z, b[None].expand_as(z))


def test_router_rowgemm_is_invariant_to_row_shift_under_layernorm(ops, dev, G):
    T, HT, WT, N = G.T, G.HT, G.WT, G.N
    """LayerNorm removes a per-row offset: rowgemm512 with folded LayerNorm must return (almost) the same q|k|v for x
    and x + c_row -- exercises the matrix-core row statistics at the full 2 N-row router size."""
    M = G.n_id * N
    x = rnd((M, 512), dev, 15)
    shift = torch.randn(M, 1, generator=torch.Generator().manual_seed(16)).to(dev) * 2
    xs = (x.float() + shift).to(torch.bfloat16)
    w, b = rnd((1536, 512), dev, 17, 512 ** -0.5), rnd((1536,), dev, 18, 0.1)
    gam, bet = rnd((512,), dev, 19, 0.2) + 1, rnd((512,), dev, 20, 0.1)
    pack = ops.pack_rowgemm512(w, b, gam, bet)
    o1, o2 = (torch.empty(M, 1536, dtype=torch.bfloat16, device=dev) for _ in range(2))
    ops.rowgemm512(x, pack, o1)
    ops.rowgemm512(xs, pack, o2)
    assert rel_fro(o2.float(), o1.float()) < 8e-3          # bf16 rounding of the shifted rows is the only difference
