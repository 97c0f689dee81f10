"""fp8 weights (BASELINE configs[4]): the row quantiser and the e4m3 GEMM against CPU restatements on the same operands.

The reference has no fp8 path, so the oracle here is the definition in include/bya.h written out in
torch on the CPU: symmetric per-row scales amax / 448, OCP e4m3fn with round-to-nearest-even (torch.float8_e4m3fn), exact
products accumulated in fp32 -- the quantiser is compared BYTE FOR BYTE, the GEMM against the fp64 product of the very
bytes the GPU multiplied, rounded to bf16 (tolerance 1e-3 relative Frobenius, the bar of every bf16-output kernel)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_fro

pytestmark = pytest.mark.gpu


def rnd(shape, seed, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * std).to(torch.bfloat16)


def quant_ref(x):
    """include/bya.h, bya_quantize_rows_fp8, on the CPU."""
    xf = x.float()
    amax = xf.abs().amax(dim=-1, keepdim=True)
    # tensor / tensor: the correctly rounded quotient (torch evaluates `scalar / tensor` as reciprocal * scalar, which is
    # one ulp off for a third of the rows and flips 0.13 % of the bytes at exact ties)
    c448 = torch.full_like(amax, 448.0)
    inv = torch.where(amax > 0, c448 / amax, torch.zeros_like(amax))
    scale = torch.where(amax > 0, amax / c448, torch.ones_like(amax))
    q = (xf * inv).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale.squeeze(-1)


def dequant(q_u8):
    return q_u8.view(torch.float8_e4m3fn).double()


@pytest.mark.parametrize("M,K", [(300, 3072), (64, 12288), (17, 128), (5, 1000)])
def test_quantize_rows_matches_the_definition_byte_for_byte(dev, M, K):
    from bind_your_avatar_implementation_amd import ops
    x = rnd((M, K), 1, std=3.0)
    x[1] = 0                                    # an all-zero row: scale 1, zeros
    x[2, :7] = torch.tensor([448., -448., 1e-3, -0.0, 0.0, 1e4, -3e-5]).to(torch.bfloat16)
    q, s = ops.quantize_rows_fp8(x.to(dev))
    q_ref, s_ref = quant_ref(x)
    assert torch.equal(s.cpu(), s_ref)
    same = (q.cpu() == q_ref)
    print(f"{M}x{K}: {int((~same).sum())} of {same.numel()} bytes differ")
    assert same.all()


@pytest.mark.parametrize("M,N,K,kw", [
    (384, 512, 1024, {}),
    (300, 260, 256, {"bias": True}),                                  # ragged tiles in both directions
    (1000, 768, 3072, {"bias": True, "act": "gelu_tanh"}),
    (777, 1024, 512, {"bias": True, "gate_res": True}),
    (640, 1536, 1024, {"bias": True, "split": True}),
    # >= 200 tiles of 256 x 256: the persistent one-wave-per-SIMD kernel (csrc/gemm_fp8_v4.hip), ragged in M and N,
    # its shortest K (four K-tiles), every epilogue kind
    (4000, 3592, 512, {"bias": True, "act": "gelu_tanh"}),
    (4100, 3328, 1536, {"bias": True, "gate_res": True}),
    (3900, 3840, 1024, {"bias": True, "split": True}),
    (3700, 3848, 3072, {}),
    (4200, 3600, 1664, {"bias": True}),            # an odd number of K-tiles: the LDS ring changes parity from tile to tile
])
def test_gemm_fp8_vs_exact_product_of_the_same_bytes(dev, M, N, K, kw):
    from bind_your_avatar_implementation_amd import ops
    a, w = rnd((M, K), 2), rnd((N, K), 3, std=K ** -0.5)
    bias = rnd((N,), 4) if kw.get("bias") else None
    a8, sa = ops.quantize_rows_fp8(a.to(dev))
    w8, sw = ops.quantize_rows_fp8(w.to(dev))
    ref = (dequant(a8.cpu()) @ dequant(w8.cpu()).T) * sa.cpu().double()[:, None] * sw.cpu().double()[None, :]
    if bias is not None:
        ref = ref + bias.double()
    if kw.get("act") == "gelu_tanh":
        ref = F.gelu(ref, approximate="tanh")
    args = {}
    if kw.get("gate_res"):
        gate, res = rnd((2, N), 5), rnd((M, N), 6)
        split_row = 226
        g = torch.where(torch.arange(M)[:, None] < split_row, gate[0].double()[None], gate[1].double()[None])
        ref = res.double() + g * ref
        args = dict(res=res.to(dev), gate0=gate[0].to(dev).contiguous(), gate1=gate[1].to(dev).contiguous(),
                    gate_split=split_row)
    if kw.get("split"):
        out = torch.empty(3, M, N // 3, dtype=torch.bfloat16, device=dev)
        ops.gemm_fp8(a8, sa, w8, sw, out[0], bias=None if bias is None else bias.to(dev), split=(N // 3, M * (N // 3)))
        got = out.permute(1, 0, 2).reshape(M, N)
    else:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ops.gemm_fp8(a8, sa, w8, sw, out, bias=None if bias is None else bias.to(dev), act=kw.get("act"), **args)
        got = out
    err = rel_fro(got.float().cpu(), ref.float().to(torch.bfloat16).float())
    print(f"{M}x{N}x{K} {kw}: rel-Fro vs bf16(exact) = {err:.3e}")
    assert err <= 1e-3
    # and the quantisation itself costs what e4m3 costs: a few percent against the unquantised bf16 product
    if not kw:
        full = a.double() @ w.double().T
        print(f"   vs the unquantised product: {rel_fro(got.float().cpu(), full.float()):.3e}")
        assert rel_fro(got.float().cpu(), full.float()) < 6e-2


def test_gemm_fp8_persistent_kernel_equals_the_128_tile_kernel(dev, monkeypatch):
    """The two e4m3 GEMM kernels differ only in how they walk K and the output tiles: same bytes in, same bf16 out up to
    the order of fp32 additions inside a K-tile (a last-bit flip on a handful of elements of a long K)."""
    from bind_your_avatar_implementation_amd import ops
    M, N, K = 4500, 3200, 2048
    a, w = rnd((M, K), 12), rnd((N, K), 13, std=K ** -0.5)
    bias, res = rnd((N,), 14).to(dev), rnd((M, N), 15).to(dev)
    a8, sa = ops.quantize_rows_fp8(a.to(dev))
    w8, sw = ops.quantize_rows_fp8(w.to(dev))
    outs = {}
    for kern in ("128", "v4"):
        outs[kern] = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        with ops.options(fp8_kernel=int(kern == "128")):
            ops.gemm_fp8(a8, sa, w8, sw, outs[kern], bias=bias, res=res)
    diff = (outs["v4"].float() - outs["128"].float()).abs()
    frac = float((diff > 0).float().mean())
    print(f"elements that differ: {frac:.2e}, max abs {float(diff.max()):.3e}")
    assert torch.isfinite(outs["v4"].float()).all() and frac < 1e-4 and float(diff.max()) <= 0.0625


def test_fp8_linear_selection(dev, monkeypatch):
    """enable_fp8_weights(linears=...): the default is the four DiT Linears, "all" adds the two query projections, an unknown
    name is refused when the engine packs; BYA_FP8_LINEARS is the same choice for a caller that cannot reach the model
    object (a comma list, read when the engine packs; an explicit ``linears=`` wins)."""
    from bind_your_avatar_implementation_amd import BindyouravatarTransformer3DModel
    from bind_your_avatar_implementation_amd.synth import synth_inputs
    from test_forward_gpu import SMALL_KW, to_dev
    model = BindyouravatarTransformer3DModel(**SMALL_KW, device=dev).init_synthetic(seed=2, fast=True)
    gi = to_dev(synth_inputs(batch=1, frames=3, height=16, width=24, seed=3), dev)
    monkeypatch.setenv("BYA_FP8_LINEARS", "ff2,pq")
    model.enable_fp8_weights()
    model(**gi)
    assert set(model._engine.w8) == {"ff2", "pq"}
    monkeypatch.delenv("BYA_FP8_LINEARS")
    model.enable_fp8_weights()
    model(**gi)
    assert set(model._engine.w8) == {"qkv", "out", "ff1", "ff2"}
    model.enable_fp8_weights(linears=("ff1", "aq"))
    model(**gi)
    assert set(model._engine.w8) == {"ff1", "aq"}
    model.enable_fp8_weights(linears="all")
    model(**gi)
    assert set(model._engine.w8) == {"qkv", "out", "ff1", "ff2", "pq", "aq"}
    model.enable_fp8_weights(linears=("qkv", "nope"))
    with pytest.raises(ValueError):
        model(**gi)


# ------------------------------------------------------------------------------------------ forward level
class FakeQuantLinear(torch.nn.Module):
    """The definition of bya_gemm_fp8 applied to one nn.Linear of the CPU oracle: e4m3 rows of x, e4m3 rows of W,
    exact products, fp32 accumulation and scales, then bias -- whatever dtype the oracle is run in."""

    def __init__(self, lin):
        super().__init__()
        self.lin = lin

    def forward(self, x):
        xq, sx = quant_ref(x)
        wq, sw = quant_ref(self.lin.weight)
        y = (xq.view(torch.float8_e4m3fn).float() @ wq.view(torch.float8_e4m3fn).float().T) * sx[..., None] * sw
        if self.lin.bias is not None:
            y = y + self.lin.bias.float()
        return y.to(x.dtype)


def with_fp8_dit_linears(orc, query_projections=False):
    """The engine's default set (engine.FP8_DEFAULT: the four DiT Linears); query_projections=True = linears="all"."""
    for blk in orc.transformer_blocks:
        at = blk.attn1
        at.to_q, at.to_k, at.to_v = FakeQuantLinear(at.to_q), FakeQuantLinear(at.to_k), FakeQuantLinear(at.to_v)
        at.to_out[0] = FakeQuantLinear(at.to_out[0])
        blk.ff.net[0].proj = FakeQuantLinear(blk.ff.net[0].proj)
        blk.ff.net[2] = FakeQuantLinear(blk.ff.net[2])
    if not query_projections:
        return orc
    for pc in orc.perceiver_cross_attention:                 # the query projections behind a LayerNorm of the video rows
        pc.to_q = FakeQuantLinear(pc.to_q)
    for layer in orc.audio_model.layers:
        layer["attn"].to_q = FakeQuantLinear(layer["attn"].to_q)
    return orc


def test_forward_with_fp8_weights_vs_fake_quantised_oracle(dev):
    """Small geometry (3 x 8 x 12 video tokens + 226 text rows, full 3072-wide model, 2 layers, 2 identities, CFG batch of
    2): the engine with fp8 weights -- all six kinds, linears="all" -- against the CPU oracle whose DiT Linears and perceiver / audio
    query projections are replaced by the fp8 definition above.  Bar, stage by stage, as for the bf16 engine: err(engine, fp32 oracle) <= 1.5 x err(oracle run in
    bf16, fp32 oracle) + 1e-3 (both carry the same e4m3 operands; what differs is bf16 rounding around them -- which also
    moves a few e4m3 roundings by one step, on both sides).  The distance to the unquantised model is printed and bounded
    loosely: that is the price of e4m3, not an implementation property."""
    from bind_your_avatar_implementation_amd import BindyouravatarTransformer3DModel
    from bind_your_avatar_implementation_amd.synth import synth_inputs
    from oracle.model import OracleTransformer
    from test_forward_gpu import SMALL_KW, to_dev
    model = BindyouravatarTransformer3DModel(**SMALL_KW, device=dev).init_synthetic(seed=1, fast=True)
    sd = {k: v.float().cpu() for k, v in model.state_dict().items()}
    with torch.device("meta"):
        orc = OracleTransformer(**SMALL_KW)
    orc = orc.to_empty(device="cpu")
    orc.load_state_dict(sd, strict=True)
    orc.eval()
    inp = synth_inputs(batch=2, frames=3, height=16, width=24, seed=3, uncond_first=True)
    gi = to_dev(inp, dev)
    out_bf16 = model(**gi)[0].float().cpu()
    taps32, taps16, tapsg = {}, {}, {}
    with torch.no_grad():
        plain = orc(**inp)[0]
        orc = with_fp8_dit_linears(orc, query_projections=True)
        ref = orc(taps=taps32, **inp)[0]
        orc16 = orc.to(torch.bfloat16)
        inp16 = {k: (v.to(torch.bfloat16) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in inp.items()}
        inp16["id_cond"] = [t.to(torch.bfloat16) for t in inp["id_cond"]]
        inp16["id_vit_hidden"] = [[t.to(torch.bfloat16) for t in l] for l in inp["id_vit_hidden"]]
        ref16 = orc16(taps=taps16, **inp16)[0]
    model.enable_fp8_weights(linears="all")                  # all six kinds (the default leaves the query projections in bf16)
    out = model(**gi)[0]
    assert model._engine.w8 is not None and set(model._engine.w8) == {"qkv", "out", "ff1", "ff2", "pq", "aq"}
    model._engine.step(gi["hidden_states"], gi["encoder_hidden_states"], gi["timestep"], gi["image_rotary_emb"],
                       gi["id_cond"], gi["id_vit_hidden"], gi["audio_embeds"], gi["af_matrix"], None, taps=tapsg)
    for name in ["block0", "face0", "audio0", "block1", "audio1"]:
        g, r32, r16 = tapsg[name].float().cpu(), taps32[name].float(), taps16[name].float()
        e_g, e_16 = rel_fro(g, r32), rel_fro(r16, r32)
        print(f"{name:8s} engine(fp8)-vs-fp32(fp8) {e_g:.3e}   bf16(fp8)-oracle-vs-fp32(fp8) {e_16:.3e}")
        assert e_g <= 1.5 * e_16 + 1e-3, name
    e_g, e_16 = rel_fro(out, ref), rel_fro(ref16, ref)
    print(f"output   engine(fp8)-vs-fp32(fp8) {e_g:.3e}   bf16(fp8)-oracle-vs-fp32(fp8) {e_16:.3e}")
    assert e_g <= 1.5 * e_16 + 1e-3
    d_model, d_engine = rel_fro(ref, plain), rel_fro(out.float().cpu(), out_bf16)
    print(f"price of e4m3 on this model: oracle {d_model:.3e}, engine fp8 vs engine bf16 {d_engine:.3e}")
    assert d_engine < 0.15 and abs(d_engine - d_model) < 0.5 * d_model + 5e-3
    model.enable_fp8_weights(False)
    assert torch.equal(model(**gi)[0].float().cpu(), out_bf16)            # and back: bit-identical bf16 engine


def test_config4_fp8_weights_three_identities_97_frames_vs_fake_quantised_oracle(dev):
    """BASELINE configs[4] with ALL THREE of its elements in one forward: fp8 weights + 3 identities / audio streams + a
    97-frame clip (25 latent frames), at a small spatial size (25 x 6 x 10 video tokens + 226 text rows, full 3072-wide
    model, 2 layers, cyclic audio-to-face matrix).  Oracle: oracle/model.py (whose n-identity audio weights are the
    build-defined generalisation of the reference's two-stream swap) with its four DiT Linears (the
    engine's default fp8 set) replaced by the fp8 definition of include/bya.h -- no reference counterpart exists for either
    (models/transformer.py:638-639,784,881 hard-code two identities; nothing in the reference is fp8).  Usual stage bar:
    err(engine, fp32 oracle) <= 1.5 x err(oracle in bf16, fp32 oracle) + 1e-3."""
    from bind_your_avatar_implementation_amd import BindyouravatarTransformer3DModel
    from bind_your_avatar_implementation_amd.synth import synth_inputs
    from oracle.model import OracleTransformer
    from test_forward_gpu import SMALL_KW, to_dev
    kw = dict(SMALL_KW, sample_frames=97, sample_height=12, sample_width=20)
    model = BindyouravatarTransformer3DModel(**kw, device=dev).init_synthetic(seed=11, fast=True)
    inp = synth_inputs(batch=1, frames=25, height=12, width=20, n_id=3, seed=12)
    inp["af_matrix"] = torch.roll(torch.eye(3), 1, dims=1)[None]
    with torch.device("meta"):
        orc = OracleTransformer(**kw)
    orc = orc.to_empty(device="cpu")
    orc.load_state_dict({k: v.float().cpu() for k, v in model.state_dict().items()}, strict=True)
    orc.eval()
    gi = to_dev(inp, dev)
    out_bf16 = model(**gi)[0].float().cpu()
    taps32, taps16, tapsg = {}, {}, {}
    with torch.no_grad():
        orc = with_fp8_dit_linears(orc)
        ref = orc(taps=taps32, **inp)[0]
        inp16 = {k: (v.to(torch.bfloat16) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in inp.items()}
        inp16["id_cond"] = [t.to(torch.bfloat16) for t in inp["id_cond"]]
        inp16["id_vit_hidden"] = [[t.to(torch.bfloat16) for t in l] for l in inp["id_vit_hidden"]]
        ref16 = orc.to(torch.bfloat16)(taps=taps16, **inp16)[0]
    model.enable_fp8_weights()
    out = model(**gi)[0]
    eng = model._engine
    assert eng.w8 is not None and set(eng.w8) == {"qkv", "out", "ff1", "ff2"} and eng.n_id == 3
    eng.step(gi["hidden_states"], gi["encoder_hidden_states"], gi["timestep"], gi["image_rotary_emb"], gi["id_cond"],
             gi["id_vit_hidden"], gi["audio_embeds"], gi["af_matrix"], None, taps=tapsg)
    assert tapsg["router0_b0"].shape[-1] == 3
    for name in ["block0", "face0", "audio0", "block1", "audio1"]:
        g, r32, r16 = tapsg[name].float().cpu(), taps32[name].float(), taps16[name].float()
        e_g, e_16 = rel_fro(g, r32), rel_fro(r16, r32)
        print(f"{name:8s} engine(fp8, 3 ids, 97 frames)-vs-fp32 {e_g:.3e}   bf16-oracle-vs-fp32 {e_16:.3e}")
        assert e_g <= 1.5 * e_16 + 1e-3, name
    e_g, e_16 = rel_fro(out, ref), rel_fro(ref16, ref)
    print(f"output   engine(fp8, 3 ids, 97 frames)-vs-fp32 {e_g:.3e}   bf16-oracle-vs-fp32 {e_16:.3e}")
    assert e_g <= 1.5 * e_16 + 1e-3
    assert 1e-3 < rel_fro(out.float().cpu(), out_bf16) < 0.15            # fp8 weights were really in use
    model.enable_fp8_weights(False)


def test_fp8_weights_sequence_parallel_two_ranks_matches_single(dev):
    """fp8 weights under the sharded engine (row shards, head-parallel exchange): the activation scales are per ROW, so a
    rank's rows quantise exactly as they do unsharded -- 2 ranks on the one test GPU must reproduce the single-GPU fp8
    step to the same bar as the bf16 engine's sharded test (the GEMMs see different M, hence tile raggedness, not
    different arithmetic)."""
    import os
    import torch.multiprocessing as mp
    from test_forward_gpu import _sp_worker
    os.environ["BYA_FP8_WEIGHTS"] = "1"            # inherited by the spawned ranks: every engine they build holds fp8 weights
    try:
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(_sp_worker, args=(2, 28100 + os.getpid() % 1000, ret, (16, 24), 2), nprocs=2, join=True)
    finally:
        os.environ.pop("BYA_FP8_WEIGHTS", None)
    print("fp8, sequence-parallel vs single:", dict(ret))
    for r in (0, 1):
        assert ret[r][0] <= 2e-2, ret[r]


def test_layernorm_fp8_equals_layernorm_then_quantiser(dev):
    """bya_layernorm_fp8 is bya_layernorm + bya_quantize_rows_fp8 in one pass: same bytes, same scales (AdaLN modulation with
    the text / video split, a CFG batch of 2, rows that are not a multiple of the 4 rows of a workgroup)."""
    from bind_your_avatar_implementation_amd import ops
    B, S, D, split = 2, 1001, 3072, 226
    x = rnd((B, S, D), 11, std=2.0).to(dev)
    w, b = rnd((D,), 12).to(dev), rnd((D,), 13, std=0.1).to(dev)
    mod = rnd((B, 4 * D), 14, std=0.3).to(dev)
    kw = dict(eps=1e-5, shift0=mod[:, :D], scale0=mod[:, D:], shift1=mod[:, 2 * D:], scale1=mod[:, 3 * D:], split=split,
              mod_batch_stride=mod.stride(0))
    y = torch.empty_like(x)
    ops.layernorm(x, y, w, b, **kw)
    q_ref, s_ref = ops.quantize_rows_fp8(y)
    q = torch.empty(B, S, D, dtype=torch.uint8, device=dev)
    s = torch.empty(B, S, dtype=torch.float32, device=dev)
    ops.layernorm_fp8(x, q, s, w, b, **kw)
    assert torch.equal(s, s_ref)
    assert torch.equal(q, q_ref)


def test_fused_and_unfused_fp8_engine_are_bit_identical(dev, monkeypatch):
    from bind_your_avatar_implementation_amd import BindyouravatarTransformer3DModel
    from bind_your_avatar_implementation_amd.synth import synth_inputs
    from test_forward_gpu import SMALL_KW, to_dev
    model = BindyouravatarTransformer3DModel(**SMALL_KW, device=dev).init_synthetic(seed=1, fast=True).enable_fp8_weights(linears="all")
    gi = to_dev(synth_inputs(batch=2, frames=3, height=16, width=24, seed=3, uncond_first=True), dev)
    fused = model(**gi)[0].clone()
    assert model._engine.fuse_ln_quant
    monkeypatch.setenv("BYA_FP8_FUSED_LN", "0")
    model.invalidate_engine()
    plain = model(**gi)[0]
    assert not model._engine.fuse_ln_quant
    assert torch.equal(fused, plain)


def test_fp8_engine_hip_graph_replay_matches_eager(dev):
    """The fp8 engine inside a captured hipGraph (its e4m3 staging buffers and row scales live in the graph's workspace):
    replays on new input values equal eager execution bit for bit."""
    from bind_your_avatar_implementation_amd import BindyouravatarTransformer3DModel
    from bind_your_avatar_implementation_amd.synth import synth_inputs
    from test_forward_gpu import SMALL_KW, to_dev
    model = BindyouravatarTransformer3DModel(**SMALL_KW, device=dev).init_synthetic(seed=1, fast=True).enable_fp8_weights(linears="all")
    outs = {}
    for mode in ("eager", "graph"):
        model.use_hip_graph = mode == "graph"
        for seed in (21, 22):
            inp = to_dev(synth_inputs(batch=2, frames=3, height=16, width=24, seed=seed, uncond_first=True), dev)
            inp["timestep"] = torch.tensor([seed * 10, seed * 10], dtype=torch.int64, device=dev)
            outs[mode, seed] = model(**inp)[0].clone()
    model.use_hip_graph = False
    assert model._engine.w8 is not None and len(model._graphs) == 1
    for seed in (21, 22):
        assert torch.equal(outs["eager", seed], outs["graph", seed]), seed
    assert not torch.equal(outs["graph", 21], outs["graph", 22])

