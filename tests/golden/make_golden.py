#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Runs only in the build container (needs /root/reference, which never travels to the GPU box;
nothing here is imported by the test-suite at run time).  What it does:

  1. imports the reference's own ``models/transformer.py``, ``models/router.py`` and
     ``models/audio_model.py`` unmodified.  Their third-party dependency ``diffusers`` is not
     installed, so a stand-in namespace is registered whose layer classes are the build's
     restatement in ``oracle/layers.py`` (=> the fixtures pin the reference-OWNED arithmetic and
     control flow; the diffusers-owned layers stay "parity unpinned", see oracle/__init__.py).
  2. fills the reference model with the name-keyed synthetic weights of
     ``bind_your_avatar_implementation_amd.synth`` (bf16-representable values, fp32 storage) and
     runs ``forward`` in fp32 on CPU at the reference's hard-coded geometry (13 x 30 x 45 tokens).
  3. stores small fixtures (router logits, strided taps of the hidden stream, the full output in
     fp16, the state-dict key/shape list) as ``.npz``/``.json`` next to this script.
  4. runs ``oracle.model.OracleTransformer`` on the same weights/inputs and prints the deviation,
     which ``tests/test_oracle_golden.py`` re-checks from the fixtures.

Usage:  python tests/golden/make_golden.py [--case base|cfg_forcing|modules|masks|depth|config0|config0_mono|prepare_latents|bars] [--layers 2]

``depth``   : the full 42-layer model at the reference geometry (fp32 reference + the oracle run in bf16 on the same
              weights: the fixture carries the reference output, strided per-block taps and the bf16 path's own error,
              which is the bar of tests/test_forward_gpu.py::test_depth_42_layers_vs_golden).
``config0`` : BASELINE.json configs[0] -- ONE DiT block with every injection (cross_attn_interval = 1), 1 face + 1 audio
              stream expressed as the 2-stream path with the second identity / audio stream zero-filled
              ("Single-audio fallback").
``bars``    : error of the oracle run in bf16 against the stored fp32 reference outputs (tests/golden/bf16_bars.json).
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import layers as L  # noqa: E402
from oracle.model import OracleTransformer  # noqa: E402
from bind_your_avatar_implementation_amd.synth import synth_inputs, synth_state_dict  # noqa: E402


def install_standins():
    """Register the ``diffusers`` names the reference imports."""
    import transformers  # noqa: F401  (must be imported before the torchvision stub exists)
    from transformers import T5EncoderModel, T5Tokenizer  # noqa: F401

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class ConfigMixin:
        pass

    def register_to_config(init):
        def wrapped(self, *a, **kw):
            import inspect
            sig = inspect.signature(init)
            bound = sig.bind(self, *a, **kw)
            bound.apply_defaults()
            cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
            self.config = types.SimpleNamespace(**cfg)
            init(self, *a, **kw)
        return wrapped

    class ModelMixin(torch.nn.Module):
        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

        @property
        def dtype(self):
            try:
                return next(self.parameters()).dtype
            except StopIteration:
                return torch.get_default_dtype()

    class PeftAdapterMixin:
        pass

    class _Logger:
        def warning(self, *a, **k):
            pass
        info = debug = warning

    logging = types.SimpleNamespace(get_logger=lambda name: _Logger())
    mod("diffusers", ModelMixin=ModelMixin)
    mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    mod("diffusers.loaders", PeftAdapterMixin=PeftAdapterMixin)
    mod("diffusers.utils", USE_PEFT_BACKEND=False, is_torch_version=lambda *a: True, logging=logging,
        scale_lora_layers=lambda *a, **k: None, unscale_lora_layers=lambda *a, **k: None,
        load_image=lambda *a, **k: None)
    mod("diffusers.utils.torch_utils", maybe_allow_in_graph=lambda c: c)
    mod("diffusers.models")
    mod("diffusers.models.attention", Attention=L.Attention, FeedForward=L.FeedForward)
    mod("diffusers.models.attention_processor", AttentionProcessor=L.AttentionProcessor,
        CogVideoXAttnProcessor2_0=L.CogVideoXAttnProcessor2_0,
        FusedCogVideoXAttnProcessor2_0=L.FusedCogVideoXAttnProcessor2_0)
    mod("diffusers.models.embeddings", CogVideoXPatchEmbed=L.CogVideoXPatchEmbed,
        TimestepEmbedding=L.TimestepEmbedding, Timesteps=L.Timesteps,
        get_3d_rotary_pos_embed=L.get_3d_rotary_pos_embed)
    mod("diffusers.models.modeling_outputs", Transformer2DModelOutput=dict)
    mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    mod("diffusers.models.normalization", AdaLayerNorm=L.AdaLayerNorm,
        CogVideoXLayerNormZero=L.CogVideoXLayerNormZero)
    mod("diffusers.pipelines")
    mod("diffusers.pipelines.cogvideo")
    mod("diffusers.pipelines.cogvideo.pipeline_cogvideox",
        get_resize_crop_region_for_grid=L.get_resize_crop_region_for_grid)
    # heavy CV imports of models/utils.py that the hot path never touches
    mod("cv2")
    tv = mod("torchvision")
    tvt = mod("torchvision.transforms", InterpolationMode=object)
    tvf = mod("torchvision.transforms.functional", normalize=None, resize=None)
    tv.transforms = tvt
    tvt.functional = tvf
    sys.path.insert(0, REF)


MODEL_KW = dict(num_attention_heads=48, attention_head_dim=64, in_channels=48, out_channels=16,
                use_rotary_positional_embeddings=True, use_learned_positional_embeddings=True,
                is_train_face=True, cross_attn_interval=2, local_face_scale=1.0, is_train_audio=True,
                audio_attn_interval=1)


def build(cls, layers, seed, **over):
    t0 = time.time()
    with torch.device("meta"):
        m = cls(num_layers=layers, **dict(MODEL_KW, **over))
    pos = None
    m = m.to_empty(device="cpu")
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    fill(m, seed)
    m.eval()
    print(f"built {cls.__name__} ({sum(p.numel() for p in m.parameters())/1e9:.2f} B params) "
          f"in {time.time()-t0:.0f}s", flush=True)
    return m, shapes


def fill(m, seed, dtype=None, **kw):
    """Name-keyed synthetic parameters, written tensor by tensor (a 42-layer fp32 model is 34 GB: a second full
    state dict would not fit next to it)."""
    from oracle.model import router_pos_emb
    from bind_your_avatar_implementation_amd.synth import synth_tensor
    with torch.no_grad():
        for name, t in m.state_dict().items():
            if name == "router.pos_emb":
                v = router_pos_emb(13, 45, 30, 512)          # models/router.py:312-316 constants
            else:
                v = synth_tensor(name, t.shape, seed)
                if name.endswith("pos_embedding"):
                    v[:, :226] = 0
            t.copy_(v.to(t.dtype))
            del v


class StridedTaps(dict):
    """taps dict that keeps a strided sample of every large tensor instead of the tensor (42 layers of taps do not
    fit in host memory otherwise)."""

    def __init__(self, step):
        super().__init__()
        self.step = step

    def __setitem__(self, k, v):
        if torch.is_tensor(v) and k.startswith("block"):
            v = v[:, 226:]                  # the oracle taps the joint [text | video] stream; keep the video rows
        if torch.is_tensor(v) and v.numel() > 4_000_000:
            v = torch.from_numpy(strided(v, self.step))
        super().__setitem__(k, v)


def strided(t, step=970):
    return t.detach().float().reshape(-1)[::step].numpy().copy()


def stats(t):
    t = t.detach().double()
    return np.array([t.mean().item(), t.std().item(), t.abs().max().item(), t.norm().item()])


def run_case(case, layers, seed):
    install_standins()
    from models.transformer import BindyouravatarTransformer3DModel
    torch.manual_seed(0)
    ref, shapes = build(BindyouravatarTransformer3DModel, layers, seed)
    batch = 2 if case == "cfg_forcing" else 1
    inp = synth_inputs(batch=batch, seed=seed, uncond_first=(batch == 2))
    forcing = None
    if case == "cfg_forcing":
        # hard 0/1 mask in the layout of util/utils.py:871-936 (label per token -> one-hot, background (0,0)):
        # id 0 = a box drifting right over time on the left, id 1 = a box on the right present in SOME frames
        # only, so the max-over-frames of models/transformer.py:813-819 changes the per-frame masks.
        lab = torch.full((13, 30, 45), -1, dtype=torch.long)
        for t in range(13):
            lab[t, 4:22, 2 + t:14 + t] = 0
            if t % 3 != 1:
                lab[t, 8 + (t % 4):27, 28:43 - (t % 5)] = 1
        lab = lab.reshape(-1)
        forcing = torch.zeros(1, 13 * 30 * 45, 2)
        forcing[0, lab == 0, 0] = 1
        forcing[0, lab == 1, 1] = 1
        inp["af_matrix"] = (1 - torch.eye(2))[None].repeat(batch, 1, 1)
        inp["routing_logits_forcing"] = forcing

    taps = {}

    def hook(name):
        def fn(mod, args, out):
            outs = out if isinstance(out, (tuple, list)) else (out,)
            for k, o in enumerate(outs):
                if torch.is_tensor(o):
                    taps.setdefault(f"{name}.{k}", []).append(o.detach().clone())
        return fn

    for i, blk in enumerate(ref.transformer_blocks):
        blk.register_forward_hook(hook(f"block{i}"))
    ref.router.register_forward_hook(hook("router"))
    for i, pc in enumerate(ref.perceiver_cross_attention):
        pc.register_forward_hook(hook(f"perceiver{i}"))
    ref.audio_model.register_forward_hook(hook("audio"))
    ref.local_facial_extractor.register_forward_hook(hook("lfe"))
    ref.audio_model.audio_proj_model.register_forward_hook(hook("audio_proj"))

    t0 = time.time()
    with torch.no_grad():
        out = ref(return_dict=False, denoise_step=0, **inp)
    assert len(out) == 5 and all(o is None for o in out[1:])
    out = out[0]
    print(f"reference forward ({case}, {layers} layers, B={batch}) {time.time()-t0:.0f}s "
          f"out {tuple(out.shape)} |out|={out.norm():.4f}", flush=True)

    fx = {"output_f16": out.numpy().astype(np.float16), "output_stats": stats(out)}
    for name, lst in taps.items():
        for j, t in enumerate(lst):
            key = f"{name}.call{j}"
            if name.startswith("router"):
                fx[key] = t.float().numpy()                       # full [1, N, 2]
            elif name.startswith(("lfe", "audio_proj")):
                fx[key] = t.float().numpy().astype(np.float16)    # small, keep whole
            elif name.startswith("perceiver") and not name.endswith(".0"):
                continue                                           # weight/q_out/k_out: covered by router
            else:
                fx[key + ".strided"] = strided(t)
                fx[key + ".stats"] = stats(t)
    if forcing is not None:
        fx["forcing_u8"] = forcing.numpy().astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, f"ref_forward_{case}_L{layers}_seed{seed}.npz"), **fx)
    if case == "base":
        with open(os.path.join(HERE, "ref_state_dict_keys.json"), "w") as f:
            json.dump({"layers": layers, "model_kw": MODEL_KW, "keys": shapes}, f)

    # --- the restatement against the reference, same weights / inputs
    sd = ref.state_dict()
    del ref
    with torch.device("meta"):
        orc = OracleTransformer(num_layers=layers, **MODEL_KW)
    orc = orc.to_empty(device="cpu")
    orc.load_state_dict(sd, strict=True)
    orc.eval()
    otaps = {}
    t0 = time.time()
    o = orc(taps=otaps, **inp)[0]
    rel = ((o - out).norm() / out.norm()).item()
    print(f"oracle forward {time.time()-t0:.0f}s  rel-Fro(oracle, reference) = {rel:.3e} "
          f"max-abs {(o - out).abs().max().item():.3e}", flush=True)
    for j, t in enumerate(taps.get("router.0", [])):
        ca, bj = divmod(j, batch)
        d = (otaps[f"router{ca}_b{bj}"] - t).abs().max().item()
        print(f"  router call {j}: max-abs diff {d:.3e}")
    assert rel < 1e-5, rel


def to_bf16_inputs(inp):
    out = {k: (v.to(torch.bfloat16) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in inp.items()}
    out["id_cond"] = [t.to(torch.bfloat16) for t in inp["id_cond"]]
    out["id_vit_hidden"] = [[t.to(torch.bfloat16) for t in l] for l in inp["id_vit_hidden"]]
    out["image_rotary_emb"] = inp["image_rotary_emb"]            # fp32 tables, as the pipeline passes them
    return out


def rel(a, b):
    a, b = torch.as_tensor(a).double().reshape(-1), torch.as_tensor(b).double().reshape(-1)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def bf16_oracle_pass(layers, seed, inp, step, router_tap=None, prep=None, **over):
    """The restatement with bf16 parameters and activations (what the reference's own bf16 inference path computes,
    infer.py:477) on the same synthetic weights: its distance from the fp32 result is the tolerance bar."""
    with torch.device("meta"):
        orc = OracleTransformer(num_layers=layers, **dict(MODEL_KW, **over))
    orc = orc.to_empty(device="cpu").to(torch.bfloat16)
    fill(orc, seed)
    orc.eval()
    if prep is not None:
        prep(orc)
    taps = StridedTaps(step)
    if router_tap is not None:
        orc.router.register_forward_hook(lambda m, a, o: router_tap.append(o.detach().float().clone()))
    t0 = time.time()
    with torch.no_grad():
        out = orc(taps=taps, **to_bf16_inputs(inp))[0]
    print(f"bf16 oracle forward ({layers} layers) {time.time()-t0:.0f}s", flush=True)
    return out.float(), taps


def run_deep(case, layers, seed, step, over, inp, prep=None):
    """fp32 REFERENCE + bf16 oracle on one configuration; strided taps only (memory).  ``prep(model)`` is applied to
    the reference and to the oracle after their weights are in place."""
    install_standins()
    from models.transformer import BindyouravatarTransformer3DModel
    ref, _ = build(BindyouravatarTransformer3DModel, layers, seed, **over)
    if prep is not None:
        prep(ref)
    taps = {}

    def hook(name):
        def fn(mod, args, out):
            outs = out if isinstance(out, (tuple, list)) else (out,)
            taps[name] = strided(outs[0], step)
        return fn

    for i, blk in enumerate(ref.transformer_blocks):
        blk.register_forward_hook(hook(f"block{i}"))          # output 0 = the video rows of the hidden stream
    rtaps = []
    ref.router.register_forward_hook(lambda m, a, o: rtaps.append(o.detach().float().clone()))
    t0 = time.time()
    with torch.no_grad():
        out = ref(return_dict=False, denoise_step=0, **inp)[0]
    print(f"reference forward ({case}, {layers} layers) {time.time()-t0:.0f}s |out|={out.norm():.4f}", flush=True)
    del ref
    import gc
    gc.collect()
    fx = {"output_f16": out.numpy().astype(np.float16), "output_stats": stats(out), "tap_step": np.array(step)}
    for k, v in taps.items():
        fx[k + ".strided"] = v
    for j, r in enumerate(rtaps[:2]):
        fx[f"router.call{j}"] = r.numpy()
    out16, taps16 = bf16_oracle_pass(layers, seed, inp, step, prep=prep, **over)
    fx["bf16_err_output"] = np.array(rel(out16, out))
    errs = []
    for i in range(layers):
        t16 = taps16[f"block{i}"]
        if t16.numel() != taps[f"block{i}"].size:
            raise RuntimeError("tap layouts differ")
        errs.append(rel(t16, taps[f"block{i}"]))
    fx["bf16_err_blocks"] = np.array(errs)
    print(f"[{case}] bf16-oracle vs fp32 reference: output {fx['bf16_err_output']:.3e}; blocks "
          + " ".join(f"{e:.2e}" for e in errs), flush=True)
    np.savez_compressed(os.path.join(HERE, f"ref_forward_{case}_L{layers}_seed{seed}.npz"), **fx)


def install_pipeline_standins():
    """The extra ``diffusers`` names models/pipeline_bindyouravatar.py imports at module level (none of them is reached by
    ``prepare_latents`` except ``randn_tensor``, restated in oracle/vae.py)."""
    from oracle import vae as OV

    def mod(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Any:
        def __init__(self, *a, **k):
            pass

    mod("diffusers.callbacks", MultiPipelineCallbacks=_Any, PipelineCallback=_Any)
    mod("diffusers.image_processor", PipelineImageInput=object)
    mod("diffusers.models", AutoencoderKLCogVideoX=_Any, CogVideoXTransformer3DModel=_Any)
    mod("diffusers.pipelines.pipeline_utils", DiffusionPipeline=_Any)
    mod("diffusers.schedulers", CogVideoXDDIMScheduler=_Any, CogVideoXDPMScheduler=_Any)
    sys.modules["diffusers.utils"].replace_example_docstring = lambda doc: (lambda f: f)
    mod("diffusers.utils.torch_utils", randn_tensor=OV.randn_tensor)
    mod("diffusers.video_processor", VideoProcessor=_Any)
    mod("diffusers.pipelines.cogvideo.pipeline_output", CogVideoXPipelineOutput=_Any)
    sys.modules["diffusers.loaders"].CogVideoXLoraLoaderMixin = type("CogVideoXLoraLoaderMixin", (), {})


def run_prepare_latents():
    """The REFERENCE's ``BindyouravatarPipeline.prepare_latents`` (models/pipeline_bindyouravatar.py:376-458), called the way
    ``__call__`` calls it (:831-860: once for the image -- which draws the noise when the caller gave none -- and once
    more for the background image with the latents of the first call), on the stub VAE of tests/golden/stub_vae.py."""
    install_standins()
    install_pipeline_standins()
    from models import pipeline_bindyouravatar as P
    from oracle.vae import DiagonalGaussianDistribution
    sys.path.insert(0, HERE)
    from stub_vae import StubVAE, prepare_latents_cases, prepare_latents_inputs
    fx = {}
    for name, dtype, batch, kps, bg, gk in prepare_latents_cases():
        vae = StubVAE(DiagonalGaussianDistribution)
        me = types.SimpleNamespace(vae=vae, vae_scale_factor_temporal=4, vae_scale_factor_spatial=8,
                                   vae_scaling_factor_image=vae.config.scaling_factor,
                                   scheduler=types.SimpleNamespace(init_noise_sigma=1.0))
        inp = prepare_latents_inputs(name, dtype, batch)
        gen = ([torch.Generator().manual_seed(100 + i) for i in range(batch)] if gk == "list"
               else torch.Generator().manual_seed(100))
        given = inp["latents"] if "given_latents" in name else None
        kp = inp["kps"] if kps else None
        lat, img = P.BindyouravatarPipeline.prepare_latents(me, inp["image"], batch, 16, 9, 32, 48, dtype, torch.device("cpu"),
                                                            gen, given, kp)
        fx[name + ".latents"], fx[name + ".image_latents"] = lat.float().numpy(), img.float().numpy()
        if bg:
            lat2, bgl = P.BindyouravatarPipeline.prepare_latents(me, inp["bg"], batch, 16, 9, 32, 48, dtype,
                                                                 torch.device("cpu"), gen, lat, kp)
            assert torch.equal(lat2, lat)
            fx[name + ".image_bg_latents"] = bgl.float().numpy()
        print(name, tuple(lat.shape), tuple(img.shape), "vae.encode calls:", vae.calls)
        fx[name + ".encode_calls"] = np.array(vae.calls)
    np.savez_compressed(os.path.join(HERE, "ref_prepare_latents.npz"), **fx)


def config0_inputs(seed):
    """BASELINE.json configs[0]: 1 audio + 1 face stream = the 2-stream call with the second identity's
    ``id_cond`` / ``id_vit_hidden`` and the second audio stream zero-filled."""
    inp = synth_inputs(batch=1, seed=seed)
    inp["id_cond"][1] = torch.zeros_like(inp["id_cond"][1])
    inp["id_vit_hidden"][1] = [torch.zeros_like(t) for t in inp["id_vit_hidden"][1]]
    inp["audio_embeds"][:, 1] = 0
    return inp


def run_config0_mono(seed):
    """BASELINE.json configs[0] as the reference itself takes it: ONE audio stream ([B, F, 12, 768]); the reference
    completes it with the "mute" stream it reads from tests/input/ae_mute.pt relative to the working directory
    (models/audio_model.py:201-221).  That file is not shipped: a seeded stand-in is written into a scratch directory and
    the reference runs from there, so its own torch.load finds it."""
    import tempfile
    from bind_your_avatar_implementation_amd.synth import mono_audio_extras
    ae_mute, tokens = mono_audio_extras(seed)
    inp = config0_inputs(seed)
    inp["audio_embeds"] = inp["audio_embeds"][:, 0].contiguous()
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "tests", "input"))
    torch.save(ae_mute, os.path.join(tmp, "tests", "input", "ae_mute.pt"))

    def prep(model):
        am = model.audio_model
        am.mute_learnable_tokens.data.copy_(tokens.to(am.mute_learnable_tokens.dtype))
        if hasattr(am, "mute_audio_embeds"):
            am.mute_audio_embeds = ae_mute       # the oracle takes the tensor; the reference reads the file
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        run_deep("config0_mono", 1, seed, 970, dict(cross_attn_interval=1), inp, prep=prep)
    finally:
        os.chdir(cwd)


def run_bars(seed):
    """bf16-oracle error against the STORED fp32 reference outputs of the 2-layer fixtures."""
    bars = {}
    for case, batch in (("base", 1), ("cfg_forcing", 2)):
        fxp = os.path.join(HERE, f"ref_forward_{case}_L2_seed{seed}.npz")
        fx = np.load(fxp)
        inp = synth_inputs(batch=batch, seed=seed, uncond_first=(batch == 2))
        if case == "cfg_forcing":
            inp["af_matrix"] = (1 - torch.eye(2))[None].repeat(batch, 1, 1)
            inp["routing_logits_forcing"] = torch.from_numpy(fx["forcing_u8"].astype(np.float32))
        rt = []
        out16, _ = bf16_oracle_pass(2, seed, inp, 970, router_tap=rt)
        bars[case] = rel(out16, torch.from_numpy(fx["output_f16"].astype(np.float32)))
        print(f"{case}: bf16 oracle vs stored fp32 reference output = {bars[case]:.3e}", flush=True)
        if case == "base":       # the first router call's sigmoid outputs: bf16 path vs the reference's fp32 values
            d = (rt[0].reshape(-1) - torch.from_numpy(fx["router.0.call0"]).float().reshape(-1)).abs()
            bars["base_router_logits_max_abs"], bars["base_router_logits_mean_abs"] = float(d.max()), float(d.mean())
            print(f"{case}: router logits bf16 vs fp32: max-abs {d.max():.3e} mean {d.mean():.3e}", flush=True)
    with open(os.path.join(HERE, "bf16_bars.json"), "w") as f:
        json.dump(bars, f, indent=1)


def module_cases(seed):
    """Small inputs for each reference-OWNED module; used identically here (reference) and in
    tests/test_oracle_golden.py (oracle).  Returns {case: (ctor_kwargs, input builder)}."""
    from bind_your_avatar_implementation_amd.synth import synth_tensor
    rn = lambda name, shape, std=1.0: synth_tensor("modin." + name + ".proj", shape, seed) * std * (shape[-1] ** 0.5)
    return {
        "perceiver": lambda: (rn("pc_x", (2, 32, 2048)), rn("pc_lat", (2, 270, 3072))),
        "router": lambda: (None, rn("r_q", (2, 16, 17550, 128)), rn("r_k", (2, 16, 32, 128)), 1),
        "lfe": lambda: (rn("lfe_id", (2, 1280)), [rn(f"lfe_vit{i}", (2, 577, 1024)) for i in range(5)]),
        "audio_layer": lambda: (rn("a_ctx", (2, 4, 32, 768), 0.5), rn("a_hid", (2, 4 * 35, 3072)), 4, 1),
        "block": lambda: (rn("b_hid", (1, 120, 3072)), rn("b_enc", (1, 226, 3072)), rn("b_temb", (1, 512)),
                          tuple(t for t in __import__("bind_your_avatar_implementation_amd.synth", fromlist=["x"])
                                .rope_table((2, 6, 10)))),
    }


def build_module(cls, prefix, seed, **kw):
    with torch.device("meta"):
        m = cls(**kw)
    m = m.to_empty(device="cpu")
    sd = synth_state_dict([(prefix + k, tuple(v.shape)) for k, v in m.state_dict().items()], seed=seed,
                          skip=(prefix + "pos_emb",))
    sd = {k[len(prefix):]: v for k, v in sd.items()}
    if "pos_emb" in m.state_dict():
        from oracle.model import router_pos_emb
        sd["pos_emb"] = router_pos_emb(13, 45, 30, 512)
    m.load_state_dict(sd, strict=True)
    return m.eval()


MODULE_SPECS = {   # case -> (reference module path, class name, oracle class name, prefix, ctor kwargs)
    "perceiver": ("models.router", "PerceiverCrossAttention", "PerceiverCrossAttention", "perceiver_cross_attention.0.",
                  dict(dim=3072, dim_head=128, heads=16, kv_dim=2048)),
    "router": ("models.router", "MultiIPRouter", "MultiIPRouter", "router.", dict(num_layers=2)),
    "lfe": ("models.router", "LocalFacialExtractor", "LocalFacialExtractor", "local_facial_extractor.", dict()),
    "audio_layer": ("models.audio_model", "AudioAwareModel", "AudioAwareModel", "audio_model.", dict(num_layers=2)),
    "block": ("models.transformer", "CogVideoXBlock", "CogVideoXBlock", "transformer_blocks.0.",
              dict(dim=3072, num_attention_heads=48, attention_head_dim=64, time_embed_dim=512, attention_bias=True)),
}


def run_modules(seed):
    install_standins()
    import importlib
    import oracle.model as om
    cases = module_cases(seed)
    fx = {}
    for case, (modpath, cname, oname, prefix, kw) in MODULE_SPECS.items():
        if case == "audio_layer":
            # AudioAwareModel() also builds the 1.2 B-parameter projector; only the attention layer is exercised
            # here, so the projector keeps meta/empty storage and is never called.
            pass
        t0 = time.time()
        ref = build_module(getattr(importlib.import_module(modpath), cname), prefix, seed, **kw)
        args = cases[case]()
        with torch.no_grad():
            out = ref(*args)
        outs = out if isinstance(out, (tuple, list)) else (out,)
        orc = getattr(om, oname)(**kw) if case != "audio_layer" else None
        if orc is None:
            with torch.device("meta"):
                orc = getattr(om, oname)(**kw)
            orc = orc.to_empty(device="cpu")
        orc.load_state_dict(ref.state_dict(), strict=True)
        with torch.no_grad():
            oo = orc.eval()(*args)
        oo = oo if isinstance(oo, (tuple, list)) else (oo,)
        for j, (a, b) in enumerate(zip(outs, oo)):
            d = (a - b).abs().max().item()
            print(f"{case}[{j}] {tuple(a.shape)} reference-vs-oracle max-abs {d:.3e}  ({time.time()-t0:.0f}s)", flush=True)
            assert d < 1e-5, (case, j, d)
            if a.numel() <= 2_000_000:
                fx[f"{case}.{j}"] = a.numpy().astype(np.float32 if a.numel() < 200_000 else np.float16)
            else:
                fx[f"{case}.{j}.strided"] = strided(a, 211)
                fx[f"{case}.{j}.stats"] = stats(a)
    np.savez_compressed(os.path.join(HERE, f"ref_modules_seed{seed}.npz"), **fx)


def run_masks(seed):
    """Tracking masks -> routing_logits_forcing through the reference's own util/utils.py functions (PNG frames on
    disk exactly as its stage-2 inference reads them).  Imports need stand-ins for three I/O-only packages."""
    import tempfile
    from PIL import Image
    from mask_cases import synthetic_masks
    for name in ("imageio", "spandrel"):
        m = types.ModuleType(name)
        m.ModelLoader = object
        sys.modules[name] = m
    sys.modules["diffusers.utils"].export_to_video = lambda *a, **k: None
    import importlib
    U = importlib.import_module("util.utils")
    from oracle.masks import masks_to_routing_logits
    fx = {}
    for sd in (seed, seed + 1):
        masks = synthetic_masks(sd)
        with tempfile.TemporaryDirectory() as d:
            for i in range(2):
                os.makedirs(os.path.join(d, str(i + 1)))
                for t in range(masks.shape[1]):
                    Image.fromarray(masks[i, t]).save(os.path.join(d, str(i + 1), f"annotated_frame_{t:05d}.png"))
            ref = U.process_masks_to_routing_logits(d)
        orc = masks_to_routing_logits(torch.from_numpy(masks))
        assert ref.shape == (1, 17550, 2) and torch.equal(ref, orc), "oracle differs from the reference"
        print(f"masks seed {sd}: id1 {int(ref[0, :, 0].sum())} tokens, id2 {int(ref[0, :, 1].sum())} tokens", flush=True)
        fx[f"logits.seed{sd}"] = np.packbits(ref[0].numpy().astype(np.uint8), axis=0)
    np.savez_compressed(os.path.join(HERE, f"ref_masks_seed{seed}.npz"), **fx)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="base", choices=["base", "cfg_forcing", "modules", "masks", "depth", "config0", "config0_mono", "prepare_latents",
                                                        "bars"])
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    if a.case == "masks":
        install_standins()
        run_masks(a.seed)
    elif a.case == "prepare_latents":
        run_prepare_latents()
    elif a.case == "modules":
        run_modules(a.seed)
    elif a.case == "depth":
        # the oracle taps the joint stream, the reference hook the video rows: sample the video rows on both sides
        run_deep("depth", 42 if a.layers == 2 else a.layers, a.seed, 9973, {}, synth_inputs(batch=1, seed=a.seed))
    elif a.case == "config0":
        run_deep("config0", 1, a.seed, 970, dict(cross_attn_interval=1), config0_inputs(a.seed))
    elif a.case == "config0_mono":
        run_config0_mono(a.seed)
    elif a.case == "bars":
        run_bars(a.seed)
    else:
        run_case(a.case, a.layers, a.seed)
