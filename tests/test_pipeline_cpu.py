"""CPU: host logic of the pipeline mirror -- CFG batching helpers (reference models/utils.py:630-670,
models/pipeline_bindyouravatar.py:877-884) and the DDIM scheduler algebra."""
import pytest
import torch

from bind_your_avatar_implementation_amd.pipeline import (DDIMScheduler, cfg_af_matrix, cfg_audio, cfg_id_cond,
                                                            cfg_id_vit_hidden, get_af_matrix_infer)


def test_cfg_helpers_order_and_zeroing():
    idc = [torch.ones(1, 1280), 2 * torch.ones(1, 1280)]
    out = cfg_id_cond(idc)
    assert [tuple(t.shape) for t in out] == [(2, 1280)] * 2 and torch.equal(out[1][0], out[1][1])
    out0 = cfg_id_cond(idc, zero2cond_cfg_flag=True)
    assert out0[0][0].abs().sum() == 0 and torch.equal(out0[0][1], idc[0][0])          # [uncond = 0, cond]
    vit = [[torch.randn(1, 577, 1024) for _ in range(5)] for _ in range(2)]
    v2 = cfg_id_vit_hidden(vit)
    assert len(v2) == 2 and len(v2[0]) == 5 and v2[0][0].shape == (2, 577, 1024)
    af = get_af_matrix_infer("right")[None]
    assert torch.equal(af[0], torch.tensor([[0., 1.], [1., 0.]]))
    assert torch.equal(cfg_af_matrix(af), af.repeat(2, 1, 1))
    assert cfg_af_matrix(af, True)[0].abs().sum() == 0
    a = torch.randn(1, 2, 53, 12, 768)
    ca = cfg_audio(a)
    assert ca.shape[0] == 2 and ca[0].abs().sum() == 0 and torch.equal(ca[1], a[0])   # uncond half hears silence
    with pytest.raises(ValueError):
        get_af_matrix_infer("middle")
    with pytest.raises(ValueError):
        cfg_id_cond(None)


def test_scheduler_tables_and_oracle_algebra():
    """Host side of the schedulers (tables, spacing, per-step scalars) + the oracle restatement's algebra; the
    element-wise step itself is a HIP kernel and is checked against this oracle in the GPU suite."""
    from oracle import scheduler as osch
    from bind_your_avatar_implementation_amd.pipeline import DPMScheduler
    s = DDIMScheduler()
    ac = s.alphas_cumprod
    assert ac.dtype == torch.float64 and torch.equal(ac, osch.alphas_cumprod())
    assert ac.shape == (1000,) and torch.all(ac[1:] <= ac[:-1]) and ac[-1].abs() < 1e-12    # zero terminal SNR
    ts = s.set_timesteps(50)
    assert ts[0] == 999 and ts[-1] == 19 and len(ts) == 50                                  # trailing spacing
    assert torch.equal(ts, osch.trailing_timesteps(1000, 50))
    # v-prediction consistency of the oracle: the true velocity of (x0, eps) lands on the exact DDIM point
    o = osch.DDIM()
    o.set_timesteps(50)
    x0, eps = torch.randn(2, 3, 4, dtype=torch.float64), torch.randn(2, 3, 4, dtype=torch.float64)
    t = int(ts[10])
    a_t, a_p = ac[t], ac[t - 20]
    x_t = a_t.sqrt() * x0 + (1 - a_t).sqrt() * eps
    v = a_t.sqrt() * eps - (1 - a_t).sqrt() * x0
    assert torch.allclose(o.step(v, t, x_t), a_p.sqrt() * x0 + (1 - a_p).sqrt() * eps, atol=1e-12)
    # the product's coefficients are the oracle's scalars
    c = s.coefficients(t, guidance=6.0)
    a = ((1 - a_p) / (1 - a_t)) ** 0.5
    assert float(c["k_sample"]) == float(a) and float(c["k_denoised"]) == -float(a_p ** 0.5 - a_t ** 0.5 * a)
    assert float(c["sqrt_alpha"]) == float(a_t ** 0.5) and c["guidance"] == 6.0
    last = s.coefficients(19)                                   # previous timestep < 0 -> alpha_prev = 1 -> prev = x0
    assert float(last["k_sample"]) == 0.0 and abs(float(last["k_denoised"]) + 1.0) < 1e-12
    # DPM: first-order on the first and last step, second-order (k_cur - k_old = 1) in between; with zero injected
    # noise weight the first-order update is the DDIM point
    d = DPMScheduler()
    d.set_timesteps(50)
    c1, second = d.coefficients(999, None, have_old=False)
    assert not second and c1["k_cur"] == 1.0 and c1["k_old"] == 0.0
    c2, second = d.coefficients(int(ts[10]), int(ts[9]))
    assert second and abs(float(c2["k_cur"]) - float(c2["k_old"]) - 1.0) < 1e-12 and float(c2["k_noise"]) > 0
    cl, second = d.coefficients(19, 39)
    assert not second
    od = osch.DPM()
    od.set_timesteps(50)
    prev, x0_hat = od.step(v, None, t, None, x_t, torch.zeros_like(x_t))
    assert torch.allclose(x0_hat, x0, atol=1e-12)
    h = ((a_p / (1 - a_p)) ** 0.5).log() - ((a_t / (1 - a_t)) ** 0.5).log()
    expect = ((1 - a_p) / (1 - a_t)) ** 0.5 * (-h).exp() * x_t - (-2 * h).expm1() * a_p ** 0.5 * x0
    assert torch.allclose(prev, expect, atol=1e-12)


def test_pipeline_rejects_out_of_scope_inputs():
    from types import SimpleNamespace
    from bind_your_avatar_implementation_amd.pipeline import BindyouravatarPipeline
    fake = SimpleNamespace(config=SimpleNamespace(in_channels=48, patch_size=2, attention_head_dim=64,
                                                  use_rotary_positional_embeddings=True),
                           device=torch.device("cpu"), dtype=torch.bfloat16, _engine=None)
    pipe = BindyouravatarPipeline(fake)
    with pytest.raises(ValueError):
        pipe(num_frames=53, prompt_embeds=torch.zeros(1, 226, 4096), image_latents=torch.zeros(1))
    with pytest.raises(NotImplementedError):
        pipe(prompt="a talking head", image_latents=torch.zeros(1))
    with pytest.raises(NotImplementedError):
        pipe(prompt_embeds=torch.zeros(1, 226, 4096), image_latents=torch.zeros(1), output_type="pil")


def test_callback_on_step_end_receives_and_replaces_tensors():
    """Reference models/pipeline_bindyouravatar.py:950-958: the callback gets the tensors named in
    ``callback_on_step_end_tensor_inputs`` and may replace ``latents`` / ``prompt_embeds`` /
    ``negative_prompt_embeds``.  Host logic only: a stand-in transformer and a caller-supplied scheduler."""
    from types import SimpleNamespace
    from bind_your_avatar_implementation_amd.pipeline import BindyouravatarPipeline

    class Tr:
        device, dtype = torch.device("cpu"), torch.float32
        config = SimpleNamespace(in_channels=48, patch_size=2, attention_head_dim=64,
                                 use_rotary_positional_embeddings=False)
        seen = []

        def precompute_conditioning(self, *a):
            pass

        def release_conditioning(self):
            pass

        def __call__(self, hidden_states, encoder_hidden_states, **kw):
            self.seen.append(encoder_hidden_states.clone())
            return (torch.ones_like(hidden_states[:, :, :16]),)

    class Sch:
        def set_timesteps(self, n, dev):
            return torch.arange(n - 1, -1, -1, device=dev)

        def scale_model_input(self, x, t):
            return x

        def step(self, n32, t, latents, return_dict=False):
            return (latents - 0.5 * n32,)

    calls = []

    def cb(pipe, i, t, kw):
        assert set(kw) == {"latents", "prompt_embeds"}
        calls.append((i, int(t), kw["latents"].clone()))
        return {"latents": kw["latents"] + 10.0, "prompt_embeds": kw["prompt_embeds"] * 2.0}

    tr = Tr()
    pipe = BindyouravatarPipeline(tr, scheduler=Sch())
    lat = torch.zeros(1, 3, 16, 4, 6)
    out = pipe(height=32, width=48, num_frames=9, num_inference_steps=2, guidance_scale=1.0, latents=lat,
               prompt_embeds=torch.ones(1, 4, 8), image_latents=torch.zeros(1, 3, 16, 4, 6),
               image_bg_latents=torch.zeros(1, 3, 16, 4, 6),
               callback_on_step_end=cb, callback_on_step_end_tensor_inputs=["latents", "prompt_embeds"], output_type="latent").frames
    assert [c[0] for c in calls] == [0, 1] and [c[1] for c in calls] == [1, 0]
    assert torch.allclose(calls[0][2], torch.full_like(lat, -0.5))             # latents AFTER the scheduler step
    assert torch.allclose(calls[1][2], torch.full_like(lat, 9.0))              # the callback's replacement was used
    assert torch.allclose(out, torch.full_like(lat, 19.0))
    assert torch.equal(tr.seen[1], 2.0 * tr.seen[0])                            # replaced prompt_embeds reach the model


import os as _os
import sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(__file__), "golden"))      # stub_vae.py: shared with make_golden.py


def _stub_pipeline(in_channels=48, dtype=torch.float32):
    from types import SimpleNamespace
    from bind_your_avatar_implementation_amd.pipeline import BindyouravatarPipeline
    from bind_your_avatar_implementation_amd.vae import DiagonalGaussian
    from stub_vae import StubVAE
    fake = SimpleNamespace(config=SimpleNamespace(in_channels=in_channels, patch_size=2, attention_head_dim=64, sample_frames=49,
                                                  use_rotary_positional_embeddings=False),
                           device=torch.device("cpu"), dtype=dtype, _engine=None)
    return BindyouravatarPipeline(fake, vae=StubVAE(DiagonalGaussian)), fake


def test_prepare_latents_matches_the_reference_draw_for_draw():
    """``prepare_latents`` against the REFERENCE's own (models/pipeline_bindyouravatar.py:376-458), run by
    tests/golden/make_golden.py --case prepare_latents on the same stub VAE and the same seeded generators, called the way
    ``__call__`` calls it (image first -- drawing the noise when none is given -- then the background image with the first
    call's latents).  Bit for bit: one posterior draw per image in batch order, the key-point image encoded as the SECOND
    latent frame with ``num_frames - 2`` zero frames behind it (:412-446), per-sample generator lists, bf16 arithmetic of the
    posterior sample, the 0.7 scaling, and the number of ``vae.encode`` calls."""
    import os
    import numpy as np
    from stub_vae import prepare_latents_cases, prepare_latents_inputs  # noqa: E402  (path set by _stub_pipeline)
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_prepare_latents.npz"))
    for name, dtype, batch, kps, bg, gk in prepare_latents_cases():
        pipe, _ = _stub_pipeline(dtype=dtype)
        inp = prepare_latents_inputs(name, dtype, batch)
        gen = ([torch.Generator().manual_seed(100 + i) for i in range(batch)] if gk == "list"
               else torch.Generator().manual_seed(100))
        given = inp["latents"] if "given_latents" in name else None
        kp = inp["kps"] if kps else None
        lat, img = pipe.prepare_latents(inp["image"], batch, 16, 9, 32, 48, dtype, torch.device("cpu"), gen, given, kp)
        assert lat.dtype == dtype and img.dtype == dtype
        assert torch.equal(lat.float(), torch.from_numpy(fx[name + ".latents"])), name
        assert torch.equal(img.float(), torch.from_numpy(fx[name + ".image_latents"])), name
        if kps:                                              # frame 0 = the image, frame 1 = the key points, then zeros
            assert img[:, 1].abs().sum() > 0 and img[:, 2:].abs().sum() == 0
        else:
            assert img[:, 1:].abs().sum() == 0
        if bg:
            lat2, bgl = pipe.prepare_latents(inp["bg"], batch, 16, 9, 32, 48, dtype, torch.device("cpu"), gen, lat, kp)
            assert torch.equal(lat2, lat)
            assert torch.equal(bgl.float(), torch.from_numpy(fx[name + ".image_bg_latents"])), name
        assert pipe.vae.calls == int(fx[name + ".encode_calls"]), name


def test_call_follows_the_reference_channel_rule_and_rejects_what_it_does_not_do():
    """reference :827-830: the noise gets in_channels // 3 channels only when a background stream is passed, else
    in_channels // 2 (with the stock 48-channel model and no background the reference fails in the patch embedding: here a
    ValueError that says why).  ``kps_cond`` travels into the condition latents; ``num_videos_per_prompt`` / ``eta`` /
    raw key-point lists are refused instead of silently ignored."""
    from types import SimpleNamespace
    seen = []

    class Tr:
        device, dtype = torch.device("cpu"), torch.float32
        config = SimpleNamespace(in_channels=48, patch_size=2, attention_head_dim=64, sample_frames=49,
                                 use_rotary_positional_embeddings=False)

        def precompute_conditioning(self, *a):
            pass

        def release_conditioning(self):
            pass

        def __call__(self, hidden_states, encoder_hidden_states, **kw):
            seen.append(hidden_states.clone())
            return (torch.zeros_like(hidden_states[:, :, :hidden_states.shape[2] // 3]),)

    class Sch:
        init_noise_sigma = 1.0

        def set_timesteps(self, n, dev):
            return torch.arange(n - 1, -1, -1, device=dev)

        def scale_model_input(self, x, t):
            return x

        def step(self, n32, t, latents, return_dict=False):
            return (latents,)

    from bind_your_avatar_implementation_amd.pipeline import BindyouravatarPipeline
    from bind_your_avatar_implementation_amd.vae import DiagonalGaussian
    from stub_vae import StubVAE
    pipe = BindyouravatarPipeline(Tr(), scheduler=Sch(), vae=StubVAE(DiagonalGaussian))
    g = torch.Generator().manual_seed(3)
    image, kps, bg = (torch.rand(1, 3, 32, 48, generator=g) * 2 - 1 for _ in range(3))
    kw = dict(height=32, width=48, num_frames=9, num_inference_steps=1, guidance_scale=1.0, prompt_embeds=torch.ones(1, 4, 8),
              output_type="latent")
    out = pipe(image=image, image_bg=bg, kps_cond=kps, use_inpaint=True, generator=torch.Generator().manual_seed(5), **kw).frames
    x = seen[-1]
    assert out.shape == (1, 3, 16, 4, 6) and x.shape == (1, 3, 48, 4, 6)
    assert x[:, 0, 16:32].abs().sum() > 0 and x[:, 1, 16:32].abs().sum() > 0 and x[:, 2, 16:32].abs().sum() == 0   # image | key points | zeros
    assert x[:, 0, 32:].abs().sum() > 0 and x[:, 1, 32:].abs().sum() > 0                                           # background stream too
    pipe(image=image, image_bg=bg, use_inpaint=False, generator=torch.Generator().manual_seed(5), **kw)
    assert seen[-1][:, :, 32:].abs().sum() == 0 and seen[-1][:, 1, 16:32].abs().sum() == 0      # zero-filled background, no key points
    with pytest.raises(ValueError, match="in_channels // 2"):
        pipe(image=image, **kw)                              # 24 noise + 16 image channels != 48, as in the reference
    with pytest.raises(NotImplementedError, match="num_videos_per_prompt"):
        pipe(image=image, image_bg=bg, num_videos_per_prompt=2, **kw)
    with pytest.raises(NotImplementedError, match="eta"):
        pipe(image=image, image_bg=bg, eta=0.5, **kw)
    with pytest.raises(NotImplementedError, match="key-point IMAGE"):
        pipe(image=image, image_bg=bg, kps_cond=[[1.0, 2.0]] * 5, **kw)


def test_output_type_defaults_to_the_references_pil_and_postprocesses_like_diffusers():
    """reference models/pipeline_bindyouravatar.py:645 (``output_type: str = "pil"``) and :967-971: anything but "latent" is
    decoded and passed through diffusers' ``VideoProcessor.postprocess_video``: x / 2 + 0.5 clamped to [0, 1]; "pt" ->
    [B, F, 3, H, W], "np" -> float32 [B, F, H, W, 3], "pil" -> per sample a list of F uint8 RGB images."""
    import inspect
    import numpy as np
    from bind_your_avatar_implementation_amd.pipeline import BindyouravatarPipeline
    assert inspect.signature(BindyouravatarPipeline.__call__).parameters["output_type"].default == "pil"
    g = torch.Generator().manual_seed(0)
    video = torch.rand(2, 3, 5, 8, 12, generator=g) * 3 - 1.5                  # values outside [-1, 1] get clamped
    pt = BindyouravatarPipeline.postprocess_video(video, "pt")
    want = (video / 2 + 0.5).clamp(0, 1).permute(0, 2, 1, 3, 4)
    assert pt.shape == (2, 5, 3, 8, 12) and torch.equal(pt, want)
    arr = BindyouravatarPipeline.postprocess_video(video, "np")
    assert arr.shape == (2, 5, 8, 12, 3) and arr.dtype == np.float32 and np.array_equal(arr, want.permute(0, 1, 3, 4, 2).numpy())
    pil = BindyouravatarPipeline.postprocess_video(video, "pil")
    assert len(pil) == 2 and len(pil[0]) == 5 and pil[0][0].size == (12, 8) and pil[0][0].mode == "RGB"
    assert np.array_equal(np.asarray(pil[1][4]), (arr[1, 4] * 255).round().astype("uint8"))
    with pytest.raises(ValueError, match="does not exist"):
        BindyouravatarPipeline.postprocess_video(video, "mp4")
