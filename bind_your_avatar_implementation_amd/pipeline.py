"""Thin host pipeline around the MI355X denoise engine, mirroring the call surface of the reference
``BindyouravatarPipeline.__call__`` (reference models/pipeline_bindyouravatar.py:626-660) for the part of it that
is the per-step loop (:877-962): CFG batching ([uncond, cond] order, zero audio for the uncond half, duplicated or
zeroed identity / audio-face conditioning), ``transformer.forward``, CFG combine in fp32, scheduler step.

Out of scope (third-party models that run once outside the loop): T5 prompt encoding, VAE
encode/decode, key-point drawing.  Consequently ``prompt_embeds`` / ``negative_prompt_embeds`` and the condition
latents are passed as tensors and ``output_type`` must be ``"latent"`` (the default is the reference's ``"pil"``, which
needs the VAE); passing ``prompt`` / ``image`` raises.

Schedulers: ``DDIMScheduler`` / ``DPMScheduler`` hold the host side (float64 coefficient tables, per-step scalars) of
diffusers' ``CogVideoXDDIMScheduler`` / ``CogVideoXDPMScheduler`` (v-prediction, scaled-linear betas with SNR shift 3.0,
zero-terminal-SNR rescale, trailing spacing); the CFG combine and the step itself are one fused HIP launch
(``bya_cfg_scheduler_step``).  Their configuration ships with the HF checkpoint, not with
the reference repo, so they are "parity unpinned" like the other diffusers-owned pieces.
"""
import math
import os
from types import SimpleNamespace
from typing import Callable, List, Optional

import torch


# ---- CFG helpers (reference models/utils.py:630-657, models/pipeline_bindyouravatar.py:877-884) ------------------
def cfg_id_vit_hidden(id_vit_hidden, zero2cond_cfg_flag=False):
    if id_vit_hidden is None:
        raise ValueError("id_vit_hidden is None")
    return [[torch.cat([torch.zeros_like(t) if zero2cond_cfg_flag else t, t], dim=0) for t in inner]
            for inner in id_vit_hidden]


def cfg_id_cond(id_cond, zero2cond_cfg_flag=False):
    if id_cond is None:
        raise ValueError("id_cond is None")
    return [torch.cat([torch.zeros_like(t) if zero2cond_cfg_flag else t, t], dim=0) for t in id_cond]


def cfg_af_matrix(af_matrix, zero2cond_cfg_flag=False):
    return torch.cat([torch.zeros_like(af_matrix), af_matrix], 0) if zero2cond_cfg_flag else af_matrix.repeat(2, 1, 1)


def cfg_audio(audio_embs):
    return torch.cat([torch.zeros_like(audio_embs), audio_embs], dim=0)     # uncond half hears silence (:884)


def get_af_matrix_infer(speaker_pos):
    """reference models/utils.py:660-670"""
    if speaker_pos == "left":
        return torch.eye(2)
    if speaker_pos == "right":
        return 1 - torch.eye(2)
    raise ValueError("speaker is not left or right")


def _alphas_cumprod(num_train_timesteps, beta_start, beta_end, snr_shift_scale):
    """float64: scaled-linear betas -> cumprod -> SNR shift -> zero-terminal-SNR rescale (CogVideoX schedulers)."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
    ac = torch.cumprod(1.0 - betas, dim=0)
    ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)
    s = ac.sqrt()
    s0, sT = s[0].clone(), s[-1].clone()
    s = (s - sT) * (s0 / (s0 - sT))
    return s ** 2


def get_routing_logits_from_masks(masks, latent_frames=13, height=60, width=90, patch=2):
    """Stage-2 inference of the reference (infer.py:368-413 -> util/utils.py:871-936): per-identity tracking masks
    ``uint8 [n_id, T, H, W]`` on the GPU -> ``routing_logits_forcing [1, N, n_id]`` for ``__call__`` / ``forward``.
    Reading the PNG frames (``annotated_frame_%05d.png`` under ``<dir>/1``, ``<dir>/2``) stays with the caller."""
    from . import ops
    return ops.masks_to_routing_logits(masks.contiguous(), latent_frames, height // patch, width // patch)


class _SchedulerBase:
    """Host side of a scheduler: float64 coefficient tables and per-step scalars.  The element-wise work (CFG combine
    + the step itself) runs in ONE HIP launch, ``ops.cfg_scheduler_step`` -- there is no torch implementation here."""

    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, snr_shift_scale=3.0,
                 prediction_type="v_prediction"):
        if prediction_type != "v_prediction":
            raise NotImplementedError(prediction_type)
        self.alphas_cumprod = _alphas_cumprod(num_train_timesteps, beta_start, beta_end, snr_shift_scale)
        self.final_alpha_cumprod = torch.tensor(1.0, dtype=torch.float64)
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        self.timesteps = None
        self.init_noise_sigma = 1.0

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        t = torch.arange(self.num_train_timesteps, 0, -self.num_train_timesteps / num_inference_steps)
        self.timesteps = (t.round() - 1).long().to(device)                   # "trailing" spacing
        return self.timesteps

    def scale_model_input(self, sample, t):
        return sample

    def _alphas(self, timestep):
        t = int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return a_t, a_prev, prev_t


class DDIMScheduler(_SchedulerBase):
    """CogVideoXDDIMScheduler, eta = 0."""

    def coefficients(self, timestep, guidance=1.0):
        a_t, a_prev, _ = self._alphas(timestep)
        a = ((1 - a_prev) / (1 - a_t)) ** 0.5
        b = a_prev ** 0.5 - a_t ** 0.5 * a
        return dict(guidance=guidance, sqrt_alpha=a_t ** 0.5, sqrt_beta=(1 - a_t) ** 0.5, k_sample=a, k_denoised=-b,
                    k_noise=0.0, k_cur=1.0, k_old=0.0)

    def step(self, model_output, timestep, sample, guidance=1.0):
        """``model_output``: the bf16 prediction, [1, ...] or the CFG pair [2, ...]; returns the new bf16 latents."""
        from . import ops
        return ops.cfg_scheduler_step(model_output.to(torch.bfloat16).contiguous(), sample.contiguous(),
                                      self.coefficients(timestep, guidance))


class DPMScheduler(_SchedulerBase):
    """CogVideoXDPMScheduler (SDE DPM-Solver++ 2M; what the reference's infer.py:202 configures).  Stochastic: every
    step consumes Gaussian draws from the caller's generator exactly like diffusers does (one draw on the first and on
    the last step, two on the others, of which the second reaches the sample)."""

    def coefficients(self, timestep, timestep_back=None, guidance=1.0, have_old=True):
        a_t, a_prev, prev_t = self._alphas(timestep)
        lamb = ((a_t / (1 - a_t)) ** 0.5).log()
        lamb_next = ((a_prev / (1 - a_prev)) ** 0.5).log()
        h = lamb_next - lamb
        m0 = ((1 - a_prev) / (1 - a_t)) ** 0.5 * (-h).exp()
        m1 = (-2 * h).expm1() * a_prev ** 0.5
        m_noise = (1 - a_prev) ** 0.5 * (1 - (-2 * h).exp()) ** 0.5
        second = have_old and timestep_back is not None and prev_t >= 0
        k_cur, k_old = 1.0, 0.0
        if second:
            a_back = self.alphas_cumprod[int(timestep_back)]
            lamb_prev = ((a_back / (1 - a_back)) ** 0.5).log()
            r = (lamb - lamb_prev) / h
            k_cur, k_old = 1 + 1 / (2 * r), 1 / (2 * r)
        return dict(guidance=guidance, sqrt_alpha=a_t ** 0.5, sqrt_beta=(1 - a_t) ** 0.5, k_sample=m0, k_denoised=m1,
                    k_noise=m_noise, k_cur=k_cur, k_old=k_old), second

    def step(self, model_output, old_pred_original_sample, timestep, timestep_back, sample, guidance=1.0,
             generator=None):
        """-> (new bf16 latents, fp32 x0 for the next call)."""
        from . import ops
        coef, second = self.coefficients(timestep, timestep_back, guidance, old_pred_original_sample is not None)
        gdev = sample.device if generator is None else generator.device
        draw = lambda: torch.randn(sample.shape, generator=generator, device=gdev, dtype=sample.dtype).to(sample.device)
        noise = draw()
        if second:
            noise = draw()                    # diffusers draws again for the corrected sample; the first is unused
        x0 = torch.empty(sample.shape, dtype=torch.float32, device=sample.device)
        prev = ops.cfg_scheduler_step(model_output.to(torch.bfloat16).contiguous(), sample.contiguous(), coef,
                                      old_x0=old_pred_original_sample if second else None, noise=noise, x0_out=x0)
        return prev, x0


class BindyouravatarPipeline:
    """See module docstring.  ``transformer``: ``BindyouravatarTransformer3DModel`` of this package."""

    def __init__(self, transformer, scheduler=None, vae_scale_factor_spatial=8, vae_scale_factor_temporal=4, vae=None):
        self.transformer = transformer
        self.scheduler = scheduler or DDIMScheduler()
        # ``vae``: a ``BindyouravatarVAE`` (this package's AutoencoderKLCogVideoX on the HIP kernels).  With it the pipeline
        # takes ``image`` / ``image_bg`` (encoded like the reference's prepare_latents, models/pipeline_bindyouravatar.py
        # :406-424) and returns frames (``decode_latents``, :461-466); without it, latents in and latents out.
        self.vae = vae
        if vae is not None:                                   # reference :231-240
            vae_scale_factor_spatial = 2 ** (len(vae.config.block_out_channels) - 1)
            vae_scale_factor_temporal = vae.config.temporal_compression_ratio
        self.vae_scale_factor_spatial = vae_scale_factor_spatial
        self.vae_scale_factor_temporal = vae_scale_factor_temporal
        self.vae_scaling_factor_image = vae.config.scaling_factor if vae is not None else 0.7
        self._guidance_scale, self._interrupt, self._num_timesteps = 1.0, False, 0

    def fuse_lora(self, lora_scale=1.0, **kw):
        """diffusers pipeline API the reference calls (infer.py:279): fold the transformer's staged LoRA adapters."""
        return self.transformer.fuse_lora(lora_scale)

    guidance_scale = property(lambda self: self._guidance_scale)
    num_timesteps = property(lambda self: self._num_timesteps)
    interrupt = property(lambda self: self._interrupt)

    def _rotary(self, height, width, frames, device):
        from .synth import rope_table
        p = self.transformer.config.patch_size
        gh, gw = height // (self.vae_scale_factor_spatial * p), width // (self.vae_scale_factor_spatial * p)
        cos, sin = rope_table((frames, gh, gw), self.transformer.config.attention_head_dim)
        return cos.to(device), sin.to(device)

    @torch.no_grad()
    def __call__(self, image=None, prompt=None, negative_prompt=None, height: int = 480, width: int = 720,
                 num_frames: int = 49, num_inference_steps: int = 50, timesteps: Optional[List[int]] = None,
                 guidance_scale: float = 6, use_inpaint: bool = False, use_dynamic_cfg: bool = False,
                 num_videos_per_prompt: int = 1, eta: float = 0.0, generator=None, latents=None,
                 prompt_embeds=None, negative_prompt_embeds=None, output_type: str = "pil",
                 return_dict: bool = True, callback_on_step_end: Optional[Callable] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ["latents"], max_sequence_length: int = 226,
                 id_vit_hidden=None, id_cond=None, kps_cond=None, audio_embs=None, af_matrix=None,
                 zero2cond_cfg_flag: bool = False, routing_logits_zeros_flag: bool = False,
                 routing_logits_forcing=None, image_bg=None, image_latents=None, image_bg_latents=None,
                 kps_cond_latents=None):
        max_frames = int(getattr(self.transformer.config, "sample_frames", 49))
        if num_frames > max_frames:     # reference :739-742 with its constant 49 = the stock config's sample_frames
            raise ValueError(f"The number of frames must be less than {max_frames} for now due to static positional "
                             "embeddings. This will be updated in the future to remove this limitation.")
        if prompt is not None or negative_prompt is not None:
            raise NotImplementedError("the T5 text encoder is outside the hot path: pass prompt_embeds / negative_prompt_embeds")
        if (image is not None or image_bg is not None or output_type != "latent") and self.vae is None:
            raise NotImplementedError("pass vae=BindyouravatarVAE(...) to the pipeline to encode `image` / `image_bg` and to "
                                      "decode frames; without it give image_latents and use output_type='latent'")
        if num_videos_per_prompt != 1:
            raise NotImplementedError("num_videos_per_prompt != 1: the reference repeats only the prompt embeddings "
                                      "(pipeline_bindyouravatar.py:786-799), not the identity / audio conditioning; pass a batch")
        if eta != 0.0:
            raise NotImplementedError("eta != 0: the built-in schedulers are the deterministic DDIM step and DPM-Solver++ "
                                      "(which takes no eta); bring a diffusers scheduler object for stochastic DDIM")
        if kps_cond is not None and not torch.is_tensor(kps_cond):
            raise NotImplementedError("kps_cond must be the key-point IMAGE tensor [B, 3, H, W] in [-1, 1] (the reference "
                                      "draws it with cv2, models/utils.py draw_kps + VideoProcessor.preprocess, :814-818: "
                                      "image-side preprocessing outside the engine), or pass kps_cond_latents")
        if prompt_embeds is None or (image is None and image_latents is None):
            raise ValueError("prompt_embeds and image (or image_latents) are required")
        tr = self.transformer
        dev, dtype = tr.device, tr.dtype
        self._guidance_scale, self._interrupt = guidance_scale, False
        cfg = guidance_scale > 1.0
        batch = prompt_embeds.shape[0]
        prompt_embeds = prompt_embeds.to(dev, dtype)
        if cfg:
            if negative_prompt_embeds is None:
                raise ValueError("classifier-free guidance needs negative_prompt_embeds")
            prompt_embeds = torch.cat([negative_prompt_embeds.to(dev, dtype), prompt_embeds], dim=0)   # [uncond, cond]
        # reference :827-830: the background stream decides how in_channels splits (noise | image [| background])
        has_bg = image_bg is not None or image_bg_latents is not None
        ch = tr.config.in_channels // (3 if has_bg else 2)
        latents, image_latents = self.prepare_latents(image, batch, ch, num_frames, height, width, dtype, dev, generator,
                                                      latents, kps_cond, image_latents, kps_cond_latents)
        lat_frames = latents.shape[1]
        if has_bg:                                           # :845-860 (the second call never draws new noise)
            _, image_bg_latents = self.prepare_latents(image_bg, batch, ch, num_frames, height, width, dtype, dev, generator,
                                                       latents, kps_cond, image_bg_latents, kps_cond_latents)
            if not use_inpaint:
                image_bg_latents = torch.zeros_like(image_latents)
        got = latents.shape[2] + image_latents.shape[2] * (2 if has_bg else 1)
        if got != tr.config.in_channels:
            raise ValueError(f"noise ({latents.shape[2]}) + condition latents ({got - latents.shape[2]}) = {got} channels, but the "
                             f"transformer takes in_channels = {tr.config.in_channels}: like the reference (:827-830) the "
                             "pipeline gives the noise in_channels // 3 channels only when a background stream is passed "
                             "(image_bg / image_bg_latents; with use_inpaint=False it is zero-filled), else in_channels // 2")
        ts = self.scheduler.set_timesteps(num_inference_steps, dev) if timesteps is None else torch.tensor(timesteps, device=dev)
        self._num_timesteps = len(ts)
        rope = self._rotary(height, width, lat_frames, dev) if tr.config.use_rotary_positional_embeddings else None

        if cfg:
            id_vit_hidden = cfg_id_vit_hidden(id_vit_hidden, zero2cond_cfg_flag) if id_vit_hidden is not None else None
            id_cond = cfg_id_cond(id_cond, zero2cond_cfg_flag) if id_cond is not None else None
            af_matrix = cfg_af_matrix(af_matrix, zero2cond_cfg_flag) if af_matrix is not None else None
            audio_embs = cfg_audio(audio_embs) if audio_embs is not None else None
        old_x0 = None
        # the identities and the audio do not change between steps: face tokens, audio context and their K/V once
        tr.precompute_conditioning(id_cond, id_vit_hidden, audio_embs, lat_frames)
        # one synchronisation per step (~0.35 s of GPU work each): the hand-off counters of the split kernels (ops.heal_handoffs).
        # Unsharded runs only -- ranks of a sharded step would have to agree to repeat it; they keep the end-of-clip check.
        heal = (torch.device(dev).type == "cuda" and getattr(tr, "_seq_world", 1) == 1 and getattr(tr, "_cfg", None) is None
                and os.environ.get("BYA_SELF_HEAL", "1") != "0")
        if heal:
            from . import ops
        for i, t in enumerate(ts):
            if self._interrupt:
                continue
            x = torch.cat([latents] * 2) if cfg else latents
            x = self.scheduler.scale_model_input(x, t)
            cond = (torch.cat([image_latents] * 2) if not zero2cond_cfg_flag else
                    torch.cat([torch.zeros_like(image_latents), image_latents], 0)) if cfg else image_latents
            if image_bg_latents is not None:
                bg = torch.cat([image_bg_latents] * 2) if cfg else image_bg_latents
                cond = torch.cat([cond, bg], dim=2)
            x = torch.cat([x, cond], dim=2)
            def predict():
                return tr(hidden_states=x, encoder_hidden_states=prompt_embeds, timestep=t.expand(x.shape[0]),
                          image_rotary_emb=rope, return_dict=False, id_vit_hidden=id_vit_hidden, id_cond=id_cond,
                          audio_embeds=audio_embs, af_matrix=af_matrix, denoise_step=i,
                          routing_logits_zeros_flag=routing_logits_zeros_flag,
                          routing_logits_forcing=routing_logits_forcing)[0]
            noise = predict()
            if heal and ops.heal_handoffs(dev):
                # a split-K / stream-K hand-off of this step timed out (the GPU is shared: the grids were not co-resident);
                # the library is in its unsplit mode now -- compute the step again instead of failing the clip at its end
                noise = predict()
            if use_dynamic_cfg:
                self._guidance_scale = 1 + guidance_scale * (
                    (1 - math.cos(math.pi * ((num_inference_steps - t.item()) / num_inference_steps) ** 5.0)) / 2)
            # CFG combine (fp32, [uncond, cond]) + scheduler step: one fused launch (ops.cfg_scheduler_step)
            g = self.guidance_scale if cfg else 1.0
            if isinstance(self.scheduler, DPMScheduler):
                latents, old_x0 = self.scheduler.step(noise, old_x0, t, ts[i - 1] if i > 0 else None, latents,
                                                      guidance=g, generator=generator)
            elif isinstance(self.scheduler, DDIMScheduler):
                latents = self.scheduler.step(noise, t, latents, guidance=g)
            else:                                  # a scheduler object brought by the caller (diffusers API)
                n32 = noise.float()
                if cfg:
                    u, c = n32.chunk(2)
                    n32 = u + self.guidance_scale * (c - u)
                latents = self.scheduler.step(n32, t, latents, return_dict=False)[0].to(dtype)
            if callback_on_step_end is not None:         # reference models/pipeline_bindyouravatar.py:950-958
                scope = locals()                          # (a comprehension has its own scope before Python 3.12)
                callback_kwargs = {}
                for k in callback_on_step_end_tensor_inputs:
                    callback_kwargs[k] = scope[k]
                callback_outputs = callback_on_step_end(self, i, t, callback_kwargs)
                latents = callback_outputs.pop("latents", latents)
                prompt_embeds = callback_outputs.pop("prompt_embeds", prompt_embeds)
                negative_prompt_embeds = callback_outputs.pop("negative_prompt_embeds", negative_prompt_embeds)
        tr.release_conditioning()
        if torch.device(dev).type == "cuda":    # end of the clip: no split-K hand-off timed out on the way (one sync, once)
            from . import ops
            ops.check_gemm_workspace(dev)
        if output_type != "latent":                            # reference :967-971
            latents = self.postprocess_video(self.decode_latents(latents), output_type)
        if not return_dict:
            return (latents,)
        return SimpleNamespace(frames=latents)

    @staticmethod
    def postprocess_video(video, output_type="pil"):
        """diffusers ``VideoProcessor.postprocess_video`` (reference models/pipeline_bindyouravatar.py:969): decoded video
        [B, 3, F, H, W] in about [-1, 1] -> per sample [F, 3, H, W] denormalised to [0, 1] (x / 2 + 0.5, clamped), returned as
        ``"pt"``: one tensor [B, F, 3, H, W]; ``"np"``: float32 array [B, F, H, W, 3]; ``"pil"`` (the reference's default):
        a list per sample of F ``PIL.Image`` frames (uint8 = round(255 x))."""
        if output_type not in ("pt", "np", "pil"):
            raise ValueError(f"{output_type} does not exist. Please choose one of ['np', 'pt', 'pil']")
        vid = (video.float() / 2 + 0.5).clamp(0, 1).permute(0, 2, 1, 3, 4)           # [B, F, 3, H, W]
        if output_type == "pt":
            return vid
        arr = vid.permute(0, 1, 3, 4, 2).cpu().numpy()                                 # [B, F, H, W, 3]
        if output_type == "np":
            return arr
        from PIL import Image
        return [[Image.fromarray(f) for f in (sample * 255).round().astype("uint8")] for sample in arr]

    def encode_image(self, image, generator=None):
        """image [B, 3, H, W] in [-1, 1] -> scaled latents [B, 1, C, H / 8, W / 8]: the posterior SAMPLE of every image on
        its own, in batch order, from the caller's generator (reference :406-424: one ``vae.encode`` + one draw per image)."""
        gens = generator if isinstance(generator, (list, tuple)) else [generator] * image.shape[0]
        lat = [self.vae.encode(img[None, :, None]).latent_dist.sample(g) for img, g in zip(image, gens)]
        return self.vae_scaling_factor_image * torch.cat(lat, dim=0).permute(0, 2, 1, 3, 4)

    def prepare_latents(self, image, batch_size=1, num_channels_latents=16, num_frames=13, height=60, width=90, dtype=None,
                        device=None, generator=None, latents=None, kps_cond=None, image_latents=None, kps_cond_latents=None):
        """reference models/pipeline_bindyouravatar.py:376-458, draw for draw: encode the first frame (one posterior draw
        per image), then the key-point image when given (one draw per image), then -- only when the caller passed no
        ``latents`` -- the initial noise.  The encoded frame comes first, the key-point frame second, zeros for the
        ``num_frames - 1`` (``- 2`` with key points) latent frames to generate.  ``image_latents`` / ``kps_cond_latents``
        [B, 1, C, h, w] (already scaled) stand in for ``image`` / ``kps_cond`` when the pipeline has no VAE.
        -> (latents [B, F, C, h, w], condition latents [B, F, C', h, w])."""
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective "
                             f"batch size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if (image is not None or (kps_cond is not None and kps_cond_latents is None)) and self.vae is None:
            raise NotImplementedError("pass vae=BindyouravatarVAE(...) to the pipeline to encode `image` / `image_bg` / `kps_cond`; "
                                      "without it give image_latents (and kps_cond_latents)")
        frames = (num_frames - 1) // self.vae_scale_factor_temporal + 1
        h, w = height // self.vae_scale_factor_spatial, width // self.vae_scale_factor_spatial
        shape = (batch_size, frames, num_channels_latents, h, w)
        if image is not None and image_latents is None:
            image_latents = self.encode_image(image.to(device, dtype), generator)
            if kps_cond is not None and kps_cond_latents is None:
                kps_cond_latents = self.encode_image(kps_cond.to(device, dtype), generator)
        elif kps_cond is not None and kps_cond_latents is None:
            kps_cond_latents = self.encode_image(kps_cond.to(device, dtype), generator)
        parts = [image_latents.to(device, dtype)]
        if kps_cond_latents is not None:
            parts.append(kps_cond_latents.to(device, dtype))
        have = sum(p.shape[1] for p in parts)
        if have < frames:
            parts.append(torch.zeros(batch_size, frames - have, *parts[0].shape[2:], device=device, dtype=dtype))
        image_latents = torch.cat(parts, dim=1)
        if latents is None:          # diffusers randn_tensor: drawn on the generator's device, in the target dtype
            if isinstance(generator, (list, tuple)):
                latents = torch.cat([torch.randn((1,) + shape[1:], generator=g, device=g.device, dtype=dtype).to(device)
                                     for g in generator], dim=0)
            else:
                gdev = device if generator is None else generator.device
                latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
        else:
            latents = latents.to(device)
        if tuple(latents.shape) != shape:
            raise ValueError(f"latents {tuple(latents.shape)} != {shape}")
        return latents.to(dtype) * getattr(self.scheduler, "init_noise_sigma", 1.0), image_latents

    def decode_latents(self, latents):
        """reference :461-466."""
        return self.vae.decode(latents.permute(0, 2, 1, 3, 4) / self.vae_scaling_factor_image).sample
