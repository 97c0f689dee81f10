"""Thin host pipeline around the MI355X denoise engine, mirroring the call surface of the reference
``BindyouravatarPipeline.__call__`` (reference models/pipeline_bindyouravatar.py:626-660) for the part of it that
is the per-step loop (:877-962): CFG batching ([uncond, cond] order, zero audio for the uncond half, duplicated or
zeroed identity / audio-face conditioning), ``transformer.forward``, CFG combine in fp32, scheduler step.

Out of scope (third-party models that run once outside the loop): T5 prompt encoding, VAE
encode/decode, key-point drawing.  Consequently ``prompt_embeds`` / ``negative_prompt_embeds`` and the condition
latents are passed as tensors and ``output_type`` must be ``"latent"``; passing ``prompt`` / ``image`` raises.

The scheduler is a restatement of diffusers' ``CogVideoXDDIMScheduler`` (v-prediction, scaled-linear betas with SNR
shift 3.0, zero-terminal-SNR rescale, trailing spacing).  Its configuration ships with the HF checkpoint, not with
the reference repo, so it is "parity unpinned" like the other diffusers-owned pieces.
"""
import math
from types import SimpleNamespace
from typing import Callable, Dict, List, Optional

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


class DDIMScheduler:
    """CogVideoXDDIMScheduler restated (eta = 0 path).  fp32 math on the latents' device."""

    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, snr_shift_scale=3.0,
                 prediction_type="v_prediction"):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
        ac = torch.cumprod(1.0 - betas, dim=0)
        ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)             # SNR shift
        s = ac.sqrt()                                                        # zero terminal SNR rescale
        s0, sT = s[0].clone(), s[-1].clone()
        s = (s - sT) * s0 / (s0 - sT)
        self.alphas_cumprod = (s ** 2).float()
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        self.timesteps = None

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        t = torch.arange(self.num_train_timesteps, 0, -self.num_train_timesteps / num_inference_steps)
        self.timesteps = (t.round() - 1).long().to(device)                   # "trailing" spacing
        return self.timesteps

    def scale_model_input(self, sample, t):
        return sample

    def step(self, model_output, timestep, sample):
        t = int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t].to(sample.device)
        a_prev = (self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod).to(sample.device)
        x, v = sample.float(), model_output.float()
        if self.prediction_type != "v_prediction":
            raise NotImplementedError(self.prediction_type)
        x0 = a_t.sqrt() * x - (1 - a_t).sqrt() * v
        # diffusers' CogVideoX form: prev = a * x_t + b * x0
        a = ((1 - a_prev) / (1 - a_t)).sqrt()
        b = a_prev.sqrt() - a_t.sqrt() * a
        return a * x + b * x0


class BindyouravatarPipeline:
    """See module docstring.  ``transformer``: ``BindyouravatarTransformer3DModel`` of this package."""

    def __init__(self, transformer, scheduler=None, vae_scale_factor_spatial=8, vae_scale_factor_temporal=4):
        self.transformer = transformer
        self.scheduler = scheduler or DDIMScheduler()
        self.vae_scale_factor_spatial = vae_scale_factor_spatial
        self.vae_scale_factor_temporal = vae_scale_factor_temporal
        self._guidance_scale, self._interrupt, self._num_timesteps = 1.0, False, 0

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
                 prompt_embeds=None, negative_prompt_embeds=None, output_type: str = "latent",
                 return_dict: bool = True, callback_on_step_end: Optional[Callable] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ["latents"], max_sequence_length: int = 226,
                 id_vit_hidden=None, id_cond=None, kps_cond=None, audio_embs=None, af_matrix=None,
                 zero2cond_cfg_flag: bool = False, routing_logits_zeros_flag: bool = False,
                 routing_logits_forcing=None, image_bg=None, image_latents=None, image_bg_latents=None):
        if num_frames > 49:
            raise ValueError("The number of frames must be less than 49 for now due to static positional embeddings. "
                             "This will be updated in the future to remove this limitation.")       # reference :739-742
        if prompt is not None or negative_prompt is not None or image is not None or image_bg is not None:
            raise NotImplementedError("T5 / VAE are outside the hot path: pass prompt_embeds, negative_prompt_embeds, "
                                      "image_latents (and image_bg_latents) as tensors")
        if output_type != "latent":
            raise NotImplementedError("VAE decode is outside the hot path: use output_type='latent'")
        if prompt_embeds is None or image_latents is None:
            raise ValueError("prompt_embeds and image_latents are required")
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
        lat_frames = (num_frames - 1) // self.vae_scale_factor_temporal + 1
        ch = tr.config.in_channels // (3 if image_bg_latents is not None or tr.config.in_channels % 3 == 0 else 2)
        shape = (batch, lat_frames, ch, height // self.vae_scale_factor_spatial, width // self.vae_scale_factor_spatial)
        if latents is None:
            latents = torch.randn(shape, generator=generator, device=dev if generator is None else generator.device)
        latents = latents.to(dev, dtype)
        assert tuple(latents.shape) == shape, (tuple(latents.shape), shape)
        image_latents = image_latents.to(dev, dtype)
        if image_bg_latents is None or not use_inpaint:
            image_bg_latents = torch.zeros_like(image_latents) if tr.config.in_channels == 3 * ch else None
        elif image_bg_latents is not None:
            image_bg_latents = image_bg_latents.to(dev, dtype)
        ts = self.scheduler.set_timesteps(num_inference_steps, dev) if timesteps is None else torch.tensor(timesteps, device=dev)
        self._num_timesteps = len(ts)
        rope = self._rotary(height, width, lat_frames, dev) if tr.config.use_rotary_positional_embeddings else None

        if cfg:
            id_vit_hidden = cfg_id_vit_hidden(id_vit_hidden, zero2cond_cfg_flag) if id_vit_hidden is not None else None
            id_cond = cfg_id_cond(id_cond, zero2cond_cfg_flag) if id_cond is not None else None
            af_matrix = cfg_af_matrix(af_matrix, zero2cond_cfg_flag) if af_matrix is not None else None
            audio_embs = cfg_audio(audio_embs) if audio_embs is not None else None
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
            noise = tr(hidden_states=x, encoder_hidden_states=prompt_embeds, timestep=t.expand(x.shape[0]),
                       image_rotary_emb=rope, return_dict=False, id_vit_hidden=id_vit_hidden, id_cond=id_cond,
                       audio_embeds=audio_embs, af_matrix=af_matrix, denoise_step=i,
                       routing_logits_zeros_flag=routing_logits_zeros_flag,
                       routing_logits_forcing=routing_logits_forcing)[0].float()
            if tr._engine is not None:
                tr._engine.cache_invariants = True           # conditioning does not change between steps
            if use_dynamic_cfg:
                self._guidance_scale = 1 + guidance_scale * (
                    (1 - math.cos(math.pi * ((num_inference_steps - t.item()) / num_inference_steps) ** 5.0)) / 2)
            if cfg:
                u, c = noise.chunk(2)
                noise = u + self.guidance_scale * (c - u)
            latents = self.scheduler.step(noise, t, latents).to(dtype)
            if callback_on_step_end is not None:
                kw = {k: locals()[k] for k in callback_on_step_end_tensor_inputs}
                out = callback_on_step_end(self, i, t, kw)
                latents = out.pop("latents", latents)
        if not return_dict:
            return (latents,)
        return SimpleNamespace(frames=latents)
