"""Restatement of the third-party layers the reference hot path instantiates.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  PARITY UNPINNED for this file:
the algorithms below belong to ``diffusers==0.34.0.dev0`` (reference
``requirements.txt:23``), which is neither vendored under ``/root/reference`` nor
installed here.  They are restated from the library's published behaviour and
anchored on the reference's own call sites:

  * ``Timesteps`` / ``TimestepEmbedding``   models/transformer.py:397-398, 679-686
  * ``CogVideoXPatchEmbed``                 models/transformer.py:378-393, 690
  * ``CogVideoXLayerNormZero``              models/transformer.py:198, 212, 233, 251
  * ``Attention`` + ``CogVideoXAttnProcessor2_0``   models/transformer.py:200-209, 241-245
  * ``Attention`` + default SDPA processor  models/router.py:430-452, models/audio_model.py:179-185
  * ``FeedForward`` (gelu-approximate)      models/transformer.py:214-221, 257
  * ``AdaLayerNorm`` (chunk_dim=1)          models/transformer.py:420-426, 948
  * ``apply_rotary_emb`` / ``get_3d_rotary_pos_embed``   models/pipeline_bindyouravatar.py:586-610

Parameter names follow the library's so a real checkpoint's state dict loads
unchanged (SURVEY.md section 8b lists the key families).
"""
import math

import torch
import torch.nn.functional as F
from torch import nn


# --------------------------------------------------------------------------- time embedding
def sinusoidal_timestep_embedding(timesteps, dim, flip_sin_to_cos=True, freq_shift=0.0, max_period=10000):
    """fp32 sinusoid table lookup: [cos | sin] when ``flip_sin_to_cos``."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - freq_shift)
    ang = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift, scale=1):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        return sinusoidal_timestep_embedding(
            timesteps, self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift
        )


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu"):
        super().__init__()
        assert act_fn == "silu"
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample, condition=None):
        assert condition is None  # timestep_cond is never passed (pipeline_bindyouravatar.py:910-923)
        return self.linear_2(F.silu(self.linear_1(sample)))


# --------------------------------------------------------------------------- patch embed
def sincos_1d(embed_dim, pos):
    omega = torch.arange(embed_dim // 2, dtype=torch.float64)
    omega = 1.0 / (10000 ** (omega / (embed_dim / 2.0)))
    out = pos.reshape(-1).double()[:, None] * omega[None, :]
    return torch.cat([torch.sin(out), torch.cos(out)], dim=1)


def sincos_3d_pos_embed(embed_dim, spatial_size, temporal_size, spatial_scale=1.0, temporal_scale=1.0):
    """Fixed 3-D sin/cos table; only used to *initialise* ``pos_embedding`` (a checkpoint buffer)."""
    w, h = spatial_size
    d_sp, d_t = 3 * embed_dim // 4, embed_dim // 4
    gh = torch.arange(h, dtype=torch.float32) / spatial_scale
    gw = torch.arange(w, dtype=torch.float32) / spatial_scale
    grid = torch.stack(torch.meshgrid(gw, gh, indexing="xy"), dim=0)  # [2, h, w]
    emb_h = sincos_1d(d_sp // 2, grid[0])
    emb_w = sincos_1d(d_sp // 2, grid[1])
    sp = torch.cat([emb_h, emb_w], dim=1)  # [h*w, d_sp]
    tp = sincos_1d(d_t, torch.arange(temporal_size, dtype=torch.float32) / temporal_scale)  # [T, d_t]
    sp = sp[None].expand(temporal_size, -1, -1)
    tp = tp[:, None].expand(-1, h * w, -1)
    return torch.cat([tp, sp], dim=-1).float()  # [T, h*w, D]


class CogVideoXPatchEmbed(nn.Module):
    def __init__(self, patch_size=2, patch_size_t=None, in_channels=16, embed_dim=1920, text_embed_dim=4096,
                 bias=True, sample_width=90, sample_height=60, sample_frames=49, temporal_compression_ratio=4,
                 max_text_seq_length=226, spatial_interpolation_scale=1.875, temporal_interpolation_scale=1.0,
                 use_positional_embeddings=True, use_learned_positional_embeddings=True):
        super().__init__()
        assert patch_size_t is None
        self.patch_size = patch_size
        self.embed_dim = embed_dim
        self.sample_height, self.sample_width, self.sample_frames = sample_height, sample_width, sample_frames
        self.temporal_compression_ratio = temporal_compression_ratio
        self.max_text_seq_length = max_text_seq_length
        self.use_positional_embeddings = use_positional_embeddings
        self.use_learned_positional_embeddings = use_learned_positional_embeddings
        self.proj = nn.Conv2d(in_channels, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)
        self.text_proj = nn.Linear(text_embed_dim, embed_dim)
        if use_positional_embeddings or use_learned_positional_embeddings:
            ph, pw = sample_height // patch_size, sample_width // patch_size
            frames = (sample_frames - 1) // temporal_compression_ratio + 1
            pe = sincos_3d_pos_embed(embed_dim, (pw, ph), frames, spatial_interpolation_scale,
                                     temporal_interpolation_scale).flatten(0, 1)
            joint = torch.zeros(1, max_text_seq_length + pe.shape[0], embed_dim)
            joint[0, max_text_seq_length:] = pe
            self.register_buffer("pos_embedding", joint, persistent=use_learned_positional_embeddings)

    def forward(self, text_embeds, image_embeds):
        text = self.text_proj(text_embeds)
        b, f, c, h, w = image_embeds.shape
        x = self.proj(image_embeds.reshape(-1, c, h, w))
        x = x.view(b, f, *x.shape[1:]).flatten(3).transpose(2, 3).flatten(1, 2)  # (f, h, w) token order
        out = torch.cat([text, x], dim=1).contiguous()
        if self.use_positional_embeddings or self.use_learned_positional_embeddings:
            if self.use_learned_positional_embeddings and (self.sample_width != w or self.sample_height != h):
                raise ValueError("learned positional embeddings need the configured sample height/width")
            out = out + self.pos_embedding.to(dtype=out.dtype)
        return out


# --------------------------------------------------------------------------- norms
class CogVideoXLayerNormZero(nn.Module):
    def __init__(self, conditioning_dim, embedding_dim, elementwise_affine=True, eps=1e-5, bias=True):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(conditioning_dim, 6 * embedding_dim, bias=bias)
        self.norm = nn.LayerNorm(embedding_dim, eps=eps, elementwise_affine=elementwise_affine)

    def forward(self, hidden_states, encoder_hidden_states, temb):
        shift, scale, gate, e_shift, e_scale, e_gate = self.linear(self.silu(temb)).chunk(6, dim=1)
        h = self.norm(hidden_states) * (1 + scale)[:, None, :] + shift[:, None, :]
        e = self.norm(encoder_hidden_states) * (1 + e_scale)[:, None, :] + e_shift[:, None, :]
        return h, e, gate[:, None, :], e_gate[:, None, :]


class AdaLayerNorm(nn.Module):
    def __init__(self, embedding_dim, num_embeddings=None, output_dim=None, norm_elementwise_affine=False,
                 norm_eps=1e-5, chunk_dim=0):
        super().__init__()
        assert chunk_dim == 1 and num_embeddings is None
        self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, output_dim)
        self.norm = nn.LayerNorm(output_dim // 2, norm_eps, norm_elementwise_affine)

    def forward(self, x, timestep=None, temb=None):
        shift, scale = self.linear(self.silu(temb)).chunk(2, dim=1)  # shift FIRST for chunk_dim=1
        return self.norm(x) * (1 + scale[:, None, :]) + shift[:, None, :]


# --------------------------------------------------------------------------- rotary
def apply_rotary_emb(x, freqs):
    """x [B,H,N,D]; freqs = (cos, sin) each [N,D]; interleaved (2i, 2i+1) pairs; math in fp32."""
    cos, sin = freqs
    cos, sin = cos[None, None].to(x.device), sin[None, None].to(x.device)
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * cos + rot.float() * sin).to(x.dtype)


def rope_1d(dim, pos, theta=10000.0):
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
    ang = torch.outer(pos.float(), freqs)
    return ang.cos().repeat_interleave(2, dim=1).float(), ang.sin().repeat_interleave(2, dim=1).float()


def get_3d_rotary_pos_embed(embed_dim, crops_coords, grid_size, temporal_size, theta=10000, use_real=True,
                            grid_type="linspace", max_size=None, device=None):
    """3-D RoPE table (cos, sin) each [T*H*W, embed_dim]; split t/h/w = D/4, 3D/8, 3D/8."""
    assert use_real and grid_type == "linspace"
    (h0, w0), (h1, w1) = crops_coords
    gh, gw = grid_size
    grid_h = torch.linspace(h0, h1 * (gh - 1) / gh, gh, dtype=torch.float32)
    grid_w = torch.linspace(w0, w1 * (gw - 1) / gw, gw, dtype=torch.float32)
    grid_t = torch.arange(temporal_size, dtype=torch.float32)
    dt, dh, dw = embed_dim // 4, embed_dim // 8 * 3, embed_dim // 8 * 3
    ct, st = rope_1d(dt, grid_t, theta)
    ch, sh = rope_1d(dh, grid_h, theta)
    cw, sw = rope_1d(dw, grid_w, theta)

    def combine(t, h, w):
        t = t[:, None, None, :].expand(-1, gh, gw, -1)
        h = h[None, :, None, :].expand(temporal_size, -1, gw, -1)
        w = w[None, None, :, :].expand(temporal_size, gh, -1, -1)
        return torch.cat([t, h, w], dim=-1).reshape(temporal_size * gh * gw, -1)

    return combine(ct, ch, cw), combine(st, sh, sw)


def get_resize_crop_region_for_grid(src, tgt_width, tgt_height):
    """Restated from models/pipeline_bindyouravatar.py:98-113 (reference-owned copy)."""
    th, tw = tgt_height, tgt_width
    h, w = src
    r = h / w
    if r > th / tw:
        resize_h, resize_w = th, int(round(th / h * w))
    else:
        resize_w, resize_h = tw, int(round(tw / w * h))
    top = int(round((th - resize_h) / 2.0))
    left = int(round((tw - resize_w) / 2.0))
    return (top, left), (top + resize_h, left + resize_w)


# --------------------------------------------------------------------------- attention
class AttnProcessor2_0:
    """Default processor: plain (cross-)attention through SDPA, scale = dim_head**-0.5."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        b = hidden_states.shape[0]
        q, k, v = attn.to_q(hidden_states), attn.to_k(ctx), attn.to_v(ctx)
        hd = q.shape[-1] // attn.heads
        q = q.view(b, -1, attn.heads, hd).transpose(1, 2)
        k = k.view(b, -1, attn.heads, hd).transpose(1, 2)
        v = v.view(b, -1, attn.heads, hd).transpose(1, 2)
        if attn.norm_q is not None:
            q, k = attn.norm_q(q), attn.norm_k(k)
        o = F.scaled_dot_product_attention(q, k, v)
        o = o.transpose(1, 2).reshape(b, -1, attn.heads * hd).to(q.dtype)
        return attn.to_out[1](attn.to_out[0](o))


class CogVideoXAttnProcessor2_0:
    """Joint text+video attention; q/k LayerNorm per head; RoPE on the video rows only."""

    def __call__(self, attn, hidden_states, encoder_hidden_states, attention_mask=None, image_rotary_emb=None):
        t_len = encoder_hidden_states.size(1)
        x = torch.cat([encoder_hidden_states, hidden_states], dim=1)
        b, s, _ = x.shape
        q, k, v = attn.to_q(x), attn.to_k(x), attn.to_v(x)
        hd = q.shape[-1] // attn.heads
        q = q.view(b, -1, attn.heads, hd).transpose(1, 2)
        k = k.view(b, -1, attn.heads, hd).transpose(1, 2)
        v = v.view(b, -1, attn.heads, hd).transpose(1, 2)
        if attn.norm_q is not None:
            q, k = attn.norm_q(q), attn.norm_k(k)
        if image_rotary_emb is not None:
            q[:, :, t_len:] = apply_rotary_emb(q[:, :, t_len:], image_rotary_emb)
            k[:, :, t_len:] = apply_rotary_emb(k[:, :, t_len:], image_rotary_emb)
        o = F.scaled_dot_product_attention(q, k, v)
        o = o.transpose(1, 2).reshape(b, -1, attn.heads * hd)
        o = attn.to_out[1](attn.to_out[0](o))
        enc_o, hid_o = o.split([t_len, s - t_len], dim=1)
        return hid_o, enc_o


class FusedCogVideoXAttnProcessor2_0(CogVideoXAttnProcessor2_0):
    pass


class AttentionProcessor:  # type alias only (models/transformer.py:22)
    pass


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 qk_norm=None, eps=1e-5, out_bias=True, processor=None, **unused):
        super().__init__()
        inner = heads * dim_head
        kv_dim = query_dim if cross_attention_dim is None else cross_attention_dim
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv_dim, inner, bias=bias)
        self.to_v = nn.Linear(kv_dim, inner, bias=bias)
        if qk_norm == "layer_norm":
            self.norm_q = nn.LayerNorm(dim_head, eps=eps, elementwise_affine=True)
            self.norm_k = nn.LayerNorm(dim_head, eps=eps, elementwise_affine=True)
        else:
            assert qk_norm is None
            self.norm_q = self.norm_k = None
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=out_bias), nn.Dropout(dropout)])
        self.processor = processor if processor is not None else AttnProcessor2_0()

    def get_processor(self):
        return self.processor

    def set_processor(self, processor):
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)


# --------------------------------------------------------------------------- feed-forward
class GELU(nn.Module):
    def __init__(self, dim_in, dim_out, approximate="none", bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)
        self.approximate = approximate

    def forward(self, x):
        return F.gelu(self.proj(x), approximate=self.approximate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False,
                 inner_dim=None, bias=True):
        super().__init__()
        assert activation_fn == "gelu-approximate"
        inner_dim = inner_dim or int(dim * mult)
        dim_out = dim_out or dim
        mods = [GELU(dim, inner_dim, approximate="tanh", bias=bias), nn.Dropout(dropout),
                nn.Linear(inner_dim, dim_out, bias=bias)]
        if final_dropout:
            mods.append(nn.Dropout(dropout))
        self.net = nn.ModuleList(mods)

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x
