"""Deterministic synthetic weights and inputs for the denoise-step engine.

There is no network on the build or GPU boxes, so checkpoints and datasets are replaced by
tensors drawn from a name-keyed generator: the same name + seed gives the same values in every
process (torch CPU Philox/MT streams are version-stable inside one image), which lets the
golden fixtures generated from the reference in the build container be re-checked on the GPU
box without shipping weights.  Input shapes/statistics follow SURVEY.md section 8(d).

All values are rounded to bf16-representable numbers so an fp32 consumer (the reference / the
oracle) and the bf16 engine hold bit-identical parameters.
"""
import zlib

import torch


def _gen(name, seed, device="cpu"):
    g = torch.Generator(device=device)
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def synth_tensor(name, shape, seed=0, device="cpu"):
    """One parameter/buffer, fp32 with bf16-representable values; the rule is chosen from the name."""
    shape = tuple(shape)
    g = _gen(name, seed, device)
    leaf = name.rsplit(".", 1)[-1]
    rn = lambda: torch.randn(shape, generator=g, device=device, dtype=torch.float32)
    if leaf == "mute_learnable_tokens":
        t = torch.zeros(shape, device=device)
    elif leaf == "learnable_scale":
        t = torch.full(shape, 0.01, device=device)
    elif leaf == "pos_embedding":            # learned PE buffer: text rows are zero (diffusers layout)
        t = rn() * 0.02
    elif len(shape) == 1 and leaf == "weight":   # every 1-D weight on this path is a LayerNorm gain
        t = 1.0 + 0.1 * rn()
    elif len(shape) == 1:
        t = 0.05 * rn()
    elif leaf == "latents":
        t = rn() / (shape[-1] ** 0.5)
    elif leaf == "proj_out" and len(shape) == 2:  # LocalFacialExtractor.proj_out is [dim, out] (x @ W)
        t = rn() / (shape[0] ** 0.5)
    else:                                    # Linear [out, in], Conv [out, in, k...]
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        t = rn() / (fan_in ** 0.5)
    return _bf16_round(t)


def synth_state_dict(named_shapes, seed=0, device="cpu", dtype=torch.float32, skip=("router.pos_emb",),
                     text_rows=226):
    """``named_shapes``: iterable of (name, shape).  Entries in ``skip`` keep their constructor values."""
    out = {}
    for name, shape in named_shapes:
        if name in skip:
            continue
        t = synth_tensor(name, shape, seed, device)
        if name.endswith("pos_embedding"):
            t[:, :text_rows] = 0
        out[name] = t.to(dtype)
    return out


def rope_table(grid, head_dim=64, theta=10000.0):
    """3-D RoPE (cos, sin), fp32 [T*Ht*Wt, head_dim]; t/h/w split D/4, 3D/8, 3D/8."""
    t, ht, wt = grid

    def axis(dim, n):
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
        ang = torch.outer(torch.arange(n, dtype=torch.float32), freqs)
        return ang.cos().repeat_interleave(2, dim=1), ang.sin().repeat_interleave(2, dim=1)

    dt, dh, dw = head_dim // 4, head_dim // 8 * 3, head_dim // 8 * 3
    (ct, st), (ch, sh), (cw, sw) = axis(dt, t), axis(dh, ht), axis(dw, wt)

    def comb(a, b, c):
        a = a[:, None, None, :].expand(-1, ht, wt, -1)
        b = b[None, :, None, :].expand(t, -1, wt, -1)
        c = c[None, None, :, :].expand(t, ht, -1, -1)
        return torch.cat([a, b, c], dim=-1).reshape(t * ht * wt, -1).contiguous()

    return comb(ct, ch, cw), comb(st, sh, sw)


def synth_inputs(batch=1, frames=13, height=60, width=90, in_channels=48, text_len=226, text_dim=4096,
                 n_id=2, seed=0, device="cpu", dtype=torch.float32, head_dim=64, patch=2, uncond_first=False):
    """Keyword arguments for ``transformer.forward`` (models/transformer.py:615-633) with synthetic values.

    ``uncond_first``: row 0 is the classifier-free "uncond" half: its audio is zero
    (models/pipeline_bindyouravatar.py:884).
    """
    def rn(name, shape, std=1.0):
        g = _gen("input." + name, seed, "cpu")
        return _bf16_round(torch.randn(shape, generator=g) * std).to(device=device, dtype=dtype)

    audio_frames = (frames - 1) * 4 + 1 + 4
    audio = rn("audio", (batch, n_id, audio_frames, 12, 768), 0.26)
    if uncond_first:
        audio[0] = 0
    cos, sin = rope_table((frames, height // patch, width // patch), head_dim)
    return dict(
        hidden_states=rn("latents", (batch, frames, in_channels, height, width)),
        encoder_hidden_states=rn("text", (batch, text_len, text_dim)),
        timestep=torch.full((batch,), 999, dtype=torch.int64, device=device),
        image_rotary_emb=(cos.to(device), sin.to(device)),
        id_cond=[rn(f"id_cond{i}", (batch, 1280)) for i in range(n_id)],
        id_vit_hidden=[[rn(f"vit{i}_{k}", (batch, 577, 1024)) for k in range(5)] for i in range(n_id)],
        audio_embeds=audio,
        af_matrix=torch.eye(n_id, device=device, dtype=dtype)[None].repeat(batch, 1, 1),
    )


def mono_audio_extras(seed=0, device="cpu"):
    """Synthetic stand-ins for the two tensors the reference's single-stream audio path adds
    (models/audio_model.py:201-221): the "mute" wav2vec embedding the reference reads from
    ``tests/input/ae_mute.pt`` (not shipped; 60 frames here so the ``[:num_frames * 4 + 1]`` cut is exercised) and a
    non-zero ``mute_learnable_tokens`` (zeros at init, trained in the checkpoints).  Returns (ae_mute, tokens)."""
    ae = synth_tensor("modin.ae_mute.proj", (60, 12, 768), seed, device) * (768 ** 0.5) * 0.26
    tok = synth_tensor("modin.mute_tokens.proj", (1, 32, 768), seed, device) * (768 ** 0.5) * 0.3
    return _bf16_round(ae), _bf16_round(tok)
