"""The video VAE either side of the denoise loop on the HIP kernels.

Mirrors the part of diffusers' ``AutoencoderKLCogVideoX`` the reference pipeline uses:

* ``vae.decode(latents).sample``            (models/pipeline_bindyouravatar.py:461-466, ``decode_latents``)
* ``vae.encode(image).latent_dist.sample()`` on ONE conditioning frame (``:406-424``, ``image.unsqueeze(2)``)
* ``vae.config.scaling_factor`` (``vae_scaling_factor_image``), ``vae.config.block_out_channels`` /
  ``temporal_compression_ratio`` (``vae_scale_factor_spatial`` / ``_temporal``, ``:231-240``)

and the state-dict keys of that class (``encoder.down_blocks.0.resnets.0.conv1.conv.weight`` ...), so the CogVideoX
checkpoint's ``vae/diffusion_pytorch_model.safetensors`` loads with ``load_state_dict`` as it is.  The arithmetic is the
published algorithm of that class (restated for the tests in ``oracle/vae.py``: **parity unpinned**, the layer lives in
an un-vendored third-party dependency): causal 3x3x3 convolutions with the chunk-to-chunk frame cache, GroupNorm(32) /
SpatialNorm3D, nearest up-sampling with the first-frame rule, decoding in chunks of two latent frames.

How it runs here: activations are channels-last bf16 ``[T, H, W, C]``; every convolution is ``ops.vae_patches`` (gather of
the causal patch matrix, up-sampling folded into the gather) + ``ops.gemm`` (bias and the residual add in its epilogue),
1x1x1 convolutions are ``ops.gemm`` alone, GroupNorm + spatial modulation + SiLU is one statistics pass and one apply pass.
The patch matrix of a slab of frames is materialised in HBM (a first version: an implicit-GEMM convolution that builds the
patches in LDS is the next step -- DESIGN.md); 288 GB of HBM make slabs of whole frames affordable.  No torch math on the
activations: torch holds memory, views and the final layout copy.
"""
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops


class _Causal(nn.Module):           # parameter holders with diffusers' names; never called
    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, (k, k, k))


class _SpatialNorm(nn.Module):
    def __init__(self, ch, zq, groups):
        super().__init__()
        self.norm_layer = nn.GroupNorm(groups, ch, eps=1e-6)
        self.conv_y, self.conv_b = _Causal(zq, ch, 1), _Causal(zq, ch, 1)


class _Resnet(nn.Module):
    def __init__(self, cin, cout, groups, zq=None):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.norm1 = _SpatialNorm(cin, zq, groups) if zq else nn.GroupNorm(groups, cin, eps=1e-6)
        self.norm2 = _SpatialNorm(cout, zq, groups) if zq else nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv1, self.conv2 = _Causal(cin, cout, 3), _Causal(cout, cout, 3)
        if cin != cout:
            self.conv_shortcut = nn.Conv3d(cin, cout, 1)


class _Sampler(nn.Module):
    def __init__(self, ch, stride):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=stride)


class _Block(nn.Module):
    def __init__(self, resnets, up=None, down=None):
        super().__init__()
        self.resnets = nn.ModuleList(resnets)
        if up is not None:
            self.upsamplers = nn.ModuleList([up])
        if down is not None:
            self.downsamplers = nn.ModuleList([down])


class _Decoder(nn.Module):
    def __init__(self, zc, out_ch, chans, layers, groups):
        super().__init__()
        rev = list(reversed(chans))
        self.conv_in = _Causal(zc, rev[0], 3)
        self.mid_block = _Block([_Resnet(rev[0], rev[0], groups, zc) for _ in range(2)])
        blocks, cin = [], rev[0]
        for i, cout in enumerate(rev):
            blocks.append(_Block([_Resnet(cin if j == 0 else cout, cout, groups, zc) for j in range(layers + 1)],
                                 up=_Sampler(cout, 1) if i < len(rev) - 1 else None))
            cin = cout
        self.up_blocks = nn.ModuleList(blocks)
        self.norm_out = _SpatialNorm(rev[-1], zc, groups)
        self.conv_out = _Causal(rev[-1], out_ch, 3)


class _Encoder(nn.Module):
    def __init__(self, in_ch, zc, chans, layers, groups):
        super().__init__()
        self.conv_in = _Causal(in_ch, chans[0], 3)
        blocks, cin = [], chans[0]
        for i, cout in enumerate(chans):
            blocks.append(_Block([_Resnet(cin if j == 0 else cout, cout, groups) for j in range(layers)],
                                 down=_Sampler(cout, 2) if i < len(chans) - 1 else None))
            cin = cout
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = _Block([_Resnet(cin, cin, groups) for _ in range(2)])
        self.norm_out = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv_out = _Causal(cin, 2 * zc, 3)


class DiagonalGaussian:
    """``latent_dist`` of ``encode``: mean | logvar; ``sample`` draws with the caller's generator.  diffusers'
    DiagonalGaussianDistribution, arithmetic included: logvar clamped to [-30, 20], ``std = exp(0.5 logvar)``, the noise
    drawn in the PARAMETERS' dtype on the generator's device (randn_tensor) and ``mean + std * noise`` evaluated in that
    dtype -- with the same seed the draw stream and the bf16 roundings are the reference pipeline's."""

    def __init__(self, moments):
        self.mean, logvar = moments.chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        gdev = self.mean.device if generator is None else generator.device
        noise = torch.randn(self.mean.shape, generator=generator, device=gdev, dtype=self.mean.dtype).to(self.mean.device)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class BindyouravatarVAE(nn.Module):
    """Drop-in for the ``AutoencoderKLCogVideoX`` instance of the reference pipeline (``decode`` / ``encode`` / ``config``)."""

    PATCH_BYTES = 4 << 30          # bytes of patch matrix materialised per slab of frames

    def __init__(self, in_channels=3, out_channels=3, latent_channels=16, block_out_channels=(128, 256, 256, 512),
                 layers_per_block=3, norm_num_groups=32, temporal_compression_ratio=4, scaling_factor=0.7,
                 num_latent_frames_batch_size=2, device=None):
        super().__init__()
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels, latent_channels=latent_channels,
                                      block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                      norm_num_groups=norm_num_groups, temporal_compression_ratio=temporal_compression_ratio,
                                      scaling_factor=scaling_factor)
        self.num_latent_frames_batch_size = num_latent_frames_batch_size
        self.n_time = temporal_compression_ratio.bit_length() - 1          # up / down blocks that also resample time
        with torch.device(device if device is not None else "cpu"):
            self.encoder = _Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
            self.decoder = _Decoder(latent_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.to(torch.bfloat16)
        self._packed = {}
        self._ws = {}
        # 3 x 3 x 3 resnet convolutions as implicit GEMMs (bya_vae_conv3d); "0" = patch matrix + bya_gemm_bf16 (A/B, tests)
        self.implicit_conv = os.environ.get("BYA_VAE_IMPLICIT_CONV", "1") != "0"

    def forward(self, *a, **k):
        raise RuntimeError("use decode() / encode()")

    def init_synthetic(self, seed=0):
        """Random weights of the right shapes and scales (there is no checkpoint in the build / test environment)."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        with torch.no_grad():
            for name, p in self.named_parameters():
                if p.dim() == 1 and name.endswith("weight"):                  # GroupNorm gains
                    v = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
                elif p.dim() == 1:
                    v = 0.05 * torch.randn(p.shape, generator=g)
                else:
                    v = torch.randn(p.shape, generator=g) * (p[0].numel() ** -0.5)
                    if "conv_y" in name:
                        v = v * 0.3
                p.copy_(v.to(p.dtype))
                if "conv_y.conv.bias" in name:
                    p.add_(1.0)                                                # spatial gain around one
        self._packed = {}
        return self

    # ------------------------------------------------------------------------------------------ packing
    @staticmethod
    def _pad_k(k):
        return (k + 63) // 64 * 64

    def _pack_conv(self, key, weight, bias):
        """[Cout, Cin, (kt,) kh, kw] -> bf16 [Cout8, Kpad] with columns (kt, kh, kw, cin); Cout padded to a multiple of 8."""
        hit = self._packed.get(key)
        if hit is not None and hit[2] == (weight.data_ptr(), weight._version):
            return hit[0], hit[1]
        w = weight
        if w.dim() == 4:
            w = w[:, :, None]
        cout, cin = w.shape[:2]
        w2 = w.permute(0, 2, 3, 4, 1).reshape(cout, -1)
        K, cout4 = w2.shape[1], (cout + 7) // 8 * 8
        wp = torch.zeros(cout4, self._pad_k(K), dtype=torch.bfloat16, device=w.device)
        wp[:cout, :K] = w2
        bp = torch.zeros(cout4, dtype=torch.bfloat16, device=w.device)
        bp[:cout] = bias
        self._packed[key] = (wp, bp, (weight.data_ptr(), weight._version))
        return wp, bp

    def _pack_yb(self, key, sn):
        """conv_y | conv_b of a SpatialNorm3D stacked: [2 C, Kpad(zc)], bias [2 C]."""
        hit = self._packed.get(key)
        wy = sn.conv_y.conv.weight
        if hit is not None and hit[2] == (wy.data_ptr(), wy._version):
            return hit[0], hit[1]
        C, zc = wy.shape[0], wy.shape[1]
        wp = torch.zeros(2 * C, self._pad_k(zc), dtype=torch.bfloat16, device=wy.device)
        wp[:C, :zc] = wy.reshape(C, zc)
        wp[C:, :zc] = sn.conv_b.conv.weight.reshape(C, zc)
        bp = torch.cat([sn.conv_y.conv.bias, sn.conv_b.conv.bias]).contiguous()
        self._packed[key] = (wp, bp, (wy.data_ptr(), wy._version))
        return wp, bp

    def _buf(self, name, *shape, dtype=torch.bfloat16):
        t = self._ws.get(name)
        n = 1
        for s in shape:
            n *= s
        if t is None or t.numel() < n or t.dtype != dtype:
            t = self._ws[name] = torch.empty(n, dtype=dtype, device=self._dev)
        return t[:n].view(*shape)

    # ------------------------------------------------------------------------------------------ building blocks
    def _conv(self, key, mod, x, cache=None, res=None, KT=3, stride=1, pad=1, up=False, tmode=0, out=None):
        """x [T, H, W, C] -> [To, Ho, Wo, Cout] (+ res); returns (y, new_cache)."""
        conv = mod.conv if hasattr(mod, "conv") else mod
        wp, bp = self._pack_conv(key, conv.weight, conv.bias)
        T, H, W, C = x.shape
        cout, cout4, Kpad = conv.weight.shape[0], wp.shape[0], wp.shape[1]
        if up:
            To = T if tmode == 0 else (2 * T if tmode == 1 else 2 * T - 1)
            Ho, Wo = 2 * H, 2 * W
        elif stride == 2:
            To, Ho, Wo = T, (H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1
        else:
            To, Ho, Wo = T, H, W
        y = out if out is not None else torch.empty(To, Ho, Wo, cout4, dtype=torch.bfloat16, device=x.device)
        rows_f = Ho * Wo
        nt_max = max(1, int(self.PATCH_BYTES // (rows_f * Kpad * 2)))
        y2 = y.view(To * rows_f, cout4)
        r2 = None if res is None else res.reshape(To * rows_f, cout4)
        for t0 in range(0, To, nt_max):
            nt = min(nt_max, To - t0)
            patches = self._buf("patches", nt * rows_f, Kpad)
            ops.vae_patches(x, cache, patches, KT, stride, pad, up, tmode, Ho, Wo, t0, nt)
            sl = slice(t0 * rows_f, (t0 + nt) * rows_f)
            ops.gemm(patches, wp, y2[sl], bias=bp, res=None if r2 is None else r2[sl])
        new_cache = None
        if KT == 3:                                           # last two frames of [cache | x]
            if T >= 2:
                new_cache = x[-2:].clone()
            else:
                first = cache[1:] if cache is not None else x[:1]
                new_cache = torch.cat([first, x[:1]], 0).contiguous()
        if cout4 != cout:
            y = y[..., :cout]
        return y, new_cache

    def _norm(self, key, norm, x, zctx, act="silu", out_pad=None):
        """GroupNorm (+ spatial modulation from the latent chunk ``zctx`` = (z64 rows, (Tz, hz, wz))) + activation.
        ``out_pad``: write into this zero-padded conv input [T + 2, H + 2, W + 2, C] instead of a fresh tensor."""
        T, H, W, C = x.shape
        gn = norm.norm_layer if hasattr(norm, "norm_layer") else norm
        sums = self._buf("gn_sums", 2 * gn.num_groups, dtype=torch.float32)
        part = self._buf("gn_partial", (T * H * W + 511) // 512 * gn.num_groups * 2, dtype=torch.float32)
        ops.vae_groupnorm_stats(x.view(-1, C), sums, gn.num_groups, part)
        y = out_pad if out_pad is not None else torch.empty_like(x)
        pad = out_pad is not None
        if hasattr(norm, "norm_layer"):
            z64, lat = zctx
            wp, bp = self._pack_yb(key, norm)
            zyb = self._buf("zyb", z64.shape[0], 2 * C)
            ops.gemm(z64, wp, zyb, bias=bp)
            tmode = 2 if (T > 1 and T % 2 == 1) else 1
            ops.vae_norm_act(x, y, sums, gn.weight, gn.bias, gn.num_groups, act=act, eps=gn.eps, zy=zyb[:, :C], zb=zyb[:, C:],
                             latent_shape=lat, tmode=tmode, out_pad=pad)
        else:
            ops.vae_norm_act(x, y, sums, gn.weight, gn.bias, gn.num_groups, act=act, eps=gn.eps, out_pad=pad)
        return y

    def _pad_buf(self, T, H, W, C):
        """Zero-padded conv input [T + 2, H + 2, W + 2, C] (bya_vae_conv3d): one buffer per shape, zero-filled when it is
        created -- the border is never written afterwards (the norm kernel fills the interior, the context frames are
        whole padded frames)."""
        key = ("pad", T, H, W, C)
        t = self._ws.get(key)
        if t is None:
            t = self._ws[key] = torch.zeros(T + 2, H + 2, W + 2, C, dtype=torch.bfloat16, device=self._dev)
        return t

    def _upsample_conv(self, key, mod, x, tmode):
        """CogVideoXUpsample3D: nearest up-sampling (never kept at 9x as patches: written once, zero-padded) + its per-frame
        3 x 3 convolution as an implicit GEMM."""
        T, H, W, C = x.shape
        conv = mod.conv if hasattr(mod, "conv") else mod
        if not self.implicit_conv or C not in (128, 256, 512):
            return self._conv(key, mod, x, KT=1, up=True, tmode=tmode)[0]
        To = T if tmode == 0 else (2 * T if tmode == 1 else 2 * T - 1)
        kb = ("pad1", To, 2 * H, 2 * W, C)
        ypad = self._ws.get(kb)
        if ypad is None:
            ypad = self._ws[kb] = torch.zeros(To, 2 * H + 2, 2 * W + 2, C, dtype=torch.bfloat16, device=self._dev)
        ops.vae_upsample_pad(x, ypad, tmode)
        wp, bp = self._pack_conv(key, conv.weight, conv.bias)
        cout = conv.weight.shape[0]
        y = torch.empty(To, 2 * H, 2 * W, wp.shape[0], dtype=torch.bfloat16, device=x.device)
        ops.vae_conv3d(ypad, wp, bp, y, KT=1)
        return y if wp.shape[0] == cout else y[..., :cout]

    def _norm_conv(self, key, norm, conv_mod, x, zctx, cache, res=None):
        """GroupNorm (+ modulation) + SiLU -> causal 3 x 3 x 3 convolution (+ res), the convolution as an implicit GEMM: the
        norm kernel writes the zero-padded conv input, no patch matrix exists.  ``cache``: the two context frames in PADDED
        form [2, H + 2, W + 2, C] (or None: the first frame twice).  Returns (y, new_cache)."""
        T, H, W, C = x.shape
        conv = conv_mod.conv if hasattr(conv_mod, "conv") else conv_mod
        cout = conv.weight.shape[0]
        if not self.implicit_conv or C not in (128, 256, 512) or conv.weight.shape[2:] != (3, 3, 3):
            h = self._norm(key + ".n", norm, x, zctx)
            if cache is not None and cache.shape[1] == H + 2:                 # (a padded cache from the other path)
                cache = cache[:, 1:-1, 1:-1].contiguous()
            return self._conv(key + ".c", conv_mod, h, cache, res=res)
        xpad = self._pad_buf(T, H, W, C)
        self._norm(key + ".n", norm, x, zctx, out_pad=xpad)
        if cache is None:
            xpad[0].copy_(xpad[2])
            xpad[1].copy_(xpad[2])
        else:
            if cache.shape[1] == H:                                           # (an unpadded cache from the other path)
                xpad[:2, 1:-1, 1:-1].copy_(cache)
            else:
                xpad[:2].copy_(cache)
        wp, bp = self._pack_conv(key + ".c", conv.weight, conv.bias)
        cout8 = wp.shape[0]                                   # (conv_out: 3 channels computed as 8)
        y = torch.empty(T, H, W, cout8, dtype=torch.bfloat16, device=x.device)
        ops.vae_conv3d(xpad, wp, bp, y, res=res)
        return (y if cout8 == cout else y[..., :cout]), xpad[T:T + 2].clone()

    def _resnet(self, key, blk, x, zctx, cache):
        cache = cache or {}
        new = {}
        h, new["conv1"] = self._norm_conv(key + ".1", blk.norm1, blk.conv1, x, zctx, cache.get("conv1"))
        if blk.cin != blk.cout:
            wp, bp = self._pack_conv(key + ".sc", blk.conv_shortcut.weight, blk.conv_shortcut.bias)
            sc = torch.empty(*x.shape[:3], blk.cout, dtype=torch.bfloat16, device=x.device)
            if wp.shape[1] != blk.cin:                       # K padded: pad the rows once
                xp = self._buf("sc_in", x.shape[0] * x.shape[1] * x.shape[2], wp.shape[1])
                xp.zero_()
                xp[:, :blk.cin] = x.view(-1, blk.cin)
                ops.gemm(xp, wp, sc.view(-1, blk.cout), bias=bp)
            else:
                ops.gemm(x.view(-1, blk.cin), wp, sc.view(-1, blk.cout), bias=bp)
        else:
            sc = x
        h, new["conv2"] = self._norm_conv(key + ".2", blk.norm2, blk.conv2, h, zctx, cache.get("conv2"), res=sc)
        return h, new

    # ------------------------------------------------------------------------------------------ decode
    def _decode_chunk(self, z, cache):
        """z [Tz, hz, wz, zc] channels-last -> frames [To, 8 hz, 8 wz, 3]."""
        d = self.decoder
        cache = cache or {}
        new = {}
        Tz, hz, wz, zc = z.shape
        kz = self._pad_k(zc)
        z64 = torch.zeros(Tz * hz * wz, kz, dtype=torch.bfloat16, device=z.device)
        z64[:, :zc] = z.view(-1, zc)
        zctx = (z64, (Tz, hz, wz))
        h, new["conv_in"] = self._conv("d.conv_in", d.conv_in, z, cache.get("conv_in"))
        for j, blk in enumerate(d.mid_block.resnets):
            h, new[f"mid{j}"] = self._resnet(f"d.mid{j}", blk, h, zctx, cache.get(f"mid{j}"))
        for i, ub in enumerate(d.up_blocks):
            for j, blk in enumerate(ub.resnets):
                h, new[f"up{i}_{j}"] = self._resnet(f"d.up{i}_{j}", blk, h, zctx, cache.get(f"up{i}_{j}"))
            if hasattr(ub, "upsamplers"):
                T = h.shape[0]
                tmode = 0 if (i >= self.n_time or T == 1) else (2 if T % 2 == 1 else 1)
                h = self._upsample_conv(f"d.up{i}.us", ub.upsamplers[0], h, tmode)
        y, new["conv_out"] = self._norm_conv("d.out", d.norm_out, d.conv_out, h, zctx, cache.get("conv_out"))
        return y, new

    @torch.no_grad()
    def decode(self, z, return_dict=True):
        """z [B, zc, T, h, w] -> ``.sample`` [B, 3, 4 (T - 1) + 1, 8 h, 8 w] (diffusers' ``_decode`` chunking: 2 latent frames
        at a time, the first chunk takes the remainder)."""
        p = next(self.parameters())
        if not p.is_cuda:
            raise RuntimeError("the VAE runs on the HIP kernels only: move it to a GPU (there is no CPU fallback)")
        self._dev = p.device
        B, zc, T = z.shape[:3]
        fb = self.num_latent_frames_batch_size
        n, rem = max(T // fb, 1), T % fb
        outs = []
        with ops.pinned_stream():
            for b in range(B):
                zl = z[b].to(self._dev, torch.bfloat16).permute(1, 2, 3, 0).contiguous()        # [T, h, w, zc]
                cache, frames = None, []
                for i in range(n):
                    a, e = fb * i + (0 if i == 0 else rem), fb * (i + 1) + rem
                    y, cache = self._decode_chunk(zl[a:e].contiguous(), cache)
                    frames.append(y)
                outs.append(torch.cat(frames, 0).permute(3, 0, 1, 2))                          # [3, F, H, W]
        sample = torch.stack(outs, 0).contiguous()
        return SimpleNamespace(sample=sample) if return_dict else (sample,)

    # ------------------------------------------------------------------------------------------ encode (one frame)
    @torch.no_grad()
    def encode(self, x, return_dict=True):
        """x [B, 3, 1, H, W] (the reference encodes the conditioning FRAME, ``image.unsqueeze(2)``) -> ``.latent_dist``."""
        p = next(self.parameters())
        if not p.is_cuda:
            raise RuntimeError("the VAE runs on the HIP kernels only: move it to a GPU (there is no CPU fallback)")
        if x.shape[2] != 1:
            raise NotImplementedError("the reference pipeline encodes single conditioning frames (models/pipeline_"
                                      "bindyouravatar.py:406-421); clips need the temporal pooling of CogVideoXDownsample3D")
        self._dev = p.device
        e = self.encoder
        moments = []
        with ops.pinned_stream():
            for b in range(x.shape[0]):
                h = x[b].to(self._dev, torch.bfloat16).permute(1, 2, 3, 0).contiguous()          # [1, H, W, 3]
                h, _ = self._conv("e.conv_in", e.conv_in, h)
                for i, db in enumerate(e.down_blocks):
                    for j, blk in enumerate(db.resnets):
                        h, _ = self._resnet(f"e.down{i}_{j}", blk, h, None, None)
                    if hasattr(db, "downsamplers"):
                        h, _ = self._conv(f"e.down{i}.ds", db.downsamplers[0], h, KT=1, stride=2, pad=0)
                        h = h.contiguous()
                for j, blk in enumerate(e.mid_block.resnets):
                    h, _ = self._resnet(f"e.mid{j}", blk, h, None, None)
                h = self._norm("e.norm_out", e.norm_out, h, None)
                h, _ = self._conv("e.conv_out", e.conv_out, h)
                moments.append(h.permute(3, 0, 1, 2))                                           # [2 zc, 1, h, w]
        dist = DiagonalGaussian(torch.stack(moments, 0).contiguous())
        return SimpleNamespace(latent_dist=dist) if return_dict else (dist,)
