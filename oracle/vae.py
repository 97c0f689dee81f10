"""TEST INFRASTRUCTURE (oracle): CPU restatement of the video VAE either side of the denoise loop.

The reference calls ``self.vae.encode(image)`` on ONE conditioning frame (``models/pipeline_bindyouravatar.py:406-421``,
``image.unsqueeze(2)``: F = 1) and ``self.vae.decode(latents).sample`` on the finished latents (``:461-466``); ``self.vae``
is diffusers' ``AutoencoderKLCogVideoX`` (``infer.py`` loads it from the CogVideoX-5B-I2V checkpoint), i.e. an
UN-VENDORED THIRD-PARTY layer: ``diffusers==0.34.0.dev0`` (``requirements.txt:23``), not installed here, no network.
The reference holds no tests or golden vectors at this boundary, so **parity is unpinned**: what follows restates the
published algorithm of that class (``autoencoder_kl_cogvideox.py`` of diffusers 0.31 - 0.34) and is anchored on the
reference's two call sites only.  What the restatement encodes:

* ``CogVideoXCausalConv3d`` (pad_mode "first"): zero padding in space (inside the convolution), CAUSAL in time -- the
  input is extended at the front by the previous chunk's last ``k_t - 1`` frames (``conv_cache``) or, for the first
  chunk, by copies of its first frame; the new cache is the last ``k_t - 1`` frames of the extended input.
* ``CogVideoXSpatialNorm3D`` (decoder): ``GroupNorm(32, eps 1e-6)(f) * conv_y(zq') + conv_b(zq')`` with 1x1x1 convolutions
  of the latent ``zq`` resized to ``f`` by nearest neighbour -- first frame and the rest resized separately when ``f`` has
  an odd number (> 1) of frames.
* ``CogVideoXResnetBlock3D``: norm -> SiLU -> conv 3x3x3 -> norm -> SiLU -> conv 3x3x3, + input (1x1x1 convolution when
  the channel count changes).
* ``CogVideoXUpsample3D``: nearest x2 in space (and in time for the first two blocks: first frame kept single when the
  chunk has an odd number (> 1) of frames), then a per-frame 3x3 convolution.  ``CogVideoXDownsample3D``: average over
  frame pairs (first frame kept when the count is odd) for the first two blocks, zero pad (0, 1, 0, 1), per-frame 3x3
  convolution of stride 2.
* Decoder: conv_in 16 -> 512, mid block (2 resnets), four up blocks of 4 resnets (512, 256, 256, 128; up-sampling after
  the first three, temporal in the first two), SpatialNorm, SiLU, conv_out -> 3.  Encoder: conv_in 3 -> 128, four down
  blocks of 3 resnets (128, 256, 256, 512), mid block, GroupNorm, SiLU, conv_out -> 2 x 16 (mean | logvar).
* ``_decode`` walks the latent frames in chunks of ``num_latent_frames_batch_size = 2`` (the first chunk takes the
  remainder too: 13 frames = 3 + 2 + 2 + 2 + 2 + 2), carrying every causal convolution's cache from chunk to chunk.
  GroupNorm statistics are therefore PER CHUNK -- that is the published behaviour, and the product reproduces it.
* ``decode_latents``: ``latents.permute(0, 2, 1, 3, 4) / scaling_factor`` in, ``[B, 3, F, H, W]`` out (``:461-466``);
  ``prepare_latents``: the posterior SAMPLE of the encoded frame times ``scaling_factor`` (``:406-424``).

Only ``tests/`` may import this module (it is the checker, never the product)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class CausalConv3d(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.kt = k
        self.conv = nn.Conv3d(cin, cout, (k, k, k), padding=(0, (k - 1) // 2, (k - 1) // 2))

    def forward(self, x, cache=None):
        if self.kt > 1:
            front = [cache] if cache is not None else [x[:, :, :1]] * (self.kt - 1)
            x = torch.cat(front + [x], dim=2)
        new_cache = x[:, :, -self.kt + 1:].clone() if self.kt > 1 else None
        return self.conv(x), new_cache


def resize_like(zq, f):
    """nearest-neighbour resize of the latent to f's (T, H, W); first frame apart when f has an odd (> 1) frame count."""
    if f.shape[2] > 1 and f.shape[2] % 2 == 1:
        z_first = F.interpolate(zq[:, :, :1], size=(1,) + tuple(f.shape[-2:]))
        z_rest = F.interpolate(zq[:, :, 1:], size=(f.shape[2] - 1,) + tuple(f.shape[-2:]))
        return torch.cat([z_first, z_rest], dim=2)
    return F.interpolate(zq, size=tuple(f.shape[-3:]))


class SpatialNorm3D(nn.Module):
    def __init__(self, ch, zq_ch, groups):
        super().__init__()
        self.norm_layer = nn.GroupNorm(groups, ch, eps=1e-6, affine=True)
        self.conv_y = CausalConv3d(zq_ch, ch, 1)
        self.conv_b = CausalConv3d(zq_ch, ch, 1)

    def forward(self, f, zq):
        z = resize_like(zq, f)
        return self.norm_layer(f) * self.conv_y(z)[0] + self.conv_b(z)[0]


class ResnetBlock3D(nn.Module):
    def __init__(self, cin, cout, groups, zq_ch=None):
        super().__init__()
        self.spatial = zq_ch is not None
        if self.spatial:
            self.norm1, self.norm2 = SpatialNorm3D(cin, zq_ch, groups), SpatialNorm3D(cout, zq_ch, groups)
        else:
            self.norm1, self.norm2 = nn.GroupNorm(groups, cin, eps=1e-6), nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv1, self.conv2 = CausalConv3d(cin, cout, 3), CausalConv3d(cout, cout, 3)
        self.conv_shortcut = nn.Conv3d(cin, cout, 1) if cin != cout else None

    def forward(self, x, zq, cache):
        cache = cache or {}
        new = {}
        h = self.norm1(x, zq) if self.spatial else self.norm1(x)
        h, new["conv1"] = self.conv1(F.silu(h), cache.get("conv1"))
        h = self.norm2(h, zq) if self.spatial else self.norm2(h)
        h, new["conv2"] = self.conv2(F.silu(h), cache.get("conv2"))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return h + x, new


class Upsample3D(nn.Module):
    def __init__(self, ch, compress_time):
        super().__init__()
        self.compress_time = compress_time
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        if self.compress_time:
            if x.shape[2] > 1 and x.shape[2] % 2 == 1:
                first = F.interpolate(x[:, :, 0], scale_factor=2.0)[:, :, None]
                rest = F.interpolate(x[:, :, 1:], scale_factor=2.0)
                x = torch.cat([first, rest], dim=2)
            elif x.shape[2] > 1:
                x = F.interpolate(x, scale_factor=2.0)
            else:
                x = F.interpolate(x.squeeze(2), scale_factor=2.0)[:, :, None]
        else:
            b, c, t, h, w = x.shape
            x = F.interpolate(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w), scale_factor=2.0)
            x = x.reshape(b, t, c, 2 * h, 2 * w).permute(0, 2, 1, 3, 4)
        b, c, t, h, w = x.shape
        y = self.conv(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w))
        return y.reshape(b, t, -1, h, w).permute(0, 2, 1, 3, 4)


class Downsample3D(nn.Module):
    def __init__(self, ch, compress_time):
        super().__init__()
        self.compress_time = compress_time
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=0)

    def forward(self, x):
        if self.compress_time:
            b, c, t, h, w = x.shape
            y = x.permute(0, 3, 4, 1, 2).reshape(b * h * w, c, t)
            if t % 2 == 1:
                first, rest = y[..., 0], y[..., 1:]
                if rest.shape[-1] > 0:
                    rest = F.avg_pool1d(rest, kernel_size=2, stride=2)
                y = torch.cat([first[..., None], rest], dim=-1)
            else:
                y = F.avg_pool1d(y, kernel_size=2, stride=2)
            x = y.reshape(b, h, w, c, y.shape[-1]).permute(0, 3, 4, 1, 2)
        x = F.pad(x, (0, 1, 0, 1))
        b, c, t, h, w = x.shape
        y = self.conv(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w))
        return y.reshape(b, t, -1, y.shape[-2], y.shape[-1]).permute(0, 2, 1, 3, 4)


class _Block(nn.Module):
    """``resnets`` (+ ``upsamplers`` / ``downsamplers``): the module tree -- hence the state-dict keys -- of diffusers'
    CogVideoX{Mid,Up,Down}Block3D, so a checkpoint of ``AutoencoderKLCogVideoX`` loads into this restatement as it is."""

    def __init__(self, resnets, up=None, down=None):
        super().__init__()
        self.resnets = nn.ModuleList(resnets)
        if up is not None:
            self.upsamplers = nn.ModuleList([up])
        if down is not None:
            self.downsamplers = nn.ModuleList([down])


class Decoder3D(nn.Module):
    def __init__(self, latent_channels=16, out_channels=3, block_out_channels=(128, 256, 256, 512), layers_per_block=3,
                 groups=32, temporal_compression_ratio=4):
        super().__init__()
        rev = list(reversed(block_out_channels))
        self.conv_in = CausalConv3d(latent_channels, rev[0], 3)
        self.mid_block = _Block([ResnetBlock3D(rev[0], rev[0], groups, latent_channels) for _ in range(2)])
        self.up_blocks = nn.ModuleList()
        n_time = temporal_compression_ratio.bit_length() - 1
        cin = rev[0]
        for i, cout in enumerate(rev):
            self.up_blocks.append(_Block([ResnetBlock3D(cin if j == 0 else cout, cout, groups, latent_channels)
                                          for j in range(layers_per_block + 1)],
                                         up=Upsample3D(cout, compress_time=i < n_time) if i < len(rev) - 1 else None))
            cin = cout
        self.norm_out = SpatialNorm3D(rev[-1], latent_channels, groups)
        self.conv_out = CausalConv3d(rev[-1], out_channels, 3)

    def forward(self, z, cache=None):
        cache = cache or {}
        new = {}
        h, new["conv_in"] = self.conv_in(z, cache.get("conv_in"))
        for j, blk in enumerate(self.mid_block.resnets):
            h, new[f"mid{j}"] = blk(h, z, cache.get(f"mid{j}"))
        for i, ub in enumerate(self.up_blocks):
            for j, blk in enumerate(ub.resnets):
                h, new[f"up{i}_{j}"] = blk(h, z, cache.get(f"up{i}_{j}"))
            if hasattr(ub, "upsamplers"):
                h = ub.upsamplers[0](h)
        h = F.silu(self.norm_out(h, z))
        h, new["conv_out"] = self.conv_out(h, cache.get("conv_out"))
        return h, new


class Encoder3D(nn.Module):
    def __init__(self, in_channels=3, latent_channels=16, block_out_channels=(128, 256, 256, 512), layers_per_block=3,
                 groups=32, temporal_compression_ratio=4):
        super().__init__()
        self.conv_in = CausalConv3d(in_channels, block_out_channels[0], 3)
        self.down_blocks = nn.ModuleList()
        n_time = temporal_compression_ratio.bit_length() - 1
        cin = block_out_channels[0]
        for i, cout in enumerate(block_out_channels):
            self.down_blocks.append(_Block([ResnetBlock3D(cin if j == 0 else cout, cout, groups) for j in range(layers_per_block)],
                                           down=Downsample3D(cout, compress_time=i < n_time)
                                           if i < len(block_out_channels) - 1 else None))
            cin = cout
        self.mid_block = _Block([ResnetBlock3D(cin, cin, groups) for _ in range(2)])
        self.norm_out = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv_out = CausalConv3d(cin, 2 * latent_channels, 3)

    def forward(self, x, cache=None):
        cache = cache or {}
        new = {}
        h, new["conv_in"] = self.conv_in(x, cache.get("conv_in"))
        for i, db in enumerate(self.down_blocks):
            for j, blk in enumerate(db.resnets):
                h, new[f"down{i}_{j}"] = blk(h, None, cache.get(f"down{i}_{j}"))
            if hasattr(db, "downsamplers"):
                h = db.downsamplers[0](h)
        for j, blk in enumerate(self.mid_block.resnets):
            h, new[f"mid{j}"] = blk(h, None, cache.get(f"mid{j}"))
        h = F.silu(self.norm_out(h))
        h, new["conv_out"] = self.conv_out(h, cache.get("conv_out"))
        return h, new


class OracleVAE(nn.Module):
    """``AutoencoderKLCogVideoX`` as the reference uses it: ``encode(x).latent_dist`` (mean | logvar of one frame or a
    clip) and ``decode(z).sample``; no quant / post-quant convolution, no spatial tiling (the reference never enables it)."""

    def __init__(self, scaling_factor=0.7, latent_frames_per_chunk=2, sample_frames_per_chunk=8, **kw):
        super().__init__()
        self.scaling_factor = scaling_factor
        self.lchunk, self.schunk = latent_frames_per_chunk, sample_frames_per_chunk
        self.encoder, self.decoder = Encoder3D(**kw), Decoder3D(**{k: v for k, v in kw.items() if k != "in_channels"})

    @torch.no_grad()
    def decode(self, z):
        """z [B, C, T, h, w] -> [B, 3, 4 (T - 1) + 1, 8 h, 8 w]."""
        T = z.shape[2]
        n, rem = max(T // self.lchunk, 1), T % self.lchunk
        cache, out = None, []
        for i in range(n):
            a = self.lchunk * i + (0 if i == 0 else rem)
            b = self.lchunk * (i + 1) + rem
            y, cache = self.decoder(z[:, :, a:b], cache)
            out.append(y)
        return torch.cat(out, dim=2)

    @torch.no_grad()
    def encode_moments(self, x):
        """x [B, 3, F, H, W] -> (mean, logvar), each [B, C, (F - 1) / 4 + 1, H / 8, W / 8]."""
        Fr = x.shape[2]
        n, rem = max(Fr // self.schunk, 1), Fr % self.schunk
        cache, out = None, []
        for i in range(n):
            a = self.schunk * i + (0 if i == 0 else rem)
            b = self.schunk * (i + 1) + rem
            y, cache = self.encoder(x[:, :, a:b], cache)
            out.append(y)
        return torch.cat(out, dim=2).chunk(2, dim=1)

    def decode_latents(self, latents):
        """models/pipeline_bindyouravatar.py:461-466."""
        return self.decode(latents.permute(0, 2, 1, 3, 4) / self.scaling_factor)

    def encode_image_latents(self, image, noise=None):
        """models/pipeline_bindyouravatar.py:406-424 for one conditioning frame: image [B, 3, H, W] -> scaled posterior
        sample [B, 1, C, H / 8, W / 8] (noise = the standard-normal draw of ``latent_dist.sample``; None = the mode)."""
        mean, logvar = self.encode_moments(image.unsqueeze(2))
        z = mean if noise is None else mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise
        return self.scaling_factor * z.permute(0, 2, 1, 3, 4)


class DiagonalGaussianDistribution:
    """diffusers ``models/autoencoders/vae.py`` DiagonalGaussianDistribution as published (0.31 - 0.34), the object behind
    ``vae.encode(x).latent_dist`` that the reference samples once per image (models/pipeline_bindyouravatar.py:177-181,
    :409-420): ``mean | logvar = chunk(parameters, 2, dim=1)``, logvar clamped to [-30, 20], ``std = exp(0.5 logvar)``;
    ``sample`` draws ``randn_tensor(mean.shape, generator, device=parameters.device, dtype=parameters.dtype)`` -- on the
    generator's device when that is the CPU -- and returns ``mean + std * noise`` in the parameters' dtype.
    diffusers-owned: parity unpinned (no diffusers in this image)."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        return self.mean + self.std * randn_tensor(self.mean.shape, generator, self.parameters.device, self.parameters.dtype)

    def mode(self):
        return self.mean


def randn_tensor(shape, generator=None, device=None, dtype=None):
    """diffusers ``utils/torch_utils.py`` randn_tensor for the cases the reference pipeline reaches: one generator (drawn
    on ITS device, moved afterwards) or a list of per-sample generators."""
    device = torch.device(device) if device is not None else torch.device("cpu")
    if isinstance(generator, (list, tuple)):
        shape1 = (1,) + tuple(shape[1:])
        return torch.cat([torch.randn(shape1, generator=g, device=g.device, dtype=dtype).to(device) for g in generator], dim=0)
    rdev = device if generator is None else generator.device
    return torch.randn(tuple(shape), generator=generator, device=rdev, dtype=dtype).to(device)
