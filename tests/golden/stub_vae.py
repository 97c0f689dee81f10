"""A deterministic stand-in "VAE" for the pipeline-plumbing fixtures (tests/golden/ref_prepare_latents.npz): the moments
of a frame are a fixed 1 x 1 channel mix of its 8 x 8 average-pooled pixels.  It only has to be a function both sides can
evaluate bit for bit -- the reference's ``prepare_latents`` (run by tests/golden/make_golden.py --case prepare_latents with
the restated diffusers distribution) and this package's pipeline (tests/test_pipeline_cpu.py with the product's
``DiagonalGaussian``); the real encoder has its own tests (tests/test_vae_gpu.py)."""
from types import SimpleNamespace

import torch
import torch.nn.functional as F


class StubVAE:
    def __init__(self, dist_cls, latent_channels=16, scaling_factor=0.7):
        g = torch.Generator().manual_seed(1234)
        self.mix = (torch.randn(2 * latent_channels, 3, generator=g) * 0.5).to(torch.bfloat16).float()
        self.dist_cls = dist_cls
        self.config = SimpleNamespace(block_out_channels=(1, 2, 3, 4), temporal_compression_ratio=4,
                                      scaling_factor=scaling_factor, latent_channels=latent_channels)
        self.calls = 0

    def encode(self, x):                       # [B, 3, F, H, W] -> latent_dist over [B, 2C, F, H / 8, W / 8]
        self.calls += 1
        B, C, Fr, H, W = x.shape
        p = F.avg_pool2d(x.permute(0, 2, 1, 3, 4).reshape(B * Fr, C, H, W).float(), 8)
        m = torch.einsum("oc,nchw->nohw", self.mix, p).reshape(B, Fr, -1, H // 8, W // 8).permute(0, 2, 1, 3, 4)
        return SimpleNamespace(latent_dist=self.dist_cls(m.to(x.dtype).contiguous()))


def prepare_latents_cases():
    """(name, dtype, batch, kps?, bg?, generator kind) of the fixture; inputs are regenerated from the seeds below."""
    return [("plain_f32", torch.float32, 2, False, False, "one"), ("kps_f32", torch.float32, 2, True, False, "one"),
            ("kps_bg_bf16", torch.bfloat16, 2, True, True, "one"), ("bg_list_bf16", torch.bfloat16, 2, False, True, "list"),
            ("kps_given_latents_bf16", torch.bfloat16, 1, True, True, "one")]


def prepare_latents_inputs(name, dtype, batch, seed=7):
    g = torch.Generator().manual_seed(seed + sum(map(ord, name)))
    mk = lambda: (torch.rand(batch, 3, 32, 48, generator=g) * 2 - 1).to(dtype)
    return dict(image=mk(), kps=mk(), bg=mk(), latents=torch.randn(batch, 3, 16, 4, 6, generator=g).to(dtype))
