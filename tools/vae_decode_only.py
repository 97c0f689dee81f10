"""One full-size VAE decode (13 x 60 x 90 latents -> 49 frames of 480 x 720) for profiling.  usage: python tools/vae_decode_only.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bind_your_avatar_implementation_amd import BindyouravatarVAE
dev = torch.device("cuda:0")
vae = BindyouravatarVAE(device=dev).init_synthetic(9)
z = torch.randn(1, 16, 13, 60, 90, generator=torch.Generator().manual_seed(10)).to(dev)
vae.decode(z[:, :, :3])
torch.cuda.synchronize()
t0 = time.time()
out = vae.decode(z).sample
torch.cuda.synchronize()
print(f"decode {tuple(out.shape)} in {time.time() - t0:.3f} s", flush=True)
