#!/usr/bin/env python3
"""End-to-end denoise LOOP (BASELINE config 1/2 wording: "50 DDIM steps") through pipeline.BindyouravatarPipeline:
conditioning computed once (precompute_conditioning), fused CFG-combine + scheduler step, latents in / latents out.
  python tools/pipeline_loop.py [steps] [guidance] [ddim|dpm] [vae]      (guidance > 1 -> CFG batch of 2; "vae": the clip is
  decoded to frames by BindyouravatarVAE inside the timed call; BYA_FP8_WEIGHTS=1 for the fp8 engine)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bind_your_avatar_implementation_amd import BindyouravatarTransformer3DModel
from bind_your_avatar_implementation_amd.pipeline import BindyouravatarPipeline, DDIMScheduler, DPMScheduler
from bind_your_avatar_implementation_amd.synth import synth_inputs
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
guidance = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
sched = DPMScheduler() if (len(sys.argv) > 3 and sys.argv[3] == "dpm") else DDIMScheduler()
dev = torch.device("cuda:0")
model = BindyouravatarTransformer3DModel(**bench.MODEL_KW, device=dev).init_synthetic(seed=0, fast=True)
inp = synth_inputs(batch=1, seed=0, device="cpu")
bf = lambda t: t.to(dev, torch.bfloat16)
lat = bf(inp["hidden_states"][:, :, :16]).contiguous()
img = bf(inp["hidden_states"][:, :, 16:32]).contiguous()
with_vae = len(sys.argv) > 4 and sys.argv[4] == "vae"
vae = None
if with_vae:
    from bind_your_avatar_implementation_amd import BindyouravatarVAE
    vae = BindyouravatarVAE(device=dev).init_synthetic(9)
pipe = BindyouravatarPipeline(model, scheduler=sched, vae=vae)
kw = dict(height=480, width=720, num_frames=49, num_inference_steps=steps, guidance_scale=guidance, latents=lat,
          prompt_embeds=bf(inp["encoder_hidden_states"]), negative_prompt_embeds=torch.zeros_like(bf(inp["encoder_hidden_states"])),
          image_latents=img, image_bg_latents=img, id_vit_hidden=[[bf(t) for t in l] for l in inp["id_vit_hidden"]],
          id_cond=[bf(t) for t in inp["id_cond"]], audio_embs=bf(inp["audio_embeds"]), af_matrix=bf(inp["af_matrix"]),
          generator=torch.Generator(device=dev).manual_seed(0), output_type="pt" if with_vae else "latent")
pipe(**dict(kw, num_inference_steps=2))
torch.cuda.synchronize()
t0 = time.perf_counter()
out = pipe(**kw).frames
torch.cuda.synchronize()
dt = time.perf_counter() - t0
assert torch.isfinite(out.float()).all()
print(f"{steps} steps, guidance {guidance} ({type(sched).__name__}){' + VAE decode' if with_vae else ''}"
      f"{' [fp8 weights]' if os.environ.get('BYA_FP8_WEIGHTS') == '1' else ''}: {dt:.2f} s = {steps/dt:.3f} steps/s "
      f"({dt/steps*1e3:.1f} ms/step, batch {2 if guidance > 1 else 1}); output {tuple(out.shape)} std {out.float().std():.3f}")
