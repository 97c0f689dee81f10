"""The step-invariant conditioning alone (LocalFacialExtractor, perceiver to_kv / router keys, AudioProjModel + audio K/V): what the
engine recomputes every step on its side stream.  For rocprofv3 --kernel-trace --stats, or stand-alone timing.
python tools/invariants_only.py [repeats]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = sys.argv[:2]
import torch
from bind_your_avatar_implementation_amd import BindyouravatarTransformer3DModel
from bind_your_avatar_implementation_amd.synth import synth_inputs
dev = torch.device("cuda:0")
from bench import MODEL_KW
model = BindyouravatarTransformer3DModel(**MODEL_KW, device=dev).init_synthetic(seed=0, fast=True)
inp = synth_inputs(batch=1, seed=0)
to = lambda t: t.to(dev, torch.bfloat16) if t.is_floating_point() else t.to(dev)
id_cond = [to(t) for t in inp["id_cond"]]
vit = [[to(t) for t in l] for l in inp["id_vit_hidden"]]
audio = to(inp["audio_embeds"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for i in range(n):
    torch.cuda.synchronize(); t0 = time.time()
    model.precompute_conditioning(id_cond, vit, audio, latent_frames=13)
    torch.cuda.synchronize()
    print(f"invariants pass {i}: {1e3 * (time.time() - t0):.2f} ms", flush=True)
    model.release_conditioning()
