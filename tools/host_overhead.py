#!/usr/bin/env python3
"""Host-side cost of enqueueing one step (no synchronisation inside the timed region) vs the GPU time of the step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bind_your_avatar_implementation_amd import BindyouravatarTransformer3DModel
from bind_your_avatar_implementation_amd.synth import synth_inputs
dev = torch.device("cuda:0")
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 42
model = BindyouravatarTransformer3DModel(**dict(bench.MODEL_KW, num_layers=layers), device=dev).init_synthetic(seed=0, fast=True)
inp = synth_inputs(batch=1, seed=0, device="cpu")
inp = {k: (v.to(dev, torch.bfloat16) if torch.is_tensor(v) and v.is_floating_point() else (v.to(dev) if torch.is_tensor(v) else v)) for k, v in inp.items()}
inp["image_rotary_emb"] = tuple(t.to(dev, torch.float32) for t in inp["image_rotary_emb"])
inp["id_cond"] = [t.to(dev, torch.bfloat16) for t in inp["id_cond"]]
inp["id_vit_hidden"] = [[t.to(dev, torch.bfloat16) for t in l] for l in inp["id_vit_hidden"]]
for _ in range(2):
    model(**inp)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    model(**inp)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.1f} ms, step {1e3*(t2-t0):.1f} ms", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); model(**inp); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
