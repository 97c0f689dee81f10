#!/usr/bin/env python3
"""Where the GEMM time of one denoise step goes, by shape: runs the headline step (bench.py's model and inputs) with the
per-launch HIP-event timers keyed by (batch x M x N x K, epilogue) and prints launches / ms per step / TFLOP/s per shape,
sorted by time.  `python tools/gemm_breakdown.py [--out gpurun_out/gemm_breakdown.json]`"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import MODEL_KW
from bind_your_avatar_implementation_amd import BindyouravatarTransformer3DModel, ops
from bind_your_avatar_implementation_amd.synth import synth_inputs


def build_model_and_inputs(dev):
    model = BindyouravatarTransformer3DModel(**MODEL_KW, device=dev).init_synthetic(seed=0, fast=True)
    inp = synth_inputs(batch=1, seed=0, device="cpu")
    inp = {k: (v.to(dev, torch.bfloat16) if torch.is_tensor(v) and v.is_floating_point() else
               (v.to(dev) if torch.is_tensor(v) else v)) for k, v in inp.items()}
    inp["image_rotary_emb"] = tuple(t.to(dev, torch.float32) for t in inp["image_rotary_emb"])
    inp["id_cond"] = [t.to(dev, torch.bfloat16) for t in inp["id_cond"]]
    inp["id_vit_hidden"] = [[t.to(dev, torch.bfloat16) for t in l] for l in inp["id_vit_hidden"]]
    return model, inp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--steps", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model, inputs = build_model_and_inputs(dev)
    with torch.no_grad():
        model(return_dict=False, denoise_step=0, **inputs)
        torch.cuda.synchronize()
        ops.enable_kernel_timers(by_shape=True)
        for _ in range(a.steps):
            model(return_dict=False, denoise_step=0, **inputs)
        flops = ops.kernel_timer_flops()
        times = ops.collect_kernel_timers()
    rows = []
    for k, v in times.items():
        if not k.startswith("bya_gemm_bf16:"):
            continue
        t = sum(v)
        rows.append(dict(shape=k.split(":", 1)[1], launches_per_step=len(v) // a.steps, ms_per_step=t / a.steps * 1e3,
                         tflops=flops[k] / t / 1e12 if t else 0.0, tflop_per_step=flops[k] / a.steps / 1e12))
    rows.sort(key=lambda r: -r["ms_per_step"])
    tot_ms, tot_tf = sum(r["ms_per_step"] for r in rows), sum(r["tflop_per_step"] for r in rows)
    for r in rows:
        print(f"{r['shape']:44s} x{r['launches_per_step']:4d}  {r['ms_per_step']:8.3f} ms  {r['tflops']:7.0f} TFLOP/s")
    print(f"total {tot_ms:.2f} ms/step, {tot_tf / tot_ms * 1e3:.0f} TFLOP/s")
    others = {k: sum(v) / a.steps * 1e3 for k, v in times.items() if not k.startswith("bya_gemm_bf16:")}
    for k, v in sorted(others.items(), key=lambda kv: -kv[1]):
        n, fl = len(times[k]) // a.steps, flops.get(k, 0.0)
        rate = f"{fl / sum(times[k]) / 1e12:7.0f} TFLOP/s" if fl else ""
        print(f"{k:60s} x{n:4d}  {v:8.3f} ms  {rate}")
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(dict(gemm=rows, other_ms_per_step=others), open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
