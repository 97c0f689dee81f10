#!/usr/bin/env python3
"""How far apart are two summation orders of the same 42-layer denoise step?  (the floor under bench.py's N > 1 check)

bench.py --gpus N compares the sharded step in its default mode (split-K tails in the GEMMs, stream-K joint attention) with
the unsharded step in the unsplit mode and bounds the relative Frobenius distance.  The sharded default mode cannot run on
the builder's one-GPU box at N > 2 (several processes on one GPU starve each other's co-resident grids), so the bound is
set from what CAN be measured here: the same unsharded step under each combination of the two switches, against the unsplit
one.  Every pair differs only by fp32 summation order inside bf16 activations.

  python tools/default_mode_noise.py [--layers 42] [--out gpurun_out/default_mode_noise.json]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=42)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from bench import MODEL_KW
    from bind_your_avatar_implementation_amd import BindyouravatarTransformer3DModel, _hip, ops
    from bind_your_avatar_implementation_amd.synth import synth_inputs
    dev = torch.device("cuda:0")
    model = BindyouravatarTransformer3DModel(**dict(MODEL_KW, num_layers=a.layers), device=dev).init_synthetic(seed=0, fast=True)
    inp = synth_inputs(batch=1, seed=0, device="cpu")
    inp = {k: (v.to(dev, torch.bfloat16) if torch.is_tensor(v) and v.is_floating_point() else
               (v.to(dev) if torch.is_tensor(v) else v)) for k, v in inp.items()}
    inp["image_rotary_emb"] = tuple(t.to(dev, torch.float32) for t in inp["image_rotary_emb"])
    inp["id_cond"] = [t.to(dev, torch.bfloat16) for t in inp["id_cond"]]
    inp["id_vit_hidden"] = [[t.to(dev, torch.bfloat16) for t in l] for l in inp["id_vit_hidden"]]

    def run(splitk, streamk):
        os.environ["BYA_GEMM_SPLITK"], os.environ["BYA_ATTN_STREAMK"] = splitk, streamk     # read per call by the library
        _hip.apply_env_options()      # (the library reads no environment: hand the change to its option table)
        model(return_dict=False, denoise_step=0, **inp)
        out = model(return_dict=False, denoise_step=0, **inp)[0].float().clone()
        torch.cuda.synchronize()
        return out

    ref = run("0", "0")
    again = run("0", "0")
    res = {"layers": a.layers, "unsplit_twice_bit_identical": bool(torch.equal(ref, again)), "rel_fro_vs_unsplit": {}}
    for name, sk, st in (("split-K only", "1", "0"), ("stream-K only", "0", "1"), ("default (both)", "1", "1")):
        out = run(sk, st)
        res["rel_fro_vs_unsplit"][name] = ((out - ref).norm() / ref.norm()).item()
    ops.check_gemm_workspace()
    print(json.dumps(res))
    if a.out:
        with open(a.out, "w") as f:
            f.write(json.dumps(res) + "\n")


if __name__ == "__main__":
    main()
