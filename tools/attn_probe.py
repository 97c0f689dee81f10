#!/usr/bin/env python3
"""Joint-attention kernel on all-zero vs normalised gaussian q / k / v with board power and shader clock sampled while
each arm loops (the attention counterpart of tools/gemm_probe.py): separates what the kernel does per CLOCK from what
the power limit takes away on realistic data.  `python tools/attn_probe.py [--out gpurun_out/attn_probe.json]`"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gemm_probe import Smi, median, time_once
from bind_your_avatar_implementation_amd import ops

S, H, D = 17776, 48, 64
PEAK_PER_GHZ = 256 * 4 * 512 * 2 / 1e3          # TFLOP/s per GHz of shader clock (dense bf16 MFMA)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--seconds", type=float, default=4.0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    flop = 4.0 * S * S * D * H
    res = {}
    for data in ("zeros", "gaussian"):
        if data == "zeros":
            q = torch.zeros(1, S, H * D, dtype=torch.bfloat16, device=dev)
            k, v = torch.zeros_like(q), torch.zeros_like(q)
        else:
            g = torch.Generator(device=dev).manual_seed(0)
            nrm = lambda t: (t / t.view(1, S, H, D).float().norm(dim=-1, keepdim=True).repeat_interleave(D, -1).view(1, S, H * D) * 8).to(torch.bfloat16)
            q = nrm(torch.randn(1, S, H * D, device=dev, generator=g))
            k = (nrm(torch.randn(1, S, H * D, device=dev, generator=g)).float() * (0.125 * 1.4426950408889634)).to(torch.bfloat16)
            v = torch.randn(1, S, H * D, device=dev, generator=g).to(torch.bfloat16)
        o = torch.empty_like(q)
        for name, kw in (("bounded", dict(prescaled=True, score_bound=11.8)), ("prescaled", dict(prescaled=True)),
                         ("running_max", dict())):
            fn = lambda: ops.self_attention(q, k, v, o, heads=H, **kw)
            t1 = time_once(fn, 3)
            iters = max(5, int(a.seconds / t1))
            smi = Smi()
            smi.start()
            ts = [time_once(fn, iters // 3 + 1) for _ in range(3)]
            smi.stop_flag = True
            smi.join()
            t = median(ts)
            sm = smi.summary()
            r = dict(ms=t * 1e3, tflops=flop / t / 1e12, **sm)
            if sm.get("sclk_mhz_avg"):
                r["frac_of_per_clock_peak"] = r["tflops"] / (PEAK_PER_GHZ * sm["sclk_mhz_avg"] / 1e3)
            res[f"{data}:{name}"] = r
            print(f"{data:9s} {name:12s} {t * 1e3:7.3f} ms  {r['tflops']:7.0f} TFLOP/s  {sm}")
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
