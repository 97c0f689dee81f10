#!/usr/bin/env python3
"""Why does cutting K not help a launch with fewer tiles than CUs?  2222 x 3072 x 3072 (108 tiles of 256 x 256 on 256 CUs: an
8-rank shard's attention-out / audio projections) on the persistent kernel: unsplit, split in two (BYA_GEMM_SPLITK), and the
unsplit kernel on HALF the K (what a split half computes, without the exchange) -- plus the 128 x 128 kernel the library
picks today.  python tools/small_grid_splitk_probe.py [out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bind_your_avatar_implementation_amd import ops  # noqa: E402
from bind_your_avatar_implementation_amd import _hip  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters * 1e3)
    return best


def main():
    ops.ensure_gemm_workspace(dev)
    res = {}
    g = torch.Generator().manual_seed(0)
    for M, N, K in ((2222, 3072, 3072), (2222, 3072, 1536), (2222, 3072, 768), (2222, 3072, 12288), (2222, 3072, 6144)):
        a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        row = {}
        for name, env in (("256p unsplit", {"BYA_GEMM_TILE": "4", "BYA_GEMM_SPLITK": "0"}),
                          ("256p split", {"BYA_GEMM_TILE": "4", "BYA_GEMM_SPLITK": "1", "BYA_GEMM_SPLITK_MIN": "3"}),
                          ("128x128", {"BYA_GEMM_TILE": "1"}), ("library's choice", {})):
            for k in ("BYA_GEMM_TILE", "BYA_GEMM_SPLITK", "BYA_GEMM_SPLITK_MIN"):
                os.environ.pop(k, None)
            os.environ.update(env)
            _hip.apply_env_options()      # (the library reads no environment: hand the change to its option table)
            us = timed(lambda: ops.gemm(a, w, out))
            row[name] = {"us": round(us, 1), "tflops": round(2.0 * M * N * K / us * 1e-6)}
        res[f"{M}x{N}x{K}"] = row
        print(f"{M}x{N}x{K}", row, flush=True)
    ops.check_gemm_workspace()
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
