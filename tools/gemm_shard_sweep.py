"""Which GEMM structure is fastest at the PER-RANK shapes of a W-rank sequence-parallel step?  For every Linear of the step at
M = 17776 / W rows: the library's own choice against every tile structure forced (BYA_GEMM_TILE) and, for the persistent
256 x 256 kernel, several split-K thresholds (BYA_GEMM_SPLITK_MIN).  python tools/gemm_shard_sweep.py [out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bind_your_avatar_implementation_amd import ops  # noqa: E402
from bind_your_avatar_implementation_amd import _hip  # noqa: E402

dev = torch.device("cuda:0")
VARIANTS = [("default", {}), ("128x128", {"BYA_GEMM_TILE": "1"}), ("256x128", {"BYA_GEMM_TILE": "2"}),
            ("256x256 w8", {"BYA_GEMM_TILE": "3"}), ("256p no split", {"BYA_GEMM_TILE": "4", "BYA_GEMM_SPLITK": "0"}),
            ("256p split>=8", {"BYA_GEMM_TILE": "4", "BYA_GEMM_SPLITK_MIN": "8"}),
            ("256p split>=16", {"BYA_GEMM_TILE": "4", "BYA_GEMM_SPLITK_MIN": "16"}),
            ("256p split>=24", {"BYA_GEMM_TILE": "4", "BYA_GEMM_SPLITK_MIN": "24"})]


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters)
    return best * 1e3


res = {}
for W in ((8, 4, 2) if "--w1" not in sys.argv else (1,)):
    S_loc, N_loc = 17776 // W, 17550 // W
    shapes = [("qkv", S_loc, 9216, 3072, {}), ("attn_out", S_loc, 3072, 3072, {"res": True}),
              ("ff1", S_loc, 12288, 3072, {"act": "gelu_tanh"}), ("ff2", S_loc, 3072, 12288, {"res": True}),
              ("audio_q", N_loc, 3072, 3072, {}), ("perceiver_q", N_loc, 2048, 3072, {}),
              ("perceiver_out", N_loc, 3072, 2048, {"res": True}), ("router_q", N_loc, 2048, 2048, {})]
    for name, M, N, K, kw in shapes:
        x = (torch.randn(M, K, device=dev)).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev).to(torch.bfloat16)
        out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        kwargs = dict(bias=b)
        if kw.get("res"):
            kwargs["res"] = out
        if kw.get("act"):
            kwargs["act"] = kw["act"]
        row = {}
        for vname, env in VARIANTS:
            for k_, v_ in env.items():
                os.environ[k_] = v_
            _hip.apply_env_options()           # (the library reads no environment: hand the change to its option table)
            try:
                us = timed(lambda: ops.gemm(x, w, out, **kwargs))
                row[vname] = round(2.0 * M * N * K / us / 1e6, 0)
            except Exception as ex:      # a structure that does not take this epilogue
                row[vname] = str(ex)[:40]
            for k_ in env:
                del os.environ[k_]
            _hip.apply_env_options()
        best = max((v for v in row.values() if isinstance(v, float)), default=0)
        res[f"W{W} {name} {M}x{N}x{K}"] = row
        print(f"W{W} {name:14s} {M}x{N}x{K}: " + "  ".join(f"{k_}={v_}" for k_, v_ in row.items()), flush=True)
outs = [a for a in sys.argv[1:] if not a.startswith("--")]
if outs:
    json.dump(res, open(outs[0], "w"), indent=1)
