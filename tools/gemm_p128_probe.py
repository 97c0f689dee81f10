"""The 128 x 256 persistent GEMM (csrc/gemm_v5.hip) against the other tile structures at the PER-RANK shapes of a W-rank
sequence-parallel step and at the one-GPU shapes: TFLOP/s per forced structure (BYA_OPT_GEMM_TILE: 1 = 128 x 128, 4 = 256 x 256
persistent, 5 = 128 x 256 persistent), the library's own choice, and whether the forced-5 output equals the forced-4 output bit
for bit.  Variants are timed interleaved, best of several rounds.  python tools/gemm_p128_probe.py [out.json] [--quick]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bind_your_avatar_implementation_amd import ops  # noqa: E402
from bind_your_avatar_implementation_amd.synth import rope_table  # noqa: E402

dev = torch.device("cuda:0")
VARIANTS = [("choice", {}), ("128x128", {"gemm_tile": 1}), ("256p", {"gemm_tile": 4}), ("128p", {"gemm_tile": 5}), ("128s", {"gemm_tile": 6})]


def bench(fns, iters=20, rounds=4):
    """fns: name -> (library options of the variant, callable); the options are set once around each timing loop"""
    best = {k: 1e9 for k in fns}
    for k, (opts, f) in fns.items():
        with ops.options(**opts):
            for _ in range(3):
                f()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, (opts, f) in fns.items():
            with ops.options(**opts):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(iters):
                    f()
                e.record()
                torch.cuda.synchronize()
            best[k] = min(best[k], s.elapsed_time(e) / iters * 1e3)
    return best


def main():
    out_path = next((a for a in sys.argv[1:] if not a.startswith("--")), None)
    worlds = (8, 4) if "--quick" in sys.argv else (8, 4, 2, 1)
    res = {}
    for W in worlds:
        S_loc, N_loc = 17776 // W, 17550 // W
        shapes = [("qkv+norm", S_loc, 9216, 3072, "qkn"), ("attn_out", S_loc, 3072, 3072, "res"), ("ff1", S_loc, 12288, 3072, "gelu"),
                  ("ff2", S_loc, 3072, 12288, "res"), ("audio_q", N_loc, 3072, 3072, ""), ("perceiver_q", N_loc, 2048, 3072, ""),
                  ("perceiver_out", N_loc, 3072, 2048, "res"), ("router_q", N_loc, 2048, 2048, "")]
        for name, M, N, K, kind in shapes:
            x = torch.randn(M, K, device=dev).to(torch.bfloat16)
            w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
            b = torch.randn(N, device=dev).to(torch.bfloat16)
            outs = {v: torch.zeros(M, N, dtype=torch.bfloat16, device=dev) for v, _ in VARIANTS}
            r0 = torch.randn(M, N, device=dev).to(torch.bfloat16)
            if kind == "qkn":
                width, text = N // 3, min(226, M // 2)
                grid = (1, M - text, 1)
                cos, sin = (c.to(dev).contiguous() for c in rope_table(grid))
                nw = [torch.randn(64, device=dev).to(torch.bfloat16) for _ in range(4)]
                outs = {v: torch.zeros(3, 1, M, width, dtype=torch.bfloat16, device=dev) for v, _ in VARIANTS}

            def make(vname, opts):
                o = outs[vname]

                def run():
                    if kind == "qkn":
                        ops.gemm_qkv_norm_rope(x.view(1, M, K), w, o[0], b, (width, M * width), *nw, cos, sin, text, eps=1e-6, k_scale=0.18)
                    elif kind == "res":
                        o.copy_(r0)
                        ops.gemm(x, w, o, bias=b, res=o)
                    elif kind == "gelu":
                        ops.gemm(x, w, o, bias=b, act="gelu_tanh")
                    else:
                        ops.gemm(x, w, o, bias=b)
                return opts, run

            fns = {v: make(v, o) for v, o in VARIANTS if not (kind == "qkn" and v == "128x128")}
            t = bench(fns)
            if kind == "res":       # the copy of the residual rides in every variant alike: time it alone and take it out
                tc = bench({"copy": ({}, lambda: outs["choice"].copy_(r0))})["copy"]
                t = {k: v - tc for k, v in t.items()}
            torch.cuda.synchronize()
            row = {k: round(2.0 * M * N * K / (v * 1e-6) / 1e12, 1) for k, v in t.items()}
            row["us"] = {k: round(v, 1) for k, v in t.items()}
            row["128p_equals_256p"] = bool(torch.equal(outs["128p"], outs["256p"]))
            row["128s_equals_256p"] = bool(torch.equal(outs["128s"], outs["256p"]))
            row["choice_equals_256p"] = bool(torch.equal(outs["choice"], outs["256p"]))
            res[f"W{W} {name} {M}x{N}x{K}"] = row
            print(f"W{W} {name:14s} {M}x{N}x{K}: " + "  ".join(f"{k}={v}" for k, v in row.items() if k != "us"), flush=True)
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
