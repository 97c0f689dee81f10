"""A last round of 256 x 256 tiles cut along K (gemm_v4.hip: slabs + hand-off counters) against the row plan that needs no hand-off
(rows of the full rounds on 256-row tiles, the rest on 128-row tiles with loader waves): the shapes where the K-split applies
(K >= 5120), same process, interleaved.  Run from the repository root:  python tools/gemm_splitk_vs_rowplan.py"""
import json
import sys

import torch

sys.path.insert(0, ".")
from bind_your_avatar_implementation_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
ops.ensure_gemm_workspace(dev)


def timed(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


res = {}
for M, N, K in ((17776, 3072, 12288), (8888, 3072, 12288), (4444, 3072, 12288), (2222, 3072, 12288)):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    o = torch.randn(M, N, device=dev).to(torch.bfloat16)
    best = {"k_split": 1e9, "row_plan": 1e9}
    for _ in range(4):
        best["k_split"] = min(best["k_split"], timed(lambda: ops.gemm(x, w, o, bias=b, res=o)))
        with ops.options(gemm_splitk=0):
            best["row_plan"] = min(best["row_plan"], timed(lambda: ops.gemm(x, w, o, bias=b, res=o)))
    res[f"{M}x{N}x{K}"] = {k: [round(v, 1), round(2.0 * M * N * K / v / 1e6)] for k, v in best.items()}
    print(f"{M}x{N}x{K}", res[f"{M}x{N}x{K}"], flush=True)
ops.check_gemm_workspace()
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
