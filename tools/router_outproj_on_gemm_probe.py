"""The router's N = 512 out-projections (+ residual, no LayerNorm in front: 336 launches per step on rowgemm512q_kernel) on the
128 x 256 persistent GEMM with loader waves instead (option gemm_tile = 6; K = 512 is eight K-tiles, and that kernel's ring runs
across output tiles, so there is no fill / drain per tile): bit-equality with the row kernel and time, at the one-GPU row count and
at the shards' row counts.  Run from the repository root:  python tools/router_outproj_on_gemm_probe.py [out.json]"""
import json
import sys

import torch

sys.path.insert(0, ".")
from bind_your_avatar_implementation_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return round(best, 1)


res = {}
g = torch.Generator().manual_seed(0)
w = (torch.randn(512, 512, generator=g) * 512 ** -0.5).to(torch.bfloat16).to(dev)
b = (torch.randn(512, generator=g) * 0.2).to(torch.bfloat16).to(dev)
pack = ops.pack_rowgemm512(w, b)
for M in (35100, 17550, 8788, 4394):
    x = torch.randn(M, 512, generator=g).to(torch.bfloat16).to(dev)
    r = torch.randn(M, 512, generator=g).to(torch.bfloat16).to(dev)
    o_row, o_gemm = torch.empty_like(r), torch.empty_like(r)
    ops.rowgemm512(x, pack, o_row, res=r)
    row = {"rowgemm_us": timed(lambda: ops.rowgemm512(x, pack, o_row, res=r))}
    for tile in (6, 1, 4):
        with ops.options(gemm_tile=tile):
            ops.gemm(x, w, o_gemm, bias=b, res=r)
            torch.cuda.synchronize()
            row[f"gemm_tile{tile}_equal"] = bool(torch.equal(o_gemm, o_row))
            row[f"gemm_tile{tile}_us"] = timed(lambda: ops.gemm(x, w, o_gemm, bias=b, res=r))
    res[M] = row
    print(M, row, flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
