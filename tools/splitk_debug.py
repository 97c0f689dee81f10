#!/usr/bin/env python3
"""Per-tile error map of the split-K GEMM path (debugging aid): which 256 x 256 output tiles differ from the fp32 product."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bind_your_avatar_implementation_amd import ops

dev = torch.device("cuda:0")


def run(M, N, K, mode):
    g = torch.Generator().manual_seed(1)
    a = (torch.randn(M, K, generator=g)).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    os.environ["BYA_GEMM_SPLITK"] = mode
    os.environ["BYA_GEMM_TILE"] = "4"
    ops.gemm(a, w, out)
    torch.cuda.synchronize()
    ref = a.float() @ w.float().T
    tm, tn = (M + 255) // 256, (N + 255) // 256
    err = torch.zeros(tm, tn)
    for i in range(tm):
        for j in range(tn):
            r, o = ref[i * 256:(i + 1) * 256, j * 256:(j + 1) * 256], out[i * 256:(i + 1) * 256, j * 256:(j + 1) * 256].float()
            err[i, j] = ((o - r).norm() / r.norm()).item()
    bad = (err > 5e-3).nonzero().tolist()
    print(f"M={M} N={N} K={K} mode={mode}: tiles {tm}x{tn}, bad tiles {len(bad)}; worst {err.max():.3e}")
    if bad:
        print("   bad (tile_m, tile_n, err):", [(i, j, round(err[i, j].item(), 3)) for i, j in bad[:40]])
    return err


for mode in ("0", "1", "2"):
    run(1024, 1024, 2048, mode)
    run(2222, 3072, 3072, mode)
    run(17776, 3072, 3072, mode)
for mode in ("1", "2"):
    for _ in range(3):
        run(1024, 1024, 2048, mode)
