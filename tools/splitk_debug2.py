#!/usr/bin/env python3
"""Structured-input debug of the split-K path: constant / column-coded / row-coded operands so that what the finisher adds,
and where, can be read off the output."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bind_your_avatar_implementation_amd import ops

dev = torch.device("cuda:0")
M, N, K = 1024, 1024, 2048          # 16 tiles: 2 per XCD, 4 K-ranges each (512 deep)
os.environ["BYA_GEMM_TILE"] = "4"


def show(name, a, w, mode):
    os.environ["BYA_GEMM_SPLITK"] = mode
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(a, w, out)
    torch.cuda.synchronize()
    ref = (a.float() @ w.float().T)
    o = out.float()
    t = o[:256, :256]
    r = ref[:256, :256]
    bad = (t != r)
    print(f"--- {name} mode={mode}: tile(0,0) wrong elements {bad.sum().item()} / 65536")
    if bad.any():
        vals, cnt = torch.unique((t / r)[bad].round(decimals=3), return_counts=True)
        print("   ratio out/ref among wrong elements:", [(round(v.item(), 3), c.item()) for v, c in zip(vals[:12], cnt[:12])])
        rows = bad.any(dim=1).nonzero().flatten().tolist()
        cols = bad.any(dim=0).nonzero().flatten().tolist()
        print("   wrong rows:", rows[:40], "...", len(rows))
        print("   wrong cols:", cols[:40], "...", len(cols))
        print("   sample row 0, cols 0..15 out:", t[0, :16].tolist())
        print("   sample row 0, cols 0..15 ref:", r[0, :16].tolist())
        print("   sample col 0, rows 0..15 out:", t[:16, 0].tolist())
        print("   sample col 0, rows 0..15 ref:", r[:16, 0].tolist())


ones_a = torch.ones(M, K, dtype=torch.bfloat16, device=dev)
ones_w = torch.ones(N, K, dtype=torch.bfloat16, device=dev)
colcode = ((torch.arange(N, device=dev) % 256) + 1).to(torch.bfloat16)[:, None].expand(N, K).contiguous()
rowcode = ((torch.arange(M, device=dev) % 256) + 1).to(torch.bfloat16)[:, None].expand(M, K).contiguous()
# K-coded: A[m, k] = 1 for k in K-range q, else 0 -> out = 512 if that range was summed
for mode in ("1",):
    show("ones", ones_a, ones_w, mode)
    show("column-coded W", ones_a, colcode, mode)
    show("row-coded A", rowcode, ones_w, mode)
    for q in range(4):
        a = torch.zeros(M, K, dtype=torch.bfloat16, device=dev)
        a[:, q * 512:(q + 1) * 512] = 1
        show(f"only K-range {q} nonzero", a, ones_w, mode)
