"""Profiler driver: the fused router group attention (temporal shape), N launches.  python tools/router_group_attn_only.py [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bind_your_avatar_implementation_amd import ops

dev = torch.device("cuda:0")
M = 35100
x = torch.randn(M, 512, device=dev).to(torch.bfloat16)
w = (torch.randn(1536, 512, device=dev) * 512 ** -0.5).to(torch.bfloat16)
pack = ops.pack_rowgemm512(w, torch.zeros(1536, device=dev), torch.ones(512, device=dev), torch.zeros(512, device=dev))
out = torch.empty(M, 512, dtype=torch.bfloat16, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    ops.router_group_attn(x, pack, out, 13, 2, 1350, 17550, 1350)
    ops.router_group_attn(x, pack, out, 2, 1, 17550, 35100, 17550)
torch.cuda.synchronize()
