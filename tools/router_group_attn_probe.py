"""Times the fused LayerNorm -> q|k|v -> group attention launch (bya_router_group_attn) against the unfused pair it
replaces (bya_rowgemm512 N = 1536 + bya_attn_tiny) at the router's shapes.  python tools/router_group_attn_probe.py [out.json]"""
import json
import sys

import torch

sys.path.insert(0, ".")
from bind_your_avatar_implementation_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M = 35100
x = (torch.randn(M, 512, generator=g)).to(torch.bfloat16).to(dev)
w = (torch.randn(1536, 512, generator=g) * 512 ** -0.5).to(torch.bfloat16).to(dev)
b = (torch.randn(1536, generator=g) * 0.1).to(torch.bfloat16).to(dev)
gam = torch.ones(512, dtype=torch.bfloat16, device=dev)
bet = torch.zeros(512, dtype=torch.bfloat16, device=dev)
pack = ops.pack_rowgemm512(w, b, gam, bet)
out = torch.empty(M, 512, dtype=torch.bfloat16, device=dev)
qkv = torch.empty(M, 1536, dtype=torch.bfloat16, device=dev)


def timed(fn, n=50):
    for _ in range(5):
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
for name, (L, no, ni, os_, ss) in {"temporal": (13, 2, 1350, 17550, 1350), "multi_id": (2, 1, 17550, 35100, 17550)}.items():
    fused = timed(lambda: ops.router_group_attn(x, pack, out, L, no, ni, os_, ss))
    gemm = timed(lambda: ops.rowgemm512(x, pack, qkv))
    tiny = timed(lambda: ops.attn_tiny(qkv, qkv[:, 512:], qkv[:, 1024:], out, L, 8, no, ni, os_, ss, 1536, 512, 0.125))
    res[name] = dict(fused_us=round(fused, 1), rowgemm_qkv_us=round(gemm, 1), attn_tiny_us=round(tiny, 1),
                     unfused_us=round(gemm + tiny, 1))
    print(name, res[name])
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
