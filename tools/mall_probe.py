"""Memory-side evidence for "the GEMM's fabric traffic is served by the 256 MB Infinity Cache, not HBM" (rocprofv3 on this
image exposes no HBM-side counter): time the same launch (a) over ONE operand set, re-used launch after launch (A + W + C
fit the MALL), and (b) rotating over R operand sets whose total is far above 256 MB, so every launch finds its operands
evicted.  Control: the HBM-bound LayerNorm kernel under the same treatment.  python tools/mall_probe.py [out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bind_your_avatar_implementation_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timed(fns, iters=48):
    for f in fns:
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fns[i % len(fns)]()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


res = {}
for name, M, N, K in (("attn_out 17776x3072x3072", 17776, 3072, 3072), ("qkv 17776x9216x3072", 17776, 9216, 3072),
                      ("ff2 17776x3072x12288", 17776, 3072, 12288)):
    def make():
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        return lambda: ops.gemm(a, w, c)
    mb = (M * K + N * K + M * N) * 2 / 1e6
    one = timed([make()])
    R = max(3, int(3000 / mb) + 1)
    many = timed([make() for _ in range(R)])
    fl = 2.0 * M * N * K
    res[name] = dict(operand_set_MB=round(mb), one_set_us=round(one, 1), one_set_tflops=round(fl / one / 1e6),
                     rotating_sets=R, rotating_us=round(many, 1), rotating_tflops=round(fl / many / 1e6))
    print(name, res[name], flush=True)
# control: an HBM-bound kernel
S, D = 17776, 3072
def make_ln():
    x, y = torch.randn(1, S, D, device=dev).to(torch.bfloat16), torch.empty(1, S, D, dtype=torch.bfloat16, device=dev)
    w, b = torch.ones(D, dtype=torch.bfloat16, device=dev), torch.zeros(D, dtype=torch.bfloat16, device=dev)
    return lambda: ops.layernorm(x, y, w, b)
one, many = timed([make_ln()]), timed([make_ln() for _ in range(16)])
res["layernorm 17776x3072 (control, 218 MB per launch)"] = dict(one_set_us=round(one, 1), one_set_GBps=round(2 * S * D * 2 / one / 1e3),
                                                                 rotating_sets=16, rotating_us=round(many, 1),
                                                                 rotating_GBps=round(2 * S * D * 2 / many / 1e3))
print(res["layernorm 17776x3072 (control, 218 MB per launch)"])
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
