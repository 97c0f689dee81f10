"""The router's spatial attention (26 (id, frame) pairs x 8 heads x 1350 x 1350, q|k|v strided inside the [R, 1536] projection
output): the running-maximum kernel it uses today against the static-bound kernels (two-block and one-wave-per-SIMD w4,
with and without stream-K) that a weight-derived score bound would unlock.  python tools/spatial_attn_probe.py [out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bind_your_avatar_implementation_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
pairs, pf, F = 26, 1350, 512
R = pairs * pf
qkv = (torch.randn(R, 3 * F, device=dev) * 0.5).to(torch.bfloat16)
qkv[:, F:2 * F] *= 0.125 * 1.4426950408889634
out = torch.empty(R, F, dtype=torch.bfloat16, device=dev)


def run(**kw):
    ops.attention(qkv, qkv[:, F:], qkv[:, 2 * F:], out, head_dim=64, heads=8, nb1=pairs, nb2=1, Sq=pf, Skv=pf,
                  q_strides=(pf * 3 * F, 0, 3 * F), k_strides=(pf * 3 * F, 0, 3 * F), v_strides=(pf * 3 * F, 0, 3 * F),
                  o_strides=(pf * F, 0, F), scale=0.125, **kw)


def timed(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


fl = 4.0 * pairs * 8 * pf * pf * 64
res = {}
for name, env, kw in [("running_max (today)", {}, {}),
                      ("prescaled running_max", {}, dict(prescaled=True)),
                      ("prescaled running_max, 8-byte epilogue stores", {"BYA_ATTN_WIDE_STORE": "0"}, dict(prescaled=True)),
                      ("prescaled running_max (again)", {}, dict(prescaled=True)),
                      ("prescaled running_max, 8-byte epilogue stores (again)", {"BYA_ATTN_WIDE_STORE": "0"}, dict(prescaled=True)),
                      ("static bound, w4 per item", {"BYA_ATTN_STREAMK": "0"}, dict(prescaled=True, score_bound=20.0)),
                      ("static bound, w4 stream-K", {}, dict(prescaled=True, score_bound=20.0))]:
    for k_, v_ in env.items():
        os.environ[k_] = v_
    us = timed(lambda: run(**kw))
    for k_ in env:
        del os.environ[k_]
    res[name] = dict(us=round(us, 1), tflops=round(fl / us / 1e6, 0))
    print(name, res[name], flush=True)
run(prescaled=True)
wide = out.clone()
os.environ["BYA_ATTN_WIDE_STORE"] = "0"
out.zero_()
run(prescaled=True)
del os.environ["BYA_ATTN_WIDE_STORE"]
res["wide_stores_bit_identical"] = bool(torch.equal(wide, out))
print("16-byte epilogue stores bit-identical to the 8-byte ones:", res["wide_stores_bit_identical"])
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
