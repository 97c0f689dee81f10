"""Times the router chains of csrc/rowchain.hip (bya_router_mlp_fused, bya_router_group_attn_out) against the two launches each
replaces, at the one-GPU router shape (35100 rows) and at the 2- / 4- / 8-rank shard shapes, for every first-pass size.
python tools/router_chain_probe.py [out.json]"""
import json
import sys

import torch

sys.path.insert(0, ".")
from bind_your_avatar_implementation_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def rnd(*shape, std=1.0):
    return (torch.randn(*shape, generator=g) * std).to(torch.bfloat16).to(dev)


w, b = rnd(1536, 512, std=512 ** -0.5), rnd(1536, std=0.1)
wo, bo = rnd(512, 512, std=512 ** -0.5), rnd(512, std=0.1)
w1, b1 = rnd(512, 512, std=512 ** -0.5), rnd(512, std=0.1)
gam, bet = torch.ones(512, dtype=torch.bfloat16, device=dev), torch.zeros(512, dtype=torch.bfloat16, device=dev)
pack, po = ops.pack_rowgemm512(w, b, gam, bet), ops.pack_rowgemm512(wo, bo)
p1 = ops.pack_rowgemm512(w1, b1, gam, bet)


def timed(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)


res = {}
T, n_id = 13, 2
for world in (1, 2, 4, 8):
    loc = -(-1350 // world)                      # locations of one rank (location-major partition)
    M = n_id * T * loc
    x = rnd(M, 512)
    a, h = torch.empty_like(x), torch.empty_like(x)
    geo = {"temporal": (T, n_id, loc, T * loc, loc), "multi_id": (n_id, 1, T * loc, 0 if world > 1 else n_id * T * loc, T * loc)}
    if world == 1:
        geo = {"temporal": (13, 2, 1350, 17550, 1350), "multi_id": (2, 1, 17550, 35100, 17550)}
    r = {"rows": M}
    for name, (L, no, ni, os_, ss) in geo.items():
        pair_a = timed(lambda: ops.router_group_attn(x, pack, a, L, no, ni, os_, ss))
        pair_b = timed(lambda: ops.rowgemm512(a, po, x, res=x))
        r[name] = {"group_attn_us": pair_a, "out_proj_us": pair_b, "pair_us": round(pair_a + pair_b, 1), "chain_us": {}}
        for tp0 in (0, 8, 7, 6, 5, 4):
            r[name]["chain_us"][str(tp0)] = timed(lambda: ops.router_group_attn_out(x, pack, po, L, no, ni, os_, ss, tiles_pass0=tp0))
    m0 = timed(lambda: ops.rowgemm512(x, p1, h, act="gelu_erf"))
    m1 = timed(lambda: ops.rowgemm512(h, po, x, res=x))
    r["mlp"] = {"mlp0_us": m0, "mlp2_us": m1, "pair_us": round(m0 + m1, 1), "chain_us": {}}
    for tp0 in (0, 8, 7, 6, 5, 4):
        r["mlp"]["chain_us"][str(tp0)] = timed(lambda: ops.router_mlp_fused(x, p1, po, tiles_pass0=tp0))
    res[f"world{world}"] = r
    print(world, json.dumps(r))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
