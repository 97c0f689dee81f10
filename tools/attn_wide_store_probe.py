"""Joint attention (48 heads x 17776 tokens, the step's kernel): 16-byte epilogue stores (v_permlane32_swap pairs, the default)
against the 8-byte stores (BYA_ATTN_WIDE_STORE=0), interleaved launches.  python tools/attn_wide_store_probe.py [out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bind_your_avatar_implementation_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
S, H, D = 17776, 48, 64
nrm = lambda t: (t / t.view(1, S, H, D).float().norm(dim=-1, keepdim=True).repeat_interleave(D, -1).view(1, S, H * D) * 8)
q, k, v = (torch.randn(1, S, H * D, device=dev) for _ in range(3))
q, k, v = nrm(q).to(torch.bfloat16), (nrm(k) * (0.125 * 1.4426950408889634)).to(torch.bfloat16), v.to(torch.bfloat16)
out = torch.empty_like(q)
ops.ensure_attn_workspace(dev)
run = lambda: ops.self_attention(q, k, v, out, heads=H, prescaled=True, score_bound=11.8, tag="joint")
t, outs = {"wide": [], "narrow": []}, {}
for rep in range(4):
    for mode in ("wide", "narrow"):
        os.environ["BYA_ATTN_WIDE_STORE"] = "1" if mode == "wide" else "0"
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        t[mode].append(round(e0.elapsed_time(e1) / 10, 4))
        outs[mode] = out.clone()
res = {"ms_per_launch": t, "bit_identical": bool(torch.equal(outs["wide"], outs["narrow"]))}
print(res)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
