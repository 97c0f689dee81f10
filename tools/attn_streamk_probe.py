"""Joint attention (48 heads x 17776 tokens, static-bound softmax): 256 persistent stream-K workgroups against one workgroup
per (head, q-tile) item, interleaved launches.  python tools/attn_streamk_probe.py [out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bind_your_avatar_implementation_amd import ops  # noqa: E402
from bind_your_avatar_implementation_amd import _hip  # noqa: E402

dev = torch.device("cuda:0")
res = {}
for S, H in ((17776, 48), (8888 * 2, 24), (17776, 6), (47026, 48)):
    D = 64
    nrm = lambda t: (t / t.view(1, S, H, D).float().norm(dim=-1, keepdim=True).repeat_interleave(D, -1).view(1, S, H * D) * 8)
    q, k, v = (torch.randn(1, S, H * D, device=dev) for _ in range(3))
    q, k, v = nrm(q).to(torch.bfloat16), (nrm(k) * (0.125 * 1.4426950408889634)).to(torch.bfloat16), v.to(torch.bfloat16)
    out = torch.empty_like(q)
    run = lambda: ops.self_attention(q, k, v, out, heads=H, prescaled=True, score_bound=11.8, tag="joint")
    t = {}
    for rep in range(3):
        for mode in ("stream_k", "per_item"):
            os.environ["BYA_ATTN_STREAMK"] = "1" if mode == "stream_k" else "0"
            _hip.apply_env_options()      # (the library reads no environment: hand the change to its option table)
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            t.setdefault(mode, []).append(e0.elapsed_time(e1) / 10)
    fl = 4.0 * H * S * S * D
    res[f"S{S}_H{H}"] = {m: dict(ms=round(min(v_), 4), tflops=round(fl / min(v_) / 1e9, 1)) for m, v_ in t.items()}
    print(f"S={S} H={H}", res[f"S{S}_H{H}"], flush=True)
    del q, k, v, out
assert ops.attn_workspace_status() == 0
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
