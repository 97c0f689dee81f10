#!/usr/bin/env python3
"""How fast does the push kernel move bytes when the destination is local HBM?  (one GPU: P2PGroup solo mode, every "peer" a
local buffer).  The packed q|k|v exchange of an 8-rank step: 24 pieces of 2222 x 384 bf16 = 41 MB.  Over xGMI the same
kernel is bound by the links (7 x ~77 GB/s per direction); what this shows is whether the kernel itself keeps enough loads
and stores in flight -- a kernel that cannot reach a multiple of the link rate against local memory will not fill the links
at their higher latency either.   python tools/p2p_push_bandwidth.py [out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bind_your_avatar_implementation_amd.p2p import P2PGroup  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    res = {}
    for W in (8, 4, 2):
        g = P2PGroup(solo=(1, W), device=dev)
        S_loc, Dl = 17776 // W, 3072 // W
        blocks = torch.randn(3 * W, S_loc, Dl, device=dev).to(torch.bfloat16)
        name = f"qkvh{W}"
        g.symmetric(name, (3, 17776, Dl), torch.bfloat16)
        pieces = [(blocks[t * W + j], j, name, (t * 17776 + 1 * S_loc) * Dl) for j in range(W) for t in range(3)]
        ch = g.channel(("probe", W), pieces)
        nbytes = blocks.numel() * 2
        row = {"bytes": nbytes, "control_block": g.ctrl_kind}
        for form, fn in (("exchange (one launch)", lambda: ch.exchange()), ("push + wait (two launches)", lambda: ch.push().wait()),
                         ("push alone", lambda: ch.push())):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(20):
                    fn()
                e.record()
                torch.cuda.synchronize()
                best = min(best, s.elapsed_time(e) / 20 * 1e3)
            row[form] = {"us": round(best, 1), "GBps": round(nbytes / best * 1e-3)}
            if form == "push alone":              # (the pushes ran ahead of the waits: bring the channel's counters back in step)
                for _ in range(5 + 5 * 20):
                    ch.wait()
                torch.cuda.synchronize()
        res[f"W={W}"] = row
        # a small exchange (the router's repartitions and logits: fixed cost only)
        tiny = torch.randn(W, 4096, device=dev).to(torch.bfloat16)
        g.symmetric(f"tiny{W}", (W, 4096), torch.bfloat16)
        ct = g.channel(("tiny", W), [(tiny[j], j, f"tiny{W}", 1 * 4096) for j in range(W)])
        for _ in range(5):
            ct.exchange()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(50):
                ct.exchange()
            e.record()
            torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 50 * 1e3)
        row["64 KB exchange (one launch)"] = {"us": round(best, 1)}
        print(W, res[f"W={W}"], flush=True)
        assert g.timeouts() == 0
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
