"""bya_router_scores at the step's shape (17550 tokens x 2 identities): the identity's keys resident in LDS (default) against the
kernel that re-reads them per wave (BYA_ROUTER_SCORES_LDS=0), interleaved, and their bit-identity.
python tools/router_scores_probe.py [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bind_your_avatar_implementation_amd import ops
dev = torch.device("cuda:0")
res = {}
for N, NID in ((17550, 2), (33750, 3), (4394, 2)):
    rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
    qr, kr = rnd(N, 2048), rnd(NID, 32, 2048)
    w, b, pos = rnd(512) * 0.2 + 1, rnd(512) * 0.2, rnd(N, 512) * 0.1
    out = torch.empty(NID, N, 512, dtype=torch.bfloat16, device=dev)
    run = lambda: ops.router_scores(qr, kr, w, b, pos, out, NID, N)
    t, outs = {"lds": [], "per_wave": []}, {}
    for rep in range(3):
        for name, flag in (("lds", "1"), ("per_wave", "0")):
            os.environ["BYA_ROUTER_SCORES_LDS"] = flag
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                run()
            e1.record(); torch.cuda.synchronize()
            t[name].append(round(e0.elapsed_time(e1) / 30 * 1e3, 1))
            outs[name] = out.clone()
    res[f"N{N}_ids{NID}"] = dict(us=t, bit_identical=bool(torch.equal(outs["lds"], outs["per_wave"])))
    print(N, NID, res[f"N{N}_ids{NID}"], flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
