"""AdaLN LayerNorm at the step's shape (17776 x 3072, text / video modulation split at row 226): the rows-per-wave kernel with the
parameter vectors in registers (default) against the one-row-per-wave kernel that re-reads them (BYA_LN_ROWS=0), interleaved.
python tools/layernorm_probe.py [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bind_your_avatar_implementation_amd import ops
dev = torch.device("cuda:0")
S, D = 17776, 3072
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
x, y = rnd(1, S, D), torch.empty(1, S, D, dtype=torch.bfloat16, device=dev)
w, b = rnd(D), rnd(D)
mods = [rnd(D) * 0.3 for _ in range(4)]
run = lambda: ops.layernorm(x, y, w, b, 1e-5, shift0=mods[0], scale0=mods[1], shift1=mods[2], scale1=mods[3], split=226)
res = {"rows": [], "one_row": []}
outs = {}
for rep in range(4):
    for name, flag in (("rows", "1"), ("one_row", "0")):
        os.environ["BYA_LN_ROWS"] = flag
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run()
        e1.record(); torch.cuda.synchronize()
        res[name].append(round(e0.elapsed_time(e1) / 50 * 1e3, 2))
        outs[name] = y.clone()
d = (outs["rows"].float() - outs["one_row"].float()).abs()
res["max_abs_diff"] = float(d.max()); res["fraction_of_elements_that_differ"] = float((d > 0).float().mean())
res["gb_per_s_rows"] = round(2 * S * D * 2 / min(res["rows"]) / 1e3, 0)
print(res)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
