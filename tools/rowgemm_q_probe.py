"""N = 512 row GEMM: the W-stationary barrier-free kernel (default) against the chunk-balanced one (BYA_ROWGEMM_Q=0)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bind_your_avatar_implementation_amd import ops
dev = torch.device("cuda:0")
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
res = {}
for M in (35100, 17550, 8788, 4394, 2194, 43875, 52650, 70200):
    x = torch.randn(M, 512, device=dev).to(torch.bfloat16)
    w = (torch.randn(512, 512, device=dev) * 512 ** -0.5).to(torch.bfloat16)
    b = torch.randn(512, device=dev).to(torch.bfloat16)
    pack = ops.pack_rowgemm512(w, b)
    out = torch.randn(M, 512, device=dev).to(torch.bfloat16)
    row = {}
    for name, env in (("w_stationary", None), ("chunk_balanced", "0")):
        if env is not None: os.environ["BYA_ROWGEMM_Q"] = env
        row[name + "_res_us"] = round(timed(lambda: ops.rowgemm512(x, pack, out, res=out)), 1)
        o2 = torch.empty_like(out)
        row[name + "_plain_us"] = round(timed(lambda: ops.rowgemm512(x, pack, o2)), 1)
        os.environ.pop("BYA_ROWGEMM_Q", None)
    # same bits?
    a_ = torch.empty_like(out); b_ = torch.empty_like(out)
    ops.rowgemm512(x, pack, a_)
    os.environ["BYA_ROWGEMM_Q"] = "0"; ops.rowgemm512(x, pack, b_); os.environ.pop("BYA_ROWGEMM_Q")
    row["bit_identical"] = bool(torch.equal(a_, b_))
    res[f"M{M}"] = row
    print(M, row, flush=True)
if len(sys.argv) > 1: json.dump(res, open(sys.argv[1], "w"), indent=1)
