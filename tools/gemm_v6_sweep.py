"""The two 128 x 256 persistent GEMMs side by side (forced tile 5 = four waves, csrc/gemm_v5.hip; 6 = loader waves, csrc/gemm_v6.hip)
on per-rank shapes, one JSON line; BYA_HIP_LIB selects a side build (e.g. another barrier slot of the generated schedule).
Run from the repository root:  python tools/gemm_v6_sweep.py"""
import sys, torch, json
sys.path.insert(0, ".")
from bind_your_avatar_implementation_amd import ops
dev = torch.device("cuda:0")
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
res = {}
for (M, N, K, kind) in ((2222, 3072, 3072, "bias"), (2222, 3072, 3072, "res"), (2193, 2048, 3072, "bias"), (2222, 3072, 12288, "res"), (4444, 12288, 3072, "gelu"), (2222, 12288, 3072, "gelu"), (1392, 3072, 3072, "res")):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    o = torch.randn(M, N, device=dev).to(torch.bfloat16)
    kw = {"bias": dict(bias=b), "res": dict(bias=b, res=o), "gelu": dict(bias=b, act="gelu_tanh")}[kind]
    for tile in (5, 6):
        with ops.options(gemm_tile=tile, gemm_splitk=0):
            t = timed(lambda: ops.gemm(x, w, o, **kw))
        res[f"{M}x{N}x{K} {kind} tile{tile}"] = round(2.0 * M * N * K / t / 1e6)
print(json.dumps(res))
