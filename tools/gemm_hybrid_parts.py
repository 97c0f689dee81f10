"""The two launches of a row plan timed apart (csrc/gemm.hip: rows of the full rounds of 256 x 256 tiles, then the remaining rows on
128 x 256 tiles): whole launch on 256-row tiles, the library's choice, the main rows alone, the tail rows alone, whole launch on
128-row tiles.  Run from the repository root:  python tools/gemm_hybrid_parts.py"""
import sys, torch
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
    return round(best, 1)
for (M, N, K, act, m0) in ((4444, 12288, 3072, "gelu_tanh", 4096), (4444, 9216, 3072, None, 3584), (2222, 9216, 3072, None, 1792), (17776, 3072, 3072, None, 16384)):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kw = dict(bias=b, act=act) if act else dict(bias=b)
    r = {}
    with ops.options(gemm_variant=2):
        r["256p whole"] = timed(lambda: ops.gemm(x, w, o, **kw))
    r["choice"] = timed(lambda: ops.gemm(x, w, o, **kw))
    with ops.options(gemm_variant=2):
        r["256p main rows"] = timed(lambda: ops.gemm(x[:m0], w, o[:m0], **kw))
    with ops.options(gemm_tile=5):
        r["128p tail rows"] = timed(lambda: ops.gemm(x[m0:], w, o[m0:], **kw))
        r["128p whole"] = timed(lambda: ops.gemm(x, w, o, **kw))
    print(M, N, K, act, r, flush=True)
