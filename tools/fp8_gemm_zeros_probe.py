"""Is the persistent e4m3 GEMM's K-loop bound by its schedule or by the board's power limit?  The same launch on gaussian
operands and on all-zero operands (no switching activity: the clock stays up), and the tile-order group size.
usage: python tools/fp8_gemm_zeros_probe.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bind_your_avatar_implementation_amd import ops
from bind_your_avatar_implementation_amd import _hip  # noqa: E402
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, N, K = 17776, 9216, 3072
def bench(a8, sa, w8, sw, out, n=20):
    best = 1e9
    for rep in range(3):
        for _ in range(3):
            ops.gemm_fp8(a8, sa, w8, sw, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n):
            ops.gemm_fp8(a8, sa, w8, sw, out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
a8, sa = ops.quantize_rows_fp8(torch.randn(M, K, device=dev).to(torch.bfloat16))
w8, sw = ops.quantize_rows_fp8((torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16))
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
z8a, z8w = torch.zeros_like(a8), torch.zeros_like(w8)
for kern in ("128", "v4"):
    os.environ["BYA_FP8_KERNEL"] = kern
    _hip.apply_env_options()      # (the library reads no environment: hand the change to its option table)
    for rep in range(2):
        g = bench(a8, sa, w8, sw, out)
        z = bench(z8a, sa, z8w, sw, out)
        print(f"kernel {kern} run {rep}: gaussian {g:7.1f} us = {2.0*M*N*K/g*1e-6:7.1f} TF   zeros {z:7.1f} us = {2.0*M*N*K/z*1e-6:7.1f} TF", flush=True)
