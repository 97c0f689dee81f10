#!/usr/bin/env python3
"""Reference point only (not used by the engine): torch.nn.functional.linear = hipBLASLt/rocBLAS on the DiT shapes,
next to bya_gemm_bf16 on the same operands."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from bind_your_avatar_implementation_amd import ops
dev = torch.device("cuda:0")

def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

for (M, N, K) in [(17776, 9216, 3072), (17776, 12288, 3072), (17776, 3072, 12288), (17776, 3072, 3072), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t_v = timeit(lambda: F.linear(a, w, b))
    t_o = timeit(lambda: ops.gemm(a, w, out, bias=b))
    fl = 2 * M * N * K / 1e12
    print(f"M={M} N={N} K={K}: vendor {t_v*1e3:.3f} ms {fl/t_v:.0f} TF/s | bya_gemm_bf16 {t_o*1e3:.3f} ms {fl/t_o:.0f} TF/s", flush=True)
