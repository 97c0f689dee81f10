"""Implicit-GEMM convolution (bya_vae_conv3d) against the plain persistent GEMM of the same M x N x K, per decoder level.
usage: python tools/vae_conv_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bind_your_avatar_implementation_amd import ops
from bind_your_avatar_implementation_amd import _hip  # noqa: E402
dev = torch.device("cuda:0")
torch.manual_seed(0)

def timeit(fn, n=5):
    best = 1e9
    for rep in range(3):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best

for (T, H, W, C, Cout) in [(8, 480, 720, 128, 128), (8, 240, 360, 256, 256), (4, 120, 180, 512, 512), (2, 60, 90, 512, 512), (8, 480, 720, 256, 128)]:
    xpad = torch.zeros(T + 2, H + 2, W + 2, C, dtype=torch.bfloat16, device=dev)
    xpad[:, 1:-1, 1:-1] = torch.randn(T + 2, H, W, C, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, 27 * C, device=dev) * (27 * C) ** -0.5).to(torch.bfloat16)
    b = torch.randn(Cout, device=dev).to(torch.bfloat16)
    y = torch.empty(T, H, W, Cout, dtype=torch.bfloat16, device=dev)
    t_conv = timeit(lambda: ops.vae_conv3d(xpad, w, b, y))
    fl = 2.0 * T * H * W * Cout * 27 * C
    # the plain GEMM of the same size: A = [rows, 27 C] (what the patch path multiplied), rows capped to fit memory
    rows = min(T * H * W, (6 << 30) // (27 * C * 2))
    a = torch.randn(rows, 27 * C, device=dev).to(torch.bfloat16)
    y2 = torch.empty(rows, Cout, dtype=torch.bfloat16, device=dev)
    os.environ["BYA_GEMM_TILE"] = "4"
    _hip.apply_env_options()      # (the library reads no environment: hand the change to its option table)
    t_gemm = timeit(lambda: ops.gemm(a, w, y2, bias=b)) * (T * H * W / rows)
    os.environ.pop("BYA_GEMM_TILE")
    _hip.apply_env_options()      # (the library reads no environment: hand the change to its option table)
    t_gemm128 = timeit(lambda: ops.gemm(a, w, y2, bias=b)) * (T * H * W / rows)
    print(f"{T}x{H}x{W} C={C} Cout={Cout}: conv3d {t_conv:7.3f} ms = {fl/t_conv*1e-9:6.0f} TF   persistent GEMM on a patch matrix {t_gemm:7.3f} ms = {fl/t_gemm*1e-9:6.0f} TF"
          f"   default-tile GEMM {t_gemm128:7.3f} ms = {fl/t_gemm128*1e-9:6.0f} TF", flush=True)
