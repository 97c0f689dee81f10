"""Sweep of the K-split threshold of the persistent GEMM's last partial round (BYA_GEMM_SPLITK_MIN, read per call) on the
step's 3.28-round shapes.  usage: python tools/splitk_min_probe.py [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bind_your_avatar_implementation_amd import ops
from bind_your_avatar_implementation_amd import _hip  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
SHAPES = [(17776, 3072, 3072, True), (17550, 3072, 3072, False), (17550, 3072, 2048, True), (17550, 2048, 3072, False),
          (17550, 2048, 2048, False), (17776, 3072, 12288, True)]
ops.ensure_gemm_workspace(dev) if hasattr(ops, "ensure_gemm_workspace") else None
result = {}
for (M, N, K, has_res) in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16) if has_res else None
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ref = None
    row = {}
    for mn in (1000, 24, 16, 12, 8):
        os.environ["BYA_GEMM_SPLITK_MIN"] = str(mn)
        _hip.apply_env_options()      # (the library reads no environment: hand the change to its option table)
        best = 1e9
        for rep in range(3):
            for _ in range(3):
                ops.gemm(a, w, out, res=res)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                ops.gemm(a, w, out, res=res)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        if ref is None:
            ref = out.clone()
        row[str(mn)] = {"us": best, "tflops": 2.0 * M * N * K / best * 1e-6,
                        "diff_vs_unsplit": float((out.float() - ref.float()).abs().max())}
    result[f"{M}x{N}x{K}{'+res' if has_res else ''}"] = row
    print(M, N, K, has_res, json.dumps(row), flush=True)
ops.check_gemm_workspace()
if len(sys.argv) > 1:
    json.dump(result, open(sys.argv[1], "w"), indent=1)
