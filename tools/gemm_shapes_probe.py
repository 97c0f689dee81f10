"""Times bya_gemm_bf16 on the step's big Linear shapes (one library per process: BYA_HIP_LIB selects an ablation build).
usage: python tools/gemm_shapes_probe.py [label]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bind_your_avatar_implementation_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
ops.ensure_gemm_workspace(dev)
SHAPES = [("ff1", 17776, 12288, 3072, "gelu_tanh", False), ("ff2", 17776, 3072, 12288, None, True),
          ("qkv", 17776, 9216, 3072, None, False), ("attn_out", 17776, 3072, 3072, None, True), ("to_q", 17550, 3072, 3072, None, False)]
row = {}
for name, M, N, K, act, has_res in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16) if has_res else None
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    best = 1e9
    for rep in range(3):
        for _ in range(5):
            ops.gemm(a, w, out, bias=b, res=res, act=act)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            ops.gemm(a, w, out, bias=b, res=res, act=act)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    row[name] = (round(best, 1), round(2.0 * M * N * K / best * 1e-6), float(out.float().abs().mean()))
ops.check_gemm_workspace()
print(sys.argv[1] if len(sys.argv) > 1 else "", " ".join(f"{k} {v[0]}us {v[1]}TF" for k, v in row.items()), " checksum", round(sum(v[2] for v in row.values()), 6), flush=True)
