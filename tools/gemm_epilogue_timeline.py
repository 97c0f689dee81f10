"""Where an unsplit persistent-GEMM tile spends its time: K-loop vs epilogue, from wall-clock stamps inside the kernel.
Needs a library whose gemm_v4.hip was built with -DBYA_GEMM_TIMELINE (BYA_HIP_LIB points at it; same register allocation
as the shipped build).  usage: BYA_HIP_LIB=... python tools/gemm_epilogue_timeline.py [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bind_your_avatar_implementation_amd import ops
from bind_your_avatar_implementation_amd import _hip  # noqa: E402

dev = torch.device("cuda:0")
os.environ["BYA_GEMM_SPLITK_MIN"] = "1000"          # unsplit instance
_hip.apply_env_options()      # (the library reads no environment: hand the change to its option table)
ws = ops.ensure_gemm_workspace(dev)
SLAB = 256 * 256 * 4
SHAPES = [("ff1", 17776, 12288, 3072, "gelu_tanh", True, False), ("qkv", 17776, 9216, 3072, None, True, False),
          ("attn_out", 17776, 3072, 3072, None, True, True), ("to_q", 17550, 3072, 3072, None, False, False),
          ("ff2_unsplit", 17776, 3072, 12288, None, True, True)]
result = {}
for name, M, N, K, act, has_bias, has_res in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16) if has_bias else None
    res = torch.randn(M, N, device=dev).to(torch.bfloat16) if has_res else None
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(10):
        ops.gemm(a, w, out, bias=b, res=res, act=act)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(30):                                 # the stamps read below are the last launch's, at sustained clocks
        ops.gemm(a, w, out, bias=b, res=res, act=act)
    e1.record()
    torch.cuda.synchronize()
    raw = ws[4096 + 255 * SLAB: 4096 + 255 * SLAB + 256 * 2 * 8 * 8].cpu().numpy().view(np.uint64).reshape(256, 2, 8).astype(np.int64)
    v = raw[(raw[:, :, 7] > 0) & (raw[:, :, 6] == K // 64)]
    kl, ep = (v[:, 1] - v[:, 0]) / 100.0, (v[:, 4] - v[:, 1]) / 100.0
    result[name] = {"shape": [M, N, K], "act": act, "bias": has_bias, "res": has_res, "launch_us": e0.elapsed_time(e1) / 30 * 1e3,
                    "units": int(len(v)), "kloop_us": float(kl.mean()), "kloop_us_per_ktile": float(kl.mean() / (K // 64)),
                    "epilogue_us": float(ep.mean()), "epilogue_us_min_max": [float(ep.min()), float(ep.max())],
                    "epilogue_share": float(ep.mean() / (ep.mean() + kl.mean())),
                    "kloop_tflops_chip": float(256 * 2.0 * 256 * 256 * 64 / (kl.mean() / (K // 64)) * 1e-6)}
    print(name, json.dumps(result[name]), flush=True)
if len(sys.argv) > 1:
    json.dump(result, open(sys.argv[1], "w"), indent=1)
