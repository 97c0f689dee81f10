"""A/B of the two e4m3 GEMM kernels (128 x 128 two-barrier: gemm_fp8_kernel.h; persistent 256 x 256 hand-placed:
gemm_fp8_v4.hip) on the four DiT Linear shapes, same process (BYA_FP8_KERNEL is read per call); both against the exact
product of the same bytes.  usage: python tools/fp8_gemm_probe.py [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bind_your_avatar_implementation_amd import ops
from bind_your_avatar_implementation_amd import _hip  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
SHAPES = [("qkv", 17776, 9216, 3072, None, False), ("attn_out", 17776, 3072, 3072, None, True),
          ("ff1", 17776, 12288, 3072, "gelu_tanh", False), ("ff2", 17776, 3072, 12288, None, True),
          ("ragged", 1500, 1280, 1024, None, True)]
result = {}
for name, M, N, K, act, has_res in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16) if has_res else None
    a8, sa = ops.quantize_rows_fp8(a)
    w8, sw = ops.quantize_rows_fp8(w)
    # exact reference on a row sample (fp64 on the GPU)
    rows = torch.arange(0, M, max(1, M // 512), device=dev)
    ref = (a8[rows].view(torch.float8_e4m3fn).double() @ w8.view(torch.float8_e4m3fn).double().T) * sa[rows].double()[:, None] * sw.double()[None, :]
    ref = ref + b.double()
    if act:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    if has_res:
        ref = ref + res[rows].double()
    entry, outs = {}, {}
    for which in ("128", "v4"):
        os.environ["BYA_FP8_KERNEL"] = "128" if which == "128" else "v4"
        _hip.apply_env_options()      # (the library reads no environment: hand the change to its option table)
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        best = 1e9
        for rep in range(3):
            for _ in range(3):
                ops.gemm_fp8(a8, sa, w8, sw, out, bias=b, res=res, act=act)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                ops.gemm_fp8(a8, sa, w8, sw, out, bias=b, res=res, act=act)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        outs[which] = out
        entry[f"us_{which}"] = round(best, 1)
        entry[f"tflops_{which}"] = round(2.0 * M * N * K / best * 1e-6, 1)
        entry[f"err_{which}"] = float((out[rows].double() - ref).norm() / ref.norm())
        entry[f"finite_{which}"] = bool(torch.isfinite(out.float()).all())
    entry["v4_vs_128_max_abs"] = float((outs["v4"].float() - outs["128"].float()).abs().max())
    entry["v4_vs_128_mismatch_frac"] = float((outs["v4"] != outs["128"]).float().mean())
    result[name] = entry
    print(name, json.dumps(entry), flush=True)
if len(sys.argv) > 1:
    json.dump(result, open(sys.argv[1], "w"), indent=1)
