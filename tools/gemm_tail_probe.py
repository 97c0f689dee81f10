#!/usr/bin/env python3
"""The rows behind the last full round of 256 x 256 tiles (17776 x 3072 is 3.28 rounds; K = 3072 is too short for the K-split) on the
128 x 256 persistent kernel (csrc/gemm_v5.hip) as a second launch, against one launch that pays a whole fourth round: the product
build and a side build without the row split (-DBYA_GEMM_NO_P128_TAIL), the step's big Linears timed in child processes, alternating,
on gaussian operands.   python tools/gemm_tail_probe.py --build | --run [--out x.json]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "bind_your_avatar_implementation_amd")
OUT = os.path.join(PKG, "build", "ablate")


def build():
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, ROOT)
    from bind_your_avatar_implementation_amd.build import SOURCES
    objs = [os.path.join(PKG, "build", f.replace(".hip", ".o")) for f in SOURCES if f != "gemm.hip"]
    obj = os.path.join(OUT, "gemm_no_p128_tail.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                           "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-DBYA_GEMM_NO_P128_TAIL", "-c", os.path.join(PKG, "csrc", "gemm.hip"), "-o", obj])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                           os.path.join(OUT, "libbya_no_p128_tail.so")] + objs + [obj, "-ldl"])
    print("built")


CHILD = r'''
import json, sys, torch
sys.path.insert(0, ".")
from bind_your_avatar_implementation_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
rnd = lambda *s, std=1.0: (torch.randn(*s, generator=g) * std).to(torch.bfloat16).to(dev)
def timed(fn, n=30):
    for _ in range(6): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
res = {}
for name, M, N, K, kw in (("qkv 17776x9216x3072", 17776, 9216, 3072, {}), ("ff1 17776x12288x3072 gelu", 17776, 12288, 3072, {"act": "gelu_tanh"}),
                          ("ff2 17776x3072x12288 +res", 17776, 3072, 12288, {"res": True}), ("out 17776x3072x3072 +res", 17776, 3072, 3072, {"res": True}),
                          ("audio q 17550x3072x3072", 17550, 3072, 3072, {}), ("perceiver q 17550x2048x3072", 17550, 2048, 3072, {}),
                          ("perceiver out 17550x3072x2048 +res", 17550, 3072, 2048, {"res": True}), ("router q 17550x2048x2048", 17550, 2048, 2048, {})):
    a, w, b = rnd(M, K), rnd(N, K, std=K ** -0.5), rnd(N)
    out = rnd(M, N)
    args = dict(bias=b)
    if kw.get("res"): args["res"] = out
    if kw.get("act"): args["act"] = kw["act"]
    us = timed(lambda: ops.gemm(a, w, out, **args))
    res[name] = {"us": round(us, 1), "tflops": round(2.0 * M * N * K / us / 1e6)}
print(json.dumps(res))
'''


def run(out):
    res = {}
    for rep in range(3):                                   # twice, interleaved: the board's clock drifts over a run
        for lib in ("no_p128_tail", "product"):
            env = dict(os.environ, BYA_HIP_LIB=os.path.join(OUT, "libbya_no_p128_tail.so")) if lib != "product" else dict(os.environ)
            r = subprocess.run([sys.executable, "-c", CHILD], env=env, cwd=ROOT, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            res[f"{lib} run{rep}"] = json.loads(line[-1]) if line else {"error": r.stderr[-400:]}
            print(f"{lib} run{rep}", {k: v.get("tflops") for k, v in res[f"{lib} run{rep}"].items()} if line else res[f"{lib} run{rep}"], flush=True)
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    if "--run" in sys.argv:
        run(sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None)
