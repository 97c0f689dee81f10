#!/usr/bin/env python3
"""Ablation builds of the W-stationary N = 512 row GEMM (csrc/rowgemm.hip rowgemm512q_kernel, -DBYA_ROWGEMM_ABLATE=mask; the
router's out-projections and mlp[2]: 336 launches of ~35 us per step) timed at 35100 rows with a residual.  Results are
meaningless, only the time is read.   python tools/rowgemm_q_ablate.py --build | --run [--out gpurun_out/x.json]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "bind_your_avatar_implementation_amd")
OUT = os.path.join(PKG, "build", "ablate")
VARIANTS = {"full": 0, "no_x_loads": 1, "no_stores": 2, "no_residual": 4, "no_w_preload": 8, "no_mfma": 16, "no_w_reads": 32,
            "no_memory": 7, "no_mfma_no_w_reads": 48, "only_memory": 56, "no_x_no_res": 5}


def build():
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, ROOT)
    from bind_your_avatar_implementation_amd.build import SOURCES
    objs = [os.path.join(PKG, "build", f.replace(".hip", ".o")) for f in SOURCES if f != "rowgemm.hip"]
    for name, mask in VARIANTS.items():
        obj = os.path.join(OUT, f"rowgemm_{name}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                               "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-slp-vectorize", f"-DBYA_ROWGEMM_ABLATE={mask}",
                               "-c", os.path.join(PKG, "csrc", "rowgemm.hip"), "-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, f"libbya_rowgemm_{name}.so")] + objs + [obj, "-ldl"])
        print("built", name)


CHILD = r'''
import json, sys, torch
sys.path.insert(0, ".")
from bind_your_avatar_implementation_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
rnd = lambda *s, std=1.0: (torch.randn(*s, generator=g) * std).to(torch.bfloat16).to(dev)
wo, bo = rnd(512, 512, std=512 ** -0.5), rnd(512, std=0.1)
po = ops.pack_rowgemm512(wo, bo)
def timed(fn, n=60):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)
res = {}
for M in (35100, 17550, 8788):
    a, x = rnd(M, 512), rnd(M, 512)
    res[f"res_M{M}"] = timed(lambda: ops.rowgemm512(a, po, x, res=x))
    res[f"plain_M{M}"] = timed(lambda: ops.rowgemm512(a, po, x))
print(json.dumps(res))
'''


def run(out):
    res = {}
    for name in VARIANTS:
        env = dict(os.environ, BYA_HIP_LIB=os.path.join(OUT, f"libbya_rowgemm_{name}.so"))
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, cwd=ROOT, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        res[name] = json.loads(line[-1]) if line else {"error": r.stderr[-400:]}
        print(name, res[name], flush=True)
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    if "--run" in sys.argv:
        run(sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None)
