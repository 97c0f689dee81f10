#!/usr/bin/env python3
"""Ablation builds of the router chain kernels (csrc/rowchain.hip, -DBYA_ROWCHAIN_ABLATE=mask): every variant is compiled into a
side copy of the library (bind_your_avatar_implementation_amd/build/ablate/) and timed in a child process.  Results are
meaningless, only the time is read: what is left when a piece goes away tells what that piece costs.
python tools/rowchain_ablate.py --build | --run [--out gpurun_out/x.json]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "bind_your_avatar_implementation_amd")
OUT = os.path.join(PKG, "build", "ablate")
VARIANTS = {"full": 0, "no_compute": 1, "no_dma": 2, "no_gelu": 4, "no_w_reads": 8, "no_barrier": 16, "no_dma_no_barrier": 18,
            "no_w_reads_no_gelu": 12, "mfma_only": 30}


def build():
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, ROOT)
    from bind_your_avatar_implementation_amd.build import SOURCES
    objs = [os.path.join(PKG, "build", f.replace(".hip", ".o")) for f in SOURCES if f != "rowchain.hip"]
    for name, mask in VARIANTS.items():
        obj = os.path.join(OUT, f"rowchain_{name}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                               "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-slp-vectorize", f"-DBYA_ROWCHAIN_ABLATE={mask}",
                               "-c", os.path.join(PKG, "csrc", "rowchain.hip"), "-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, f"libbya_rowchain_{name}.so")] + objs + [obj, "-ldl"])
        print("built", name)


CHILD = r'''
import json, sys, torch
sys.path.insert(0, ".")
from bind_your_avatar_implementation_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
rnd = lambda *s, std=1.0: (torch.randn(*s, generator=g) * std).to(torch.bfloat16).to(dev)
w, b = rnd(1536, 512, std=512 ** -0.5), rnd(1536, std=0.1)
wo, bo = rnd(512, 512, std=512 ** -0.5), rnd(512, std=0.1)
w1, b1 = rnd(512, 512, std=512 ** -0.5), rnd(512, std=0.1)
gam, bet = torch.ones(512, dtype=torch.bfloat16, device=dev), torch.zeros(512, dtype=torch.bfloat16, device=dev)
pack, po, p1 = ops.pack_rowgemm512(w, b, gam, bet), ops.pack_rowgemm512(wo, bo), ops.pack_rowgemm512(w1, b1, gam, bet)
def timed(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)
res = {}
x0 = rnd(35100, 512)
for M, tp0s in ((35100, (8, 4)), (32768, (8,)), (17550, (5,))):
    x = x0[:M].clone()
    for tp0 in tp0s:
        res[f"mlp_M{M}_tp{tp0}"] = timed(lambda: ops.router_mlp_fused(x, p1, po, tiles_pass0=tp0))
x = x0.clone()
for tp0 in (8, 4):
    res[f"temporal_tp{tp0}"] = timed(lambda: ops.router_group_attn_out(x, pack, po, 13, 2, 1350, 17550, 1350, tiles_pass0=tp0))
    res[f"multi_id_tp{tp0}"] = timed(lambda: ops.router_group_attn_out(x, pack, po, 2, 1, 17550, 35100, 17550, tiles_pass0=tp0))
print(json.dumps(res))
'''


def run(out):
    res = {}
    for name in VARIANTS:
        env = dict(os.environ, BYA_HIP_LIB=os.path.join(OUT, f"libbya_rowchain_{name}.so"))
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
