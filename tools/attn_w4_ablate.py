#!/usr/bin/env python3
"""Ablation builds of the hand-placed joint-attention kernel (csrc/attn_w4.hip): the generator leaves pieces of the hot
loop out (tools/gen_attn_w4_schedule.py --ablate MASK), every variant is compiled into a side copy of the library
(bind_your_avatar_implementation_amd/build/ablate/, built HERE or on the GPU box) and timed in a child process on the same
normalised gaussian data.  Results are meaningless, only the time is read: what is left when a piece goes away tells
what that piece costs.   python tools/attn_w4_ablate.py --build | --run [--out gpurun_out/x.json]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "bind_your_avatar_implementation_amd")
OUT = os.path.join(PKG, "build", "ablate")
VARIANTS = {"full": 0, "exp_as_mov": 1, "no_adds": 2, "no_cvt": 4, "no_lds_dma_barrier": 8, "q_in_vgpr": 16, "no_valu": 32,
            "no_valu_no_lds": 40, "no_adds_no_cvt": 6, "exp_only_as_mov": 7, "no_barrier": 128, "no_dma": 256,
            "no_ds_reads": 512, "no_lgkm_waits": 1024, "no_vmcnt_wait": 2048, "no_barrier_no_vmcnt": 2176, "no_k_reads": 4096, "no_v_reads": 8192, "v_reads_burst": 16384, "v_reads_b128": 32768, "dma_behind_valu": 65536}
if "--only" in sys.argv:
    keep = sys.argv[sys.argv.index("--only") + 1].split(",")
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in keep}


def build():
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(PKG, "build", f) for f in os.listdir(os.path.join(PKG, "build")) if f.endswith(".o") and f != "attn_w4.o"]
    for name, mask in VARIANTS.items():
        src = os.path.join(OUT, f"attn_w4_{name}.hip")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_attn_w4_schedule.py"), "--ablate", str(mask), "--out", src])
        txt = open(src).read().replace('#include "attn_common.h"', f'#include "{PKG}/csrc/attn_common.h"')
        txt = txt.replace('#include "bya_common.h"', f'#include "{PKG}/csrc/bya_common.h"')
        if mask & 32768:
            txt = txt.replace("u32x2 vh[2][4][2];", "u32x2 vh[2][4][2];\n    u32x4 vq[2][4] = {};")
        open(src, "w").write(txt)
        obj = src.replace(".hip", ".o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{PKG}/csrc", "-c", src, "-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, f"libbya_{name}.so")] + objs + [obj, "-ldl"])
        print("built", name)


CHILD = r'''
import sys, os, json, torch
sys.path.insert(0, %r)
from bind_your_avatar_implementation_amd import ops
dev = torch.device("cuda:0")
S, H, D = 17776, 48, 64
g = torch.Generator(device=dev).manual_seed(0)
nrm = lambda t: t / t.view(1, S, H, D).float().norm(dim=-1, keepdim=True).repeat_interleave(D, -1).view(1, S, H * D) * 8
q = nrm(torch.randn(1, S, H * D, device=dev, generator=g)).to(torch.bfloat16)
k = (nrm(torch.randn(1, S, H * D, device=dev, generator=g)).float() * (0.125 * 1.4426950408889634)).to(torch.bfloat16)
v = torch.randn(1, S, H * D, device=dev, generator=g).to(torch.bfloat16)
o = torch.empty_like(q)
f = lambda: ops.self_attention(q, k, v, o, heads=H, prescaled=True, score_bound=11.8)
for _ in range(5): f()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) / 20)
print(json.dumps({"ms": sorted(ts)[2], "ms_min": min(ts)}))
''' % ROOT


def run(out):
    res = {}
    for name in VARIANTS:
        lib = os.path.join(OUT, f"libbya_{name}.so")
        env = dict(os.environ, BYA_HIP_LIB=lib)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        res[name] = json.loads(line[0]) if line else {"error": (r.stdout + r.stderr)[-300:]}
        if "ms" in res[name]:
            res[name]["tflops_equiv"] = 4.0 * 17776 ** 2 * 64 * 48 / res[name]["ms"] / 1e9
        print(name, res[name], flush=True)
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    if "--run" in sys.argv:
        run(sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None)
