#!/usr/bin/env python3
"""Ablation builds of the 128 x 256 persistent GEMMs (csrc/gemm_v5.hip, -DBYA_GEMM5_ABLATE=mask; with --v6: csrc/gemm_v6.hip, the
form with loader waves, -DBYA_GEMM6_ABLATE=mask): every variant is compiled into a side copy of the library
(bind_your_avatar_implementation_amd/build/ablate/) and timed in a child process.  Results are meaningless, only the time is read:
what is left when a piece goes away tells what that piece costs.
python tools/gemm_p128_ablate.py [--v6] --build | --run [--out gpurun_out/x.json]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "bind_your_avatar_implementation_amd")
OUT = os.path.join(PKG, "build", "ablate")
VARIANTS = {"full": 0, "no_epilogue": 1, "no_dma": 2, "no_barrier": 4, "no_frag_reads": 8, "no_dma_no_barrier": 6, "mfma_only": 15,
            "no_epilogue_no_dma": 3, "all_l2_hits": 16, "w_pieces_only": 32, "a_pieces_only": 64}
V6 = "--v6" in sys.argv
if V6:
    VARIANTS = {"full": 0, "no_epilogue": 1, "no_dma": 2, "no_epilogue_no_dma": 3, "all_l2_hits": 16}
SRC, MACRO, TILE = ("gemm_v6.hip", "BYA_GEMM6_ABLATE", 6) if V6 else ("gemm_v5.hip", "BYA_GEMM5_ABLATE", 5)


def build():
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, ROOT)
    from bind_your_avatar_implementation_amd.build import SOURCES
    objs = [os.path.join(PKG, "build", f.replace(".hip", ".o")) for f in SOURCES if f != SRC]
    procs = []
    for name, mask in VARIANTS.items():
        obj = os.path.join(OUT, f"{SRC[:-4]}_{name}.o")
        procs.append((name, obj, subprocess.Popen(
            ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
             f"-D{MACRO}={mask}", "-c", os.path.join(PKG, "csrc", SRC), "-o", obj])))
    for name, obj, pr in procs:
        assert pr.wait() == 0, name
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, f"libbya_{SRC[:-4]}_{name}.so")] + objs + [obj, "-ldl"])
        print("built", name)


CHILD = r'''
import json, sys, torch
sys.path.insert(0, ".")
from bind_your_avatar_implementation_amd import ops
dev = torch.device("cuda:0")
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
res = {}
with ops.options(gemm_tile=TILE_CODE):
    for M, N, K in ((2222, 3072, 3072), (2222, 9216, 3072), (4444, 12288, 3072), (2222, 3072, 12288)):
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev).to(torch.bfloat16)
        o = torch.randn(M, N, device=dev).to(torch.bfloat16)
        for name, kw in (("plain", {}), ("bias", {"bias": b}), ("bias+res", {"bias": b, "res": o})):
            t = timed(lambda: ops.gemm(x, w, o, **kw))
            res[f"{M}x{N}x{K} {name}"] = [round(t, 1), round(2.0 * M * N * K / t / 1e6, 0)]
print(json.dumps(res))
'''


def run(out):
    res = {}
    for name in VARIANTS:
        env = dict(os.environ, BYA_HIP_LIB=os.path.join(OUT, f"libbya_{SRC[:-4]}_{name}.so"))
        r = subprocess.run([sys.executable, "-c", CHILD.replace("TILE_CODE", str(TILE))], env=env, cwd=ROOT, capture_output=True, text=True)
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
