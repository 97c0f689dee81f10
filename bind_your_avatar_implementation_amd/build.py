"""Builds ``libbya_hip.so`` (the C-ABI library of hand-written gfx950 kernels) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the ``.so`` is
git-ignored but travels with the repo snapshot to the GPU box.
"""
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libbya_hip.so")
SOURCES = ["gemm.hip", "gemm_v4.hip", "gemm_v5.hip", "gemm_v6.hip", "gemm_fp8.hip", "gemm_fp8_v4.hip", "attn.hip", "attn_w4.hip", "norm.hip", "misc.hip", "router.hip", "rowgemm.hip", "rowchain.hip", "comm.hip", "vae.hip", "calib.hip"]
# VALU-only kernels are built WITHOUT the SLP vectoriser, i.e. without packed-fp32 (v_pk_mul/fma/mov_f32) instructions:
# with them the q/k-norm + RoPE kernel returned wrong values in lanes 48..63 of some waves whenever ANOTHER PROCESS kept
# MFMA-heavy workgroups resident on the same CUs (19-20 of 20 runs; 0 of 20 for the same source built with
# -fno-slp-vectorize; tools/timeslice/repro.py, profiles/history/r2_timeslice_repro_run*.json).  These
# kernels are HBM-bound, the packed forms bought nothing.
NO_SLP_SOURCES = {"norm.hip", "misc.hip", "router.hip", "gemm_fp8.hip", "vae.hip", "rowgemm.hip", "rowchain.hip"}     # (gemm_fp8: its row quantiser)
# translation units whose kernels keep their accumulators in AGPRs (one wave per SIMD, 512 registers)
AGPR_SOURCES = {"gemm_v4.hip", "gemm_v5.hip", "gemm_v6.hip", "gemm_fp8_v4.hip", "attn_w4.hip", "calib.hip"}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(PKG_DIR, "..", "include", "bya.h"),
                                                               os.path.abspath(__file__)]       # (compile flags live here)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_hip_library(force=False, verbose=True):
    if not force and not needs_build():
        return LIB_PATH
    objs = []
    build_dir = os.path.join(PKG_DIR, "build")
    os.makedirs(build_dir, exist_ok=True)
    hipcc = _hipcc()
    procs = []
    for src in SOURCES:
        obj = os.path.join(build_dir, src.replace(".hip", ".o"))
        # -amdgpu-mfma-vgpr-form: MFMA results stay in arch VGPRs (gfx950's register file is unified), which removes
        # the v_accvgpr_read/write traffic hipcc otherwise inserts wherever VALU code touches an accumulator.
        form = [] if src in AGPR_SOURCES else ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
        if src in NO_SLP_SOURCES:
            form = form + ["-fno-slp-vectorize"]
        # -fvisibility=hidden: the dynamic symbol table is include/bya.h (which pushes default visibility around its declarations)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", *form, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    # every symbol must resolve (a kernel whose host launch stub was not emitted only fails at dlopen time); checked
    # in a child process so this one keeps a single HIP runtime
    subprocess.check_call([sys.executable, "-c", f"import ctypes; ctypes.CDLL({LIB_PATH!r})"])
    return LIB_PATH


if __name__ == "__main__":
    print(build_hip_library(force="--force" in sys.argv))
