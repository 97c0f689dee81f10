"""Builds tools/ablate/libbya_timeline.so: the shipped objects with gemm_v4.hip recompiled under -DBYA_GEMM_TIMELINE (wall-clock
stamps per tile of the unsplit persistent GEMM; tools/gemm_epilogue_timeline.py reads them).  Run after
`python -m bind_your_avatar_implementation_amd.build`; hipcc cross-compiles, no GPU needed."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bind_your_avatar_implementation_amd import build as B

out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ablate")
os.makedirs(out_dir, exist_ok=True)
B.build_hip_library(verbose=False)
hipcc = B._hipcc()
obj = os.path.join(out_dir, "gemm_v4_timeline.o")
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DBYA_GEMM_TIMELINE", "-c",
                       os.path.join(B.CSRC, "gemm_v4.hip"), "-o", obj])
objs = [obj if s == "gemm_v4.hip" else os.path.join(B.PKG_DIR, "build", s.replace(".hip", ".o")) for s in B.SOURCES]
lib = os.path.join(out_dir, "libbya_timeline.so")
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"])
print(lib)
