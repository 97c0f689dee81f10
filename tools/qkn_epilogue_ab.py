#!/usr/bin/env python3
"""A/B of the q|k|v projection's fused q/k-norm epilogue (bya_gemm_qkv_norm_rope) against the two launches (BYA_QKN_EPILOGUE=0),
whole bench.py steps in child processes, interleaved, one box.  An optional side build (BYA_HIP_LIB) can be put beside them.
  python tools/qkn_epilogue_ab.py        -> gpurun_out/r5_h_qkn_epilogue_ab.json"""
import os, sys, json, subprocess
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
out = {}
for rnd in range(2):
    for tag, env in (("branchfree", {}), *([("side_build", {"BYA_HIP_LIB": os.environ["QKN_AB_SIDE_LIB"]})] if os.environ.get("QKN_AB_SIDE_LIB") else []), ("two_launches", {"BYA_QKN_EPILOGUE": "0"})):
        r = subprocess.run([sys.executable, R + "/bench.py", "--no-cpu-baseline", "--no-fp8-variant", "--no-qk-gain-variant", "--steps", "6", "--warmup", "2"],
                           env=dict(os.environ, **env), capture_output=True, text=True)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        out.setdefault(tag, []).append((round(d["ms_per_step"], 2), d["kernel_ms_per_step"]["bya_gemm_bf16"], d["kernel_ms_per_step"].get("bya_qknorm_rope")))
        print(tag, out[tag][-1], flush=True)
json.dump(out, open(R + "/gpurun_out/r5_h_qkn_epilogue_ab.json", "w"), indent=1)
