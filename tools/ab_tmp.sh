export TMPDIR=/tmp
mkdir -p gpurun_out/fp8
timeout 1200 python -m pytest tests/test_fp8_gpu.py -q -s 2>&1 | grep -E "bytes differ|rel-Fro|unquantised|engine\(fp8\)|price of|passed|failed|Error|error" | head -40
timeout 600 python tools/gemm_probe.py --variants v4 --fp8 --data gaussian --rounds 3 --shapes ff2 2>&1 | grep -E "gaussian"
