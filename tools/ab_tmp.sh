export TMPDIR=/tmp
mkdir -p gpurun_out/fp8
timeout 1500 python -m pytest tests/test_fp8_gpu.py -q -s 2>&1 | grep -E "bytes differ|rel-Fro|unquantised|engine\(fp8\)|price of|passed|failed|Error|error|sequence-parallel" | head -40
timeout 900 python bench.py --fp8-weights --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/fp8/bench_fp8_fused.json 2> /dev/null; tail -c 900 gpurun_out/fp8/bench_fp8_fused.json | head -c 700
