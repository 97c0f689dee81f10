export TMPDIR=/tmp
mkdir -p gpurun_out/fp8
timeout 1500 python -m pytest tests/test_fp8_gpu.py -q -s 2>&1 | grep -E "bytes differ|engine\(fp8\)|price of|passed|failed|Error|error|sequence-parallel" | head -30
timeout 900 python bench.py --fp8-weights --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/fp8/bench_fp8_q.json 2> /dev/null; python -c "
import json;d=json.loads(open('gpurun_out/fp8/bench_fp8_q.json').read().strip().splitlines()[-1]);print('fp8', d['value'],d['ms_per_step']); print(d['kernel_ms_per_step'])"
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/fp8/bench_bf16_q.json 2> /dev/null; python -c "
import json;d=json.loads(open('gpurun_out/fp8/bench_bf16_q.json').read().strip().splitlines()[-1]);print('bf16', d['value'],d['ms_per_step'])"
