set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2c15
mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py -m gpu -q > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -4 $O/pytest_kernels.log
timeout 600 python tools/gemm_breakdown.py --out $O/gemm_breakdown.json > $O/gemm_breakdown.log 2>&1; head -10 $O/gemm_breakdown.log; grep -E "^total" $O/gemm_breakdown.log
timeout 2400 python -m pytest tests/test_forward_gpu.py tests/test_properties_gpu.py tests/test_weights_gpu.py -m gpu -q > $O/pytest_rest.log 2>&1; echo "rest rc=$?"; tail -4 $O/pytest_rest.log
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['kernel_ms_per_step'])"
