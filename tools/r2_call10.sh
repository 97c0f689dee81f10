set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2c10
mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -s -k "split_k" > $O/pytest_splitk.log 2>&1; echo "splitk rc=$?"; grep -E "split vs unsplit|passed|failed|Error|assert" $O/pytest_splitk.log | head -30
timeout 1500 python -m pytest tests/test_kernels_gpu.py -m gpu -q > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -4 $O/pytest_kernels.log
timeout 600 python tools/gemm_breakdown.py --out $O/gemm_breakdown.json > $O/gemm_breakdown.log 2>&1; head -12 $O/gemm_breakdown.log; grep -E "^total|routed_mix|rowgemm" $O/gemm_breakdown.log
timeout 300 python tools/shard_shape_probe.py --world 8 --out $O/shard_shapes_w8.json > $O/shard_w8.log 2>&1; grep -E "^gemm|projected|compute_per_rank" $O/shard_w8.log
timeout 1500 python -m pytest tests/test_forward_gpu.py -m gpu -q -s -k "golden or depth or config0 or three_identities or small_geometry" > $O/pytest_fwd.log 2>&1; echo "fwd rc=$?"; grep -E "passed|failed|FAILED|output" $O/pytest_fwd.log | head
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['kernel_ms_per_step'])"
