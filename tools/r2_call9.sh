set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2c9
mkdir -p $O
timeout 900 python -m pytest tests/test_forward_gpu.py -m gpu -q -s -x -k "three_identities" > $O/pytest_3id.log 2>&1; echo "3id rc=$?"; grep -E "engine-vs|passed|failed|Error|error|forcing" $O/pytest_3id.log | head -30
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -5 $O/pytest_kernels.log
timeout 1500 python -m pytest tests/test_forward_gpu.py -m gpu -q -s -k "golden or small_geometry or config0 or wide_aspect or both_softmax or two_ranks" > $O/pytest_fwd.log 2>&1; echo "fwd rc=$?"; grep -E "passed|failed|router logits|FAILED" $O/pytest_fwd.log | head
timeout 300 python tools/shard_shape_probe.py --world 8 --out $O/shard_shapes_w8.json > $O/shard_w8.log 2>&1; grep -E "joint attention|projected|compute_per_rank" $O/shard_w8.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['kernel_ms_per_step'])"
