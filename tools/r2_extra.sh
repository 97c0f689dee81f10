# extra end-of-round measurements: power picture of the final GEMM kernel, CFG batch on one GPU, 2-rank shard shapes
set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2extra
mkdir -p $O
timeout 600 python tools/gemm_probe.py --variants v4,w8 --data zeros --rounds 3 --shapes qkv,ff2,sq8192 --out $O/gemm_probe_zeros.json > $O/gemm_probe_zeros.log 2>&1; grep -v amdgpu $O/gemm_probe_zeros.log
timeout 600 python tools/gemm_probe.py --variants v4,w8 --data gaussian --rounds 3 --shapes qkv,ff2,sq8192 --out $O/gemm_probe_gauss.json > $O/gemm_probe_gauss.log 2>&1; grep -v amdgpu $O/gemm_probe_gauss.log
timeout 900 python bench.py --batch 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_b2.json 2> $O/bench_b2.err; echo "b2 rc=$?"; python -c "
import json;d=json.loads(open('$O/bench_b2.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['config']['workload'])"
timeout 300 python tools/shard_shape_probe.py --world 2 --out $O/shard_shapes_w2.json > $O/shard_w2.log 2>&1; tail -2 $O/shard_w2.log
