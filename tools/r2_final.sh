# end-of-round evidence: full GPU suite, bench, rocprofv3 kernel stats, PMC traffic (run as ONE gpurun call: same box)
set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2final5
mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-fp8-variant > $O/prof.log 2>&1
cd $R
find $O/prof -name "*kernel_stats.csv" | head -2
find $O/prof -name "*.csv" ! -name "*kernel_stats.csv" -size +2M -delete
bash tools/run_pmc.sh r2_v5 > $O/pmc.log 2>&1; ls -la gpurun_out/r2_v5_pmc_traffic.json
cp gpurun_out/r2_v5_pmc_traffic.json profiles/r2_v5_pmc_traffic.json
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 2600 $O/bench.json
timeout 300 python tools/shard_shape_probe.py --world 8 --out $O/shard_shapes_w8.json > $O/shard_w8.log 2>&1; grep -E "joint attention|compute_per_rank" $O/shard_w8.log
timeout 600 python bench.py --latent-hw 90 160 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_720x1280.json 2> /dev/null; python -c "
import json;d=json.loads(open('$O/bench_720x1280.json').read().strip().splitlines()[-1]);print('720x1280', d['value'],d['ms_per_step'])"
timeout 600 python bench.py --latent-frames 25 --identities 3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_97f_3id.json 2> /dev/null; python -c "
import json;d=json.loads(open('$O/bench_97f_3id.json').read().strip().splitlines()[-1]);print('97f3id', d['value'],d['ms_per_step'])"
timeout 600 python bench.py --fp8-weights --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_fp8.json 2> /dev/null; python -c "
import json;d=json.loads(open('$O/bench_fp8.json').read().strip().splitlines()[-1]);print('fp8', d['value'],d['ms_per_step'])"
