set -x
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_r1_v4.json 2> gpurun_out/bench_r1_v4.err; tail -c 3000 gpurun_out/bench_r1_v4.json
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_v4 --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers > $R/gpurun_out/prof_v4.log 2>&1
cd $R
find gpurun_out/prof_v4 -name "*kernel_stats.csv" | head; 
find gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/prof_v4 -name "*.csv" ! -name "*kernel_stats.csv" -size +2M -delete
