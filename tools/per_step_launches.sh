export TMPDIR=/tmp
R=$PWD
cd /tmp
for K in 1 5; do
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ps_$K --output-format csv -- python3 $R/bench.py --steps $K --warmup 1 --no-cpu-baseline --no-kernel-timers --no-fp8-variant --no-qk-gain-variant > $R/gpurun_out/ps_$K.log 2>&1
cp $(find $R/gpurun_out/ps_$K -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r5_y_stats_steps$K.csv
find $R/gpurun_out/ps_$K -name "*.csv" -size +1M -delete
done
