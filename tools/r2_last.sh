# rocprofv3 kernel stats of the fp8-weights step (evidence for bya_gemm_fp8 / the quantiser kernels)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2fp8prof
mkdir -p $O
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof --output-format csv -- python3 $R/bench.py --fp8-weights --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers > $O/prof.log 2>&1
cd $R
K=$(find $O/prof -name "*kernel_stats.csv" | head -1); echo $K; head -12 $K | cut -c1-170
find $O/prof -name "*.csv" ! -name "*kernel_stats.csv" -size +2M -delete
tail -c 300 $O/prof.log
