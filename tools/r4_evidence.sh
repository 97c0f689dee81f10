# End-of-round evidence, one box, one call:  bash tools/r4_evidence.sh <tag>   (writes gpurun_out/<tag>_*)
TAG=${1:-r4_e}
export TMPDIR=/tmp
R=$PWD
python bench.py --steps 10 --warmup 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-fp8-variant > $R/gpurun_out/${TAG}_prof.log 2>&1
cd $R
cp $(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_bench_kernel_stats.csv
find gpurun_out/${TAG}_prof -name "*.csv" -size +2M -delete
bash tools/run_pmc.sh ${TAG} > gpurun_out/${TAG}_pmc.log 2>&1
bash tools/pmc_kernel.sh ${TAG}_gemm tools/gemm_only.py 17776 12288 3072 3 > gpurun_out/${TAG}_sq_counters_gemm.txt 2>&1
bash tools/pmc_kernel.sh ${TAG}_attn tools/attn_only.py 3 bounded > gpurun_out/${TAG}_sq_counters_attn.txt 2>&1
bash tools/pmc_kernel.sh ${TAG}_rowattn tools/router_group_attn_only.py 3 > gpurun_out/${TAG}_sq_counters_router_group_attn.txt 2>&1
find gpurun_out -name "*counter_collection.csv" -size +1M -delete
