# Last evidence of round 5, one box, one call:  bash tools/r5_final_evidence.sh   (writes gpurun_out/r5_final_*)
export TMPDIR=/tmp
R=$PWD
python -m pytest tests -q -m gpu --durations=12 > gpurun_out/r5_final_pytest_gpu.txt 2>&1
python __graft_entry__.py --smoke > gpurun_out/r5_final_smoke.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r5_final_bench.json 2> gpurun_out/r5_final_bench.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r5_final_prof --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-fp8-variant --no-qk-gain-variant > $R/gpurun_out/r5_final_prof.log 2>&1
cd $R
cp $(find gpurun_out/r5_final_prof -name "*kernel_stats.csv" | head -1) gpurun_out/r5_final_bench_kernel_stats.csv
find gpurun_out/r5_final_prof -name "*.csv" -size +1M -delete
bash tools/run_pmc.sh r5_final > gpurun_out/r5_final_pmc.log 2>&1
find gpurun_out -name "*counter_collection.csv" -size +1M -delete
python tools/solo_rank_step.py --world 8 --single --out gpurun_out/r5_final_solo_rank_step_w8.json > gpurun_out/r5_final_solo_w8.log 2>&1
