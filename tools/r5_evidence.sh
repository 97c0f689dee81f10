# End-of-round evidence, one box, one call:  bash tools/r5_evidence.sh <tag>   (writes gpurun_out/<tag>_*)
TAG=${1:-r5_z}
export TMPDIR=/tmp
R=$PWD
python bench.py --steps 10 --warmup 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-fp8-variant --no-qk-gain-variant > $R/gpurun_out/${TAG}_prof.log 2>&1
cd $R
cp $(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_bench_kernel_stats.csv
find gpurun_out/${TAG}_prof -name "*.csv" -size +2M -delete
bash tools/run_pmc.sh ${TAG} > gpurun_out/${TAG}_pmc.log 2>&1
# what one GPU can measure of the N-GPU step: one rank's real step alone (solo), and the per-kernel sum (shard shapes)
for W in 8 4 2; do
  python tools/solo_rank_step.py --world $W $([ $W = 8 ] && echo --single) --out gpurun_out/${TAG}_solo_rank_step_w$W.json > gpurun_out/${TAG}_solo_w$W.log 2>&1
  python tools/shard_shape_probe.py --world $W --out gpurun_out/${TAG}_shard_shapes_w$W.json > gpurun_out/${TAG}_shard_shapes_w$W.log 2>&1
done
find gpurun_out -name "*counter_collection.csv" -size +1M -delete
# the driver's N > 1 command shape with all ranks on this one GPU (BYA_BENCH_SHARE_GPU=1: gloo group; never a measurement): the
# whole path of bench.py -- unsharded reference, ladder, bit-identity check, timing -- at 8 ranks and as the 2 x 2 CFG split
BYA_BENCH_SHARE_GPU=1 python bench.py --gpus 8 --layers 2 --steps 2 --warmup 1 --no-kernel-timers > gpurun_out/${TAG}_bench_8_ranks_on_one_gpu_2_layers.json 2> gpurun_out/${TAG}_bench_8_ranks.err
BYA_BENCH_SHARE_GPU=1 python bench.py --gpus 4 --batch 2 --layers 2 --steps 2 --warmup 1 --no-kernel-timers > gpurun_out/${TAG}_bench_cfg_2x2_on_one_gpu_2_layers.json 2> gpurun_out/${TAG}_bench_cfg_2x2.err
python __graft_entry__.py --smoke > gpurun_out/${TAG}_smoke.txt 2>&1
# SURVEY 8(d) form of the CPU baseline (configs[0] through the oracle's transformer.forward; minutes of CPU on the box's host)
python bench.py --steps 5 --warmup 2 --cpu-baseline-config0 --no-fp8-variant --no-qk-gain-variant > gpurun_out/${TAG}_cpu_baseline_config0.json 2> gpurun_out/${TAG}_cpu_baseline_config0.err
