# The evidence set of round 6, from ONE commit:   bash tools/r6_final_evidence.sh [A] [B] [C] [D]   (default: all four parts)
# writes gpurun_out/r6_final_*; copy into profiles/ what is to be judged (profiles/README.md lists the set).
#   A  GPU suite with durations, smoke, the bench line (20 steps)            ~15 min
#   B  rocprofv3 kernel stats + the two PMC passes (FETCH_SIZE first)        ~20 min
#   C  solo rank-steps W = 2 / 4 / 8, shard shapes, router chain probe + ablations  ~8 min
#   D  bench.py --gpus 2 / 4 / 8 at full depth with the ranks on this one GPU (rehearsal of the N > 1 path)  ~15 min
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
PARTS="${*:-A B C D}"
for P in $PARTS; do
case $P in
A)
  python -m pytest tests -q -m gpu --durations=25 > gpurun_out/r6_final_pytest_gpu.txt 2>&1
  python __graft_entry__.py --smoke > gpurun_out/r6_final_smoke.txt 2>&1
  python bench.py --steps 20 --warmup 5 > gpurun_out/r6_final_bench.json 2> gpurun_out/r6_final_bench.err
  ;;
B)
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r6_final_prof --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-fp8-variant --no-qk-gain-variant --no-calibration > $R/gpurun_out/r6_final_prof.log 2>&1
  cd $R
  cp $(find gpurun_out/r6_final_prof -name "*kernel_stats.csv" | head -1) gpurun_out/r6_final_bench_kernel_stats.csv
  find gpurun_out/r6_final_prof -name "*.csv" -size +1M -delete
  cd /tmp
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 1500 rocprofv3 --pmc $C -d $R/gpurun_out/pmc_${C}_r6_final --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-fp8-variant --no-qk-gain-variant --no-calibration > $R/gpurun_out/pmc_${C}_r6_final.log 2>&1
  done
  cd $R
  python tools/pmc_aggregate.py gpurun_out/pmc_FETCH_SIZE_r6_final gpurun_out/pmc_WRITE_SIZE_r6_final gpurun_out/r6_final_pmc_traffic.json > gpurun_out/r6_final_pmc.log 2>&1
  find gpurun_out/pmc_FETCH_SIZE_r6_final gpurun_out/pmc_WRITE_SIZE_r6_final -name "*.csv" -size +2M -delete
  ;;
C)
  for W in 8 4 2; do
    python tools/solo_rank_step.py --world $W --steps 5 $([ $W = 8 ] && echo --single) --out gpurun_out/r6_final_solo_rank_step_w$W.json > gpurun_out/r6_final_solo_w$W.log 2>&1
  done
  python tools/shard_shape_probe.py --world 8 --out gpurun_out/r6_final_shard_shapes_w8.json > gpurun_out/r6_final_shard_shapes_w8.log 2>&1
  python tools/router_chain_probe.py gpurun_out/r6_final_router_chain_probe.json > gpurun_out/r6_final_router_chain_probe.log 2>&1
  python tools/rowchain_ablate.py --build --run --out gpurun_out/r6_final_rowchain_ablate.json > gpurun_out/r6_final_rowchain_ablate.log 2>&1
  python tools/rowgemm_q_ablate.py --build --run --out gpurun_out/r6_final_rowgemm_q_ablate.json > gpurun_out/r6_final_rowgemm_q_ablate.log 2>&1
  ;;
D)
  for N in 2 4 8; do
    BYA_BENCH_SHARE_GPU=1 timeout 1200 python bench.py --gpus $N --steps 3 --warmup 2 --no-cpu-baseline --no-fp8-variant --no-qk-gain-variant > gpurun_out/r6_final_bench_${N}_ranks_on_one_gpu_42_layers.json 2> gpurun_out/r6_final_bench_${N}_ranks.err
  done
  ;;
esac
done
