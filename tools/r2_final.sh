# end-of-round evidence: full GPU suite, bench, rocprofv3 kernel stats, PMC traffic, vendor probe, extra geometries
set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2final
mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 2500 $O/bench.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers > $O/prof.log 2>&1
cd $R
find $O/prof -name "*kernel_stats.csv" | head -2
find $O/prof -name "*.csv" ! -name "*kernel_stats.csv" -size +2M -delete
bash tools/run_pmc.sh r2_v2 > $O/pmc.log 2>&1; ls -la gpurun_out/r2_v2_pmc_traffic.json
timeout 600 python tools/gemm_probe.py --variants v4,w8 --data gaussian --rounds 4 --shapes qkv,ff1,ff2,attn_out,audio_q,perc_q,sq8192 --out $O/gemm_probe_final.json > $O/gemm_probe_final.log 2>&1; grep -v amdgpu $O/gemm_probe_final.log; timeout 300 python tools/gemm_breakdown.py --out $O/gemm_breakdown.json > $O/gemm_breakdown.log 2>&1; head -10 $O/gemm_breakdown.log; grep "^total" $O/gemm_breakdown.log
timeout 600 python bench.py --latent-frames 25 --identities 3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_97f_3id.json 2> $O/bench_97f_3id.err; echo "bench97 rc=$?"; tail -c 1200 $O/bench_97f_3id.json
timeout 300 python tools/shard_shape_probe.py --world 8 --out $O/shard_shapes_w8.json > $O/shard_w8.log 2>&1; tail -3 $O/shard_w8.log
