# PMC traffic of the fp8-weights step (two separate passes, as tools/run_pmc.sh)
TAG=r2_fp8
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 700 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_$TAG --output-format csv -- python3 $R/bench.py --fp8-weights --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timers > $R/gpurun_out/pmc_fetch_$TAG.log 2>&1
timeout 700 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_$TAG --output-format csv -- python3 $R/bench.py --fp8-weights --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timers > $R/gpurun_out/pmc_write_$TAG.log 2>&1
cd $R
python tools/pmc_aggregate.py gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG gpurun_out/${TAG}_pmc_traffic.json | grep -E "gemm_fp8|quant_rows|layernorm_kernel<8, 6, true>|gemm256p"
find gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG -name "*.csv" -size +2M -delete
