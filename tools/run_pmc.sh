# HBM traffic per kernel: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over one bench step.
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timers > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timers > $R/gpurun_out/pmc_write.log 2>&1
cd $R
python tools/pmc_aggregate.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/r1_pmc_traffic.json
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" -size +2M -delete
