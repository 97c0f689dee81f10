# HBM traffic per kernel: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over one bench step.
#   bash tools/run_pmc.sh <tag>      -> gpurun_out/<tag>_pmc_traffic.json  (copy into profiles/)
TAG=${1:-r2}
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 700 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_$TAG --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-fp8-variant --no-qk-gain-variant > $R/gpurun_out/pmc_fetch_$TAG.log 2>&1
timeout 700 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_$TAG --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-fp8-variant --no-qk-gain-variant > $R/gpurun_out/pmc_write_$TAG.log 2>&1
cd $R
python tools/pmc_aggregate.py gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG gpurun_out/${TAG}_pmc_traffic.json
find gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG -name "*.csv" -size +2M -delete
