set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2c5
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc; grep -E "passed|failed|FAILED|error" $O/pytest.log | tail -8
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 3500 $O/bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers > $O/prof.log 2>&1; echo "prof rc=$?"
cd $R
find $O/prof -name "*.csv" ! -name "*kernel_stats.csv" -size +2M -delete
bash tools/run_pmc.sh r2_v1; echo "pmc rc=$?"
ls $O gpurun_out | head -40
