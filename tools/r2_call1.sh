# Round 2, GPU call 1: new parity tests, multi-process corruption reproducer, GEMM evidence probe (vendor / zeros /
# epilogues / power), vendor kernel names, effective clock, baseline bench on this box.
set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2c1
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc; tail -5 $O/pytest.log
timeout 600 python tools/timeslice/repro.py --runs 30 --out $O/timeslice_repro.json > $O/timeslice.log 2>&1; echo "repro rc=$?"; tail -20 $O/timeslice.log
timeout 900 python tools/gemm_probe.py --out $O/gemm_probe.json > $O/gemm_probe.log 2>&1; echo "probe rc=$?"; cat $O/gemm_probe.log
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gemm -- python3 $R/tools/gemm_probe.py --quick --data gaussian > $O/prof_gemm.log 2>&1; echo "prof rc=$?"
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_clk -- python3 $R/tools/gemm_probe.py --quick --shapes ff2,sq8192 > $O/pmc_clk.log 2>&1; echo "pmc rc=$?"
cd $R
find $O -name "*.csv" ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -size +3M -delete
find $O -name "*kernel_trace.csv" -size +3M -delete
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 2500 $O/bench.json
ls -la $O
