# last call of the round: the whole GPU suite at HEAD, then the default bench line (with its fp8-weights variant object)
set -x
export TMPDIR=/tmp
O=gpurun_out/r2last2
mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
/usr/bin/time -v timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep -E "Elapsed|Maximum resident" $O/bench.err; tail -c 1200 $O/bench.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
