# last call of the round: the whole GPU suite at HEAD, with the slowest tests listed
export TMPDIR=/tmp
O=gpurun_out/r2last4
mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 $O/pytest_gpu.log
