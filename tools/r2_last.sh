export TMPDIR=/tmp
O=gpurun_out/r2last3
mkdir -p $O
timeout 1500 python -m pytest tests/test_forward_gpu.py -q -s -k "depth_42" > $O/pytest_depth.log 2>&1; echo "pytest rc=$?"; grep -E "fp8-engine|passed|failed" $O/pytest_depth.log | tail -12
