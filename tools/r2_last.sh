export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fp8_gpu.py -q 2>&1 | tail -3
