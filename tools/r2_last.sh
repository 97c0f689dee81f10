export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_fp8_gpu.py -q -s -k "depth" 2>&1 | grep -E "block|output|passed|failed|Error|error" | head -20
