set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2c16
mkdir -p $O
timeout 900 python tools/attn_ablate.py --out $O/attn_ablate.json > $O/attn_ablate.log 2>&1; cat $O/attn_ablate.log | grep -v amdgpu.ids
timeout 1800 python -m pytest tests/test_forward_gpu.py -m gpu -q -s -k "sequence_parallel or cfg_split" > $O/pytest_sp.log 2>&1; echo "sp rc=$?"; grep -E "passed|failed|vs single|FAILED" $O/pytest_sp.log | cut -c1-400
