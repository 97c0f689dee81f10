set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2c14
mkdir -p $O
timeout 300 python tools/splitk_debug.py 2>&1 | grep -c "bad tiles 0"
timeout 1500 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -s -k "split_k" > $O/pytest_splitk.log 2>&1; echo "splitk rc=$?"; grep -E "split vs unsplit|passed|failed|Error|assert" $O/pytest_splitk.log | head -30
timeout 600 python tools/gemm_breakdown.py --out $O/gemm_breakdown_split.json > $O/gemm_breakdown_split.log 2>&1; head -10 $O/gemm_breakdown_split.log; grep -E "^total" $O/gemm_breakdown_split.log
BYA_GEMM_SPLITK=0 timeout 600 python tools/gemm_breakdown.py --out $O/gemm_breakdown_nosplit.json > $O/gemm_breakdown_nosplit.log 2>&1; head -10 $O/gemm_breakdown_nosplit.log; grep -E "^total" $O/gemm_breakdown_nosplit.log
timeout 300 python tools/shard_shape_probe.py --world 8 --out $O/shard_shapes_w8_split.json > $O/shard_w8.log 2>&1; grep -E "^gemm|projected|compute_per_rank" $O/shard_w8.log
BYA_GEMM_SPLITK=0 timeout 300 python tools/shard_shape_probe.py --world 8 > $O/shard_w8_nosplit.log 2>&1; grep -E "^gemm|compute_per_rank" $O/shard_w8_nosplit.log
