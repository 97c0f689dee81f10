set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2c8
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "rowgemm or router" > $O/pytest_rowgemm.log 2>&1; echo "rowgemm tests rc=$?"; tail -3 $O/pytest_rowgemm.log
timeout 600 python tools/gemm_probe.py --variants v4 --data gaussian --rounds 4 --shapes router_qkv,router_out --out $O/gemm_probe_router.json > $O/gemm_probe_router.log 2>&1; cat $O/gemm_probe_router.log
timeout 600 python tools/gemm_breakdown.py --out $O/gemm_breakdown.json > $O/gemm_breakdown.log 2>&1; cat $O/gemm_breakdown.log
timeout 600 python tools/attn_probe.py --out $O/attn_probe.json > $O/attn_probe.log 2>&1; cat $O/attn_probe.log
timeout 900 python -m pytest tests/test_forward_gpu.py -m gpu -q -x -k "golden or depth" > $O/pytest_fwd.log 2>&1; echo "fwd rc=$?"; tail -3 $O/pytest_fwd.log
