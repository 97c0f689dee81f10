set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2c4
mkdir -p $O
BYA_GEMM_TILE=4 BYA_GEMM_VARIANT=v4 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm and not rowgemm and not whole_suite" > $O/pytest_v4.log 2>&1; echo "v4 suite rc=$?"; grep -E "passed|failed|FAILED|rel-Fro" $O/pytest_v4.log | tail -30
BYA_GEMM_VARIANT=v4 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm or routed_mix" > $O/pytest_v4b.log 2>&1; echo "v4 natural-tiles rc=$?"; grep -E "passed|failed|FAILED" $O/pytest_v4b.log | tail -8
timeout 900 python tools/gemm_probe.py --variants default,v3,v4 --data gaussian --rounds 4 --out $O/gemm_probe.json > $O/gemm_probe.log 2>&1; echo "probe rc=$?"; cat $O/gemm_probe.log
timeout 600 python tools/timeslice/repro.py --runs 20 --disturbers none,rowgemm_n1536_140k,torch_matmul --victims qknorm_rope_inplace,qknorm_rope_dbg1_inplace,qknorm_rope_dbg2_inplace --out $O/timeslice_repro_v3.json > $O/timeslice.log 2>&1; echo "repro rc=$?"; grep -E "^none|^rowgemm|^torch" $O/timeslice.log
