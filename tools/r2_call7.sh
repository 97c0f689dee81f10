set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2c7
mkdir -p $O
timeout 600 python tools/timeslice/repro.py --runs 30 --disturbers none,rowgemm_n1536_140k,torch_matmul,gemm256_128k --victims qknorm_rope_inplace,layernorm_inplace,qknorm_norope_inplace --out $O/timeslice_repro_v5_product_build.json > $O/timeslice.log 2>&1; echo "repro rc=$?"; grep -E "^none|^rowgemm|^torch|^gemm" $O/timeslice.log | cut -c1-300
for i in 1 2 3; do timeout 900 python -m pytest tests/test_forward_gpu.py -m gpu -q -s -k "sequence_parallel or cfg_split" > $O/pytest_sp_$i.log 2>&1; echo "sp run $i rc=$?"; grep -E "passed|failed|vs single" $O/pytest_sp_$i.log | cut -c1-600; done
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $O/pytest_kernels.log
timeout 600 python tools/gemm_probe.py --variants v4,w8 --data gaussian --rounds 4 --shapes router_qkv,router_out,attn_out,ff2 --out $O/gemm_probe_router.json > $O/gemm_probe_router.log 2>&1; cat $O/gemm_probe_router.log
timeout 600 python tools/shard_shape_probe.py --world 8 --out $O/shard_shapes_w8.json > $O/shard_w8.log 2>&1; cat $O/shard_w8.log
timeout 300 python tools/shard_shape_probe.py --world 2 --out $O/shard_shapes_w2.json > $O/shard_w2.log 2>&1; tail -3 $O/shard_w2.log
