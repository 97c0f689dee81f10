# Round 2, GPU call 2: v3 GEMM correctness + probe, batched-load epilogue A/B, corruption classification, full tests, bench.
set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2c2
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm and not rowgemm" > $O/pytest_gemm.log 2>&1; echo "pytest gemm rc=$?"; tail -15 $O/pytest_gemm.log
timeout 900 python tools/gemm_probe.py --variants default,w4,v3 --data gaussian --rounds 4 --out $O/gemm_probe.json > $O/gemm_probe.log 2>&1; echo "probe rc=$?"; cat $O/gemm_probe.log
timeout 300 python tools/gemm_probe.py --variants default,v3 --data zeros --rounds 3 --shapes qkv,ff2,sq8192 --out $O/gemm_probe_zeros.json > $O/gemm_probe_zeros.log 2>&1; cat $O/gemm_probe_zeros.log
timeout 600 python tools/timeslice/repro.py --runs 20 --disturbers none,rowgemm_n1536_140k,torch_matmul,gemm256_128k --out $O/timeslice_repro.json > $O/timeslice.log 2>&1; echo "repro rc=$?"; tail -12 $O/timeslice.log
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc; grep -E "passed|failed|error" $O/pytest.log | tail -5
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1800 $O/bench.json
