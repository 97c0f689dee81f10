set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2c6
mkdir -p $O
timeout 600 python tools/timeslice/repro.py --runs 20 --disturbers none,rowgemm_n1536_140k,torch_matmul --victims qknorm_rope_inplace,qknorm_rope_noslp_inplace,qknorm_norope_inplace --out $O/timeslice_repro_v4.json > $O/timeslice.log 2>&1; echo "repro rc=$?"; grep -E "^none|^rowgemm|^torch|groups" $O/timeslice.log | cut -c1-420
timeout 1200 python -m pytest tests/test_properties_gpu.py tests/test_weights_gpu.py tests/test_forward_gpu.py -m gpu -q -s -k "properties or weights or wide_aspect or G or sharded" > $O/pytest_new.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error|engine-vs|checkpoint" $O/pytest_new.log | tail -25
timeout 600 python bench.py --latent-hw 90 160 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_720x1280.json 2> $O/bench_720x1280.err; echo "bench720 rc=$?"; tail -c 1500 $O/bench_720x1280.json
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1200 $O/bench.json
