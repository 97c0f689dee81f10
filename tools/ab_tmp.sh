export TMPDIR=/tmp
mkdir -p gpurun_out/fp8
timeout 1200 python -m pytest tests/test_fp8_gpu.py -x -q -s 2>&1 | tail -40
for t in 128 256; do
echo "== tile $t"
BYA_FP8_TILE=$t timeout 600 python tools/gemm_probe.py --variants v4 --fp8 --data gaussian --rounds 3 --shapes qkv,ff1,ff2,attn_out --out gpurun_out/fp8/probe_tile$t.json 2>&1 | grep gaussian
done
timeout 900 python bench.py --fp8-weights --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/fp8/bench_fp8.json 2> gpurun_out/fp8/bench_fp8.err; tail -c 1500 gpurun_out/fp8/bench_fp8.json; tail -3 gpurun_out/fp8/bench_fp8.err
