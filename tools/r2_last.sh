# last call of the round: the whole GPU suite at HEAD + BASELINE configs[4] on one GPU (97 frames, 3 identities, fp8 weights)
set -x
export TMPDIR=/tmp
O=gpurun_out/r2last
mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --latent-frames 25 --identities 3 --fp8-weights --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_97f_3id_fp8.json 2> /dev/null; python -c "
import json;d=json.loads(open('$O/bench_97f_3id_fp8.json').read().strip().splitlines()[-1]);print('97f3id fp8', d['value'],d['ms_per_step'])"
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
