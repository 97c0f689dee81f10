# last call of the round: the default bench line at HEAD (with its fp8-weights variant object), wall-clocked
export TMPDIR=/tmp
O=gpurun_out/r2last2
mkdir -p $O
T0=$(date +%s)
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2last2/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['steps'], d['warmup'], d['roofline']['frac'], d['attn_roofline']['frac'])
print(d.get('fp8_weights_variant')); print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
