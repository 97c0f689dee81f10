# FETCH_SIZE / WRITE_SIZE (separate passes) of the kernels that are new or changed late in round 4 and of the persistent fp8
# GEMM (whose traffic file was still the old kernel's):  bash tools/r4_new_kernels_pmc.sh <tag>
export TMPDIR=/tmp
R=$PWD; TAG=${1:-r4_z}
cd /tmp
for spec in "fp8gemm tools/gemm_fp8_only.py 17776 12288 3072 3" "kvmix tools/kv_mix_only.py" "rowgemmq tools/rowgemm_only.py 512 0 1 0"; do
  set -- $spec; name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c -d $R/gpurun_out/pmc_${TAG}_${name}_$c --output-format csv -- python3 $R/$@ > /dev/null 2>&1
    echo "$name $c rc=$?"
  done
  python3 $R/tools/pmc_aggregate.py $R/gpurun_out/pmc_${TAG}_${name}_FETCH_SIZE $R/gpurun_out/pmc_${TAG}_${name}_WRITE_SIZE $R/gpurun_out/${TAG}_pmc_${name}.json
done
cd $R
python3 - <<PY
import json
for n in ("fp8gemm", "kvmix", "rowgemmq"):
    d = json.load(open("gpurun_out/${TAG}_pmc_%s.json" % n))
    for k, v in d.items():
        if isinstance(v, dict) and any(s in k for s in ("gemm256p_fp8", "kv_mix", "rowgemm512")):
            print(n, k[:60], {a: round(b / 1e6, 1) if "bytes" in a else b for a, b in v.items()})
PY
find gpurun_out -name "*counter_collection.csv" -size +1M -delete
