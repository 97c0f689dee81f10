export TMPDIR=/tmp
R=$PWD
cd /tmp
s=$(date +%s)
timeout 1500 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_r4_o --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-fp8-variant > $R/gpurun_out/pmc_fetch_r4_o.log 2>&1
echo "fetch pass rc=$? seconds=$(( $(date +%s) - s ))"
cd $R
cp -r gpurun_out/pmc_write_r4_n gpurun_out/pmc_write_r4_o 2>/dev/null
ls gpurun_out/pmc_fetch_r4_o/*/ | head
python tools/pmc_aggregate.py gpurun_out/pmc_fetch_r4_o gpurun_out/pmc_write_r4_o gpurun_out/r4_o_pmc_fetch_only.json | grep -i "attn_joint\|gemm256p\|rowattn" 
find gpurun_out/pmc_fetch_r4_o -name "*.csv" -size +2M -delete
python tools/mall_probe.py gpurun_out/r4_o_mall_probe.json 2>&1 | grep -v amdgpu
