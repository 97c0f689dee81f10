set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2c22
mkdir -p $O
for rep in 1 2; do for v in a b c; do
BYA_HIP_LIB=$R/tools/ablate/libbya_hip_$v.so timeout 600 python tools/gemm_probe.py --variants v4,w8 --data gaussian --rounds 3 --shapes qkv,ff1,attn_out,audio_q,sq8192 --out $O/probe_${v}_$rep.json > $O/probe_${v}_$rep.log 2>&1; echo "== variant $v rep $rep"; grep -v amdgpu $O/probe_${v}_$rep.log
done; done
