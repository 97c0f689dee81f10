set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2c3
mkdir -p $O
timeout 900 python tools/timeslice/repro.py --runs 20 --disturbers none,rowgemm_n1536_140k,rowgemm_sentinel,torch_matmul --victims qknorm_rope_inplace,qknorm_rope_sc1_inplace,bcast_table,bcast_table_sc1,qknorm_norope_inplace,layernorm_inplace --out $O/timeslice_repro_v2.json > $O/timeslice.log 2>&1; echo "repro rc=$?"; tail -30 $O/timeslice.log | cut -c1-900
