# usage: bash tools/pmc_probe.sh <counter> <script> [args...]  -- one short PMC pass with a hard time limit
export TMPDIR=/tmp
R=$PWD; C=$1; shift
cd /tmp
timeout 150 rocprofv3 --pmc $C -d $R/gpurun_out/pmc_probe --output-format csv -- python3 $R/$@ > $R/gpurun_out/pmc_probe.log 2>&1
echo "rc=$? for $C $@"
cd $R
