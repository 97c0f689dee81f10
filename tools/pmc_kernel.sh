# usage: bash tools/pmc_kernel.sh <tag> <script> [args...]   -- SQ counters of one kernel, two passes
export TMPDIR=/tmp
R=$PWD; TAG=$1; shift
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/pmc_${TAG}_a --output-format csv -- python3 $R/$@ > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU -d $R/gpurun_out/pmc_${TAG}_b --output-format csv -- python3 $R/$@ > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace -d $R/gpurun_out/pmc_${TAG}_c --output-format csv -- python3 $R/$@ > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
for p in ("a", "b", "c"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
    for f in glob.glob("gpurun_out/pmc_${TAG}_%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
            if "gemm" in k or "attn" in k or "rowgemm" in k:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(p, k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
