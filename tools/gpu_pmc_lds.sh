#!/bin/bash
# rocprofv3 PMC pass for the LDS side of the trace kernel (counters only).  usage: tools/gpu_pmc_lds.sh <tag> [spp]
set -o pipefail
TAG=${1:-x}; SPP=${2:-64}; REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d $REPO/gpurun_out/pmclds_${TAG} -o pmc -- python3 $REPO/bench.py --steps 1 --warmup 0 --spp $SPP --no-cpu-baseline --no-extra-configs > $REPO/gpurun_out/pmclds_${TAG}.log 2>&1
echo "rc=$?"
cd $REPO
python3 - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/pmclds_${TAG}/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "pt_trace" in r["Kernel_Name"] or "pt_fold" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:30], r["Counter_Name"])] += float(r["Counter_Value"])
    for k in sorted(agg): print(k[0], k[1], "%.6g" % agg[k])
PY
