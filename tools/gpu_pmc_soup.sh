#!/bin/bash
# rocprofv3 PMC passes (counters only) for the 10^6-triangle soup through the LBVH.  usage: tools/gpu_pmc_soup.sh <tag> [spp]
set -o pipefail
TAG=${1:-r02}
SPP=${2:-8}
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
run_pass () {
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $REPO/gpurun_out/pmcsoup_${TAG}_$name -o pmc -- python3 $REPO/bench.py --soup 1000000 --steps 1 --warmup 0 --spp $SPP --no-cpu-baseline --no-extra-configs > $REPO/gpurun_out/pmcsoup_${TAG}_$name.log 2>&1
  echo "pass $name rc=$?"
}
if [ -n "$SOUP_PMC_QUICK" ]; then
run_pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
else
run_pass sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pass sq2 SQ_WAVES SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS
run_pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
run_pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
run_pass fetch FETCH_SIZE
fi
cd $REPO
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmcsoup_${TAG}_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if "trace" not in r["Kernel_Name"]: continue
            k = (r["Kernel_Name"][:28], r["Counter_Name"])
            agg[k] += float(r["Counter_Value"]); n[k] += 1
        for k in sorted(agg): print(d.split("/")[-2], k[0], k[1], "sum=%.6g" % agg[k], "dispatches=%d" % n[k])
PY
