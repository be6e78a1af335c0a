#!/bin/bash
# rocprofv3 PMC passes (counters only: never combined with tracing) for the bench workload.
# usage: tools/gpu_pmc.sh <tag> [spp]   -> gpurun_out/pmc_<tag>_<pass>/
set -o pipefail
TAG=${1:-r01}
SPP=${2:-256}
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
run_pass () {
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $REPO/gpurun_out/pmc_${TAG}_$name -o pmc -- python3 $REPO/bench.py --steps 1 --warmup 0 --spp $SPP --no-cpu-baseline --no-extra-configs $BENCH_ARGS > $REPO/gpurun_out/pmc_${TAG}_$name.log 2>&1
  echo "pass $name rc=$?"
}
run_pass sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pass sq2 SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
cd $REPO
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_${TAG}_*/")):
    for f in glob.glob(d + "*counter_collection.csv"):
        agg = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:40], r["Counter_Name"])
            agg[k] += float(r["Counter_Value"]); n[k] += 1
        for k in sorted(agg): print(d.split("/")[-2], k[0], k[1], "sum=%.6g" % agg[k], "dispatches=%d" % n[k])
PY
