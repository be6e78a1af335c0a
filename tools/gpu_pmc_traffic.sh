#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the trace and fold kernels for a given build.  usage: PT_SHIM_LIB=... tools/gpu_pmc_traffic.sh <tag> [spp]
set -o pipefail
TAG=${1:-x}; SPP=${2:-256}; REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $REPO/gpurun_out/pmct_${TAG}_$c -o pmc -- python3 $REPO/bench.py --steps 1 --warmup 0 --spp $SPP --no-cpu-baseline --no-extra-configs $BENCH_ARGS > $REPO/gpurun_out/pmct_${TAG}_$c.log 2>&1
done
cd $REPO
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmct_${TAG}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "trace" if "pt_trace" in r["Kernel_Name"] else "fold" if "pt_fold" in r["Kernel_Name"] else None
        if k: agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
n = 1024 * 1024 * $SPP
for k in ("trace", "fold"):
    f, w = agg[(k, "FETCH_SIZE")], agg[(k, "WRITE_SIZE")]
    print("$TAG %-5s fetch %.2f B/sample (x2 = %.2f)  write %.2f B/sample  hbm %.2f" % (k, f * 1024 / n, 2 * f * 1024 / n, w * 1024 / n, (2 * f + w) * 1024 / n))
PY
