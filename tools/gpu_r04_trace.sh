#!/bin/bash
# kernel timeline of a few steps (rocprofv3 --kernel-trace): tools/gpu_r04_trace.sh TAG [bench args]
TAG=$1; shift
REPO=$(pwd)
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/r04/trace_$TAG -o t -- python3 $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra-configs "$@" > $REPO/gpurun_out/r04/trace_$TAG.log 2>&1
echo "rocprof rc=$?"
cd $REPO
f=$(find gpurun_out/r04/trace_$TAG -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
sel = [r for r in rows if "pt_trace" in r["Kernel_Name"] or "pt_fold_kernel" in r["Kernel_Name"]]
# the last ~40 launches
for r in sel[-40:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-16s start %12.3f ms  dur %8.3f ms  queue %s" % (r["Kernel_Name"][:16], s / 1e6, (e - s) / 1e6, r.get("Queue_Id", "")))
PY
