#!/bin/bash
# same-box A/B of the fold kernel: libptshim_old.so (the commit before) against libptshim.so, one render lane (a fold that runs beside the other
# lane's trace launch reads long), trace / fold totals per step from the bench's HIP events
set -o pipefail
for lib in libptshim_old.so libptshim.so libptshim_old.so libptshim.so; do
  PT_SHIM_LIB=$(pwd)/oclpathtracer_amd/$lib timeout -k 10 200 python bench.py --steps 10 --warmup 3 --lanes 1 --no-cpu-baseline --no-extra-configs > /tmp/ab_fold.json || exit 1
  python3 - "$lib" <<'PY'
import json, sys
for l in open("/tmp/ab_fold.json"):
    if l.startswith("{"):
        d = json.loads(l); k = d["kernels"]
        print("%-20s %.3f ms per step; trace %.3f ms, folds %.3f ms per step" % (sys.argv[1], d["ms_per_step"], k["pt_trace_kernel_ms_total"] / d["steps"], k["pt_fold_kernel_ms_total"] / d["steps"]))
PY
done
