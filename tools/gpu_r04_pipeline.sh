#!/bin/bash
# round 4: the streaming renderer -- GPU tests, then the lanes / chunk-size matrix on configs[2]
# usage: tools/gpu_r04_pipeline.sh [tag] ; results under gpurun_out/r04/
set -o pipefail
TAG=${1:-a}
OUT=gpurun_out/r04
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/ -q -m gpu > $OUT/pytest_$TAG.log 2>&1
rc=$?; echo "pytest -m gpu rc=$rc"; tail -25 $OUT/pytest_$TAG.log
[ $rc -ne 0 ] && [ "$2" != "force" ] && exit $rc
line() {  # name, bench args...
  local name=$1; shift
  timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extra-configs "$@" > $OUT/bench_${TAG}_$name.json 2> $OUT/bench_${TAG}_$name.err || { echo "$name FAILED"; tail -5 $OUT/bench_${TAG}_$name.err; return 1; }
  python3 - $OUT/bench_${TAG}_$name.json $name <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
k, r = d["kernels"], d["roofline"]
n = k["pt_trace_kernel_launches"]
print("%-22s %8.1f Msamples/s  %7.3f ms/step  trace launches %4d  avg %.3f ms  union/launch %.3f ms  fold total/step %.3f ms  frac %.4f  frac_excl %.4f  ws %.0f MB"
      % (sys.argv[2], d["value"], d["ms_per_step"], n, k["pt_trace_kernel_ms_total"] / n, k["pt_trace_kernel_ms_union"] / n,
         k["pt_fold_kernel_ms_total"] / d["steps"], r["frac"], r["frac_exclusive"], d["config"]["workspace_bytes"] / 1e6))
PY
}
line default &&
line lanes1 --lanes 1 &&
line nocarry --checkpoint 0 &&
line nocarry_lanes1 --checkpoint 0 --lanes 1 &&
line onechunk --staging-mb 6400 &&
line onechunk_nocarry --staging-mb 6400 --checkpoint 0 --lanes 1 &&
line chunk8 --chunk-frames 8 &&
line chunk4 --chunk-frames 4 &&
line chunk32 --staging-mb 800 --chunk-frames 32
