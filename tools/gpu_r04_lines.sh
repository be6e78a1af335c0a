#!/bin/bash
# round 4: bench lines only (no tests): tools/gpu_r04_lines.sh TAG  name:args ...
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/r04
mkdir -p $OUT
for spec in "$@"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extra-configs $args > $OUT/bench_${TAG}_$name.json 2> $OUT/bench_${TAG}_$name.err || { echo "$name FAILED"; tail -5 $OUT/bench_${TAG}_$name.err; exit 1; }
  python3 - $OUT/bench_${TAG}_$name.json $name <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
k, r = d["kernels"], d["roofline"]
n = k["pt_trace_kernel_launches"]
print("%-22s %8.1f Msamples/s  %7.3f ms/step  trace launches %4d  sum/step %.3f ms  union/step %.3f ms  fold/step %.3f ms  frac %.4f  frac_excl %.4f  ws %.0f MB"
      % (sys.argv[2], d["value"], d["ms_per_step"], n, k["pt_trace_kernel_ms_total"] / d["steps"], k["pt_trace_kernel_ms_union"] / d["steps"],
         k["pt_fold_kernel_ms_total"] / d["steps"], r["frac"], r["frac_exclusive"], d["config"]["workspace_bytes"] / 1e6))
PY
done
