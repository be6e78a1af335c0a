#!/bin/bash
# A/B several builds of libptshim (same ABI) on one box: parity + bench each
set -o pipefail
mkdir -p gpurun_out
for lib in "$@"; do
  export PT_SHIM_LIB=$(pwd)/oclpathtracer_amd/$lib
  tag=${lib%.so}
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/ab_${tag}_pytest.log 2>&1
  rc=$?; echo "$lib pytest rc=$rc $(tail -1 gpurun_out/ab_${tag}_pytest.log)"
  [ $rc -ne 0 ] && continue
  for rep in 1 2; do
  timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/ab_${tag}_bench.log 2>&1
  python3 - <<PY
import json
for l in open("gpurun_out/ab_${tag}_bench.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("  $lib  %.1f Msamples/s  trace %.2f ms  fold %.2f ms" % (d["value"], d["roofline"]["avg_launch_ms"], d["kernels"]["pt_fold_kernel_ms_total"]/d["kernels"]["pt_fold_kernel_launches"]))
PY
  done
done
