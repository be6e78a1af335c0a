#!/bin/bash
# A/B several builds of libptshim (same ABI) on one box.  usage: tools/gpu_ab.sh lib1.so lib2.so ...
# full parity file for the first library, the golden/oracle subset for the others, then the bench line of each
set -o pipefail
mkdir -p gpurun_out
first=1
for lib in "$@"; do
  export PT_SHIM_LIB=$(pwd)/oclpathtracer_amd/$lib
  tag=${lib%.so}
  if [ $first -eq 1 ]; then sel=""; first=0; else sel="-k golden or matches_oracle or specialisations"; fi
  timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q ${sel:+-k "${sel#-k }"} > gpurun_out/ab_${tag}_pytest.log 2>&1
  rc=$?; echo "$lib pytest rc=$rc $(tail -1 gpurun_out/ab_${tag}_pytest.log)"
  [ $rc -ne 0 ] && { tail -30 gpurun_out/ab_${tag}_pytest.log; continue; }
  for rep in 1 2; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs > gpurun_out/ab_${tag}_bench.log 2>&1
  python3 - <<PY
import json
for l in open("gpurun_out/ab_${tag}_bench.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("  $lib  %.1f Msamples/s  trace %.2f ms  fold %.2f ms" % (d["value"], d["roofline"]["avg_launch_ms"], d["kernels"]["pt_fold_kernel_ms_total"]/d["kernels"]["pt_fold_kernel_launches"]))
PY
  done
done
