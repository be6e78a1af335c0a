#!/bin/bash
# bench-only A/B of several builds of libptshim on one box, interleaved twice.  usage: tools/gpu_bench_libs.sh lib1.so lib2.so ...
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for lib in "$@"; do
  export PT_SHIM_LIB=$(pwd)/oclpathtracer_amd/$lib
  tag=${lib%.so}
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs > gpurun_out/bl_${tag}.log 2>&1
  python3 - <<PY
import json
for l in open("gpurun_out/bl_${tag}.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("  %-26s %.1f Msamples/s  trace %.2f ms  fold %.2f ms" % ("$lib", d["value"], d["roofline"]["avg_launch_ms"], d["kernels"]["pt_fold_kernel_ms_total"]/d["kernels"]["pt_fold_kernel_launches"]))
PY
done
done
