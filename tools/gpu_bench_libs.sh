#!/bin/bash
# timing-only A/B of experimental builds (no parity: experiments may break results)
for lib in "$@"; do
  export PT_SHIM_LIB=$(pwd)/oclpathtracer_amd/$lib
  timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline $BENCH_ARGS > gpurun_out/exp_${lib%.so}.log 2>&1
  python3 - <<PY
import json
for l in open("gpurun_out/exp_${lib%.so}.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("  $lib  %.1f Msamples/s  trace %.2f ms" % (d["value"], d["roofline"]["avg_launch_ms"]))
PY
done
