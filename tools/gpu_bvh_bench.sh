#!/bin/bash
# LBVH vs brute force on soups of growing size (exploration; small images so brute force finishes)
set -o pipefail
mkdir -p gpurun_out
run () { timeout -k 10 280 python bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-70s %10.2f Msamples/s  %9.2f ms/step' % (d['config']['workload'][:70], d['value'], d['ms_per_step']))"; }
run --soup 2000 --width 256 --height 256 --spp 16 --accel 1
run --soup 2000 --width 256 --height 256 --spp 16 --accel 2
run --soup 20000 --width 256 --height 256 --spp 4 --accel 1
run --soup 20000 --width 256 --height 256 --spp 4 --accel 2
run --soup 200000 --width 256 --height 256 --spp 16 --accel 2
run --soup 1000000 --width 512 --height 512 --spp 16 --accel 2
run --width 512 --height 512 --spp 64 --accel 2
run --width 512 --height 512 --spp 64 --accel 1
