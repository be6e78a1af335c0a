#!/bin/bash
# same-box A/B of the round-3 tree (git archive 4d8c396 under r03_ab/, built there) against this tree: bench.py, headline config
mkdir -p gpurun_out/r04
run() {  # dir label args...
  local dir=$1 label=$2; shift 2
  (cd $dir && timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernels']; n = k['pt_trace_kernel_launches']
        print('%-34s %.1f Msamples/s  %.3f ms per step; trace kernels %.3f ms per step in %d launches, folds %.3f' % ('$label', d['value'], d['ms_per_step'], k['pt_trace_kernel_ms_total'] / d['steps'], n // d['steps'], k['pt_fold_kernel_ms_total'] / d['steps']))")
}
for k in 1 2; do
  run r03_ab "round-3 tree"
  run . "this tree, shipped ring"
  run . "this tree, one launch per render" --staging-mb 6400 --lanes 1 --checkpoint 0
  run . "this tree, one launch + checkpoint" --staging-mb 6400 --lanes 1
  run . "this tree, shipped ring, lanes 1" --lanes 1
done | tee gpurun_out/r04/ab_r03_vs_r04.txt
