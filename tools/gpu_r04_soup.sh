#!/bin/bash
# configs[4] scene on one GPU: the shipped ring (16 launches per render, lanes alternate per chunk) against one launch per render, same box
mkdir -p gpurun_out/r04
for spec in "default:" "nocarry:--checkpoint 0" "onechunk:--staging-mb 6400 --lanes 1" "onechunk_nocarry:--staging-mb 6400 --lanes 1 --checkpoint 0"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout -k 10 400 python bench.py --config 4 --steps 2 --warmup 1 $args > gpurun_out/r04/soup_$name.json 2> gpurun_out/r04/soup_$name.err || { tail -3 gpurun_out/r04/soup_$name.err; exit 1; }
  grep -h '^{' gpurun_out/r04/soup_$name.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=d['kernels']
print('soup %-10s %.1f Msamples/s  ms/step %.1f  trace launches %d sum %.1f union %.1f ms' % ('$name', d['value'], d['ms_per_step'], k['pt_trace_kernel_launches'], k['pt_trace_kernel_ms_total'], k['pt_trace_kernel_ms_union']))"
done
