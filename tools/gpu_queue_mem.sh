#!/bin/bash
# A/B of the memory type of the work queues (PT_SHIM_QUEUE_MEM 0 cached / 1 fine-grained / 2 uncached): launch stamps + step time, one box
set -o pipefail
mkdir -p gpurun_out/r04
out=gpurun_out/r04/queue_mem.txt; : > $out
for m in 0 1 2; do
  echo "== PT_SHIM_QUEUE_MEM=$m" >> $out
  PT_SHIM_QUEUE_MEM=$m PT_SHIM_LIB=$(pwd)/oclpathtracer_amd/libptshim_lstamps.so timeout -k 10 200 python tools/launch_stamps.py 1 >> $out 2>&1 || exit 1
  for rep in 1 2; do
    PT_SHIM_QUEUE_MEM=$m timeout -k 10 200 python tools/step_time.py >> $out 2>&1 || exit 1
  done
  PT_SHIM_QUEUE_MEM=$m PT_STAGING_MB=100000 timeout -k 10 200 python tools/step_time.py >> $out 2>&1 || exit 1
done
cat $out
