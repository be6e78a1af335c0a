#!/bin/bash
# kernel timeline (rocprofv3 --kernel-trace) of a few pipelined renders for given (chains, ring slots, staging MiB) settings
set -o pipefail
mkdir -p gpurun_out/r04/timeline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in ${CFGS:-"1 2 360" "1 3 360" "2 5 360"}; do
  set -- $cfg
  tag=c$1_r$2_s$3
  PT_SHIM_CHAINS=$1 PT_SHIM_RING_SLOTS=$2 PT_STAGING_MB=$3 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -- python3 $R/tools/step_time.py 1024 1024 256 16 3 > /tmp/tl_$tag.log 2>&1 || { tail -20 /tmp/tl_$tag.log; exit 1; }
  f=$(find /tmp/tl_$tag -name '*kernel_trace.csv' | head -1)
  python3 $R/tools/timeline_summary.py $f > $R/gpurun_out/r04/timeline/$tag.txt
  tail -3 /tmp/tl_$tag.log
  head -70 $R/gpurun_out/r04/timeline/$tag.txt
done
