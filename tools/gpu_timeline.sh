#!/bin/bash
# kernel timeline (rocprofv3 --kernel-trace) of a few pipelined renders; environment settings for tools/step_time.py pass through (PT_FOLD_IN, PT_STAGING_MB ...)
# usage: tools/gpu_timeline.sh <tag>
set -o pipefail
tag=${1:-run}
mkdir -p gpurun_out/r04/timeline
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -- python3 $R/tools/step_time.py 1024 1024 256 16 3 > /tmp/tl_$tag.log 2>&1 || { tail -20 /tmp/tl_$tag.log; exit 1; }
f=$(find /tmp/tl_$tag -name '*kernel_trace.csv' | head -1)
python3 $R/tools/timeline_summary.py $f > $R/gpurun_out/r04/timeline/$tag.txt
grep "ms per render" /tmp/tl_$tag.log
head -${LINES_OUT:-60} $R/gpurun_out/r04/timeline/$tag.txt
