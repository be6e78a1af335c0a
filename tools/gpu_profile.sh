#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench command; summaries land in gpurun_out/prof_<tag>/
set -o pipefail
TAG=${1:-r01}
STEPS=${2:-3}
REPO=$(pwd)
mkdir -p gpurun_out
echo "nproc=$(nproc) affinity=$(python3 -c 'import os;print(len(os.sched_getaffinity(0)))') cpu.max=$(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o bench -- python3 $REPO/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-extra-configs > $REPO/gpurun_out/prof_$TAG.log 2>&1
echo "rocprof rc=$?"
cd $REPO
tail -3 gpurun_out/prof_$TAG.log
find gpurun_out/prof_$TAG -name "*stats*" | head
for f in $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv"); do head -12 $f; done
