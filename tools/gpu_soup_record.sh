#!/bin/bash
# the 10^6-triangle soup at full size (configs[4] scene, 1024^2 x 256 spp): bench line + rocprofv3 kernel stats
set -o pipefail
mkdir -p gpurun_out
REPO=$(pwd)
timeout -k 10 600 python bench.py --soup 1000000 --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs > gpurun_out/soup_bench.json 2> gpurun_out/soup_bench.err
echo "bench rc=$?"; grep -h '^{' gpurun_out/soup_bench.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.1f Msamples/s  launch %.1f ms  nodes/ray %.1f tris/ray %.2f node occ %.2f  alg %.0f GB/s  traffic/launch %.3g' % (d['value'], r['avg_launch_ms'], r['nodes_per_ray'], r['tris_per_ray'], r['node_phase_lane_occupancy'], r['achieved'], r['traffic'] or 0))"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_soup -o soup -- python3 $REPO/bench.py --soup 1000000 --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs > $REPO/gpurun_out/prof_soup.log 2>&1
echo "rocprof rc=$?"
cd $REPO
for f in $(find gpurun_out/prof_soup -name "*kernel_stats.csv"); do cut -c1-160 $f | head -14; done
