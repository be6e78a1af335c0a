#!/bin/bash
# round 4 iteration check: GPU tests, step time of configs[2] at a few batch sizes, the configs[4] scene line
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/ -q -m gpu > gpurun_out/r04/pytest_iter.log 2>&1
rc=$?; echo "pytest -m gpu rc=$rc"; tail -4 gpurun_out/r04/pytest_iter.log
for b in ${BATCHES}; do PT_SHIM_BATCH=$b timeout -k 10 120 python tools/step_time.py 1024 1024 256 16 30 || exit 1; done
timeout -k 10 120 python tools/step_time.py 1024 1024 256 16 30
PT_STAGING_MB=6400 PT_LANES=1 PT_CHECKPOINT=0 timeout -k 10 120 python tools/step_time.py 1024 1024 256 16 30   # (the round-3 shape on this box: one launch per render)
timeout -k 10 300 python bench.py --config 4 --steps 2 --warmup 1 > gpurun_out/r04/soup_iter.json 2> gpurun_out/r04/soup_iter.err
grep -h '^{' gpurun_out/r04/soup_iter.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('soup value %.1f Msamples/s  ms/step %.1f launch %.1f ms' % (d['value'], d['ms_per_step'], r['avg_launch_ms']))"
exit $rc
