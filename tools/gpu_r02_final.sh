#!/bin/bash
# round-2 record run: whole GPU test tier, smoke, the driver's bench command, its rocprofv3 kernel stats, rank shares
set -o pipefail
mkdir -p gpurun_out
REPO=$(pwd)
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu --durations=8 > gpurun_out/r02f_pytest.log 2>&1
rc=$?; echo "pytest -m gpu rc=$rc"; tail -14 gpurun_out/r02f_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02f_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r02f_smoke.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02f_bench.json 2> gpurun_out/r02f_bench.err
echo "bench rc=$?"; python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r02f_bench.json") if l.startswith("{")][0])
print("value %.1f ms/step %.2f trace %.2f frac %.3f cpu %.2f x%.0f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["gpu_over_cpu"]))
for k,v in d["extra"]["configs"].items(): print(k, "%.1f Msamples/s" % v["value"], v.get("roofline",{}).get("nodes_per_ray"))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_r02f -o bench -- python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > $REPO/gpurun_out/prof_r02f.log 2>&1
echo "rocprof rc=$?"
cd $REPO
for f in $(find gpurun_out/prof_r02f -name "*kernel_stats.csv"); do head -8 $f; done
timeout -k 10 300 python tools/rank_share.py 8 > gpurun_out/r02f_rank_share.txt 2>&1; timeout -k 10 300 python tools/rank_share.py 4 >> gpurun_out/r02f_rank_share.txt 2>&1; timeout -k 10 300 python tools/rank_share.py 2 >> gpurun_out/r02f_rank_share.txt 2>&1; cat gpurun_out/r02f_rank_share.txt
