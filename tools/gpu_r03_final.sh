#!/bin/bash
# round-3 record run: the driver's bench command, its rocprofv3 kernel stats, the configs[4] line + stats, PMC passes
set -o pipefail
mkdir -p gpurun_out
REPO=$(pwd)
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03f_bench.json 2> gpurun_out/r03f_bench.err
echo "bench rc=$?"; python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03f_bench.json") if l.startswith("{")][0])
print("value %.1f ms/step %.2f trace %.2f frac %.3f cpu %.2f x%.0f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["gpu_over_cpu"]))
for k,v in d["extra"]["configs"].items(): print(k, "%.1f Msamples/s" % v["value"], (v.get("roofline") or {}).get("frac"))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_r03f -o bench -- python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > $REPO/gpurun_out/prof_r03f.log 2>&1
echo "rocprof rc=$?"
cd $REPO
for f in $(find gpurun_out/prof_r03f -name "*kernel_stats.csv"); do head -8 $f; done
timeout -k 10 600 python bench.py --config 4 --steps 1 --warmup 1 > gpurun_out/r03f_soup.json 2> gpurun_out/r03f_soup.err
echo "soup bench rc=$?"; grep -h '^{' gpurun_out/r03f_soup.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('soup value %.1f Msamples/s  launch %.1f ms  frac %.3f (valu %.3f)  %s' % (d['value'], r['avg_launch_ms'], r['frac'], d['roofline_valu']['frac'], d['config']['workload']))"
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_r03f_soup -o soup -- python3 $REPO/bench.py --config 4 --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs > $REPO/gpurun_out/prof_r03f_soup.log 2>&1
echo "rocprof soup rc=$?"
cd $REPO
for f in $(find gpurun_out/prof_r03f_soup -name "*kernel_stats.csv"); do cut -c1-170 $f | head -8; done
tools/gpu_pmc_soup.sh r03f 8 > gpurun_out/r03f_pmcsoup.txt 2>&1; tail -3 gpurun_out/r03f_pmcsoup.txt
tools/gpu_pmc_traffic.sh r03f 256
tools/gpu_pmc.sh r03f 256 > gpurun_out/r03f_pmc.txt 2>&1; grep -c "sum=" gpurun_out/r03f_pmc.txt
