#!/bin/bash
# round-4 closing record on the final build: GPU test tier, smoke, the driver's bench command, its rocprofv3 kernel stats, the configs[4] line,
# the one-GPU bound of the N-rank split.  Everything under gpurun_out/r04fin/.
set -o pipefail
REPO=$(pwd)
O=gpurun_out/r04fin
mkdir -p $O
timeout -k 10 900 python -m pytest tests/ -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest -m gpu rc=$rc"; tail -2 $O/pytest.log
[ $rc -ne 0 ] && { tail -40 $O/pytest.log; exit $rc; }
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; python3 - <<PY
import json
d=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][0])
r=d["roofline"]
print("value %.1f ms/step %.2f trace avg %.3f ms x %d frac %.4f excl %.4f cpu %.2f x%.0f alloc %.1f ms ws %.0f MB" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["launches"], r["frac"], r["frac_exclusive"], d["cpu_baseline"]["value"], d["gpu_over_cpu"], d["alloc_ms"], d["config"]["workspace_bytes"]/1e6))
for k,v in d["extra"]["configs"].items(): print(k, "%.1f Msamples/s %.2f ms" % (v["value"], v["ms_per_step"]) if "value" in v else v)
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof -o bench -- python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > $REPO/$O/prof.log 2>&1
echo "rocprof rc=$?"
cd $REPO
for f in $(find $O/prof -name "*kernel_stats.csv"); do cp $f $O/bench_kernel_stats.csv; cut -c1-200 $f | head -8; done
timeout -k 10 600 python bench.py --config 4 --steps 2 --warmup 1 > $O/soup.json 2> $O/soup.err
echo "soup bench rc=$?"; grep -h '^{' $O/soup.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('soup value %.1f Msamples/s  launch %.1f ms  frac %.3f gather %.3f (valu %.3f)' % (d['value'], r['avg_launch_ms'], r['frac'], r.get('frac_of_gather_ceiling', 0), d['roofline_valu']['frac']))"
tools/gpu_r04_share.sh > $O/rank_share.log 2>&1; cp gpurun_out/r04/rank_share.txt $O/ 2>/dev/null; tail -6 $O/rank_share.log
