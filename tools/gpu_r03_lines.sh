#!/bin/bash
# regenerates the round-3 bench lines (the driver's command; config 4; config 3 on one GPU) after the PMC summaries were committed
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/r03g_pytest.log 2>&1; echo "pytest -m gpu rc=$?"; tail -3 gpurun_out/r03g_pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03g_bench.json 2> gpurun_out/r03g_bench.err; echo "bench rc=$?"
timeout -k 10 600 python bench.py --config 4 --steps 1 --warmup 1 > gpurun_out/r03g_soup.json 2> gpurun_out/r03g_soup.err; echo "soup rc=$?"
timeout -k 10 600 python bench.py --config 3 --steps 2 --warmup 1 --no-extra-configs > gpurun_out/r03g_c3.json 2> gpurun_out/r03g_c3.err; echo "config 3 rc=$?"
timeout -k 10 600 python bench.py --config 1 --steps 200 --warmup 5 --no-extra-configs > gpurun_out/r03g_c1.json 2> gpurun_out/r03g_c1.err; echo "config 1 rc=$?"
python3 - <<PY
import json
for f in ("r03g_bench","r03g_soup","r03g_c3","r03g_c1"):
    try:
        d=json.loads([l for l in open("gpurun_out/%s.json"%f) if l.startswith("{")][0])
        r=d.get("roofline") or {}
        print(f, "%.1f Msamples/s  %.2f ms/step  frac %.3f  %s" % (d["value"], d["ms_per_step"], r.get("frac",0), d["config"]["workload"]))
        if "extra" in d:
            for k,v in d["extra"]["configs"].items(): print("   ", k, "%.1f" % v["value"], (v.get("roofline") or {}).get("frac"))
    except Exception as e: print(f, "FAILED", e)
PY
