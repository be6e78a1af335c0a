#!/bin/bash
# one optimisation iteration on the GPU box: parity first, then the bench line
set -o pipefail
TAG=${1:-iter}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/${TAG}_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs > gpurun_out/${TAG}_bench.log 2>&1
echo "bench rc=$?"
python3 - <<PY
import json
for l in open("gpurun_out/${TAG}_bench.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("VALUE %.1f Msamples/s  ms/step %.2f  trace %.2f ms  fold %.2f ms  frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["kernels"]["pt_fold_kernel_ms_total"]/d["kernels"]["pt_fold_kernel_launches"], d["roofline"]["frac"]))
PY
