#!/bin/bash
# parity for all variants, then bench each trace-kernel variant
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/var_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -6 gpurun_out/var_pytest.log
[ $rc -ne 0 ] && exit $rc
for v in 1 2 1 2; do
timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --variant $v > gpurun_out/var_bench_$v.log 2>&1
python3 - <<PY
import json
for l in open("gpurun_out/var_bench_$v.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("variant $v  %.1f Msamples/s  trace %.2f ms  fold %.2f ms" % (d["value"], d["roofline"]["avg_launch_ms"], d["kernels"]["pt_fold_kernel_ms_total"]/d["kernels"]["pt_fold_kernel_launches"]))
PY
done
