#!/bin/bash
# first GPU session: micro-experiments, smoke, parity tests, first bench line
set -o pipefail
mkdir -p gpurun_out
echo "== ubench" ; timeout -k 10 240 ./tools/ubench > gpurun_out/ubench.log 2>&1 ; echo "ubench rc=$?" ; tail -20 gpurun_out/ubench.log
echo "== smoke" ; timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -5 gpurun_out/smoke.log
echo "== pytest" ; timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -30 gpurun_out/pytest_gpu.log
echo "== bench" ; timeout -k 10 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench1.log 2>&1 ; echo "bench rc=$?" ; tail -5 gpurun_out/bench1.log
