#!/bin/bash
# full GPU test tier as the driver runs it, plus smoke
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/pytest_gpu_all.log 2>&1
rc=$?; echo "pytest -m gpu rc=$rc"; tail -15 gpurun_out/pytest_gpu_all.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout -k 10 300 ./oclpathtracer_amd/raytrace_test --dim 512 --frames 2000 --scene oclpathtracer_amd/data/cornellbox.bin --out-dir gpurun_out --only RayCast 2>&1 | tail -4
exit $rc
