#!/bin/bash
# round 4: GPU tests + the phase shares of the trace kernel (PT_STAMPS build)
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 1100 python -m pytest tests/ -q -m gpu > gpurun_out/r04/pytest_check.log 2>&1
rc=$?; echo "pytest -m gpu rc=$rc"; tail -6 gpurun_out/r04/pytest_check.log
PT_SHIM_LIB=$(pwd)/oclpathtracer_amd/libptshim_stamps.so timeout -k 10 300 python tools/stamps.py 1024 1024 64 16 | tee gpurun_out/r04/stamps_r04.txt
exit $rc
