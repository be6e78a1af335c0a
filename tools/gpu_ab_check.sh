#!/bin/bash
# a kernel change: the parity tests, then the same-box step-time A/B against libptshim_old.so (tools/gpu_ab_step.sh)
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_streaming.py -m gpu -x -q > gpurun_out/r04/ab_check_pytest.log 2>&1
rc=$?; tail -2 gpurun_out/r04/ab_check_pytest.log
[ $rc -ne 0 ] && { tail -40 gpurun_out/r04/ab_check_pytest.log; exit $rc; }
tools/gpu_ab_step.sh
