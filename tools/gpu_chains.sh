#!/bin/bash
# the two-chain streaming renderer: streaming tests, then step time over (chains, ring slots, staging) on one box
set -o pipefail
mkdir -p gpurun_out/r04
out=gpurun_out/r04/chains.txt; : > $out
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout -k 10 900 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_lbvh_robust.py -m gpu -x -q > gpurun_out/r04/chains_pytest.log 2>&1
  rc=$?; tail -5 gpurun_out/r04/chains_pytest.log
  [ $rc -ne 0 ] && { tail -40 gpurun_out/r04/chains_pytest.log; exit $rc; }
fi
for cfg in ${CFGS:-"1 2 360" "1 3 360" "2 4 360" "2 5 360" "2 6 360" "2 5 450" "2 5 100000"}; do
  set -- $cfg
  echo "== chains $1, ring slots $2, staging $3 MiB" >> $out
  PT_SHIM_CHAINS=$1 PT_SHIM_RING_SLOTS=$2 PT_STAGING_MB=$3 timeout -k 10 200 python tools/step_time.py >> $out 2>&1 || { tail -5 $out; exit 1; }
done
cat $out
