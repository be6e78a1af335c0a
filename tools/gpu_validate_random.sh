#!/bin/bash
# pass-1 filter validator on the random quad scenes of the fuzz test (strongest filter)
set -o pipefail
mkdir -p gpurun_out
export PT_SHIM_LIB=$(pwd)/oclpathtracer_amd/libptshim_validate.so
rc=0
for seed in $(seq 1000 1023); do
  timeout -k 10 300 python tools/validate_filter.py random:$seed 512 512 64 0 2>&1 | tee -a gpurun_out/filter_validation_random.txt
  [ ${PIPESTATUS[0]} -ne 0 ] && rc=1
done
exit $rc
