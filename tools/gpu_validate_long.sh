#!/bin/bash
# long run of the pass-1 filter validator (strongest filter only): 1024 x 1024 x 1024 spp per scene
set -o pipefail
mkdir -p gpurun_out
export PT_SHIM_LIB=$(pwd)/oclpathtracer_amd/libptshim_validate.so
rc=0
for scene in cornell scaled skewed tiny; do
  timeout -k 10 600 python tools/validate_filter.py $scene 1024 1024 1024 0 2>&1 | tee -a gpurun_out/filter_validation_long.txt
  [ ${PIPESTATUS[0]} -ne 0 ] && rc=1
done
exit $rc
