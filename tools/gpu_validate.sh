#!/bin/bash
# pass-1 filter validator over the scene set (needs oclpathtracer_amd/libptshim_validate.so:
# make -C oclpathtracer_amd/csrc ../libptshim_validate.so).  usage: tools/gpu_validate.sh [W H spp]
set -o pipefail
W=${1:-1024}; H=${2:-1024}; SPP=${3:-64}
mkdir -p gpurun_out
export PT_SHIM_LIB=$(pwd)/oclpathtracer_amd/libptshim_validate.so
rc=0
for qf in 0 3; do
  for scene in cornell scaled skewed tiny rolled soup; do
    timeout -k 10 300 python tools/validate_filter.py $scene $W $H $SPP $qf 2>&1 | tee -a gpurun_out/filter_validation.txt
    [ ${PIPESTATUS[0]} -ne 0 ] && rc=1
  done
done
exit $rc
