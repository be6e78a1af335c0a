#!/bin/bash
# pass-1 filter validator over the scene set (needs oclpathtracer_amd/libptshim_validate.so:
# make -C oclpathtracer_amd/csrc ../libptshim_validate.so).  usage: tools/gpu_validate.sh [W H spp]
# Every (ray, triangle) pair traced is checked against the reference predicate: the packed filter, the
# independent-triangle filter AND the primary rays' cached candidate masks (several image geometries).
set -o pipefail
W=${1:-1024}; H=${2:-1024}; SPP=${3:-64}
mkdir -p gpurun_out
export PT_SHIM_LIB=$(pwd)/oclpathtracer_amd/libptshim_validate.so
rc=0
for qf in 0 1; do
  for scene in cornell scaled skewed tiny rolled soup random:1000 random:1003 random:1007; do
    timeout -k 10 300 python tools/validate_filter.py $scene $W $H $SPP $qf 2>&1 | tee -a gpurun_out/filter_validation.txt
    [ ${PIPESTATUS[0]} -ne 0 ] && rc=1
  done
done
for qf in 0 1; do
  for scene in lbvh:2000 lbvh:200000; do   # the big triangles' search inside the LBVH kernel (packed filter / independent triangles)
    timeout -k 10 300 python tools/validate_filter.py $scene 512 512 $SPP $qf 2>&1 | tee -a gpurun_out/filter_validation.txt
    [ ${PIPESTATUS[0]} -ne 0 ] && rc=1
  done
done
for geo in "64 64 256" "200 50 256" "33 97 256" "2048 2048 8" "512 512 512"; do
  for scene in cornell scaled skewed tiny random:1001 random:1005; do
    timeout -k 10 300 python tools/validate_filter.py $scene $geo 0 2>&1 | tee -a gpurun_out/filter_validation.txt
    [ ${PIPESTATUS[0]} -ne 0 ] && rc=1
  done
done
exit $rc
