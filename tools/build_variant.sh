#!/bin/bash
# builds oclpathtracer_amd/libptshim_<name>.so from the current sources with extra -D flags (A/B runs: tools/gpu_ab.sh, tools/gpu_bvh.sh)
# usage: tools/build_variant.sh <name> [-DFOO=1 ...]     (SRC_DIR=<dir> builds another checkout's csrc instead)
set -e
name=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${SRC_DIR:-$ROOT/oclpathtracer_amd/csrc}
cd "$SRC"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -Wno-unused-value -Wno-unused-result "$@" -shared -o "$ROOT/oclpathtracer_amd/libptshim_$name.so" pt_kernels.hip pt_shim.hip pt_bvh.hip
echo "built libptshim_$name.so ($*)"
