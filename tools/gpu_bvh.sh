#!/bin/bash
# LBVH iteration: the BVH parity tests, then the soup bench line (with tallies) for each library given
# usage: tools/gpu_bvh.sh lib1.so [lib2.so ...]
set -o pipefail
mkdir -p gpurun_out
first=1
for lib in "$@"; do
  export PT_SHIM_LIB=$(pwd)/oclpathtracer_amd/$lib
  tag=${lib%.so}
  if [ $first -eq 1 ]; then
    first=0
    timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${BVH_TESTS:-bvh or soup or lbvh or random_quad or ties or golden or corner}" > gpurun_out/bvh_${tag}_pytest.log 2>&1
    rc=$?; echo "$lib pytest rc=$rc $(tail -1 gpurun_out/bvh_${tag}_pytest.log)"
    # (a timing of a kernel that failed parity must not reach profiles/ or a DESIGN table by way of a missed log line)
    [ $rc -ne 0 ] && { tail -30 gpurun_out/bvh_${tag}_pytest.log; echo "PARITY FAILED for $lib: no timing taken"; exit $rc; }
  fi
  timeout -k 10 300 python bench.py --soup 1000000 --spp 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > gpurun_out/bvh_${tag}_bench.log 2>&1
  python3 - <<PY
import json
for l in open("gpurun_out/bvh_${tag}_bench.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d.get("roofline") or {}
        print("  %-22s %.1f Msamples/s  nodes/ray %.1f  tris/ray %.2f  lane occupancy node %.2f tri %.2f  rays/sample %.2f" % ("$lib", d["value"], r.get("nodes_per_ray", 0), r.get("tris_per_ray", 0), r.get("node_phase_lane_occupancy", 0), r.get("tri_phase_lane_occupancy", 0), d["config"]["rays_per_sample"]))
PY
done
