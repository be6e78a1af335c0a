#!/bin/bash
# registers, spills, scratch and LDS of every kernel in csrc/pt_kernels.hip (from the code object's metadata)
cd "$(dirname "$0")/../oclpathtracer_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize --cuda-device-only -S -o /tmp/pt_kernels.s pt_kernels.hip "$@" 2>/dev/null || exit 1
python3 - <<'PY'
import re
t = open("/tmp/pt_kernels.s").read()
md = t[t.index(".amdgpu_metadata"):]
print("%-78s %5s %5s %7s %7s %6s %8s" % ("kernel", "VGPR", "SGPR", "v-spill", "s-spill", "LDS", "scratch"))
for blk in md.split("  - .agpr_count")[1:]:
    g = lambda k: re.search(r"\.%s:\s*(\S+)" % k, blk).group(1)
    print("%-78s %5s %5s %7s %7s %6s %8s" % (g("name")[:78], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
PY
