#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes of tools/gpu_pmc.sh (gpurun_out/pmc_<tag>_*) for BASELINE configs[2]: per-sample instruction counts of
the trace and fold kernels (all their dispatches of one render summed) and the bytes beyond the L2.
usage: tools/pmc_cornell_summary.py <tag> <spp> [out.txt [out.json]]   (the json is what bench.py's `traffic` fields read)"""
import collections, csv, glob, json, subprocess, sys
tag, spp = sys.argv[1], int(sys.argv[2])
agg = collections.defaultdict(float); calls = collections.Counter()
for f in glob.glob("gpurun_out/pmc_%s_*/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k = "trace" if "pt_trace" in r["Kernel_Name"] else "fold" if "pt_fold_kernel" in r["Kernel_Name"] else None
        if k:
            agg[(k, r["Counter_Name"])] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
n = 1024 * 1024 * spp
g = lambda k, c: agg.get((k, c), 0.0)
try:
    head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip()
except Exception:
    head = None
lines = ["BASELINE configs[2] (cornellbox 1024^2 x %d spp, depth 16), rocprofv3 --pmc passes (tools/gpu_pmc.sh %s %d: bench.py --steps 1 --warmup 0, counters only);"
         % (spp, tag, spp), "sums over all dispatches of the render (%d trace launches, %d folds); %d samples" % (calls[("trace", "SQ_INSTS_VALU")], calls[("fold", "SQ_INSTS_VALU")], n)]
for k in ("trace", "fold"):
    v = g(k, "SQ_INSTS_VALU")
    lines.append("pt_%s_kernel: SQ_INSTS_VALU %.4g = %.2f per sample; SALU %.4g (%.2f of VALU); SMEM %.3g; lanes active per VALU instruction %.1f of 64; VMEM rd %.3g wr %.3g; waves %d"
                 % (k, v, v / n, g(k, "SQ_INSTS_SALU"), g(k, "SQ_INSTS_SALU") / max(v, 1), g(k, "SQ_INSTS_SMEM"), g(k, "SQ_THREAD_CYCLES_VALU") / max(v, 1),
                    g(k, "SQ_INSTS_VMEM_RD"), g(k, "SQ_INSTS_VMEM_WR"), int(g(k, "SQ_WAVES"))))
    lines.append("    SQ_BUSY_CYCLES %.4g  SQ_WAVE_CYCLES %.4g  SQ_WAIT_INST_ANY %.4g  GRBM_GUI_ACTIVE %.4g (/8 XCDs = %.4g)"
                 % (g(k, "SQ_BUSY_CYCLES"), g(k, "SQ_WAVE_CYCLES"), g(k, "SQ_WAIT_INST_ANY"), g(k, "GRBM_GUI_ACTIVE"), g(k, "GRBM_GUI_ACTIVE") / 8))
    # (MI355X_MICROARCH.md: the SQ counters count quad-cycles; a wave64 vector instruction holds the issue for one quad and the SIMD sustains two per quad)
    quads = g(k, "GRBM_GUI_ACTIVE") / 8 / 4 * 1024
    if quads > 0 and g(k, "SQ_ACTIVE_INST_VALU") > 0:
        lines.append("    vector-ALU issue: SQ_ACTIVE_INST_VALU %.4g quad-cycles over 1 024 SIMDs x %.4g quad-cycles of GRBM_GUI_ACTIVE / 8 = %.2f per SIMD and quad-cycle, "
                     "%.0f %% of the two a SIMD sustains (v_fma_f32, wave64: 2 cycles); per wave: issuing %.0f %%, issue-stalled %.0f %%, parked in a wait %.0f %% of its cycles"
                     % (g(k, "SQ_ACTIVE_INST_VALU"), quads / 1024, g(k, "SQ_ACTIVE_INST_VALU") / quads, 50.0 * g(k, "SQ_ACTIVE_INST_VALU") / quads,
                        100.0 * g(k, "SQ_ACTIVE_INST_ANY") / max(g(k, "SQ_WAVE_CYCLES"), 1), 100.0 * g(k, "SQ_WAIT_INST_ANY") / max(g(k, "SQ_WAVE_CYCLES"), 1),
                        100.0 * g(k, "SQ_WAIT_ANY") / max(g(k, "SQ_WAVE_CYCLES"), 1)))
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/gpu_pmc.sh %s %d: bench.py --spp %d --steps 1 (the full configs[2] render: "
                 "every trace and fold launch of it summed), units KiB; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE under-reports wide "
                 "streaming reads by 2x, MI355X_MICROARCH.md HBM).  These are bytes beyond the L2; the Infinity Cache lies behind it" % (tag, spp, spp),
       "head": head, "samples_in_profiled_launch": n}
for k in ("trace", "fold"):
    f, w = g(k, "FETCH_SIZE"), g(k, "WRITE_SIZE")
    out["pt_%s_kernel" % k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_sample": (2 * f + w) * 1024 / n, "launches": calls[(k, "FETCH_SIZE")]}
    lines.append("pt_%s_kernel: FETCH_SIZE %.4g KiB (x2 = %.2f B per sample), WRITE_SIZE %.4g KiB (%.2f B per sample): %.2f B per sample beyond the L2"
                 % (k, f, 2 * f * 1024 / n, w, w * 1024 / n, (2 * f + w) * 1024 / n))
txt = "\n".join(lines) + "\n"
sys.stdout.write(txt)
if len(sys.argv) > 3: open(sys.argv[3], "w").write(txt)
if len(sys.argv) > 4: json.dump(out, open(sys.argv[4], "w"), indent=1)
