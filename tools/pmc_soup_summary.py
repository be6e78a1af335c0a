#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes of tools/gpu_pmc_soup.sh (gpurun_out/pmcsoup_<tag>_*) per traced ray.

usage: tools/pmc_soup_summary.py <tag> <spp of the timed render> [out.txt [out.json]]
The profiled command renders the soup twice: `spp` frames (the timed kernel) and 16 frames (the tallied render of bench.py's roofline, the
measurement variant of the kernel: heavier, never timed).  Only the dispatches of the TIMED variant (template argument TALLY = false) are
summed, and divided by the rays of its `spp` frames (samples x rays per sample of the bench line).  (Rounds 2 and 3 summed both variants.)
"""
import collections
import csv
import glob
import json
import os
import sys

tag, spp = sys.argv[1], int(sys.argv[2])
agg = collections.defaultdict(float)
for d in sorted(glob.glob("gpurun_out/pmcsoup_%s_*/" % tag)):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "pt_trace_bvh_kernel" in r["Kernel_Name"] and ", false, " in r["Kernel_Name"]:   # <DET_BOUNDED, TALLY = false, BIGQ>
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
rps = None
for f in glob.glob("gpurun_out/pmcsoup_%s_*.log" % tag):
    for line in open(f):
        if line.startswith("{"):
            j = json.loads(line)
            rps = j["config"]["rays_per_sample"]
            W = H = 1024
if rps is None:
    raise SystemExit("no bench line found in gpurun_out/pmcsoup_%s_*.log" % tag)
rays = W * H * spp * rps
lines = ["10^6-triangle soup, 1024^2 x %d spp, rocprofv3 --pmc passes (tools/gpu_pmc_soup.sh %s %d); sums over the dispatches of the timed variant of"
         % (spp, tag, spp), "pt_trace_bvh_kernel (TALLY = false); %.4g rays" % rays]
for k in sorted(agg):
    lines.append("  %-34s %-14.6g (%.4g per ray)" % (k, agg[k], agg[k] / rays))
g = lambda k: agg.get(k, 0.0)
out = {"rays_in_profiled_launches": rays}
if g("TCC_REQ_sum"):
    out.update(l2_hit_rate=g("TCC_HIT_sum") / g("TCC_REQ_sum"), l2_requests_per_ray=g("TCC_REQ_sum") / rays,
               l2_misses_per_ray=g("TCC_MISS_sum") / rays, TCC_REQ=g("TCC_REQ_sum"), TCC_HIT=g("TCC_HIT_sum"), TCC_MISS=g("TCC_MISS_sum"),
               TCC_EA0_RDREQ=g("TCC_EA0_RDREQ_sum"))
    lines.append("L2 (TCC) hit rate %.1f %%; requests to the L2 per ray %.1f, misses %.1f" % (100 * out["l2_hit_rate"], out["l2_requests_per_ray"], out["l2_misses_per_ray"]))
if g("FETCH_SIZE"):
    out.update(FETCH_SIZE_KiB=g("FETCH_SIZE"), fetched_beyond_l2_bytes_per_ray_as_tallied=g("FETCH_SIZE") * 1024 / rays)
    lines.append("FETCH_SIZE %.0f B per ray as tallied (%.1f B per TCC_EA0_RDREQ)" % (out["fetched_beyond_l2_bytes_per_ray_as_tallied"],
                 g("FETCH_SIZE") * 1024 / max(g("TCC_EA0_RDREQ_sum"), 1)))
if g("SQ_INSTS_VALU"):
    out.update(valu_instructions_per_ray=g("SQ_INSTS_VALU") / rays, lanes_per_valu_instruction=g("SQ_THREAD_CYCLES_VALU") / max(g("SQ_INSTS_VALU"), 1) / 1.0)
    lines.append("VALU: %.1f wave-instructions per ray, SALU %.1f, LDS %.1f, vector loads %.2f; lane-cycles per VALU instruction %.1f"
                 % (g("SQ_INSTS_VALU") / rays, g("SQ_INSTS_SALU") / rays, g("SQ_INSTS_LDS") / rays, g("SQ_INSTS_VMEM_RD") / rays,
                    g("SQ_THREAD_CYCLES_VALU") / max(g("SQ_INSTS_VALU"), 1)))
if g("TCP_TOTAL_CACHE_ACCESSES_sum"):
    out.update(tcp_accesses_per_ray=g("TCP_TOTAL_CACHE_ACCESSES_sum") / rays)
    lines.append("L1 (TCP): %.1f lane accesses per ray, %.1f requests to the L2, mean L1->L2 read latency %.0f cycles"
                 % (out["tcp_accesses_per_ray"], g("TCP_TCC_READ_REQ_sum") / rays, g("TCP_TCC_READ_REQ_LATENCY_sum") / max(g("TCP_TCC_READ_REQ_sum"), 1)))
txt = "\n".join(lines) + "\n"
sys.stdout.write(txt)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(txt)
if len(sys.argv) > 4:
    json.dump(out, open(sys.argv[4], "w"), indent=1)
