#!/usr/bin/env python3
"""Prices an XCD-affine region scheme for the LBVH search on the CPU (tools/xcd_pricing.c has the model and the question).
usage: python tools/xcd_pricing.py [triangles rays inflight]   (defaults: BASELINE configs[4]'s 10^6-triangle soup, 400 000 rays, 40 960 in flight per XCD)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oclpathtracer_amd import scene
a = sys.argv[1:]
ntri, nrays, inflight = (int(x) for x in (a[:3] + ["1000000", "400000", "40960"][len(a[:3]):]))
tris, _ = scene.make_soup(ntri)
raw = np.concatenate([tris["p1"][:, :3], tris["p2"][:, :3], tris["p3"][:, :3]], axis=1).astype(np.float32)
path = "/tmp/xcd_soup_%d.bin" % ntri
with open(path, "wb") as f:
    f.write(np.int32(ntri).tobytes()); f.write(raw.tobytes())
exe = "/tmp/xcd_pricing"
subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tools", "xcd_pricing.c"), "-lm"])
out = subprocess.check_output([exe, path, str(nrays), str(inflight)]).decode()
vals = {l.split()[0]: float(l.split()[1]) for l in out.splitlines() if l.split() and l.split()[0] in ("BASE_MISSES", "REGION_MISSES", "REGION_HANDOVERS", "ORIGIN_MISSES")}
print("\n".join(l for l in out.splitlines() if not l.startswith(("BASE_", "REGION_", "ORIGIN_"))))
MEASURED = 23.6   # L2 misses per ray of pt_trace_bvh_kernel on this scene, profiles/r03/pmc_traffic_soup.json
ratio = vals["REGION_MISSES"] / vals["BASE_MISSES"]
priced = MEASURED * ratio
hand = vals["REGION_HANDOVERS"]
print("\npriced against the measurement: %.1f misses per ray x %.3f (model's regions / baseline) = %.1f misses + %.2f hand-overs of 64 bytes = %.1f per ray"
      % (MEASURED, ratio, priced, hand, priced + hand))
print("threshold for building it (VERDICT r03): <= 14 per ray  ->  %s" % ("BUILD" if priced + hand <= 14.0 else "do not build: the scheme cannot reach it"))
o = MEASURED * vals["ORIGIN_MISSES"] / vals["BASE_MISSES"]
print("origin-affine variant (rays binned by the region they start in, between bounces; no hand-over inside a search): %.1f x %.3f = %.1f misses + 1 hand-over = %.1f per ray"
      % (MEASURED, vals["ORIGIN_MISSES"] / vals["BASE_MISSES"], o, o + 1.0))
