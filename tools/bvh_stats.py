#!/usr/bin/env python3
"""Diagnostic: LBVH traversal work per ray (needs the -DPT_BVH_STATS=1 build).
usage: PT_SHIM_LIB=.../libptshim_bvhstats.so python tools/bvh_stats.py [ntri W H spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oclpathtracer_amd import adl, scene, shim
from oclpathtracer_amd.render import Renderer
a = sys.argv[1:]
ntri, W, H, spp = (int(x) for x in (a[:4] + ["200000", "256", "256", "4"][len(a[:4]):]))
t, m = scene.make_soup(ntri) if ntri > 36 else scene.load_model()
assert adl.init()
dev = adl.DeviceUtils.allocate()
dev.setOption(shim.PT_OPT_ACCEL, 2)
r = Renderer(dev, t, m, W, H, want_stats=True)
r.render(spp)
out = np.zeros(shim.PT_STAT_WORDS, np.uint64)
r.stats.read(out, shim.PT_STAT_WORDS); dev.waitForCompletion()
samples, rays, nodes, leaves, iters, brays = (int(x) for x in out[:6])
print("%d triangles %dx%d x %d: %d rays; per ray: %.1f nodes entered, %.2f triangles tested, %.1f wave-loop iterations (lane efficiency %.0f%%)"
      % (len(t), W, H, spp, rays, nodes / max(brays, 1), leaves / max(brays, 1), iters / max(brays, 1), 100.0 * nodes / max(iters, 1)))
r.release(); adl.DeviceUtils.deallocate(dev)
