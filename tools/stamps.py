#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime shares of the trace kernel (needs the PT_STAMPS=1 build).
usage: PT_SHIM_LIB=.../libptshim_stamps.so python tools/stamps.py [W H spp depth]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oclpathtracer_amd import adl, scene, shim
from oclpathtracer_amd.render import Renderer
args = sys.argv[1:]
variant = 1
W, H, spp, depth = (int(x) for x in (args[:4] + ["1024", "1024", "64", "16"][len(args[:4]):]))
quad_filter = int(args[4]) if len(args) > 4 else 0
assert adl.init()
dev = adl.DeviceUtils.allocate()
t, m = scene.load_model()
dev.setOption(shim.PT_OPT_QUAD_FILTER, quad_filter)
r = Renderer(dev, t, m, W, H, want_stats=True)
r.render(spp, max_bounces=depth)
out = np.zeros(shim.PT_STAT_WORDS, np.uint64)
r.stats.read(out, shim.PT_STAT_WORDS); dev.waitForCompletion()
samples, rays, c_regen, c_loop, c_shade, iters, c_sort, c_full = (int(x) for x in out[:8])
c_p1 = 0
if variant == 1:  # stats[6] = pass-1 ticks, a part of c_loop
    c_p1, c_sort = c_sort, 0
tot = c_regen + c_loop + c_shade + c_sort
print("variant", variant, " stats[7] per wave-iteration (v1: pass-2 steps, v2: triangles fully tested): %.1f" % (c_full / max(iters, 1)))
print("samples %d rays %d wave-iterations %d  lanes busy per iteration %.1f/64" % (samples, rays, iters, rays / max(iters, 1)))
for n, c in (("regen", c_regen), ("sort", c_sort), ("loop", c_loop), ("shade", c_shade)):
    print("%-6s %5.1f%%   %.0f ticks per wave-iteration" % (n, 100.0 * c / max(tot, 1), c / max(iters, 1)))
if variant == 1:
    print("       of loop: pass 1 %.1f%% of the total (%.0f ticks per wave-iteration), pass 2 the rest" % (100.0 * c_p1 / max(tot, 1), c_p1 / max(iters, 1)))
r.release(); adl.DeviceUtils.deallocate(dev)
