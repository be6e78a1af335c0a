#!/usr/bin/env python3
"""Time every rank's share of an N-rank render on ONE GPU (stripes dealt as in the multi-GPU run):
shows the load balance between ranks and the per-rank fixed costs.  usage: python tools/rank_share.py [N W H spp stripe_rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oclpathtracer_amd import adl, scene
from oclpathtracer_amd.render import Renderer
a = sys.argv[1:]
N, W, H, spp, SR = (int(x) for x in (a[:5] + ["8", "1024", "1024", "256", "16"][len(a[:5]):]))
t, m = scene.load_model()
assert adl.init()
dev = adl.DeviceUtils.allocate()
times = []
for r in range(N):
    R = Renderer(dev, t, m, W, H, n_ranks=N, rank=r, stripe_rows=SR)
    R.render(spp); dev.waitForCompletion()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); R.render(spp); dev.waitForCompletion(); best = min(best, time.perf_counter() - t0)
    times.append(best * 1e3)
    R.release()
one = Renderer(dev, t, m, W, H)
one.render(spp); dev.waitForCompletion()
t0 = time.perf_counter(); one.render(spp); dev.waitForCompletion(); full = (time.perf_counter() - t0) * 1e3
one.release()
print("stripe %d rows: full image %.2f ms; %d ranks: %s ms; max %.2f -> strong-scaling efficiency bound %.1f%% (render only, no gather)"
      % (SR, full, N, " ".join("%.2f" % x for x in times), max(times), 100.0 * full / (N * max(times))))
adl.DeviceUtils.deallocate(dev)
