#!/usr/bin/env python3
"""Diagnostic (needs the PT_LAUNCH_STAMPS=1 build: make -C oclpathtracer_amd/csrc ../libptshim_lstamps.so): where the time of ONE checkpointed
trace launch goes.  Renders configs[2]'s image for three chunks and stamps the launch of chunk `which` (0: no checkpoint to resume; 1: resumes
one) with s_memrealtime (100 MHz): first wave's start, first wave that stopped, last wave's exit, the sum of the waves' lifetimes.
usage: PT_SHIM_LIB=.../libptshim_lstamps.so python tools/launch_stamps.py [which]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oclpathtracer_amd import adl, scene, shim
from oclpathtracer_amd.render import Renderer
which = int(sys.argv[1]) if len(sys.argv) > 1 else 1
t, m = scene.load_model()
assert adl.init()
dev = adl.DeviceUtils.allocate()
r = Renderer(dev, t, m, 1024, 1024, want_stats=True)
WAVES = 8192
r.stats.release()
r.stats = adl.Buffer(dev, shim.PT_STAT_WORDS + 4 * WAVES, np.uint64)     # the diagnostic build writes four words per wave behind the tallies
r.render(48); dev.waitForCompletion()          # warm
for rep in range(3):
    init = np.zeros(shim.PT_STAT_WORDS + 4 * WAVES, np.uint64); init[14] = np.uint64(2**64 - 1); init[13] = which
    r.stats.write(init, len(init)); dev.waitForCompletion()
    r.render(48, frame_begin=0); dev.waitForCompletion()
    st = np.zeros(len(init), np.uint64); r.stats.read(st, len(st)); dev.waitForCompletion()
    w = st[16:].reshape(WAVES, 4).astype(np.int64)
    w = w[w[:, 0] != 0]
    t0 = w[:, 0].min()
    start, lastb, leave = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0, (w[:, 2] - t0) / 100.0
    iters, last_iters, bounds = w[:, 3] & 0xfffff, (w[:, 3] >> 20) & 0xfffff, (w[:, 3] >> 40) & 0xfffff
    full = (int(st[14]) - t0) / 100.0
    pc = lambda a: "min %.0f / 10%% %.0f / median %.0f / 90%% %.0f / max %.0f" % tuple(np.percentile(a, [0, 10, 50, 90, 100]))
    print("chunk %d's launch, %d waves, times in us from the first wave's start: starts %s; stop word complete at %.0f; waves leave %s;"
          % (which, len(w), pc(start), full, pc(leave)))
    print("   per wave %.1f iterations and %.1f boundaries; a wave's LAST generation: %s us, %.1f iterations; slots idle before the last wave left: %.1f %% of the launch"
          % (iters.mean(), bounds.mean(), pc(leave - lastb), last_iters.mean(), 100.0 * (leave.max() - leave).sum() / (leave.max() * len(w))))
r.release(); adl.DeviceUtils.deallocate(dev)
