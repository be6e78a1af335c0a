#!/usr/bin/env python3
"""Diagnostic: s_memtime shares of the sub-phases of pt_shade (needs the PT_STAMPS=2 build:
make -C oclpathtracer_amd/csrc ../libptshim_stamps2.so).  usage: PT_SHIM_LIB=... python tools/stamps2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oclpathtracer_amd import adl, scene, shim
from oclpathtracer_amd.render import Renderer
assert adl.init()
dev = adl.DeviceUtils.allocate()
t, m = scene.load_model()
r = Renderer(dev, t, m, 1024, 1024, want_stats=True)
r.render(64)
out = np.zeros(shim.PT_STAT_WORDS, np.uint64)
r.stats.read(out, shim.PT_STAT_WORDS); dev.waitForCompletion()
sub = [int(x) for x in out[2:8]]
tot = sum(sub)
for n, c in zip(("rng+sincos", "hit record, normal, material", "basis (tv, sv)", "sample dir (2 sqrt, normalize)", "brdf eval", "throughput, next ray, store"), sub):
    print("%-34s %5.1f%%" % (n, 100.0 * c / max(tot, 1)))
r.release(); adl.DeviceUtils.deallocate(dev)
