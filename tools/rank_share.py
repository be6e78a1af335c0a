#!/usr/bin/env python3
"""Time every rank's share of an N-rank render on ONE GPU (stripes dealt as in the multi-GPU run):
shows the load balance between ranks and the per-rank fixed costs.  usage: python tools/rank_share.py [N W H spp stripe_rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np
from oclpathtracer_amd import adl, scene, shim
from oclpathtracer_amd.render import Renderer
a = sys.argv[1:]
N, W, H, spp, SR = (int(x) for x in (a[:5] + ["8", "1024", "1024", "256", "16"][len(a[:5]):]))
t, m = scene.load_model()
assert adl.init()
dev = adl.DeviceUtils.allocate()
times = []
lib = shim.load()
def kernel_ms(fn):
    shim.check(lib.pt_profile_enable(dev._h, 1)); shim.check(lib.pt_profile_reset(dev._h))
    fn(); dev.waitForCompletion()
    out = []
    for kind in (shim.PT_PROF_TRACE, shim.PT_PROF_FOLD):
        ms, n = ctypes.c_double(), ctypes.c_uint64()
        shim.check(lib.pt_profile_query(dev._h, kind, ctypes.byref(ms), ctypes.byref(n)))
        out.append(ms.value / max(n.value, 1))
    shim.check(lib.pt_profile_enable(dev._h, 0))
    return out
# warm the device first with throw-away renders of the LAST rank's share (clocks, code and scene caches, the radiance
# staging allocation): round 2 measured rank 0 straight after start-up and read it 5-8 % high "on every run"
warm = Renderer(dev, t, m, W, H, n_ranks=N, rank=N - 1, stripe_rows=SR)
for _ in range(20):
    warm.render(spp)
dev.waitForCompletion()
warm.release()
order = list(range(N)) + [0]          # ... and rank 0 is measured a second time at the end
for r in order:
    R = Renderer(dev, t, m, W, H, n_ranks=N, rank=r, stripe_rows=SR)
    R.render(spp); dev.waitForCompletion()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); R.render(spp); dev.waitForCompletion(); best = min(best, time.perf_counter() - t0)
    times.append(best * 1e3)
    if r == N - 1 and len(times) == N:
        share_k = kernel_ms(lambda: R.render(spp))
    R.release()
one = Renderer(dev, t, m, W, H)
one.render(spp); dev.waitForCompletion()
t0 = time.perf_counter(); one.render(spp); dev.waitForCompletion(); full = (time.perf_counter() - t0) * 1e3
full_k = kernel_ms(lambda: one.render(spp))
one.release()
print("kernels: full image trace %.3f fold %.3f ms; last rank's share trace %.3f (x%d = %.2f) fold %.3f (x%d = %.2f)"
      % (full_k[0], full_k[1], share_k[0], N, share_k[0] * N, share_k[1], N, share_k[1] * N))
again = times.pop()
print("stripe %d rows: full image %.2f ms; %d ranks: %s ms (rank 0 measured again last: %.2f); max %.2f -> strong-scaling efficiency bound %.1f%% (render only, no gather)"
      % (SR, full, N, " ".join("%.2f" % x for x in times), again, max(times), 100.0 * full / (N * max(times))))
adl.DeviceUtils.deallocate(dev)
