#!/usr/bin/env python3
"""Time every rank's share of an N-rank render on ONE GPU (stripes dealt as in the multi-GPU run):
shows the load balance between ranks and the per-rank fixed costs.  usage: python tools/rank_share.py [N W H spp stripe_rows]
A share is timed as the N-rank loop runs it (bench.py): REPS renders enqueued back to back, one wait at the end -- consecutive
renders overlap (the next one's first trace launch beside the last, draining launch of this one); `single` is one render
followed by a wait, the latency of an isolated image."""
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
REPS = 20
single = 0.0
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
    warm.render(spp, frame_begin=0)
dev.waitForCompletion()
warm.release()
order = list(range(N)) + [0]          # ... and rank 0 is measured a second time at the end
for r in order:
    R = Renderer(dev, t, m, W, H, n_ranks=N, rank=r, stripe_rows=SR)
    R.render(spp, frame_begin=0); dev.waitForCompletion()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(REPS):
            R.render(spp, frame_begin=0)
        dev.waitForCompletion(); best = min(best, (time.perf_counter() - t0) / REPS)
    times.append(best * 1e3)
    if r == N - 1 and len(times) == N:
        t0 = time.perf_counter(); R.render(spp, frame_begin=0); dev.waitForCompletion(); single = (time.perf_counter() - t0) * 1e3
    if r == N - 1 and len(times) == N:
        share_k = kernel_ms(lambda: R.render(spp, frame_begin=0))
    R.release()
one = Renderer(dev, t, m, W, H)
one.render(spp, frame_begin=0); dev.waitForCompletion()
t0 = time.perf_counter()
for _ in range(REPS):
    one.render(spp, frame_begin=0)
dev.waitForCompletion(); full = (time.perf_counter() - t0) / REPS * 1e3
full_k = kernel_ms(lambda: one.render(spp, frame_begin=0))
one.release()
print("kernels (mean per launch; a render is several trace launches, the last one draining): full image trace %.3f fold %.3f ms; last rank's share trace %.3f fold %.3f; "
      "one isolated render of that share, with its wait: %.2f ms" % (full_k[0], full_k[1], share_k[0], share_k[1], single))
again = times.pop()
print("stripe %d rows: full image %.2f ms; %d ranks: %s ms (rank 0 measured again last: %.2f); max %.2f -> strong-scaling efficiency bound %.1f%% (render only, no gather)"
      % (SR, full, N, " ".join("%.2f" % x for x in times), again, max(times), 100.0 * full / (N * max(times))))
adl.DeviceUtils.deallocate(dev)
