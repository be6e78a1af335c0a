#!/usr/bin/env python3
"""Wall time per render of REPS renders enqueued back to back (no per-kernel event pairs, one wait at the end).
usage: python tools/step_time.py [W H spp depth reps] ; options through the environment: PT_LANES, PT_CHECKPOINT, PT_CHUNK, PT_STAGING_MB"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclpathtracer_amd import adl, scene, shim
from oclpathtracer_amd.render import Renderer
a = sys.argv[1:]
W, H, spp, depth, reps = (int(x) for x in (a[:5] + ["1024", "1024", "256", "16", "40"][len(a[:5]):]))
t, m = scene.load_model()
assert adl.init()
dev = adl.DeviceUtils.allocate()
dev.setOption(shim.PT_OPT_RENDER_LANES, int(os.environ.get("PT_LANES", "2")))
dev.setOption(shim.PT_OPT_CHECKPOINT, int(os.environ.get("PT_CHECKPOINT", "1")))
dev.setOption(shim.PT_OPT_CHUNK_FRAMES, int(os.environ.get("PT_CHUNK", "0")))
dev.reserveStaging(int(os.environ.get("PT_STAGING_MB", "0")) << 20)
r = Renderer(dev, t, m, W, H)
fbs = [r.fb, adl.Buffer(dev, W * H, adl.float4)]   # one scene, two framebuffers, alternating (as bench.py's loop)
for k in range(6):
    r.render(spp, frame_begin=0, max_bounces=depth, fb=fbs[k & 1])
dev.waitForCompletion()
best, enq = 1e9, 0.0
for _ in range(3):
    t0 = time.perf_counter()
    for k in range(reps):
        r.render(spp, frame_begin=0, max_bounces=depth, fb=fbs[k & 1])
    t1 = time.perf_counter()
    dev.waitForCompletion()
    if (time.perf_counter() - t0) / reps < best:
        best, enq = (time.perf_counter() - t0) / reps, (t1 - t0) / reps
print("%dx%d x %d spp depth %d: %.3f ms per render (host: %.3f ms to enqueue it), %.1f Msamples/s (lanes %s checkpoint %s chunk %s staging %s MiB)"
      % (W, H, spp, depth, best * 1e3, enq * 1e3, W * H * spp / best / 1e6, os.environ.get("PT_LANES", "2"), os.environ.get("PT_CHECKPOINT", "1"),
         os.environ.get("PT_CHUNK", "0"), os.environ.get("PT_STAGING_MB", "0")))
fbs[1].release()
r.release()
adl.DeviceUtils.deallocate(dev)
