import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from oclpathtracer_amd import adl, scene, shim
from oclpathtracer_amd.render import Renderer
t, m = scene.load_model()
assert adl.init()
dev = adl.DeviceUtils.allocate()
for W, spp in ((1024, 64), (256, 64)):
    r = Renderer(dev, t, m, W, W, want_stats=True)
    r.render(spp); dev.waitForCompletion()
    st = r.read_stats_raw()
    print(W, spp, "samples", st[0], "rays", st[1], "carried", st[7], "words", st)
    r.release()
adl.DeviceUtils.deallocate(dev)
