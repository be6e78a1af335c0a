#!/usr/bin/env python3
"""Diagnostic: empirical check of the pass-1 conservative filter (needs the PT_VALIDATE_FILTER=1
build).  For every (ray, triangle) pair actually traced, the reference's predicate of
GenerateColors.cl:100,109 (literal form, IEEE division) is evaluated beside the filter; a pair the
reference keeps but the filter dropped is a VIOLATION and must never occur.
usage: PT_SHIM_LIB=.../libptshim_validate.so python tools/validate_filter.py [scene] [W H spp] [quad_filter]
(make -C oclpathtracer_amd/csrc ../libptshim_validate.so; quad_filter = PT_OPT_QUAD_FILTER, 0 = auto)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oclpathtracer_amd import adl, scene, shim
from oclpathtracer_amd.render import Renderer

kind = sys.argv[1] if len(sys.argv) > 1 else "cornell"
W, H, spp = (int(x) for x in (sys.argv[2:5] + ["512", "512", "64"][len(sys.argv[2:5]):]))
quad_filter = int(sys.argv[5]) if len(sys.argv) > 5 else 0
t, m = scene.load_model()
if kind == "rolled":          # pairs broken: per-triangle filter
    t = np.roll(t, 1)
elif kind == "soup":
    t, m = scene.make_soup(200)
elif kind.startswith("lbvh:"):    # a soup large enough for the LBVH: what is validated is the filter of the brute-force search over
    t, m = scene.make_soup(int(kind.split(":")[1]))   # the BIG triangles kept out of the hierarchy (the Cornell box's 36: quads -> packed filter)
elif kind == "scaled":
    t = t.copy()
    for f in ("p1", "p2", "p3"):
        t[f][:, :3] = t[f][:, :3] * np.float32(37.5) + np.array([3.0, -80.0, 11.0], np.float32)
elif kind == "skewed":        # (a,b,c),(c,d,a) pairs far from parallelograms: d moved by up to 0.4
    t = t.copy()
    rng = np.random.default_rng(7)
    t["p2"][1::2, :3] += rng.uniform(-0.4, 0.4, (len(t) // 2, 3)).astype(np.float32)
elif kind == "tiny":          # the box shrunk around the eye: margins near their floor
    t = t.copy()
    eye = np.array([0.0, 2.75, 4.0], np.float32)
    for f in ("p1", "p2", "p3"):
        t[f][:, :3] = (t[f][:, :3] - eye) * np.float32(0.01) + eye + np.array([0.0, 0.0, -0.05], np.float32)
elif kind.startswith("random:"):  # the fuzz scenes of tests/test_gpu_parity.py::_random_quad_scene
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests"))
    spec = importlib.util.spec_from_file_location("tgp", os.path.join(root, "tests", "test_gpu_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    t, m = mod._random_quad_scene(int(kind.split(":")[1]))
assert adl.init()
dev = adl.DeviceUtils.allocate()
dev.setOption(shim.PT_OPT_QUAD_FILTER, quad_filter)
r = Renderer(dev, t, m, W, H, want_stats=True)
r.render(spp)
out = np.zeros(shim.PT_STAT_WORDS, np.uint64)
r.stats.read(out, shim.PT_STAT_WORDS); dev.waitForCompletion()
samples, rays, pairs, ref_keep, flt_keep, viol = (int(x) for x in out[:6])
print("%-8s qf=%d %dx%d x %d: %d rays, %.4g pairs examined; reference keeps %.3f%%, filter keeps %.3f%%; VIOLATIONS: %d"
      % (kind, quad_filter, W, H, spp, rays, pairs, 100.0 * ref_keep / max(pairs, 1), 100.0 * flt_keep / max(pairs, 1), viol))
r.release(); adl.DeviceUtils.deallocate(dev)
if out[6] or out[7]:  # shared-u filter active: headroom of its error bounds (must be <= 1)
    r1, r3 = (float(np.array([x], np.uint64).astype(np.uint32).view(np.float32)[0]) for x in out[6:8])
    # r1 is rounding only (large headroom expected); r3 includes the Cauchy-Schwarz bound
    # |w.pvec| <= |w||e2| of a non-parallelogram pair, which rays do attain: it approaches 1 by design
    if quad_filter in (0, 4):  # packed Pluecker form: its own roundings against the reference's floats (both rounding only)
        print("         packed filter: max |un_here-un_ref|/deltaP = %.4f, max |det_here-det_ref| c/deltaD = %.4f" % (r1, r3))
    else:
        print("         shared-u bounds: max |un'+unA|/delta1 = %.4f, max (|det'-detA| c + |un'+unA|)/delta3 = %.4f" % (r1, r3))
    viol += int(r1 > 1.0 or r3 > 1.0)
sys.exit(1 if viol else 0)
