#!/usr/bin/env python3
"""Randomised parity campaign on the GPU box: random quad scenes (the generator of tests/test_gpu_parity.py), random image
geometries, frame counts, depth caps, stripe splits, search modes and options -- every render compared bit for bit
(NaN masks equal) with the CPU oracle, ray counts included.  usage: python tools/gpu_fuzz.py [first_seed] [count]"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oclpathtracer_amd import adl, scene, shim
from oclpathtracer_amd.render import Renderer
from oracle import ptoracle

spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
tgp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tgp)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ptoracle.build()
assert adl.init()
dev = adl.DeviceUtils.allocate()
cornell = scene.load_model()
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    kind = rng.integers(0, 4)
    if kind == 0:
        tris, mats = cornell
    elif kind == 1:
        tris, mats = tgp._random_quad_scene(seed)
    elif kind == 2:   # nested Cornell copies: 72 ... 468 triangles (chunks, tiled mode)
        parts = []
        for c in range(int(rng.integers(2, 14))):
            t = cornell[0].copy(); k = np.float32(1.0 - 0.06 * c)
            for f in ("p1", "p2", "p3"):
                t[f][:, :3] = t[f][:, :3] * k + np.array([0.0, 2.7, -2.8], np.float32) * (np.float32(1.0) - k)
            parts.append(t)
        tris, mats = np.concatenate(parts), cornell[1]
    else:
        tris, mats = scene.make_soup(int(rng.integers(40, 3000)))
    W, H = int(rng.integers(1, 200)), int(rng.integers(1, 120))
    frames = int(rng.integers(1, 7)); depth = int(rng.choice([1, 2, 3, 16, 16, 16, 40]))
    n_ranks = int(rng.choice([1, 1, 2, 3, 8])); rank = int(rng.integers(0, n_ranks)); stripe = int(rng.integers(1, 20))
    accel = int(rng.choice([0, 0, 1, 2])) if len(tris) >= 2 else 0
    opts = {shim.PT_OPT_ACCEL: accel, shim.PT_OPT_PRIMARY_MASKS: int(rng.integers(0, 2)), shim.PT_OPT_QUAD_FILTER: int(rng.choice([0, 0, 1])),
            shim.PT_OPT_CHUNK_FRAMES: int(rng.choice([0, 0, 2]))}
    for k, v in opts.items(): dev.setOption(k, v)
    r = Renderer(dev, tris, mats, W, H, n_ranks=n_ranks, rank=rank, stripe_rows=stripe, want_stats=True)
    try:
        r.render(frames, max_bounces=depth)
        got, gst, rows = r.read(), r.read_stats(), r.global_rows()
    finally:
        r.release()
        for k in opts: dev.setOption(k, 1 if k == shim.PT_OPT_PRIMARY_MASKS else 0)
    fb = np.zeros((H * W, 4), np.float32)
    rays = 0
    for row in rows:
        _, st = ptoracle.render(tris, mats, W, H, frames, max_bounces=depth, fb=fb, gid_begin=int(row) * W, gid_count=W, nthreads=4, want_stats=True)
        rays += st["rays"]
    want = fb.reshape(H, W, 4)[rows].reshape(-1, 4) if len(rows) else np.zeros((0, 4), np.float32)
    try:
        tgp.assert_fb_equal(got, want, "seed %d" % seed)
        assert gst["rays"] == rays, (gst["rays"], rays)
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed %d kind %d ntri %d %dx%d f%d d%d ranks %d/%d stripe %d opts %s: %s" % (seed, kind, len(tris), W, H, frames, depth, rank, n_ranks, stripe, opts, str(e)[:300]), flush=True)
    if (seed - first) % 250 == 249:   # (a line now and then: a run that prints nothing for minutes is taken to be hung)
        print("  ... %d cases, %d mismatches, %.0f s" % (seed - first + 1, bad, time.time() - t0), flush=True)
print("fuzz: %d cases from seed %d, %d mismatches, %.1f s" % (count, first, bad, time.time() - t0), flush=True)
adl.DeviceUtils.deallocate(dev)
sys.exit(1 if bad else 0)
