"""Pins the CPU oracle (oracle/pt_oracle.c) with every known answer the reference offers.

The reference's own tests hold no golden vectors for this path (no test asserts a
pixel, a random number or an intersection), so the pins are the hand-derivable known answers
listed there, each traced to the reference line it follows.
"""
import json
import os

import mpmath
import numpy as np

from conftest import GOLDEN, assert_fb_equal


# ---- GenerateColors.cl:47-71 -----------------------------------------------------------------
def test_hash_known_answers(oracle):
    # hashUInt32(x) = 1103515245*x + 12345 mod 2^32 (:57)
    for x, want in [(0, 12345), (1, 1103527590), (2, 2207042835), (255, 2223525580), (9999, 277963676)]:
        assert oracle.hash_u32(x) == want
        assert want == (1103515245 * x + 12345) % 2**32


def test_rng_known_answers(oracle):
    states, vals = oracle.random_floats(0, 3)
    assert states == [0x4E6EBE5B, 0xC0FBEE26, 0x83912C50]
    assert [float(v) for v in vals] == [float(np.float32(s) * np.float32(2.3283064365386963e-10)) for s in states]
    assert vals == [np.float32(0.30637732), np.float32(0.75384414), np.float32(0.5139339)]
    states, vals = oracle.random_floats(12345, 3)
    assert states == [0x3D8A7E50, 0x57DB8A89, 0xD9B36308]
    assert vals == [np.float32(0.24039449), np.float32(0.34319368), np.float32(0.8503935)]


def test_rng_python_model(oracle):
    """Independent integer model of getRandomFloat (:63-70) over many seeds."""
    def step(s):
        s = (s ^ 61) ^ (s >> 16)
        s = (s + (s << 3)) & 0xFFFFFFFF
        s = s ^ (s >> 4)
        s = (s * 0x27D4EB2D) & 0xFFFFFFFF
        s = s ^ (s >> 15)
        return (1103515245 * s + 12345) & 0xFFFFFFFF
    rng = np.random.default_rng(7)
    for seed in [0, 1, 0xFFFFFFFF, 0x80000000] + [int(x) for x in rng.integers(0, 2**32, 50)]:
        states, vals = oracle.random_floats(seed, 4)
        s = seed
        for st, v in zip(states, vals):
            s = step(s)
            assert st == s
            assert v == np.float32(s) * np.float32(2.0**-32)
            assert 0.0 <= v <= 1.0  # (float) rounds to nearest: 1.0 is reachable


def test_uint_to_float_rounds_to_one():
    assert np.float32(0xFFFFFFFF) * np.float32(2.3283064365386963e-10) == np.float32(1.0)


# ---- camera constants, GenerateColors.cl:263-288 ---------------------------------------------
def test_camera_constants():
    fov = np.float32((60.0 * np.pi) / 180.0)  # evaluated in double, narrowed (:267)
    assert float(fov) == 1.0471975803375244
    half = np.float32(0.5) * fov
    mpmath.mp.prec = 200
    t = mpmath.tan(mpmath.mpf(float(half)))
    lo, hi = np.float32(0.57735026), np.nextafter(np.float32(0.57735026), np.float32(1))
    # correctly rounded tan(0.5f*fov) is 0x1.279a74p-1 (= 0.57735026f), the constant PTSPEC fixes
    assert abs(t - mpmath.mpf(float(lo))) < abs(t - mpmath.mpf(float(hi)))
    assert float(lo) == float.fromhex("0x1.279a74p-1")
    assert np.float32(1.0) / np.float32(2.2) == np.float32(0.45454544)
    assert np.float32(6.28318530718) == np.float32(6.2831855)
    assert np.float32(0.31830988618) == np.float32(0.31830987)


def test_centre_pixel_zero_jitter_looks_down_minus_z(oracle):
    """x = (float)xc + xi - 0.5; with xc = W/2 and both draws exactly 0.5 the ray is (0,0,-1)
    (:278-284).  Find a seed state whose next two draws are known and check the mapping formula
    against an independent float32 evaluation instead."""
    W = H = 64
    for seed in (0, 12345, 99):
        org, d, s_after = oracle.generate_ray(W // 2, H // 2, W, H, seed)
        states, xi = oracle.random_floats(seed, 2)
        assert s_after == states[1]
        f = np.float32
        invW = f(1) / f(W)
        angle = f(float.fromhex("0x1.279a74p-1"))
        x = f(f(W // 2) + xi[0]) - f(0.5)
        y = f(f(H // 2) + xi[1]) - f(0.5)
        x = f(f(f(f(2) * f(f(x + f(0.5)) * invW)) - f(1)) * angle) * f(f(W) / f(H))
        y = f(-(f(1) - f(f(2) * f(f(y + f(0.5)) * invW)))) * angle
        dd = np.array([x, -y, -1], np.float64)
        dd /= np.linalg.norm(dd)
        assert np.allclose(org, [0, 2.75, 4])
        assert np.allclose(d, dd, atol=3e-7)
        assert abs(np.linalg.norm(d.astype(np.float64)) - 1) < 2e-7


# ---- scene decode, RaytraceTest.cpp:87-198 ----------------------------------------------------
def test_scene_table_matches_fixture(cornell):
    tris, mats = cornell
    with open(os.path.join(GOLDEN, "cornell_scene_table.json")) as f:
        table = json.load(f)
    assert len(tris) == 36 == len(table["triangles"]) and len(mats) == 18 == len(table["materials"])
    for t, w in zip(tris, table["triangles"]):
        assert t["p1"].tolist() == w["p1"] and t["p2"].tolist() == w["p2"] and t["p3"].tolist() == w["p3"]
        assert int(t["id"]) == w["id"]
    for m, w in zip(mats, table["materials"]):
        assert m["albedo"].tolist() == w["albedo"] and m["emissive"].tolist() == w["emissive"]
        assert int(m["type"]) == w["type"] and float(m["roughness"]) == w["roughness"]


def test_scene_structure(cornell):
    from oclpathtracer_amd import scene

    tris, mats = cornell
    meshes = scene.parse_meshes(open(scene.DEFAULT_SCENE, "rb").read())
    assert [(len(i), len(v)) for _, i, v in meshes] == [(2, 8), (3, 12), (1, 4), (1, 4), (1, 4), (10, 40)]
    assert [float(t) for t, _, _ in meshes] == [0.5, 0.5, 5.0, 0.5, 0.5, 0.5]
    assert os.path.getsize(scene.DEFAULT_SCENE) == 1516
    # quad (a,b,c,d) -> (a,b,c),(c,d,a) with one id
    assert np.array_equal(tris["id"], np.repeat(np.arange(18), 2))
    for q in range(18):
        t1, t2 = tris[2 * q], tris[2 * q + 1]
        assert np.array_equal(t1["p3"], t2["p1"]) and np.array_equal(t1["p1"], t2["p3"])
    assert np.all(tris["p1"][:, 3] == 0) and np.all(tris["p2"][:, 3] == 0)
    # materials: light = quad 5 (mesh 2), emissive 30, albedo overridden to 0.7
    assert mats["emissive"][5].tolist() == [30.0, 30.0, 30.0, 1.0]
    assert np.allclose(mats["albedo"][5], [0.7, 0.7, 0.7, 1.0])
    assert all(mats["emissive"][i].tolist() == [0.0, 0.0, 0.0, 1.0] for i in range(18) if i != 5)
    assert np.allclose(mats["albedo"][6], [0.6, 0, 0, 1]) and np.allclose(mats["albedo"][7], [0, 0.6, 0, 1])
    assert np.all(mats["type"][:8] == 1) and np.all(mats["type"][8:] == 2)
    assert np.allclose(mats["albedo"][8:], [0.5, 0.35, 0.05, 0.0]) and np.all(mats["roughness"][8:] == np.float32(0.008))


# ---- intersectWorld invariants -----------------------------------------------
def test_intersect_world_invariants(oracle, cornell):
    tris, _ = cornell
    hit, t, p, n, tri = oracle.intersect_world(tris, (0, 4.5, 4), (0, 0, -1))
    assert hit and tri == 3 and abs(t - 9.592) < 1e-5 and abs(p[2] + 5.592) < 1e-5  # back wall, quad 1
    hit, t, p, n, tri = oracle.intersect_world(tris, (0, 2.75, -2.8), (0, 1, 0))
    assert hit and tris["id"][tri] == 5 and abs(t - 2.73) < 1e-5  # the light, front-facing from below
    hit, t, p, n, tri = oracle.intersect_world(tris, (0, 6.0, -2.8), (0, -1, 0))
    assert hit and tris["id"][tri] != 5 and abs(t - 6.0) < 1e-5  # light culled from above (:100): floor
    hit, t, p, n, tri = oracle.intersect_world(tris, (0, 2.75, 4), (0, 0, -1))
    assert hit and tris["id"][tri] >= 8 and p[2] > -3.4  # the camera's centre ray meets the tall box
    hit, *_ = oracle.intersect_world(tris, (0, 2.75, 4), (0, 0, 1))
    assert not hit  # the open front
    # exact tie: first triangle in buffer order wins (strict t < tmax, :125)
    dup = np.concatenate([tris[2:4], tris[2:4]])
    hit, t, p, n, tri = oracle.intersect_world(dup, (0, 4.5, 4), (0, 0, -1))
    assert hit and tri in (0, 1)


def test_radiance_invariants(oracle, cornell):
    tris, mats = cornell
    # a scene of only the light: a path that sees it returns 3*30 = 90 on first hit (:241) plus
    # whatever the next bounce adds (the open scene then misses: + mask*0.45)
    W = H = 32
    light = tris[10:12].copy()
    lm = mats.copy()
    seen = 0
    for gid in range(0, W * H, 7):
        c = oracle.radiance(light, lm, gid, W, H, 1, 16)
        assert np.all(c >= 0)
        if c[0] > 1:
            seen += 1
            assert c[0] >= 90.0
        else:
            assert np.allclose(c, 0.45)  # miss: mask(1) * bg
    # depth cap 1 in the full scene: radiance is emissive*3 of the first hit or the background
    for gid in range(0, W * H, 11):
        c = oracle.radiance(tris, mats, gid, W, H, 3, 1)
        assert np.allclose(c, 0.0) or np.allclose(c, 90.0) or np.allclose(c, 0.45)


def test_accumulate_semantics(oracle, cornell):
    """GenerateColors.cl:314-321: frame 0 stores gamma(c); frame z stores
    gamma((degamma(old)*(z-1)+c)/z) -- so frame 1 discards frame 0, w == 1 always."""
    tris, mats = cornell
    W = H = 16
    fb0 = oracle.render(tris, mats, W, H, 1)
    assert np.all(fb0[:, 3] == 1.0)
    g = np.float32(1.0) / np.float32(2.2)
    for gid in (0, 17, 200):
        c = oracle.radiance(tris, mats, gid, W, H, 0, 16)
        assert np.allclose(fb0[gid, :3], oracle.pow_array(c, float(g)), rtol=0, atol=0)
    a = oracle.render(tris, mats, W, H, 3)                       # frames 0,1,2
    junk = np.full((W * H, 4), 123.25, np.float32)
    b = oracle.render(tris, mats, W, H, 2, frame_begin=1, fb=junk)  # frames 1,2 over garbage
    assert_fb_equal(a, b, "frame 0 is discarded by frame 1")
    # two-frame mean by hand: after frame 2 the value is gamma((degamma(gamma(c1))*1 + c2)/2)
    for gid in (3, 99):
        c1 = oracle.radiance(tris, mats, gid, W, H, 1, 16)
        c2 = oracle.radiance(tris, mats, gid, W, H, 2, 16)
        v1 = oracle.pow_array((oracle.pow_array(fb0[gid, :3], 2.2) * np.float32(0) + c1) / np.float32(1), float(g))
        v2 = oracle.pow_array((oracle.pow_array(v1, 2.2) * np.float32(1) + c2) / np.float32(2), float(g))
        assert np.array_equal(v2.view(np.uint32), a[gid, :3].view(np.uint32))


# ---- PTSPEC transcendental helpers vs high precision ---------------------------------------------
def _ulp_err(got32, exact64):
    got = got32.astype(np.float64)
    ulp = np.spacing(np.abs(exact64).astype(np.float32)).astype(np.float64)
    return np.abs(got - exact64) / ulp


def test_sincos_accuracy(oracle):
    rng = np.random.default_rng(3)
    xi = np.concatenate([rng.random(200000, dtype=np.float32), np.array([0, 1, 0.25, 0.5, 0.75], np.float32)])
    phi = np.float32(6.28318530718) * xi
    s, c = oracle.sincos(phi)
    es, ec = np.sin(phi.astype(np.float64)), np.cos(phi.astype(np.float64))
    # PTSPEC on [0, 2 pi]: binary32 evaluation, <= 1.43 ulp over EVERY binary32 angle of the range
    # (tools/check_sincos_f32.c, exhaustive); here a sample, zeros of the functions included
    assert _ulp_err(s, es).max() <= 1.5
    assert _ulp_err(c, ec).max() <= 1.5
    assert np.max(np.abs(s.astype(np.float64) - es)) < 1e-7 and np.max(np.abs(c.astype(np.float64) - ec)) < 1e-7
    s0, c0 = oracle.sincos(np.array([0.0], np.float32))
    assert s0[0] == 0.0 and c0[0] == 1.0
    # larger angles (the spec covers phi >= 0 only): binary64 evaluation rounded once -- correctly
    # rounded except double-rounding near-ties
    far = rng.uniform(6.2832, 100.0, 100000).astype(np.float32)
    s, c = oracle.sincos(far)
    es, ec = np.sin(far.astype(np.float64)), np.cos(far.astype(np.float64))
    big = np.abs(es) > 1e-3
    assert _ulp_err(s[big], es[big]).max() <= 0.5 + 1e-4
    big = np.abs(ec) > 1e-3
    assert _ulp_err(c[big], ec[big]).max() <= 0.5 + 1e-4


def test_pow_accuracy_and_edges(oracle):
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.random(100000, dtype=np.float32) * 100, 10.0 ** rng.uniform(-30, 30, 50000).astype(np.float32),
                        np.array([1.0, 0.45, 90.0, 1e-40, 3e38], np.float32)]).astype(np.float32)
    mpmath.mp.prec = 120
    for y in (2.2, float(np.float32(1.0) / np.float32(2.2))):
        yf = np.float32(y)
        got = oracle.pow_array(x, float(yf))
        exact = np.exp(np.log(x.astype(np.float64)) * float(yf))
        fin = np.isfinite(exact) & (exact < 3.4e38) & (exact > 1.2e-38)
        assert _ulp_err(got[fin], exact[fin]).max() <= 0.5 + 2e-3  # float64 reference itself is ~1e-16 relative
        assert np.array_equal(np.isinf(got), exact > 3.4028235677973366e38)
        for xv in (0.3, 7.5, 1e-20, 12345.678):  # spot-check against 120-bit arithmetic
            e = mpmath.power(mpmath.mpf(float(np.float32(xv))), mpmath.mpf(float(yf)))
            g = float(oracle.pow_array(np.array([xv], np.float32), float(yf))[0])
            assert abs(g - float(e)) <= 0.5000001 * float(np.spacing(np.float32(float(e))))
    edge = np.array([0.0, -0.0, np.inf, -1.0, np.nan, 1.0], np.float32)
    r = oracle.pow_array(edge, 2.2)
    assert r[0] == 0 and r[1] == 0 and not np.signbit(r[1]) and np.isinf(r[2]) and np.isnan(r[3]) and np.isnan(r[4]) and r[5] == 1.0
    # y == 2 is x*x exactly (PTSPEC; GGX denominator GenerateColors.cl:177)
    xs = rng.random(1000, dtype=np.float32) * 3
    assert np.array_equal(oracle.pow_array(xs, 2.0), xs * xs)


# ---- golden framebuffers: the oracle is platform-independent -------------------------------------
def test_oracle_reproduces_golden_framebuffers(oracle, cornell):
    tris, mats = cornell
    with open(os.path.join(GOLDEN, "work_counters.json")) as f:
        meta = json.load(f)
    for name, m in meta.items():
        want = np.load(os.path.join(GOLDEN, name + ".npy"))
        got, st = oracle.render(tris, mats, m["W"], m["H"], m["frames"], max_bounces=m["max_bounces"], want_stats=True)
        assert_fb_equal(got, want, name)
        for k, v in st.items():
            assert v == m[k], (name, k)
        # tally identities
        assert st["tests"] == st["rays"] * 36
        assert st["tests"] == st["cull"] + st["rej_u"] + st["rej_v"] + st["reach_t"]
        assert st["samples"] == st["miss"] + st["term_pdf"] + st["term_depth"]
        assert st["rays"] == st["miss"] + st["shade_diffuse"] + st["shade_specular"]


def test_oracle_threads_and_ranges_are_deterministic(oracle, cornell):
    tris, mats = cornell
    W, H, frames = 40, 24, 3
    a = oracle.render(tris, mats, W, H, frames, nthreads=1)
    b = oracle.render(tris, mats, W, H, frames, nthreads=7)
    assert_fb_equal(a, b, "thread count")
    c = np.zeros_like(a)
    for g0 in range(0, W * H, 100):
        oracle.render(tris, mats, W, H, frames, fb=c, gid_begin=g0, gid_count=min(100, W * H - g0))
    assert_fb_equal(a, c, "gid ranges")
    d = oracle.render(tris, mats, W, H, 1)
    oracle.render(tris, mats, W, H, 2, frame_begin=1, fb=d)
    assert_fb_equal(a, d, "resume")


def test_converged_image_matches_reference_render_colours(oracle, cornell):
    """Weak visual oracle: after f2c(sqrt(.)) the left wall is green, the right
    wall red, the ceiling light saturated -- as in the reference's FinalRendered_Specular.jpg."""
    from oclpathtracer_amd import scene

    tris, mats = cornell
    W = H = 64
    fb = oracle.render(tris, mats, W, H, 48)
    img = scene.f2c(fb[:, :3]).reshape(H, W, 3).astype(np.int64)
    left, right = img[20:44, 2:6].mean((0, 1)), img[20:44, 58:62].mean((0, 1))
    assert left[1] > 150 and left[0] < 90 and left[2] < 90
    assert right[0] > 150 and right[1] < 90 and right[2] < 90
    assert np.all(img[8:10, 30:34] == 255)  # the light


def test_converged_image_matches_reference_render_quantitatively(oracle, cornell):
    """The one artefact of the REAL OpenCL path tracer the reference holds is its rendered image,
    FinalRendered_Specular.jpg (512x512, 8-bit, lossy).  Its 32x32-pixel block means are committed as
    tests/golden/reference_jpg_blocks_16x16.npy (made by tests/golden/make_jpg_blocks.py).  The oracle,
    rendered at 128x128 x 800 frames and pushed through the reference's own output stage
    f2c(sqrt(.)) (RaytraceTest.cpp:78-83, 280-285), must reproduce them within Monte-Carlo + JPEG
    noise: measured mean |diff| 1.9 of 255, worst block 11.5, correlation 0.9996 (5.7 / 27 / 0.9988 at
    200 frames: the difference shrinks as the render converges).  This pins camera, scene decode,
    materials, both BRDFs, the running gamma mean and the tonemap against the reference's real
    output -- statistically; bit-level parity with OpenCL stays unpinned."""
    from oclpathtracer_amd import scene

    ref = np.load(os.path.join(GOLDEN, "reference_jpg_blocks_16x16.npy")).astype(np.float64)
    tris, mats = cornell
    W, frames, G = 128, 800, 16
    fb = oracle.render(tris, mats, W, W, frames)
    img = scene.f2c(fb[:, :3]).reshape(W, W, 3).astype(np.float64)
    blocks = img.reshape(G, W // G, G, W // G, 3).mean(axis=(1, 3))
    diff = np.abs(blocks - ref)
    corr = np.corrcoef(blocks.ravel(), ref.ravel())[0, 1]
    assert diff.mean() < 3.5, diff.mean()
    assert diff.max() < 22.0, diff.max()
    assert corr > 0.999, corr
