"""The oracle against an INDEPENDENT float64 model of the reference kernel, and against itself under the other legal
convention for OpenCL's dot / cross (CPU tier; VERDICT r02 "pin the oracle with an independent numeric model").

The oracle (oracle/pt_oracle.c) and the HIP kernels implement ONE arithmetic contract written by the same builder, and every
GPU parity test compares the two bit for bit: an error in the builder's READING of test/ClKernels/GenerateColors.cl that both
inherited would be invisible to all of them (only the 8-bit statistics of the reference's JPG, sensitivity ~1 %, would see
it).  tests/f64_model.py restates the kernel from the OpenCL source in different arithmetic (numpy float64, libm, no fused
multiply-add) and knows, for every path, how far each of the kernel's comparisons was from its boundary.  Two correct
evaluations of the same formulas in different arithmetic take the same decisions except next to a boundary, and where they
take the same decisions their per-path radiances agree to rounding.
"""
import numpy as np
import pytest

import f64_model as fm

U32 = 2.0 ** -24
N_PATHS = 20000


def _ggx_room(cornell, seed):
    """GGX-heavy scene: the Cornell box's 36 triangles with 70 % of the quads turned into glossy GGX surfaces of roughness
    0.003 ... 0.5 and random albedo, two emitters kept, plus ten random quads inside the room (half of them glossy)."""
    from oclpathtracer_amd import scene

    ctris, cmats = cornell
    rng = np.random.default_rng(seed)
    nq_extra = 10
    tris = np.zeros(len(ctris) + 2 * nq_extra, scene.TRIANGLE_DTYPE)
    mats = np.zeros(len(cmats) + nq_extra, scene.MATERIAL_DTYPE)
    tris[: len(ctris)] = ctris
    mats[: len(cmats)] = cmats
    for q in range(len(cmats)):
        if mats[q]["emissive"][0] > 0:
            continue
        mats[q]["albedo"] = tuple(rng.uniform(0.2, 0.9, 3)) + (1.0,)
        if rng.random() < 0.7:
            mats[q]["type"] = scene.SPECULAR
            mats[q]["roughness"] = np.float32(10.0 ** rng.uniform(-2.5, -0.3))
        else:
            mats[q]["type"] = scene.DIFFUSE
            mats[q]["roughness"] = 0.0
    eye = np.array([0.0, 2.75, 4.0], np.float32)
    for k in range(nq_extra):
        a = np.array([rng.uniform(-2.2, 2.2), rng.uniform(0.3, 4.8), rng.uniform(-5.0, -0.5)], np.float32)
        e1 = rng.uniform(-1, 1, 3).astype(np.float32) * np.float32(0.9)
        e2 = rng.uniform(-1, 1, 3).astype(np.float32) * np.float32(0.9)
        if np.dot(np.cross(e2, e1), a - eye) < 0:
            e1, e2 = e2, e1
        b, c, d = a + e1, a + e1 + e2, a + e2
        q = len(cmats) + k
        for j, (p1, p2, p3) in enumerate(((a, b, c), (c, d, a))):
            t = tris[len(ctris) + 2 * k + j]
            t["p1"][:3], t["p2"][:3], t["p3"][:3] = p1, p2, p3
            t["id"] = q
        mats[q]["albedo"] = tuple(rng.uniform(0.2, 0.9, 3)) + (1.0,)
        mats[q]["emissive"] = (0.0, 0.0, 0.0, 1.0)
        glossy = k % 2 == 0
        mats[q]["type"] = scene.SPECULAR if glossy else scene.DIFFUSE
        mats[q]["roughness"] = np.float32(10.0 ** rng.uniform(-2.5, -0.3)) if glossy else 0.0
    return tris, mats


def _compare(oracle, tris, mats, seed, what, share_1e4=0.99):
    W = H = 256
    rng = np.random.default_rng(seed)
    gids = rng.integers(0, W * H, N_PATHS)
    frames = rng.integers(0, 10000, N_PATHS)          # the reference renders frames 0 .. 9999 (RaytraceTest.cpp:250)
    rad, hits, margin, sens, _ = fm.trace(tris, mats, gids, frames, W, H)
    orad, ohits = oracle.paths(tris, mats, gids, frames, W, H)
    same = (hits == ohits).all(axis=1)
    n_flip = int((~same).sum())
    near5 = margin < 1e-5
    report = ("%s: %d paths, %.2f bounces on average; hit sequences differ in %d (%.3f %%); within 1e-5 of a decision boundary: "
              "%.2f %% of the paths, %d of the differing ones; largest margin of a differing path %.2e"
              % (what, N_PATHS, float((hits >= 0).sum(axis=1).mean()), n_flip, 100.0 * n_flip / N_PATHS, 100.0 * near5.mean(),
                 int((~same & near5).sum()), float(margin[~same].max()) if n_flip else 0.0))
    print(report)
    # 1. decisions.  A misread operand order, sign, constant or branch changes decisions everywhere (most paths, margins of
    #    order 0.1 .. 1); two correct evaluations differ only next to a boundary.  A path takes ~6 bounces x 36+ triangles x 4
    #    comparisons, so its SMALLEST margin is small (below 1e-3 for 40-50 % of the paths) -- and binary32 reaches a boundary
    #    from further away than 1e-5 after a glossy bounce: sin(theta) = sqrt(1 - cos^2(theta)) at roughness 0.008 is known to
    #    5e-4 only, which moves the next hit point by up to ~1e-4 of the scene (f64_model's docstring).  Measured: 0.02-0.08 %
    #    of the paths differ, all with margins below 2e-3.  Hard bounds: at most 0.2 % differ, none with a margin of 1e-2 or
    #    more -- an order of magnitude below where an error of reading would show; the 1e-5 figures are reported.
    assert n_flip <= N_PATHS * 2e-3, report
    assert not (~same & (margin >= 1e-2)).any(), report
    # 2. radiance of the paths that took the same decisions: binary32 rounding only.  Bound: 2e-5 + 64 u sens (sens = 0 for
    #    a path without glossy bounces, so those must agree to 2e-5 outright; measured: 7e-7).
    ok = same & np.isfinite(rad).all(axis=1) & np.isfinite(orad).all(axis=1)
    assert np.array_equal(np.isfinite(rad[same]), np.isfinite(orad[same])), what + ": non-finite radiances differ"
    rel = np.abs(rad[ok] - orad[ok]).max(axis=1) / np.maximum(np.abs(rad[ok]).max(axis=1), 1e-3)
    bound = 2e-5 + 64.0 * U32 * sens[ok]
    rep2 = ("%s: per-path radiance, relative difference to the float64 model: max %.2e, 99.9th percentile %.2e, within 1e-4: %.2f %%, "
            "within 1e-5: %.2f %%; paths without a glossy bounce: max %.2e"
            % (what, rel.max(), np.quantile(rel, 0.999), 100.0 * (rel <= 1e-4).mean(), 100.0 * (rel <= 1e-5).mean(),
               rel[sens[ok] == 0].max() if (sens[ok] == 0).any() else 0.0))
    print(rep2)
    assert (rel <= bound).all(), rep2 + "; worst ratio to the bound %.2f" % float((rel / bound).max())
    assert (rel <= 1e-4).mean() >= share_1e4, rep2
    return report, rep2


def test_oracle_agrees_with_the_float64_model_on_the_cornell_box(oracle, cornell):
    tris, mats = cornell
    _compare(oracle, tris, mats, 7, "cornell box")


def test_oracle_agrees_with_the_float64_model_on_a_glossy_scene(oracle, cornell):
    tris, mats = _ggx_room(cornell, 11)
    assert (mats["type"] == 2).sum() >= 12
    _compare(oracle, tris, mats, 13, "glossy room (%d triangles, %d of %d materials GGX)" % (len(tris), int((mats["type"] == 2).sum()), len(mats)),
             share_1e4=0.97)


def test_float64_model_reproduces_the_integer_kats():
    """the model's own RNG against the hand-derived known answers of SURVEY.md S8c (so that it is not the oracle's RNG twice)"""
    assert int(fm.hash_u32(np.array([0, 1, 2, 255, 9999]))[3]) == 2223525580
    assert [int(v) for v in fm.hash_u32(np.array([0, 1, 2, 255, 9999]))] == [12345, 1103527590, 2207042835, 2223525580, 277963676]
    s = np.array([0, 12345], np.uint64)
    states, vals = [], []
    for _ in range(3):
        s, v = fm.random_float(s)
        states.append([int(x) for x in s])
        vals.append(v.copy())
    assert [st[0] for st in states] == [0x4E6EBE5B, 0xC0FBEE26, 0x83912C50]
    assert [st[1] for st in states] == [0x3D8A7E50, 0x57DB8A89, 0xD9B36308]
    np.testing.assert_allclose([v[0] for v in vals], [0.30637732, 0.75384414, 0.5139339], rtol=0, atol=3e-8)
    np.testing.assert_allclose([v[1] for v in vals], [0.24039449, 0.34319368, 0.8503935], rtol=0, atol=3e-8)


def test_convention_to_convention_distance(oracle, cornell):
    """How far two CONFORMING evaluations of the reference lie apart: the oracle's contract (FMA forms of dot / cross) against
    the same restatement with every product and sum rounded (libptoracle_nofma.so).  OpenCL allows either, so this -- not
    0 -- is the honest tolerance for "matches the OpenCL reference" on an image: the RMS is dominated by the few paths that
    flip a decision, each worth a whole sample, and it falls like 1 / spp."""
    tris, mats = cornell
    W = H = 64
    spp = 256
    a = oracle.render(tris, mats, W, H, spp)
    b = oracle.render(tris, mats, W, H, spp, variant="nofma")
    assert np.array_equal(np.isnan(a), np.isnan(b))
    d = (a.astype(np.float64) - b.astype(np.float64))[:, :3]
    fin = np.isfinite(d)
    rms = float(np.sqrt(np.mean(d[fin] ** 2)))
    changed = float((np.abs(d).max(axis=1) > 0).mean())
    big = float((np.abs(d).max(axis=1) > 1e-3).mean())
    print("FMA vs no-FMA convention, cornell %dx%d x %d spp (gamma-encoded running mean): RMS %.3e, pixels that differ at all %.1f %%, "
          "by more than 1e-3: %.2f %%, max %.3e" % (W, H, spp, rms, 100 * changed, 100 * big, float(np.abs(d[fin]).max())))
    # per path: the two conventions take different decisions in ~1e-4 of the paths
    rng = np.random.default_rng(3)
    gids = rng.integers(0, 256 * 256, 200000)
    frames = rng.integers(0, 10000, 200000)
    ra, ha = oracle.paths(tris, mats, gids, frames, 256, 256)
    rb, hb = oracle.paths(tris, mats, gids, frames, 256, 256, variant="nofma")
    flips = int((ha != hb).any(axis=1).sum())
    same = (ha == hb).all(axis=1)
    rel = np.abs(ra[same].astype(np.float64) - rb[same]).max(axis=1) / np.maximum(np.abs(ra[same]).max(axis=1), 1e-3)
    print("per path (200 000 paths): %d take different decisions (%.4f %%); same-decision paths differ by at most %.2e relative, %.1f %% bit-identical"
          % (flips, 100.0 * flips / 200000, float(rel.max()), 100.0 * float((rel == 0).mean())))
    assert 0.0 < rms < 2e-2, rms            # the conventions DO differ, and by Monte-Carlo flips only
    assert big < 0.05
    assert flips <= 200000 * 1e-3
    assert rel.max() < 2e-3


def test_image_distance_to_the_float64_model(oracle, cornell):
    """The whole path including the accumulation (GenerateColors.cl:290-300,314-321), as an image: the oracle's framebuffer
    against the float64 model's paths folded by the same gamma -> mean -> degamma recurrence in float64.  Per-path radiances
    agree to ~1e-6 and 0.04 % of the paths take another decision, each worth one sample of one pixel: the RMS is
    Monte-Carlo-flip noise that falls like 1 / spp, and it is the measured distance between this build's arithmetic and an
    independent evaluation of the reference."""
    tris, mats = cornell
    W = H = 20
    spp = 96
    gid = np.repeat(np.arange(W * H), spp)
    frame = np.tile(np.arange(spp), W * H)
    rad, _, _, _, _ = fm.trace(tris, mats, gid, frame, W, H)
    rad = rad.reshape(W * H, spp, 3)
    g = 1.0 / float(np.float32(2.2))    # 1.0f / 2.2f is a float expression (:292); 2.2f a float literal (:298)
    fb = np.zeros((W * H, 3))
    with np.errstate(invalid="ignore", divide="ignore"):
        for z in range(spp):
            c = rad[:, z, :]
            fb = c ** float(np.float32(g)) if z == 0 else ((fb ** float(np.float32(2.2)) * (z - 1) + c) / z) ** float(np.float32(g))
    want = oracle.render(tris, mats, W, H, spp)[:, :3].astype(np.float64)
    fin = np.isfinite(fb) & np.isfinite(want)
    assert np.array_equal(np.isfinite(fb), np.isfinite(want))
    d = (fb - want)[fin]
    rms = float(np.sqrt(np.mean(d * d)))
    print("oracle vs float64 model, cornell %dx%d x %d spp, gamma-encoded running mean: RMS %.3e, max %.3e, pixels differing by more than 1e-4: %.2f %%"
          % (W, H, spp, rms, float(np.abs(d).max()), 100.0 * float((np.abs(fb - want).max(axis=1) > 1e-4).mean())))
    assert rms < 2e-3, rms
