"""GPU parity tests proper: the HIP hot path (through the C ABI of libptshim.so) against the CPU
oracle on the same inputs, and against the committed golden fixtures.

Bar: BIT-EXACT on every finite value and identical NaN masks.  The north-star tolerance
(pixels within 1e-4 RMS) is asserted as well, written out as RMS_TOL; it is implied by
bit-exactness and kept so a future relaxation of the arithmetic contract still has a gate.
Nothing here reads /root/reference.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_fb_equal, rms_diff

pytestmark = pytest.mark.gpu

RMS_TOL = 1e-4  # BASELINE.json north_star: "pixels within 1e-4 RMS of the OpenCL reference"




def _render_gpu(device, tris, mats, W, H, frames, *, depth=16, frame_begin=0, fb_init=None, **kw):
    from oclpathtracer_amd import shim
    from oclpathtracer_amd.render import Renderer

    r = Renderer(device, tris, mats, W, H, **kw)
    try:
        if fb_init is not None:
            r.fb.write(np.ascontiguousarray(fb_init, np.float32), r.local_pixels)
        r.render(frames, frame_begin=frame_begin, max_bounces=depth)
        return r.read()
    finally:
        r.release()


GOLDEN_CASES = ["cornell_64x64_f1_d16", "cornell_64x64_f2_d16", "cornell_64x64_f8_d16", "cornell_64x64_f8_d2",
                "cornell_40x24_f5_d16"]


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_fused_render_matches_golden(device, cornell, name):
    with open(os.path.join(GOLDEN, "work_counters.json")) as f:
        meta = json.load(f)[name]
    want = np.load(os.path.join(GOLDEN, name + ".npy"))
    tris, mats = cornell
    got = _render_gpu(device, tris, mats, meta["W"], meta["H"], meta["frames"], depth=meta["max_bounces"])
    assert_fb_equal(got, want, name)
    assert rms_diff(got, want) <= RMS_TOL


@pytest.mark.parametrize("W,H,frames,depth", [(256, 256, 1, 16),      # BASELINE configs[0]
                                              (128, 128, 16, 16),
                                              (512, 512, 4, 2),        # configs[1] shape, fewer frames
                                              (96, 33, 7, 16),         # ragged
                                              (1, 1, 3, 16), (64, 1, 2, 16), (1, 70, 2, 5)])
def test_fused_render_matches_oracle(device, cornell, oracle, W, H, frames, depth):
    tris, mats = cornell
    want, st = oracle.render(tris, mats, W, H, frames, max_bounces=depth, want_stats=True)
    from oclpathtracer_amd import shim
    from oclpathtracer_amd.render import Renderer

    r = Renderer(device, tris, mats, W, H, want_stats=True)
    try:
        r.render(frames, max_bounces=depth)
        got = r.read()
        gst = r.read_stats()
    finally:
        r.release()
    assert_fb_equal(got, want, "%dx%d f%d d%d" % (W, H, frames, depth))
    assert rms_diff(got, want) <= RMS_TOL
    # integer work counters: exact
    assert gst["samples"] == st["samples"] == W * H * frames
    assert gst["rays"] == st["rays"]


def test_configs1_direct_full_size(device, cornell, oracle):
    """BASELINE configs[1]: 512x512, 64 spp, depth cap 2 -- full size, bit-exact."""
    tris, mats = cornell
    want = oracle.render(tris, mats, 512, 512, 64, max_bounces=2)
    got = _render_gpu(device, tris, mats, 512, 512, 64, depth=2)
    assert_fb_equal(got, want, "C2")


def test_configs2_full_size_sampled_pixels(device, cornell, oracle):
    """BASELINE configs[2]: 1024x1024, 256 spp, depth 16 rendered in full on the GPU; the oracle
    recomputes 2048 seeded pixel positions through all 256 frames (pixels are independent,
    GenerateColors.cl:305-321) and those must match bit for bit.  Whole-image properties: w == 1
    everywhere, no negative values, frame-0 independence."""
    tris, mats = cornell
    W = H = 1024
    frames = 256
    got = _render_gpu(device, tris, mats, W, H, frames).reshape(H * W, 4)
    assert np.all(got[:, 3] == 1.0)
    assert not np.any(got[:, :3] < 0)
    rng = np.random.default_rng(20261004)
    # 32 runs of 64 consecutive pixels (whole waves of the reference's launch) at seeded places
    starts = rng.integers(0, W * H - 64, 32)
    fb = np.zeros((H * W, 4), np.float32)
    for s in starts:
        oracle.render(tris, mats, W, H, frames, fb=fb, gid_begin=int(s), gid_count=64)
        assert_fb_equal(got[s:s + 64], fb[s:s + 64], "C3 pixels %d.." % s)


@pytest.mark.parametrize("W,H,frames", [(64, 64, 8), (200, 50, 4), (33, 97, 3)])
def test_primary_masks_on_and_off(device, cornell, oracle, W, H, frames):
    """PT_OPT_PRIMARY_MASKS: fresh waves of primary rays take their pass-1 survivors from per-pixel candidate masks
    (default) or run the filter like any other ray (0).  Same pixels, same ray counts, at footprints from 1/33 to
    1/200 of the image and aspect ratios 4:1 and 1:3 (the masks' margin grows with the pixel's footprint)."""
    from oclpathtracer_amd import shim
    from oclpathtracer_amd.render import Renderer

    tris, mats = cornell
    want, st = oracle.render(tris, mats, W, H, frames, want_stats=True)
    for on in (1, 0):
        device.setOption(shim.PT_OPT_PRIMARY_MASKS, on)
        r = Renderer(device, tris, mats, W, H, want_stats=True)
        try:
            r.render(frames)
            got, gst = r.read(), r.read_stats()
        finally:
            r.release()
            device.setOption(shim.PT_OPT_PRIMARY_MASKS, 1)
        assert_fb_equal(got, want, "primary masks %d, %dx%d" % (on, W, H))
        assert gst["rays"] == st["rays"]


def test_resume_equals_one_shot(device, cornell):
    """Accumulation is resumable (frame is an argument, GenerateColors.cl:314-321): 3+5 frames in
    two calls == 8 frames in one, and chunked staging == unchunked."""
    from oclpathtracer_amd import shim
    from oclpathtracer_amd.render import Renderer

    tris, mats = cornell
    one = _render_gpu(device, tris, mats, 64, 64, 8)
    r = Renderer(device, tris, mats, 64, 64)
    try:
        r.render(3)
        r.render(5)
        two = r.read()
    finally:
        r.release()
    assert_fb_equal(two, one, "resume")
    device.setOption(shim.PT_OPT_CHUNK_FRAMES, 3)
    try:
        chunked = _render_gpu(device, tris, mats, 64, 64, 8)
    finally:
        device.setOption(shim.PT_OPT_CHUNK_FRAMES, 0)
    assert_fb_equal(chunked, one, "chunked")
    want = np.load(os.path.join(GOLDEN, "cornell_64x64_f8_d16.npy"))
    assert_fb_equal(one, want, "one-shot vs golden")


def test_frame1_discards_frame0(device, cornell, oracle):
    """Frame 1 multiplies the stored value by (z-1) = 0 (GenerateColors.cl:320): whatever frame 0
    left (garbage included, but finite) does not influence frames >= 1."""
    tris, mats = cornell
    W = H = 32
    junk = np.random.default_rng(1).random((W * H, 4)).astype(np.float32) * 7
    a = _render_gpu(device, tris, mats, W, H, 4, frame_begin=1, fb_init=junk)
    b = _render_gpu(device, tris, mats, W, H, 5)
    assert_fb_equal(a, b, "frame0 discarded")
    # an infinite stored value poisons the pixel: 0 * inf = NaN, on both sides alike
    junk[5, 0] = np.inf
    got = _render_gpu(device, tris, mats, W, H, 2, frame_begin=1, fb_init=junk)
    want = oracle.render(tris, mats, W, H, 2, frame_begin=1, fb=junk.copy())
    assert_fb_equal(got, want, "inf poison")
    assert np.isnan(got[5, 0])


def test_reference_loop_matches_fused_and_oracle(device, cornell, oracle):
    """The RaytraceTest-shaped per-frame Launcher loop (both immediate and with deferred frame
    batching) gives the same bits as the fused call and the oracle."""
    from oclpathtracer_amd import shim
    from oclpathtracer_amd.render import raycast_reference_loop

    tris, mats = cornell
    dim, frames = 64, 8
    want = np.load(os.path.join(GOLDEN, "cornell_64x64_f8_d16.npy"))
    for batch in (1, 0):
        device.setOption(shim.PT_OPT_BATCH_FRAMES, batch)
        try:
            got = raycast_reference_loop(device, tris, mats, dim, frames)
        finally:
            device.setOption(shim.PT_OPT_BATCH_FRAMES, 1)
        assert_fb_equal(got, want, "reference loop, batch=%d" % batch)


@pytest.mark.parametrize("n_ranks,stripe_rows,H", [(2, 16, 64), (4, 8, 64), (8, 4, 64), (3, 5, 47), (8, 16, 40)])
def test_stripe_sharding_reassembles_bit_exact(device, cornell, n_ranks, stripe_rows, H):
    """Rows dealt to n_ranks in stripes, each rank rendered separately with GLOBAL pixel ids,
    then assembled on the device == the single-device image."""
    from oclpathtracer_amd import adl, shim
    from oclpathtracer_amd.render import Renderer

    tris, mats = cornell
    W, frames = 48, 3
    full = _render_gpu(device, tris, mats, W, H, frames).reshape(H, W, 4)
    lib = shim.load()
    slab_rows = max(lib.pt_local_rows(H, stripe_rows, n_ranks, r) for r in range(n_ranks))
    gathered = np.zeros((n_ranks, slab_rows, W, 4), np.float32)
    total_rows = 0
    for rank in range(n_ranks):
        r = Renderer(device, tris, mats, W, H, n_ranks=n_ranks, rank=rank, stripe_rows=stripe_rows)
        try:
            r.render(frames)
            loc = r.read().reshape(r.local_rows, W, 4)
            rows = r.global_rows()
        finally:
            r.release()
        assert len(rows) == loc.shape[0]
        total_rows += len(rows)
        assert_fb_equal(loc, full[rows], "rank %d rows" % rank)
        gathered[rank, : loc.shape[0]] = loc
    assert total_rows == H
    gbuf = adl.Buffer(device, gathered.size // 4, adl.float4)
    ibuf = adl.Buffer(device, W * H, adl.float4)
    try:
        gbuf.write(gathered, gathered.size // 4)
        shim.check(lib.pt_assemble_stripes(device._h, gbuf._h, ibuf._h, W, H, stripe_rows, n_ranks, slab_rows, None))
        out = np.empty((H * W, 4), np.float32)
        ibuf.read(out, W * H)
        device.waitForCompletion()
    finally:
        gbuf.release()
        ibuf.release()
    assert_fb_equal(out, full, "assembled image")


def test_soup_scene_runtime_triangle_count(device, oracle):
    """Runtime N_tri / N_mat (BASELINE configs[4] shape at a size the oracle finishes quickly)."""
    from oclpathtracer_amd import scene

    tris, mats = scene.make_soup(2000)
    W, H, frames = 48, 32, 2
    want = oracle.render(tris, mats, W, H, frames)
    got = _render_gpu(device, tris, mats, W, H, frames)
    assert_fb_equal(got, want, "soup 2000")


def test_tonemap_matches_host_f2c(device, cornell):
    from oclpathtracer_amd import adl, scene, shim

    tris, mats = cornell
    fb = _render_gpu(device, tris, mats, 64, 64, 4)
    fb[3, 0] = np.nan
    fb[4, 1] = np.inf
    fb[5, 2] = 0.0
    src = adl.Buffer(device, 64 * 64, adl.float4)
    dst = adl.Buffer(device, 64 * 64 * 3, np.int32)
    try:
        src.write(fb, 64 * 64)
        shim.check(shim.load().pt_tonemap_ppm(device._h, src._h, dst._h, 64 * 64, None))
        out = np.empty(64 * 64 * 3, np.int32)
        dst.read(out, out.size)
        device.waitForCompletion()
    finally:
        src.release()
        dst.release()
    # test/RaytraceTest.cpp:78-83 and :280-285 restated HERE (not the product's scene.f2c): `a *= 255; return min((int)a, 255)`
    # of sqrtf(v); (int) of a float on the reference's x86 host is cvttss2si: NaN and out-of-range values give INT_MIN
    with np.errstate(invalid="ignore", over="ignore"):
        a = np.sqrt(fb[:, :3].astype(np.float32)) * np.float32(255.0)
    trunc = np.where(np.isfinite(a) & (np.abs(a) < 2147483648.0), np.trunc(np.where(np.isfinite(a), a, 0.0)), -2147483648.0).astype(np.int64)
    want = np.minimum(trunc, 255).astype(np.int32).reshape(-1)
    assert np.array_equal(out, want)
    assert np.array_equal(want, scene.f2c(fb[:, :3]).reshape(-1))   # ... and the product's host-side f2c agrees with it


def _variant_scene(kind):
    """Scenes that steer the shim into each trace-kernel specialisation."""
    from oclpathtracer_amd import scene

    tris, mats = scene.load_model()
    tris = tris.copy()
    if kind == "quads_scaled":      # still (2k, 2k+1) quads, other numbers: quad filter
        for f in ("p1", "p2", "p3"):
            tris[f][:, :3] = tris[f][:, :3] * np.float32(0.73) + np.array([0.11, 0.4, -0.2], np.float32)
    elif kind == "pairs_broken":    # same triangles, rotated by one: no pair is a quad: generic filter
        tris = np.roll(tris, 1)
    elif kind == "odd_count":       # 35 triangles
        tris = tris[:35].copy()
    elif kind == "huge_extent":     # |e1||e2| > 2e19: exact-division kernel (DET_BOUNDED = false)
        for f in ("p1", "p2", "p3"):
            tris[f][:, :3] = tris[f][:, :3] * np.float32(3.0e10)
    elif kind == "one_triangle":
        tris = tris[2:3].copy()
    elif kind == "degenerate":      # zero-area and NaN triangles among the real ones
        tris[4]["p2"] = tris[4]["p1"]
        tris[7]["p3"][:3] = np.nan
    elif kind == "quads_skewed":    # (a,b,c),(c,d,a) pairs far from parallelograms: shared-u filter, wide margins
        rng = np.random.default_rng(7)
        tris["p2"][1::2, :3] += rng.uniform(-0.4, 0.4, (len(tris) // 2, 3)).astype(np.float32)
    elif kind == "quads_tiny":      # the box shrunk to 5 cm in front of the eye: shared-u margins at their floor
        eye = np.array([0.0, 2.75, 4.0], np.float32)
        for f in ("p1", "p2", "p3"):
            tris[f][:, :3] = (tris[f][:, :3] - eye) * np.float32(0.01) + eye + np.array([0.0, 0.0, -0.05], np.float32)
    elif kind == "quads_detached":  # second triangles translated: e2' == -e2 still, p1' != p3: pair filter only
        for f in ("p1", "p2", "p3"):
            tris[f][1::2, :3] += np.array([0.25, -0.125, 0.5], np.float32)
    elif kind == "quads_nan_second":  # a NaN in the second triangle's e1 only: the pair structure survives
        tris["p2"][9, 1] = np.nan
    elif kind == "quads_17":        # an odd number of quads: the packed filter's last table entry is half padding
        tris = tris[:34].copy()
    elif kind == "quads_2":         # one pair of quads only
        tris = tris[4:8].copy()
    elif kind == "quads_72tri":     # three 32-triangle chunks: the box plus a shrunk copy of itself inside it
        inner = tris.copy()
        for f in ("p1", "p2", "p3"):
            inner[f][:, :3] = inner[f][:, :3] * np.float32(0.4) + np.array([0.3, 1.2, -1.9], np.float32)
        tris = np.concatenate([tris, inner])
    elif kind == "quads_far":       # scene far from the eye relative to its size: large radius, small triangles
        for f in ("p1", "p2", "p3"):
            tris[f][:, :3] = tris[f][:, :3] * np.float32(4.0) + np.array([0.0, -8.25, -160.0], np.float32)
    return tris, mats


@pytest.mark.parametrize("kind", ["quads_scaled", "pairs_broken", "odd_count", "huge_extent", "one_triangle", "degenerate",
                                  "quads_skewed", "quads_tiny", "quads_detached", "quads_nan_second", "quads_far",
                                  "quads_17", "quads_2", "quads_72tri"])
@pytest.mark.parametrize("quad_filter", [0, 1])
def test_kernel_specialisations_match_oracle(device, oracle, kind, quad_filter):
    """quad_filter = PT_OPT_QUAD_FILTER: 0 = the strongest pass-1 filter the scene allows (packed shared-u for
    quad scenes), 1 = independent triangles."""
    from oclpathtracer_amd import shim

    tris, mats = _variant_scene(kind)
    W, H, frames = 64, 48, 3
    want, st = oracle.render(tris, mats, W, H, frames, want_stats=True)
    from oclpathtracer_amd.render import Renderer

    device.setOption(shim.PT_OPT_QUAD_FILTER, quad_filter)
    r = Renderer(device, tris, mats, W, H, want_stats=True)
    try:
        r.render(frames)
        got = r.read()
        gst = r.read_stats()
    finally:
        r.release()
        device.setOption(shim.PT_OPT_QUAD_FILTER, 0)
    assert_fb_equal(got, want, kind)
    assert gst["rays"] == st["rays"]


@pytest.mark.parametrize("copies", [8, 13])
def test_tiled_brute_force_between_the_lds_table_and_the_lbvh(device, oracle, cornell, copies):
    """Scenes of 257 ... 511 triangles are searched by brute force with the records of the current 32-triangle chunk
    streamed through a per-wave LDS tile (north_star's "LDS staging of triangle tiles"; the reference's loop is
    GenerateColors.cl:137-154 with a runtime count).  Nested, shrunk copies of the Cornell box make MANY lanes hold
    survivors in most chunks, so the tile path (not just the tail) does the work: 288 and 468 triangles."""
    from oclpathtracer_amd.render import Renderer

    tris, mats = cornell
    parts = []
    for c in range(copies):
        t = tris.copy()
        k = np.float32(1.0 - 0.06 * c)
        for f in ("p1", "p2", "p3"):
            t[f][:, :3] = t[f][:, :3] * k + np.array([0.0, 2.7, -2.8], np.float32) * (np.float32(1.0) - k)
        parts.append(t)
    big = np.concatenate(parts)
    assert 256 < len(big) < 512
    W, H, frames = 64, 48, 3
    want, st = oracle.render(big, mats, W, H, frames, want_stats=True)
    r = Renderer(device, big, mats, W, H, want_stats=True)
    try:
        r.render(frames)
        got, gst = r.read(), r.read_stats()
    finally:
        r.release()
    assert_fb_equal(got, want, "tiled brute force, %d triangles" % len(big))
    assert gst["rays"] == st["rays"]


@pytest.mark.gpu
@pytest.mark.parametrize("kind,ntri", [("cornell", 36), ("soup", 300), ("soup", 2000), ("soup", 20000), ("degenerate", 36),
                                        ("pairs_broken", 36), ("quads_far", 36)])
def test_bvh_matches_oracle(device, oracle, kind, ntri):
    """PT_OPT_ACCEL = 2: the LBVH closest-hit search against the brute-force oracle.
    Asserted bit for bit -- what the order-free argmin (t, index) argument gives whenever no accepted hit
    lies outside its triangle's grown box; north_star's tolerance (RMS 1e-4) is the documented bar for
    the near-parallel rays that argument cannot cover (csrc/pt_bvh.hip)."""
    from oclpathtracer_amd import scene, shim
    from oclpathtracer_amd.render import Renderer

    if kind == "cornell":
        tris, mats = scene.load_model()
    elif kind == "soup":
        tris, mats = scene.make_soup(ntri)
    else:
        tris, mats = _variant_scene(kind)
    W, H, frames = (64, 48, 3) if ntri <= 2000 else (48, 32, 2)
    want, st = oracle.render(tris, mats, W, H, frames, want_stats=True)
    device.setOption(shim.PT_OPT_ACCEL, 2)
    r = Renderer(device, tris, mats, W, H, want_stats=True)
    try:
        r.render(frames)
        got = r.read()
        gst = r.read_stats()
    finally:
        r.release()
        device.setOption(shim.PT_OPT_ACCEL, 0)
    assert rms_diff(got, want) <= 1e-4
    assert_fb_equal(got, want, "bvh %s %d" % (kind, ntri))
    assert gst["rays"] == st["rays"]


def _bvh_edge_scene(kind):
    """Scenes that steer the LBVH builder into its corner cases (csrc/pt_bvh.hip: radix tree, eight-child collapse, big
    triangles outside the tree)."""
    from oclpathtracer_amd import scene

    box, mats = scene.load_model()
    rng = np.random.default_rng(99)

    def small(n, centres, size=0.05):
        t = np.zeros(n, scene.TRIANGLE_DTYPE)
        c = centres.astype(np.float32)
        t["p1"][:, :3] = c
        t["p2"][:, :3] = c + rng.uniform(-size, size, (n, 3)).astype(np.float32)
        t["p3"][:, :3] = c + rng.uniform(-size, size, (n, 3)).astype(np.float32)
        t["id"] = rng.integers(0, len(mats), n)
        return t

    lo, span = np.array([-2.5, 0.2, -5.2]), np.array([5.0, 5.0, 5.0])
    if kind == "two":            # the smallest hierarchy: one node, two leaves
        return box[20:22].copy(), mats
    if kind == "three":
        return box[20:23].copy(), mats
    if kind == "nine":           # one more leaf than a node holds
        return small(9, lo + span * rng.random((9, 3)), 0.8), mats
    if kind == "duplicates":     # 300 copies of one triangle (equal Morton codes: the tree splits on the index bits) in the box
        t = small(1, (lo + span * 0.5)[None, :], 0.6)
        return np.concatenate([box, np.repeat(t, 300)]), mats
    if kind == "clustered":      # centres at 1 - 2^-k along the diagonal: every radix split peels one leaf off, a deep chain
        k = np.arange(1, 25)
        c = lo[None, :] + span[None, :] * (1.0 - 2.0 ** -k)[:, None]
        return np.concatenate([box, small(24, c, 0.02), small(400, lo + span * rng.random((400, 3)))]), mats
    if kind == "many_big":       # more big triangles than the brute-force table holds (64): they all stay in the tree
        return np.concatenate([box, small(90, lo + span * rng.random((90, 3)), 2.5), small(500, lo + span * rng.random((500, 3)))]), mats
    if kind == "flat":           # every centre in one plane: one Morton axis carries no information
        c = lo + span * rng.random((700, 3))
        c[:, 1] = 2.0
        return np.concatenate([box, small(700, c)]), mats
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["two", "three", "nine", "duplicates", "clustered", "many_big", "flat"])
def test_bvh_builder_corner_cases_match_oracle(device, oracle, kind):
    """PT_OPT_ACCEL = 2 on scenes chosen for the builder, bit for bit against the brute-force oracle."""
    from oclpathtracer_amd import shim
    from oclpathtracer_amd.render import Renderer

    tris, mats = _bvh_edge_scene(kind)
    W, H, frames = 48, 40, 3
    want, st = oracle.render(tris, mats, W, H, frames, want_stats=True)
    device.setOption(shim.PT_OPT_ACCEL, 2)
    r = Renderer(device, tris, mats, W, H, want_stats=True)
    try:
        r.render(frames)
        got, gst = r.read(), r.read_stats()
    finally:
        r.release()
        device.setOption(shim.PT_OPT_ACCEL, 0)
    assert_fb_equal(got, want, "bvh corner case %s" % kind)
    assert gst["rays"] == st["rays"]


def test_bvh_matches_gpu_brute_force_200k_triangles(device):
    """A soup too large for the CPU oracle: the LBVH against the brute-force kernel on the same GPU."""
    from oclpathtracer_amd import scene, shim
    from oclpathtracer_amd.render import Renderer

    tris, mats = scene.make_soup(200_000)
    W, H, frames = 64, 32, 2
    out = {}
    for accel in (1, 2):
        device.setOption(shim.PT_OPT_ACCEL, accel)
        r = Renderer(device, tris, mats, W, H, want_stats=True)
        try:
            r.render(frames)
            out[accel] = (r.read(), r.read_stats())
        finally:
            r.release()
            device.setOption(shim.PT_OPT_ACCEL, 0)
    assert_fb_equal(out[2][0], out[1][0], "bvh vs brute force, 200k triangles")
    assert out[2][1]["rays"] == out[1][1]["rays"]


def _erode(mask: np.ndarray) -> np.ndarray:
    """A block stays only if its 3x3 neighbourhood is all inside the mask (keeps regions off silhouettes)."""
    m = np.pad(mask, 1, constant_values=False)
    out = np.ones_like(mask)
    for dr in (0, 1, 2):
        for dc in (0, 1, 2):
            out &= m[dr:dr + mask.shape[0], dc:dc + mask.shape[1]]
    return out


def jpg_regions(ref: np.ndarray) -> dict:
    """Regions of the reference's rendered image in its 64x64 grid of 8x8-pixel blocks, derived from the
    committed block fixture itself by colour class and image position, eroded by one block: flat areas inside
    one surface where a Monte-Carlo + JPEG residual averages out and a SYSTEMATIC restatement error (a biased
    BRDF lobe, a wrong light radiance, a wrong wall albedo) would not.  Saturated blocks (any channel > 250:
    the light and its mirror images in the two near-specular GGX boxes, roughness 0.008) are clipped by the
    8-bit output and carry no signed information; they form their own class and are only checked to stay
    saturated."""
    R, G, B = ref[..., 0], ref[..., 1], ref[..., 2]
    rows = np.arange(ref.shape[0])[:, None] + np.zeros(ref.shape[1], int)[None, :]
    sat = ref.max(axis=2) >= 250.0
    green = (R < 12) & (B < 12) & (G >= 60) & ~sat
    red = (G < 12) & (B < 12) & (R >= 60) & ~sat
    # the two near-mirror GGX boxes (roughness 0.008) where they reflect the open front of the box: the
    # path leaves after one specular bounce, the pixel is f2c(sqrt(gamma(0.45 * albedo * 2))) = (213, 196, 126)
    # with no Monte-Carlo noise at all -- the sharpest pin of the GGX weight f * cos / pdf
    mirror_bg = (np.abs(R - 213) <= 1.5) & (np.abs(G - 196) <= 1.5) & (np.abs(B - 126) <= 1.5)
    other = ~(sat | green | red | mirror_bg)
    regions = {
        "left wall (green, diffuse)": _erode(green),
        "right wall (red, diffuse)": _erode(red),
        "GGX boxes mirroring the background": _erode(mirror_bg),
        "other unsaturated, rows 0-20 (ceiling)": _erode(other & (rows < 21)),
        "other unsaturated, rows 21-41": _erode(other & (rows >= 21) & (rows < 42)),
        "other unsaturated, rows 42-63 (floor, boxes)": _erode(other & (rows >= 42)),
    }
    regions = {k: v for k, v in regions.items() if v.sum() >= 4}
    regions["saturated (clipped by the 8-bit output)"] = _erode(sat)
    return regions


def test_gpu_render_matches_reference_jpg(device, cornell):
    """The HIP path at the reference's own settings -- 512x512, all 10 000 frames of its loop
    (RaytraceTest.cpp:250) -- through the device output stage f2c(sqrt(.)) (RaytraceTest.cpp:78-83,280-285)
    against the 8x8-pixel block means of the reference's rendered image
    (tests/golden/reference_jpg_blocks_64x64.npy, from FinalRendered_Specular.jpg).  The statistical pin of
    the whole path against the real OpenCL output:
      * unsigned: mean |difference| over all blocks, worst block, correlation (Monte-Carlo + JPEG noise);
      * SIGNED mean residual per region (jpg_regions): a systematic error in one surface's shading shows up
        here long before it moves the unsigned mean; every region must be within REGION_TOL of zero."""
    from oclpathtracer_amd import scene
    from oclpathtracer_amd.render import Renderer

    ref = np.load(os.path.join(GOLDEN, "reference_jpg_blocks_64x64.npy")).astype(np.float64)
    tris, mats = cornell
    W, frames, G = 512, 10000, 64
    r = Renderer(device, tris, mats, W, W)
    try:
        r.render(frames)
        fb = r.read()
    finally:
        r.release()
    img = scene.f2c(fb[:, :3]).reshape(W, W, 3).astype(np.float64)
    blocks = img.reshape(G, W // G, G, W // G, 3).mean(axis=(1, 3))
    diff = np.abs(blocks - ref)
    corr = np.corrcoef(blocks.ravel(), ref.ravel())[0, 1]
    lines = ["gpu (10000 frames) vs reference JPG, 64x64 blocks: mean |diff| %.3f max %.2f corr %.6f" % (diff.mean(), diff.max(), corr)]
    worst = 0.0
    for name, mask in jpg_regions(ref).items():
        res = (blocks[mask] - ref[mask]).mean(axis=0)
        lines.append("  %-48s %4d blocks  signed mean residual R %+.3f G %+.3f B %+.3f  (reference level %s)" % (
            name, int(mask.sum()), res[0], res[1], res[2], np.round(ref[mask].mean(axis=0), 1)))
        if name.startswith("saturated"):
            top = ref[mask].argmax(axis=1)   # the channel the reference clips in each of these blocks
            assert np.take_along_axis(blocks[mask], top[:, None], axis=1).min() > 240.0, "a channel the reference saturates is not saturated here"
        else:
            # the large mixed regions carry the diffuse / GI information and average the JPEG's artefacts out;
            # flat saturated-colour regions keep a DC quantisation offset of the JPEG's chroma planes
            # (a channel at 0 cannot ring below 0; a flat (213,196,126) area reconstructs R one level off)
            tol = 0.25 if name.startswith("other") and mask.sum() >= 100 else REGION_TOL
            worst = max(worst, float(np.abs(res).max()) / tol)
    report = "\n".join(lines)
    print(report)
    out_dir = os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "jpg_region_residuals.txt"), "w") as f:
            f.write(report + "\n")
    except OSError:
        pass
    assert diff.mean() < 1.5, diff.mean()
    assert diff.max() < 16.0, diff.max()
    assert corr > 0.9998, corr
    assert worst < 1.0, report


REGION_TOL = 1.25  # of 255; measured: profiles/r02/jpg_region_residuals.txt (<= 0.08 in the large mixed regions, <= 0.94 in flat colour areas)


def _random_quad_scene(seed: int):
    """Random (a,b,c),(c,d,a) quads around the view volume: parallelograms, perturbed parallelograms,
    slivers, huge and tiny ones, some facing away; random diffuse / glossy / emissive materials."""
    from oclpathtracer_amd import scene

    rng = np.random.default_rng(seed)
    nq = int(rng.integers(1, 40))
    tris = np.zeros(2 * nq, scene.TRIANGLE_DTYPE)
    mats = np.zeros(nq, scene.MATERIAL_DTYPE)
    scale = np.float32(10.0 ** rng.uniform(-1.5, 1.5))          # scene size: 0.03 ... 30
    centre = np.array([0.0, 2.75, 4.0], np.float32) + np.array([0.0, 0.0, -1.0], np.float32) * scale * np.float32(1.5)
    for q in range(nq):
        a = centre + rng.uniform(-1, 1, 3).astype(np.float32) * scale
        e1 = rng.uniform(-1, 1, 3).astype(np.float32) * scale * np.float32(10.0 ** rng.uniform(-1.5, 0.5))
        e2 = rng.uniform(-1, 1, 3).astype(np.float32) * scale * np.float32(10.0 ** rng.uniform(-1.5, 0.5))
        if rng.random() < 0.7 and np.dot(np.cross(e2, e1), a - np.array([0.0, 2.75, 4.0], np.float32)) < 0:
            e1, e2 = e2, e1                                      # most quads face the eye (cull test :100)
        b, c = a + e1, a + e1 + e2
        d = a + e2
        if rng.random() < 0.5:                                   # not a parallelogram
            d = d + rng.uniform(-0.3, 0.3, 3).astype(np.float32) * np.float32(np.abs(e1).max())
        for k, (p1, p2, p3) in enumerate(((a, b, c), (c, d, a))):
            t = tris[2 * q + k]
            t["p1"][:3], t["p2"][:3], t["p3"][:3] = p1, p2, p3
            t["id"] = q
        m = mats[q]
        m["albedo"] = tuple(rng.uniform(0.05, 0.95, 3)) + (1.0,)
        m["emissive"] = ((30.0, 30.0, 30.0, 1.0) if rng.random() < 0.15 else (0.0, 0.0, 0.0, 1.0))
        m["type"] = scene.SPECULAR if rng.random() < 0.3 else scene.DIFFUSE
        m["roughness"] = np.float32(10.0 ** rng.uniform(-2.5, -0.3)) if m["type"] == scene.SPECULAR else 0.0
    return tris, mats


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_quad_scenes_match_oracle(device, oracle, seed):
    """Fuzz of the scene-dependent machinery: whatever filter mode, slack and kernel variant the shim
    picks for a random quad scene (auto), and with the LBVH forced, pixels and ray counts equal the
    brute-force oracle's."""
    from oclpathtracer_amd import shim
    from oclpathtracer_amd.render import Renderer

    tris, mats = _random_quad_scene(1000 + seed)
    W, H, frames = 48, 40, 3
    want, st = oracle.render(tris, mats, W, H, frames, want_stats=True)
    for accel in (0, 2) if len(tris) >= 2 else (0,):
        device.setOption(shim.PT_OPT_ACCEL, accel)
        r = Renderer(device, tris, mats, W, H, want_stats=True)
        try:
            r.render(frames)
            got = r.read()
            gst = r.read_stats()
        finally:
            r.release()
            device.setOption(shim.PT_OPT_ACCEL, 0)
        assert_fb_equal(got, want, "random quads seed %d accel %d (%d triangles)" % (seed, accel, len(tris)))
        assert gst["rays"] == st["rays"]


def test_configs3_one_rank_of_eight_full_size(device, cornell, oracle):
    """BASELINE configs[3] at FULL size for one rank: 2048x2048 dealt to 8 ranks in 16-row stripes; rank 5's
    share (256 rows, 524 288 pixels) rendered for all 1024 frames (537 M samples), 16 runs of 64 of its
    pixels recomputed by the oracle through all 1024 frames from their GLOBAL ids (seed parity across
    the split, GenerateColors.cl:305-308).  Whole-share properties: w == 1, nothing negative, the
    sample counter equals pixels x frames."""
    from oclpathtracer_amd.render import Renderer

    tris, mats = cornell
    W = H = 2048
    frames, n_ranks, rank, stripe = 1024, 8, 5, 16
    r = Renderer(device, tris, mats, W, H, n_ranks=n_ranks, rank=rank, stripe_rows=stripe, want_stats=True)
    try:
        r.render(frames)
        got = r.read()
        st = r.read_stats()
        rows = r.global_rows()
    finally:
        r.release()
    assert len(rows) == H // n_ranks and got.shape == (len(rows) * W, 4)
    assert st["samples"] == len(rows) * W * frames
    assert np.all(got[:, 3] == 1.0)
    assert not np.any(got[:, :3] < 0)
    rng = np.random.default_rng(3)
    fb = np.zeros((H * W, 4), np.float32)
    for lr in rng.integers(0, len(rows), 16):
        x0 = int(rng.integers(0, W - 64))
        g0 = int(rows[lr]) * W + x0
        oracle.render(tris, mats, W, H, frames, fb=fb, gid_begin=g0, gid_count=64)
        assert_fb_equal(got[lr * W + x0: lr * W + x0 + 64], fb[g0: g0 + 64], "C4 rank %d local row %d" % (rank, lr))


def test_configs4_million_triangle_soup_full_size(device, oracle):
    """BASELINE configs[4] scene at FULL size on one GPU: the 10^6-triangle soup, 1024x1024, 256 spp, depth 16,
    through the LBVH (the default from 512 triangles on).
      * whole image: w == 1, nothing negative, sample counter == pixels x frames;
      * 16 seeded pixels recomputed by the brute-force CPU oracle through all 256 frames (one oracle thread per
        pixel; ~10^6 exact tests per ray): identical bits;
      * LBVH against the GPU's own brute-force search (PT_OPT_ACCEL = 1) on the first 64 rows (65 536 pixels),
        frame 0: identical bits and ray counts."""
    from concurrent.futures import ThreadPoolExecutor

    from oclpathtracer_amd import scene, shim
    from oclpathtracer_amd.render import Renderer

    tris, mats = scene.make_soup(1_000_000)
    W = H = 1024
    frames = 256
    r = Renderer(device, tris, mats, W, H, want_stats=True)
    try:
        r.render(frames)
        got = r.read()
        st = r.read_stats()
    finally:
        r.release()
    assert st["samples"] == W * H * frames
    assert np.all(got[:, 3] == 1.0)
    assert not np.any(got[:, :3] < 0)

    rng = np.random.default_rng(4)
    gids = [int(g) for g in rng.integers(0, W * H, 16)]

    def one(gid):
        fb = np.zeros((H * W, 4), np.float32)
        oracle.render(tris, mats, W, H, frames, fb=fb, gid_begin=gid, gid_count=1, nthreads=1)
        return fb[gid].copy()

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        want = list(ex.map(one, gids))
    for gid, w in zip(gids, want):
        assert_fb_equal(got[gid], w, "1M-triangle soup, LBVH vs oracle, pixel %d" % gid)

    rows_ab = 64
    out = {}
    for accel in (0, 1):
        device.setOption(shim.PT_OPT_ACCEL, accel)
        r = Renderer(device, tris, mats, W, H, n_ranks=H // rows_ab, rank=0, stripe_rows=rows_ab, want_stats=True)
        try:
            r.render(1)
            out[accel] = (r.read(), r.read_stats())
        finally:
            r.release()
            device.setOption(shim.PT_OPT_ACCEL, 0)
    assert out[0][0].shape == (rows_ab * W, 4)
    assert_fb_equal(out[0][0], out[1][0], "1M-triangle soup, LBVH vs brute force, 65 536 pixels")
    assert out[0][1]["rays"] == out[1][1]["rays"]


def test_bvh_exact_ties_go_to_the_lowest_index(device, oracle, cornell):
    """Every Cornell triangle 17 more times, shuffled, behind the originals (648 triangles: LBVH by
    default): each hit is an exact tie in t between 18 copies, which the reference's ascending loop with
    its strict `t < tmax` (GenerateColors.cl:125,145-151) gives to the lowest index.  The copies carry
    OTHER materials than the originals, so a traversal that resolved ties by visiting order would
    change the image."""
    from oclpathtracer_amd import scene, shim
    from oclpathtracer_amd.render import Renderer

    tris, mats = cornell
    rng = np.random.default_rng(5)
    copies = np.concatenate([tris[rng.permutation(len(tris))] for _ in range(17)])
    copies["id"] = (copies["id"] + 7) % len(mats)       # a copy would shade differently
    big = np.concatenate([tris, copies])
    W, H, frames = 64, 48, 3
    want, st = oracle.render(big, mats, W, H, frames, want_stats=True)
    base = oracle.render(tris, mats, W, H, frames)
    assert np.array_equal(want.view(np.uint32), base.view(np.uint32))  # the oracle itself: copies never win
    for accel in (0, 1):
        device.setOption(shim.PT_OPT_ACCEL, accel)
        r = Renderer(device, big, mats, W, H, want_stats=True)
        try:
            r.render(frames)
            got = r.read()
            gst = r.read_stats()
        finally:
            r.release()
            device.setOption(shim.PT_OPT_ACCEL, 0)
        assert_fb_equal(got, want, "tie scene, accel %d" % accel)
        assert gst["rays"] == st["rays"]


@pytest.mark.parametrize("depth", [1, 3, 40])
def test_depth_caps_outside_the_baseline_configs(device, cornell, oracle, depth):
    """max_bounces is a runtime parameter of the fused entry point (the reference compiles BOUNCES = 16,
    GenerateColors.cl:5): one bounce, three, and more than the reference's cap."""
    tris, mats = cornell
    W, H, frames = 64, 40, 4
    want = oracle.render(tris, mats, W, H, frames, max_bounces=depth)
    got = _render_gpu(device, tris, mats, W, H, frames, depth=depth)
    assert_fb_equal(got.reshape(-1, 4), want, "depth %d" % depth)


def test_empty_scene_and_zero_frames(device, cornell, oracle):
    """No triangles at all: every path leaves on its first ray (GenerateColors.cl:233-237).  And a
    render of zero frames is a no-op that leaves the framebuffer alone."""
    from oclpathtracer_amd import scene
    from oclpathtracer_amd.render import Renderer

    tris, mats = cornell
    none = np.zeros(0, scene.TRIANGLE_DTYPE)
    W, H, frames = 48, 32, 3
    want = oracle.render(none, mats, W, H, frames)
    r = Renderer(device, none, mats, W, H)
    try:
        r.render(frames)
        got = r.read()
        r.render(0)
        again = r.read()
    finally:
        r.release()
    assert_fb_equal(got, want, "empty scene")
    assert_fb_equal(again, got, "zero frames")


def test_stripes_with_the_lbvh(device, oracle):
    """configs[4] runs the soup on 8 ranks: image stripes and the LBVH together (every rank builds its
    own hierarchy of the replicated scene).  Rank 1 of 4, 8-row stripes, 3000 triangles."""
    from oclpathtracer_amd import scene
    from oclpathtracer_amd.render import Renderer

    tris, mats = scene.make_soup(3000)
    W, H, frames = 96, 80, 2
    want = oracle.render(tris, mats, W, H, frames).reshape(H, W, 4)
    r = Renderer(device, tris, mats, W, H, n_ranks=4, rank=1, stripe_rows=8)
    try:
        r.render(frames)
        got = r.read().reshape(-1, W, 4)
        rows = r.global_rows()
    finally:
        r.release()
    assert_fb_equal(got.reshape(-1, 4), want[rows].reshape(-1, 4), "stripes + LBVH")
