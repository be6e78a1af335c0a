"""The fold kernel's short forms return the bits of the literal operations (GPU tier).

`pt_fold_kernel` replays the reference's gamma running mean (test/ClKernels/GenerateColors.cl:314-321)
    o = pow(m, 2.2f);  m = pow((o * (z - 1) + c) / z, 1.0f / 2.2f)
with three short forms (csrc/pt_kernels.hip, "fold kernel"): the encode without special-case tests, the decode
from the previous encode's binary64 value with Ziv's rounding test, and Markstein's three-instruction quotient.
Each is compared here, ON THE GPU and over EVERY operand it can meet, with the literal PTSPEC operation -- which
`test_device_transcendentals_match_oracle_bitwise` pins to the CPU oracle:

  * encode: all 2^32 binary32 patterns (the regular ones are compared, the others take the literal pow by construction);
  * decode: all 2^32 patterns again, through the encode -> decode chain the kernel runs;
  * division: every frame number 1 .. 2047 x every 23-bit significand x three exponents (the ends and the middle
    of the range in which the short form is used);
  * whole chains of 32 frames on arbitrary radiance (zeros, huge, tiny, subnormal, negative, infinite, NaN), fresh
    and resumed: 2^22 chains.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(device, mode, first, count):
    from oclpathtracer_amd import adl

    k = device.getKernel("PtShimTest", "FoldCheckKernel")
    assert k is not None
    out = adl.Buffer(device, 6, np.uint64)
    try:
        out.write(np.zeros(6, np.uint64), 6)
        launcher = adl.Launcher(device, k)
        launcher.setBuffers([adl.BufferInfo(out)])
        launcher.setConst(np.int32(mode))
        launcher.setConst(np.uint32(first))
        launcher.setConst(np.uint64(count))
        launcher.launch1D(1)
        res = np.empty(6, np.uint64)
        out.read(res, 6)
        device.waitForCompletion()
    finally:
        out.release()
    return [int(v) for v in res]


def test_encode_equals_the_literal_pow_for_every_binary32(device):
    bad, _, _, _, seen, _ = _check(device, 0, 0, 1 << 32)
    # the regular range [2^-80, 2^80): 160 binades of 2^23 significands
    assert seen == 160 << 23
    assert bad == 0


def test_decode_equals_the_literal_pow_of_the_encoded_value_for_every_binary32(device):
    _, bad, slow, _, seen, _ = _check(device, 1, 0, 1 << 32)
    assert seen == 160 << 23
    assert bad == 0
    print("decode: %d of %d operands failed the rounding test and took the literal pow (%.4f %%)" % (slow, seen, 100.0 * slow / seen))
    assert slow < seen // 1000


def test_quotient_equals_ieee_division_for_every_frame_number_and_significand(device):
    _, _, _, bad, seen, _ = _check(device, 2, 1, 2047 << 23)
    assert seen == 3 * (2047 << 23)
    assert bad == 0


def test_chains_on_arbitrary_radiance_equal_the_literal_fold(device):
    res = _check(device, 3, 20261004, 1 << 22)
    assert res[4] == 32 << 22
    assert res[5] == 0


@pytest.mark.parametrize("case", ["tiny", "huge", "mixed", "nonfinite"])
def test_fold_on_extreme_radiance_matches_the_oracle(device, oracle, cornell, case):
    """The same through the product path and against the CPU oracle: emitters whose radiance puts the running mean outside
    the range of the short forms (below 2^-80, above 2^80, negative sums clamped at :260, infinities and NaN), rendered
    from frame 0 and resumed from pixels holding arbitrary values."""
    from conftest import assert_fb_equal
    from test_gpu_parity import _render_gpu

    tris, mats = cornell
    mats = mats.copy()
    em = {"tiny": [3e-27, 1e-33, 2e-25, 0.0, 4e-38, 1e-30],
          "huge": [3e25, 1e30, 2e24, 7e28, 0.0, 1e33],
          "mixed": [1e-27, 2e26, 0.5, -3.0, 0.0, 1e-40],
          "nonfinite": [np.inf, 1.0, np.nan, -np.inf, 0.0, 2.0]}[case]
    for i in range(len(mats)):
        e = np.float32(em[i % len(em)])
        mats[i]["emissive"] = (e, e * np.float32(0.5), e * np.float32(2.0), 1.0)
    W, H, frames = 48, 32, 6
    with np.errstate(all="ignore"):
        want = oracle.render(tris, mats, W, H, frames)
        got = _render_gpu(device, tris, mats, W, H, frames)
        assert_fb_equal(got, want, "extreme radiance (%s), from frame 0" % case)
        rng = np.random.default_rng(3)
        junk = (2.0 ** rng.uniform(-140, 120, (W * H, 4))).astype(np.float32)
        junk[rng.random((W * H, 4)) < 0.1] = 0.0
        junk[7, 1], junk[9, 2], junk[11, 0], junk[13, 1] = np.inf, np.nan, -1.5, 1e-45
        want = oracle.render(tris, mats, W, H, 4, frame_begin=3, fb=junk.copy())
        got = _render_gpu(device, tris, mats, W, H, 4, frame_begin=3, fb_init=junk)
        assert_fb_equal(got, want, "extreme radiance (%s), resumed at frame 3" % case)
    if case in ("tiny", "huge"):
        # the scene must really drive the mean out of the regular range
        fin = got[:, :3][np.isfinite(got[:, :3]) & (got[:, :3] > 0)]
        enc = fin.astype(np.float64) ** 2.2
        assert ((enc < 2.0 ** -80) | (enc > 2.0 ** 80)).mean() > 0.2


@pytest.mark.parametrize("case", ["albedo", "roughness"])
def test_throughput_quotients_outside_the_short_forms_range_match_the_oracle(device, oracle, cornell, case):
    """color * dot / pdf (:253-255) goes through one exact reciprocal and Markstein's correction while the three numerators
    and the pdf are in a guarded range (csrc/pt_device_math.h, pt_div3), through the generic division otherwise: materials
    whose albedo (numerators: tiny, huge, zero, negative, NaN) or roughness (pdf: 1e13 and beyond, 0/0) leave that range."""
    from conftest import assert_fb_equal
    from oclpathtracer_amd import scene
    from test_gpu_parity import _render_gpu

    tris, mats = cornell
    mats = mats.copy()
    if case == "albedo":
        vals = [1e-25, 3e22, 0.0, -0.5, 1e-38, np.nan, 0.7, 2e19]
        for i in range(len(mats)):
            a = np.float32(vals[i % len(vals)])
            mats[i]["albedo"] = (a, a * np.float32(3.0), np.float32(0.6), 1.0)
    else:
        vals = [1e-7, 3e-6, 0.0, 1e-9, 0.02, 1e-5]
        for i in range(len(mats)):
            mats[i]["type"] = scene.SPECULAR
            mats[i]["roughness"] = np.float32(vals[i % len(vals)])
    W, H, frames = 64, 40, 4
    with np.errstate(all="ignore"):
        want, st = oracle.render(tris, mats, W, H, frames, want_stats=True)
        got = _render_gpu(device, tris, mats, W, H, frames)
    assert_fb_equal(got, want, "quotients outside the guarded range (%s)" % case)
    assert st["rays"] > 2 * W * H * frames     # paths do continue past the first hit
