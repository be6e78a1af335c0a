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
