"""CPU witness of the identity behind the fold kernel's short decode (csrc/pt_kernels.hip, pt_fold_decode).

The reference's running mean (test/ClKernels/GenerateColors.cl:314-321) decodes the pixel it encoded one frame earlier:
m = pow(v, 1/2.2f) rounded to binary32, then o = pow(m, 2.2f).  The kernel replaces the second pow by
    o = fl32(v + v * (2.2f * delta + eta * ln v)),   delta = (m - E) / E,   eta = fl32(1/2.2f) * 2.2f - 1,
(E = the unrounded pow(v, 1/2.2f)) whenever both ends of a +-2^-36 v interval round to the same binary32, and evaluates the
literal pow otherwise.  The GPU tier proves the device code against the literal operation for every binary32 v
(tests/test_gpu_fold_exact.py); this test restates the identity with numpy and the ORACLE's pow on two million operands,
so that the argument can be checked without a GPU."""
import numpy as np


def test_decode_identity_against_the_oracle_pow(oracle):
    rng = np.random.default_rng(1)
    n = 2_000_000
    v = (2.0 ** rng.uniform(-79.9, 79.9, n)).astype(np.float32)
    y1 = float(np.float32(1.0) / np.float32(2.2))
    y2 = float(np.float32(2.2))
    eta_ln2 = np.float32((y1 * y2 - 1.0) * np.log(2.0))
    assert eta_ln2 == np.float32(float.fromhex("-0x1.4f889ep-27"))   # PT_FOLD_ETA_LN2
    m = oracle.pow_array(v, y1)
    want = oracle.pow_array(m, y2)
    E = np.power(v.astype(np.float64), y1)                  # the unrounded encode, to binary64 accuracy
    lf = np.log2(v.astype(np.float64)).astype(np.float32)
    df = (m.astype(np.float64) - E).astype(np.float32)
    c = (eta_ln2 * lf + (np.float32(2.2) * df) * (np.float32(1.0) / m)).astype(np.float32)
    t1 = (v * c).astype(np.float32)
    u = (v * np.float32(2.0 ** -36)).astype(np.float32)
    lo = (v + (t1 - u)).astype(np.float32)
    hi = (v + (t1 + u)).astype(np.float32)
    ok = lo == hi
    assert ok.mean() > 0.999                                # Ziv's test accepts all but ~0.04 %
    assert np.array_equal(lo[ok].view(np.uint32), want[ok].view(np.uint32))
    # the decoded value is the encoded one moved by a few ulps (eta is not zero: it is NOT the identity)
    ulps = (want.astype(np.float64) - v) / np.spacing(v)
    assert np.abs(ulps).max() <= 16 and (ulps != 0).mean() > 0.5


def _rn32(x):
    """Fraction -> nearest binary32 (ties to even), exactly; normal range only."""
    from fractions import Fraction

    if x == 0:
        return Fraction(0)
    s = -1 if x < 0 else 1
    x = abs(x)
    e = x.numerator.bit_length() - x.denominator.bit_length()
    if Fraction(2) ** e > x:
        e -= 1
    assert Fraction(2) ** e <= x < Fraction(2) ** (e + 1) and -126 <= e <= 127
    ulp = Fraction(2) ** (e - 23)
    m = x / ulp
    k = m.numerator // m.denominator
    rem = m - k
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (k & 1)):
        k += 1
    return s * k * ulp


def test_markstein_quotient_in_exact_rational_arithmetic():
    """q0 = RN(a y), r = RN(a - b q0), q = RN(q0 + r y) with y = RN(1/b) is RN(a/b): the three-instruction quotient of
    csrc/pt_device_math.h (pt_div_markstein), replayed with exact rationals and an exact round-to-nearest-even on 20 000 random
    significand pairs plus pairs built to sit next to rounding boundaries.  (The GPU tier runs ALL 2^46 pairs: tools/div_exhaustive.py.)
    The textbook statement of Markstein's theorem asks for a FAITHFUL q0, which makes the residual exact; RN(a RN(1/b)) can be
    1.5 ulp off when a < b, and then the residual needs 25 bits and the fma rounds it (about one pair in 700) -- the quotient comes
    out correctly rounded all the same, which is why the claim rests on the exhaustive run and not on the theorem."""
    from fractions import Fraction

    rng = np.random.default_rng(7)
    a_m = rng.integers(1 << 23, 1 << 24, 20_000)
    b_m = rng.integers(1 << 23, 1 << 24, 20_000)
    # adversarial: b with long runs of ones / near powers of two, a = b * k +- 1 (quotients next to representable numbers)
    extra_b = np.array([(1 << 24) - 1, (1 << 24) - 2, (1 << 23) + 1, (1 << 23), 0xAAAAAB, 0xFFFFF1, 0xC00001, 0x800003] * 8)
    extra_a = np.array([(int(b) * k >> 1 | 1 << 23) % (1 << 24) | 1 << 23 for b, k in zip(extra_b, range(3, 3 + len(extra_b)))])
    a_m = np.concatenate([a_m, extra_a, extra_b])
    b_m = np.concatenate([b_m, extra_b, extra_a])
    inexact_r = 0
    for am, bm in zip(a_m.tolist(), b_m.tolist()):
        a, b = Fraction(am, 1 << 23), Fraction(bm, 1 << 23)
        y = _rn32(1 / b)
        q0 = _rn32(a * y)
        r_exact = a - b * q0
        r = _rn32(r_exact) if r_exact != 0 else Fraction(0)
        inexact_r += r != r_exact
        q = _rn32(q0 + r * y)
        assert q == _rn32(a / b), (am, bm)
    assert 0 < inexact_r < 100   # the residual is NOT always representable -- and the quotient is right regardless
