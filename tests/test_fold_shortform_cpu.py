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
