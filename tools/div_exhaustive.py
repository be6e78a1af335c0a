#!/usr/bin/env python3
"""Markstein's three-instruction quotient against IEEE division for EVERY pair of binary32 significands (GPU).

    y = RN(1/b) (pt_rcp_fast);  q0 = a * y;  r = fma(-b, q0, a);  q = fma(r, y, q0)          (csrc/pt_kernels.hip, pt_fold_div)

While no operand or intermediate leaves the normal range, the quotient's significand depends on the operands'
significands alone: 2^23 x 2^23 pairs decide the matter for every operand the callers' range guards let through.
Usage: python tools/div_exhaustive.py [first_chunk [n_chunks]]   (128 chunks of 2^16 divisors; all of them ~ 1 minute)
"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch  # noqa: F401  (first: the shim binds to the HIP runtime torch loaded)

from oclpathtracer_amd import adl  # noqa: E402


def main():
    c0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    nc = int(sys.argv[2]) if len(sys.argv) > 2 else 128 - c0
    assert adl.init(adl.TYPE_HIP)
    dev = adl.DeviceUtils.allocate(adl.TYPE_HIP, adl.Config(0))
    k = dev.getKernel("PtShimTest", "FoldCheckKernel")
    out = adl.Buffer(dev, 6, np.uint64)
    out.write(np.zeros(6, np.uint64), 6)
    t0 = time.time()
    res = np.zeros(6, np.uint64)
    for c in range(c0, c0 + nc):
        la = adl.Launcher(dev, k)
        la.setBuffers([adl.BufferInfo(out)])
        la.setConst(np.int32(4))
        la.setConst(np.uint32(c << 16))
        la.setConst(np.uint64(1 << 16))
        la.launch1D(1)
        out.read(res, 6)
        dev.waitForCompletion()
        if c % 8 == 7 or c == c0 + nc - 1:
            print("divisor significands %#08x .. %#08x: %d divisors x 2^23 numerators checked so far, %d quotients differ from IEEE division (%.0f s)"
                  % (c0 << 16, ((c + 1) << 16) - 1, int(res[4]), int(res[3]), time.time() - t0), flush=True)
    out.release()
    adl.DeviceUtils.deallocate(dev)
    print("TOTAL: %d x 2^23 = %.4g significand pairs, %d mismatches" % (int(res[4]), float(res[4]) * 2.0**23, int(res[3])))
    return 0 if int(res[3]) == 0 else 1


if __name__ == "__main__":
    raise SystemExit(main())
