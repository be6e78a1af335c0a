import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure, oracle/pt_oracle.c), built on demand with gcc."""
    from oracle import ptoracle

    ptoracle.build()
    ptoracle.lib()
    return ptoracle


@pytest.fixture(scope="session")
def cornell():
    from oclpathtracer_amd import scene

    return scene.load_model()


@pytest.fixture(scope="session")
def device():
    """One HIP shim device for the whole GPU session (fails loudly without a GPU)."""
    try:
        import torch  # noqa: F401  (first: tests that hand torch tensors to the shim need ONE HIP runtime in the process)
    except ImportError:
        pass
    from oclpathtracer_amd import adl

    assert adl.init(adl.TYPE_HIP), "adl.init failed: no usable MI355X / libptshim.so"
    dev = adl.DeviceUtils.allocate(adl.TYPE_HIP, adl.Config(0))
    yield dev
    adl.DeviceUtils.deallocate(dev)
    adl.quit(adl.TYPE_HIP)


def assert_fb_equal(got: np.ndarray, want: np.ndarray, what: str = ""):
    """Bit-exact framebuffer comparison with NaNs canonicalised (NaN payload/sign differ between
    x86 and gfx950; SURVEY.md Appendix B 'non-finite pixels')."""
    got = np.asarray(got, np.float32).reshape(-1)
    want = np.asarray(want, np.float32).reshape(-1)
    assert got.shape == want.shape, what
    gn, wn = np.isnan(got), np.isnan(want)
    assert np.array_equal(gn, wn), "%s: NaN masks differ at %d entries" % (what, int((gn != wn).sum()))
    g = got.view(np.uint32)[~gn]
    w = want.view(np.uint32)[~wn]
    if not np.array_equal(g, w):
        bad = np.flatnonzero(g != w)
        fin = np.isfinite(got[~gn]) & np.isfinite(want[~wn])
        diff = np.abs(got[~gn][fin].astype(np.float64) - want[~wn][fin].astype(np.float64))
        rms = float(np.sqrt(np.mean(diff * diff))) if diff.size else 0.0
        raise AssertionError("%s: %d of %d values differ bitwise (first at %d: got %r want %r); rms %.3e"
                             % (what, bad.size, g.size, int(bad[0]), got[~gn][bad[0]], want[~wn][bad[0]], rms))


def rms_diff(got: np.ndarray, want: np.ndarray) -> float:
    got = np.asarray(got, np.float64).reshape(-1)
    want = np.asarray(want, np.float64).reshape(-1)
    fin = np.isfinite(got) & np.isfinite(want)
    d = got[fin] - want[fin]
    return float(np.sqrt(np.mean(d * d))) if d.size else 0.0
