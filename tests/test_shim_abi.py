"""The C-ABI library loads and exports every symbol include/pt_shim.h declares (no GPU, no
compute calls), and the product has no import path into oracle/."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "pt_shim.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from oclpathtracer_amd import shim

    names = _declared_functions()
    assert len(names) >= 40
    lib = ctypes.CDLL(shim.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libptshim.so does not export %s" % n
    assert sorted(shim.SIGNATURES) == names, "shim.py binding and include/pt_shim.h disagree"
    assert shim.load().pt_abi_version() == shim.PT_SHIM_ABI_VERSION == 2
    hdr = open(os.path.join(ROOT, "include", "pt_shim.h")).read()
    assert re.search(r"#define\s+PT_SHIM_ABI_VERSION\s+2\b", hdr)


def test_struct_layouts_match_header():
    from oclpathtracer_amd import shim

    assert ctypes.sizeof(shim.LaunchArg) == 4 + 4 + 8 + 8 + 64
    assert ctypes.sizeof(shim.RenderParams) == 16 * 4


def test_no_gpu_means_loud_failure_not_fallback():
    """Host-only calls work; anything needing the device reports an error code (never a CPU path)."""
    from oclpathtracer_amd import adl, shim

    lib = shim.load()
    assert lib.pt_local_rows(64, 16, 2, 1) == 32 and lib.pt_local_rows(40, 16, 8, 2) == 8
    assert lib.pt_local_rows(40, 16, 8, 3) == 0 and lib.pt_local_rows(47, 5, 3, 0) == 17
    assert lib.pt_local_rows(10, 0, 1, 0) == -1 and lib.pt_local_rows(10, 1, 2, 2) == -1
    if lib.pt_device_count() == 0:
        assert adl.init(adl.TYPE_HIP) is False
        out = ctypes.c_void_p()
        rc = lib.pt_device_create(0, ctypes.byref(out))
        assert rc == shim.PT_ERR_NO_DEVICE and not out.value
        assert b"device" in lib.pt_last_error().lower()
        with pytest.raises(shim.ShimError):
            adl.DeviceUtils.allocate(adl.TYPE_HIP)
    assert adl.DeviceUtils.allocate(adl.TYPE_HOST) is None  # unknown backend -> 0 (Adl/Adl.cpp:188-189)
    assert adl.init(adl.TYPE_HOST) is False


def test_product_does_not_reach_into_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "oclpathtracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "ptoracle" not in text and "libptoracle" not in text, fn  # python binding / .so name
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), fn
                assert not re.search(r"#\s*include\s*[\"<][^\">]*(oracle|ptor_)", text), fn
                assert not re.search(r"-[IL]\S*oracle|-lptoracle", text), fn
    hdr = open(os.path.join(ROOT, "include", "pt_shim.h")).read()
    assert "oracle" not in hdr.lower()
