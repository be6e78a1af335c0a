"""ctypes binding of the CPU oracle (oracle/libptoracle.so).

TEST INFRASTRUCTURE -- imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (oclpathtracer_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libptoracle.so")

STATS_FIELDS = (
    "samples", "rays", "tests", "cull", "rej_u", "rej_v", "reach_t", "accept",
    "shade_diffuse", "shade_specular", "miss", "term_pdf", "term_depth",
)


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (gcc only)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(os.path.join(_HERE, f)) for f in ("pt_oracle.c", "ptor_constants.h", "Makefile")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None
_libs = {}


def lib(variant: str = ""):
    """The oracle ("") or the same restatement under another arithmetic convention ("nofma": dot / cross without fused
    multiply-add; tests only -- see oracle/Makefile)."""
    global _lib
    if variant:
        if variant not in _libs:
            path = os.path.join(_HERE, "libptoracle_%s.so" % variant)
            if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "pt_oracle.c")):
                subprocess.check_call(["make", "-C", _HERE, "-s", os.path.basename(path)])
            L = ctypes.CDLL(path)
            _bind(L)
            _libs[variant] = L
        return _libs[variant]
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        _bind(L)
        _lib = L
    return _lib


def _bind(L):
    if True:
        L.ptor_kat_hash.restype = ctypes.c_uint32
        L.ptor_kat_hash.argtypes = [ctypes.c_uint32]
        L.ptor_kat_random.restype = ctypes.c_float
        L.ptor_kat_random.argtypes = [ctypes.POINTER(ctypes.c_uint32)]
        L.ptor_kat_pow.restype = ctypes.c_float
        L.ptor_kat_pow.argtypes = [ctypes.c_float, ctypes.c_float]
        L.ptor_render.restype = ctypes.c_int
        L.ptor_render.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
            ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
        ]
        assert L.ptor_stats_words() == len(STATS_FIELDS)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def hash_u32(x: int) -> int:
    return lib().ptor_kat_hash(x & 0xFFFFFFFF)


def random_floats(seed: int, n: int):
    """n successive getRandomFloat draws; returns (states, floats)."""
    s = ctypes.c_uint32(seed & 0xFFFFFFFF)
    states, vals = [], []
    for _ in range(n):
        v = lib().ptor_kat_random(ctypes.byref(s))
        states.append(s.value)
        vals.append(np.float32(v))
    return states, vals


def sincos(phi: np.ndarray):
    phi = np.ascontiguousarray(phi, np.float32)
    s = np.empty_like(phi)
    c = np.empty_like(phi)
    lib().ptor_kat_sincos_array(_ptr(phi), _ptr(s), _ptr(c), ctypes.c_int64(phi.size))
    return s, c


def pow_array(x: np.ndarray, y: float) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().ptor_kat_pow_array(_ptr(x), ctypes.c_float(y), _ptr(out), ctypes.c_int64(x.size))
    return out


def generate_ray(xc: int, yc: int, W: int, H: int, seed: int):
    s = ctypes.c_uint32(seed & 0xFFFFFFFF)
    out = np.zeros(6, np.float32)
    lib().ptor_kat_generate_ray(xc, yc, W, H, ctypes.byref(s), _ptr(out))
    return out[:3].copy(), out[3:].copy(), s.value


def intersect_world(tris: np.ndarray, origin, direction):
    o = np.ascontiguousarray(origin, np.float32)
    d = np.ascontiguousarray(direction, np.float32)
    out = np.zeros(7, np.float32)
    tri = ctypes.c_int(-1)
    tris = np.ascontiguousarray(tris)
    hit = lib().ptor_kat_intersect_world(_ptr(tris), len(tris), _ptr(o), _ptr(d), _ptr(out), ctypes.byref(tri))
    return bool(hit), float(out[0]), out[1:4].copy(), out[4:7].copy(), tri.value


def radiance(tris, mats, gid, W, H, frame, max_bounces=16):
    tris = np.ascontiguousarray(tris)
    mats = np.ascontiguousarray(mats)
    out = np.zeros(3, np.float32)
    lib().ptor_kat_radiance(_ptr(tris), len(tris), _ptr(mats), gid, W, H, frame, max_bounces, _ptr(out))
    return out


def paths(tris, mats, gids, frames, W, H, max_bounces=16, variant=""):
    """Radiance (n, 3) and hit sequences (n, max_bounces + 1) of the paths (gid[k], frame[k]); a sequence is the triangle
    hit at every bounce, then -1 (missed) or -2 (pdf <= 0), -3 beyond the path's end."""
    tris = np.ascontiguousarray(tris)
    mats = np.ascontiguousarray(mats)
    gids = np.ascontiguousarray(gids, np.int32)
    frames = np.ascontiguousarray(frames, np.int32)
    assert gids.shape == frames.shape and gids.ndim == 1
    rad = np.zeros((gids.size, 3), np.float32)
    hits = np.zeros((gids.size, max_bounces + 1), np.int32)
    lib(variant).ptor_kat_paths(_ptr(tris), len(tris), _ptr(mats), _ptr(gids), _ptr(frames), ctypes.c_int64(gids.size), W, H,
                                max_bounces, _ptr(rad), _ptr(hits))
    return rad, hits


def render(tris, mats, W, H, frames, *, frame_begin=0, max_bounces=16, fb=None,
           gid_begin=0, gid_count=None, nthreads=None, want_stats=False, variant=""):
    """Run frames [frame_begin, frame_begin+frames) over a gid range.

    Returns the (H*W, 4) float32 framebuffer (the one passed in, updated in place, or a new
    zero-initialised one) and, if asked, the work tallies as a dict.
    """
    tris = np.ascontiguousarray(tris)
    mats = np.ascontiguousarray(mats)
    if fb is None:
        fb = np.zeros((H * W, 4), np.float32)
    assert fb.dtype == np.float32 and fb.size == W * H * 4 and fb.flags.c_contiguous
    if gid_count is None:
        gid_count = W * H - gid_begin
    if nthreads is None:
        nthreads = os.cpu_count() or 1
    st = np.zeros(len(STATS_FIELDS), np.uint64)
    rc = lib(variant).ptor_render(_ptr(tris), len(tris), _ptr(mats), len(mats), _ptr(fb), W, H, frame_begin,
                           frames, max_bounces, gid_begin, gid_count, nthreads, _ptr(st))
    if rc != 0:
        raise ValueError("ptor_render rejected the arguments (rc=%d)" % rc)
    if want_stats:
        return fb, dict(zip(STATS_FIELDS, (int(v) for v in st)))
    return fb
