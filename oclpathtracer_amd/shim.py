"""ctypes binding of libptshim.so -- the C ABI declared in include/pt_shim.h.

This is the only way Python reaches the hot path; there is no Python/NumPy/torch fallback.
If the library is missing, ``load()`` raises: build it with ``__graft_entry__.build()``
(``make -C oclpathtracer_amd/csrc``).
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PT_SHIM_LIB lets an experiment point at an alternative build of the same ABI (A/B kernel variants)
LIB_PATH = os.environ.get("PT_SHIM_LIB") or os.path.join(_HERE, "libptshim.so")

PT_OK = 0
PT_ERR_INVALID, PT_ERR_NO_DEVICE, PT_ERR_OOM, PT_ERR_HIP, PT_ERR_NOT_FOUND, PT_ERR_ARGS, PT_ERR_RANGE, PT_ERR_TRAVERSAL = range(1, 9)
PT_INFO_NAME, PT_INFO_BOARD, PT_INFO_VENDOR, PT_INFO_VERSION = range(4)
PT_OPT_BATCH_FRAMES, PT_OPT_CHUNK_FRAMES, PT_OPT_PROFILE_RETURN_TIME = 0, 1, 2   # (3 is not assigned)
PT_OPT_QUAD_FILTER, PT_OPT_ACCEL, PT_OPT_BVH_TALLY, PT_OPT_PRIMARY_MASKS, PT_OPT_BVH_STACK_LIMIT, PT_OPT_RENDER_LANES, PT_OPT_CHECKPOINT, PT_OPT_BVH_BUILD_COUNT = 4, 5, 6, 7, 8, 9, 10, 11
PT_SHIM_ABI_VERSION = 2  # include/pt_shim.h; load() refuses a library of another version
PT_MAX_ARG_SIZE = 64
PT_MAX_ARG_COUNT = 64
PT_STAT_SAMPLES, PT_STAT_RAYS, PT_STAT_WORDS = 0, 1, 16
PT_STAT_BVH_NODES, PT_STAT_BVH_TRIS, PT_STAT_BVH_STEPS, PT_STAT_BVH_TRI_STEPS, PT_STAT_BVH_MAX_STACK, PT_STAT_CARRIED, PT_STAT_BVH_GRAZING = 2, 3, 4, 5, 6, 7, 8
PT_PROF_TRACE, PT_PROF_FOLD = 0, 1
PT_STREAM_LEGACY = 1  # hipStreamLegacy: how a caller names the legacy default stream to pt_device_set_stream

_c = ctypes
_H = _c.c_void_p  # opaque handles


class LaunchArg(_c.Structure):
    """pt_launch_arg (mirrors Launcher::Args, Adl/AdlKernel.h:133-140)."""

    _fields_ = [
        ("is_buffer", _c.c_int32),
        ("read_only", _c.c_int32),
        ("size", _c.c_uint64),
        ("buffer", _H),
        ("data", _c.c_ubyte * PT_MAX_ARG_SIZE),
    ]


class RenderParams(_c.Structure):
    """pt_render_params."""

    _fields_ = [
        ("width", _c.c_int32), ("height", _c.c_int32),
        ("frame_begin", _c.c_int32), ("frame_count", _c.c_int32),
        ("max_bounces", _c.c_int32), ("num_triangles", _c.c_int32), ("num_materials", _c.c_int32),
        ("stripe_rows", _c.c_int32), ("n_ranks", _c.c_int32), ("rank", _c.c_int32),
        ("reserved", _c.c_int32 * 6),
    ]


# every symbol include/pt_shim.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "pt_last_error": (_c.c_char_p, []),
    "pt_abi_version": (_c.c_int, []),
    "pt_init": (_c.c_int, []),
    "pt_quit": (None, []),
    "pt_device_count": (_c.c_int, []),
    "pt_device_create": (_c.c_int, [_c.c_int, _c.POINTER(_H)]),
    "pt_device_destroy": (_c.c_int, [_H]),
    "pt_device_info": (_c.c_int, [_H, _c.c_int, _c.c_char_p]),
    "pt_device_max_alloc": (_c.c_uint64, [_H]),
    "pt_device_mem_size": (_c.c_uint64, [_H]),
    "pt_device_used_memory": (_c.c_uint64, [_H]),
    "pt_device_peak_memory": (_c.c_uint64, [_H]),
    "pt_device_workspace_memory": (_c.c_uint64, [_H]),
    "pt_device_reserve_staging": (_c.c_int, [_H, _c.c_size_t]),
    "pt_device_num_cus": (_c.c_int, [_H]),
    "pt_device_set_stream": (_c.c_int, [_H, _c.c_void_p]),
    "pt_device_get_stream": (_c.c_void_p, [_H]),
    "pt_device_wait_stream": (_c.c_int, [_H, _c.c_void_p]),
    "pt_device_wait_hip_event": (_c.c_int, [_H, _c.c_void_p]),
    "pt_event_wait_on": (_c.c_int, [_H, _c.c_void_p]),
    "pt_sync": (_c.c_int, [_H]),
    "pt_flush": (_c.c_int, [_H]),
    "pt_device_set_option": (_c.c_int, [_H, _c.c_int, _c.c_int64]),
    "pt_device_get_option": (_c.c_int64, [_H, _c.c_int]),
    "pt_buffer_alloc": (_c.c_int, [_H, _c.c_size_t, _c.POINTER(_H)]),
    "pt_buffer_wrap": (_c.c_int, [_H, _c.c_void_p, _c.c_size_t, _c.POINTER(_H)]),
    "pt_buffer_free": (_c.c_int, [_H]),
    "pt_buffer_size": (_c.c_size_t, [_H]),
    "pt_buffer_device_ptr": (_c.c_void_p, [_H]),
    "pt_buffer_address": (_c.c_void_p, [_H]),
    "pt_buffer_write": (_c.c_int, [_H, _c.c_void_p, _c.c_size_t, _c.c_size_t, _H]),
    "pt_buffer_read": (_c.c_int, [_H, _c.c_void_p, _c.c_size_t, _c.c_size_t, _H]),
    "pt_buffer_copy": (_c.c_int, [_H, _H, _c.c_size_t, _c.c_size_t, _c.c_size_t, _H]),
    "pt_buffer_map": (_c.c_void_p, [_H, _c.c_size_t, _c.c_int]),
    "pt_buffer_unmap": (_c.c_int, [_H, _c.c_void_p]),
    "pt_host_alloc": (_c.c_int, [_c.c_size_t, _c.POINTER(_c.c_void_p)]),
    "pt_host_free": (_c.c_int, [_c.c_void_p]),
    "pt_event_create": (_c.c_int, [_H, _c.POINTER(_H)]),
    "pt_event_destroy": (_c.c_int, [_H]),
    "pt_event_wait": (_c.c_int, [_H]),
    "pt_event_is_complete": (_c.c_int, [_H]),
    "pt_event_elapsed_ns": (_c.c_int, [_H, _c.POINTER(_c.c_uint64)]),
    "pt_kernel_get": (_c.c_int, [_H, _c.c_char_p, _c.c_char_p, _c.POINTER(_H)]),
    "pt_launch_2d": (_c.c_int, [_H, _H, _c.POINTER(LaunchArg), _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _H,
                                _c.POINTER(_c.c_float)]),
    "pt_local_rows": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "pt_render_frames": (_c.c_int, [_H, _H, _H, _H, _c.POINTER(RenderParams), _H, _H]),
    "pt_profile_enable": (_c.c_int, [_H, _c.c_int]),
    "pt_profile_query": (_c.c_int, [_H, _c.c_int, _c.POINTER(_c.c_double), _c.POINTER(_c.c_uint64)]),
    "pt_profile_query_union": (_c.c_int, [_H, _c.c_int, _c.POINTER(_c.c_double)]),
    "pt_profile_reset": (_c.c_int, [_H]),
    "pt_assemble_stripes": (_c.c_int, [_H, _H, _H, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _H]),
    "pt_assemble_stripes_on": (_c.c_int, [_H, _H, _H, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    "pt_tonemap_ppm": (_c.c_int, [_H, _H, _H, _c.c_size_t, _H]),
}


class ShimError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__("pt_shim error %d: %s" % (code, message))
        self.code = code


_lib = None


def load():
    """Load libptshim.so and declare every entry point.  Raises if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ShimError(PT_ERR_NO_DEVICE, "libptshim.so is not built (run __graft_entry__.build()); "
                                              "there is no CPU fallback for the hot path")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = ABI drift; do not mask it
            fn.restype = res
            fn.argtypes = args
        if lib.pt_abi_version() != PT_SHIM_ABI_VERSION:
            raise ShimError(PT_ERR_INVALID, "libptshim.so speaks ABI version %d, this binding %d: rebuild (__graft_entry__.build())"
                            % (lib.pt_abi_version(), PT_SHIM_ABI_VERSION))
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != PT_OK:
        raise ShimError(rc, load().pt_last_error().decode("utf-8", "replace"))
