"""Host drivers of the hot path.

* ``raycast_reference_loop`` -- the reference harness's frame loop, call for call
  (test/RaytraceTest.cpp:202-291 ``TEST_F(DeviceTest, RayCast)``), through the Adl-shaped API.
* ``Renderer``               -- the fused entry point ``pt_render_frames`` (one call = many frames)
  with image-stripe sharding parameters for multi-GPU.

All compute happens in libptshim.so on the GPU; nothing here has a CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np

from . import adl, scene, shim

BOUNCES = 16        # GenerateColors.cl:5
NUM_TRIANGLES = 36  # GenerateColors.cl:6


def upload_scene(dev: adl.Device, triangles: np.ndarray, materials: np.ndarray):
    """Scene upload exactly as RaytraceTest.cpp:222-246 does it: map, element-wise copy, unmap.

    (The reference over-allocates both buffers 64x by passing bytes as the element count,
    :222-223; that quirk is not reproduced.)"""
    tbuf = adl.Buffer(dev, len(triangles), scene.TRIANGLE_DTYPE)
    mbuf = adl.Buffer(dev, len(materials), scene.MATERIAL_DTYPE)
    tb = tbuf.getHostPtr()
    mb = mbuf.getHostPtr()
    adl.DeviceUtils.waitForCompletion(dev)
    tb[:] = triangles
    mb[:] = materials
    tbuf.returnHostPtr(tb)
    mbuf.returnHostPtr(mb)
    adl.DeviceUtils.waitForCompletion(dev)
    return tbuf, mbuf


def raycast_reference_loop(dev: adl.Device, triangles: np.ndarray, materials: np.ndarray, dimension: int = 512,
                           frames: int = 10000, kernel_path: str = "../test/ClKernels/GenerateColors") -> np.ndarray:
    """RaytraceTest.cpp:216-290 with ``dimension`` and the frame count as parameters.

    Returns the framebuffer as an (H*W, 4) float32 array (what the reference maps at :272)."""
    frameBuff = adl.Buffer(dev, dimension * dimension, adl.float4)
    tBuffer, materialBuffer = upload_scene(dev, triangles, materials)
    frameCount = 0
    while frameCount != frames:
        res = np.array([dimension, dimension, frameCount, 0], np.int32)
        frameCount += 1
        bInfo = [adl.BufferInfo(tBuffer), adl.BufferInfo(materialBuffer), adl.BufferInfo(frameBuff)]
        launcher = adl.Launcher(dev, dev.getKernel(kernel_path, "GenerateColors"))
        launcher.setBuffers(bInfo, len(bInfo))
        launcher.setConst(res)
        launcher.launch1D(dimension * dimension)
        adl.DeviceUtils.waitForCompletion(dev)
    h = frameBuff.getHostPtr()
    adl.DeviceUtils.waitForCompletion(dev)
    out = np.array(h, copy=True)
    for b in (frameBuff, tBuffer, materialBuffer):
        b.release()
    return out


class Renderer:
    """One device's share of an image, rendered with the fused multi-frame entry point.

    ``n_ranks``/``rank``/``stripe_rows`` select the rows this device owns (rows are
    dealt in stripes, round-robin); the local framebuffer holds exactly those rows.  With the
    defaults it is the whole image.  ``fb_device_ptr`` lets the framebuffer live in caller-owned
    device memory (a torch tensor) so a collective can move it without a copy.
    """

    def __init__(self, dev: adl.Device, triangles: np.ndarray, materials: np.ndarray, width: int, height: int, *,
                 n_ranks: int = 1, rank: int = 0, stripe_rows: int = 16, fb_device_ptr: Optional[int] = None,
                 want_stats: bool = False):
        self.dev = dev
        self.width, self.height = int(width), int(height)
        self.n_ranks, self.rank, self.stripe_rows = int(n_ranks), int(rank), int(stripe_rows)
        self.num_triangles, self.num_materials = len(triangles), len(materials)
        lib = shim.load()
        self.local_rows = lib.pt_local_rows(self.height, self.stripe_rows, self.n_ranks, self.rank)
        if self.local_rows < 0:
            raise ValueError("invalid stripe geometry")
        self.local_pixels = self.local_rows * self.width
        self.tbuf, self.mbuf = upload_scene(dev, triangles, materials)
        self.fb = adl.Buffer(dtype=adl.float4)
        if fb_device_ptr is None:
            self.fb.allocate(dev, max(self.local_pixels, 1))
        else:
            self.fb.setRawPtr(dev, fb_device_ptr, max(self.local_pixels, 1))
        self.stats = None
        if want_stats:
            self.stats = adl.Buffer(dev, shim.PT_STAT_WORDS, np.uint64)
            self.stats.write(np.zeros(shim.PT_STAT_WORDS, np.uint64), shim.PT_STAT_WORDS)
        self.frames_done = 0

    def render(self, frames: int, *, frame_begin: Optional[int] = None, max_bounces: int = BOUNCES,
               sync: Optional[adl.SyncObject] = None, fb: Optional[adl.Buffer] = None) -> None:
        """Enqueue frames [frame_begin, frame_begin+frames) (default: continue after the last call).
        ``fb``: another framebuffer of the same size to render into (a pipeline's second buffer)."""
        if frame_begin is None:
            frame_begin = self.frames_done
        p = shim.RenderParams()
        p.width, p.height = self.width, self.height
        p.frame_begin, p.frame_count = int(frame_begin), int(frames)
        p.max_bounces = int(max_bounces)
        p.num_triangles, p.num_materials = self.num_triangles, self.num_materials
        p.stripe_rows, p.n_ranks, p.rank = self.stripe_rows, self.n_ranks, self.rank
        shim.check(shim.load().pt_render_frames(self.dev._h, self.tbuf._h, self.mbuf._h, (fb or self.fb)._h, ctypes.byref(p),
                                                self.stats._h if self.stats else None,
                                                sync._h if sync is not None else None))
        self.frames_done = frame_begin + frames

    def read(self) -> np.ndarray:
        """Local framebuffer as (local_rows*W, 4) float32 (synchronises)."""
        out = np.empty((self.local_pixels, 4), np.float32)
        if self.local_pixels:
            self.fb.read(out, self.local_pixels)
        self.dev.waitForCompletion()
        return out

    def read_stats_raw(self) -> np.ndarray:
        """All PT_STAT_WORDS work counters (uint64), as accumulated since the buffer was last zeroed."""
        if self.stats is None:
            raise ValueError("renderer was created without want_stats")
        out = np.zeros(shim.PT_STAT_WORDS, np.uint64)
        self.stats.read(out, shim.PT_STAT_WORDS)
        self.dev.waitForCompletion()
        return out

    def read_stats(self) -> dict:
        out = self.read_stats_raw()
        return {"samples": int(out[shim.PT_STAT_SAMPLES]), "rays": int(out[shim.PT_STAT_RAYS])}

    def global_rows(self) -> np.ndarray:
        """Global row index of every local row (ascending)."""
        rows = np.arange(self.height)
        return rows[(rows // self.stripe_rows) % self.n_ranks == self.rank]

    def release(self) -> None:
        for b in (self.tbuf, self.mbuf, self.fb, self.stats):
            if b is not None:
                b.release()
