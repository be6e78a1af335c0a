"""Progressive driver: persistent accumulation with double-buffered asynchronous readback.

How the reference is actually used (RaytraceTest.cpp:250-268): a 10 000-frame refinement loop whose
image one wants to look at while it converges.  The kernel's ``frame``
argument already makes accumulation resumable (GenerateColors.cl:314-321); this driver adds the
host side: every :meth:`step` enqueues the next frames and an asynchronous device-to-host copy of the
framebuffer into one of two page-locked host buffers, followed by an event -- the host never waits for
the GPU, the GPU never waits for the host, and :meth:`latest` returns the newest image whose copy has
completed.  Pixels are those of a one-shot render of the same frames, bit for bit.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import numpy as np

from . import adl, shim
from .render import BOUNCES, Renderer


class ProgressiveRenderer:
    def __init__(self, dev: adl.Device, triangles: np.ndarray, materials: np.ndarray, width: int, height: int, *,
                 frames_per_step: int = 16, max_bounces: int = BOUNCES):
        self.dev = dev
        self.renderer = Renderer(dev, triangles, materials, width, height)
        self.width, self.height = int(width), int(height)
        self.frames_per_step, self.max_bounces = int(frames_per_step), int(max_bounces)
        self._lib = shim.load()
        self._bytes = self.width * self.height * 16
        self._host, self._views, self._events, self._frames = [], [], [], [0, 0]
        for _ in range(2):
            p = ctypes.c_void_p()
            shim.check(self._lib.pt_host_alloc(self._bytes, ctypes.byref(p)))
            self._host.append(p)
            buf = (ctypes.c_float * (self.width * self.height * 4)).from_address(p.value)
            self._views.append(np.frombuffer(buf, np.float32).reshape(self.width * self.height, 4))
            self._events.append(adl.SyncObject(dev))
        self._steps = 0

    @property
    def frames_done(self) -> int:
        """Frames enqueued so far (the next step starts at this frame index)."""
        return self.renderer.frames_done

    def step(self, frames: Optional[int] = None) -> None:
        """Enqueue the next ``frames`` frames and the readback of their result; returns at once."""
        n = self.frames_per_step if frames is None else int(frames)
        slot = self._steps % 2
        # the slot's previous copy (two steps ago) must have landed before it is overwritten
        if self._steps >= 2:
            self._events[slot].waitForCompletion()
        self.renderer.render(n, max_bounces=self.max_bounces)
        shim.check(self._lib.pt_buffer_read(self.renderer.fb._h, self._host[slot], self._bytes, 0, self._events[slot]._h))
        self._frames[slot] = self.renderer.frames_done
        self._steps += 1

    def latest(self, block: bool = False) -> Optional[Tuple[int, np.ndarray]]:
        """``(frames, image)`` of the newest completed snapshot, ``image`` a (W*H, 4) float32 view of
        the page-locked buffer (valid until two more steps have been enqueued); None before the
        first one has landed.  ``block`` waits for the most recently enqueued snapshot."""
        if self._steps == 0:
            return None
        if block:
            self._events[(self._steps - 1) % 2].waitForCompletion()
        for index in (self._steps - 1, self._steps - 2):  # newest snapshot first, then the one before
            if index >= 0 and self._events[index % 2].isComplete():
                return self._frames[index % 2], self._views[index % 2]
        return None

    def release(self) -> None:
        self.dev.waitForCompletion()
        for e in self._events:
            e.release()
        for p in self._host:
            self._lib.pt_host_free(p)
        self._host, self._views, self._events = [], [], []
        self.renderer.release()
