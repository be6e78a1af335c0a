"""Multi-GPU sharding of the framebuffer: one process per GPU, image stripes, one gather.

The path shards by independent units: every (pixel, frame) sample depends only
on its GLOBAL pixel id and frame (GenerateColors.cl:305-312), so ranks share nothing while
rendering.  The only exchange is the final image assembly: rank k sends its slab of rows to
rank 0 (``torch.distributed.gather``; backend "nccl" = RCCL over xGMI -- 7 peers write into the
root over 7 distinct links, so no ring), and rank 0 scatters the slabs into image order with
the ``pt_assemble_stripes`` HIP kernel.

Rows are dealt in stripes of ``stripe_rows`` rows, round-robin over ranks, because path length
(hence cost) varies with image height: ceiling rows are cheap, floor/box rows are deep.

torch is plumbing here (device memory, streams, the collective); import it BEFORE the shim so
both bind to the same HIP runtime.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from . import adl, shim
from .render import Renderer


class StripePlan:
    """Which image rows each rank owns.  Pure index math; mirrors pt_local_rows() and the
    row mapping of pt_trace_kernel / pt_assemble_kernel."""

    def __init__(self, height: int, stripe_rows: int, world: int):
        if height < 1 or stripe_rows < 1 or world < 1:
            raise ValueError("invalid stripe plan")
        self.height, self.stripe_rows, self.world = int(height), int(stripe_rows), int(world)

    def global_rows(self, rank: int) -> np.ndarray:
        rows = np.arange(self.height)
        return rows[(rows // self.stripe_rows) % self.world == rank]

    def local_rows(self, rank: int) -> int:
        period = self.stripe_rows * self.world
        full, rem = divmod(self.height, period)
        start = rank * self.stripe_rows
        return full * self.stripe_rows + (min(self.stripe_rows, rem - start) if rem > start else 0)

    @property
    def slab_rows(self) -> int:
        """Rows per gathered slab (the largest share; smaller shares are zero-padded)."""
        return max(self.local_rows(r) for r in range(self.world))


def gather_slabs(local_slab: torch.Tensor, world: int, rank: int, dst: int = 0, group=None,
                 out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """Gather equally-shaped per-rank slabs to ``dst``; returns [world, *slab.shape] there, else None.

    ``out`` (on ``dst`` only) is an optional preallocated destination.  Works with any initialised
    backend (nccl on GPUs; gloo in the CPU rehearsal tests)."""
    if world == 1:
        return local_slab.unsqueeze(0)
    if local_slab.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal of the N-rank path on a box with fewer GPUs than ranks (bench.py --rehearse, the
        # one-GPU world-2 test): gloo has no device gather, the slabs travel through the host
        host = gather_slabs(local_slab.cpu(), world, rank, dst, group)
        if rank != dst:
            return None
        if out is None:
            return host.to(local_slab.device)
        out.copy_(host)
        return out
    if rank == dst:
        if out is None:
            out = torch.empty((world,) + tuple(local_slab.shape), dtype=local_slab.dtype, device=local_slab.device)
        dist.gather(local_slab, [out[i] for i in range(world)], dst=dst, group=group)
        return out
    dist.gather(local_slab, None, dst=dst, group=group)
    return None


class StripeImage:
    """One rank's renderer + the gather/assemble step.  The local framebuffer is a torch CUDA
    tensor (so the collective moves it with no staging copy) wrapped by the shim."""

    def __init__(self, dev: adl.Device, triangles, materials, width: int, height: int, *, world: int = 1, rank: int = 0,
                 stripe_rows: int = 16, want_stats: bool = False, pipelined: bool = False):
        """``pipelined``: two local framebuffers (and, for N > 1, two gather buffers), so that the collective of one render can run
        while the next render is already on the GPU -- and consecutive renders overlap: the next image's first trace launch fills
        the machine while this image's last one runs its paths out (``render`` returns the slot it rendered into, ``gather(slot)``
        takes it; see ``bench.py``).  Every render must then start at frame 0 or continue its own slot's frames."""
        self.dev, self.world, self.rank = dev, int(world), int(rank)
        self.width, self.height = int(width), int(height)
        self.plan = StripePlan(height, stripe_rows, world)
        self.cuda = torch.device("cuda", torch.cuda.current_device())
        # The device handle keeps its OWN stream and render lanes (pt_shim.h): consecutive renders then overlap on the GPU --
        # the next image's first trace launch fills the machine while this image's last paths drain -- which a stream shared
        # with torch would serialise.  Every hand-over between the shim and the stream torch ops / the collective run on is a
        # device-side event wait (Device.waitStream / waitHipEvent, SyncObject.waitOnStream), never a host sync.
        self.pipelined = bool(pipelined)
        nslots = 2 if self.pipelined else 1
        self._locals = [torch.zeros((self.plan.slab_rows, self.width, 4), dtype=torch.float32, device=self.cuda) for _ in range(nslots)]
        self.local = self._locals[0]
        dev.waitStream(torch.cuda.current_stream(self.cuda).cuda_stream)  # the zero fill precedes the first render
        self.renderer = Renderer(dev, triangles, materials, width, height, n_ranks=world, rank=rank,
                                 stripe_rows=stripe_rows, fb_device_ptr=self.local.data_ptr(), want_stats=want_stats)
        assert self.renderer.local_rows == self.plan.local_rows(rank)
        self._fbs = [self.renderer.fb]
        for t in self._locals[1:]:
            b = adl.Buffer(dtype=adl.float4)
            b.setRawPtr(dev, t.data_ptr(), max(self.renderer.local_pixels, 1))
            self._fbs.append(b)
        # per slot: "its render is done" (a shim event, behind the render's last fold), "its collective has read it" (on torch's stream)
        self._rendered = [adl.SyncObject(dev) for _ in range(nslots)]
        self._collected = [None] * nslots
        self._slot = 0       # the slot the next render goes to
        self._last = 0       # the slot the last render went to
        self.image = None
        self._slabs = []
        self._gbufs = []
        self._ibuf = None
        self._asm_stream = None
        if rank == 0 and world > 1:
            # the assembly kernel gets a stream of its own: behind the collective, beside the next render (on the shim
            # stream it would queue behind that render and every image would arrive one render late)
            self._asm_stream = torch.cuda.Stream(device=self.cuda)
            # gather destinations and the assembled image live for the object's lifetime: a render loop
            # allocates, wraps and synchronises nothing per step
            self.image = torch.empty((self.height, self.width, 4), dtype=torch.float32, device=self.cuda)
            self._ibuf = adl.Buffer(dtype=adl.float4)
            self._ibuf.setRawPtr(dev, self.image.data_ptr(), self.image.numel() // 4)
            for _ in range(nslots):
                sl = torch.empty((self.world, self.plan.slab_rows, self.width, 4), dtype=torch.float32, device=self.cuda)
                g = adl.Buffer(dtype=adl.float4)
                g.setRawPtr(dev, sl.data_ptr(), sl.numel() // 4)
                self._slabs.append(sl)
                self._gbufs.append(g)

    def render(self, frames: int, *, frame_begin: Optional[int] = None, max_bounces: int = 16) -> int:
        """Enqueue frames; returns the slot rendered into (always 0 unless ``pipelined``).
        ``self.local`` may be consumed by torch ops on the current stream after ``ready()`` (or ``gather()``),
        without a host synchronisation."""
        slot = self._slot
        if self.pipelined:
            # only the collective that last read THIS slot has to be over; the other slot's may still be running
            if self._collected[slot] is not None:
                self.dev.waitHipEvent(self._collected[slot].cuda_event)
        else:
            # torch work already queued on the current stream that touches self.local (a previous
            # gather reading it, a user op) must finish before the render overwrites it
            self.dev.waitStream(torch.cuda.current_stream(self.cuda).cuda_stream)
        self.renderer.render(frames, frame_begin=frame_begin, max_bounces=max_bounces, fb=self._fbs[slot], sync=self._rendered[slot])
        self._last = slot
        self.local = self._locals[slot]
        if self.pipelined:
            self._slot ^= 1
        return slot

    def ready(self) -> torch.Tensor:
        """Order torch's current stream after the last render and return the local framebuffer tensor
        (device-side wait only)."""
        self._rendered[self._last].waitOnStream(torch.cuda.current_stream(self.cuda).cuda_stream)
        return self.local

    def gather(self, slot: Optional[int] = None) -> Optional[torch.Tensor]:
        """Assemble the full image of the last render (or of ``slot``) on rank 0 (returns it there; None elsewhere).

        Pipelined use: ``s = img.render(...)`` of the NEXT image first, then ``img.gather(previous_slot)``: the
        collective runs on torch's stream beside that render, the assembly kernel follows it on a stream of its own."""
        cur = torch.cuda.current_stream(self.cuda)
        slot = self._last if slot is None else int(slot)
        self._rendered[slot].waitOnStream(cur.cuda_stream)   # this slot's render -> collective / consumer (a later render may be in flight)
        if self.world == 1:
            self.image = self._locals[slot][: self.height]
            return self.image
        gather_slabs(self._locals[slot], self.world, self.rank, out=self._slabs[slot] if self.rank == 0 else None)
        if self.pipelined:
            ev = torch.cuda.Event()
            ev.record(cur)
            self._collected[slot] = ev             # the slot may be rendered into again after this
        if self.rank != 0:
            return None
        # collective -> assembly kernel (reads _slabs[slot], writes self.image) on its own stream -> whoever consumes
        # self.image on the current stream.  (The next collective into this slot's slab and the next assembly into
        # self.image are enqueued on / behind the current stream, which has waited for this assembly: no overlap.)
        self._asm_stream.wait_stream(cur)
        shim.check(shim.load().pt_assemble_stripes_on(self.dev._h, self._gbufs[slot]._h, self._ibuf._h, self.width, self.height,
                                                      self.plan.stripe_rows, self.world, self.plan.slab_rows,
                                                      self._asm_stream.cuda_stream))
        cur.wait_stream(self._asm_stream)
        return self.image

    def reset_stats(self) -> None:
        if self.renderer.stats is not None:
            self.renderer.stats.write(np.zeros(shim.PT_STAT_WORDS, np.uint64), shim.PT_STAT_WORDS)
            self.dev.waitForCompletion()

    def read_stats(self) -> dict:
        return self.renderer.read_stats()

    def release(self) -> None:
        for b in self._gbufs + [self._ibuf] + self._fbs[1:]:
            if b is not None:
                b.release()
        self._gbufs, self._ibuf, self._fbs = [], None, self._fbs[:1]
        self.renderer.release()
        for e in self._rendered:
            e.release()
        self._rendered = []
