"""The streaming renderer (GPU tier): a bounded staging ring sized once, checkpointed trace launches, render
lanes -- and none of it may change a pixel.

What replaces the reference's per-frame ``launch1D`` + ``clFinish`` (test/RaytraceTest.cpp:250-268) is one asynchronous call per
render; its frames pass through two fixed radiance slots in chunks, every chunk's trace launch ends with a checkpoint that the next
one resumes, and every pixel still folds its frames in ascending order (test/ClKernels/GenerateColors.cl:314-321).
"""
import ctypes

import numpy as np
import pytest

from conftest import assert_fb_equal

pytestmark = pytest.mark.gpu

MIB = 1 << 20


def _render(device, tris, mats, W, H, frames, *, depth=16, begin=0, **opts):
    """One render through the fused entry point with the given device options; returns (pixels, raw stats)."""
    from oclpathtracer_amd import shim
    from oclpathtracer_amd.render import Renderer

    names = {"lanes": shim.PT_OPT_RENDER_LANES, "checkpoint": shim.PT_OPT_CHECKPOINT, "chunk": shim.PT_OPT_CHUNK_FRAMES}
    defaults = {"lanes": 2, "checkpoint": 1, "chunk": 0}
    for k, v in opts.items():
        device.setOption(names[k], v)
    try:
        r = Renderer(device, tris, mats, W, H, want_stats=True)
        if begin:
            r.render(begin, frame_begin=0, max_bounces=depth)
        r.render(frames, frame_begin=begin, max_bounces=depth)
        got, st = r.read(), r.read_stats_raw()
        r.release()
    finally:
        for k in opts:
            device.setOption(names[k], defaults[k])
    return got, st


def test_chunking_checkpoints_and_lanes_never_change_a_pixel(device, cornell, oracle):
    """The same 23 frames as ONE launch, as 2-, 3- and 7-frame chunks, with and without checkpoints, on one lane and on two: bit for
    bit the oracle's image, the same ray count -- and the checkpointed runs really did hand paths from launch to launch."""
    from oclpathtracer_amd import shim

    tris, mats = cornell
    W, H, frames = 96, 64, 23
    want, ost = oracle.render(tris, mats, W, H, frames, want_stats=True)
    for opts in ({"chunk": 0}, {"chunk": 7}, {"chunk": 3}, {"chunk": 2, "lanes": 1}, {"chunk": 3, "checkpoint": 0}, {"chunk": 3, "checkpoint": 0, "lanes": 1},
                 {"chunk": 1}):
        got, st = _render(device, tris, mats, W, H, frames, **opts)
        assert_fb_equal(got, want, "options %r" % (opts,))
        assert int(st[shim.PT_STAT_SAMPLES]) == W * H * frames and int(st[shim.PT_STAT_RAYS]) == ost["rays"], opts
        carried = int(st[shim.PT_STAT_CARRIED])
        if opts.get("checkpoint", 1) and opts["chunk"]:
            assert carried > 0, "no path crossed a launch boundary: %r" % (opts,)
        if not opts.get("checkpoint", 1) or not opts["chunk"]:
            assert carried == 0, opts     # (a render of one chunk has nothing to hand on: no checkpoint, no draining launch)


def test_checkpoints_on_a_launch_with_fewer_batches_than_waves(device, cornell, oracle):
    """A tiny image: most waves of the grid get no batch at all, the queue is empty almost at once and launches stop while paths of
    the PREVIOUS chunk are still under way -- which must hold them back until those are finished (their fold follows)."""
    tris, mats = cornell
    for W, H, frames, chunk in ((16, 8, 40, 1), (33, 7, 19, 2), (8, 8, 64, 5)):
        want = oracle.render(tris, mats, W, H, frames)
        got, _ = _render(device, tris, mats, W, H, frames, chunk=chunk)
        assert_fb_equal(got, want, "%dx%d x %d frames in chunks of %d" % (W, H, frames, chunk))


def test_resumed_renders_across_chunk_boundaries(device, cornell, oracle):
    """frame_begin > 0 (GenerateColors.cl:318-320 resumes from the framebuffer) with chunks that do not divide the call."""
    tris, mats = cornell
    W, H = 64, 48
    want = oracle.render(tris, mats, W, H, 5 + 17)
    got, _ = _render(device, tris, mats, W, H, 17, begin=5, chunk=4)
    assert_fb_equal(got, want, "5 frames, then 17 more in chunks of 4")


def test_the_staging_ring_is_sized_once_and_a_render_loop_allocates_nothing(cornell):
    """configs[2]'s radiance staging was 3.2 GB, grown on demand with a device wait inside whatever timed region met it.
    Now: 2 x 192 MiB reserved once; BASELINE configs[2] at full size fits the handle's whole workspace in 512 MB, and neither a longer
    render, nor a larger image, nor a render loop moves it."""
    from oclpathtracer_amd import adl
    from oclpathtracer_amd.render import Renderer

    tris, mats = cornell
    dev = adl.DeviceUtils.allocate(adl.TYPE_HIP, adl.Config(0))
    try:
        base = dev.getWorkspaceMemory()
        assert base < 8 * MIB                                   # side tables and queues only
        dev.reserveStaging(0)
        ring = dev.getWorkspaceMemory() - base
        assert ring == 2 * 192 * MIB
        r = Renderer(dev, tris, mats, 1024, 1024)
        r.render(256)                                           # BASELINE configs[2]
        dev.waitForCompletion()
        ws = dev.getWorkspaceMemory()
        assert ws <= 512 * 1000 * 1000, "configs[2] workspace %d bytes" % ws
        assert dev.getUsedMemory() == 36 * 64 + 18 * 64 + 1024 * 1024 * 16    # the caller's buffers, as the reference counts them
        for _ in range(3):
            r.render(256, frame_begin=0)
            r.render(1024, frame_begin=256)                     # a longer call walks more chunks through the same ring
        dev.waitForCompletion()
        assert dev.getWorkspaceMemory() == ws
        r.release()
        big = Renderer(dev, tris, mats, 2048, 2048)             # configs[3]'s image: 4 frames per slot
        big.render(8)
        dev.waitForCompletion()
        assert dev.getWorkspaceMemory() <= ws + 2048 * 2048 * 8 + (1 << 20)   # + its primary-ray masks
        big.release()
        # a caller's own size; and what is too small for one frame grows once instead of failing
        dev.reserveStaging(64 * MIB)
        assert dev.getWorkspaceMemory() < ws - 250 * MIB          # (2 x 32 MiB instead of 2 x 192; the 2048^2 masks stay)
        small = Renderer(dev, tris, mats, 256, 256)
        small.render(40)
        huge = Renderer(dev, tris, mats, 4096, 1024)            # 48 MiB per frame > a 32 MiB slot
        huge.render(2)
        dev.waitForCompletion()
        assert dev.getWorkspaceMemory() >= 2 * 48 * MIB
        small.release()
        huge.release()
    finally:
        adl.DeviceUtils.deallocate(dev)


def test_renders_return_before_the_gpu_has_finished_and_overlap(device, cornell, oracle):
    """Back-to-back renders into two framebuffers (the N-rank loop's two slots): enqueued in a fraction of their run time, and each
    image bit-exact although the next render's first launch ran beside its last, draining one."""
    import time

    from oclpathtracer_amd.render import Renderer

    tris, mats = cornell
    W, H, frames = 512, 512, 96
    want = oracle.render(tris, mats, W, H, frames)
    from oclpathtracer_amd import adl
    r = Renderer(device, tris, mats, W, H)               # ONE scene (a second Renderer would upload a second copy, and alternating
    fbs = [r.fb, adl.Buffer(device, W * H, adl.float4)]  # between them prepare the scene anew for every render), two framebuffers
    try:
        r.render(frames, frame_begin=0)
        device.waitForCompletion()
        t0 = time.perf_counter()
        for k in range(12):
            r.render(frames, frame_begin=0, fb=fbs[k & 1])
        t_enqueue = time.perf_counter() - t0
        device.waitForCompletion()
        t_total = time.perf_counter() - t0
        for fb in fbs:
            got = np.empty((W * H, 4), np.float32)
            fb.read(got, W * H)
            device.waitForCompletion()
            assert_fb_equal(got, want, "overlapped renders")
    finally:
        fbs[1].release()
        r.release()
    print("12 renders: enqueued in %.2f ms, finished after %.2f ms" % (t_enqueue * 1e3, t_total * 1e3))
    assert t_enqueue < 0.5 * t_total


def test_same_framebuffer_renders_fold_in_call_order(device, cornell, oracle):
    """Consecutive calls on ONE framebuffer with no wait in between: call k+1 continues the running mean call k left
    (GenerateColors.cl:318-320), although its trace launches start while call k's folds are still to come."""
    from oclpathtracer_amd.render import Renderer

    tris, mats = cornell
    W, H = 80, 60
    r = Renderer(device, tris, mats, W, H)
    try:
        for n in (3, 1, 7, 2, 11):
            r.render(n)                     # continues at frames_done
        got = r.read()
        total = r.frames_done
    finally:
        r.release()
    assert total == 24
    assert_fb_equal(got, oracle.render(tris, mats, W, H, total), "five calls, one wait")


def test_events_complete_with_the_render_and_hand_over_to_other_streams(device, cornell, oracle):
    """``ev`` of pt_render_frames sits behind the render's LAST fold (on whichever lane that ran); pt_event_wait_on makes a torch
    stream wait for it on the device, pt_device_wait_stream the other way round."""
    import torch

    from oclpathtracer_amd import adl
    from oclpathtracer_amd.render import Renderer

    tris, mats = cornell
    W, H, frames = 128, 96, 20
    want = oracle.render(tris, mats, W, H, frames)
    t = torch.zeros((H * W, 4), dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    device.waitStream(torch.cuda.current_stream().cuda_stream)       # the zero fill precedes the render
    r = Renderer(device, tris, mats, W, H, fb_device_ptr=t.data_ptr())
    ev = adl.SyncObject(device)
    try:
        r.render(frames, sync=ev)
        ev.waitOnStream(side.cuda_stream)
        with torch.cuda.stream(side):
            copy = t.clone()                                          # reads the framebuffer behind the event, on another stream
        device.waitStream(side.cuda_stream)                           # ... and the next render overwrites it only after that read
        r.render(frames, frame_begin=0, max_bounces=2)
        side.synchronize()
        assert_fb_equal(copy.cpu().numpy(), want, "consumer stream behind the render's event")
        assert ev.isComplete() and ev.getExecutionTimeNanoseconds() > 0
        device.waitForCompletion()
        assert_fb_equal(t.cpu().numpy(), oracle.render(tris, mats, W, H, frames, max_bounces=2), "second render")
    finally:
        ev.release()
        r.release()


def test_option_3_of_abi_version_1_is_still_accepted(device):
    """option id 3 (a kernel-variant switch) was removed without a version bump; version 2 takes its two values again."""
    from oclpathtracer_amd import shim

    lib = shim.load()
    assert lib.pt_abi_version() == 2
    assert lib.pt_device_set_option(device._h, 3, 0) == shim.PT_OK and lib.pt_device_set_option(device._h, 3, 1) == shim.PT_OK
    assert lib.pt_device_get_option(device._h, 3) == 0
    assert lib.pt_device_set_option(device._h, 3, 2) == shim.PT_ERR_INVALID
    assert lib.pt_device_set_option(device._h, shim.PT_OPT_RENDER_LANES, 3) == shim.PT_ERR_INVALID


def test_profile_union_counts_overlapped_launches_once(device, cornell):
    from oclpathtracer_amd import shim
    from oclpathtracer_amd.render import Renderer

    tris, mats = cornell
    lib = shim.load()
    rs = [Renderer(device, tris, mats, 512, 512) for _ in range(2)]
    try:
        shim.check(lib.pt_profile_enable(device._h, 1))
        shim.check(lib.pt_profile_reset(device._h))
        for k in range(6):
            rs[k & 1].render(16, frame_begin=0)
        tot, n, uni = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_double()
        shim.check(lib.pt_profile_query(device._h, shim.PT_PROF_TRACE, ctypes.byref(tot), ctypes.byref(n)))
        shim.check(lib.pt_profile_query_union(device._h, shim.PT_PROF_TRACE, ctypes.byref(uni)))
    finally:
        lib.pt_profile_enable(device._h, 0)
        for r in rs:
            r.release()
    assert n.value == 6                       # six renders of one chunk each: one launch per render (nothing to checkpoint, no draining launch)
    assert 0.0 < uni.value <= tot.value * 1.0001


def test_a_render_longer_than_32768_frames_goes_in_parts(device, cornell, oracle):
    """A path keeps its frame (counted from the render's first) in 16 bits: longer calls are split into parts of 32 768 frames, each
    its own sequence of launches; the fold chain joins them."""
    tris, mats = cornell
    W, H, frames = 4, 4, 32768 + 300
    want = oracle.render(tris, mats, W, H, frames)
    got, st = _render(device, tris, mats, W, H, frames)
    assert_fb_equal(got, want, "33 068 frames of a 4x4 image")
    assert int(st[0]) == W * H * frames


def test_a_callers_stream_keeps_plain_stream_order(device, cornell, oracle):
    """pt_device_set_stream: the handle runs ON a stream of the caller's, who is promised stream order without events -- a torch op
    enqueued on that stream right after a render reads the finished framebuffer, and a render enqueued right after a torch fill of
    the framebuffer's neighbour does not start before it (the lanes fork behind the stream and the stream joins them, every render)."""
    import torch

    from oclpathtracer_amd import shim
    from oclpathtracer_amd.render import Renderer

    tris, mats = cornell
    W, H, frames = 160, 120, 24
    want = oracle.render(tris, mats, W, H, frames)
    lib = shim.load()
    st = torch.cuda.Stream()
    t = torch.empty((H * W, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    shim.check(lib.pt_device_set_stream(device._h, ctypes.c_void_p(st.cuda_stream)))
    try:
        r = Renderer(device, tris, mats, W, H, fb_device_ptr=t.data_ptr())
        with torch.cuda.stream(st):
            t.fill_(7.0)                              # on the caller's stream, before the render: must not land on top of it
            r.render(frames, frame_begin=0)
            snap = t.clone()                          # ... and right behind it, no event: stream order alone
            r.render(5, frame_begin=0, max_bounces=1)
            snap2 = t.clone()
        st.synchronize()
        assert_fb_equal(snap.cpu().numpy(), want, "torch op right behind a render on the caller's stream")
        assert_fb_equal(snap2.cpu().numpy(), oracle.render(tris, mats, W, H, 5, max_bounces=1), "second render on the caller's stream")
        r.release()
    finally:
        shim.check(lib.pt_device_set_stream(device._h, None))
