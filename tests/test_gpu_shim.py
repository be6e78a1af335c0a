"""GPU tests of the drop-in boundary itself: the Adl-shaped API over the C ABI (the reference's
commented-out smoke tests, test/main.cpp:74-152), error behaviour, and the C++ RaytraceTest-shaped
harness built on include/pt_adl.hpp."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, assert_fb_equal

pytestmark = pytest.mark.gpu


def test_device_info(device):
    assert "gfx950" in device.getDeviceVersion()
    assert device.getDeviceVendor().startswith("Advanced Micro Devices")
    assert device.getMaxAllocationSize() > (64 << 30)  # 288 GB part
    from oclpathtracer_amd import adl
    assert adl.DeviceUtils.getNDevices() >= 1 and adl.DeviceUtils.getNCUs(device) == 256


def test_write_read_copy_roundtrip(device):
    from oclpathtracer_amd import adl

    n = 128
    host = (np.arange(n, dtype=np.int32) * 3 + 1)
    b = adl.Buffer(device, n, np.int32)
    c = adl.Buffer(device, n, np.int32)
    used0 = device.getUsedMemory()
    b.write(host, n)
    c.write(b, n)  # device-to-device
    back = np.full(64, -1, np.int32)
    c.read(back, 64, 64)  # offset read
    device.waitForCompletion()
    assert np.array_equal(back, host[64:])
    assert used0 >= 2 * n * 4
    b.release()
    c.release()
    assert device.getUsedMemory() == used0 - 2 * n * 4
    assert device.getPeakMemory() >= used0


def test_map_unmap(device):
    from oclpathtracer_amd import adl

    b = adl.Buffer(device, 1024, np.float32)
    p = b.getHostPtr()
    device.waitForCompletion()
    p[:] = np.arange(1024, dtype=np.float32) * 0.5
    b.returnHostPtr(p)
    device.waitForCompletion()
    q = b.getHostPtr(blocking=True)
    assert np.array_equal(q, np.arange(1024, dtype=np.float32) * 0.5)
    with pytest.raises(Exception):
        b.getHostPtr()  # already mapped
    b.returnHostPtr(q)
    b.release()


def test_partial_map_writes_back_only_the_mapped_range(device):
    """Buffer::getHostPtr(size) maps a sub-range (Adl/Adl.inl:247-252).  The pinned staging range is kept
    at its largest size ever, so a partial unmap must copy back exactly the bytes of ITS map: a full
    map/unmap, then a change of the buffer on the device, then a partial map/unmap must leave the rest
    of the device buffer as the device last wrote it (not revert it to the stale staging snapshot)."""
    from oclpathtracer_amd import adl

    n = 256
    b = adl.Buffer(device, n, np.int32)
    first = np.arange(n, dtype=np.int32)
    p = b.getHostPtr(blocking=True)          # full map: staging now holds n elements
    p[:] = first
    b.returnHostPtr(p)
    second = first * 7 + 3
    b.write(second, n)                        # the device contents move on behind the staging copy
    q = b.getHostPtr(16, blocking=True)       # partial map: refreshes 16 elements of the staging range
    assert q.shape[0] == 16 and np.array_equal(q, second[:16])
    q[:] = -5
    b.returnHostPtr(q)
    back = np.empty(n, np.int32)
    b.read(back, n)
    device.waitForCompletion()
    want = second.copy()
    want[:16] = -5
    assert np.array_equal(back, want)
    b.release()


def test_wrapped_buffers_are_never_deferred(device, cornell):
    """launch1D + waitForCompletion is clFinish in the reference (RaytraceTest.cpp:264-267): memory the
    caller can reach behind the ABI (a wrapped torch tensor; a buffer whose device pointer was handed
    out) must hold the frame after the wait, and a scene rewritten between launches must only affect
    later frames -- so such launches are not batched."""
    import torch

    from oclpathtracer_amd import adl, shim
    from oclpathtracer_amd.render import upload_scene

    tris, mats = cornell
    dim = 32
    want1 = np.load(os.path.join(GOLDEN, "cornell_64x64_f1_d16.npy"))  # noqa: F841 (shape reference only)
    lib = shim.load()
    assert device._lib.pt_device_get_option(device._h, shim.PT_OPT_BATCH_FRAMES) == 1

    def launch(tb, mb, fb, z):
        launcher = adl.Launcher(device, device.getKernel("../test/ClKernels/GenerateColors", "GenerateColors"))
        launcher.setBuffers([adl.BufferInfo(tb), adl.BufferInfo(mb), adl.BufferInfo(fb)], 3)
        launcher.setConst(np.array([dim, dim, z, 0], np.int32))
        launcher.launch1D(dim * dim)
        adl.DeviceUtils.waitForCompletion(device)

    # reference image of frames 0..2 through owned, unexposed buffers (deferred path)
    tb, mb = upload_scene(device, tris, mats)
    fb = adl.Buffer(device, dim * dim, adl.float4)
    for z in range(3):
        launch(tb, mb, fb, z)
    ref = np.empty((dim * dim, 4), np.float32)
    fb.read(ref, dim * dim)
    device.waitForCompletion()

    # (1) framebuffer = caller-owned torch memory: after each wait the tensor already holds the frame
    t = torch.zeros((dim * dim, 4), dtype=torch.float32, device="cuda")
    wfb = adl.Buffer(dtype=adl.float4)
    wfb.setRawPtr(device, t.data_ptr(), dim * dim)
    snaps = []
    for z in range(3):
        launch(tb, mb, wfb, z)
        torch.cuda.synchronize()
        snaps.append(t.cpu().numpy().copy())
    assert np.array_equal(snaps[2].view(np.uint32), ref.view(np.uint32))
    assert not np.array_equal(snaps[0], snaps[1])  # frame 0 really was on the device after its wait

    # (2) an owned buffer whose device pointer is handed out mid-stream: pending frames are submitted
    fb2 = adl.Buffer(device, dim * dim, adl.float4)
    launch(tb, mb, fb2, 0)
    launch(tb, mb, fb2, 1)
    assert fb2.m_ptr != 0                 # the address as a value: changes nothing
    ptr = fb2.getInternalObject()         # pt_buffer_device_ptr: flushes, switches batching off for fb2
    assert ptr == fb2.m_ptr
    device.waitForCompletion()
    raw = np.empty((dim * dim, 4), np.float32)
    assert lib.pt_buffer_read(fb2._h, raw.ctypes.data_as(ctypes.c_void_p), raw.nbytes, 0, None) == 0
    device.waitForCompletion()
    launch(tb, mb, fb2, 2)
    out = torch.empty((dim * dim, 4), dtype=torch.float32, device="cuda")
    src = adl.Buffer(dtype=adl.float4)
    src.setRawPtr(device, ptr, dim * dim)
    dst = adl.Buffer(dtype=adl.float4)
    dst.setRawPtr(device, out.data_ptr(), dim * dim)
    dst.write(src, dim * dim)
    device.waitForCompletion()
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    for b in (tb, mb, fb, wfb, fb2, src, dst):
        b.release()


def test_fill_kernel_and_events(device):
    from oclpathtracer_amd import adl

    n = 1000
    b = adl.Buffer(device, n, np.int32)
    k = device.getKernel("../test/PtShimTest", "FillKernel")
    assert k is not None
    assert device.getKernel("../test/ClKernels/NoSuchKernel", "Nope") is None  # reference: returns 0
    assert device.getKernel("../test/ClKernels/GenerateColors.cl", "GenerateColors") is not None  # extension ok
    launcher = adl.Launcher(device, k)
    launcher.setBuffers([adl.BufferInfo(b)])
    launcher.setConst(np.int32(42))
    sync = adl.SyncObject(device)
    launcher.launch1D(n, 64, sync)
    sync.waitForCompletion()
    assert sync.isComplete() and sync.getExecutionTimeNanoseconds() >= 0
    out = np.zeros(n, np.int32)
    b.read(out, n)
    device.waitForCompletion()
    assert np.all(out == 42)
    sync.release()
    b.release()


def test_error_behaviour(device, cornell):
    from oclpathtracer_amd import adl, shim
    from oclpathtracer_amd.render import upload_scene

    lib = shim.load()
    b = adl.Buffer(device, 16, np.int32)
    with pytest.raises(shim.ShimError) as e:
        b.read(np.zeros(64, np.int32), 64)  # past the end
    assert e.value.code == shim.PT_ERR_RANGE
    # wrong argument list for GenerateColors
    k = device.getKernel("GenerateColors", "GenerateColors")
    launcher = adl.Launcher(device, k)
    launcher.setBuffers([adl.BufferInfo(b)])
    with pytest.raises(shim.ShimError) as e:
        launcher.launch1D(64)
    assert e.value.code == shim.PT_ERR_ARGS
    # a null kernel is reported, not dereferenced (the reference would crash)
    with pytest.raises(shim.ShimError) as e:
        adl.Launcher(device, None).launch1D(64)
    assert e.value.code == shim.PT_ERR_NOT_FOUND
    # framebuffer too small for the launch
    tris, mats = cornell
    tb, mb = upload_scene(device, tris, mats)
    small = adl.Buffer(device, 16, adl.float4)
    launcher = adl.Launcher(device, k)
    launcher.setBuffers([adl.BufferInfo(tb), adl.BufferInfo(mb), adl.BufferInfo(small)])
    launcher.setConst(np.array([64, 64, 0, 0], np.int32))
    with pytest.raises(shim.ShimError) as e:
        launcher.launch1D(64 * 64)
    assert e.value.code == shim.PT_ERR_RANGE
    # invalid render parameters
    p = shim.RenderParams()
    p.width, p.height, p.frame_count, p.max_bounces = 8, 8, 1, 0
    p.num_triangles, p.num_materials, p.stripe_rows, p.n_ranks, p.rank = 36, 18, 1, 1, 0
    fb = adl.Buffer(device, 64, adl.float4)
    assert lib.pt_render_frames(device._h, tb._h, mb._h, fb._h, ctypes.byref(p), None, None) == shim.PT_ERR_INVALID
    p.max_bounces, p.rank = 16, 3
    assert lib.pt_render_frames(device._h, tb._h, mb._h, fb._h, ctypes.byref(p), None, None) == shim.PT_ERR_INVALID
    # an OOM allocation follows the reference: size 0, null pointer, no exception
    huge = adl.Buffer(device, 1 << 50, np.uint8)
    assert huge.getSize() == 0 and huge.m_ptr == 0
    # destroying a device with live buffers is refused (the reference asserts used-memory == 0)
    for x in (b, tb, mb, small, fb):
        x.release()


def test_launch_profile_return_time(device, cornell):
    """Device::toggleProfiling(PROFILE_RETURN_TIME): launch returns its duration in ms."""
    from oclpathtracer_amd import adl
    from oclpathtracer_amd.render import upload_scene

    tris, mats = cornell
    tb, mb = upload_scene(device, tris, mats)
    fb = adl.Buffer(device, 128 * 128, adl.float4)
    device.toggleProfiling(adl.Device.PROFILE_RETURN_TIME)
    try:
        launcher = adl.Launcher(device, device.getKernel("GenerateColors", "GenerateColors"))
        launcher.setBuffers([adl.BufferInfo(tb), adl.BufferInfo(mb), adl.BufferInfo(fb)])
        launcher.setConst(np.array([128, 128, 0, 0], np.int32))
        ms = launcher.launch1D(128 * 128)
        assert 0.0 < ms < 1000.0
    finally:
        device.toggleProfiling(adl.Device.PROFILE_NON)
        for x in (tb, mb, fb):
            x.release()


def test_ragged_launch_is_guarded(device, cornell):
    """The reference kernel has no gid < W*H guard and rounds the NDRange up to a multiple of 64
    (Adl/CL/AdlKernelUtilsCL.cpp:461-468): W*H % 64 != 0 writes out of bounds there.  Here the tail
    is guarded: a 40x24 image (960 px) launched with 960 work-items touches exactly 960 pixels."""
    from oclpathtracer_amd import adl
    from oclpathtracer_amd.render import upload_scene

    tris, mats = cornell
    W, H = 40, 24
    tb, mb = upload_scene(device, tris, mats)
    fb = adl.Buffer(device, W * H + 64, adl.float4)
    sentinel = np.full((W * H + 64, 4), -7.0, np.float32)
    fb.write(sentinel, W * H + 64)
    for z in range(5):
        launcher = adl.Launcher(device, device.getKernel("GenerateColors", "GenerateColors"))
        launcher.setBuffers([adl.BufferInfo(tb), adl.BufferInfo(mb), adl.BufferInfo(fb)])
        launcher.setConst(np.array([W, H, z, 0], np.int32))
        launcher.launch1D(W * H)
    out = np.empty_like(sentinel)
    fb.read(out, W * H + 64)
    device.waitForCompletion()
    assert np.all(out[W * H:] == -7.0)
    assert_fb_equal(out[: W * H], np.load(os.path.join(GOLDEN, "cornell_40x24_f5_d16.npy")), "ragged")
    for x in (tb, mb, fb):
        x.release()


def test_cpp_harness_matches_golden(tmp_path):
    """The C++ RaytraceTest-shaped harness (own process, pt_adl.hpp facade): all fixture tests pass
    and the RayCast framebuffer equals the golden oracle image, with and without frame batching."""
    exe = os.path.join(ROOT, "oclpathtracer_amd", "raytrace_test")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    scene_path = os.path.join(ROOT, "oclpathtracer_amd", "data", "cornellbox.bin")
    want = np.load(os.path.join(GOLDEN, "cornell_64x64_f8_d16.npy"))
    for extra in ([], ["--no-batch"]):
        dump = str(tmp_path / "fb.raw")
        r = subprocess.run([exe, "--dim", "64", "--frames", "8", "--scene", scene_path, "--out-dir", str(tmp_path),
                            "--dump", dump] + extra, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert r.stdout.count("[       OK ]") == 7 and "FAILED" not in r.stdout
        got = np.fromfile(dump, np.float32).reshape(-1, 4)
        assert_fb_equal(got, want, "C++ harness " + " ".join(extra))
    ppm = [f for f in os.listdir(tmp_path) if f.endswith(".ppm")]
    assert len(ppm) == 1 and ppm[0].startswith("rayCastAo_")
    from oclpathtracer_amd import scene
    toks = open(os.path.join(tmp_path, ppm[0])).read().split()
    assert toks[:4] == ["P3", "64", "64", "255"]
    assert np.array_equal(np.array(toks[4:], np.int64).reshape(-1, 3), scene.f2c(want[:, :3]))


def test_device_transcendentals_match_oracle_bitwise(device, oracle):
    """PTSPEC sin/cos/pow as the kernels evaluate them, directly against the CPU oracle: 4M inputs
    each, every bit (NaN masks aside) must agree.  sin/cos over phi = TWO_PI*xi for the RNG's whole
    output lattice region plus the ends; pow over 60 decades, denormals, 0, inf, negatives, NaN."""
    from oclpathtracer_amd import adl

    rng = np.random.default_rng(11)
    n = 1 << 22
    phi = (np.float32(6.28318530718) * rng.random(n, dtype=np.float32)).astype(np.float32)
    phi[:4] = [0.0, np.float32(6.28318530718), np.float32(3.14159274), np.float32(1.57079637)]
    # arguments outside [0, 2 pi] take PTSPEC's binary64 evaluation: exercise it too
    phi[4:1028] = rng.uniform(6.2832, 300.0, 1024).astype(np.float32)
    phi[1028:1030] = [np.float32(6.283186), np.float32(6.2831864)]  # the binary32 path's last angle, the first beyond it
    xs = np.concatenate([
        (10.0 ** rng.uniform(-44, 38, n // 2)).astype(np.float32),
        rng.random(n // 2 - 16, dtype=np.float32) * np.float32(100.0),
        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -1.0, 1.0, 1e-45, 1.17549435e-38, 3.4028235e38,
                  0.45, 90.0, 2.0, 0.5, 1.0000001, 0.99999994], np.float32)]).astype(np.float32)
    k = device.getKernel("PtShimTest", "MathKernel")
    assert k is not None
    for inp, cols in ((phi, (0, 1)), (xs, (2, 3))):
        src = adl.Buffer(device, n, np.float32)
        dst = adl.Buffer(device, 4 * n, np.float32)
        try:
            src.write(inp, n)
            launcher = adl.Launcher(device, k)
            launcher.setBuffers([adl.BufferInfo(src, True), adl.BufferInfo(dst)])
            launcher.launch1D(n)
            out = np.empty(4 * n, np.float32)
            dst.read(out, 4 * n)
            device.waitForCompletion()
        finally:
            src.release()
            dst.release()
        out = out.reshape(n, 4)
        if cols == (0, 1):
            s, c = oracle.sincos(inp)
            want = (s, c)
        else:
            with np.errstate(all="ignore"):
                want = (oracle.pow_array(inp, 2.2), oracle.pow_array(inp, float(np.float32(1.0) / np.float32(2.2))))
        for col, w in zip(cols, want):
            g = out[:, col]
            gn, wn = np.isnan(g), np.isnan(w)
            assert np.array_equal(gn, wn)
            assert np.array_equal(g.view(np.uint32)[~gn], w.view(np.uint32)[~wn]), "column %d" % col


def test_progressive_driver_matches_one_shot(device, cornell, oracle):
    """resumable accumulation + double-buffered asynchronous readback.  Snapshots
    arrive in order, never block the producer, and every one equals a one-shot render of its frames --
    the GPU's own AND the CPU oracle's (GenerateColors.cl:314-321 makes a pixel a function of its frames alone)."""
    from oclpathtracer_amd.progressive import ProgressiveRenderer
    from oclpathtracer_amd.render import Renderer

    tris, mats = cornell
    W, H, per_step, steps = 96, 64, 5, 6
    prog = ProgressiveRenderer(device, tris, mats, W, H, frames_per_step=per_step)
    snaps = {}
    try:
        assert prog.latest() is None
        for _ in range(steps):
            prog.step()
            got = prog.latest()
            if got is not None:
                snaps[got[0]] = got[1].copy()
        frames, img = prog.latest(block=True)
        snaps[frames] = img.copy()
        assert frames == per_step * steps == prog.frames_done
    finally:
        prog.release()
    assert len(snaps) >= 1 and all(f % per_step == 0 for f in snaps)
    for f, img in sorted(snaps.items()):
        r = Renderer(device, tris, mats, W, H)
        try:
            r.render(f)
            want = r.read()
        finally:
            r.release()
        assert_fb_equal(img, want, "progressive snapshot at %d frames" % f)
        assert_fb_equal(img, oracle.render(tris, mats, W, H, f), "progressive snapshot at %d frames vs the oracle" % f)
