"""GPU tests of the N-rank path: one process per rank, image stripes with GLOBAL pixel ids,
gather to rank 0, device-side assembly -- and of the stream hand-over between the shim and torch.

* ``test_stripe_image_is_ordered_against_torch_without_host_sync`` (1 GPU): results are consumed by torch ops
  with no host synchronisation in between (what a collective does).
* ``test_world2_processes_share_one_gpu_gloo`` (1 GPU): two rank PROCESSES on the same MI355X, the product's
  StripeImage in each, the slabs gathered over gloo through the host (``gather_slabs``' rehearsal branch),
  assembled by ``pt_assemble_stripes`` on rank 0 -- everything of the N-rank path except RCCL itself.
* ``test_world2_nccl`` (needs >= 2 devices, skips otherwise): the same with backend nccl (= RCCL) over xGMI.
* ``test_bench_starts_its_own_ranks`` (1 GPU): ``python bench.py --gpus 2 --rehearse`` with no launcher
  environment starts two ranks itself and prints rank 0's JSON line.
All compare with the one-process image bit for bit.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_fb_equal

pytestmark = pytest.mark.gpu

W, H, FRAMES, STRIPE = 256, 192, 6, 16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _single_gpu_image(device, cornell):
    from oclpathtracer_amd.render import Renderer

    tris, mats = cornell
    r = Renderer(device, tris, mats, W, H)
    try:
        r.render(FRAMES)
        return r.read()
    finally:
        r.release()


def test_stripe_image_is_ordered_against_torch_without_host_sync(device, cornell):
    """StripeImage runs the shim on its own torch stream; ``ready()`` / ``gather()`` make torch's current stream
    wait for it with an event.  A heavy render (1024x1024 x 32 frames, ~5 ms of GPU time) is consumed by a torch
    op enqueued immediately after, and overwritten by the next render right after that: without the two
    waits the copy would read a half-written framebuffer (or the second render would clobber it early)."""
    import torch

    from oclpathtracer_amd.distributed import StripeImage
    from oclpathtracer_amd.render import Renderer

    tris, mats = cornell
    w = h = 1024
    r = Renderer(device, tris, mats, w, h)
    try:
        r.render(32)
        want32 = r.read()
        r.render(32, frame_begin=32)
        want64 = r.read()
    finally:
        r.release()
    img = StripeImage(device, tris, mats, w, h)
    try:
        img.render(32, frame_begin=0)
        snap32 = img.ready().clone()           # torch op on the current stream, no host sync before it
        img.render(32, frame_begin=32)          # must not overwrite .local before the clone has read it
        snap64 = img.gather().clone()
        torch.cuda.synchronize()
        assert_fb_equal(snap32.cpu().numpy().reshape(-1, 4), want32, "first 32 frames via torch, no host sync")
        assert_fb_equal(snap64.cpu().numpy().reshape(-1, 4), want64, "64 frames via torch, no host sync")
    finally:
        img.release()


def _run_world(world, backend, tmp_path, ndev_needed, extra=(), geom=None):
    port = _free_port()
    out = str(tmp_path / "image.npy")
    w, h, frames, stripe = geom or (W, H, FRAMES, STRIPE)
    procs = []
    for rank in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), backend, out,
                                       str(w), str(h), str(frames), str(stripe)] + list(extra), env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode("utf-8", "replace"))
    for rank, (p, log) in enumerate(zip(procs, logs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, log[-3000:])
    return np.load(out)


def test_world2_processes_share_one_gpu_gloo(device, cornell, tmp_path):
    want = _single_gpu_image(device, cornell)
    got = _run_world(2, "gloo", tmp_path, 1)
    assert_fb_equal(got.reshape(-1, 4), want, "world-2 (gloo through the host) vs one process")


def test_world2_pipelined_gathers(device, cornell, tmp_path):
    """StripeImage(pipelined=True), the loop bench.py --gpus N times: every gather is issued after the NEXT render has
    been enqueued into the other framebuffer; the last gather must deliver the last image (gloo through the host on a
    one-GPU box; RCCL when two devices are visible)."""
    from oclpathtracer_amd import shim

    want = _single_gpu_image(device, cornell)
    two = shim.load().pt_device_count() >= 2
    got = _run_world(2, "nccl" if two else "gloo", tmp_path, 2 if two else 1, extra=("pipelined",))
    assert_fb_equal(got.reshape(-1, 4), want, "world-2 pipelined gathers vs one process")


def test_world3_pipelined_with_a_ragged_last_period(device, cornell, tmp_path):
    """Three ranks, 4-row stripes (bench.py's), a height that leaves the last period of stripes incomplete (200 = 16 x 12 + 8:
    rank 0 and 1 own one stripe more than rank 2, whose slab is zero-padded in the gather), the pipelined loop, the assembly
    on its own stream.  gloo through the host when fewer than three devices are visible."""
    from oclpathtracer_amd import shim
    from oclpathtracer_amd.render import Renderer

    w, h, frames, stripe = 192, 200, 5, 4
    tris, mats = cornell
    r = Renderer(device, tris, mats, w, h)
    try:
        r.render(frames)
        want = r.read()
    finally:
        r.release()
    three = shim.load().pt_device_count() >= 3
    got = _run_world(3, "nccl" if three else "gloo", tmp_path, 3 if three else 1, extra=("pipelined",), geom=(w, h, frames, stripe))
    assert_fb_equal(got.reshape(-1, 4), want, "world-3 pipelined, ragged stripes vs one process")


def test_configs3_geometry_gather_and_assembly_rehearsal(device, cornell, oracle, tmp_path):
    """BASELINE configs[3]'s image -- 2048 x 2048, 64 MiB of float4 -- through the whole N-rank path at full size: two rank
    processes (sharing the one MI355X: gloo through the host; RCCL with two devices) render their 4-row stripes, gather
    32 MiB slabs and assemble.  Few frames (the arithmetic at full spp is test_configs3_one_rank_of_eight_full_size's
    business); 24 sampled rows against the oracle, the w lane and the finiteness of the whole image."""
    from oclpathtracer_amd import shim

    w = h = 2048
    frames, stripe = 3, 4
    two = shim.load().pt_device_count() >= 2
    got = _run_world(2, "nccl" if two else "gloo", tmp_path, 2 if two else 1, extra=("pipelined",), geom=(w, h, frames, stripe))
    assert got.shape == (h, w, 4)
    assert np.all(got[..., 3] == 1.0) and np.isfinite(got).all()
    tris, mats = cornell
    rows = np.unique(np.concatenate([np.array([0, 3, 4, 7, 8, h - 1, h - 4, h - 5]), np.random.default_rng(3).integers(0, h, 16)]))
    for row in rows:
        fb = np.zeros((h * w, 4), np.float32)
        oracle.render(tris, mats, w, h, frames, fb=fb, gid_begin=int(row) * w, gid_count=w)
        assert_fb_equal(got[row].reshape(-1, 4), fb[row * w:(row + 1) * w], "2048^2 two-rank image, row %d" % row)


def test_world4_pipelined_configs2_geometry(device, cornell, tmp_path):
    """Four rank processes on the one MI355X (the box admits six processes on its card; eight ranks are the driver's to start):
    BASELINE configs[2]'s image, bench.py's 4-row stripes, the pipelined loop -- every rank with its own device handle, staging
    ring and render lanes beside the others' -- against the one-process image."""
    from oclpathtracer_amd.render import Renderer

    w, h, frames, stripe = 1024, 1024, 3, 4
    tris, mats = cornell
    r = Renderer(device, tris, mats, w, h)
    try:
        r.render(frames)
        want = r.read()
    finally:
        r.release()
    got = _run_world(4, "gloo", tmp_path, 1, extra=("pipelined",), geom=(w, h, frames, stripe))
    assert_fb_equal(got.reshape(-1, 4), want, "world-4 pipelined at 1024 x 1024 vs one process")


@pytest.mark.parametrize("config,extra", [(3, ["--spp", "2"]), (4, ["--spp", "1"])])
def test_bench_rehearses_four_ranks_on_the_named_multi_gpu_configs(device, config, extra):
    """bench.py --gpus 4 --rehearse on BASELINE configs[3] (2048 x 2048) and configs[4] (the 10^6-triangle soup through the LBVH, every
    rank building its own hierarchy): the N-rank code path of the two workloads BASELINE names for 8 GPUs, with the line's own record
    of what the collective backend connected (ranks_seen, backend)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rehearse", "--config", str(config), "--steps", "2", "--warmup", "1"] + extra
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode("utf-8", "replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["ranks_seen"] == 4 and d["backend"] == "gloo" and len(d["devices_seen"]) == 4
    assert d["value"] > 0 and "rehearsal" in d


def test_world2_nccl(device, cornell, tmp_path):
    from oclpathtracer_amd import shim

    if shim.load().pt_device_count() < 2:
        pytest.skip("needs two MI355X: RCCL refuses two ranks on one device")
    want = _single_gpu_image(device, cornell)
    got = _run_world(2, "nccl", tmp_path, 2)
    assert_fb_equal(got.reshape(-1, 4), want, "world-2 (RCCL gather) vs one process")


def test_bench_starts_its_own_ranks(device):
    """`python bench.py --gpus 2` without a launcher environment must start its ranks itself (round-1 review:
    it used to exit).  On a one-GPU box the two ranks share the device (--rehearse: gloo gather through the
    host, flagged in the line as not a measurement); with two devices it is the real thing."""
    from oclpathtracer_amd import shim

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--width", "256", "--height", "256", "--spp", "8"]
    if shim.load().pt_device_count() < 2:
        cmd.append("--rehearse")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode("utf-8", "replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0
    assert d["config"]["rays_per_sample"] > 1.0


def test_bench_line_keeps_the_drivers_contract(device):
    """The one JSON line `python bench.py` prints on one GPU: every field the driver and the judge read, with the types they expect
    (a small image so that the CPU leg and the extra legs stay short; the numbers themselves are the record run's business)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-extra-configs"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode("utf-8", "replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "alloc_ms", "ranks_seen"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "Msamples/s" and d["dtype"] == "f32" and d["scaling"] in ("weak", "strong")
    assert "configs[2]" in d["config"]["workload"] and "model" not in d["config"]
    assert abs(d["value"] - 1024 * 1024 * 256 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma", "valu") and 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["unit"] == "TFLOP/s"
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "Msamples/s" and c["sample"]
    assert d["config"]["workspace_bytes"] <= 512 * 1000 * 1000 and d["config"]["staging_ring_bytes"] == 2 * 192 * (1 << 20)
