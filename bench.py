#!/usr/bin/env python3
"""bench.py -- Msamples/s of the path-tracing hot path on N MI355X (BASELINE.json metric).

A "step" is one complete render of the workload: cornellbox.bin, 1024 x 1024, 256 spp
(frames 0..255), depth 16 (BASELINE.json configs[2]) -- every rank renders its image stripes
with the fused HIP path, rank 0 gathers the stripes (RCCL over xGMI) and assembles the image.
The image is fixed as N grows: "scaling": "strong".

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (pt_trace_kernel, timed with
HIP events on its own stream inside the timed region); `cpu_baseline` times the CPU oracle
("port": this repo's restatement of the reference kernel -- the reference has no CPU executor)
on a bounded sample of the same workload on rank 0's host cores, at N=1 only.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_VALU_TFLOPS = 157.3   # MI355X_MICROARCH.md chip table: peak FP32 vector (= the dense f32 MFMA peak)
PEAK_HBM_GBS = 8000.0           # HBM3E spec peak
# algorithmic flop weights, SURVEY.md S8(d)
F_GEN, F_ACC = 60.0, 36.0
F_TRI = {"cull": 20.0, "rej_u": 30.0, "rej_v": 46.0, "reach_t": 52.0}
F_ACCEPT, F_SHADE_DIFFUSE, F_SHADE_SPECULAR = 33.0, 120.0, 160.0
BYTES_PER_SAMPLE = 32.0         # 16 B read + 16 B write of one float4 pixel (GenerateColors.cl:314-321)


def pmc_traffic():
    """HBM bytes per sample from the newest committed rocprofv3 PMC passes (profiles/*/pmc_traffic.json;
    FETCH_SIZE and WRITE_SIZE collected in separate passes by tools/gpu_pmc.sh and corrected as
    MI355X_MICROARCH.md prescribes).  Counters cannot be read inside the timed run, so the measured
    per-sample figure is scaled to this launch; None when no PMC summary is committed."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        d["file"] = os.path.relpath(files[-1], ROOT)
        return d
    except (OSError, ValueError):
        return None


def flops_per_ray_from_tallies(st: dict) -> float:
    """Outcome-weighted algorithmic flops per traced ray (intersectWorld + shading)."""
    rays = max(st["rays"], 1)
    tri = sum(F_TRI[k] * st[k] for k in F_TRI) + F_ACCEPT * st["accept"]
    shade = F_SHADE_DIFFUSE * st["shade_diffuse"] + F_SHADE_SPECULAR * st["shade_specular"]
    return (tri + shade) / rays


def host_cores() -> int:
    """CPU threads this process may actually use: affinity mask, capped by the cgroup quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            n = min(n, max(1, q // p))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(tris, mats, depth, target_seconds=12.0):
    """Time the CPU oracle (all host cores) on a bounded 256x256 sample of the workload."""
    from oracle import ptoracle

    ptoracle.build()
    cores = host_cores()
    W = H = 256
    t0 = time.perf_counter()
    ptoracle.render(tris, mats, W, H, 2, max_bounces=depth, nthreads=cores)  # warm-up + calibration
    dt = max(time.perf_counter() - t0, 1e-3)
    frames = int(max(4, min(4096, target_seconds / (dt / 2))))
    t0 = time.perf_counter()
    _, st = ptoracle.render(tris, mats, W, H, frames, max_bounces=depth, nthreads=cores, want_stats=True)
    dt = time.perf_counter() - t0
    return {
        "value": W * H * frames / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": "oracle/pt_oracle.c, cornellbox %dx%d x %d frames depth %d, %d threads, %.1f s" % (W, H, frames, depth, cores, dt),
    }, st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--depth", type=int, default=16)
    ap.add_argument("--stripe-rows", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=0, help="trace kernel: 0 library default, 1 lane-regenerating, 2 octant-sorted")
    ap.add_argument("--quad-filter", type=int, default=0, help="PT_OPT_QUAD_FILTER (A/B timing): 0 strongest, 1 none, 2 pairs, 3 shared u, 4 packed")
    ap.add_argument("--accel", type=int, default=0, help="PT_OPT_ACCEL: 0 auto (LBVH from 512 triangles), 1 brute force, 2 LBVH")
    ap.add_argument("--soup", type=int, default=0, help="render the synthetic N-triangle soup of BASELINE configs[4] (Cornell box + "
                    "N-36 small triangles) instead of cornellbox.bin; exploration only: no roofline / cpu_baseline objects")
    args = ap.parse_args()

    import torch  # first: the shim must bind to the HIP runtime torch already loaded
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from oclpathtracer_amd import adl, scene, shim
    from oclpathtracer_amd.distributed import StripeImage

    tris, mats = scene.make_soup(args.soup) if args.soup else scene.load_model()
    W, H, spp, depth = args.width, args.height, args.spp, args.depth

    assert adl.init(adl.TYPE_HIP), "adl.init failed"
    dev = adl.DeviceUtils.allocate(adl.TYPE_HIP, adl.Config(local_rank))
    lib = shim.load()
    dev.setOption(shim.PT_OPT_TRACE_VARIANT, args.variant)
    dev.setOption(shim.PT_OPT_QUAD_FILTER, args.quad_filter)
    dev.setOption(shim.PT_OPT_ACCEL, args.accel)
    img = StripeImage(dev, tris, mats, W, H, world=world, rank=rank, stripe_rows=args.stripe_rows, want_stats=True)

    def step():
        img.render(spp, frame_begin=0, max_bounces=depth)
        img.gather()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    img.reset_stats()
    shim.check(lib.pt_profile_enable(dev._h, 1))
    shim.check(lib.pt_profile_reset(dev._h))
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-launch duration of the dominant kernel on this rank, and its work counters
    tot_ms, launches = ctypes.c_double(), ctypes.c_uint64()
    shim.check(lib.pt_profile_query(dev._h, shim.PT_PROF_TRACE, ctypes.byref(tot_ms), ctypes.byref(launches)))
    fold_ms, fold_n = ctypes.c_double(), ctypes.c_uint64()
    shim.check(lib.pt_profile_query(dev._h, shim.PT_PROF_FOLD, ctypes.byref(fold_ms), ctypes.byref(fold_n)))
    st = img.read_stats()
    counters = torch.tensor([st["samples"], st["rays"]], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)
    total_samples, total_rays = float(counters[0].item()), float(counters[1].item())

    if rank == 0:
        samples_per_step = W * H * spp
        assert total_samples == samples_per_step * args.steps, (total_samples, samples_per_step * args.steps)
        value = samples_per_step * args.steps / dt / 1e6
        out = {
            "metric": "Msamples/sec (pixels x spp / s), cornellbox.bin %dx%d" % (W, H),
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (cornellbox.bin scene, seeded per-pixel RNG of the reference)",
            "config": {"workload": "cornellbox.bin %dx%d, %d spp, depth %d, full path (BASELINE configs[2])" % (W, H, spp, depth),
                       "sharding": "%d-row image stripes round-robin over %d rank(s), RCCL gather to rank 0" % (args.stripe_rows, world),
                       "rays_per_sample": total_rays / total_samples},
        }
        if args.soup:
            out["metric"] = "Msamples/sec (pixels x spp / s), %d-triangle soup %dx%d" % (len(tris), W, H)
            out["config"]["workload"] = "soup of %d triangles (BASELINE configs[4] generator) %dx%d, %d spp, depth %d, accel %d" % (
                len(tris), W, H, spp, depth, args.accel)
            out["roofline"] = None
            out["kernels"] = {"pt_trace_kernel_ms_total": tot_ms.value, "pt_trace_kernel_launches": int(launches.value),
                              "pt_fold_kernel_ms_total": fold_ms.value}
            print(json.dumps(out))
            sys.stdout.flush()
            img.release()
            adl.DeviceUtils.deallocate(dev)
            if world > 1:
                dist.barrier()
                dist.destroy_process_group()
            return
        cpu, tallies = (None, None)
        if world == 1 and not args.no_cpu_baseline:
            cpu, tallies = cpu_baseline(tris, mats, depth)
        # roofline of pt_trace_kernel on rank 0: algorithmic work of ONE launch / its mean duration
        n_launch = max(int(launches.value), 1)
        avg_ms = tot_ms.value / n_launch
        rank_samples = st["samples"] / n_launch
        rank_rays = st["rays"] / n_launch
        if tallies is not None:
            fpr = flops_per_ray_from_tallies(tallies)
            basis = "outcome-weighted (SURVEY S8d) from the cpu_baseline sample's tallies"
        else:
            # no CPU leg in this run (N > 1 or --no-cpu-baseline): outcome mix of the committed golden
            # tallies (tests/golden/work_counters.json, oracle, same scene and depth cap 16)
            try:
                with open(os.path.join(ROOT, "tests", "golden", "work_counters.json")) as f:
                    fpr = flops_per_ray_from_tallies(json.load(f)["cornell_64x64_f8_d16"])
                basis = "outcome-weighted (SURVEY S8d) from tests/golden/work_counters.json (cornell_64x64_f8_d16)"
            except (OSError, KeyError, ValueError):
                fpr = 36 * 52.0 + 130.0
                basis = "fallback tri_tests x 52 + 130 shading"
        flops = rank_samples * (F_GEN + F_ACC) + rank_rays * fpr
        tfl = flops / (avg_ms * 1e-3) / 1e12
        pmc = pmc_traffic()
        tr_traffic = fo_traffic = None
        if pmc is not None:
            tr_traffic = pmc["pt_trace_kernel"]["hbm_bytes_per_sample"] * rank_samples
            fo_traffic = pmc["pt_fold_kernel"]["hbm_bytes_per_sample"] * rank_samples
        out["roofline"] = {
            "bound": "valu", "achieved": tfl, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s", "frac": tfl / PEAK_FP32_VALU_TFLOPS,
            "traffic": tr_traffic, "kernel": "pt_trace_kernel", "avg_launch_ms": avg_ms, "launches": n_launch,
            "algorithmic_flops_per_launch": flops, "flops_per_ray": fpr, "flops_basis": basis,
            "traffic_basis": (pmc["file"] + ": measured HBM bytes/sample x samples of this launch") if pmc else None,
            "note": "FP32 vector-ALU bound, no MFMA (no dense contraction); the f32 MFMA peak equals the VALU peak on gfx950",
        }
        gbs = rank_samples * BYTES_PER_SAMPLE / (avg_ms * 1e-3) / 1e9
        out["roofline_hbm"] = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                               "traffic": (tr_traffic + fo_traffic) if pmc else None,
                               "algorithmic_bytes_per_sample": BYTES_PER_SAMPLE,
                               "note": "reference semantics: 16 B read + 16 B write of the pixel per sample; scene is 3.4 KB; "
                                       "traffic = trace (radiance stores) + fold (radiance reads) kernels"}
        out["kernels"] = {"pt_trace_kernel_ms_total": tot_ms.value, "pt_fold_kernel_ms_total": fold_ms.value,
                          "pt_fold_kernel_launches": int(fold_n.value)}
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["gpu_over_cpu"] = value / cpu["value"]
        print(json.dumps(out))
        sys.stdout.flush()

    img.release()
    adl.DeviceUtils.deallocate(dev)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
