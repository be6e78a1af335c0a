#!/usr/bin/env python3
"""bench.py -- Msamples/s of the path-tracing hot path on N MI355X (BASELINE.json metric).

A "step" is one complete render of the workload: cornellbox.bin, 1024 x 1024, 256 spp
(frames 0..255), depth 16 (BASELINE.json configs[2]) -- every rank renders its image stripes
with the fused HIP path, rank 0 gathers the stripes (RCCL over xGMI) and assembles the image.
The image is fixed as N grows: "scaling": "strong".

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started WITHOUT a torch.distributed launcher (no RANK in the environment) and with --gpus N > 1,
this process starts the N ranks itself -- N fresh interpreters, created before torch or the HIP
runtime is loaded here -- relays rank 0's JSON line and exits with the worst rank's code.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (pt_trace_kernel, timed with
HIP events on its own stream inside the timed region); `cpu_baseline` times the CPU oracle
("port": this repo's restatement of the reference kernel -- the reference has no CPU executor)
on a bounded sample of the same workload on rank 0's host cores, at N=1 only.  `extra.configs`
(N=1 only) carries the other single-GPU BASELINE configurations, timed in the same process
after the headline region: configs[1] (512^2 x 64 spp, depth 2) and configs[4]'s scene
(10^6-triangle soup, 1024^2 x 256 spp, through the LBVH).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import socket
import subprocess
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
# LBVH search (configs[4]): per eight-child node entered, eight slab tests (6 FMA = 12 flop, 6 min/max, 1 compare = 19
# flop each) + 9 for the node's frame; per triangle tested the reference's full test (52, reach_t) -- DESIGN.md S4
F_BVH_NODE, F_BVH_TRI = 161.0, 52.0
B_BVH_NODE, B_BVH_TRI = 64.0, 48.0   # one 64-byte node record per node entered, 48 bytes of a 64-byte leaf record per triangle tested


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
    # BASELINE configs[0] as it is written -- 256 x 256, ONE frame, depth 16 -- is a 10-ms job on these cores: the median of
    # many repetitions, thread start-up included (SURVEY S8d asked for this line beside the sustained sample)
    reps = []
    for _ in range(21):
        t1 = time.perf_counter()
        ptoracle.render(tris, mats, W, H, 1, max_bounces=16, nthreads=cores)
        reps.append(time.perf_counter() - t1)
    reps.sort()
    c1 = reps[len(reps) // 2]
    return {
        "value": W * H * frames / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": "oracle/pt_oracle.c, cornellbox %dx%d x %d frames depth %d, %d threads, %.1f s" % (W, H, frames, depth, cores, dt),
        "configs0": {"workload": "BASELINE configs[0]: cornellbox.bin 256x256, 1 spp, depth 16 (CPU)", "value": W * H / c1 / 1e6, "unit": "Msamples/s",
                     "ms": c1 * 1e3, "sample": "median of %d one-frame renders, %d threads (thread start-up included)" % (len(reps), cores)},
    }, st


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=(1, 2, 3, 4),
                    help="the BASELINE.json workload, by its index in `configs`: 1 = cornellbox 512^2 x 64 spp depth 2; 2 = cornellbox 1024^2 x 256 spp "
                         "(the headline, default); 3 = cornellbox 2048^2 x 1024 spp (BASELINE: 8 GPUs + gather); 4 = 10^6-triangle soup 1024^2 x 256 spp "
                         "(BASELINE: 8 GPUs).  --width/--height/--spp/--depth/--soup override single values")
    ap.add_argument("--steps", type=int, default=None, help="timed renders (default: a timed region of a few seconds for the chosen workload)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--stripe-rows", type=int, default=4,
                    help="rows per image stripe of the N-rank split (4: the ranks' shares of configs[2] are equal to 1 %%; 16: 2 %%)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the configs[1] / configs[3] / configs[4] legs of extra.configs")
    ap.add_argument("--quad-filter", type=int, default=0, help="PT_OPT_QUAD_FILTER (A/B timing): 0 strongest, 1 none, 4 packed")
    ap.add_argument("--accel", type=int, default=0, help="PT_OPT_ACCEL: 0 auto (LBVH from 512 triangles), 1 brute force, 2 LBVH")
    ap.add_argument("--soup", type=int, default=None, help="render the synthetic N-triangle soup of BASELINE configs[4] (Cornell box + "
                    "N-36 small triangles) instead of cornellbox.bin (no cpu_baseline: the oracle is O(N) per ray)")
    ap.add_argument("--lanes", type=int, default=2, choices=(1, 2),
                    help="PT_OPT_RENDER_LANES (A/B timing): 2 = the next trace launch fills the machine while the current one drains (default), "
                         "1 = every launch waits for the previous one")
    ap.add_argument("--checkpoint", type=int, default=1, choices=(0, 1),
                    help="PT_OPT_CHECKPOINT (A/B timing): 1 = a trace launch ends when its queue is empty and the next one resumes its paths (default), "
                         "0 = every launch runs its paths out")
    ap.add_argument("--chunk-frames", type=int, default=0, help="PT_OPT_CHUNK_FRAMES: cap on the frames per staging chunk (0 = as many as a ring slot holds)")
    ap.add_argument("--staging-mb", type=int, default=0, help="size of the radiance staging ring in MiB (pt_device_reserve_staging; 0 = the default 2 x 192 MiB)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > visible GPUs: ranks share devices (rank %% device_count) and gather over gloo through the host. "
                         "Exercises the N-rank code path on a smaller box; the line says so and is no scaling measurement")
    args = ap.parse_args(argv)
    # the named workloads: (width, height, spp, depth, soup triangles, default steps, default warm-up)
    table = {1: (512, 512, 64, 2, 0, 400, 5), 2: (1024, 1024, 256, 16, 0, 150, 5), 3: (2048, 2048, 1024, 16, 0, 8, 1),
             4: (1024, 1024, 256, 16, 1_000_000, 3, 1)}
    w, h, spp, depth, soup, steps, warm = table[args.config]
    args.named = all(v is None for v in (args.width, args.height, args.spp, args.depth, args.soup))  # exactly BASELINE's configs[k]
    args.width = w if args.width is None else args.width
    args.height = h if args.height is None else args.height
    args.spp = spp if args.spp is None else args.spp
    args.depth = depth if args.depth is None else args.depth
    args.soup = soup if args.soup is None else args.soup
    args.steps = steps if args.steps is None else args.steps
    args.warmup = warm if args.warmup is None else args.warmup
    return args


# ---------------------------------------------------------------------------------------------
# self-launch: N fresh interpreters, started before this process has loaded torch or HIP
# ---------------------------------------------------------------------------------------------
def spawn_ranks(args) -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else None))   # (other ranks print nothing on stdout; stderr is inherited)
    # supervise ALL ranks: a rank that dies at start-up would leave the others waiting in the rendezvous or a collective
    # until the backend's timeout.  On the first failure the rest are stopped and that rank's code is returned.
    out_chunks = []
    import selectors

    sel = selectors.DefaultSelector()
    sel.register(procs[0].stdout, selectors.EVENT_READ)
    failed = None
    open_out = True
    while True:
        if open_out:
            for key, _ in sel.select(timeout=0.2):
                chunk = os.read(key.fileobj.fileno(), 65536)
                if chunk:
                    out_chunks.append(chunk)
                else:
                    sel.unregister(key.fileobj)
                    open_out = False
        else:
            time.sleep(0.2)
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad and failed is None:
            failed = bad
            for p in procs:
                if p.poll() is None:
                    p.terminate()
        if all(c is not None for c in codes) and not open_out:
            break
        if all(c is not None for c in codes) and open_out:
            # drain what rank 0 left in the pipe
            rest = procs[0].stdout.read()
            if rest:
                out_chunks.append(rest)
            break
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    sys.stdout.write(b"".join(out_chunks).decode("utf-8", "replace"))
    sys.stdout.flush()
    if failed:
        sys.stderr.write("bench.py: rank(s) failed first: %s; the other ranks were stopped\n" % ", ".join("rank %d rc %d" % rc for rc in failed))
        return max(abs(c) for _, c in failed) or 1
    return 0


def time_render(img, fence, steps, warmup, spp, depth):
    """W warm-up renders, then K timed ones bracketed by fence(); returns seconds.  (The legs' images have two framebuffers like the
    headline's: consecutive renders overlap where the data allows.)"""
    for _ in range(warmup):
        img.render(spp, frame_begin=0, max_bounces=depth)
        img.gather()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        img.render(spp, frame_begin=0, max_bounces=depth)
        img.gather()
    fence()
    return time.perf_counter() - t0


def bvh_tallies(dev, lib, shim, tris, mats, W, H, depth):
    """Per-ray averages of the LBVH search (nodes entered, triangles tested) from a short tallied render
    (PT_OPT_BVH_TALLY: the same kernel with its work counters switched on; never the timed one)."""
    from oclpathtracer_amd.render import Renderer

    dev.setOption(shim.PT_OPT_BVH_TALLY, 1)
    try:
        r = Renderer(dev, tris, mats, W, H, want_stats=True)
        r.render(16, max_bounces=depth)  # (long enough that the launch's tail, lanes running dry, does not dominate the occupancies)
        st = r.read_stats_raw()
        r.release()
    finally:
        dev.setOption(shim.PT_OPT_BVH_TALLY, 0)
    rays = max(int(st[shim.PT_STAT_RAYS]), 1)
    nodes, tris_ = int(st[shim.PT_STAT_BVH_NODES]), int(st[shim.PT_STAT_BVH_TRIS])
    steps, tsteps = int(st[shim.PT_STAT_BVH_STEPS]), int(st[shim.PT_STAT_BVH_TRI_STEPS])
    return {"nodes_per_ray": nodes / rays, "tris_per_ray": tris_ / rays,
            "node_phase_lane_occupancy": nodes / max(64 * steps, 1), "tri_phase_lane_occupancy": tris_ / max(64 * tsteps, 1),
            "search_lane_occupancy": (nodes + tris_) / max(64 * (steps + tsteps), 1)}


# tools/ubench_gather on the MI355X (profiles/r03/ubench_gather_r03.txt): dependent random reads of 64-byte records, the LBVH search's
# memory pattern with NO arithmetic, 5 waves per SIMD, from an 80 MB table (the 10^6-triangle soup's hierarchy): 75.7 G records/s.
# (Aligned 128-byte records: 86 G/s; a 22 MB table: 114 G/s; an L2-resident one: 262 G/s.)  That, not the HBM figure, is the ceiling
# of a kernel whose every miss is a dependent 64-byte gather.
GATHER_CEILING_RECORDS_PER_S = 75.7e9
GATHER_RECORD_BYTES = 64.0


def soup_pmc():
    """Measured per-ray counters of the LBVH search (newest profiles/*/pmc_traffic_soup.json: rocprofv3 --pmc passes)."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic_soup.json")))
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        k = d.get("pt_trace_bvh_kernel", d)
        return {"l2_misses_per_ray": float(k["l2_misses_per_ray"]), "bytes_beyond_l2_per_ray": float(k["fetched_beyond_l2_bytes_per_ray_as_tallied"]),
                "l2_hit_rate": float(k["l2_hit_rate"]), "valu_instructions_per_ray": float(k.get("valu_instructions_per_ray", 0.0)),
                "file": os.path.relpath(files[-1], ROOT)}
    except (IndexError, OSError, KeyError, ValueError, TypeError):
        return None


def soup_roofline(tally, rays_per_launch, samples_per_launch, avg_ms):
    """(memory roofline, vector-ALU roofline) of the LBVH trace kernel.

    Memory: what the kernel pulls through the L2's miss path -- MEASURED misses per ray (PMC) x the rays of this launch / its
    duration -- against what a pure dependent gather of the same record size and working set sustains on this chip
    (GATHER_CEILING_RECORDS_PER_S).  Both are 64-byte records per second; the object states them in GB/s as the contract asks.
    The fraction is below 1 by construction of its terms (a kernel cannot miss faster than a kernel that does nothing else).
    VALU: algorithmic flops (SURVEY S8d weights) against the FP32 vector peak."""
    dt = avg_ms * 1e-3
    flops = samples_per_launch * (F_GEN + F_ACC) + rays_per_launch * (
        tally["nodes_per_ray"] * F_BVH_NODE + tally["tris_per_ray"] * F_BVH_TRI + F_SHADE_DIFFUSE)
    alg_bytes = rays_per_launch * (tally["nodes_per_ray"] * B_BVH_NODE + tally["tris_per_ray"] * B_BVH_TRI) + samples_per_launch * 12.0
    tfl = flops / dt / 1e12
    pmc = soup_pmc()
    gather_gbs = GATHER_CEILING_RECORDS_PER_S * GATHER_RECORD_BYTES / 1e9
    mem = None
    if pmc is not None:
        miss_per_s = pmc["l2_misses_per_ray"] * rays_per_launch / dt
        achieved = miss_per_s * GATHER_RECORD_BYTES / 1e9
        traffic = pmc["bytes_beyond_l2_per_ray"] * rays_per_launch
        # `peak` and `frac` against the hardware's HBM figure; what a dependent 64-byte gather can reach of it is a
        # separate field.  The per-ray counters come from a COMMITTED rocprofv3 PMC pass (counters cannot be read inside the timed
        # run), scaled by this launch's rays: the record says which file, so a stale one shows
        mem = {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS, "traffic": traffic,
               "kernel": "pt_trace_bvh_kernel", "avg_launch_ms": avg_ms,
               "from_committed_profile": True, "profile": pmc["file"],
               "achieved_basis": "%s: %.1f L2 misses per ray (hit rate %.1f %%) x %.4g rays of this launch / its duration = %.1f G misses/s, 64 bytes each"
                                 % (pmc["file"], pmc["l2_misses_per_ray"], 100.0 * pmc["l2_hit_rate"], rays_per_launch, miss_per_s / 1e9),
               "gather_ceiling_gbs": gather_gbs, "frac_of_gather_ceiling": achieved / gather_gbs,
               "gather_ceiling_basis": "dependent random gather of 64-byte records from an 80 MB table, no arithmetic: %.1f G records/s "
                                       "(tools/ubench_gather, profiles/r03/ubench_gather_r03.txt) -- what this access pattern can reach of the HBM figure"
                                       % (GATHER_CEILING_RECORDS_PER_S / 1e9),
               "algorithmic_bytes_per_launch": alg_bytes,
               "algorithmic_gbs": alg_bytes / dt / 1e9,
               "note": "algorithmic = 64 B per node entered + 48 B per triangle tested + 12 B radiance per sample: most of it is served by "
                       "L1 / L2 (upper tree levels), so it is NOT set against a memory peak; `traffic` = measured bytes beyond the L2 (FETCH_SIZE as "
                       "tallied: 64 B per miss).  Neither the miss path, nor VALU issue (`roofline_valu`, PMC: %.0f wave-instructions per ray), nor the "
                       "L1's bandwidth is saturated alone; the search is balanced between them at five waves per SIMD (profiles/r03/lbvh_bottlenecks.txt)"
                       % pmc["valu_instructions_per_ray"],
               **tally}
    valu = {"bound": "valu", "achieved": tfl, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s", "frac": tfl / PEAK_FP32_VALU_TFLOPS,
            "traffic": None, "kernel": "pt_trace_bvh_kernel", "avg_launch_ms": avg_ms, "algorithmic_flops_per_launch": flops,
            "flops_basis": "per ray: %.1f nodes entered x %.0f + %.2f triangles tested x %.0f + %.0f shading (tallied render, PT_OPT_BVH_TALLY)"
                           % (tally["nodes_per_ray"], F_BVH_NODE, tally["tris_per_ray"], F_BVH_TRI, F_SHADE_DIFFUSE)}
    if mem is None:   # no PMC summary committed: the VALU object is all that can be stated
        valu.update(tally)
        return valu, valu
    return mem, valu


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(args))

    import torch  # first: the shim must bind to the HIP runtime torch already loaded
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if world > ndev and not args.rehearse:
        raise SystemExit("--gpus %d but only %d device(s) visible (use --rehearse to share devices; not a measurement)" % (world, ndev))
    dev_idx = local_rank % ndev
    torch.cuda.set_device(dev_idx)
    rehearsal = world > ndev
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime

        tmo = datetime.timedelta(seconds=180)   # a missing peer fails the rendezvous in minutes, not the backend's default half hour
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_idx), timeout=tmo)

    from oclpathtracer_amd import adl, scene, shim
    from oclpathtracer_amd.distributed import StripeImage

    tris, mats = scene.make_soup(args.soup) if args.soup else scene.load_model()
    W, H, spp, depth = args.width, args.height, args.spp, args.depth

    assert adl.init(adl.TYPE_HIP), "adl.init failed"
    dev = adl.DeviceUtils.allocate(adl.TYPE_HIP, adl.Config(dev_idx))
    lib = shim.load()
    dev.setOption(shim.PT_OPT_QUAD_FILTER, args.quad_filter)
    dev.setOption(shim.PT_OPT_ACCEL, args.accel)
    dev.setOption(shim.PT_OPT_RENDER_LANES, args.lanes)
    dev.setOption(shim.PT_OPT_CHECKPOINT, args.checkpoint)
    dev.setOption(shim.PT_OPT_CHUNK_FRAMES, args.chunk_frames)
    # every device allocation of the renderer happens HERE, before any timed region: the staging ring is sized once per device
    # handle, the scene workspace and the local framebuffers with the StripeImage below; the render loop allocates nothing
    ws0 = int(dev.getWorkspaceMemory())
    t_alloc = time.perf_counter()
    dev.reserveStaging(args.staging_mb << 20)
    torch.cuda.synchronize()
    alloc_ms = (time.perf_counter() - t_alloc) * 1e3
    staging_bytes = int(dev.getWorkspaceMemory()) - ws0
    # two framebuffers per rank: the next step's first trace launch fills the machine while this step's last one runs its paths out, and
    # (N > 1) the RCCL gather of one step runs beside the next step's render (StripeImage(pipelined=True)); every step still renders,
    # gathers and assembles one complete image
    img = StripeImage(dev, tris, mats, W, H, world=world, rank=rank, stripe_rows=args.stripe_rows, want_stats=True, pipelined=True)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(n):
        prev = None
        for _ in range(n):
            slot = img.render(spp, frame_begin=0, max_bounces=depth)
            if prev is not None:
                img.gather(prev)   # the previous step's image: its collective overlaps the render just enqueued
            prev = slot
        if prev is not None:
            img.gather(prev)

    run_steps(args.warmup)
    fence()
    img.reset_stats()
    shim.check(lib.pt_profile_enable(dev._h, 1))
    shim.check(lib.pt_profile_reset(dev._h))
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    ranks_seen, devices_seen = 1, [dev_idx]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # what the collective backend actually connected: every rank reports itself and the device it drives
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        seen = torch.zeros(world, dtype=torch.int64, device="cpu" if rehearsal else "cuda")
        seen[rank] = 1 + torch.cuda.current_device()
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        devices_seen = [int(x) - 1 for x in seen.tolist()]
        ranks_seen = sum(1 for x in devices_seen if x >= 0)
        assert ranks_seen == world, "only %d of %d ranks answered the collective" % (ranks_seen, world)
        if not rehearsal:
            assert len(set(devices_seen)) == world, "ranks share a device: %r" % (devices_seen,)

    # per-launch duration of the dominant kernel on this rank, and its work counters
    tot_ms, launches = ctypes.c_double(), ctypes.c_uint64()
    shim.check(lib.pt_profile_query(dev._h, shim.PT_PROF_TRACE, ctypes.byref(tot_ms), ctypes.byref(launches)))
    fold_ms, fold_n = ctypes.c_double(), ctypes.c_uint64()
    shim.check(lib.pt_profile_query(dev._h, shim.PT_PROF_FOLD, ctypes.byref(fold_ms), ctypes.byref(fold_n)))
    union_ms = ctypes.c_double()
    shim.check(lib.pt_profile_query_union(dev._h, shim.PT_PROF_TRACE, ctypes.byref(union_ms)))
    shim.check(lib.pt_profile_enable(dev._h, 0))
    st = img.read_stats()
    counters = torch.tensor([st["samples"], st["rays"]], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
    if world > 1:
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)
    total_samples, total_rays = float(counters[0].item()), float(counters[1].item())

    if rank == 0:
        samples_per_step = W * H * spp
        assert total_samples == samples_per_step * args.steps, (total_samples, samples_per_step * args.steps)
        value = samples_per_step * args.steps / dt / 1e6
        step_ms = dt / args.steps * 1e3
        out = {
            "metric": "Msamples/sec (pixels x spp / s), cornellbox.bin %dx%d" % (W, H),
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (cornellbox.bin scene, seeded per-pixel RNG of the reference)",
            "config": {"workload": ("BASELINE configs[%d]: " % args.config if args.named else "") +
                                   "cornellbox.bin %dx%d, %d spp, depth %d%s" % (W, H, spp, depth, ", full path" if depth >= 16 else ""),
                       "sharding": "%d-row image stripes round-robin over %d rank(s), RCCL gather to rank 0%s" % (
                           args.stripe_rows, world, " (two framebuffers per rank: the gather of a step runs beside the next step's render)" if world > 1 else " (two framebuffers: consecutive steps overlap)"),
                       "rays_per_sample": total_rays / total_samples, "timed_region_s": dt,
                       "render_lanes": args.lanes, "checkpointed_launches": bool(args.checkpoint)},
            "ranks_seen": ranks_seen, "devices_seen": devices_seen,
            "backend": (dist.get_backend() if world > 1 else None),
            "alloc_ms": alloc_ms,
        }
        out["config"]["staging_ring_bytes"] = staging_bytes
        out["config"]["workspace_bytes"] = int(dev.getWorkspaceMemory())
        if rehearsal:
            out["rehearsal"] = "%d ranks share %d device(s), gloo gather through the host: code-path check, NOT a scaling measurement" % (world, ndev)
        n_launch = max(int(launches.value), 1)
        avg_ms = tot_ms.value / n_launch
        rank_samples = st["samples"] / n_launch
        rank_rays = st["rays"] / n_launch
        out["kernels"] = {"pt_trace_kernel_ms_total": tot_ms.value, "pt_trace_kernel_launches": n_launch,
                          "pt_trace_kernel_ms_union": union_ms.value,
                          "pt_fold_kernel_ms_total": fold_ms.value, "pt_fold_kernel_launches": int(fold_n.value),
                          "note": "total = sum of the launches' [start, stop] durations (what rocprofv3 --stats averages); union = time during which at "
                                  "least one trace launch was executing: with two render lanes launch c+1 starts while launch c drains, so the sum "
                                  "counts that overlap twice"}
        cpu = None
        if args.soup:
            out["metric"] = "Msamples/sec (pixels x spp / s), %d-triangle soup %dx%d" % (len(tris), W, H)
            out["config"]["workload"] = ("BASELINE configs[4]: " if args.named and args.config == 4 else "") + \
                "soup of %d triangles (BASELINE configs[4] generator) %dx%d, %d spp, depth %d, accel %d" % (len(tris), W, H, spp, depth, args.accel)
            if args.accel != 1 and world == 1:
                tally = bvh_tallies(dev, lib, shim, tris, mats, W, H, depth)
                out["roofline"], out["roofline_valu"] = soup_roofline(tally, rank_rays, rank_samples, avg_ms)
            else:
                out["roofline"] = None
        else:
            tallies = None
            if world == 1 and not args.no_cpu_baseline:
                try:
                    cpu, tallies = cpu_baseline(tris, mats, depth)
                except Exception as e:  # noqa: BLE001 -- the GPU measurement stands; the line says what happened to the CPU leg
                    cpu, tallies = {"error": "%s: %s" % (type(e).__name__, e)}, None
            # roofline of pt_trace_kernel on rank 0: algorithmic work of ONE launch / its mean duration
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
                "from_committed_profile": bool(pmc), "profile_head": (pmc or {}).get("head"),
                # with two render lanes consecutive launches overlap (launch c+1 becomes resident while launch c drains): `achieved` and
                # `frac` divide by the launches' own [start, stop] durations -- what rocprofv3 --stats averages -- which count the
                # overlap twice; the *_exclusive pair divides by the time during which at least one launch was executing
                "avg_launch_ms_exclusive": union_ms.value / n_launch,
                "achieved_exclusive": flops / (union_ms.value / n_launch * 1e-3) / 1e12 if union_ms.value > 0 else None,
                "frac_exclusive": flops / (union_ms.value / n_launch * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS if union_ms.value > 0 else None,
                "note": "FP32 vector-ALU bound, no MFMA (no dense contraction); the f32 MFMA peak equals the VALU peak on gfx950",
            }
            # the 32 B/sample of the reference are moved by trace + fold + framebuffer together: over the step
            gbs = (W * H * spp) * BYTES_PER_SAMPLE / (step_ms * 1e-3) / 1e9
            out["roofline_hbm"] = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                                   "traffic": (tr_traffic + fo_traffic) if pmc else None,
                                   "algorithmic_bytes_per_sample": BYTES_PER_SAMPLE,
                                   "note": "reference semantics: 16 B read + 16 B write of the pixel per sample, over the whole "
                                           "step (all ranks); scene is 3.4 KB; traffic = trace (radiance stores) + fold "
                                           "(radiance reads) kernels of rank 0"}
        if world == 1 and not args.soup and not args.no_extra_configs:
            # (the extra legs never cost the headline line: a failure in one of them is recorded in its place)
            out["extra"] = {"configs": extra_configs(dev, lib, shim, scene, adl, fence, args)}
        if cpu is not None:
            out["cpu_baseline"] = cpu
            if "value" in cpu:
                out["gpu_over_cpu"] = value / cpu["value"]
        print(json.dumps(out))
        sys.stdout.flush()

    img.release()
    adl.DeviceUtils.deallocate(dev)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def extra_configs(dev, lib, shim, scene, adl, fence, args):
    """The other single-GPU BASELINE configurations, timed in this process after the headline region
    (same fence / perf_counter bracket)."""
    from oclpathtracer_amd.distributed import StripeImage

    res = {}
    def leg_configs1():
        # configs[1]: cornellbox 512x512, 64 spp, depth cap 2 ("direct lighting only")
        tris, mats = scene.load_model()
        img = StripeImage(dev, tris, mats, 512, 512, want_stats=True, pipelined=True)
        steps = 200
        dt = time_render(img, fence, steps, 5, 64, 2)
        img.release()
        res["configs[1]"] = {"workload": "cornellbox.bin 512x512, 64 spp, depth 2", "value": 512 * 512 * 64 * steps / dt / 1e6,
                             "unit": "Msamples/s", "steps": steps, "ms_per_step": dt / steps * 1e3}

    def leg_configs3():
        # configs[3]'s image on ONE GPU (BASELINE names 8 + gather): cornellbox 2048x2048, 1024 spp, depth 16; one timed render
        tris, mats = scene.load_model()
        img = StripeImage(dev, tris, mats, 2048, 2048, want_stats=True, pipelined=True)
        dt = time_render(img, fence, 1, 1, 1024, 16)   # (the warm-up is the very call that is timed: VERDICT r03 -- a shorter one left the first
        img.release()                                  #  full-size render's one-off costs inside the timed region)
        res["configs[3]"] = {"workload": "cornellbox.bin 2048x2048, 1024 spp, depth 16, 1 GPU (BASELINE names 8 + gather)",
                             "value": 2048 * 2048 * 1024 / dt / 1e6, "unit": "Msamples/s", "steps": 1, "ms_per_step": dt * 1e3}

    def leg_configs4():
        # configs[4] scene on ONE GPU at full size: 10^6-triangle soup, 1024x1024, 256 spp, depth 16, LBVH
        tris, mats = scene.make_soup(1_000_000)
        img = StripeImage(dev, tris, mats, 1024, 1024, want_stats=True, pipelined=True)
        img.render(256, frame_begin=0, max_bounces=16)   # builds the LBVH; the same call as the timed one
        img.gather()
        fence()
        img.reset_stats()
        shim.check(lib.pt_profile_enable(dev._h, 1))
        shim.check(lib.pt_profile_reset(dev._h))
        dt = time_render(img, fence, 1, 0, 256, 16)
        tot_ms, launches = ctypes.c_double(), ctypes.c_uint64()
        shim.check(lib.pt_profile_query(dev._h, shim.PT_PROF_TRACE, ctypes.byref(tot_ms), ctypes.byref(launches)))
        shim.check(lib.pt_profile_enable(dev._h, 0))
        st = img.read_stats()
        img.release()
        tally = bvh_tallies(dev, lib, shim, tris, mats, 1024, 1024, 16)
        n_launch = max(int(launches.value), 1)
        rf, rf_valu = soup_roofline(tally, st["rays"] / n_launch, st["samples"] / n_launch, tot_ms.value / n_launch)
        res["configs[4]"] = {"workload": "10^6-triangle soup 1024x1024, 256 spp, depth 16, LBVH, 1 GPU (BASELINE names 8)",
                             "value": 1024 * 1024 * 256 / dt / 1e6, "unit": "Msamples/s", "steps": 1, "ms_per_step": dt * 1e3,
                             "rays_per_sample": st["rays"] / max(st["samples"], 1), "trace_launches": n_launch,
                             "roofline": rf, "roofline_valu": rf_valu}

    for key, leg in (("configs[1]", leg_configs1), ("configs[3]", leg_configs3), ("configs[4]", leg_configs4)):
        try:
            leg()
        except Exception as e:  # noqa: BLE001 -- reported in the line, the headline measurement stands
            res[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                fence()
            except Exception:  # noqa: BLE001
                pass
    return res


if __name__ == "__main__":
    main()
