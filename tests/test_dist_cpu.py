"""N>1 path on CPU: world_size-2, 3 and 8 gloo rehearsal of the stripe sharding + gather + assembly.

Each rank renders ITS rows with the CPU oracle (test infrastructure standing in for the GPU
renderer, using the same StripePlan the GPU path uses), the slabs travel through
``distributed.gather_slabs`` (the product's gather, here over gloo), and rank 0 assembles with a
numpy restatement of pt_assemble_kernel's index formula.  The result must equal the
single-process oracle image bit for bit -- the property the GPU path relies on (global pixel ids
in the seed).
"""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from conftest import ROOT  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def assemble_numpy(slabs, W, H, stripe_rows, world):
    """numpy restatement of pt_assemble_kernel (oclpathtracer_amd/csrc/pt_kernels.hip)."""
    rows = np.arange(H)
    stripe = rows // stripe_rows
    within = rows - stripe * stripe_rows
    rank = stripe % world
    sl = stripe // world
    return slabs[rank, sl * stripe_rows + within]  # [H, W, 4]


def _worker(rank, world, port, W, H, frames, stripe_rows, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oclpathtracer_amd import scene
        from oclpathtracer_amd.distributed import StripePlan, gather_slabs
        from oracle import ptoracle

        tris, mats = scene.load_model()
        plan = StripePlan(H, stripe_rows, world)
        rows = plan.global_rows(rank)
        full = np.zeros((H * W, 4), np.float32)
        # render this rank's stripes: contiguous gid ranges, GLOBAL ids
        r = 0
        while r < len(rows):
            e = r
            while e + 1 < len(rows) and rows[e + 1] == rows[e] + 1:
                e += 1
            ptoracle.render(tris, mats, W, H, frames, fb=full, gid_begin=int(rows[r]) * W,
                            gid_count=(e - r + 1) * W, nthreads=2)
            r = e + 1
        slab = np.zeros((plan.slab_rows, W, 4), np.float32)
        slab[: len(rows)] = full.reshape(H, W, 4)[rows]
        got = gather_slabs(torch.from_numpy(slab), world, rank)
        if rank == 0:
            assert got.shape == (world, plan.slab_rows, W, 4)
            np.save(out_path, assemble_numpy(got.numpy(), W, H, stripe_rows, world))
        else:
            assert got is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H,stripe_rows", [(2, 32, 4), (2, 30, 7), (3, 20, 3)])
def test_gloo_stripes_gather_assemble(tmp_path, oracle, cornell, world, H, stripe_rows):
    W, frames = 24, 2
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, frames, stripe_rows, out), nprocs=world, join=True)
    tris, mats = cornell
    want = oracle.render(tris, mats, W, H, frames).reshape(H, W, 4)
    got = np.load(out)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("H", [1024, 1000])
def test_gloo_world8_bench_geometry(tmp_path, oracle, cornell, H):
    """Eight rank PROCESSES (the N-rank path had never run with 8 ranks anywhere): bench.py's split -- 4-row stripes
    dealt round-robin, BASELINE's image height (and one that leaves the last period of stripes incomplete: 1000 = 31 x 32 + 8,
    ranks 2..7 own a stripe less than ranks 0 and 1, their slabs are zero-padded in the gather) -- at a small width; the slab
    list of the gather, StripePlan.slab_rows and the assembly index math at world 8, bit for bit the one-process image."""
    W, frames, world, stripe_rows = 8, 1, 8, 4
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, frames, stripe_rows, out), nprocs=world, join=True)
    tris, mats = cornell
    want = oracle.render(tris, mats, W, H, frames).reshape(H, W, 4)
    got = np.load(out)
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_single_rank_gather_is_identity():
    from oclpathtracer_amd.distributed import gather_slabs

    t = torch.arange(24, dtype=torch.float32).reshape(2, 3, 4)
    g = gather_slabs(t, 1, 0)
    assert g.shape == (1, 2, 3, 4) and torch.equal(g[0], t)
