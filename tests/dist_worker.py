"""One rank of the world-N GPU tests (tests/test_gpu_dist.py): renders its stripes with the product's
StripeImage, takes part in the gather, and -- on rank 0 -- saves the assembled image.

usage (environment: RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT):
    python dist_worker.py <nccl|gloo> <out.npy> W H frames stripe_rows [pipelined]
backend gloo: every rank uses device (LOCAL_RANK % device_count) -- ranks may share one GPU.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    backend, out = sys.argv[1], sys.argv[2]
    W, H, frames, stripe = (int(x) for x in sys.argv[3:7])
    import torch  # before the shim: both must bind to one HIP runtime
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    dev_idx = int(os.environ.get("LOCAL_RANK", "0")) % ndev
    torch.cuda.set_device(dev_idx)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_idx))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from oclpathtracer_amd import adl, scene
    from oclpathtracer_amd.distributed import StripeImage

    assert adl.init(adl.TYPE_HIP)
    dev = adl.DeviceUtils.allocate(adl.TYPE_HIP, adl.Config(dev_idx))
    tris, mats = scene.load_model()
    pipelined = len(sys.argv) > 7 and sys.argv[7] == "pipelined"
    img = StripeImage(dev, tris, mats, W, H, world=world, rank=rank, stripe_rows=stripe, pipelined=pipelined)
    if pipelined:
        # bench.py's loop: three complete images, each gathered after the NEXT render has been enqueued; the images of
        # slots 0, 1, 0 have 1, 2 and `frames` frames, the last gather must deliver the last one
        prev = None
        for n in (1, 2, frames):
            slot = img.render(n, frame_begin=0)
            if prev is not None:
                img.gather(prev)
            prev = slot
        image = img.gather(prev)
    else:
        # two halves, the second enqueued right behind the first gather: exercises render-after-gather ordering
        half = frames // 2
        img.render(half, frame_begin=0)
        img.gather()
        img.render(frames - half, frame_begin=half)
        image = img.gather()
    if rank == 0:
        torch.cuda.synchronize()
        import numpy as np

        np.save(out, image.cpu().numpy())
    dist.barrier()
    img.release()
    adl.DeviceUtils.deallocate(dev)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
