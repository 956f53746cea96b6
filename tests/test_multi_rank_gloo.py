"""The N > 1 path on CPU: two ranks (gloo) shard a frame into interleaved row bands, gather
the shards to rank 0 and rebuild the frame -- the same tiling/gather code bench.py runs over
RCCL.  Pixels come from the CPU oracle here (the test exercises sharding, not the kernels)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, BAND = 48, 37, 8


def _worker(rank, world_size, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    import raytracer_amd as rt
    from oracle import orc
    from raytracer_amd import tiling
    world, cam = rt.exercise("glass_bubble", W, H)
    ow = orc.OracleWorld(world.describe())
    slots = tiling.padded_local_rows(H, world_size, BAND)
    rows = tiling.image_rows(H, rank, world_size, BAND, n_slots=slots)
    full = np.zeros((H, W, 3))
    for y in rows[rows >= 0]:
        ow.render(cam, rows=(int(y), int(y) + 1, 1), nthreads=1, out=full)
    shard = np.zeros((slots, W, 3))
    shard[rows >= 0] = full[rows[rows >= 0]]
    t = torch.from_numpy(shard)
    gathered = [torch.empty_like(t) for _ in range(world_size)] if rank == 0 else None
    dist.gather(t, gathered, dst=0)
    if rank == 0:
        frame = tiling.assemble(gathered, H, world_size, BAND).numpy()
        ref, _ = ow.render(cam, nthreads=1)
        np.save(out_path, np.array([float(np.array_equal(frame, ref))]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_row_band_gather(tmp_path):
    out = str(tmp_path / "ok.npy")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert np.load(out)[0] == 1.0
