"""Row-band partition + framebuffer gather, covered on CPU with gloo (world_size 2 and 3)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cudaraytracer_amd.dist import gather_framebuffer, gather_strips, partition_rows, partition_strips


@pytest.mark.parametrize("ny,world", [(1080, 8), (2160, 8), (75, 2), (7, 3), (5, 8), (1, 1)])
def test_partition_tiles_the_image(ny, world):
    rows = [partition_rows(ny, world, r) for r in range(world)]
    assert rows[0][0] == 0
    for (b0, c0), (b1, _) in zip(rows, rows[1:]):
        assert b0 + c0 == b1
    assert rows[-1][0] + rows[-1][1] == ny
    counts = [c for _, c in rows]
    assert max(counts) - min(counts) <= 1


@pytest.mark.parametrize("ny,world", [(1080, 8), (2160, 8), (75, 2), (7, 3), (33, 8), (1, 1)])
def test_strips_tile_the_image(ny, world):
    rows = sorted(s for r in range(world) for s in partition_strips(ny, world, r))
    at = 0
    for y0, c in rows:
        assert y0 == at and 0 < c <= 32
        at += c
    assert at == ny
    counts = [sum(c for _, c in partition_strips(ny, world, r)) for r in range(world)]
    assert max(counts) - min(counts) <= 32


def _worker(rank, world, port, golden, ny, nx, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = torch.from_numpy(np.load(golden)["rgb"]).reshape(ny, nx, 3)
    b, c = partition_rows(ny, world, rank)
    band = full[b:b + c].reshape(-1).clone()           # stands in for this rank's rendered rows
    res = gather_framebuffer(band, nx, ny, rank, world)
    strips = partition_strips(ny, world, rank)
    piece = torch.cat([full[y0:y0 + c].reshape(-1) for y0, c in strips]) if strips else full.new_zeros(0)
    res2 = gather_strips(piece, nx, ny, rank, world)
    if rank == 0:
        np.save(out_path, res.numpy())
        np.save(out_path + ".strips.npy", res2.numpy())
    else:
        assert res is None and res2 is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_gather_reassembles_the_framebuffer(world, golden_dir, tmp_path):
    golden = os.path.join(golden_dir, "c2_cornell_100x75x16.npz")
    out = str(tmp_path / "full.npy")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, golden, 75, 100, out), nprocs=world, join=True)
    ref = np.load(golden)["rgb"]
    for f in (out, out + ".strips.npy"):
        got = np.load(f)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f
