"""Worker of tests/test_gpu_multirank.py: W ranks share ONE GPU (gloo rendezvous, CPU-staged
gather) and render the C2 golden frame with the production BandRenderer; rank 0 checks it."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cudaraytracer_amd import api, scenes  # noqa: E402
from cudaraytracer_amd.dist import BandRenderer  # noqa: E402


def main():
    out_path, mode = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "c2_cornell_100x75x16.npz"))
    nx, ny, spp = int(g["nx"]), int(g["ny"]), int(g["spp"])
    sc = scenes.cornell32()
    scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    c = sc["camera"]
    cam = api.create_camera(nx, ny, c["lookfrom"], c["lookat"], c["focus"], c["aperture"], c["vfov"])
    br = BandRenderer(scene, cam, nx, ny, rank, world, seed=0, mode=mode)
    br.render(spp)
    torch.cuda.synchronize()
    full = br.gather()
    if rank == 0:
        ok = np.array_equal(full.cpu().numpy().view(np.uint32), g["rgb"].view(np.uint32))
        with open(out_path, "w") as f:
            f.write("OK" if ok else "MISMATCH")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
