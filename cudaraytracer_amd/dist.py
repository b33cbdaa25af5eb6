"""Multi-GPU split of the list-render path: contiguous row bands, one process per GPU,
one gather of the float3 framebuffer bands to rank 0 (RCCL over xGMI when the backend is
"nccl"; gloo on CPU for the tests).

The reference is single-GPU (SURVEY.md §2 "Parallelism"), so there is nothing to mirror:
pixels are independent (the RNG stream of a pixel is its GLOBAL pixel index,
core/deviceManage.h:126-127), triangles are replicated, and the only exchange step is the
final image gather.  No collective touches the data path of the render itself.
"""
import torch
import torch.distributed as dist


def partition_rows(ny, world_size, rank):
    """Rows [begin, begin+count) of rank `rank`: bands differ by at most one row and tile
    the image exactly (rank 0 gets the bottom band; row 0 = image bottom)."""
    base, rem = divmod(ny, world_size)
    begin = rank * base + min(rank, rem)
    count = base + (1 if rank < rem else 0)
    return begin, count


def gather_framebuffer(band, nx, ny, rank, world_size, group=None):
    """Gather every rank's band (float32 tensor, count*nx*3 elements, any device) on rank 0.
    Returns the full [ny, nx, 3] tensor on rank 0 and None elsewhere.  One collective:
    torch.distributed.gather of equally sized (padded) bands — ncclGather/grouped
    send-recv under RCCL."""
    if world_size == 1:
        return band.reshape(ny, nx, 3)
    max_rows = partition_rows(ny, world_size, 0)[1]
    begin, count = partition_rows(ny, world_size, rank)
    padded = band.new_zeros(max_rows * nx * 3)
    padded[: count * nx * 3] = band.reshape(-1)
    if rank == 0:
        parts = [torch.empty_like(padded) for _ in range(world_size)]
        dist.gather(padded, gather_list=parts, dst=0, group=group)
        full = band.new_empty(ny * nx * 3)
        for r in range(world_size):
            b, c = partition_rows(ny, world_size, r)
            full[b * nx * 3:(b + c) * nx * 3] = parts[r][: c * nx * 3]
        return full.reshape(ny, nx, 3)
    dist.gather(padded, gather_list=None, dst=0, group=group)
    return None


class BandRenderer:
    """Per-rank state of a tile-split render: scene + camera + the RNG states of this
    rank's rows + its device band buffer."""

    def __init__(self, scene, camera, nx, ny, rank, world_size, seed=0, device="cuda"):
        from . import api
        self.api = api
        self.scene, self.camera = scene, camera
        self.nx, self.ny = nx, ny
        self.rank, self.world_size = rank, world_size
        self.row_begin, self.row_count = partition_rows(ny, world_size, rank)
        self.state = api.SetCurandState(nx, ny, seed, self.row_begin, self.row_count)
        self.band = torch.zeros(self.row_count * nx * 3, dtype=torch.float32, device=device)

    def render(self, samples, frame_index=0, stream=None):
        """Enqueue this rank's band (no synchronisation)."""
        self.api.render_rows(self.scene, self.camera, self.state, self.nx, self.ny, samples,
                             self.row_begin, self.row_count, self.band, frame_index=frame_index,
                             stream=stream if stream is not None else torch.cuda.current_stream())

    def gather(self, group=None):
        return gather_framebuffer(self.band, self.nx, self.ny, self.rank, self.world_size, group)
