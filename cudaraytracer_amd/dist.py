"""Multi-GPU split of the list-render path: one process per GPU, image rows dealt to the
ranks, one gather of the float3 framebuffer pieces to rank 0 (RCCL over xGMI when the
backend is "nccl"; gloo on CPU for the tests).

The reference is single-GPU (SURVEY.md §2 "Parallelism"), so there is nothing to mirror:
pixels are independent (the RNG stream of a pixel is its GLOBAL pixel index,
core/deviceManage.h:126-127), triangles are replicated, and the only exchange step is the
final image gather.  No collective touches the data path of the render itself.

Two partitions:
  * strips (default for world_size > 1): strips of 32 rows dealt round-robin — rank r owns
    strips r, r+W, r+2W, ...  With the tiled kernels the cost of a row is proportional to
    what it shows, so contiguous bands leave the ranks that own sky idle; interleaving
    balances them.
  * bands: contiguous row bands (rank r owns rows [begin, begin+count)).
"""
import torch
import torch.distributed as dist

STRIP_ROWS = 32


def partition_rows(ny, world_size, rank):
    """Contiguous band [begin, begin+count) of rank `rank`: bands differ by at most one row
    and tile the image exactly (rank 0 gets the bottom band; row 0 = image bottom)."""
    base, rem = divmod(ny, world_size)
    begin = rank * base + min(rank, rem)
    count = base + (1 if rank < rem else 0)
    return begin, count


def partition_strips(ny, world_size, rank, strip_rows=STRIP_ROWS):
    """Strips (first_row, rows) owned by `rank` when strips are dealt round-robin."""
    out = []
    s = rank
    while s * strip_rows < ny:
        y0 = s * strip_rows
        out.append((y0, min(strip_rows, ny - y0)))
        s += world_size
    return out


def _gather_padded(piece, max_elems, rank, world_size, group):
    if piece.is_cuda and dist.get_backend(group) == "gloo":
        piece = piece.cpu()          # rehearsal on a shared GPU: gloo has no device gather
    padded = piece.new_zeros(max_elems)
    padded[: piece.numel()] = piece.reshape(-1)
    if rank == 0:
        parts = [torch.empty_like(padded) for _ in range(world_size)]
        dist.gather(padded, gather_list=parts, dst=0, group=group)
        return parts
    dist.gather(padded, gather_list=None, dst=0, group=group)
    return None


def gather_framebuffer(band, nx, ny, rank, world_size, group=None):
    """Contiguous bands: gather every rank's band (float32 tensor, count*nx*3 elements, any
    device) on rank 0.  Returns the full [ny, nx, 3] tensor on rank 0 and None elsewhere.
    One collective: torch.distributed.gather of equally sized (padded) pieces —
    ncclGather / grouped send-recv under RCCL."""
    if world_size == 1:
        return band.reshape(ny, nx, 3)
    max_rows = partition_rows(ny, world_size, 0)[1]
    parts = _gather_padded(band, max_rows * nx * 3, rank, world_size, group)
    if parts is None:
        return None
    full = band.new_empty(ny * nx * 3)
    for r in range(world_size):
        b, c = partition_rows(ny, world_size, r)
        full[b * nx * 3:(b + c) * nx * 3] = parts[r][: c * nx * 3]
    return full.reshape(ny, nx, 3)


def gather_strips(piece, nx, ny, rank, world_size, strip_rows=STRIP_ROWS, group=None):
    """Round-robin strips: `piece` holds this rank's strips one after the other.  One gather,
    then rank 0 copies every strip to its place."""
    if world_size == 1:
        return piece.reshape(ny, nx, 3)
    rows_of = [sum(c for _, c in partition_strips(ny, world_size, r, strip_rows)) for r in range(world_size)]
    parts = _gather_padded(piece, max(rows_of) * nx * 3, rank, world_size, group)
    if parts is None:
        return None
    full = piece.new_empty(ny, nx, 3)
    for r in range(world_size):
        p = parts[r][: rows_of[r] * nx * 3].reshape(rows_of[r], nx, 3)
        at = 0
        for y0, c in partition_strips(ny, world_size, r, strip_rows):
            full[y0:y0 + c] = p[at:at + c]
            at += c
    return full


class BandRenderer:
    """Per-rank state of a split render: scene + camera + the RNG states of this rank's rows
    + its device buffer.  mode "strips" (default when world_size > 1) or "bands"."""

    def __init__(self, scene, camera, nx, ny, rank, world_size, seed=0, device="cuda", mode=None):
        from . import api
        self.api = api
        self.scene, self.camera = scene, camera
        self.nx, self.ny = nx, ny
        self.rank, self.world_size = rank, world_size
        self.mode = mode or ("strips" if world_size > 1 else "bands")
        if self.mode == "strips":
            self.row_count = sum(c for _, c in partition_strips(ny, world_size, rank))
            self.row_begin = rank * STRIP_ROWS
            self.state = api.CurandState.for_strips(nx, ny, rank, STRIP_ROWS, world_size, seed) \
                if self.row_count else None
        else:
            self.row_begin, self.row_count = partition_rows(ny, world_size, rank)
            self.state = api.SetCurandState(nx, ny, seed, self.row_begin, self.row_count)
        self.band = torch.zeros(self.row_count * nx * 3, dtype=torch.float32, device=device)

    def render(self, samples, frame_index=0, stream=None):
        """Enqueue this rank's rows (no synchronisation)."""
        stream = stream if stream is not None else torch.cuda.current_stream()
        if self.row_count == 0:
            return
        if self.mode == "strips":
            self.api.render_strips(self.scene, self.camera, self.state, self.nx, self.ny, samples, self.rank,
                                   STRIP_ROWS, self.world_size, self.band, frame_index=frame_index, stream=stream)
        else:
            self.api.render_rows(self.scene, self.camera, self.state, self.nx, self.ny, samples,
                                 self.row_begin, self.row_count, self.band, frame_index=frame_index,
                                 stream=stream)

    def gather(self, group=None):
        if self.mode == "strips":
            return gather_strips(self.band, self.nx, self.ny, self.rank, self.world_size, STRIP_ROWS, group)
        return gather_framebuffer(self.band, self.nx, self.ny, self.rank, self.world_size, group)
