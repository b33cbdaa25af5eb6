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


class _GatherBuffers:
    """Persistent staging of one split: this rank's padded piece and, on rank 0, one [W, max]
    buffer whose rows are the gather targets (no allocation and no padding copy per frame)."""

    def __init__(self, max_elems, rank, world_size, like):
        self.max_elems = max_elems
        self.send = like.new_zeros(max_elems)
        self.recv = like.new_empty((world_size, max_elems)) if (rank == 0 and world_size > 1) else None


def _gather_padded(piece, max_elems, rank, world_size, group, bufs=None):
    """dist.gather of equally sized (padded) pieces; returns the [W, max_elems] buffer on rank 0."""
    if piece.is_cuda and dist.get_backend(group) == "gloo":
        piece = piece.cpu()          # rehearsal on a shared GPU: gloo has no device gather
        bufs = None
    if bufs is None or bufs.send.device != piece.device:
        bufs = _GatherBuffers(max_elems, rank, world_size, piece)
    if piece.data_ptr() != bufs.send.data_ptr():      # BandRenderer renders straight into bufs.send
        bufs.send[: piece.numel()] = piece.reshape(-1)
    if rank == 0:
        dist.gather(bufs.send, gather_list=[bufs.recv[r] for r in range(world_size)], dst=0, group=group)
        return bufs.recv
    dist.gather(bufs.send, gather_list=None, dst=0, group=group)
    return None


def gather_framebuffer(band, nx, ny, rank, world_size, group=None, bufs=None):
    """Contiguous bands: gather every rank's band (float32 tensor, count*nx*3 elements, any
    device) on rank 0.  Returns the full [ny, nx, 3] tensor on rank 0 and None elsewhere.
    One collective: torch.distributed.gather of equally sized (padded) pieces —
    ncclGather / grouped send-recv under RCCL."""
    if world_size == 1:
        return band.reshape(ny, nx, 3)
    max_rows = partition_rows(ny, world_size, 0)[1]
    parts = _gather_padded(band, max_rows * nx * 3, rank, world_size, group, bufs)
    if parts is None:
        return None
    full = parts.new_empty(ny * nx * 3)
    for r in range(world_size):
        b, c = partition_rows(ny, world_size, r)
        full[b * nx * 3:(b + c) * nx * 3] = parts[r][: c * nx * 3]
    return full.reshape(ny, nx, 3)


def assemble_parts(parts, nx, ny, world_size, strip_rows=STRIP_ROWS):
    """Rank 0 after the gather: parts[r] holds rank r's strips one after the other (padded to the
    longest piece).  On the device: the library's reassembly kernel (crt_assemble_strips) on the
    current stream.  On the CPU (gloo rehearsals): all complete rounds of strips with ONE strided
    copy (full[g, r] = parts[r, g] for round g, rank r) and the ragged last round strip by strip."""
    full = parts.new_empty(ny, nx, 3)
    if parts.is_cuda:
        from . import api
        api.assemble_strips(parts, parts.stride(0), nx, ny, strip_rows, world_size, full,
                            stream=torch.cuda.current_stream())
        return full
    strip_elems = strip_rows * nx * 3
    rounds = ny // (strip_rows * world_size)               # rounds in which every rank has a full strip
    if rounds > 0:
        dst = full.reshape(-1)[: rounds * world_size * strip_elems].view(rounds, world_size, strip_elems)
        dst.copy_(parts[:, : rounds * strip_elems].view(world_size, rounds, strip_elems).transpose(0, 1))
    for r in range(world_size):                             # the last, incomplete round
        at = rounds * strip_rows
        for y0, c in partition_strips(ny, world_size, r, strip_rows)[rounds:]:
            full[y0:y0 + c] = parts[r][at * nx * 3:(at + c) * nx * 3].view(c, nx, 3)
            at += c
    return full


def gather_strips(piece, nx, ny, rank, world_size, strip_rows=STRIP_ROWS, group=None, bufs=None):
    """Round-robin strips: `piece` holds this rank's strips one after the other.  One gather of
    padded pieces to rank 0 (torch.distributed.gather: RCCL under the "nccl" backend), then
    assemble_parts there."""
    if world_size == 1:
        return piece.reshape(ny, nx, 3)
    rows_of = [sum(c for _, c in partition_strips(ny, world_size, r, strip_rows)) for r in range(world_size)]
    parts = _gather_padded(piece, max(rows_of) * nx * 3, rank, world_size, group, bufs)
    if parts is None:
        return None
    return assemble_parts(parts, nx, ny, world_size, strip_rows)


class BandRenderer:
    """Per-rank state of a split render: scene + camera + the RNG states of this rank's rows
    + its device buffer.  mode "strips" (default when world_size > 1) or "bands"."""

    def __init__(self, scene, camera, nx, ny, rank, world_size, seed=0, device="cuda", mode=None, bvh=False, gather="torch",
                 group=None, overlap=False):
        from . import api
        self.api = api
        # overlap (C path only): the exchange of frame k runs on a stream of its own while frame k + 1 renders — two pieces
        # in turn, an event each way: the exchange waits for its frame's kernels, the render of frame k + 2 waits until the
        # exchange of frame k has read the piece it is about to overwrite.  gather() then returns rank 0's frame buffer as it
        # will be once the exchange stream has run (wait() orders a caller's stream behind it; a device synchronise does too).
        self.overlap = bool(overlap) and gather == "c"
        self._xstream = None
        self._slot = 0
        self._rendered = [None, None]
        self._gathered = [None, None]
        self._bands = None
        self.bvh = bool(bvh)                 # world = the scene's BVHNode tree (crt_scene_build_bvh) instead of its list
        # gather "c": the exchange step behind the C ABI — crt_gather_strips (one grouped ncclSend / ncclRecv to rank 0 + the
        # reassembly kernel, csrc/crt_multi.cpp) on a communicator made with crt_comm_unique_id / crt_comm_init_rank; the 128
        # id bytes travel from rank 0 over the process group the launcher set up.  "torch": torch.distributed.gather.
        self.gather_path = gather if (world_size > 1 or gather == "c") else "torch"
        self.comm = None
        self._full = None
        self._parts = None
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
            self.state = api.SetCurandState(nx, ny, seed, self.row_begin, self.row_count) if self.row_count else None
        # the piece is rendered straight into the padded send buffer of the gather
        if self.mode == "strips":
            max_rows = max(sum(c for _, c in partition_strips(ny, world_size, r)) for r in range(world_size))
        else:
            max_rows = partition_rows(ny, world_size, 0)[1]
        like = torch.zeros(1, dtype=torch.float32, device=device)
        if self.gather_path == "c":
            if self.mode != "strips":
                raise ValueError("gather='c' needs the strip partition (crt_gather_strips)")
            self.bufs = None
            self._bands = [like.new_zeros(max(self.row_count, 1) * nx * 3)[: self.row_count * nx * 3] for _ in range(2 if self.overlap else 1)]
            self.band = self._bands[0]
            if self.overlap:
                self._xstream = torch.cuda.Stream(device=like.device)
            self.part_stride = api.strip_part_stride(nx, ny, STRIP_ROWS, world_size)
            if rank == 0:
                self._parts = like.new_empty(self.part_stride * world_size)
                self._full = like.new_empty(ny * nx * 3)
            self.comm = self._make_comm(group, device)
        else:
            self.bufs = _GatherBuffers(max_rows * nx * 3, rank, world_size, like)
            self.band = self.bufs.send[: self.row_count * nx * 3]

    def _make_comm(self, group, device):
        """rank 0: crt_comm_unique_id; everyone: the bytes by broadcast, then crt_comm_init_rank (collective)."""
        api = self.api
        if self.world_size == 1:
            return api.comm_init_rank(1, api.comm_unique_id(), 0)
        on = device if dist.get_backend(group) == "nccl" else "cpu"
        ident = torch.zeros(api.COMM_ID_BYTES, dtype=torch.uint8, device=on)
        if self.rank == 0:
            ident.copy_(torch.frombuffer(bytearray(api.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(ident, src=0, group=group)
        return api.comm_init_rank(self.world_size, bytes(ident.cpu().numpy().tobytes()), self.rank)

    def close(self):
        if self._xstream is not None:
            self._xstream.synchronize()
        if self.comm:
            self.api.comm_destroy(self.comm)
            self.comm = None

    def render(self, samples, frame_index=0, stream=None):
        """Enqueue this rank's rows (no synchronisation)."""
        stream = stream if stream is not None else torch.cuda.current_stream()
        if self.overlap:
            self.band = self._bands[self._slot]
            if self._gathered[self._slot] is not None:
                stream.wait_event(self._gathered[self._slot])     # the exchange two frames back has read this piece
        if self.row_count == 0:
            self._mark_rendered(stream)
            return
        if self.mode == "strips":
            self.api.render_strips(self.scene, self.camera, self.state, self.nx, self.ny, samples, self.rank,
                                   STRIP_ROWS, self.world_size, self.band, frame_index=frame_index, stream=stream, bvh=self.bvh)
        else:
            self.api.render_rows(self.scene, self.camera, self.state, self.nx, self.ny, samples,
                                 self.row_begin, self.row_count, self.band, frame_index=frame_index,
                                 stream=stream, bvh=self.bvh)
        self._mark_rendered(stream)

    def _mark_rendered(self, stream):
        if self.overlap:
            ev = self._rendered[self._slot] or torch.cuda.Event()
            ev.record(stream)
            self._rendered[self._slot] = ev

    def wait(self, stream=None):
        """Order `stream` (default: the current one) behind the last exchange (overlap mode; otherwise nothing to do)."""
        last = self._gathered[self._slot ^ 1] if self.overlap else None
        if last is not None:
            (stream if stream is not None else torch.cuda.current_stream()).wait_event(last)

    def gather(self, group=None, stream=None):
        if self.gather_path == "c" and self.overlap:
            xs, slot = self._xstream, self._slot
            if self._rendered[slot] is not None:
                xs.wait_event(self._rendered[slot])
            self.api.gather_strips(self._bands[slot] if self.row_count else None, self.nx, self.ny, STRIP_ROWS, self.rank, self.world_size,
                                   self._parts, self.part_stride, self._full, self.comm, stream=xs)
            ev = self._gathered[slot] or torch.cuda.Event()
            ev.record(xs)
            self._gathered[slot] = ev
            self._slot ^= 1
            return self._full.view(self.ny, self.nx, 3) if self.rank == 0 else None
        if self.gather_path == "c":
            stream = stream if stream is not None else torch.cuda.current_stream()
            self.api.gather_strips(self.band if self.row_count else None, self.nx, self.ny, STRIP_ROWS, self.rank, self.world_size,
                                   self._parts, self.part_stride, self._full, self.comm, stream=stream)
            return self._full.view(self.ny, self.nx, 3) if self.rank == 0 else None
        if self.mode == "strips":
            return gather_strips(self.band, self.nx, self.ny, self.rank, self.world_size, STRIP_ROWS, group, self.bufs)
        return gather_framebuffer(self.band, self.nx, self.ny, self.rank, self.world_size, group, self.bufs)
