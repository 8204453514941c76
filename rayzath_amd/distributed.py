"""Multi-GPU frame assembly: one process per GPU.  Two ways to divide a frame between the ranks (ShardedFrame(mode=...)):

"tiles" — the framebuffer sharded by interleaved
32x8-pixel tiles (numbered row by row, tile t to rank t % world, the rows rotated so that column c of row r goes to rank
(c + shard_row_offset(r)) % world; include/hiprz.h: hiprz_set_shard), the
scene replicated, and ONE collective per readback: a gather of the tile-major tone-mapped RGBA8
tiles (or the RGBA32F accumulators) to rank 0 over RCCL (torch.distributed backend "nccl") —
SURVEY.md §8e.  There is no collective on the per-pass data path: pixels are independent.  Frames are the one-GPU frame bit for bit.

"samples" — every rank renders the WHOLE frame on a seed stream of its own (`sample_shard_seed`) and ONE reduce per readback sums the
RGBA32F accumulators (colour sums and finished-path counts alike) on rank 0, which tone-maps the sum: SURVEY.md §8e's "sample-sharding
with per-frame ncclReduce of full accumulators" (33 MB per 1080p frame).  A rank's step is a whole-frame step whatever the job size, so
aggregate rays per second grow with the ranks where tile sharding is held back by its slowest tile's chain of passes; the
frame is the sum of the ranks' one-GPU frames — the same for a given job size, another one for another size.

The reference has nothing to mirror here (single device: RayZath/cuda_engine_core.cu:17).
"""
import numpy as np

TILE_W, TILE_H, TILE_PIXELS = 32, 8, 256


def tile_grid(width, height):
    return (width + TILE_W - 1) // TILE_W, (height + TILE_H - 1) // TILE_H


_ROW_OFFSETS = {8: (0, 1, 3, 7, 5, 4, 2, 6), 4: (0, 1, 3, 2)}


def shard_row_offset(row, world):
    """rayzath_amd/csrc/hiprz_shard.hpp: shard_row_offset — what is added to a tile's column before `% world` in tile row `row`
    (an integer or an array of them)."""
    table = _ROW_OFFSETS.get(world)
    return np.asarray(table)[np.asarray(row) % world] if table else np.asarray(row) % world


def shard_row_shift(row, tiles_x, world):
    """hiprz_shard.hpp: shard_row_shift — the columns by which tile row `row` is rotated."""
    return ((np.asarray(row) % world) * (tiles_x % world) + world - shard_row_offset(row, world)) % world % tiles_x


def owned_tile_count(rank, world, n_tiles):
    return (n_tiles - rank + world - 1) // world if rank < n_tiles else 0


def tile_owner(tx, ty, tiles_x, world):
    """The shard that owns tile (tx, ty) (integers or arrays): hiprz_shard.hpp, shard_of_tile."""
    shift = shard_row_shift(ty, tiles_x, world)
    return (np.asarray(ty) * tiles_x + (np.asarray(tx) - shift) % tiles_x) % world


def tile_pixel_coords(width, height, rank, world):
    """(x, y) of every slot of the tile-major local layout of shard (rank, world), shape
    (owned_tiles*256,), with -1 for slots outside the frame.  Mirrors pixel_of_thread() in
    rayzath_amd/csrc/hiprz_device.hpp; used to check the device layout and by the CPU tests."""
    tiles_x, tiles_y = tile_grid(width, height)
    n = owned_tile_count(rank, world, tiles_x * tiles_y)
    lt = np.arange(n, dtype=np.int64)[:, None]
    tid = np.arange(TILE_PIXELS, dtype=np.int64)[None, :]
    tile = lt * world + rank
    ty = tile // tiles_x
    tx = (tile % tiles_x + shard_row_shift(ty, tiles_x, world)) % tiles_x
    wave, lane = tid >> 6, tid & 63
    x = tx * TILE_W + wave * 8 + (lane & 7)
    y = ty * TILE_H + (lane >> 3)
    inside = (x < width) & (y < height)
    return np.where(inside, x, -1).reshape(-1), np.where(inside, y, -1).reshape(-1)


def gather_tiles(local_tiles, rank, world, capacity_max, dist, dst=0):
    """Gather every rank's tile-major buffer (padded to `capacity_max` rows) on `dst`.
    local_tiles: torch tensor (capacity_max, C).  Returns the list of per-rank tensors on dst,
    None elsewhere.  Works with any backend (nccl on GPUs, gloo in the CPU tests)."""
    import torch

    if world == 1:
        return [local_tiles]
    if local_tiles.is_cuda and dist.get_backend() == "gloo":  # one-GPU rehearsal: stage through the host
        host = local_tiles.cpu()
        gather_list = [torch.empty_like(host) for _ in range(world)] if rank == dst else None
        dist.gather(host, gather_list=gather_list, dst=dst)
        return [t.to(local_tiles.device) for t in gather_list] if rank == dst else None
    gather_list = [torch.empty_like(local_tiles) for _ in range(world)] if rank == dst else None
    dist.gather(local_tiles, gather_list=gather_list, dst=dst)
    return gather_list


def sample_shard_seed(seed, rank, parts=1):
    """Base seed of rank `rank`'s context in a sample-sharded job: the ranks' seed streams must not overlap, and a context over `parts`
    devices / streams in HIPRZ_SHARD_SAMPLES mode uses seed .. seed + parts - 1 itself (include/hiprz.h: hiprz_set_shard_mode)."""
    return (int(seed) + int(rank) * int(parts)) & 0xFFFFFFFF


def reduce_tiles(local_tiles, rank, world, dist, dst=0):
    """Sum every rank's tile-major accumulator buffer (same layout on every rank) into `local_tiles` on `dst`: one reduce.  Works with
    any backend (nccl on GPUs: ncclReduce over xGMI; gloo in the CPU tests and the one-GPU rehearsal, staged through the host).
    Returns the summed tensor on dst, None elsewhere."""
    if world == 1:
        return local_tiles
    if local_tiles.is_cuda and dist.get_backend() == "gloo":
        host = local_tiles.cpu()
        dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
        if rank == dst:
            local_tiles.copy_(host)
        return local_tiles if rank == dst else None
    dist.reduce(local_tiles, dst=dst, op=dist.ReduceOp.SUM)
    return local_tiles if rank == dst else None


def total_ray_count(local_rays, dist, device=None):
    """Sum of the ranks' ray counters (Camera::rayCount of the whole frame): one all-reduce of a 64-bit integer — SURVEY.md §8e.
    Works with any backend (nccl: pass the rank's cuda device; gloo: CPU tensor)."""
    import torch

    t = torch.tensor([int(local_rays)], dtype=torch.int64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


class ShardedFrame:
    """Rank-local driver: renders the owned tiles and assembles full frames on rank 0.

    Nothing synchronises with the host: render -> local tone map -> export tiles run on the context's
    render stream (wrapped as a torch ExternalStream), gather -> untile on rank 0 run on a second,
    high-priority stream that waits for the export (`overlap=True`, the default) — so the collective of
    frame k overlaps the rendering of frame k+1, the way the reference's CUDA engine hands out the
    previous frame while the next one renders (cuda_engine_core.cu:115-120, sync == false); the next export
    waits for the previous gather, so one tile buffer is enough.  `overlap=False` keeps everything on the
    render stream.  `sync()` waits for both.  The default readback gathers the tone-mapped RGBA8 tiles
    (8.3 MB per 1080p frame in total, 4x less than the RGBA32F accumulators); `gather_accum` moves the float
    accumulators instead (parity checks, float output).

    mode="samples": the context renders the whole frame (shard 0 of 1) on this rank's seed stream (`sample_shard_seed`); `reduce()` —
    export accumulators -> ONE reduce(sum) to rank 0 -> untile -> tone map of the sum — replaces `gather()`, with the same stream
    choreography (the reduce of frame k overlaps the rendering of frame k+1)."""

    def __init__(self, ctx, rank, world, width, height, dist=None, device=None, overlap=True, mode="tiles"):
        import torch

        self.ctx, self.rank, self.world, self.dist = ctx, rank, world, dist
        self.width, self.height = width, height
        self.mode = mode
        if mode == "samples":
            return self._init_samples(device, overlap)
        tiles_x, tiles_y = tile_grid(width, height)
        # a context over n devices / streams hands out n slices (sub-shard rank * n + r of world * n in slice r), each with the capacity of the
        # job's largest sub-shard: the ranks' buffers laid end to end are the sub-shards 0 .. world * n - 1 (include/hiprz.h)
        self.n_parts = ctx.device_count() if hasattr(ctx, "device_count") else 1
        self.part_capacity = owned_tile_count(0, world * self.n_parts, tiles_x * tiles_y) * TILE_PIXELS
        self.capacity_max = self.part_capacity * self.n_parts
        self.device = device
        on_gpu = device is not None and device.type == "cuda"
        self.stream = torch.cuda.ExternalStream(ctx.stream(), device=device) if on_gpu else None
        self.comm = torch.cuda.Stream(device=device, priority=-1) if on_gpu and overlap and world > 1 else self.stream
        self.local8 = torch.empty((self.capacity_max, 1), dtype=torch.int32, device=device)
        self.local = None
        root = rank == 0
        # the gathered shards live in ONE buffer (shard r = row r) so that one launch untiles them all
        self.all8 = torch.empty((world, self.capacity_max, 1), dtype=torch.int32, device=device) if root and world > 1 else None
        self.parts8 = [self.all8[r] for r in range(world)] if self.all8 is not None else None
        self.rgba8 = torch.empty((height, width), dtype=torch.int32, device=device) if root else None
        self.image = None

    def _init_samples(self, device, overlap):
        import torch

        ctx, world = self.ctx, self.world
        tiles_x, tiles_y = tile_grid(self.width, self.height)
        # the context's export layout: one slice of the whole frame (one part, or a context in HIPRZ_SHARD_SAMPLES mode), or n slices =
        # the sub-shards 0 .. n - 1 of n (a context over n streams in tile mode: the hosts' default packaging for scenes without lights)
        sample_ctx = hasattr(ctx, "shard_mode") and ctx.shard_mode() == 1
        self.n_parts = 1 if sample_ctx or not hasattr(ctx, "device_count") else ctx.device_count()
        self.part_capacity = owned_tile_count(0, self.n_parts, tiles_x * tiles_y) * TILE_PIXELS
        self.capacity_max = self.part_capacity * self.n_parts
        self.device = device
        on_gpu = device is not None and device.type == "cuda"
        self.stream = torch.cuda.ExternalStream(ctx.stream(), device=device) if on_gpu else None
        self.comm = torch.cuda.Stream(device=device, priority=-1) if on_gpu and overlap and world > 1 else self.stream
        self.local = torch.zeros((self.capacity_max, 4), dtype=torch.float32, device=device)   # export buffer; on rank 0 the reduce sums into it
        root = self.rank == 0
        self.image = torch.empty((self.height, self.width, 4), dtype=torch.float32, device=device) if root else None
        self.rgba8 = torch.empty((self.height, self.width), dtype=torch.int32, device=device) if root else None
        self.local8 = None

    def reduce(self):
        """Sum the ranks' accumulators on rank 0 (one reduce), assemble the row-major RGBA32F frame and tone-map it.  Returns the RGBA8
        frame on rank 0 (`self.image` holds the summed accumulators), None elsewhere.  Asynchronous like gather(): `sync()` waits."""
        import torch

        assert self.mode == "samples"
        ctx = self.ctx
        self._before_export()
        ctx.export_accum_tiles(self.local.data_ptr(), self.local.numel() * 4)
        if self.world > 1:
            if self.dist.get_backend() == "gloo":  # one-GPU rehearsal / CPU tests: staged through the host
                ctx.sync()
                reduce_tiles(self.local, self.rank, self.world, self.dist)
            else:
                if self.comm is not self.stream:
                    self.comm.wait_stream(self.stream)  # the export just enqueued on the render stream
                with torch.cuda.stream(self.comm):
                    self.dist.reduce(self.local, dst=0, op=self.dist.ReduceOp.SUM)
        if self.rank != 0:
            return None
        stream = self.comm.cuda_stream if self.comm is not None and self.comm is not self.stream else None
        ctx.untile_gathered(self.local.data_ptr(), self.n_parts, self.part_capacity * 16, 16, self.image.data_ptr(), stream)
        ctx.tonemap_image(self.image.data_ptr(), self.rgba8.data_ptr(), stream)
        return self.rgba8

    def _gather(self, local, parts):
        import torch

        if self.world == 1:
            return [local]
        if self.dist.get_backend() == "gloo":  # one-GPU rehearsal: stage through the host
            self.ctx.sync()
            return gather_tiles(local, self.rank, self.world, self.capacity_max, self.dist)
        if self.comm is not self.stream:
            self.comm.wait_stream(self.stream)  # the export just enqueued on the render stream
        with torch.cuda.stream(self.comm):
            self.dist.gather(local, gather_list=parts if self.rank == 0 else None, dst=0)
        return parts

    def _before_export(self):
        """The tile buffer is about to be rewritten: the previous frame's gather must have read it."""
        if self.comm is not self.stream and self.comm is not None:
            self.stream.wait_stream(self.comm)

    def _untile(self, parts, gathered, image, element_bytes):
        """Row-major frame from the gathered shards, behind the gather on its stream."""
        ctx = self.ctx
        if gathered is None or parts[0].data_ptr() != gathered.data_ptr():  # world 1, or the host-staged rehearsal path
            fn = ctx.untile_rgba8 if element_bytes == 4 else ctx.untile_accum
            for r, part in enumerate(parts):
                for k in range(self.n_parts):  # slice k of rank r = sub-shard r * n_parts + k of world * n_parts
                    fn(part.data_ptr() + k * self.part_capacity * element_bytes, r * self.n_parts + k, self.world * self.n_parts, image.data_ptr())
            return
        stream = self.comm.cuda_stream if self.comm is not None and self.comm is not self.stream else None
        ctx.untile_gathered(gathered.data_ptr(), self.world * self.n_parts, self.part_capacity * element_bytes, element_bytes, image.data_ptr(), stream)

    def ray_count(self):
        """Rays traced by ALL shards since the last reset (each context counts its own pixels)."""
        on_gpu = self.device is not None and self.device.type == "cuda" and self.dist is not None and self.dist.get_backend() != "gloo"
        return total_ray_count(self.ctx.ray_count(), self.dist if self.world > 1 else None, self.device if on_gpu else None)

    def sync(self):
        """Wait until every enqueued frame has been rendered and (rank 0) assembled."""
        self.ctx.sync()
        if self.comm is not None and self.comm is not self.stream:
            self.comm.synchronize()

    def gather(self):
        """Tone-map locally, gather the RGBA8 tiles, assemble the row-major RGBA8 frame on rank 0."""
        ctx = self.ctx
        ctx.tonemap()
        self._before_export()
        ctx.export_rgba8_tiles(self.local8.data_ptr(), self.local8.numel() * 4)
        parts = self._gather(self.local8, self.parts8)
        if self.rank != 0:
            return None
        self._untile(parts, self.all8, self.rgba8, 4)
        return self.rgba8

    def gather_accum(self):
        """Accumulators (RGBA32F) of all shards -> row-major image on rank 0."""
        import torch

        ctx = self.ctx
        if self.local is None:
            self.local = torch.empty((self.capacity_max, 4), dtype=torch.float32, device=self.device)
            self.all = torch.empty((self.world, self.capacity_max, 4), dtype=torch.float32, device=self.device) if self.rank == 0 and self.world > 1 else None
            self.parts = [self.all[r] for r in range(self.world)] if self.all is not None else None
            self.image = torch.empty((self.height, self.width, 4), dtype=torch.float32, device=self.device) if self.rank == 0 else None
        self._before_export()
        ctx.export_accum_tiles(self.local.data_ptr(), self.local.numel() * 4)
        parts = self._gather(self.local, self.parts)
        if self.rank != 0:
            return None
        self._untile(parts, self.all, self.image, 16)
        return self.image
