"""Multi-GPU frame assembly on CPU: world_size-2 (and 3) gloo process groups exercise the same
`gather_tiles` collective and tile layout the GPU path uses (rayzath_amd/distributed.py); the
device-side layout itself (pixel_of_thread in hiprz_device.hpp) is checked against
tile_pixel_coords by the GPU test test_sharded_render_equals_unsharded."""
import os
import socket
import sys

import numpy as np
import pytest

from rayzath_amd.distributed import TILE_PIXELS, owned_tile_count, tile_grid, tile_owner, tile_pixel_coords

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("size", [(1920, 1080), (256, 256), (200, 120), (33, 9), (1, 1), (3840, 2160)])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_tiles_partition_the_frame(size, world):
    W, H = size
    tiles_x, tiles_y = tile_grid(W, H)
    seen = np.zeros((H, W), np.int32)
    total_tiles = 0
    for rank in range(world):
        x, y = tile_pixel_coords(W, H, rank, world)
        n = owned_tile_count(rank, world, tiles_x * tiles_y)
        total_tiles += n
        assert len(x) == n * TILE_PIXELS
        inside = x >= 0
        np.add.at(seen, (y[inside], x[inside]), 1)
        assert (tile_owner(x[inside] // 32, y[inside] // 8, tiles_x, world) == rank).all()
        if rank:  # capacity never exceeds rank 0's (the gather pads to it)
            assert n <= owned_tile_count(0, world, tiles_x * tiles_y)
    assert total_tiles == tiles_x * tiles_y
    assert (seen == 1).all()  # every pixel owned by exactly one shard


@pytest.mark.parametrize("world", [2, 3, 4, 8])
@pytest.mark.parametrize("width", [1280, 1920, 2560, 3840])
def test_a_column_a_row_or_a_diagonal_of_tiles_is_dealt_to_all_shards(world, width):
    """What the row offsets are for (hiprz_shard.hpp): a thin feature of the image must not land on one or two shards.  With tile t going
    to shard t % world a column of tiles fell on ONE shard whenever world divides the tiles per row (1280, 2560, 3840 pixels)."""
    tiles_x, tiles_y = tile_grid(width, width * 9 // 16)
    rows, columns = np.mgrid[0:tiles_y, 0:tiles_x]
    owner = tile_owner(columns, rows, tiles_x, world)
    most = lambda line: np.bincount(line, minlength=world).max()
    for c in range(world, tiles_x):     # (the first columns of a row wrap around its end and follow the unrotated numbers)
        assert most(owner[:, c]) <= -(-tiles_y // world)
    for r in range(tiles_y):
        assert most(owner[r, :]) <= -(-tiles_x // world)
    if world in (2, 4, 8):
        for dr, dc in [(1, 1), (1, -1), (1, 2), (1, -2), (2, 1), (2, -1), (1, 3), (1, -3), (3, 1), (3, -1), (1, 4), (1, -4)]:
            for c0 in range(-tiles_x, 2 * tiles_x, 5):
                k = np.arange(256)
                rr, cc = k * dr, c0 + k * dc
                inside = (rr < tiles_y) & (cc >= world) & (cc < tiles_x)
                if inside.sum() >= 4 * world:   # at most two tiles of every `world` consecutive ones on the line
                    assert np.bincount(owner[rr[inside], cc[inside]], minlength=world).max() <= 2 * -(-inside.sum() // world), (dr, dc, c0)


def test_a_wave_is_an_8x8_pixel_square():
    x, y = tile_pixel_coords(64, 16, 0, 1)
    for wave in range(4):
        wx, wy = x[wave * 64:(wave + 1) * 64], y[wave * 64:(wave + 1) * 64]
        assert wx.max() - wx.min() == 7 and wy.max() - wy.min() == 7 and wx.min() == wave * 8


def _worker(rank, world, port, W, H, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from rayzath_amd.distributed import gather_tiles, owned_tile_count as otc, tile_grid as tg, tile_pixel_coords as tpc

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    tiles_x, tiles_y = tg(W, H)
    cap = otc(0, world, tiles_x * tiles_y) * 256
    x, y = tpc(W, H, rank, world)
    local = torch.zeros((cap, 4), dtype=torch.float32)
    inside = x >= 0
    vals = np.stack([x, y, x * 0 + rank, y * W + x], axis=-1).astype(np.float32)
    local[: len(x)][torch.from_numpy(inside)] = torch.from_numpy(vals[inside])
    parts = gather_tiles(local, rank, world, cap, dist)
    from rayzath_amd.distributed import total_ray_count
    owned = int(inside.sum())
    assert total_ray_count(7 * owned, dist) == 7 * W * H     # every pixel is owned by exactly one rank
    if rank == 0:
        image = np.full((H, W, 4), -1, np.float32)
        for r, part in enumerate(parts):
            px, py = tpc(W, H, r, world)
            ok = px >= 0
            image[py[ok], px[ok]] = part.numpy()[: len(px)][ok]
        np.save(out_path, image)
    else:
        assert parts is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,size", [(2, (200, 120)), (3, (97, 41))])
def test_gather_assembles_the_frame_over_gloo(tmp_path, world, size):
    import torch.multiprocessing as mp
    W, H = size
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, port, W, H, out), nprocs=world, join=True)
    image = np.load(out)
    yy, xx = np.mgrid[0:H, 0:W]
    assert np.array_equal(image[..., 0], xx) and np.array_equal(image[..., 1], yy)
    assert np.array_equal(image[..., 3], yy * W + xx)
    assert np.array_equal(image[..., 2], tile_owner(xx // 32, yy // 8, tile_grid(W, H)[0], world))


def _reduce_worker(rank, world, port, W, H, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from rayzath_amd.distributed import ShardedFrame, reduce_tiles, sample_shard_seed, tile_grid as tg, tile_pixel_coords as tpc, total_ray_count

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    # sample sharding: every rank holds the WHOLE frame, tile-major, in the same layout (shard 0 of 1)
    x, y = tpc(W, H, 0, 1)
    inside = x >= 0
    local = torch.zeros((len(x), 4), dtype=torch.float32)
    vals = np.stack([x * (rank + 1), y * (rank + 1), x * 0 + sample_shard_seed(100, rank) - 100, x * 0 + 3], axis=-1).astype(np.float32)
    local[torch.from_numpy(inside)] = torch.from_numpy(vals[inside])
    summed = reduce_tiles(local, rank, world, dist)
    assert total_ray_count(5 * W * H, dist) == world * 5 * W * H     # every rank traces the whole frame
    if rank == 0:
        image = np.full((H, W, 4), -1, np.float32)
        image[y[inside], x[inside]] = summed.numpy()[inside]
        np.save(out_path, image)
    else:
        assert summed is None   # (a non-root rank's buffer is scratch after the collective: it is the export buffer, never the accumulators)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,size", [(2, (200, 120)), (3, (97, 41))])
def test_reduce_sums_the_ranks_accumulators_over_gloo(tmp_path, world, size):
    """ShardedFrame.reduce's collective (rayzath_amd/distributed.py: reduce_tiles): one reduce(sum) of the tile-major RGBA32F accumulators to
    rank 0; the ranks' seed streams are distinct; the ray counters add up to world * passes * W * H."""
    import torch.multiprocessing as mp
    from rayzath_amd.distributed import sample_shard_seed
    W, H = size
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "sum.npy")
    mp.spawn(_reduce_worker, args=(world, port, W, H, out), nprocs=world, join=True)
    image = np.load(out)
    yy, xx = np.mgrid[0:H, 0:W]
    k = world * (world + 1) // 2
    assert np.array_equal(image[..., 0], xx * k) and np.array_equal(image[..., 1], yy * k)
    assert np.array_equal(image[..., 2], np.full((H, W), sum(range(world)))) and np.array_equal(image[..., 3], np.full((H, W), 3 * world))
    assert len({sample_shard_seed(7, r, 2) + k for r in range(8) for k in range(2)}) == 16     # 8 ranks x 2 parts: 16 distinct seed streams
    assert sample_shard_seed(0xFFFFFFFF, 1) == 0                                               # the seed is a u32
