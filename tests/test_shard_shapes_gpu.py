"""Tile sharding over frame sizes and shard counts that do not divide anything (the rotated tile rows of hiprz_shard.hpp: counts per shard
differ by one, rows wrap, frames narrower than `world` tiles): for every (resolution, world) the shards' readbacks are disjoint and add up
to the unsharded frame bit for bit, ray counts add up, the tile-major export of every shard scattered by hiprz_untile_* gives the same
image, a context over several streams (sub-shards of a shard) gives the same shard, and hiprz_ray_cast finds every pixel's owner."""
import ctypes as C

import numpy as np
import pytest

from rayzath_amd import scenes
from rayzath_amd.distributed import tile_owner, tile_grid, tile_pixel_coords
from rayzath_amd.engine import Context, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten

pytestmark = pytest.mark.gpu
_hip = None


def _device_buffer(nbytes):
    global _hip
    if _hip is None:
        _hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
        _hip.hipMalloc.argtypes, _hip.hipMemcpy.argtypes = [C.POINTER(C.c_void_p), C.c_size_t], [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    p = C.c_void_p()
    assert _hip.hipMalloc(C.byref(p), max(nbytes, 16)) == 0
    assert _hip.hipMemset(p, 0, max(nbytes, 16)) == 0
    assert _hip.hipDeviceSynchronize() == 0   # the memset is queued on the null stream; the contexts' streams are non-blocking and do not wait for it
    return p


@pytest.mark.parametrize("size", [(200, 120), (33, 9), (31, 8), (257, 65), (96, 64), (1, 1), (640, 24)])
@pytest.mark.parametrize("world,streams", [(2, 1), (3, 1), (5, 1), (8, 1), (2, 2), (3, 2), (4, 3)])
def test_shards_of_any_shape_add_up(built, size, world, streams):
    """`streams`: every shard of the job is a context over that many streams (hiprz_create_multi with the device named that often); a job's
    contexts must all have the same number of parts — shard r of `world` of an n-part context is the sub-shards r * n + k of world * n."""
    W, H = size
    scene = scenes.cornell_box(W, H)
    flat, cam = flatten(scene), camera_struct(scene.camera)
    cfg = RenderConfig(tracing=Tracing(4, 4)).struct()
    full = Context(0)
    full.upload_scene(flat), full.upload_camera(cam), full.set_config(cfg)
    full.render(1), full.render(4)
    want, want_depth = full.read_accum(), full.read_depth()
    picks = [(0, 0), (W - 1, H - 1), (W // 2, H // 2), (min(W - 1, 40), min(H - 1, 8))]
    want_picks = [full.pick(*p) for p in picks]
    total, rays = np.zeros_like(want), 0
    image = _device_buffer(W * H * 16)
    tiles_x, _ = tile_grid(W, H)
    for rank in range(world):
        c = Context([0] * streams) if streams > 1 else Context(0)
        c.set_shard(rank, world)
        c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
        c.render(1), c.render(4)
        part = c.read_accum()
        owned = np.abs(part).sum(-1) > 0
        assert not (np.abs(total).sum(-1) > 0)[owned].any(), "shards overlap"
        mine = np.zeros((H, W), bool)               # the host's mirror of the device's map: every lit pixel is one the context owns
        for k in range(streams):
            x, y = tile_pixel_coords(W, H, rank * streams + k, world * streams)
            mine[y[x >= 0], x[x >= 0]] = True
        assert not (owned & ~mine).any()
        yy, xx = np.nonzero(mine)
        assert (tile_owner(xx // 32, yy // 8, tiles_x, world * streams) // streams == rank).all()
        total += part
        rays += c.ray_count()
        # tile-major export -> untile into a device image that collects all shards
        cap = c.local_pixel_capacity()
        buf = _device_buffer(cap * 16)
        c.export_accum_tiles(buf.value, cap * 16)
        for k in range(streams):                  # slice k of the context = sub-shard rank * streams + k of world * streams
            c.untile_accum(buf.value + k * (cap // streams) * 16, rank * streams + k, world * streams, image.value)
        c.sync()
        for p, w in zip(picks, want_picks):       # only the owner of the pixel answers; the others report nothing
            got = c.pick(*p)
            assert got == w or got == (-1, -1)
        c.close()
    assert np.array_equal(total, want) and rays == full.ray_count()
    out = np.zeros((H, W, 4), np.float32)
    assert _hip.hipMemcpy(out.ctypes.data, image, W * H * 16, 2) == 0
    assert np.array_equal(out, want)
    assert np.array_equal(full.read_depth(), want_depth)
