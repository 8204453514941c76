"""Child process of test_sharded_frame_reduce_* (tests/test_sample_sharding_gpu.py): torch initialises the GPU before the library does.

1. ShardedFrame(mode="samples") on one rank — export -> untile -> tone map — for a one-part context, a context over two streams in tile
   mode (two slices) and a context over three parts in sample mode (one slice: the parts' sum): equals the context's own readbacks.
2. The stream choreography of reduce() with BOTH ranks of a 2-GPU job living in this process: the collective is a stand-in that adds the
   other rank's export buffer on whatever stream torch has current, as RCCL's reduce would; the reduce of frame k overlaps the rendering of
   frame k + 1 (argv[1] = 1) or runs on the render stream (0).  Every frame must be the sum of the two ranks' one-GPU frames."""
import os
import sys

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rayzath_amd import scenes  # noqa: E402
from rayzath_amd.distributed import ShardedFrame, sample_shard_seed  # noqa: E402
from rayzath_amd.engine import SHARD_SAMPLES, SHARD_TILES, Context, RenderConfig, Tracing  # noqa: E402
from rayzath_amd.scene import camera_struct, flatten  # noqa: E402

SEED = 4242


class InProcessReduce:
    ReduceOp = type("ReduceOp", (), {"SUM": "sum"})

    def __init__(self):
        self.other = None

    def get_backend(self):
        return "nccl"

    def reduce(self, tensor, dst=0, op=None):
        if self.other is None:      # rank 1 hands its buffer over
            self.other = tensor
            return
        tensor.add_(self.other)     # rank 0: in place, on the current stream
        self.other = None


def config(seed):
    return RenderConfig(tracing=Tracing(5, 4), seed=seed).struct()


def main(overlap):
    world_scene = scenes.cornell_box(200, 120)
    flat, cam = flatten(world_scene), camera_struct(world_scene.camera)
    dev = torch.device("cuda", 0)
    for devices, mode in [(0, SHARD_TILES), ([0, 0], SHARD_TILES), ([0, 0, 0], SHARD_SAMPLES)]:
        ctx = Context(devices)
        ctx.set_shard_mode(mode)
        ctx.upload_scene(flat), ctx.upload_camera(cam), ctx.set_config(config(sample_shard_seed(SEED, 3, 3)))
        ctx.render(1), ctx.render(4)
        frame = ShardedFrame(ctx, 0, 1, cam.width, cam.height, None, dev, mode="samples")
        rgba8 = frame.reduce()
        frame.sync()
        torch.cuda.synchronize()
        assert np.array_equal(frame.image.cpu().numpy(), ctx.read_accum()), "summed accumulators differ from hiprz_read_accum"
        ctx.tonemap()
        assert np.array_equal(rgba8.cpu().numpy().view(np.uint8).reshape(cam.height, cam.width, 4), ctx.read_rgba8()), "tone-mapped sum differs"
        ctx.close()

    fake = InProcessReduce()
    ctxs, frames, refs = [], [], []
    for r in (0, 1):
        c = Context([0, 0])   # the hosts' default packaging: every rank's whole frame over two streams
        c.upload_scene(flat), c.upload_camera(cam), c.set_config(config(sample_shard_seed(SEED, r)))
        ctxs.append(c)
        frames.append(ShardedFrame(c, r, 2, cam.width, cam.height, fake, dev, overlap=overlap, mode="samples"))
        ref = Context(0)
        ref.upload_scene(flat), ref.upload_camera(cam), ref.set_config(config(sample_shard_seed(SEED, r)))
        refs.append(ref)
    img8 = None
    for _ in range(4):
        for r in (1, 0):
            ctxs[r].render(4)
            refs[r].render(4)
        frames[1].reduce()
        ctxs[1].sync()   # rank 1 only hands its buffer over here: order it before rank 0's add of it
        img8 = frames[0].reduce()
    frames[0].sync()
    torch.cuda.synchronize()
    want = refs[0].read_accum() + refs[1].read_accum()
    assert np.array_equal(frames[0].image.cpu().numpy(), want), "reduced frame differs from the sum of the two one-GPU frames"
    assert ctxs[0].ray_count() + ctxs[1].ray_count() == 2 * 16 * cam.width * cam.height
    assert img8 is not None and int(img8.cpu().numpy().view(np.uint8).reshape(cam.height, cam.width, 4)[..., 3].min()) == 255
    print("frames equal")


if __name__ == "__main__":
    main(bool(int(sys.argv[1])) if len(sys.argv) > 1 else True)
