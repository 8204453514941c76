"""Child process of test_sharded_frame_streams_with_in_process_collective (tests/test_parity_gpu.py)."""
import os
import sys

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rayzath_amd import scenes  # noqa: E402
from rayzath_amd.distributed import ShardedFrame  # noqa: E402
from rayzath_amd.engine import Context, RenderConfig, Tracing  # noqa: E402
from rayzath_amd.scene import camera_struct, flatten  # noqa: E402


class InProcessGather:
    def __init__(self):
        self.locals = {}

    def get_backend(self):
        return "nccl"

    def gather(self, tensor, gather_list=None, dst=0):
        if gather_list is None:
            self.locals[1] = tensor
            return
        gather_list[0].copy_(tensor, non_blocking=True)
        gather_list[1].copy_(self.locals[1], non_blocking=True)


def main(overlap):
    world_scene = scenes.cornell_box(200, 120)
    flat, cam = flatten(world_scene), camera_struct(world_scene.camera)
    cfg = RenderConfig(tracing=Tracing(4, 4)).struct()
    dev = torch.device("cuda", 0)
    fake = InProcessGather()
    ctxs, frames = [], []
    for r in (0, 1):
        c = Context(0)
        c.set_shard(r, 2)
        c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
        ctxs.append(c)
        frames.append(ShardedFrame(c, r, 2, cam.width, cam.height, fake, dev, overlap=overlap))
    ref = Context(0)
    ref.upload_scene(flat), ref.upload_camera(cam), ref.set_config(cfg)
    img8 = None
    for _ in range(4):
        for r in (1, 0):
            ctxs[r].render(4)
        frames[1].gather()
        ctxs[1].sync()  # rank 1 only hands its buffer over here: order it before rank 0's copy of it
        img8 = frames[0].gather()
        ref.render(4)
    frames[0].sync()
    ref.tonemap()
    got = img8.cpu().numpy().view(np.uint8).reshape(cam.height, cam.width, 4)
    assert np.array_equal(got, ref.read_rgba8()), "RGBA8 frame differs"
    frames[1].gather_accum()
    ctxs[1].sync()
    acc = frames[0].gather_accum()
    frames[0].sync()
    assert np.array_equal(acc.cpu().numpy(), ref.read_accum()), "accumulator frame differs"
    print("frames equal")


if __name__ == "__main__":
    main(bool(int(sys.argv[1])))
