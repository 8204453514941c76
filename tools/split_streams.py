#!/usr/bin/env python3
"""One GPU, the frame split over K contexts of a multi-device context that all name the SAME device (hiprz_create_multi([0] * K)):
K streams, each rendering its interleaved share of the tiles, so that one share's small kernels (sorts, pass bookkeeping) run beside
another share's walks.  Prints Mrays/s per K.  usage: split_streams.py [configs...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rayzath_amd import scenes
from rayzath_amd.engine import Context, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten

for name in sys.argv[1:] or ["C", "E", "D", "B"]:
    preset = scenes.CONFIGS[name]
    world = preset["build"]()
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(tracing=Tracing(preset["max_depth"], 8)).struct()
    for k in (1, 2, 3, 4):
        ctx = Context(0) if k == 1 else Context([0] * k)
        ctx.upload_scene(flat), ctx.upload_camera(cam), ctx.set_config(cfg)
        steps = 12 if name != "E" else 5
        for _ in range(3):
            ctx.render(8), ctx.tonemap()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.render(8), ctx.tonemap()
        ctx.sync()
        ms = (time.perf_counter() - t0) / steps * 1e3
        print(f"config {name}  {k} stream(s)  {ms:8.3f} ms/step  {8 * cam.width * cam.height / ms / 1e3:9.1f} Mrays/s", flush=True)
        ctx.close()
