#!/usr/bin/env python3
"""The shadow rays' wave-level walk (rz_shadow_packet_kernel) against the cooperative one (HIPRZ_SHADOW_PACKET=0) on the living-room scene at several
frame sizes and instance counts: beams get wider as a frame gets smaller (fewer rays per light and cell).  ms per pass of 8-pass batches, hip events."""
import os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayzath_amd import scenes
from rayzath_amd.engine import Context, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten

for (w, h, n) in [(3840, 2160, 40), (1920, 1080, 40), (960, 540, 40), (480, 270, 40), (1920, 1080, 300), (480, 270, 300)]:
    world = scenes.living_room(w, h, n)
    flat, cam = flatten(world), camera_struct(world.camera)
    row = []
    for packet in ("1", "0"):
        os.environ["HIPRZ_SHADOW_PACKET"] = packet
        ctx = Context(0)
        ctx.set_tree(4)
        ctx.upload_scene(flat), ctx.upload_camera(cam), ctx.set_config(RenderConfig(tracing=Tracing(8, 8)).struct())
        ctx.render(1), ctx.render(8), ctx.render(8), ctx.sync()
        t0 = time.perf_counter()
        for _ in range(4):
            ctx.render(8)
        ctx.sync()
        row.append((time.perf_counter() - t0) / 32 * 1e3)
        ctx.close()
    print(f"{w}x{h}, {n} instances: wave-level {row[0]:.3f} ms per pass, cooperative {row[1]:.3f}  ({row[1] / row[0]:.2f}x)", flush=True)
