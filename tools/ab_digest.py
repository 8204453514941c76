#!/usr/bin/env python3
"""Digest of a config's frame after 1 + 8 + 3 passes under the hosts' default trees (for A/B libraries: HIPRZ_LIB=... tools/ab_digest.py C D E)."""
import hashlib, os, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rayzath_amd import scenes
from rayzath_amd.engine import Context, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten

for name in sys.argv[1:] or ["C"]:
    preset = scenes.CONFIGS[name]
    world = preset["build"]()
    flat, cam = flatten(world), camera_struct(world.camera)
    for pipeline in (-1, 1):
        ctx = Context(0)
        ctx.set_tree(4), ctx.set_pipeline(pipeline)
        ctx.upload_scene(flat), ctx.upload_camera(cam), ctx.set_config(RenderConfig(tracing=Tracing(preset["max_depth"], 8)).struct())
        ctx.render(1), ctx.render(8), ctx.render(3)
        counters = ctx.render_counted(1)
        print(name, "pipeline", ctx.pipeline(), hashlib.sha256(ctx.read_accum().tobytes()).hexdigest()[:16], "box tests", counters["box_tests"], "tri tests", counters["tri_tests"], flush=True)
        ctx.close()
