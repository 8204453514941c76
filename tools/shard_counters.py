#!/usr/bin/env python3
"""Executed box / triangle tests per path segment in every shard r of N of a config (front-to-back walk on the hosts' default trees): where
a frame's expensive rays are.  usage: tools/shard_counters.py [CONFIG] [N]"""
import os, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rayzath_amd import scenes
from rayzath_amd.engine import Context, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten

name = sys.argv[1] if len(sys.argv) > 1 else "D"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
preset = scenes.CONFIGS[name]
world = preset["build"]()
flat, cam = flatten(world), camera_struct(world.camera)
cfg = RenderConfig(tracing=Tracing(preset["max_depth"], 8)).struct()
for r in range(n):
    ctx = Context(0)
    ctx.set_tree(4), ctx.set_walk_order(2), ctx.set_shard(r, n)
    ctx.upload_scene(flat), ctx.upload_camera(cam), ctx.set_config(cfg)
    ctx.render(1)
    c = ctx.render_counted(8)
    seg = max(c["segments"], 1)
    print(f"config {name} shard {r}/{n}: {c['box_tests'] / seg:7.2f} box tests, {c['tri_tests'] / seg:6.2f} triangle tests per segment, {c['hits'] / seg:.3f} hits", flush=True)
    ctx.close()
