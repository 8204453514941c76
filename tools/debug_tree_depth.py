#!/usr/bin/env python3
"""First-hit depth of a config under the reference trees and under the device's SAH trees: how many pixels differ, by how much, and which
triangles the two walks found there.  usage: tools/debug_tree_depth.py [CONFIG]"""
import os, sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rayzath_amd import scenes
from rayzath_amd.engine import Context, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten

name = sys.argv[1] if len(sys.argv) > 1 else "F"
preset = scenes.CONFIGS[name]
world = preset["build"]()
flat, cam = flatten(world), camera_struct(world.camera)
cfg = RenderConfig(tracing=Tracing(preset["max_depth"], 8)).struct()
out = {}
for tree in (0, 1, 3):
    ctx = Context(0)
    ctx.set_tree(tree)
    ctx.upload_scene(flat), ctx.upload_camera(cam), ctx.set_config(cfg)
    ctx.render(1)
    out[tree] = (ctx.read_depth(), ctx)
d0 = out[0][0]
for tree in (1, 3):
    d = out[tree][0]
    diff = np.argwhere(d != d0)
    print(f"config {name}: tree {tree} against tree 0: {len(diff)} of {d.size} first-hit depths differ")
    for y, x in diff[:12]:
        a, b = out[0][1].ray_cast(int(x), int(y)), out[tree][1].ray_cast(int(x), int(y))
        print(f"  pixel ({x}, {y}): reference tree depth {d0[y, x]!r} triangle {a[3]} (instance {a[0]}) | tree {tree} depth {d[y, x]!r} triangle {b[3]} (instance {b[0]})  rel diff {(d[y, x] - d0[y, x]) / d0[y, x]:.3e}")
