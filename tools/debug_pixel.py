#!/usr/bin/env python3
"""Follow ONE pixel through the passes on the GPU and in the CPU oracle and print the first pass after which they disagree (path state
and accumulator, all bits).  usage: tools/debug_pixel.py CONFIG X Y [PASSES]   (on the GPU box; test infrastructure like tests/)"""
import os, sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT), sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from rayzath_amd import scenes
from rayzath_amd.engine import Context, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten

name, x, y = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 12
preset = scenes.CONFIGS[name]
world = preset["build"]()
flat, cam = flatten(world), camera_struct(world.camera)
cfg = RenderConfig(tracing=Tracing(preset["max_depth"], 8)).struct()
ctx = Context(0)
ctx.upload_scene(flat), ctx.upload_camera(cam), ctx.set_config(cfg)
ref = oracle.OracleRenderer(flat, cam, cfg)
f = lambda v: " ".join(f"{float(a)!r}/{np.float32(a).view(np.uint32):08x}" for a in np.atleast_1d(v))
for p in range(passes):
    ctx.render(1), ref.render(1)
    g, r = ctx.read_state(), ref.state
    ga, ra = ctx.read_accum()[y, x], ref.accum[y, x]
    same = all(np.array_equal(g[k][y, x], r[k][y, x]) for k in g) and np.array_equal(ga, ra)
    print(f"pass {p}: {'same' if same else 'DIFFERENT'}  depth {g['depth'][y, x]} / {r['depth'][y, x]}  material {g['material'][y, x]} / {r['material'][y, x]}")
    for k in ("origin", "direction", "color"):
        if not same or p == passes - 1:
            print(f"   {k:9s} gpu    {f(g[k][y, x])}\n   {'':9s} oracle {f(r[k][y, x])}")
    if not same or p == passes - 1:
        print(f"   accum     gpu    {f(ga)}\n             oracle {f(ra)}")
