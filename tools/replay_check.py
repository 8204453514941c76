#!/usr/bin/env python3
"""On the GPU box: graph REPLAYS of a batch of passes with ray reordering (the path bench.py times), checked against eager launches."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayzath_amd import scenes
from rayzath_amd.engine import Context, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten
for name in sys.argv[1:] or ["C"]:
    preset = scenes.CONFIGS[name]
    w = preset["build"]()
    flat, cam = flatten(w), camera_struct(w.camera)
    out = []
    for graph in (True, False):
        c = Context(0)
        c.set_graph(graph)
        c.upload_scene(flat), c.upload_camera(cam), c.set_config(RenderConfig(tracing=Tracing(preset["max_depth"], 8)).struct())
        c.render(1)
        for k in range(4):
            c.render(8); c.sync(); print(name, "graph" if graph else "eager", "batch", k, "ok", flush=True)
        out.append(c.read_accum()); c.close()
    print(name, "replays equal eager:", np.array_equal(out[0], out[1]), flush=True)
