#!/usr/bin/env python3
"""On the GPU box: where does the first pass of shading_inputs_scene differ from the oracle?"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle
from rayzath_amd import scenes
from rayzath_amd.engine import Context, LightSampling, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten

world = scenes.shading_inputs_scene(160, 96)
flat, cam = flatten(world), camera_struct(world.camera)
cfg = RenderConfig(LightSampling(1, 1), Tracing(6, 8)).struct()
ctx = Context(0)
ctx.upload_scene(flat), ctx.upload_camera(cam), ctx.set_config(cfg)
ref = oracle.OracleRenderer(flat, cam, cfg)
ctx.render(1), ref.render(1)
a, r = ctx.read_accum(), ref.accum
bad = np.argwhere((a != r).any(-1))
print("differing pixels", len(bad), "sky among them", int((ref.depth[bad[:, 0], bad[:, 1]] >= 999).sum()))
for y, x in bad[:12]:
    print((x, y), "depth", ref.depth[y, x], "gpu", a[y, x], "cpu", r[y, x], "rel", np.abs(a[y, x] - r[y, x]) / np.maximum(np.abs(r[y, x]), 1e-30))
