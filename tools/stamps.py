#!/usr/bin/env python3
"""Diagnostic (RZ_STAMP build only): cycle shares of the binned walk's phases on a config."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayzath_amd import scenes
from rayzath_amd.engine import Context, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten
cfgname = sys.argv[1] if len(sys.argv) > 1 else "B"
preset = scenes.CONFIGS[cfgname]
w = preset["build"]()
ctx = Context(0)
ctx.set_traversal_mode(2)
ctx.upload_scene(flatten(w)); ctx.upload_camera(camera_struct(w.camera)); ctx.set_config(RenderConfig(tracing=Tracing(preset["max_depth"], 8)).struct())
ctx.render(9); ctx.sync()
out = (C.c_uint64 * 8)()
ctx.lib.hiprz_read_stamps(out)
ctx.render(8); ctx.sync()
ctx.lib.hiprz_read_stamps(out)
v = list(out); waves = v[7]
names = ["A walk", "count+barrier", "scan+scatter+barrier", "C dense work", "C barrier wait", "-"]
tot = sum(v[:5])
for n, x in zip(names[:5], v[:5]):
    print(f"{n:24s} {x / waves:10.0f} cycles/wave  {100 * x / tot:5.1f} %")
print(f"rounds per wave {v[6] / waves:.2f}; total cycles/wave in the walk {tot / waves:.0f}")
