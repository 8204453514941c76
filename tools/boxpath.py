#!/usr/bin/env python3
"""Diagnostic (RZ_BOXPATH_STATS build): share of wave-level box tests that take the packed shared-reciprocal path."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayzath_amd import scenes
from rayzath_amd.engine import Context, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten
for cfgname in sys.argv[1:] or ["B"]:
    preset = scenes.CONFIGS[cfgname]
    w = preset["build"]()
    if cfgname in ("D", "E"):
        w.camera.width, w.camera.height = w.camera.width // 4, w.camera.height // 4
    ctx = Context(0)
    ctx.upload_scene(flatten(w)); ctx.upload_camera(camera_struct(w.camera)); ctx.set_config(RenderConfig(tracing=Tracing(preset["max_depth"], 8)).struct())
    out = (C.c_uint64 * 4)()
    ctx.render(9); ctx.sync(); ctx.lib.hiprz_read_boxpath(out)
    ctx.render(8); ctx.sync(); ctx.lib.hiprz_read_boxpath(out)
    f, s, lanes, slow_lanes = list(out)
    print(cfgname, "mode", ctx.traversal_mode(), f"fast wave-tests {f}, slow wave-tests {s} ({100*s/max(f+s,1):.1f} % slow); lanes {lanes}, non-fast lanes {slow_lanes} ({100*slow_lanes/max(lanes,1):.2f} %)")
    ctx.close()
