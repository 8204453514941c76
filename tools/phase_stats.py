#!/usr/bin/env python3
"""Diagnostic (library built with -DRZ_PHASE_STATS, loaded through HIPRZ_LIB): for every kind of step of the MODE 3 walk, how many
times a WAVE executed it and with how many active lanes — i.e. where the trace kernel's VALU instructions go."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayzath_amd import scenes
from rayzath_amd.engine import Context, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten
NAMES = ["world node box test", "instance box test", "instance entry (to local)", "mesh node box test", "triangle test", "mesh walk round"]
for cfgname in sys.argv[1:] or ["D"]:
    preset = scenes.CONFIGS[cfgname]
    w = preset["build"]()
    ctx = Context(0)
    if cfgname in ("A", "B"):
        ctx.set_traversal_mode(2), ctx.set_pipeline(1)   # workgroup-binned walk: same step kinds, "round" = one binning round
    else:
        ctx.set_traversal_mode(3), ctx.set_pipeline(1), ctx.set_ray_sort(int(os.environ.get('PHASE_RAY_SORT', '0')))
    ctx.set_tree(int(os.environ.get('PHASE_TREE', '4')))   # 4: the Engine hosts' default trees
    ctx.upload_scene(flatten(w)); ctx.upload_camera(camera_struct(w.camera)); ctx.set_config(RenderConfig(tracing=Tracing(preset["max_depth"], 8)).struct())
    out = (C.c_uint64 * 16)()
    ctx.render(9); ctx.sync(); ctx.lib.hiprz_read_phase_stats(out)
    ctx.render(1); ctx.sync(); ctx.lib.hiprz_read_phase_stats(out)
    waves = ((w.camera.width + 31) // 32) * ((w.camera.height + 7) // 8) * 4
    print(f"config {cfgname}: one pass, {waves} waves")
    for k, name in enumerate(NAMES):
        n, lanes = out[2 * k], out[2 * k + 1]
        print(f"  {name:28s} wave executions {n:12d} ({n / waves:8.1f} per wave)  lanes {lanes:13d}  mean active lanes {lanes / max(n, 1):5.1f}")
    if hasattr(ctx.lib, "hiprz_read_shadow_phase_stats") and os.environ.get("PHASE_SHADOW"):   # library built with -DRZ_PHASE_STATS in the shade unit too
        ctx.lib.hiprz_read_shadow_phase_stats(out); ctx.render(1); ctx.sync(); ctx.lib.hiprz_read_shadow_phase_stats(out)
        print("  shadow rays, wave-level walk (any_hit_packet):")
        for k, name in enumerate(NAMES[:5]):
            n, lanes = out[2 * k], out[2 * k + 1]
            print(f"  {name:28s} wave executions {n:12d} ({n / waves:8.1f} per wave)  lanes {lanes:13d}  mean active lanes {lanes / max(n, 1):5.1f}")
    ctx.close()
