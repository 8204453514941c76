#!/usr/bin/env python3
"""Timeline of the trace kernel's workgroups in one pass (hiprz_set_workgroup_timing): how long the kernel runs with
few workgroups left (the tail), and what a longest-first order could gain."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rayzath_amd import scenes
from rayzath_amd.engine import Context, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="D")
ap.add_argument("--passes", type=int, default=12)
args = ap.parse_args()
preset = scenes.CONFIGS[args.config]
w = preset["build"]()
flat, cam = flatten(w), camera_struct(w.camera)
ctx = Context(0)
ctx.set_pipeline(1)
ctx.set_graph(False)
ctx.upload_scene(flat), ctx.upload_camera(cam), ctx.set_config(RenderConfig(tracing=Tracing(preset["max_depth"], 8)).struct())
ctx.set_workgroup_timing(True)
ctx.render(args.passes)
n = ((cam.width + 31) // 32) * ((cam.height + 7) // 8)
t = ctx.read_workgroup_times(n).astype(np.int64)
t0 = t[:, 0].min()
start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0  # us
dur = end - start
total = end.max()
print(f"config {args.config}: {n} workgroups, kernel span {total:.0f} us, workgroup duration mean {dur.mean():.1f} median {np.median(dur):.1f} "
      f"p99 {np.percentile(dur, 99):.1f} max {dur.max():.1f} us")
order = np.argsort(end)
for frac in (0.5, 0.9, 0.99, 0.999):
    print(f"  {frac*100:5.1f} % of the workgroups have finished by {end[order[int(frac * n) - 1]]:.0f} us")
# resident workgroups over time
for q in (0.25, 0.5, 0.75, 0.9, 0.95, 0.99):
    tt = q * total
    print(f"  at {tt:7.0f} us ({q*100:.0f} % of the span): {int(((start <= tt) & (end > tt)).sum())} workgroups running")
slots = int(((start <= 0.25 * total) & (end > 0.25 * total)).sum())
print(f"  sum of durations / slots ({slots}) = {dur.sum() / slots:.0f} us = span if perfectly packed; longest workgroup starts at {start[np.argmax(dur)]:.0f} us")
