#!/usr/bin/env python3
"""Kernel-side scaling rehearsal on ONE GPU: time the work of shard 0 of N (N = 1, 2, 4, 8) — 8 passes + tone map +
tile export, everything a rank does per bench step except the collective — and print t(1) / (N * t(N)).
Tells how much of the 8-GPU scaling target is lost to small grids / launch overhead before any communication."""
import argparse, json, os, sys, time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from rayzath_amd import scenes
from rayzath_amd.distributed import ShardedFrame
from rayzath_amd.engine import Context, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="B")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--traversal", type=int, default=-1)
ap.add_argument("--pipeline", type=int, default=-1)
args = ap.parse_args()
preset = scenes.CONFIGS[args.config]
world_scene = preset["build"]()
flat, cam = flatten(world_scene), camera_struct(world_scene.camera)
cfg = RenderConfig(tracing=Tracing(preset["max_depth"], 8)).struct()
dev = torch.device("cuda", 0)
base = None
for n in (1, 2, 4, 8):
    ctx = Context(0)
    ctx.set_traversal_mode(args.traversal)
    if args.pipeline >= 0:
        ctx.set_pipeline(args.pipeline)
    ctx.set_shard(0, n)
    ctx.upload_scene(flat), ctx.upload_camera(cam), ctx.set_config(cfg)
    frame = ShardedFrame(ctx, 0, n, cam.width, cam.height, None, dev)
    def step():
        ctx.render(8)
        ctx.tonemap()
        ctx.export_rgba8_tiles(frame.local8.data_ptr(), frame.local8.numel() * 4)
    for _ in range(3):
        step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    base = base or ms
    print(json.dumps({"config": args.config, "shard_of": n, "ms_per_step": round(ms, 4), "ideal_ms": round(base / n, 4),
                      "kernel_side_efficiency": round(base / (n * ms), 3)}), flush=True)
    ctx.close()
