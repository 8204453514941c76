#!/usr/bin/env python3
"""Kernel-side scaling rehearsal on ONE GPU: time the work of EVERY shard r of N (N = 1, 2, 4, 8) — 8 passes + tone map +
tile export, everything a rank does per bench step except the collective — and print, per N, the slowest shard (what a job of N
ranks waits for) and t(1) / (N * max_r t(N, r)).  Tells how much of the 8-GPU scaling target is lost to small grids, launch
overhead and the slowest tile before any communication.  One JSON line per (config, N).

--mode samples: the other way to divide a frame (hiprz.h: hiprz_set_shard_mode, distributed.py: ShardedFrame.reduce) — every rank renders the
WHOLE frame on its own seed stream and exports its RGBA32F accumulators for ONE reduce(sum) per step.  A rank's step is then a whole-frame
step whatever N is; the line gives the aggregate speed-up N * t(1) / max_r t(N, r) (t(1) = the one-GPU step: 8 passes + tone map) and what
sample sharding adds per step on a rank (the export: a 33 MB device copy at 1080p) and on rank 0 (untile + tone map of the sum, on the
collective's stream); the reduce itself crosses xGMI and cannot be rehearsed on one GPU."""
import argparse, json, os, sys, time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from rayzath_amd import scenes
from rayzath_amd.distributed import ShardedFrame, sample_shard_seed
from rayzath_amd.engine import Context, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="B", help="comma-separated presets of rayzath_amd/scenes.py")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--shards", default="1,2,4,8")
ap.add_argument("--traversal", type=int, default=-1)
ap.add_argument("--pipeline", type=int, default=-1)
ap.add_argument("--streams", type=int, default=1, help="streams per shard: Context([0] * streams), tiles interleaved, one scene copy")
ap.add_argument("--lds-scene", type=int, default=-1, help="hiprz_set_lds_scene: 0 never stage the scene in LDS (the cooperative walks from global memory), -1 per scene")
ap.add_argument("--mode", default="tiles", choices=["tiles", "samples"])
ap.add_argument("--tree", type=int, default=0, help="hiprz_set_tree: 0 the snapshot's mesh trees, 1 host SAH rebuild, 2 / 3 built on the device (Morton order / SAH), 4 the hosts' default")
args = ap.parse_args()
dev = torch.device("cuda", 0)
for config in args.config.split(","):
    preset = scenes.CONFIGS[config]
    world_scene = preset["build"]()
    flat, cam = flatten(world_scene), camera_struct(world_scene.camera)
    cfg = RenderConfig(tracing=Tracing(preset["max_depth"], 8)).struct()
    base = None
    for n in [int(v) for v in args.shards.split(",")]:
        per_shard, kernel_ms = [], []
        for r in range(n):
            ctx = Context([0] * args.streams) if args.streams > 1 else Context(0)
            ctx.set_traversal_mode(args.traversal)
            if args.pipeline >= 0:
                ctx.set_pipeline(args.pipeline)
            if args.lds_scene >= 0:
                ctx.set_lds_scene(args.lds_scene)
            samples = args.mode == "samples" and n > 1
            if not samples:
                ctx.set_shard(r, n)
            ctx.set_tree(args.tree)
            rank_cfg = RenderConfig(tracing=Tracing(preset["max_depth"], 8), seed=sample_shard_seed(20240501, r)).struct() if samples else cfg
            ctx.upload_scene(flat), ctx.upload_camera(cam), ctx.set_config(rank_cfg)
            frame = ShardedFrame(ctx, r, n, cam.width, cam.height, None, dev, mode="samples" if samples else "tiles")

            def step():
                ctx.render(8)
                if samples:   # everything a rank does per step except the reduce: the accumulators into the collective's buffer
                    ctx.export_accum_tiles(frame.local.data_ptr(), frame.local.numel() * 4)
                    return
                ctx.tonemap()
                ctx.export_rgba8_tiles(frame.local8.data_ptr(), frame.local8.numel() * 4)

            for _ in range(3):
                step()
            ctx.sync()
            ctx.kernel_time_ms()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            ctx.sync()
            per_shard.append((time.perf_counter() - t0) / args.steps * 1e3)
            k_ms, k_n = ctx.kernel_time_ms()  # hip events around every render batch (on the stream, no host gaps)
            kernel_ms.append(k_ms / max(k_n, 1) * 8)
            if samples and r == 0:   # what rank 0 adds behind the reduce, on the collective's stream: untile + tone map of the summed frame
                frame.rank, frame.world = 0, 1
                frame.reduce(), ctx.sync()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    ctx.untile_gathered(frame.local.data_ptr(), frame.n_parts, frame.part_capacity * 16, 16, frame.image.data_ptr(), None)
                    ctx.tonemap_image(frame.image.data_ptr(), frame.rgba8.data_ptr(), None)
                ctx.sync()
                root_ms = (time.perf_counter() - t0) / args.steps * 1e3
            ctx.close()
        worst = max(per_shard)
        base = base or worst
        if args.mode == "samples" and n > 1:
            print(json.dumps({"config": config, "mode": "samples", "ranks": n, "ms_per_step_slowest_rank": round(worst, 4), "ms_per_step_mean": round(sum(per_shard) / n, 4),
                              "ms_per_rank": [round(v, 4) for v in per_shard], "one_gpu_ms_per_step": round(base, 4), "samples_per_pixel_per_step": 8 * n,
                              "kernel_side_speedup": round(n * base / worst, 3), "kernel_side_efficiency": round(base / worst, 3),
                              "export_ms_per_step": round(worst - kernel_ms[per_shard.index(worst)], 4), "rank0_untile_and_tonemap_ms": round(root_ms, 4),
                              "accumulator_bytes_reduced_per_step": int(frame.local.numel() * 4)}), flush=True)
            continue
        print(json.dumps({"config": config, "mode": "tiles", "shards": n, "ms_per_step_slowest_shard": round(worst, 4), "ms_per_step_mean": round(sum(per_shard) / n, 4),
                          "ms_per_shard": [round(v, 4) for v in per_shard], "render_batch_ms_on_stream": [round(v, 4) for v in kernel_ms],
                          "ideal_ms": round(base / n, 4), "kernel_side_speedup": round(base / worst, 3), "kernel_side_efficiency": round(base / (n * worst), 3)}), flush=True)
