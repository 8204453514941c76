#!/usr/bin/env python3
"""bench.py — Mrays/s of the HIPGPU path-tracing backend on BASELINE.json's config B
(Cornell box 1920x1080, max depth 8), 1..8 GPUs of one node.

A *step* is one `renderWorld`-equivalent per GPU: `rpp` = 8 passes (engine_parts.hpp:84 default; every pass traces one path
segment per pixel of the GPU's share), a tone map, and — for N > 1 — the collective that brings the frame to rank 0 over RCCL.
rays = path segments, exactly the reference's own counter (`traced_rays += W*H` per pass, cpu_engine_renderer.cpp:173; shadow
rays are not counted).  value = rays traced by ALL ranks in the timed steps / seconds / 1e6.

How N GPUs divide the frame (`--shard-mode`; printed as `shard_mode`):
  samples (default for N > 1)  every rank renders the WHOLE frame on its own seed stream, ONE reduce(sum) of the RGBA32F accumulators
                               per step to rank 0, which tone-maps the sum: a step adds N * 8 samples per pixel; per-GPU work is fixed as
                               N grows ("scaling": "weak"); the frame is the sum of the ranks' one-GPU frames
  tiles                        the frame's 32x8 tiles interleaved over the ranks, one gather of the tone-mapped tiles per step: a step adds
                               8 samples per pixel whatever N is ("strong"); the frame is the one-GPU frame bit for bit.  Its kernel-side
                               speed-up at N = 8 is capped at 5.1 - 5.6 (D: 2.7) by the slowest tile's sequential chain of passes
At N > 1 the other mode is measured too and reported under `other_shard_mode`.

Top-level `value`, `ms_per_step`, `timing`, `roofline` all describe ONE packaging: one context and one stream per GPU, whole-share
launches (so `roofline.avg_launch_us` <= `ms_per_step`).  The packaging the Engine hosts use by default (two streams per GPU for scenes
without lights, one scene copy) is measured with the same protocol and reported beside it as `hosts_default_packaging`.

Extra objects on the JSON line:
  roofline      the dominant kernel: algorithmic bytes per launch (SURVEY.md §8d formula on the work counters of an instrumented run of
                the same kernel) divided by its average launch duration, measured with hip events on the render stream over the timed
                region.  `bound` names what the kernel's counters say limits it; `counters_from` says where valu_busy / traffic come from
                (committed counter passes of the builder's profiling run — NOT measured in this run).
  cpu_baseline  the CPU oracle (oracle/, a port of cpu_engine_kernel; uncalibrated against the reference, which cannot be built here)
                timed on this host's cores on a bounded sample of the same workload.  Reported baseline only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RPP = 8
PEAK_HBM_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling ~6.3 TB/s


def algorithmic_bytes(c):
    """B = 146*segments + 32*box tests + 36*triangle tests + 64*hits + 4*texel
    fetches + 64*light samples (state+accumulator read and written once per segment; 32-B
    node, 36-B triangle, 64-B shading record, 64-B light record)."""
    return (146 * c["segments"] + 32 * c["box_tests"] + 36 * c["tri_tests"] + 64 * c["hits"] + 4 * c["texel_fetches"]
            + 64 * c["light_samples"])


def cpu_baseline(flat, cam, cfg, budget_s=12.0):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle

    cores = len(os.sched_getaffinity(0))
    try:  # the GPU box gives this job a CPU share smaller than the host (cgroup v2 quota)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    ref = oracle.OracleRenderer(flat, cam, cfg)
    ref.render(1, threads=cores)  # first pass (warm-up, as headless.cpp:203 does)
    passes, t0 = 0, time.perf_counter()
    while True:
        ref.render(1, threads=cores)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or passes >= 4096:
            break
    rays = passes * cam.width * cam.height
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "calibration": "none: the reference's cpu_engine_kernel.cpp cannot be built in this image (un-vendored Math / Graphics headers), so the port is timed as it is",
            "sample": f"{passes} cumulative passes of the same {cam.width}x{cam.height} depth-{cfg.max_depth} frame "
                      f"({rays} path segments, {dt:.1f} s) after one warm-up pass"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40, help="steps per timed repeat (each step = 8 passes: >= 64 passes per repeat from 8 steps on)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="the K timed steps are repeated at least this many times; the line reports the median repeat")
    ap.add_argument("--min-seconds", type=float, default=5.0, help="keep repeating the K timed steps until the repeats add up to this much timed wall (SURVEY.md 8d: >= 5 s)")
    ap.add_argument("--config", default="B", help="scene preset of rayzath_amd/scenes.py (B = the quoted config)")
    ap.add_argument("--shard-mode", default="auto", choices=["auto", "samples", "tiles"],
                    help="how N > 1 GPUs divide the frame: samples = whole frame per rank on its own seed stream + one reduce(sum) of the accumulators per step; "
                         "tiles = interleaved 32x8 tiles + one gather per step; auto = samples (tile sharding's kernel-side speed-up is capped by its slowest tile's chain of passes, DESIGN.md §7)")
    ap.add_argument("--no-other-mode", action="store_true", help="N > 1: do not measure the other shard mode beside the chosen one")
    ap.add_argument("--traversal", type=int, default=-1, help="-1 per-scene choice (default), 1 nested walk with LDS stack, 2 workgroup-binned, 3 skip links (front to back, cooperative triangle phase)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: collective on the render stream instead of overlapping it with the next step's rendering")
    ap.add_argument("--pipeline", type=int, default=-1, help="0 fused pass kernel, 1 trace + shade kernels, 2 resident batch kernel (-1: chosen per scene)")
    ap.add_argument("--ray-sort", type=int, default=-1, help="-1 auto, 0 off, 1 on")
    ap.add_argument("--walk-order", type=int, default=-1, help="mesh child order of the skip-link walk: 0 reference order, 1 front to back (-1: library default)")
    ap.add_argument("--streams", type=int, default=-1, help="streams per GPU of `hosts_default_packaging`: every rank's share interleaved over this many contexts-with-a-stream on its GPU sharing one scene copy (-1 = the Engine hosts' default, rayzath_amd.engine.default_streams: 2 for scenes without lights, else 1; 1 = do not measure it)")
    ap.add_argument("--mode", type=int, default=0, help="hiprz_set_mode flags: 0 = the CPU kernel (the parity-checked default), 63 = every behaviour of the reference's CUDA engine")
    ap.add_argument("--tree", type=int, default=4, help="hiprz_set_tree: 4 the Engine hosts' default (the scene's own trees when it is staged in LDS, else built on the device with a binned SAH), 0 the scene's (reference) mesh trees, 1 rebuilt on the host with a binned SAH, 2 / 3 built on the device in Morton order / with a binned SAH; frames are the same under all of them")
    ap.add_argument("--no-xcd-swizzle", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify-gather", action="store_true", help="N > 1: check the assembled frame — tiles: against an unsharded render, bit for bit; samples: against the sum of one-GPU renders on the ranks' seeds")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N ranks share GPU 0 and talk over gloo: exercises the sharded path where only one GPU exists")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` as typed: this process touches no GPU — it starts one rank per GPU (torch.distributed.run, the
        # launcher the contract names) as a child, lets rank 0's JSON line through on stdout and leaves with the child's exit code.
        # --standalone: the launcher owns its rendezvous port (no port picked here that another process could take in between).
        import subprocess

        env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    import numpy as np
    import torch
    import torch.distributed as dist

    from rayzath_amd import scenes
    from rayzath_amd.distributed import ShardedFrame, sample_shard_seed
    from rayzath_amd.engine import Context, RenderConfig, Tracing, default_streams
    from rayzath_amd.scene import camera_struct, flatten

    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    preset = scenes.CONFIGS[args.config]
    scene_world = preset["build"]()
    flat, cam = flatten(scene_world), camera_struct(scene_world.camera)
    base_config = RenderConfig(tracing=Tracing(preset["max_depth"], RPP))
    cfg = base_config.struct()
    W, H = cam.width, cam.height
    shard_mode = args.shard_mode if args.shard_mode != "auto" else ("samples" if world > 1 else "tiles")

    def config_of(mode, r=rank):
        """The rank's render config: under sample sharding every rank draws from a seed stream of its own."""
        c = base_config.struct()
        if mode == "samples":
            c.seed = sample_shard_seed(base_config.seed, r)
        return c

    def make_context(devices, mode, tree=None, r=rank, n=world):
        c = Context(devices)
        c.set_traversal_mode(args.traversal)
        if args.pipeline >= 0:
            c.set_pipeline(args.pipeline)
        c.set_ray_sort(args.ray_sort)
        if args.walk_order >= 0:
            c.set_walk_order(args.walk_order)
        if args.no_xcd_swizzle:
            c.set_xcd_swizzle(False)
        c.set_tree(args.tree if tree is None else tree)
        if args.mode:
            c.set_mode(args.mode)
        if mode == "tiles":
            c.set_shard(r, n)   # samples: the whole frame (shard 0 of 1, the default)
        c.upload_scene(flat)
        c.upload_camera(cam)
        c.set_config(config_of(mode, r))
        return c

    def fence(ctx):
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_repeats(ctx, one_step):
        """EXACTLY `steps` steps between two fences, repeated until `repeats` repeats and `min_seconds` of timed wall have been collected.
        Every rank sees the same (max-reduced) samples, so all of them stop after the same repeat."""
        out = []
        while len(out) < max(args.repeats, 1) or (sum(out) < args.min_seconds and len(out) < 4096):
            fence(ctx)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                one_step()
            fence(ctx)
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            out.append(float(t.item()))
        return out

    def verify(ctx, frame, mode):
        """The frame the job assembles after 1 + RPP passes per rank against one-GPU renders (rank 0 compares)."""
        ctx.render(RPP)
        if mode == "samples":
            frame.reduce()
            frame.sync()
            img = frame.image.cpu().numpy() if rank == 0 else None
        else:
            img = frame.gather_accum()
            frame.sync()
            img = img.cpu().numpy() if rank == 0 else None
        if rank == 0:
            expected = None
            for r in range(world if mode == "samples" else 1):
                ref = Context(local_rank)
                ref.set_traversal_mode(args.traversal)
                ref.upload_scene(flat), ref.upload_camera(cam), ref.set_config(config_of(mode, r))
                ref.render(1 + RPP)
                a = ref.read_accum()
                expected = a if expected is None else expected + a
                ref.close()
            if mode == "tiles" or world == 2:   # (a sum of two is the same in either order)
                assert np.array_equal(img, expected), f"{mode}: assembled frame differs from the one-GPU frame(s)"
            else:   # the reduce adds the ranks in the collective's order
                assert np.array_equal(img[..., 3], expected[..., 3]), "samples: finished-path counts differ from the sum of the one-GPU frames"
                assert np.allclose(img, expected, rtol=1e-5, atol=1e-6), "samples: reduced frame differs from the sum of the one-GPU frames"
        total = frame.ray_count()  # all-reduce over the ranks (every rank calls it)
        want = (1 + RPP) * W * H * (world if mode == "samples" else 1)
        assert total == want, f"ray counters of the ranks add up to {total}, not {want}"

    def measure(k_streams, mode, keep=False):
        """One packaging through the whole protocol: context(s) -> first pass -> warm-up -> timed repeats.  Returns the figures (and, with
        `keep`, the context for the roofline's instrumented runs)."""
        ctx = make_context([local_rank] * k_streams if k_streams > 1 else local_rank, mode)
        frame = ShardedFrame(ctx, rank, world, W, H, dist if world > 1 else None, device, overlap=not args.no_overlap, mode=mode)

        def step():
            ctx.render(RPP)
            if world == 1:
                ctx.tonemap()
            elif mode == "samples":
                frame.reduce()
            else:
                frame.gather()

        ctx.render(1)  # renderFirstPass; the timed steps are cumulative passes, as in steady-state rendering
        if args.verify_gather and world > 1:
            verify(ctx, frame, mode)
        for _ in range(args.warmup):
            step()
        fence(ctx)
        ctx.kernel_time_ms()  # drop the warm-up launches from the event log
        alpha_before = float(ctx.read_accum()[..., 3].sum()) if rank == 0 and world == 1 else None
        passes_before = ctx.pass_count()
        t_all0 = time.perf_counter()
        samples = timed_repeats(ctx, step)
        timed_wall = time.perf_counter() - t_all0
        elapsed = sorted(samples)[len(samples) // 2]
        kernel_ms, launches = ctx.kernel_time_ms()   # hip events around every render batch of the timed repeats (head stream)
        alpha_after = float(ctx.read_accum()[..., 3].sum()) if alpha_before is not None else None
        rays_per_step = RPP * W * H * (world if mode == "samples" else 1)
        out = {"streams": k_streams, "shard_mode": mode if world > 1 else "single GPU", "value": args.steps * rays_per_step / elapsed / 1e6, "unit": "Mrays/s",
               "ms_per_step": elapsed / args.steps * 1e3, "rays_per_step": rays_per_step, "samples_per_pixel_per_step": rays_per_step // (W * H),
               "repeats": len(samples), "timed_seconds": sum(samples), "repeat_seconds_min_median_max": [min(samples), elapsed, max(samples)],
               "passes_timed": ctx.pass_count() - passes_before, "timed_wall_seconds": timed_wall, "kernel_ms": kernel_ms, "launches": launches,
               "spp_per_s": (alpha_after - alpha_before) / (W * H) / sum(samples) if alpha_before is not None else None}
        if keep:
            return out, ctx
        fence(ctx)
        ctx.close()
        return out, None

    # ---- the line's packaging: one context, one stream per GPU
    main_run, ctx = measure(1, shard_mode, keep=True)
    elapsed_step_s = main_run["ms_per_step"] / 1e3
    kernel_ms, launches = main_run["kernel_ms"], main_run["launches"]

    # ---- outside the timed region: per-kernel events (eager launches), host readback, work counters
    ctx.time_kernels(True)
    trace_ms = shade_ms = 0.0
    timed_passes = 0
    for _ in range(3):
        ctx.render(RPP)
        b = ctx.kernel_breakdown_ms()
        trace_ms, shade_ms, timed_passes = trace_ms + b[0], shade_ms + b[1], timed_passes + b[2]
    ctx.time_kernels(False)
    breakdown = (trace_ms, shade_ms, timed_passes)
    ctx.kernel_time_ms()
    end_to_end = None
    if world == 1:  # what the reference's renderWorld hands back per call: the tone-mapped frame in host memory
        ctx.render(RPP), ctx.tonemap(), ctx.read_rgba8()
        fence(ctx)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.render(RPP)
            ctx.tonemap()
            ctx.read_rgba8()
        fence(ctx)
        e2e = time.perf_counter() - t0
        end_to_end = {"value": args.steps * RPP * W * H / e2e / 1e6, "unit": "Mrays/s", "ms_per_step": e2e / args.steps * 1e3,
                      "includes": "render + tone map + hiprz_read_rgba8 into host memory (8.3 MB per step over PCIe, synchronous)"}
        ctx.kernel_time_ms()

    # ---- the hosts' default packaging (rayzath_amd.engine.default_streams, Hip::Engine::defaultStreams): the rank's share over K streams on its
    # GPU (hiprz_create_multi with the device named K times: tiles interleaved, ONE scene copy shared) — one stream's sorts, pass bookkeeping
    # and kernel tails run beside another's walks.  Same protocol, same shard mode and collective.
    hosts_default = None
    k_streams = args.streams if args.streams > 0 else default_streams(len(flat.spot_lights) + len(flat.direct_lights))
    if k_streams > 1:
        fence(ctx)
        hosts_default, _ = measure(k_streams, shard_mode)
        for k in ("kernel_ms", "launches", "spp_per_s", "passes_timed", "timed_wall_seconds"):
            hosts_default.pop(k)
        hosts_default["note"] = ("same steps, same protocol; every rank's share interleaved over %d contexts on its GPU, each with its own stream, one scene copy "
                                 "(what Hip::Engine / rayzath_amd.engine.Engine do by default for scenes without lights)" % k_streams)
    # ---- N > 1: the other way to divide the frame, one stream per GPU, same protocol
    other_mode = None
    if world > 1 and not args.no_other_mode:
        fence(ctx)
        other_mode, _ = measure(1, "tiles" if shard_mode == "samples" else "samples")
        for k in ("kernel_ms", "launches", "spp_per_s", "passes_timed", "timed_wall_seconds"):
            other_mode.pop(k)

    if rank == 0:
        # work counters of the same kernels, same state
        counters = ctx.render_counted(RPP)
        ctx.kernel_time_ms()
        bytes_per_pass = algorithmic_bytes(counters) / RPP
        avg_pass_s = kernel_ms / 1e3 / max(launches, 1)
        pipeline = ctx.pipeline()
        split = pipeline == 1
        walk_order = args.walk_order if args.walk_order >= 0 else 1
        front_to_back = split and ctx.traversal_mode() == 3 and (walk_order != 0 or bool(args.mode & 31))
        trace_bytes = lambda c: (60 * c["segments"] + 32 * (c["box_tests"] - c["shadow_box_tests"]) + 36 * (c["tri_tests"] - c["shadow_tri_tests"])) / RPP
        traversal_kernel = eager_kernel_us = None
        if pipeline == 2 and breakdown[2]:
            # dominant (only) kernel: the resident batch kernel — one launch takes every tile through the RPP passes of the
            # step.  Algorithmic bytes: SURVEY.md §8d's per-segment figure x the segments of the launch.
            # (scenes staged in LDS: rz_batch_kernel, a workgroup per tile, the reference's visiting order; scenes that are not: rz_wave_batch_kernel,
            # a wave per 8x8 pixels on the cooperative front-to-back walk — `counters` are then the tests that walk executes)
            kernel_name = "rz_batch_kernel (resident: all passes of a step)" if ctx.traversal_mode() != 3 else "rz_wave_batch_kernel (per-wave resident: all passes of a step)"
            # duration: the hip events the context records on its stream around EVERY render batch of the timed repeats (the batch
            # kernel + the one-thread kernel that advances the pass index) — the launches `value` was timed on, not a separate run
            kernel_s = kernel_ms / 1e3 / max(launches // RPP, 1)
            eager_kernel_us = breakdown[0] / (breakdown[2] / RPP) * 1e3  # the same kernel between its own two events, after the timed region
            kernel_bytes = algorithmic_bytes(counters)
            # the BVH-traversal kernel on its own (north_star's 30 % target): the same step through the split pipeline
            ctx.set_pipeline(1)
            ctx.render(RPP)
            ctx.time_kernels(True)
            ctx.render(RPP)
            tb = ctx.kernel_breakdown_ms()
            ctx.time_kernels(False)
            ctx.set_pipeline(args.pipeline)
            if tb[2]:
                t_bytes = trace_bytes(counters)
                t_s = tb[0] / 1e3 / tb[2]
                traversal_kernel = {"kernel": "rz_trace_kernel (closest-hit walk, split pipeline)", "avg_launch_us": t_s * 1e6,
                                    "algorithmic_bytes_per_launch": t_bytes, "achieved": t_bytes / t_s / 1e9,
                                    "frac": t_bytes / t_s / 1e9 / PEAK_HBM_GBS, "shade_kernel_avg_launch_us": tb[1] / tb[2] * 1e3}
        elif split and breakdown[2]:
            # dominant kernel = the BVH-traversal kernel.  Its algorithmic bytes: the ray it reads (40 B of path state) and
            # the hit record it writes (20 B) per segment + 32 B per box test + 36 B per triangle test of the closest-hit
            # walk (shadow-ray tests run in their own kernel and are not counted here).
            mode3 = ("rz_trace_coop_compat_kernel" if args.mode & 31 else "rz_trace_coop_kernel") if front_to_back else "rz_trace_skip_kernel"
            kernel_name = {3: mode3}.get(ctx.traversal_mode(), "rz_trace_kernel") + " (closest-hit walk)"
            kernel_s = breakdown[0] / 1e3 / breakdown[2]
            kernel_bytes = trace_bytes(counters)
            if front_to_back and not (args.mode & 31):
                # `counters` walked in the reference's child order.  The timed kernel walks front to back and reaches the same hits with
                # fewer tests: the roofline is priced on the tests it EXECUTED.
                ctx.set_walk_order(2)
                counters = ctx.render_counted(RPP)
                ctx.set_walk_order(walk_order)
                ctx.kernel_time_ms()
                kernel_bytes = trace_bytes(counters)
        else:
            kernel_name, kernel_s, kernel_bytes = "rz_pass_kernel (fused pass)", avg_pass_s, bytes_per_pass
        # the reference ALGORITHM's work on the same scene: first-child-then-second walks of the snapshot's (reference) trees — a bare
        # context (HIPRZ_TREE_REFERENCE, walk order 0), whatever trees and order the timed context uses
        reference_algorithm = None
        if world == 1 and not (args.mode & 31):
            bare = Context(local_rank)
            bare.set_walk_order(0)
            bare.upload_scene(flat), bare.upload_camera(cam), bare.set_config(cfg)
            bare.render(1)
            rc = bare.render_counted(RPP)
            bare.close()
            ref_bytes = trace_bytes(rc) if split else algorithmic_bytes(rc)
            reference_algorithm = {"note": "work of the reference's first-child-then-second walk of the snapshot's (reference builder's) trees over passes 2..9 of the same frame, divided by the timed kernel's duration",
                                   "box_tests_per_segment": rc["box_tests"] / max(rc["segments"], 1), "tri_tests_per_segment": rc["tri_tests"] / max(rc["segments"], 1),
                                   "kernel_bytes_per_launch": ref_bytes, "achieved": ref_bytes / kernel_s / 1e9, "frac": ref_bytes / kernel_s / 1e9 / PEAK_HBM_GBS}
        achieved = kernel_bytes / kernel_s / 1e9
        # HBM bytes and SQ figures of the same kernel: counter passes of the builder's profiling run of this command, committed under
        # profiles/ (tools/round_profiles.sh) — replayed here, not measured in this run
        traffic, compute, counters_from = None, {}, None
        tpath, spath = os.path.join(ROOT, "profiles", f"traffic_{args.config}.json"), os.path.join(ROOT, "profiles", f"sq_{args.config}.json")
        short = kernel_name.split(" ")[0]
        if world == 1 and os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(short, {}).get("hbm_bytes_per_launch")
        if world == 1 and os.path.exists(spath):
            compute = json.load(open(spath)).get(short, {})
        if traffic is not None or compute:
            meta = json.load(open(spath)).get("_meta", {}) if os.path.exists(spath) else {}
            counters_from = (f"profiles/sq_{args.config}.json, profiles/traffic_{args.config}.json: rocprofv3 --pmc passes of `bench.py --config {args.config} --streams 1` "
                             f"run by the builder ({meta.get('collected', 'round 3')}), committed and replayed here — NOT measured in this run")
        valu_bound = bool(compute.get("valu_busy"))
        result = {
            "metric": "Mrays/s (path segments, primary+secondary) at 1920x1080 depth 8" if args.config == "B" else f"Mrays/s config {args.config}",
            "value": main_run["value"], "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_run["ms_per_step"], "higher_is_better": True, "scaling": "weak" if (world > 1 and shard_mode == "samples") else "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": preset["note"], "resolution": [W, H], "max_depth": preset["max_depth"], "passes_per_step": RPP,
                       "triangles": int(len(flat.tris)), "instances": int(len(flat.instances)),
                       "sharding": ("single GPU" if world == 1 else
                                    f"samples: every one of {world} GPUs renders the whole frame on its own seed stream, one reduce(sum) of the RGBA32F accumulators to rank 0 per step ({world * RPP} samples per pixel per step)"
                                    if shard_mode == "samples" else f"tiles: interleaved 32x8 tiles over {world} GPUs, one gather of the tone-mapped tiles to rank 0 per step"),
                       "traversal": {1: "lds-stack", 2: "workgroup-binned", 3: "skip-links"}[ctx.traversal_mode()],
                       "mesh_trees": ["reference builder (scene snapshot)", "binned SAH, rebuilt on the host at upload", "built on the device at upload (Morton order)",
                                      "built on the device at upload (binned SAH)"][ctx.tree()] + (" — hiprz_set_tree(HIPRZ_TREE_AUTO), the Engine hosts' default" if args.tree == 4 else ""),
                       "integrator": "CPU kernel (parity-checked)" if not (args.mode & 31) else f"CUDA-compat flags {args.mode} (hiprz_set_mode)",
                       "pipeline": {0: "fused (one kernel per pass)", 1: "trace+shade (two kernels per pass)", 2: "resident (one kernel per step)"}[pipeline]},
            "shard_mode": main_run["shard_mode"], "streams_per_gpu": 1, "rays_per_step": main_run["rays_per_step"],
            "value_from": "one context, one stream per GPU: whole-share launches — the packaging `roofline` and `timing` describe too",
            "timing": {"protocol": f"{main_run['repeats']} repeats of exactly {args.steps} steps ({args.steps * RPP} passes per GPU) between barrier + synchronize fences (repeated until >= {args.min_seconds:g} s of timed wall); value = median repeat",
                       "repeat_seconds_min_median_max": main_run["repeat_seconds_min_median_max"], "repeats": main_run["repeats"], "timed_seconds": main_run["timed_seconds"],
                       "passes_timed": main_run["passes_timed"], "timed_wall_seconds": main_run["timed_wall_seconds"]},
            "spp_per_s": main_run["spp_per_s"],
            "end_to_end": end_to_end,
            "hosts_default_packaging": hosts_default,
            "other_shard_mode": other_mode,
            "roofline": {"bound": "valu-issue" if valu_bound else "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
                         "traffic": traffic, "kernel": kernel_name, "avg_launch_us": kernel_s * 1e6,
                         "roofline_priced": "hbm (SURVEY.md §8d: algorithmic bytes of the launch / its duration / 8 TB/s); `traffic` is what physically crossed HBM per launch",
                         "valu_busy": compute.get("valu_busy"), "lanes_active": compute.get("lanes_active"),
                         "valu_instr_per_wave": compute.get("valu_instr_per_wave"),
                         "bound_measured": "vector-instruction issue: valu_busy of the chip's VALU issue capacity at lanes_active of the lanes (SQ counters); the algorithmic bytes of a scene that lives in LDS / L2 never reach HBM" if valu_bound else None,
                         "counters_from": counters_from,
                         "duration_from": ("hip events around every render batch of the timed repeats (%d launches)" % (launches // RPP)) if pipeline == 2 else
                                          "hip events around the kernel in an eager, event-instrumented batch after the timed region (a captured graph cannot be timed from inside)",
                         "eager_avg_launch_us": eager_kernel_us,
                         "algorithmic_bytes_per_launch": kernel_bytes,
                         "priced_on": "tests executed by the timed kernel" + (" (front-to-back walk)" if front_to_back else ""),
                         "segments_per_launch": counters["segments"] / (1 if pipeline == 2 else RPP),
                         "traversal_kernel": traversal_kernel,
                         "shade_kernel_avg_launch_us": breakdown[1] / breakdown[2] * 1e3 if split and breakdown[2] else None,
                         "whole_pass": {"avg_us": avg_pass_s * 1e6, "algorithmic_bytes": bytes_per_pass,
                                        "achieved": bytes_per_pass / avg_pass_s / 1e9, "frac": bytes_per_pass / avg_pass_s / 1e9 / PEAK_HBM_GBS,
                                        "note": "all kernels of a pass; bytes of the executed walks (closest-hit + shadow rays + shading)"},
                         "box_tests_per_segment": counters["box_tests"] / max(counters["segments"], 1),
                         "tri_tests_per_segment": counters["tri_tests"] / max(counters["segments"], 1),
                         "mesh_walk_order": None if not (split and ctx.traversal_mode() == 3) else ("front to back" if walk_order else "reference child order"),
                         "reference_algorithm": reference_algorithm},
        }
        # the line's own cross-check: the dominant kernel's launches of a step fit into the step they are part of
        per_step_us = result["roofline"]["avg_launch_us"] * (1 if pipeline == 2 else RPP)
        result["roofline"]["dominant_kernel_us_per_step"] = per_step_us
        result["roofline"]["fits_in_step"] = bool(per_step_us <= result["ms_per_step"] * 1e3 * 1.01)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(flat, cam, cfg)
        print(json.dumps(result), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
