#!/usr/bin/env python3
"""bench.py — Mrays/s of the HIPGPU path-tracing backend on BASELINE.json's config B
(Cornell box 1920x1080, max depth 8), 1..8 GPUs of one node.

A *step* is one `renderWorld`-equivalent: `rpp` = 8 passes (engine_parts.hpp:84 default;
every pass traces one path segment per pixel), a tone map, and — for N > 1 — the gather
of the accumulators to rank 0 over RCCL.  rays = path segments, exactly the reference's own
counter (`traced_rays += W*H` per pass, cpu_engine_renderer.cpp:173; shadow rays are not
counted).  value = steps * rpp * W * H / seconds / 1e6 over the whole job; the frame is
fixed, so more GPUs split the same work ("strong" scaling).  `value` is measured on the packaging the
Engine hosts use by default — for scenes without lights every GPU's share runs on two streams (one scene
copy) — and `single_stream` beside it on one stream per GPU: whole-frame launches, the ones `roofline` prices.

Extra objects on the JSON line:
  roofline      HBM roofline of the pass kernel: algorithmic bytes per launch (SURVEY.md §8d
                formula on the work counters of an instrumented run of the same kernel) divided by
                the average launch duration measured with hip events on the render stream.
  cpu_baseline  the CPU oracle (oracle/, a port of cpu_engine_kernel) timed on this host's cores
                on a bounded sample of the same workload.  Reported baseline only.
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
    ap.add_argument("--traversal", type=int, default=-1, help="-1 per-scene choice (default), 1 nested walk with LDS stack, 2 workgroup-binned, 3 skip links (front to back, cooperative triangle phase)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: gather on the render stream instead of overlapping it with the next step's rendering")
    ap.add_argument("--pipeline", type=int, default=-1, help="0 fused pass kernel, 1 trace + shade kernels, 2 resident batch kernel (-1: chosen per scene)")
    ap.add_argument("--ray-sort", type=int, default=-1, help="-1 auto, 0 off, 1 on")
    ap.add_argument("--walk-order", type=int, default=-1, help="mesh child order of the skip-link walk: 0 reference order, 1 front to back (-1: library default)")
    ap.add_argument("--streams", type=int, default=-1, help="streams per GPU of the packaging `value` is measured on: every rank's tiles interleaved over this many contexts-with-a-stream on its GPU sharing one scene copy (-1 = the Engine hosts' default, rayzath_amd.engine.default_streams: 2 for scenes without lights, else 1); the single-stream figure is always measured too (`single_stream`, `roofline`)")
    ap.add_argument("--mode", type=int, default=0, help="hiprz_set_mode flags: 0 = the CPU kernel (the parity-checked default), 63 = every behaviour of the reference's CUDA engine")
    ap.add_argument("--tree", type=int, default=4, help="hiprz_set_tree: 4 the Engine hosts' default (the scene's own trees when it is staged in LDS, else built on the device with a binned SAH), 0 the scene's (reference) mesh trees, 1 rebuilt on the host with a binned SAH, 2 / 3 built on the device in Morton order / with a binned SAH; frames are the same under all of them")
    ap.add_argument("--no-xcd-swizzle", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify-gather", action="store_true", help="check the gathered frame against an unsharded render (N > 1)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N ranks share GPU 0 and talk over gloo: exercises the sharded path where only one GPU exists")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` as typed: this process touches no GPU — it starts one rank per GPU (torch.distributed.run, the
        # launcher the contract names) as a child, lets rank 0's JSON line through on stdout and leaves with the child's exit code.
        import socket
        import subprocess

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    import torch
    import torch.distributed as dist

    from rayzath_amd import scenes
    from rayzath_amd.distributed import ShardedFrame
    from rayzath_amd.engine import Context, RenderConfig, Tracing
    from rayzath_amd.scene import camera_struct, flatten

    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    preset = scenes.CONFIGS[args.config]
    scene_world = preset["build"]()
    flat, cam = flatten(scene_world), camera_struct(scene_world.camera)
    cfg = RenderConfig(tracing=Tracing(preset["max_depth"], RPP)).struct()
    W, H = cam.width, cam.height

    def make_context(devices):
        c = Context(devices)
        c.set_traversal_mode(args.traversal)
        if args.pipeline >= 0:
            c.set_pipeline(args.pipeline)
        c.set_ray_sort(args.ray_sort)
        if args.walk_order >= 0:
            c.set_walk_order(args.walk_order)
        if args.no_xcd_swizzle:
            c.set_xcd_swizzle(False)
        c.set_tree(args.tree)
        if args.mode:
            c.set_mode(args.mode)
        c.set_shard(rank, world)
        c.upload_scene(flat)
        c.upload_camera(cam)
        c.set_config(cfg)
        return c

    ctx = make_context(local_rank)
    frame = ShardedFrame(ctx, rank, world, W, H, dist if world > 1 else None, torch.device("cuda", local_rank), overlap=not args.no_overlap)

    def step():
        ctx.render(RPP)
        if world > 1:
            frame.gather()
        else:
            ctx.tonemap()

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ctx.render(1)  # renderFirstPass; the timed steps are cumulative passes, as in steady-state rendering
    if args.verify_gather and world > 1:  # the assembled frame must equal what a single shard-less context renders
        ctx.render(RPP)
        img = frame.gather_accum()
        frame.sync()
        if rank == 0:
            ref = Context(local_rank)
            ref.set_traversal_mode(args.traversal)
            ref.upload_scene(flat), ref.upload_camera(cam), ref.set_config(cfg)
            ref.render(1 + RPP)
            import numpy as np
            assert np.array_equal(img.cpu().numpy(), ref.read_accum()), "gathered frame differs from the unsharded frame"
            ref.close()
        total_rays = frame.ray_count()  # all-reduce over the ranks (every rank calls it)
        assert total_rays == (1 + RPP) * W * H, f"ray counters of the shards add up to {total_rays}, not {(1 + RPP) * W * H}"
    for _ in range(args.warmup):
        step()
    fence()
    ctx.kernel_time_ms()  # drop the warm-up launches from the event log

    # ---- the timed region: EXACTLY `steps` steps between two fences, repeated `repeats` times; the line reports the median
    # repeat ("median of 5").  Nothing but step() runs between the fences.
    alpha_before = float(ctx.read_accum()[..., 3].sum()) if rank == 0 and world == 1 else None
    passes_before = ctx.pass_count()
    def timed_repeats(one_step):
        """EXACTLY `steps` steps between two fences, repeated until `repeats` repeats and `min_seconds` of timed wall have been collected.
        Every rank sees the same (max-reduced) samples, so all of them stop after the same repeat."""
        out = []
        while len(out) < max(args.repeats, 1) or (sum(out) < args.min_seconds and len(out) < 4096):
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                one_step()
            fence()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            out.append(float(t.item()))
        return out

    t_all0 = time.perf_counter()
    samples = timed_repeats(step)
    timed_wall = time.perf_counter() - t_all0
    elapsed = sorted(samples)[len(samples) // 2]
    kernel_ms, launches = ctx.kernel_time_ms()   # hip events around every render batch of the timed repeats
    alpha_after = float(ctx.read_accum()[..., 3].sum()) if alpha_before is not None else None
    passes_timed = ctx.pass_count() - passes_before

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
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.render(RPP)
            ctx.tonemap()
            ctx.read_rgba8()
        fence()
        e2e = time.perf_counter() - t0
        end_to_end = {"value": args.steps * RPP * W * H / e2e / 1e6, "unit": "Mrays/s", "ms_per_step": e2e / args.steps * 1e3,
                      "includes": "render + tone map + hiprz_read_rgba8 into host memory (8.3 MB per step over PCIe, synchronous)"}
        ctx.kernel_time_ms()
    # ---- the hosts' default packaging (rayzath_amd.engine.default_streams, Hip::Engine::defaultStreams): the rank's share of the frame over K
    # streams on its GPU (hiprz_create_multi with the device named K times: tiles interleaved, ONE scene copy shared) — one stream's sorts,
    # pass bookkeeping and kernel tails run beside another's walks.  The same protocol as above, sharding and gather included; when K > 1 this
    # is the line's `value` (what a host gets by default) and the single-stream figure — the one the per-launch roofline belongs to — is
    # reported beside it as `single_stream`.
    several_streams = None
    from rayzath_amd.engine import default_streams
    k_streams = args.streams if args.streams > 0 else default_streams(len(flat.spot_lights) + len(flat.direct_lights))
    if k_streams > 1:
        fence()
        fast = make_context([local_rank] * k_streams)
        fast_frame = ShardedFrame(fast, rank, world, W, H, dist if world > 1 else None, torch.device("cuda", local_rank), overlap=not args.no_overlap)

        def fast_step():
            fast.render(RPP)
            if world > 1:
                fast_frame.gather()
            else:
                fast.tonemap()

        fast.render(1)
        if args.verify_gather and world > 1:  # the frame assembled from every rank's streams == one unsharded context's
            fast.render(RPP)
            img = fast_frame.gather_accum()
            fast_frame.sync()
            if rank == 0:
                import numpy as np
                ref = Context(local_rank)
                ref.set_traversal_mode(args.traversal)
                ref.upload_scene(flat), ref.upload_camera(cam), ref.set_config(cfg)
                ref.render(1 + RPP)
                assert np.array_equal(img.cpu().numpy(), ref.read_accum()), "frame gathered from the ranks' streams differs from the unsharded frame"
                ref.close()
            assert fast_frame.ray_count() == (1 + RPP) * W * H
        for _ in range(args.warmup):
            fast_step()
        fast_samples = timed_repeats(fast_step)
        fast_elapsed = sorted(fast_samples)[len(fast_samples) // 2]
        several_streams = {"streams": k_streams, "value": args.steps * RPP * W * H / fast_elapsed / 1e6, "unit": "Mrays/s",
                           "ms_per_step": fast_elapsed / args.steps * 1e3, "repeats": len(fast_samples), "timed_seconds": sum(fast_samples),
                           "repeat_seconds_min_median_max": [min(fast_samples), fast_elapsed, max(fast_samples)],
                           "note": "same steps, same protocol; every rank's tiles interleaved over %d contexts on its GPU, each with its own stream, one scene copy (the Engine hosts' default for scenes without lights)" % k_streams}
        fence()
        fast.close()
    rays = args.steps * RPP * W * H
    result = None
    if rank == 0:
        spp_per_s = None
        if alpha_before is not None:   # finished paths per pixel per second over ALL the timed repeats
            spp_per_s = (alpha_after - alpha_before) / (W * H) / sum(samples)
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
        traversal_kernel = reference_algorithm = eager_kernel_us = None
        if pipeline == 2 and breakdown[2]:
            # dominant (only) kernel: the resident batch kernel — one launch takes every tile through the RPP passes of the
            # step.  Algorithmic bytes: SURVEY.md §8d's per-segment figure x the segments of the launch.  It walks in the
            # reference's order: executed work == the reference algorithm's work.
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
                # `counters` walked in the reference's child order (they equal the CPU kernel's).  The timed kernel walks front to
                # back and reaches the same hits with fewer tests: the roofline is priced on the tests it EXECUTED; the reference
                # algorithm's figure is kept beside it as context.
                ctx.set_walk_order(2)
                ex = ctx.render_counted(RPP)
                ctx.set_walk_order(walk_order)
                ctx.kernel_time_ms()
                ref_bytes = kernel_bytes
                kernel_bytes = trace_bytes(ex)
                reference_algorithm = {"note": "work of the reference's first-child-then-second walk on the same rays, divided by the timed kernel's duration",
                                       "box_tests_per_segment": counters["box_tests"] / max(counters["segments"], 1),
                                       "tri_tests_per_segment": counters["tri_tests"] / max(counters["segments"], 1),
                                       "kernel_bytes_per_launch": ref_bytes, "achieved": ref_bytes / kernel_s / 1e9, "frac": ref_bytes / kernel_s / 1e9 / PEAK_HBM_GBS}
                counters = ex
        else:
            kernel_name, kernel_s, kernel_bytes = "rz_pass_kernel (fused pass)", avg_pass_s, bytes_per_pass
        achieved = kernel_bytes / kernel_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", f"traffic_{args.config}.json")
        if os.path.exists(tpath) and world == 1:
            traffic = json.load(open(tpath)).get(kernel_name.split(" ")[0], {}).get("hbm_bytes_per_launch")
        # compute side of the same kernel (SQ counter passes of tools/pmc_sq.sh, committed as profiles/sq_<config>.json): these walks
        # are bound by vector-instruction issue at partial lane utilisation, not by HBM — the line says so next to the byte roofline
        compute = {}
        spath = os.path.join(ROOT, "profiles", f"sq_{args.config}.json")
        if os.path.exists(spath) and world == 1:
            compute = json.load(open(spath)).get(kernel_name.split(" ")[0], {})
        result = {
            "metric": "Mrays/s (path segments, primary+secondary) at 1920x1080 depth 8" if args.config == "B" else f"Mrays/s config {args.config}",
            "value": several_streams["value"] if several_streams else rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": several_streams["ms_per_step"] if several_streams else elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": preset["note"], "resolution": [W, H], "max_depth": preset["max_depth"], "passes_per_step": RPP,
                       "triangles": int(len(flat.tris)), "instances": int(len(flat.instances)),
                       "sharding": f"interleaved 32x8 tiles over {world} GPU(s), gather to rank 0 per step" if world > 1 else "single GPU",
                       "traversal": {1: "lds-stack", 2: "workgroup-binned", 3: "skip-links"}[ctx.traversal_mode()],
                       "mesh_trees": ["reference builder (scene snapshot)", "binned SAH, rebuilt on the host at upload", "built on the device at upload (Morton order)",
                                      "built on the device at upload (binned SAH)"][ctx.tree()] + (" — hiprz_set_tree(HIPRZ_TREE_AUTO), the Engine hosts' default" if args.tree == 4 else ""),
                       "integrator": "CPU kernel (parity-checked)" if not (args.mode & 31) else f"CUDA-compat flags {args.mode} (hiprz_set_mode)",
                       "pipeline": {0: "fused (one kernel per pass)", 1: "trace+shade (two kernels per pass)", 2: "resident (one kernel per step)"}[pipeline]},
            "timing": {"protocol": f"{len(samples)} repeats of exactly {args.steps} steps ({args.steps * RPP} passes) between barrier + synchronize fences (repeated until >= {args.min_seconds:g} s of timed wall); value = median repeat",
                       "repeat_seconds_min_median_max": [min(samples), elapsed, max(samples)], "repeats": len(samples), "timed_seconds": sum(samples),
                       "passes_timed": passes_timed, "timed_wall_seconds": timed_wall},
            "spp_per_s": spp_per_s,
            "end_to_end": end_to_end,
            "streams_per_gpu": k_streams, "value_from": "several_streams" if several_streams else "single_stream",
            "single_stream": {"value": rays / elapsed / 1e6, "unit": "Mrays/s", "ms_per_step": elapsed / args.steps * 1e3,
                              "note": "one context, one stream per GPU: whole-frame launches, what `roofline` and `timing` describe"},
            "several_streams": several_streams,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
                         "traffic": traffic, "kernel": kernel_name, "avg_launch_us": kernel_s * 1e6,
                         "valu_busy": compute.get("valu_busy"), "lanes_active": compute.get("lanes_active"),
                         "valu_instr_per_wave": compute.get("valu_instr_per_wave"),
                         "bound_measured": "vector-instruction issue (valu_busy of the chip's VALU issue capacity at lanes_active of the lanes; SQ counters, profiles/)" if compute.get("valu_busy") else None,
                         "duration_from": ("hip events around every render batch of the timed repeats (%d launches)" % (launches // RPP)) if pipeline == 2 else
                                          "hip events around the kernel in an eager, event-instrumented batch after the timed region (a captured graph cannot be timed from inside)",
                         "eager_avg_launch_us": eager_kernel_us,
                         "algorithmic_bytes_per_launch": kernel_bytes,
                         "priced_on": "tests executed by the timed kernel" + (" (front-to-back walk)" if front_to_back else " (= the reference algorithm's: same visiting order)"),
                         "segments_per_launch": counters["segments"] / (1 if pipeline == 2 else RPP),
                         "traversal_kernel": traversal_kernel,
                         "shade_kernel_avg_launch_us": breakdown[1] / breakdown[2] * 1e3 if split and breakdown[2] else None,
                         "whole_pass": {"avg_us": avg_pass_s * 1e6, "algorithmic_bytes": bytes_per_pass,
                                        "achieved": bytes_per_pass / avg_pass_s / 1e9, "frac": bytes_per_pass / avg_pass_s / 1e9 / PEAK_HBM_GBS,
                                        "note": "all kernels of a pass; bytes of the reference algorithm (closest-hit + shadow rays + shading)"},
                         "box_tests_per_segment": counters["box_tests"] / max(counters["segments"], 1),
                         "tri_tests_per_segment": counters["tri_tests"] / max(counters["segments"], 1),
                         "mesh_walk_order": None if not (split and ctx.traversal_mode() == 3) else ("front to back" if walk_order else "reference child order"),
                         "reference_algorithm": reference_algorithm},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(flat, cam, cfg)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
