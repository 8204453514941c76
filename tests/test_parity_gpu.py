"""GPU parity: the HIP pass kernel (through the C-ABI) against the CPU oracle on the same
seeded inputs.  Bars (north_star: "radiance within a stated float tolerance, pixel/sample
indexing bit-exact"):

  * bit-exact: first-pass depth buffer (hit distance), finished-path counts (accumulator
    alpha), per-pixel path depth and material id, ray counters, work counters on pass 1
  * radiance / path state: the only source of difference is libm (CPU) vs ocml (GPU)
    sinf/cosf/acosf/powf — each within a few ulp — so every pixel must agree to
    rel 1e-3 (abs 1e-3 below 1.0) except a stated small fraction of pixels whose path
    crossed a geometric edge because of such an ulp (FRACTION below).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from rayzath_amd import scenes
from rayzath_amd.engine import Context, LightSampling, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten

pytestmark = pytest.mark.gpu

REL, FRACTION = 1e-3, 0.01
# scenes with lights (next-event estimation behind expf / cosf / acosf: a sample at the `radiance < 1e-4` threshold or a shadow ray
# grazing an edge differs): measured on MI355X minus 0.2 points, see the printed reports
LIT_ALPHA, LIT_RGB = 0.997, 0.993


def _run_both(world, max_depth, passes, mode=-1, spot=1, direct=1, seed=20240501):
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(LightSampling(spot, direct), Tracing(max_depth, passes), seed).struct()
    ctx = Context(0)
    ctx.set_traversal_mode(mode)
    if mode >= 3:
        ctx.set_lds_scene(0)  # mode 3 is for scenes that are not staged whole in LDS
    ctx.upload_scene(flat)
    ctx.upload_camera(cam)
    ctx.set_config(cfg)
    ref = oracle.OracleRenderer(flat, cam, cfg)
    return ctx, ref


def _close(a, b):
    return np.abs(a - b) <= REL * np.maximum(np.abs(b), 1.0)


def _compare(ctx, ref, tag):
    acc, racc = ctx.read_accum(), ref.accum
    st, rst = ctx.read_state(), ref.state
    report = {}
    report["alpha_equal"] = float((acc[..., 3] == racc[..., 3]).mean())
    report["depth_equal"] = float((st["depth"] == rst["depth"]).mean())
    report["material_equal"] = float((st["material"] == rst["material"]).mean())
    report["rgb_close"] = float(_close(acc[..., :3], racc[..., :3]).all(-1).mean())
    report["rgb_bitexact"] = float((acc[..., :3] == racc[..., :3]).all(-1).mean())
    report["origin_close"] = float(_close(st["origin"], rst["origin"]).all(-1).mean())
    report["direction_close"] = float(_close(st["direction"], rst["direction"]).all(-1).mean())
    report["mean_abs_rgb_diff"] = float(np.abs(acc[..., :3] - racc[..., :3]).mean())
    print(tag, report)
    return report


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_first_pass_is_bit_exact_where_no_libm_is_involved(built, mode):
    world = scenes.cornell_box(256, 256)
    ctx, ref = _run_both(world, 4, 1, mode)
    ctx.render(1)
    ref.render(1)
    assert np.array_equal(ctx.read_depth(), ref.depth)
    acc, racc = ctx.read_accum(), ref.accum
    assert np.array_equal(acc[..., 3], racc[..., 3])
    # emission picked up on the primary hit involves no transcendental: bit-exact
    assert np.array_equal(acc[..., :3], racc[..., :3])
    st, rst = ctx.read_state(), ref.state
    assert np.array_equal(st["depth"], rst["depth"])
    assert np.array_equal(st["material"], rst["material"])
    assert np.array_equal(st["origin"][rst["depth"] > 0], rst["origin"][rst["depth"] > 0])
    assert ctx.ray_count() == ref.traced_rays == 256 * 256


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_cornell_config_a(built, mode):
    """BASELINE config A: Cornell box 256x256, depth 4, until >= 4 finished samples everywhere."""
    world = scenes.cornell_box(256, 256)
    ctx, ref = _run_both(world, 4, 16, mode)
    ctx.render(16)
    ref.render(16)
    rep = _compare(ctx, ref, f"config A mode {mode}")
    assert ref.accum[..., 3].min() >= 4
    assert rep["alpha_equal"] >= 1 - FRACTION and rep["depth_equal"] >= 1 - FRACTION
    assert rep["rgb_close"] >= 1 - FRACTION
    assert rep["origin_close"] >= 1 - FRACTION and rep["direction_close"] >= 1 - FRACTION
    ctx.tonemap()
    img = ctx.read_rgba8()
    assert (np.abs(img.astype(int) - ref.rgba8.astype(int)).max(-1) <= 1).mean() >= 1 - FRACTION
    assert ctx.ray_count() == ref.traced_rays


def test_counters_match_on_first_pass(built):
    world = scenes.cornell_box(128, 96)
    ctx, ref = _run_both(world, 4, 1)
    got = ctx.render_counted(1)
    want = ref.render(1, counted=True)
    assert got == want


def test_lights_glass_scattering_scene(built):
    """Config-E-like scene at low resolution: NEE with MIS (spot + direct), shadow rays,
    transmission and scattering branches."""
    world = scenes.living_room(160, 96, n_instances=24)
    ctx, ref = _run_both(world, 6, 8, spot=2, direct=1)
    ctx.render(8)
    ref.render(8)
    rep = _compare(ctx, ref, "living room")
    assert rep["alpha_equal"] >= LIT_ALPHA and rep["rgb_close"] >= LIT_RGB


@pytest.mark.parametrize("mode", [1, 3])
def test_sphere_scene_hits(built, mode):
    """Config-C-like: 6 240-triangle sphere with per-vertex normals, deep mesh tree."""
    world = scenes.cornell_sphere(320, 180, 80)
    ctx, ref = _run_both(world, 8, 1, mode)
    got = ctx.render_counted(1)
    want = ref.render(1, counted=True)
    assert np.array_equal(ctx.read_depth(), ref.depth)
    assert got == want
    ctx.render(7)
    ref.render(7)
    rep = _compare(ctx, ref, "sphere")
    assert rep["alpha_equal"] >= 1 - FRACTION and rep["rgb_close"] >= 1 - FRACTION


def test_sharded_render_equals_unsharded(built):
    """Tile sharding must not change any pixel: union of 3 shards == 1 shard, bit for bit."""
    world = scenes.cornell_box(200, 120)
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(tracing=Tracing(4, 4)).struct()
    full = Context(0)
    full.upload_scene(flat), full.upload_camera(cam), full.set_config(cfg)
    full.render(5)
    want = full.read_accum()
    total = np.zeros_like(want)
    rays = 0
    for rank in range(3):
        c = Context(0)
        c.set_shard(rank, 3)
        c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
        c.render(5)
        part = c.read_accum()
        assert not (np.abs(total).sum(-1) > 0)[np.abs(part).sum(-1) > 0].any(), "shards overlap"
        total += part
        rays += c.ray_count()
        c.close()
    assert np.array_equal(total, want)
    assert rays == full.ray_count() == 5 * 200 * 120


@pytest.mark.parametrize("mode", [1, 2])
def test_split_pipeline_equals_fused(built, mode):
    """trace kernel + shade kernel (hit record through HBM) == fused pass kernel, incl. lights and counters."""
    for world, depth in ((scenes.cornell_box(160, 96), 4), (scenes.living_room(96, 64, 16), 5), (scenes.cornell_sphere(128, 72, 32), 6)):
        flat, cam = flatten(world), camera_struct(world.camera)
        cfg = RenderConfig(LightSampling(2, 1), Tracing(depth, 4)).struct()
        out = []
        for pipeline in (0, 1):
            c = Context(0)
            c.set_traversal_mode(mode), c.set_pipeline(pipeline)
            c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
            counters = c.render_counted(1)
            c.render(5)
            out.append((c.read_accum(), c.read_depth(), c.read_state(), counters))
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][3] == out[1][3]
        for k in out[0][2]:
            assert np.array_equal(out[0][2][k], out[1][2][k]), k


@pytest.mark.parametrize("mode", [1, 2])
def test_resident_pipeline_equals_split(built, mode):
    """One launch per batch (state and accumulator in registers across passes, tone map on the way out) == one or two
    launches per pass, bit for bit: accumulator, path state, RGBA8 and counters; also sharded."""
    for world, depth in ((scenes.cornell_box(160, 96), 4), (scenes.living_room(96, 64, 16), 5), (scenes.cornell_sphere(128, 72, 32), 6)):
        flat, cam = flatten(world), camera_struct(world.camera)
        cfg = RenderConfig(LightSampling(2, 1), Tracing(depth, 4)).struct()
        for shard in ((0, 1), (1, 3)):
            out = []
            for pipeline in (1, 2):
                c = Context(0)
                c.set_traversal_mode(mode), c.set_pipeline(pipeline), c.set_shard(*shard)
                c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
                counters = [c.render_counted(3), c.render_counted(2)]
                c.render(5), c.render(1), c.render(4)
                c.tonemap()
                out.append((c.read_accum(), c.read_rgba8(), c.read_state(), counters))
            assert out[0][3] == out[1][3]
            assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
            for k in out[0][2]:
                assert np.array_equal(out[0][2][k], out[1][2][k]), k


def test_wave_resident_pipeline_equals_split(built):
    """Scenes that are not staged in LDS, without lights: ONE launch per batch in which every wave takes its 64 pixels through all the
    passes with the cooperative front-to-back walk (rz_wave_batch_kernel; what a shard of an 8-GPU job runs) == trace + shade kernels per
    pass, bit for bit — accumulator, RGBA8, path state, executed-work counters; also sharded, also chosen automatically for a small shard."""
    for world, depth in ((scenes.cornell_sphere(128, 72, 32), 6), (scenes.textured_sphere_scene(120, 80, resolution=48, map_size=32), 5)):
        flat, cam = flatten(world), camera_struct(world.camera)
        cfg = RenderConfig(LightSampling(1, 1), Tracing(depth, 4)).struct()
        for shard in ((0, 1), (1, 3)):
            out = []
            for pipeline in (1, 2, -1):
                c = Context(0)
                c.set_traversal_mode(3), c.set_lds_scene(0), c.set_pipeline(pipeline), c.set_shard(*shard), c.set_walk_order(2)
                c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
                counters = [c.render_counted(3), c.render_counted(2)]
                c.render(5), c.render(1), c.render(4)
                c.tonemap()
                out.append((c.read_accum(), c.read_rgba8(), c.read_state(), counters, c.pipeline()))
            assert [o[4] for o in out] == [1, 2, 2]                    # a shard this small runs resident by default
            for other in out[1:]:
                assert out[0][3] == other[3]
                assert np.array_equal(out[0][0], other[0]) and np.array_equal(out[0][1], other[1])
                for k in out[0][2]:
                    assert np.array_equal(out[0][2][k], other[2][k]), k


def test_ray_reordering_changes_nothing_but_the_order(built):
    """Sorted walk order (keys from the shade kernel, radix sort, permutation) == pixel order, bit for bit."""
    for world, depth, mode in ((scenes.cornell_sphere(160, 90, 40), 6, 3), (scenes.living_room(96, 64, 16), 5, 3), (scenes.cornell_box(100, 60), 4, 1)):
        flat, cam = flatten(world), camera_struct(world.camera)
        cfg = RenderConfig(LightSampling(1, 1), Tracing(depth, 4)).struct()
        out = []
        for sort in (0, 1):
            c = Context(0)
            c.set_traversal_mode(mode), c.set_lds_scene(0), c.set_ray_sort(sort)
            c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
            counters = c.render_counted(2)
            c.render(4), c.render(4)   # graph capture + replay include the sort
            out.append((c.read_accum(), c.read_state(), counters))
        assert np.array_equal(out[0][0], out[1][0]) and out[0][2] == out[1][2]
        for k in out[0][1]:
            assert np.array_equal(out[0][1][k], out[1][1][k]), k


def test_graph_replay_equals_eager_launches(built):
    world = scenes.cornell_box(160, 96)
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(tracing=Tracing(4, 4)).struct()
    out = []
    for graph in (True, False):
        c = Context(0)
        c.set_graph(graph)
        c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
        c.render(1)
        for _ in range(3):
            c.render(4)       # captured once, replayed twice
        c.render(3)           # different batch size: re-capture
        out.append((c.read_accum(), c.ray_count(), c.pass_count()))
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1:] == out[1][1:] == (16 * 160 * 96, 16)


def test_pick(built):
    world = scenes.cornell_box(128, 128)
    ctx, ref = _run_both(world, 4, 1)
    ctx.render(1)
    ref.render(1)
    for (x, y) in [(64, 64), (10, 64), (120, 64), (64, 5), (40, 90), (90, 100)]:
        assert ctx.pick(x, y) == ref.pick(x, y)


def test_shared_reciprocal_division_is_exact(built):
    """The walk's box test divides with one refined reciprocal per ray axis; the device self-test
    compares it with the correctly rounded `/` over the operand range the kernel allows it in — and the filtered box test of the
    cooperative walks (a verdict from one multiplication per quotient wherever no comparison is closer than 2^-20) with the exact one."""
    ctx = Context(0)
    for seed in (1, 2, 3):
        bad, n = ctx.selftest(256, seed)
        assert 2 * 1024 * 256 * 256 <= n <= 3 * 1024 * 256 * 256     # two quotients per case + (rays in the fast range) one filtered box test
        assert bad == 0, f"{bad} of {n} quotients / box verdicts differ"
    print(ctx.timings())


@pytest.mark.parametrize("name", ["cornell_128", "living_room_96x64", "sphere_160x90", "textured_80x48"])
@pytest.mark.parametrize("mode", [3, 2, 1])
def test_gpu_matches_committed_golden(built, name, mode):
    """Same comparison without the oracle in the loop: committed fixtures (tests/golden)."""
    from test_golden_oracle import load_golden
    g, flat, cam, cfg = load_golden(name)
    ctx = Context(0)
    ctx.set_traversal_mode(mode)
    if mode >= 3:
        ctx.set_lds_scene(0)
    ctx.upload_scene(flat), ctx.upload_camera(cam), ctx.set_config(cfg)
    first = ctx.render_counted(1)
    assert list(first.values()) == g["counters_first"].tolist()
    assert np.array_equal(ctx.read_depth(), g["depth"])
    ctx.render(int(g["passes"]) - 1)
    acc = ctx.read_accum()
    lights = len(flat.spot_lights) + len(flat.direct_lights) > 0
    frac = 0.02 if lights else FRACTION
    alpha_eq, rgb_ok = (acc[..., 3] == g["accum"][..., 3]).mean(), _close(acc[..., :3], g["accum"][..., :3]).all(-1).mean()
    print(f"golden {name} mode {mode}: alpha equal {alpha_eq:.5f}, rgb within 1e-3 {rgb_ok:.5f}")
    assert alpha_eq >= (LIT_ALPHA if lights else 1 - FRACTION)
    assert rgb_ok >= (LIT_RGB if lights else 1 - FRACTION)
    st = ctx.read_state()
    assert (st["depth"] == g["path_depth"]).mean() >= 1 - frac and (st["material"] == g["ray_material"]).mean() >= 1 - frac
    ctx.tonemap()
    assert (np.abs(ctx.read_rgba8().astype(int) - g["rgba8"].astype(int)).max(-1) <= 1).mean() >= 1 - frac


@pytest.mark.parametrize("lds", [0, 1])
def test_lds_staged_scene_is_only_a_placement_choice(built, lds):
    world = scenes.cornell_box(160, 96)
    ctx, ref = _run_both(world, 4, 6, mode=1)
    ctx.set_lds_scene(lds)
    ctx.render(6)
    ref.render(6)
    assert np.array_equal(ctx.read_accum(), ref.accum)


def test_device_tile_layout_matches_host_layout(built):
    from rayzath_amd.distributed import tile_pixel_coords
    world = scenes.cornell_box(100, 44)
    flat, cam = flatten(world), camera_struct(world.camera)
    ctx = Context(0)
    ctx.set_shard(1, 3)
    ctx.upload_scene(flat), ctx.upload_camera(cam), ctx.set_config(RenderConfig(tracing=Tracing(4, 1)).struct())
    ctx.render(1)
    owned = ctx.read_depth() != 0
    x, y = tile_pixel_coords(100, 44, 1, 3)
    want = np.zeros((44, 100), bool)
    want[y[x >= 0], x[x >= 0]] = True
    assert np.array_equal(owned, want)


def test_engine_interface_and_errors(built):
    """Engine.renderWorld mirrors the reference backend contract; misuse raises HiprzError."""
    from rayzath_amd import HiprzError
    from rayzath_amd.engine import Engine
    eng = Engine(0)
    world = scenes.cornell_box(96, 64)
    cfg = RenderConfig(tracing=Tracing(4, 3))
    eng.renderWorld(world, cfg)
    assert world.camera.image_buffer.shape == (64, 96, 4) and world.camera.ray_count == 3 * 96 * 64
    eng.renderWorld(world, cfg)
    assert world.camera.ray_count == 6 * 96 * 64                 # accumulation continues
    world.camera.position[0] += 0.25
    eng.renderWorld(world, cfg)
    assert world.camera.ray_count == 3 * 96 * 64                 # camera moved: restart (cpu_engine_renderer.cpp:108-112)
    assert "render" in eng.timingsString()
    ctx = Context(0)
    with pytest.raises(HiprzError) as e:
        ctx.render(1)
    assert e.value.code == 3
    with pytest.raises(HiprzError):
        ctx.set_config(RenderConfig(LightSampling(0, 1)).struct())   # 0 samples would be 0/0 in the reference
    with pytest.raises(HiprzError):
        ctx.set_shard(3, 3)
    flat = flatten(world)
    flat.nodes["begin"][0] = 12345
    with pytest.raises(HiprzError) as e:
        ctx.upload_scene(flat)
    assert e.value.code == 1 and "leaf range" in str(e.value)


def test_sharded_bench_path_rehearsal(built):
    """bench.py's N > 1 path (shard, render, gather to rank 0, untile, tone-map) with 2 ranks sharing GPU 0
    over gloo; the assembled frame must equal the single-rank frame."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29671", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "A",
           "--rehearse-on-one-gpu", "--no-cpu-baseline", "--verify-gather", "--shard-mode", "tiles", "--no-other-mode"]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = [l for l in proc.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["value"] > 0 and res["scaling"] == "strong"


@pytest.mark.parametrize("overlap", [1, 0])
def test_sharded_frame_streams_with_in_process_collective(built, overlap):
    """ShardedFrame's stream choreography (export on the render stream, gather + one-launch untile on the comm stream,
    next export waiting for the previous gather) with both shards of a 2-GPU job living in one process: the collective is
    a stand-in that copies the shards' tile buffers on whatever stream torch has current, as RCCL would.  Runs in a child
    process because torch has to initialise the GPU before the library does."""
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "inprocess_gather_check.py")
    r = subprocess.run([sys.executable, script, str(overlap)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "frames equal" in r.stdout


def test_scene_file_renders_like_the_model_it_was_written_from(built, tmp_path):
    """.json scene -> C++ loader -> upload -> render == the Python model rendered directly, and == the oracle."""
    from rayzath_amd import scene_io
    world = scenes.living_room(96, 64, 8)
    path = str(tmp_path / "room.json")
    scene_io.save_scene_json(world, path)
    loaded = scene_io.load_scene_file(path)
    assert loaded.errors == 0, loaded.log
    cfg = RenderConfig(LightSampling(1, 1), Tracing(5, 4)).struct()
    out = []
    for flat, cam in ((flatten(world), camera_struct(world.camera)), (loaded.flat, loaded.camera)):
        c = Context(0)
        c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
        c.render(5)
        out.append((c.read_accum(), c.read_depth()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    # ... and the oracle on the snapshot the C++ loader produced
    ref = oracle.OracleRenderer(loaded.flat, loaded.camera, cfg)
    ref.render(5)
    assert np.array_equal(out[1][1], ref.depth)
    assert (out[1][0][..., 3] == ref.accum[..., 3]).mean() >= LIT_ALPHA
    assert _close(out[1][0][..., :3], ref.accum[..., :3]).all(-1).mean() >= LIT_RGB


def test_textured_scene_files_render_like_the_model(built, tmp_path):
    """Config D's kind of scene (displaced sphere with texture, normal map and roughness map) written to .json + PNG maps, read
    back by the C++ loader (own PNG decoder) and rendered: the same frame as the Python model it was written from, bit for bit."""
    from rayzath_amd import scene_io
    world = scenes.textured_sphere_scene(128, 80, resolution=60, map_size=64)
    path = str(tmp_path / "sphere.json")
    scene_io.save_scene_json(world, path)
    assert sorted(os.listdir(tmp_path / "maps")) == ["normal", "roughness", "texture"]
    loaded = scene_io.load_scene_file(path)
    assert loaded.errors == 0, loaded.log
    cfg = RenderConfig(LightSampling(1, 1), Tracing(6, 4)).struct()
    out = []
    for flat, cam in ((flatten(world), camera_struct(world.camera)), (loaded.flat, loaded.camera)):
        c = Context(0)
        c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
        counters = c.render_counted(1)
        c.render(5)
        out.append((c.read_accum(), c.read_depth(), counters))
    assert out[0][2] == out[1][2] and out[0][2]["texel_fetches"] > 0
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    # ... and the oracle on the snapshot the C++ loader produced (maps decoded by the host library's PNG reader)
    ref = oracle.OracleRenderer(loaded.flat, loaded.camera, cfg)
    assert ref.render(1, counted=True) == out[1][2]
    ref.render(5)
    assert np.array_equal(out[1][1], ref.depth)
    assert (out[1][0][..., 3] == ref.accum[..., 3]).mean() >= 1 - FRACTION
    assert _close(out[1][0][..., :3], ref.accum[..., :3]).all(-1).mean() >= 1 - FRACTION


def test_headless_runner_end_to_end(built, tmp_path):
    """hiprz_headless --headless tasks.json: loads the scene file, renders through Hip::Engine::renderWorld with the reference's
    adaptive passes-per-call loop, writes report.txt in the reference's format and (with -r) the frame."""
    import re
    from rayzath_amd import scene_io
    world = scenes.cornell_box(128, 96)
    scene_io.save_scene_json(world, str(tmp_path / "cornell.json"))
    (tmp_path / "tasks.json").write_text('{"tasks": [{"scene path": "cornell.json", "engine": ["HIPGPU"], "rpp": 40, "timeout": 20.0, "max depth": 4}]}')
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rayzath_amd", "csrc", "hiprz_headless")
    out_dir = tmp_path / "reports" / "run1"        # does not exist yet: the runner creates it
    r = subprocess.run([exe, "--headless", str(tmp_path / "tasks.json"), str(out_dir), "-r", "--quiet"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    tmp_path = out_dir
    report = (tmp_path / "report.txt").read_text()
    m = re.fullmatch(r"Scene: cornell\.json\n\tengine: HIPGPU \| max depth: 4\n\tduration: \d+\.\d{3}s \| traced (\S+) rays \((\S+) rps\)\n", report)
    assert m, report
    assert m.group(1) == "503.8K"   # 41 * 128 * 96 rays: 40 pipelined passes + the final synchronous one
    images = [f for f in os.listdir(tmp_path) if f.endswith("_HIPGPU.png")]
    assert len(images) == 1
    frame = scene_io.read_image(str(tmp_path / images[0]))
    assert frame.shape == (96, 128, 4) and (frame[..., 3] == 255).all()
    pixels = frame[..., :3]
    assert pixels.max() > 100 and pixels.std() > 10   # an actual picture, not a blank frame


def test_deferred_shadow_rays_equal_inline_walks(built, monkeypatch):
    """rz_shade_kernel<..., DEFER> + rz_shadow_kernel (shadow rays of a pass walked in their own lean kernel, sums finished in
    the reference's order) == shadow rays walked inside the shade kernel: accumulator, state and all ten work counters."""
    for samples in ((1, 1), (3, 2)):
        world = scenes.living_room(96, 64, 16)
        flat, cam = flatten(world), camera_struct(world.camera)
        cfg = RenderConfig(LightSampling(*samples), Tracing(5, 4)).struct()
        out = []
        for defer in ("0", "1"):
            monkeypatch.setenv("HIPRZ_DEFER_SHADOWS", defer)
            c = Context(0)
            c.set_traversal_mode(3), c.set_lds_scene(0), c.set_ray_sort(0)
            c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
            counters = [c.render_counted(1), c.render_counted(2)]
            c.render(4), c.render(4)
            out.append((c.read_accum(), c.read_depth(), c.read_state(), counters))
        assert out[0][3] == out[1][3]
        assert out[0][3][0]["shadow_rays"] > 0
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
        for k in out[0][2]:
            assert np.array_equal(out[0][2][k], out[1][2][k]), k


def test_front_to_back_walk_equals_reference_order(built):
    """hiprz_set_walk_order: meshes walked front to back on per-octant skip links (trace kernel and deferred shadow kernel) give
    the accumulator, depth buffer and path state of the walk in the reference's child order, bit for bit, while testing fewer
    boxes and triangles; counted renders keep the reference's order (and so the CPU kernel's counters) unless order 2 is set."""
    cases = ((scenes.cornell_sphere(160, 96, resolution=40), (1, 1)),
             (scenes.textured_sphere_scene(160, 96, resolution=60, map_size=64), (1, 1)),
             (scenes.living_room(128, 80, 16), (2, 2)))
    for world, samples in cases:
        flat, cam = flatten(world), camera_struct(world.camera)
        cfg = RenderConfig(LightSampling(*samples), Tracing(6, 4)).struct()
        out = []
        for order in (0, 1, 2):
            c = Context(0)
            c.set_traversal_mode(3), c.set_lds_scene(0), c.set_walk_order(order)
            c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
            counters = c.render_counted(2)
            c.render(6), c.render(4)
            out.append((c.read_accum(), c.read_depth(), c.read_state(), counters))
        assert out[0][3] == out[1][3]                      # order 1 counts in the reference's order
        for k in ("segments", "hits", "finished", "light_samples", "shadow_rays", "texel_fetches"):
            assert out[2][3][k] == out[0][3][k], k          # same paths, same hits ...
        assert out[2][3]["box_tests"] < out[0][3]["box_tests"]  # ... found with fewer tests
        assert out[2][3]["tri_tests"] < out[0][3]["tri_tests"]
        for other in (1, 2):
            assert np.array_equal(out[0][0], out[other][0]) and np.array_equal(out[0][1], out[other][1])
            for k in out[0][2]:
                assert np.array_equal(out[0][2][k], out[other][2][k]), k


def test_scene_specialised_shading_equals_the_general_code(built, monkeypatch):
    """Scenes without lights (and without maps) run instantiations whose next-event estimation (texture fetches, normal mapping) is
    compiled out — RZ_SHADOW_NONE / RZ_SHADOW_PLAIN.  Frames, path state and counters equal the general instantiation's bit for bit,
    in the resident, split and fused pipelines."""
    cases = ((scenes.cornell_box(96, 64), None), (scenes.cornell_sphere(96, 64, resolution=24), 3),
             (scenes.textured_sphere_scene(96, 64, resolution=24, map_size=32), 3))       # plain LDS scene, plain global scene, no lights but maps
    for world, mode in cases:
        flat, cam = flatten(world), camera_struct(world.camera)
        cfg = RenderConfig(LightSampling(1, 1), Tracing(6, 4)).struct()
        for pipeline in (0, 1, 2):
            out = []
            for special in ("0", "1"):
                monkeypatch.setenv("HIPRZ_NOLIGHT_KERNELS", special)
                c = Context(0)
                if mode is not None:
                    c.set_traversal_mode(mode), c.set_lds_scene(0)
                c.set_pipeline(pipeline)
                c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
                counters = c.render_counted(2)
                c.render(6), c.render(4)
                c.tonemap()
                out.append((c.read_accum(), c.read_depth(), c.read_state(), counters, c.read_rgba8(), c.pipeline()))
            assert out[0][3] == out[1][3] and out[0][5] == out[1][5]
            assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][4], out[1][4])
            for k in out[0][2]:
                assert np.array_equal(out[0][2][k], out[1][2][k]), k


def test_exact_ties_pick_the_triangle_the_reference_meets_first(built, monkeypatch):
    """A mesh whose every triangle exists twice (the copy with ANOTHER material, shuffled in among the originals) and an instance that
    exists twice in the same place: every hit is an exact tie, inside a leaf, across leaves and across instances.  The reference keeps
    the triangle it meets first (`t >= far` rejects the later one), so the material a pixel sees tells which one won.  The front-to-back
    cooperative walk must agree with the reference-order walk — and all of them with the CPU oracle — on every pixel."""
    from rayzath_amd.scene import Instance, Material, Mesh, generate_sphere
    rng = np.random.default_rng(11)
    base = generate_sphere(20, normals=False, texture_coordinates=False)
    T = len(base.tri_vertices)
    order = rng.permutation(2 * T)
    tri_vertices = np.concatenate([base.tri_vertices, base.tri_vertices])[order]
    tri_materials = np.concatenate([np.zeros(T), np.ones(T)]).astype(np.uint32)[order]
    twin = Mesh(base.vertices, tri_vertices, tri_materials=tri_materials)
    world = scenes.cornell_box(160, 100)
    red, blue = world.add(Material((220, 40, 40, 255), 0.0, 1.0)), world.add(Material((40, 40, 220, 255), 0.3, 0.2))
    twin = world.add(twin)
    for _ in range(2):   # the same instance twice: ties across instances (the world tree keeps the reference's order in every variant)
        world.add(Instance(twin, [red, blue], position=(0.2, 0.6, 0.1), rotation=(0.3, 0.5, 0.1), scale=(1.3, 1.3, 1.3)))
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(LightSampling(1, 1), Tracing(6, 4)).struct()
    ref = oracle.OracleRenderer(flat, cam, cfg)
    ref.render(6)
    out = []
    for order_ in (0, 1, 2):
        c = Context(0)
        c.set_traversal_mode(3), c.set_lds_scene(0), c.set_walk_order(order_)
        c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
        c.render(6)
        out.append((c.read_accum(), c.read_depth(), c.read_state()))
    for other in out[1:]:
        assert np.array_equal(out[0][0], other[0]) and np.array_equal(out[0][1], other[1])
        for k in out[0][2]:
            assert np.array_equal(out[0][2][k], other[2][k]), k
    assert np.array_equal(out[0][1], ref.depth)
    assert np.array_equal(out[0][2]["material"], ref.state["material"]) and np.array_equal(out[0][0][..., 3], ref.accum[..., 3])
    assert _close(out[0][0][..., :3], ref.accum[..., :3]).all(-1).mean() >= 1 - FRACTION
    # both copies win somewhere (the shuffle decides which comes first in a leaf): red-diffuse and blue-glossy pixels are in the picture
    rgb = ref.accum[..., :3]
    assert (rgb[..., 0] > 3 * rgb[..., 2]).sum() > 30 and (rgb[..., 2] > 1.5 * rgb[..., 0]).sum() > 30


def test_engine_frames_replay_the_captured_graph(built):
    """Engine.renderWorld calls hiprz_set_config before every frame; with an unchanged config the second steady-state frame must
    replay the graph the first one captured (split pipeline) — and a changed config must re-capture."""
    from rayzath_amd.engine import Engine
    world = scenes.cornell_sphere(160, 96, resolution=24)     # not staged in LDS by default? force the split pipeline below
    eng = Engine(0)
    eng.context.set_pipeline(1)
    eng.context.set_ray_sort(0)                    # batches that reorder rays are launched eagerly (the library sort is kept out of graphs)
    cfg = RenderConfig(LightSampling(1, 1), Tracing(6, 4))
    eng.renderWorld(world, cfg)                   # first pass + 3 cumulative: eager
    assert eng.context.graph_captures() == 0
    eng.renderWorld(world, cfg)
    assert eng.context.graph_captures() == 1
    for _ in range(3):
        eng.renderWorld(world, cfg)
    assert eng.context.graph_captures() == 1      # replayed, not re-captured
    eng.renderWorld(world, RenderConfig(LightSampling(1, 1), Tracing(5, 4)))
    assert eng.context.graph_captures() == 2      # max depth changed: kernel arguments differ
    # the frames themselves: the same passes through a fresh context, eagerly
    flat, cam = flatten(world), camera_struct(world.camera)
    ref = Context(0)
    ref.set_pipeline(1), ref.set_graph(False), ref.set_ray_sort(0)
    ref.upload_scene(flat), ref.upload_camera(cam)
    ref.set_config(cfg.struct())
    for _ in range(5):
        ref.render(4)
    ref.set_config(RenderConfig(LightSampling(1, 1), Tracing(5, 4)).struct())
    ref.render(4)
    assert np.array_equal(eng.context.read_accum(), ref.read_accum())
