"""Boundary features of SURVEY.md §8(b) beyond one camera on one GPU:

  * one context over several devices (hiprz_create_multi; here: several shards on GPU 0) == the single-device frame, bit for bit
  * every camera of the world with its own accumulation state (hiprz_select_camera) == one context per camera
  * materials / lights replaced in place (hiprz_update_shading) == a full re-upload
  * group transformations: the CPU engine's behaviour (default) and the CUDA engine's (composed transformation)
"""
import numpy as np
import pytest

import oracle
from rayzath_amd import scenes
from rayzath_amd.engine import Context, Engine, LightSampling, RenderConfig, Tracing
from rayzath_amd.scene import Camera, Group, Instance, Material, camera_struct, flatten, generate_cube

pytestmark = pytest.mark.gpu


def _all(ctx):
    ctx.tonemap()
    return ctx.read_accum(), ctx.read_depth(), ctx.read_rgba8(), ctx.read_state()


def _same(a, b):
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    for k in a[3]:
        assert np.array_equal(a[3][k], b[3][k]), k


@pytest.mark.parametrize("scene", ["cornell", "living room"])
@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0, 0, 0]])
def test_multi_device_context_equals_single_device(built, scene, devices):
    world = scenes.cornell_box(200, 120) if scene == "cornell" else scenes.living_room(160, 96, 16)
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(LightSampling(2, 1), Tracing(5, 4)).struct()
    one, many = Context(0), Context(devices)
    assert many.device_count() == len(devices)
    counted = []
    for c in (one, many):
        c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
        counted.append(c.render_counted(1))
        c.render(4), c.render(4)
    _same(_all(one), _all(many))
    assert counted[0] == counted[1]                                # the shards' work counters add up to the frame's
    assert one.ray_count() == many.ray_count() == 9 * 200 * 120 if scene == "cornell" else True
    for xy in [(10, 10), (100, 60), (150, 100), (60, 90)]:
        assert one.pick(*xy) == many.pick(*xy)
    # a further split of the multi-device context's share, as an 8-GPU job of 2 processes x 4 devices would do
    halves = []
    for rank in range(2):
        c = Context(devices)
        c.set_shard(rank, 2)
        c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
        c.render(1), c.render(4), c.render(4)
        halves.append(c.read_accum())
    assert np.array_equal(halves[0] + halves[1], one.read_accum())  # disjoint tiles, zero elsewhere
    assert (halves[0][..., 3] > 0).sum() + (halves[1][..., 3] > 0).sum() == (one.read_accum()[..., 3] > 0).sum()


def test_multi_device_context_over_two_physical_gpus(built):
    """hiprz_create_multi with two DISTINCT devices (peer access, cross-device event waits, hipMemcpyPeerAsync gather): the frame of one
    device, bit for bit.  Needs a box with two GPUs; the one-GPU boxes of this pool skip it (the [0, 0] variants above run the same code
    with both shards on one device)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    world = scenes.living_room(160, 96, 16)
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(LightSampling(2, 1), Tracing(5, 4)).struct()
    one, two = Context(0), Context([0, 1])
    for c in (one, two):
        c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
        c.render(1), c.render(4), c.render(4)
    _same(_all(one), _all(two))
    assert one.ray_count() == two.ray_count()
    for xy in [(10, 10), (100, 60), (150, 90)]:
        assert one.pick(*xy) == two.pick(*xy)


def test_tile_export_of_a_context_with_several_streams(built):
    """hiprz_export_*_tiles of a context over several streams: n slices (sub-shard rank * n + r of world * n in slice r) that
    hiprz_untile_accum / hiprz_untile_gathered assemble into the frame one single-stream context renders — alone and as the two ranks of a
    job.  (Device buffers straight from the HIP runtime: the test process keeps torch off the GPU.)"""
    import ctypes as C
    hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")                     # the runtime libhiprz.so itself is linked against (one HIP runtime per process)
    hip.hipMalloc.argtypes, hip.hipMemset.argtypes = [C.POINTER(C.c_void_p), C.c_size_t], [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipMemcpy.argtypes, hip.hipFree.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int], [C.c_void_p]
    world = scenes.cornell_sphere(200, 120, 24)
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(tracing=Tracing(5, 4)).struct()
    one = Context(0)
    one.upload_scene(flat), one.upload_camera(cam), one.set_config(cfg)
    one.render(1), one.render(4)
    want = one.read_accum()
    total = np.zeros_like(want)
    for rank, n_ranks, streams in ((0, 1, 3), (0, 2, 2), (1, 2, 2)):
        many = Context([0] * streams)
        many.set_shard(rank, n_ranks)
        many.upload_scene(flat), many.upload_camera(cam), many.set_config(cfg)
        many.render(1), many.render(4)
        capacity = many.local_pixel_capacity()
        part_capacity = capacity // streams                           # every slice has the capacity of the job's largest sub-shard
        buf, img = C.c_void_p(), C.c_void_p()
        assert hip.hipMalloc(C.byref(buf), capacity * 16) == 0 and hip.hipMalloc(C.byref(img), want.nbytes) == 0
        hip.hipMemset(img, 0, want.nbytes)
        hip.hipDeviceSynchronize()   # (the memset is on the null stream; the context's streams are non-blocking and do not wait for it)
        many.export_accum_tiles(buf.value, capacity * 16)
        for k in range(streams):
            many.untile_accum(buf.value + k * part_capacity * 16, rank * streams + k, n_ranks * streams, img.value)
        many.sync()
        got = np.zeros_like(want)
        assert hip.hipMemcpy(got.ctypes.data, img, want.nbytes, 2) == 0   # hipMemcpyDeviceToHost
        hip.hipFree(buf), hip.hipFree(img)
        if n_ranks == 1:
            assert np.array_equal(got, want)
        else:
            assert not ((np.abs(total).sum(-1) > 0) & (np.abs(got).sum(-1) > 0)).any()
            total += got
        many.close()
    assert np.array_equal(total, want)


def test_every_camera_has_its_own_frame(built):
    world = scenes.cornell_sphere(160, 96, resolution=24)
    flat = flatten(world)
    cams = [world.camera, Camera(position=(1.0, 1.5, -3.0), rotation=(0.1, -0.2, 0.0), resolution=(96, 128), fov=1.2, focal_distance=4.0),
            Camera(position=(-1.0, 0.5, -3.2), rotation=(0.0, 0.25, 0.0), resolution=(64, 40), fov=1.0, focal_distance=4.0)]
    cfg = RenderConfig(tracing=Tracing(5, 4)).struct()
    multi = Context(0)
    multi.set_pipeline(1)
    multi.upload_scene(flat), multi.set_config(cfg)
    multi.set_camera_count(3)
    assert multi.camera_count() == 3
    for k, cam in enumerate(cams):
        multi.select_camera(k)
        multi.upload_camera(camera_struct(cam))
    for n in (1, 4, 4, 3):                      # interleaved: the cameras advance independently
        for k in (2, 0, 1):
            multi.select_camera(k)
            multi.render(n)
    multi.select_camera(1)
    multi.render(4)                              # camera 1 is four passes ahead
    for k, cam in enumerate(cams):
        single = Context(0)
        single.set_pipeline(1)
        single.upload_scene(flat), single.upload_camera(camera_struct(cam)), single.set_config(cfg)
        for n in (1, 4, 4, 3) + ((4,) if k == 1 else ()):
            single.render(n)
        multi.select_camera(k)
        assert multi.pass_count() == single.pass_count() == (16 if k == 1 else 12)
        assert multi.ray_count() == single.ray_count()
        _same(_all(multi), _all(single))
    # the Engine renders every enabled camera of a world per call and skips disabled ones
    world.cameras = cams[1:]
    cams[2].enabled = False
    eng = Engine(0)
    eng.renderWorld(world, RenderConfig(tracing=Tracing(5, 4)))
    assert world.camera.image_buffer.shape[:2] == (96, 160) and cams[1].image_buffer.shape[:2] == (128, 96)
    assert not hasattr(cams[2], "image_buffer") and cams[1].ray_count == 4 * 96 * 128


def test_engine_ray_casts_every_camera_after_each_frame(built):
    """Kernel::rayCast after every frame (cpu_engine_renderer.cpp:176): the camera's ray-cast pixel -> the instance and the instance's
    material it looks at (camera.hpp:55-56), against the oracle's rzo_pick; moving the pixel does not restart accumulation."""
    world = scenes.cornell_box(160, 96)
    second = Camera(position=(0.6, 1.2, -3.2), rotation=(0.05, -0.1, 0.0), resolution=(96, 64), fov=1.3, focal_distance=4.0)
    world.cameras = [second]
    cfg = RenderConfig(tracing=Tracing(5, 3))
    flat = flatten(world)
    eng = Engine(0, streams=1)
    seen = set()
    rays = 0
    for px, py in [(80, 48), (20, 80), (150, 10), (40, 40), (110, 60), (5000, 5000)]:
        world.camera.ray_cast_at(px, py)
        second.ray_cast_at(px // 2, py // 2)
        eng.renderWorld(world, cfg)
        rays += 3 * 160 * 96
        assert world.camera.ray_count == rays                      # still the same accumulation
        for cam in (world.camera, second):
            x, y = cam.ray_cast_pixel
            ref = oracle.OracleRenderer(flat, camera_struct(cam), cfg.struct())
            ref.render(1, threads=1)
            inst, mat = ref.pick(x, y)
            got_inst = world.instances.index(cam.raycasted_instance) if cam.raycasted_instance is not None else -1
            assert got_inst == inst, (x, y)
            got_mat = 2 + world.materials.index(cam.raycasted_material) if cam.raycasted_material is not None else -1  # [0] world, [1] default
            assert got_mat == mat, (x, y)
            seen.add(got_inst)
    assert len(seen) >= 4                                           # walls, boxes, lamp: the pixels really look at different things


def test_update_shading_equals_a_full_upload(built):
    world = scenes.living_room(128, 80, 12)
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(LightSampling(1, 1), Tracing(5, 4)).struct()
    a = Context(0)
    a.upload_scene(flat), a.upload_camera(cam), a.set_config(cfg)
    a.render(5)
    world.materials[1].color = (30, 200, 90, 255)            # the red wall turns green, the mirror rough, a light goes away
    world.materials[4].roughness = 0.4
    world.spot_lights.pop()
    changed = flatten(world)
    assert np.array_equal(changed.nodes, flat.nodes) and np.array_equal(changed.tris, flat.tris)
    a.update_shading(changed)
    a.render(5)
    b = Context(0)
    b.upload_scene(changed), b.upload_camera(cam), b.set_config(cfg)
    b.render(5)
    _same(_all(a), _all(b))
    assert a.pass_count() == 5                               # the change restarted accumulation
    with pytest.raises(Exception):
        a.update_shading(flatten(scenes.cornell_box(64, 64)))  # another material count: refused


def test_update_shading_with_repointed_maps_falls_back_to_a_full_upload(built):
    """Two materials swap their uploaded textures: the first-use numbering of the maps changes, so the indices of the in-place path
    would name the wrong texels.  Context.update_shading notices (identity list of the uploaded maps) and uploads the scene."""
    world = scenes.shading_inputs_scene(96, 64)
    textured = [m for m in world.materials if m.texture is not None]
    assert len(textured) >= 2
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(LightSampling(1, 1), Tracing(5, 4)).struct()
    a = Context(0)
    a.upload_scene(flat), a.upload_camera(cam), a.set_config(cfg)
    a.render(5)
    textured[0].texture, textured[1].texture = textured[1].texture, textured[0].texture
    changed = flatten(world)
    assert changed.map_ids != flat.map_ids and sorted(changed.map_ids) == sorted(flat.map_ids)
    a.update_shading(changed)
    a.render(5)
    b = Context(0)
    b.upload_scene(changed), b.upload_camera(cam), b.set_config(cfg)
    b.render(5)
    _same(_all(a), _all(b))
    ref = oracle.OracleRenderer(changed, cam, cfg)
    ref.render(5, threads=2)
    assert np.array_equal(a.read_depth(), ref.depth) and np.array_equal(a.read_accum()[..., 3], ref.accum[..., 3])


def _grouped_world(mode):
    world = scenes.cornell_box(160, 100)
    blue = world.add(Material((40, 40, 220, 255), 0.2, 0.4))
    cube = world.add(generate_cube())
    a = world.add(Instance(cube, [blue], position=(0.3, 0.0, 0.0), rotation=(0.0, 0.4, 0.0), scale=(0.5, 0.7, 0.5)))
    b = world.add(Instance(cube, [blue], position=(-0.4, 0.2, 0.3), rotation=(0.2, 0.0, 0.1), scale=(0.4, 0.4, 0.4)))
    inner = world.add(Group(position=(0.2, 0.6, 0.0), rotation=(0.0, 0.0, 0.3), scale=(1.0, 1.2, 1.0), objects=[a]))
    world.add(Group(position=(0.0, 0.3, 0.4), rotation=(0.0, 0.5, 0.0), scale=(1.1, 1.0, 0.9), objects=[b], groups=[inner]))
    world.group_transforms = mode
    return world, (a, b)


def test_group_transformations(built):
    cfg = RenderConfig(tracing=Tracing(5, 4)).struct()
    frames = {}
    for mode in ("cpu", "cuda"):
        world, _ = _grouped_world(mode)
        flat, cam = flatten(world), camera_struct(world.camera)
        ctx = Context(0)
        ctx.upload_scene(flat), ctx.upload_camera(cam), ctx.set_config(cfg)
        ctx.render(5)
        ref = oracle.OracleRenderer(flat, cam, cfg)             # the oracle follows the flattened records in either mode
        ref.render(5)
        frames[mode] = ctx.read_accum()
        assert np.array_equal(ctx.read_depth(), ref.depth) and np.array_equal(frames[mode][..., 3], ref.accum[..., 3])
        assert (np.abs(frames[mode][..., :3] - ref.accum[..., :3]) <= 1e-3 * np.maximum(np.abs(ref.accum[..., :3]), 1.0)).all(-1).mean() >= 0.998
    assert not np.array_equal(frames["cpu"], frames["cuda"])
    # "cuda": exactly the scene whose instances carry the composed transformation themselves, outside any group
    world, (a, b) = _grouped_world("cuda")
    flat = flatten(world)
    loose, _ = _grouped_world("cuda")
    recs = flat.instances[-2:]
    for inst in loose.instances[-2:]:
        inst.group = None
    plain = flatten(loose)
    plain.instances[-2:] = recs                                # same composed position / axes / scale / box
    ctx = Context(0)
    ctx.upload_scene(plain), ctx.upload_camera(camera_struct(world.camera)), ctx.set_config(cfg)
    ctx.render(5)
    assert np.array_equal(ctx.read_accum(), frames["cuda"])


def test_engine_splits_a_world_without_lights_over_two_streams(built):
    """The hosts' default on one GPU (engine.default_streams / Hip::Engine::defaultStreams): two contexts-with-a-stream for a world
    without lights, one for a world with lights — and the frames do not depend on it."""
    from rayzath_amd.engine import Engine, default_streams
    assert default_streams(0) == 2 and default_streams(3) == 1
    cfg = RenderConfig(LightSampling(1, 1), Tracing(5, 4))
    frames = []
    for streams in (None, 1, 3):
        world = scenes.cornell_sphere(160, 96, 12)
        engine = Engine(0, streams=streams)
        engine.renderWorld(world, cfg), engine.renderWorld(world, cfg)
        assert engine.context.device_count() == (streams or 2)
        frames.append((world.camera.image_buffer.copy(), world.camera.depth_buffer.copy(), world.camera.ray_count))
    for f in frames[1:]:
        assert np.array_equal(f[0], frames[0][0]) and np.array_equal(f[1], frames[0][1]) and f[2] == frames[0][2]
    lit = scenes.living_room(96, 64, 8)
    engine = Engine(0)
    engine.renderWorld(lit, cfg)
    assert engine.context.device_count() == 1
