"""The world and instance levels of the cooperative walks (hiprz_device.hpp: closest_hit_coop / any_hit_coop — a lane steps through the
world tree until it HOLDS a leaf, tests the boxes of the leaf's instances until it ENTERS one) over world trees of many shapes: 1 to 300
instances (one leaf; a tree a few levels deep; a tree with hundreds of leaves), instances without a mesh (they are in no leaf), groups,
lights (the shadow walk has the same levels).  Every packaging that walks this way — split pipeline on the snapshot's trees, on host SAH
trees, on the device's trees, the per-wave resident kernel — renders the frame of the reference-order walk (pipeline 1, walk order 0:
the nested loops the work counters are anchored on), bit for bit; and the frames do not depend on how far the lanes may advance."""
import numpy as np
import pytest

from rayzath_amd import scenes
from rayzath_amd.engine import Context, LightSampling, RenderConfig, Tracing
from rayzath_amd.scene import Group, Instance, camera_struct, flatten

pytestmark = pytest.mark.gpu


def _world(n_instances, lights, seed):
    world = scenes.living_room(160, 96, n_instances, seed=seed)
    if not lights:
        world.spot_lights.clear(), world.direct_lights.clear()
        world.material.emission = 1.0           # the sky lights the room instead
    rng = np.random.default_rng(seed)
    for k in range(0, n_instances, 7):          # some instances have no mesh: they are left out of the world tree (bvh.hpp:40-47)
        world.add(Instance(None, [], position=tuple(rng.uniform(-1, 1, 3)), name=f"empty {k}"))
    if n_instances >= 9:
        inner = world.add(Group(position=(0.1, 0.2, 0.0), rotation=(0.0, 0.3, 0.0), objects=[world.instances[8]], name="inner"))
        world.add(Group(position=(-0.2, 0.0, 0.1), scale=(1.1, 1.0, 0.9), objects=[world.instances[9 % len(world.instances)]], groups=[inner], name="outer"))
    return world


def _frames(flat, cam, cfg, monkeypatch, env=None, **settings):
    for k in ("HIPRZ_WORLD_ADVANCE", "HIPRZ_WALK_ADVANCE", "HIPRZ_SHADOW_PACKET", "HIPRZ_SHADOW_TREE"):
        monkeypatch.delenv(k, raising=False)
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    c = Context(0)
    for k, v in settings.items():
        getattr(c, "set_" + k)(v)
    c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
    for n in (1, 4, 3):
        c.render(n)
    out = (c.read_accum().copy(), c.read_depth().copy(), c.ray_count())
    c.close()
    return out


def _same(a, b, what):
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2], what


@pytest.mark.parametrize("n_instances,lights", [(1, False), (2, True), (8, False), (9, True), (17, False), (33, True), (64, True), (300, False), (300, True)])
def test_walk_levels_over_world_trees_of_many_shapes(built, monkeypatch, n_instances, lights):
    world = _world(n_instances, lights, seed=100 + n_instances)
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(LightSampling(2, 1), Tracing(5, 4)).struct()
    reference = _frames(flat, cam, cfg, monkeypatch, pipeline=1, walk_order=0, lds_scene=0)
    for name, settings in {"front to back, the snapshot's trees": dict(pipeline=1, lds_scene=0),
                           "host SAH trees": dict(tree=1), "device trees, Morton order": dict(tree=2), "device SAH trees": dict(tree=3),
                           "the hosts' default trees": dict(tree=4)}.items():
        _same(reference, _frames(flat, cam, cfg, monkeypatch, **settings), name)
    if lights:       # the shadow rays walked by the wave (rz_shadow_packet_kernel: what big frames get) and lane by lane (the cooperative walk)
        for tree in (0, 3):
            for packet in ("1", "0"):
                _same(reference, _frames(flat, cam, cfg, monkeypatch, env={"HIPRZ_SHADOW_PACKET": packet}, pipeline=1, lds_scene=0, tree=tree), f"shadow walk: packet {packet}, tree {tree}")
            # ... by the wave on the reference's world tree instead of the shadow rays' own (build_shadow_world_tree)
            _same(reference, _frames(flat, cam, cfg, monkeypatch, env={"HIPRZ_SHADOW_PACKET": "1", "HIPRZ_SHADOW_TREE": "0"}, pipeline=1, lds_scene=0, tree=tree), f"wave-level walk on the reference's world tree, tree {tree}")
    if not lights:   # the per-wave resident kernel (scenes without lights)
        _same(reference, _frames(flat, cam, cfg, monkeypatch, pipeline=2, lds_scene=0, tree=3), "per-wave resident kernel")
    for world_advance, walk_advance in (("0", "0"), ("1", "3"), ("64", "64")):
        got = _frames(flat, cam, cfg, monkeypatch, env={"HIPRZ_WORLD_ADVANCE": world_advance, "HIPRZ_WALK_ADVANCE": walk_advance}, tree=3)
        _same(reference, got, f"world advance {world_advance}, instance advance {walk_advance}")


@pytest.mark.parametrize("flags", [1 | 2 | 8 | 16 | 32, 63])
def test_compat_integrator_over_every_kind_of_tree(built, monkeypatch, flags):
    """The CUDA-compat integrator (hiprz_set_mode) walks the same way (rz_trace_coop_compat_kernel, the deferred shadow walk; flag 4 — the
    coloured shadow mask — walks inside the shade kernel): its frames must not depend on the trees either, nor on the walk's levels."""
    world = _world(33, True, seed=7)
    for m in world.materials[:6]:
        m.scattering = max(m.scattering, 0.05)    # every medium scatters a little: the compat draws run
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(LightSampling(2, 1), Tracing(5, 4)).struct()
    reference = _frames(flat, cam, cfg, monkeypatch, mode=flags)
    for tree in (1, 2, 3):
        _same(reference, _frames(flat, cam, cfg, monkeypatch, mode=flags, tree=tree), f"tree {tree}")
    _same(reference, _frames(flat, cam, cfg, monkeypatch, env={"HIPRZ_WORLD_ADVANCE": "0", "HIPRZ_WALK_ADVANCE": "0"}, mode=flags, tree=3), "levels off")
    _same(reference, _frames(flat, cam, cfg, monkeypatch, mode=flags, pipeline=0), "one fused kernel per pass")
    for packet in ("1", "0"):   # the shadow rays walked by the wave (what big frames get; with flag 4 its mask-collecting instantiation) / lane by lane
        _same(reference, _frames(flat, cam, cfg, monkeypatch, env={"HIPRZ_SHADOW_PACKET": packet}, mode=flags, tree=3), f"shadow walk: packet {packet}")


def test_moved_instances_move_the_shadow_rays_world_tree(built, monkeypatch):
    """hiprz_update_instances rebuilds the shadow rays' own world tree (build_shadow_world_tree) with the reference's: a scene with lights whose
    instances moved on the device == a fresh upload of the moved world, with the shadow rays walked by the wave on that tree."""
    monkeypatch.setenv("HIPRZ_SHADOW_PACKET", "1")
    def build(moved):
        world = _world(17, True, seed=31)
        if moved:
            for k, inst in enumerate(world.instances[2:12]):
                inst.position = (np.asarray(inst.position, dtype=np.float32) + np.array([0.3 * ((k % 3) - 1), 0.15, -0.2 * (k % 2)], dtype=np.float32)).astype(np.float32)
        return world
    before, after = build(False), build(True)
    flat0, flat1, cam = flatten(before), flatten(after), camera_struct(before.camera)
    cfg = RenderConfig(LightSampling(2, 1), Tracing(5, 4)).struct()
    out = []
    for moved_on_device in (True, False):
        c = Context(0)
        c.set_tree(3), c.set_pipeline(1), c.set_lds_scene(0)     # TREE_DEVICE_SAH: what hiprz_update_instances needs
        c.upload_scene(flat0 if moved_on_device else flat1), c.upload_camera(cam), c.set_config(cfg)
        if moved_on_device:
            c.render(2)
            c.update_instances(flat1.instances)
        for n in (1, 4, 3):
            c.render(n)
        out.append((c.read_accum().copy(), c.read_depth().copy()))
        c.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    still = Context(0)
    still.set_tree(3), still.set_pipeline(1), still.set_lds_scene(0)
    still.upload_scene(flat0), still.upload_camera(cam), still.set_config(cfg)
    for n in (1, 4, 3):
        still.render(n)
    assert not np.array_equal(still.read_accum(), out[0][0])      # the move is visible
    still.close()
