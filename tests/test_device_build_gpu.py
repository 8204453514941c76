"""Trees built, refitted and rebuilt ON THE DEVICE (hiprz_set_tree(HIPRZ_TREE_DEVICE / HIPRZ_TREE_DEVICE_SAH), rayzath_amd/csrc/hiprz_build.hip; SURVEY.md §8 f4 —
the reference rebuilds on the host at every change: bvh_tree_node.hpp:117-215, component_container.hpp:259-363):

  * frames with device-built trees == frames with the reference trees == frames with the host SAH trees, bit for bit, on a deep textured
    mesh, an instanced scene with lights, and a scene made of exact ties;
  * the downloaded device output passes the host's scene validation (every index in range, trees disjoint, every walk terminates);
  * the device-built WORLD tree is the host builder's, node for node (the order of the instances is part of a ray's arithmetic);
  * hiprz_update_triangles (refit) and hiprz_update_instances (world-tree rebuild) == a fresh upload of the changed scene.
"""
import ctypes as C

import numpy as np
import pytest

from rayzath_amd import _abi, _lib, scenes
from rayzath_amd.engine import Context, LightSampling, RenderConfig, Tracing
from rayzath_amd.scene import FlatScene, Instance, Material, camera_struct, flatten
from test_trees_gpu import _ties_world

pytestmark = pytest.mark.gpu
DEVICE, DEVICE_SAH = 2, 3   # HIPRZ_TREE_DEVICE (Morton order), HIPRZ_TREE_DEVICE_SAH (binned surface-area build)


def _render(flat, cam, cfg, tree, passes=(1, 5, 4)):
    c = Context(0)
    c.set_tree(tree)
    c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
    for n in passes:
        c.render(n)
    return c


def _same_frames(a, b):
    assert np.array_equal(a.read_accum(), b.read_accum()) and np.array_equal(a.read_depth(), b.read_depth())
    sa, sb = a.read_state(), b.read_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k


def _worlds(name):
    return {
        "textured": (lambda: scenes.textured_sphere_scene(200, 120, resolution=160, map_size=64), (1, 1)),
        "living room": (lambda: scenes.living_room(128, 80, 16), (2, 2)),
        "shading inputs": (lambda: scenes.shading_inputs_scene(160, 96), (2, 1)),
        "exact ties": (_ties_world, (1, 1)),
    }[name]


@pytest.mark.parametrize("device", [DEVICE, DEVICE_SAH])
@pytest.mark.parametrize("name", ["textured", "living room", "shading inputs", "exact ties"])
def test_device_built_trees_give_the_same_frames_and_pass_validation(built, name, device):
    build, samples = _worlds(name)
    world = build()
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(LightSampling(*samples), Tracing(6, 4)).struct()
    ref, sah, dev = (_render(flat, cam, cfg, t) for t in (0, 1, device))
    _same_frames(ref, dev)
    _same_frames(sah, dev)
    for xy in [(10, 10), (100, 60), (60, 70)]:
        assert ref.pick(*xy) == dev.pick(*xy)
    print(dev.timings())
    # the device's output as a snapshot: the host's validation (hiprz_validate_scene = check_scene + the walk-table proofs) accepts it
    nodes, root, order, roots, refpos = dev.download_trees(len(flat.instances), len(flat.tris), len(flat.tlas_order))
    assert sorted(refpos.tolist()) == list(range(len(flat.tris)))           # a permutation of the uploaded triangles
    inst = flat.instances.copy()
    inst["blas_root"] = roots
    snap = FlatScene(nodes=nodes, tlas_root=root, tlas_order=order, tris=flat.tris[refpos], tri_attrs=flat.tri_attrs[refpos], instances=inst,
                     inst_materials=flat.inst_materials, materials=flat.materials, textures=flat.textures, texels=flat.texels,
                     spot_lights=flat.spot_lights, direct_lights=flat.direct_lights)
    msg = C.create_string_buffer(256)
    rc = _lib.load().hiprz_validate_scene(C.byref(snap.struct), msg, 256)
    if rc != _abi.OK:  # keep the refused snapshot for a look on the host
        import os
        os.makedirs("gpurun_out/r03", exist_ok=True)
        np.savez_compressed("gpurun_out/r03/refused_snapshot_%s.npz" % name.replace(" ", "_"), **snap.to_npz_dict())
    assert rc == _abi.OK, msg.value
    # ... and rendered from that snapshot with the plain (reference-order, stack / skip-link) walks it is the same frame again
    again = _render(snap, cam, cfg, 0)
    assert np.array_equal(again.read_depth(), dev.read_depth())
    if name != "exact ties":   # (a re-uploaded snapshot ranks equally distant triangles by ITS order: another twin may win a tie)
        assert (again.read_accum()[..., 3] == dev.read_accum()[..., 3]).all()


def _world_tree_of(nodes, root, order):
    """The world tree as a nested tuple: (box, ptype, children) / (box, instances of the leaf)."""
    out = []
    def walk(i):
        n = nodes[i]
        box = (tuple(n["bb_min"].tolist()), tuple(n["bb_max"].tolist()))
        if n["meta"] & _abi.NODE_LEAF:
            count = int(n["meta"] & 0x1FFFFFFF)
            return (box, tuple(order[int(n["begin"]):int(n["begin"]) + count].tolist()))
        return (box, int(n["meta"] >> 29) & 3, walk(int(n["begin"])), walk(int(n["begin"]) + 1))
    return walk(root)


@pytest.mark.parametrize("n_instances", [5, 46, 400])
def test_device_built_world_tree_is_the_host_builders(built, n_instances):
    """The reference's top-down builder run by one device thread: the same nodes, partition types and leaf order as
    hiprz_build_world_tree on the host (the snapshot's own world tree), bit for bit."""
    rng = np.random.default_rng(n_instances)
    world = scenes.living_room(96, 64, min(n_instances, 40))
    cube = world.instances[-1].mesh
    paint = world.materials[0]
    for _ in range(max(0, n_instances - len(world.instances))):
        world.add(Instance(cube, [paint], position=tuple(rng.uniform(-1.8, 1.8, 3) + np.array([0, 1, 0])), rotation=tuple(rng.uniform(0, 1, 3)),
                           scale=tuple(rng.uniform(0.02, 0.3, 3))))
    flat, cam = flatten(world), camera_struct(world.camera)
    dev = Context(0)
    dev.set_tree(DEVICE)
    dev.upload_scene(flat), dev.upload_camera(cam)
    nodes, root, order, _, _ = dev.download_trees(len(flat.instances), len(flat.tris), len(flat.tlas_order))
    assert _world_tree_of(nodes, root, order) == _world_tree_of(flat.nodes, flat.tlas_root, flat.tlas_order)
    assert np.array_equal(order, flat.tlas_order)


@pytest.mark.parametrize("device", [DEVICE, DEVICE_SAH])
def test_refit_and_world_rebuild_equal_a_fresh_upload(built, device):
    """A mesh is deformed and an instance moved: hiprz_update_triangles + hiprz_update_instances on the device against a fresh upload of the
    changed world (host trees)."""
    def build(deformed):
        world = scenes.textured_sphere_scene(200, 120, resolution=96, map_size=64)
        if deformed:
            inst = next(i for i in world.instances if i.name == "bugatti stand-in")
            v = inst.mesh.vertices
            inst.mesh.vertices = np.ascontiguousarray(v * np.array([1.0, 1.25, 0.9], dtype=np.float32) + np.sin(v[:, [1, 2, 0]] * 7.0).astype(np.float32) * np.float32(0.03), dtype=np.float32)
            inst.position = (inst.position + np.array([0.25, -0.1, 0.2], dtype=np.float32)).astype(np.float32)
            inst.rotation = (inst.rotation + np.array([0.1, 0.4, 0.0], dtype=np.float32)).astype(np.float32)
        return world
    before, after = build(False), build(True)
    flat0, flat1, cam = flatten(before), flatten(after), camera_struct(before.camera)
    cfg = RenderConfig(tracing=Tracing(6, 4)).struct()
    dev = _render(flat0, cam, cfg, device)
    # the changed triangles in the order they were uploaded in: a mesh's triangles are identified by their index before leaf reordering
    sphere = next(k for k, i in enumerate(before.instances) if i.name == "bugatti stand-in")
    def mesh_range(flat):
        roots = flat.instances["blas_root"]
        # triangles of the sphere's mesh: the leaves of its tree tile one range
        stack, lo, hi = [int(roots[sphere])], 1 << 62, 0
        while stack:
            n = flat.nodes[stack.pop()]
            if n["meta"] & _abi.NODE_LEAF:
                c = int(n["meta"] & 0x1FFFFFFF)
                if c:
                    lo, hi = min(lo, int(n["begin"])), max(hi, int(n["begin"]) + c)
            else:
                stack += [int(n["begin"]), int(n["begin"]) + 1]
        return lo, hi
    (lo0, hi0), (lo1, hi1) = mesh_range(flat0), mesh_range(flat1)
    assert hi0 - lo0 == hi1 - lo1 > 1000
    src0, src1 = flat0.tris["source_index"][lo0:hi0], flat1.tris["source_index"][lo1:hi1]
    pos1 = np.empty(hi1 - lo1, dtype=np.int64)
    pos1[src1] = np.arange(hi1 - lo1)
    pick = lo1 + pos1[src0]                                       # flat1's record of the triangle flat0 holds at lo0 + k
    dev.update_triangles(lo0, flat1.tris[pick], flat1.tri_attrs[pick])
    dev.update_instances(flat1.instances)
    print(dev.timings())
    for n in (1, 5, 4):
        dev.render(n)
    fresh = _render(flat1, cam, cfg, 0)
    _same_frames(fresh, dev)
    assert not np.array_equal(fresh.read_depth(), _render(flat0, cam, cfg, 0).read_depth())   # the change is visible


@pytest.mark.parametrize("device", [DEVICE, DEVICE_SAH])
def test_config_d_built_on_the_device(built, device):
    """BASELINE config D (301 400 triangles): tree built on the device, the frame of the reference trees; build and refit times printed."""
    preset = scenes.CONFIGS["D"]
    world = preset["build"]()
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(tracing=Tracing(preset["max_depth"], 4)).struct()
    ref, dev = _render(flat, cam, cfg, 0, passes=(1, 2)), _render(flat, cam, cfg, device, passes=(1, 2))
    _same_frames(ref, dev)
    dev.update_triangles(0, flat.tris, flat.tri_attrs)            # a refit of every triangle (to the same place): the frame does not change
    dev.render(1), dev.render(2)
    _same_frames(ref, dev)
    print(dev.timings())


def test_the_hosts_default_trees_fall_back_where_a_rebuild_is_impossible(built):
    """A snapshot whose mesh trees cannot be rebuilt (here: two distinct trees over the SAME triangles — their leaves do not tile ranges
    of their own) is a valid snapshot for the reference's walks.  hiprz_set_tree(1 / 2 / 3) refuse it; HIPRZ_TREE_AUTO, the hosts'
    default, must upload it on its own trees and render the reference trees' frame."""
    from rayzath_amd.engine import TREE_AUTO
    from rayzath_amd._lib import HiprzError
    world = scenes.textured_sphere_scene(200, 120, resolution=96, map_size=64)
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(tracing=Tracing(6, 4)).struct()
    # a second copy of the biggest mesh tree's nodes
    roots = flat.instances["blas_root"]
    def subtree(root):
        out, stack = [], [int(root)]
        while stack:
            i = stack.pop()
            out.append(i)
            n = flat.nodes[i]
            if not (n["meta"] & _abi.NODE_LEAF):
                stack += [int(n["begin"]), int(n["begin"]) + 1]
        return out
    big = max(range(len(roots)), key=lambda k: len(subtree(roots[k])) if roots[k] < len(flat.nodes) else 0)
    old = sorted(subtree(roots[big]))
    where = {o: len(flat.nodes) + k for k, o in enumerate(old)}
    # children stay adjacent: a pair (b, b + 1) is copied to (where[b], where[b] + 1) because `old` is sorted and pairs are adjacent
    copy = flat.nodes[old].copy()
    for k, o in enumerate(old):
        if not (copy[k]["meta"] & _abi.NODE_LEAF):
            assert where[int(copy[k]["begin"]) + 1] == where[int(copy[k]["begin"])] + 1
            copy[k]["begin"] = where[int(copy[k]["begin"])]
    # another instance of the world (one with a mesh of its own) now enters the COPY: two distinct trees over the same triangles, both in use
    other = next(k for k in flat.tlas_order.tolist() if k != big and int(roots[k]) != int(roots[big]) and int(roots[k]) < len(flat.nodes))
    instances = flat.instances.copy()
    instances["blas_root"][other] = where[int(roots[big])]
    snap = FlatScene(nodes=np.concatenate([flat.nodes, copy]), tlas_root=flat.tlas_root, tlas_order=flat.tlas_order, tris=flat.tris, tri_attrs=flat.tri_attrs,
                     instances=instances, inst_materials=flat.inst_materials, materials=flat.materials, textures=flat.textures,
                     texels=flat.texels, spot_lights=flat.spot_lights, direct_lights=flat.direct_lights)
    for tree in (1, DEVICE, DEVICE_SAH):
        with pytest.raises(HiprzError):
            _render(snap, cam, cfg, tree)
    ref, auto = _render(snap, cam, cfg, 0), _render(snap, cam, cfg, TREE_AUTO)
    assert auto.tree() == 0
    _same_frames(ref, auto)
    assert _render(flat, cam, cfg, TREE_AUTO).tree() == DEVICE_SAH   # ... while the plain world gets the device's trees


def _awkward_world():
    """Meshes chosen against the device builders: 5 ... 40 triangles (one small subtree, no top phase), 200 copies of ONE triangle (no
    plane separates anything: the top phase halves runs), zero-area triangles among real ones, triangles along one line of centroids, a
    mesh spanning six orders of magnitude, and a fan of slivers around one vertex."""
    from rayzath_amd.scene import Mesh
    rng = np.random.default_rng(5)
    world = scenes.cornell_box(160, 100)
    paint = [world.add(Material((200, 60, 40, 255), 0.0, 1.0)), world.add(Material((40, 160, 220, 255), 0.2, 0.4))]
    def soup(n, spread=0.5):
        v = rng.uniform(-spread, spread, (n, 3, 3)).astype(np.float32) * np.float32(0.35) + rng.uniform(-spread, spread, (n, 1, 3)).astype(np.float32)
        return v.reshape(-1, 3), np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)
    meshes = []
    for n in (5, 8, 9, 33, 40):
        meshes.append(Mesh(*soup(n), name=f"soup {n}"))
    one = np.array([[-0.3, -0.2, 0.0], [0.3, -0.2, 0.1], [0.0, 0.3, -0.1]], np.float32)
    meshes.append(Mesh(np.tile(one, (200, 1)), np.arange(600, dtype=np.uint32).reshape(-1, 3), tri_materials=(np.arange(200) % 2).astype(np.uint32), name="200 copies"))
    v, t = soup(60)
    v[t[::3, 2]] = v[t[::3, 1]]                                   # every third triangle has no area
    meshes.append(Mesh(v, t, name="zero areas"))
    line = np.linspace(-0.5, 0.5, 70, dtype=np.float32)
    v = np.stack([np.stack([line, 0 * line, 0 * line], 1) + d for d in (np.float32([0, 0.1, 0]), np.float32([0.01, -0.1, 0.05]), np.float32([-0.01, -0.1, -0.05]))], 1).reshape(-1, 3)
    meshes.append(Mesh(v, np.arange(210, dtype=np.uint32).reshape(-1, 3), name="centroids on a line"))
    v, t = soup(50)
    v[: 3 * 10] *= np.float32(1e-4)
    v[3 * 40:] *= np.float32(30.0)
    meshes.append(Mesh(v, t, name="six orders of magnitude"))
    k = 96
    ang = np.linspace(0, 2 * np.pi, k, endpoint=False, dtype=np.float32)
    rim = np.stack([np.cos(ang), np.sin(ang), 0.05 * np.sin(5 * ang)], 1).astype(np.float32) * np.float32(0.5)
    v = np.concatenate([np.zeros((1, 3), np.float32), rim])
    meshes.append(Mesh(v, np.stack([np.zeros(k), 1 + np.arange(k), 1 + (np.arange(k) + 1) % k], 1).astype(np.uint32), name="fan"))
    for i, m in enumerate(meshes):
        m = world.add(m)
        world.add(Instance(m, paint, position=(-0.9 + 0.45 * (i % 5), 0.45 + 0.6 * (i // 5), -0.3 + 0.15 * (i % 3)), rotation=(0.2 * i, 0.3, 0.1 * i),
                           scale=(0.02, 0.02, 0.02) if m.name == "six orders of magnitude" else (0.6, 0.6, 0.6)))
    return world


@pytest.mark.parametrize("device", [DEVICE, DEVICE_SAH])
def test_awkward_meshes_through_the_device_builders(built, device):
    world = _awkward_world()
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(tracing=Tracing(6, 4)).struct()
    ref, dev = _render(flat, cam, cfg, 0), _render(flat, cam, cfg, device)
    assert dev.tree() == device
    _same_frames(ref, dev)
    nodes, root, order, roots, refpos = dev.download_trees(len(flat.instances), len(flat.tris), len(flat.tlas_order))
    assert sorted(refpos.tolist()) == list(range(len(flat.tris)))
    dev.update_triangles(0, flat.tris, flat.tri_attrs)            # a refit to the same place over these trees
    for n in (1, 5, 4):
        dev.render(n)
    _same_frames(ref, dev)


def test_python_engine_refits_a_moved_world_on_the_device(built):
    """World.mark_moved() through rayzath_amd.engine.Engine: with the hosts' default trees (device SAH for this scene) the engine calls
    update_triangles / update_instances and renders what a fresh engine renders from the moved world."""
    from rayzath_amd.engine import Engine
    def build(deformed):
        world = scenes.textured_sphere_scene(200, 120, resolution=96, map_size=64)
        if deformed:
            deform(world)
        return world
    def deform(world):
        inst = next(i for i in world.instances if i.name == "bugatti stand-in")
        v = inst.mesh.vertices
        inst.mesh.vertices = np.ascontiguousarray(v * np.array([1.0, 1.25, 0.9], dtype=np.float32) + np.sin(v[:, [1, 2, 0]] * 7.0).astype(np.float32) * np.float32(0.03), dtype=np.float32)
        inst.position = (inst.position + np.array([0.25, -0.1, 0.2], dtype=np.float32)).astype(np.float32)
        inst.rotation = (inst.rotation + np.array([0.1, 0.4, 0.0], dtype=np.float32)).astype(np.float32)
    cfg = RenderConfig(tracing=Tracing(6, 4))
    world, fresh = build(False), build(True)
    engine, reference = Engine(0, streams=1), Engine(0, streams=1)
    engine.renderWorld(world, cfg)
    assert engine.context.tree() == DEVICE_SAH
    before = world.camera.image_buffer.copy()
    deform(world)
    world.mark_moved()
    engine.renderWorld(world, cfg)
    assert "refit mesh trees (device)" in engine.context.timings() and "build world tree (device)" in engine.context.timings()
    reference.renderWorld(fresh, cfg)
    assert np.array_equal(world.camera.image_buffer, fresh.camera.image_buffer) and np.array_equal(world.camera.depth_buffer, fresh.camera.depth_buffer)
    assert world.camera.ray_count == fresh.camera.ray_count
    assert not np.array_equal(before, world.camera.image_buffer)
    # moved frame after moved frame: every REBUILD_EVERY-th one the device builds the refitted trees again; frames stay a fresh engine's
    engine.REBUILD_EVERY = 2
    for k in range(3):
        for w in (world, fresh):
            inst = next(i for i in w.instances if i.name == "bugatti stand-in")
            inst.mesh.vertices = np.ascontiguousarray(inst.mesh.vertices * np.float32(1.0 + 0.05 * (k + 1)), dtype=np.float32)
        world.mark_moved()
        fresh._dirty = True
        engine.renderWorld(world, cfg), reference.renderWorld(fresh, cfg)
        assert np.array_equal(world.camera.image_buffer, fresh.camera.image_buffer), k
    assert "build mesh trees (device)" in engine.context.timings() and engine._moved_frames == 4
    # on host trees the same call is an ordinary modification
    plain = build(False)
    host = Engine(0, streams=1)
    host.set_tree(0)
    host.renderWorld(plain, cfg)
    deform(plain)
    plain.mark_moved()
    host.renderWorld(plain, cfg)
    again = build(True)
    Engine(0, streams=1).renderWorld(again, cfg)
    assert host.context.tree() == 0 and np.array_equal(plain.camera.image_buffer, again.camera.image_buffer)


@pytest.mark.parametrize("device", [DEVICE, DEVICE_SAH])
def test_refit_reaches_the_meshes_that_are_one_leaf(built, device):
    """A mesh of at most 4 triangles (a wall of a Cornell box) is not built on the device: it stays the single leaf it was uploaded as, and
    hiprz_update_triangles has to fit that leaf's box again too — a sheared wall rendered through its old box loses the hits outside it."""
    def build(sheared):
        world = scenes.cornell_box(160, 100)
        if sheared:
            seen = set()
            for inst in world.instances:
                if inst.mesh is None or id(inst.mesh) in seen:
                    continue
                seen.add(id(inst.mesh))
                v = inst.mesh.vertices
                inst.mesh.vertices = np.ascontiguousarray(v * np.array([1.2, 0.9, 1.1], dtype=np.float32) + v[:, [2, 0, 1]] * np.float32(0.15), dtype=np.float32)
        return world
    before, after = build(False), build(True)
    flat0, flat1, cam = flatten(before), flatten(after), camera_struct(before.camera)
    assert min(len(i.mesh.tri_vertices) for i in before.instances if i.mesh is not None) <= 4
    cfg = RenderConfig(tracing=Tracing(6, 4)).struct()
    dev = _render(flat0, cam, cfg, device)
    # the new records in the uploaded order: the triangle with flat0's source index, mesh by mesh (a mesh's range is the same in both)
    order = np.empty(len(flat0.tris), dtype=np.int64)
    done = 0
    seen = set()
    for inst in before.instances:
        if inst.mesh is None or id(inst.mesh) in seen:
            continue
        seen.add(id(inst.mesh))
        T = len(inst.mesh.tri_vertices)
        where = np.empty(T, dtype=np.int64)
        where[flat1.tris["source_index"][done:done + T]] = np.arange(T)
        order[done:done + T] = done + where[flat0.tris["source_index"][done:done + T]]
        done += T
    assert done == len(flat0.tris)
    dev.update_triangles(0, flat1.tris[order], flat1.tri_attrs[order])
    dev.update_instances(flat1.instances)
    for n in (1, 5, 4):
        dev.render(n)
    fresh = _render(flat1, cam, cfg, 0)
    _same_frames(fresh, dev)
    assert not np.array_equal(fresh.read_depth(), _render(flat0, cam, cfg, 0).read_depth())


def _twisted(v, turns):
    """The vertices turned about the y axis by an angle that grows with y: what a refitted tree copes with badly."""
    a = (v[:, 1] * np.float32(turns)).astype(np.float32)
    c, s_ = np.cos(a).astype(np.float32), np.sin(a).astype(np.float32)
    return np.ascontiguousarray(np.stack([c * v[:, 0] + s_ * v[:, 2], v[:, 1], -s_ * v[:, 0] + c * v[:, 2]], 1), dtype=np.float32)


def test_rebuild_on_the_device_after_a_deformation(built):
    """hiprz_rebuild_trees: a mesh is twisted through hiprz_update_triangles (its refitted tree keeps the old topology), then the device
    builds the trees again over the vertices it holds — Morton order and SAH; frames stay those of a fresh upload of the twisted world,
    further updates keep addressing triangles in the uploaded order, and twisting back gives the first frame again."""
    def build(turns):
        world = scenes.textured_sphere_scene(200, 120, resolution=96, map_size=64)
        inst = next(i for i in world.instances if i.name == "bugatti stand-in")
        if turns:
            inst.mesh.vertices = _twisted(inst.mesh.vertices, turns)
        return world
    worlds = [build(0.0), build(4.0)]
    flats = [flatten(w) for w in worlds]
    cam = camera_struct(worlds[0].camera)
    cfg = RenderConfig(tracing=Tracing(6, 4)).struct()
    sphere = next(k for k, i in enumerate(worlds[0].instances) if i.name == "bugatti stand-in")
    def records_in_uploaded_order(new):
        """flat `new`'s triangles, mesh by mesh, in flats[0]'s order (meshes keep their ranges: same meshes, same counts)."""
        order, done, seen = np.empty(len(flats[0].tris), dtype=np.int64), 0, set()
        for inst in worlds[0].instances:
            if inst.mesh is None or id(inst.mesh) in seen:
                continue
            seen.add(id(inst.mesh))
            T = len(inst.mesh.tri_vertices)
            where = np.empty(T, dtype=np.int64)
            where[new.tris["source_index"][done:done + T]] = np.arange(T)
            order[done:done + T] = done + where[flats[0].tris["source_index"][done:done + T]]
            done += T
        return new.tris[order], new.tri_attrs[order]
    dev = _render(flats[0], cam, cfg, DEVICE_SAH)
    first = (dev.read_accum().copy(), dev.read_depth().copy())
    fresh = _render(flats[1], cam, cfg, 0)
    def frames():
        for n in (1, 5, 4):
            dev.render(n)
    dev.update_triangles(0, *records_in_uploaded_order(flats[1])), dev.update_instances(flats[1].instances)
    frames()
    _same_frames(fresh, dev)                       # refitted
    for tree in (DEVICE, DEVICE_SAH):
        dev.rebuild_trees(tree)
        assert dev.tree() == tree
        frames()
        _same_frames(fresh, dev)                   # rebuilt over the twisted vertices
        nodes, root, order, roots, refpos = dev.download_trees(len(flats[0].instances), len(flats[0].tris), len(flats[0].tlas_order))
        assert sorted(refpos.tolist()) == list(range(len(flats[0].tris)))
    print(dev.timings())
    dev.update_triangles(0, flats[0].tris, flats[0].tri_attrs), dev.update_instances(flats[0].instances)   # back, over the rebuilt trees
    frames()
    assert np.array_equal(dev.read_accum(), first[0]) and np.array_equal(dev.read_depth(), first[1])
    host = Context(0)
    host.upload_scene(flats[0])
    with pytest.raises(Exception):
        host.rebuild_trees()                       # host-built trees: refused


@pytest.mark.parametrize("scene", ["textured sphere", "living room"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_engine_through_a_random_sequence_of_changes(built, seed, scene):
    """Two engines over twin worlds through the same random sequence — frames, moved frames (vertices and transformations), ordinary
    modifications, camera moves, ray-cast pixel moves: one takes the hosts' defaults (device SAH trees, the moved path with a device
    rebuild every third moved frame), the other the snapshot's trees and a full upload at every change.  After every step the images,
    depth buffers, ray counts and ray casts agree."""
    from rayzath_amd.engine import Engine
    rng = np.random.default_rng(seed)
    def build():   # one leaf of 8 instances and no lights / a world tree a few levels deep, lights, instanced meshes
        return scenes.textured_sphere_scene(160, 96, resolution=64, map_size=32) if scene == "textured sphere" else scenes.living_room(160, 96, 40)
    a_world, b_world = build(), build()
    for w in (a_world, b_world):   # a second, smaller camera: every enabled camera is rendered per call, each with a frame state of its own
        second = scenes._camera(96, 64)
        second.position = (second.position + np.array([0.6, 0.2, 0.3], dtype=np.float32)).astype(np.float32)
        w.cameras.append(second)
    a, b = Engine(0, streams=1), Engine(0, streams=1)
    a.REBUILD_EVERY = 3
    b.set_tree(0)
    cfg = RenderConfig(tracing=Tracing(5, 3))
    def both(fn):
        fn(a_world), fn(b_world)
    def compare(step):
        a.renderWorld(a_world, cfg), b.renderWorld(b_world, cfg)
        for ca, cb in zip([a_world.camera] + a_world.cameras, [b_world.camera] + b_world.cameras):
            assert np.array_equal(ca.image_buffer, cb.image_buffer) and np.array_equal(ca.depth_buffer, cb.depth_buffer), step
            assert ca.ray_count == cb.ray_count, step
        ca, cb = a_world.camera, b_world.camera
        ia = a_world.instances.index(ca.raycasted_instance) if getattr(ca, "raycasted_instance", None) is not None else -1
        ib = b_world.instances.index(cb.raycasted_instance) if getattr(cb, "raycasted_instance", None) is not None else -1
        assert ia == ib, step
    compare("first frame")
    assert a.context.tree() == DEVICE_SAH and b.context.tree() == 0
    for step in range(14):
        op = int(rng.integers(0, 6))
        amount = float(rng.uniform(0.02, 0.2))
        which = int(rng.integers(0, len(a_world.instances)))
        if op == 0:      # another frame of the same world: accumulation goes on
            pass
        elif op == 1:    # a mesh deforms, an instance moves: the moved path on one side, an ordinary modification on the other
            def deform(w):
                inst = max((i for i in w.instances if i.mesh is not None), key=lambda i: len(i.mesh.tri_vertices))
                v = inst.mesh.vertices
                inst.mesh.vertices = np.ascontiguousarray(v * np.float32(1.0 + amount) + np.sin(v[:, [2, 0, 1]] * 5.0).astype(np.float32) * np.float32(0.02), dtype=np.float32)
                w.instances[which].position = (w.instances[which].position + np.float32(amount)).astype(np.float32)
            both(deform)
            a_world.mark_moved()
            b_world._dirty = True
        elif op == 2:    # only a transformation
            def turn(w):
                w.instances[which].rotation = (w.instances[which].rotation + np.float32(amount)).astype(np.float32)
            both(turn)
            a_world.mark_moved()
            b_world._dirty = True
        elif op == 3:    # an ordinary modification on both sides (a material's colour)
            def paint(w):
                m = w.materials[which % len(w.materials)]
                m.color = (int(m.color[0]) ^ 0x40, m.color[1], m.color[2], m.color[3])
                w._dirty = True
            both(paint)
        elif op == 4:    # the camera moves: both restart
            def move(w):
                w.camera.position = (w.camera.position + np.array([amount, 0.0, -amount], dtype=np.float32)).astype(np.float32)
            both(move)
        else:            # the ray-cast pixel moves: accumulation goes on
            x, y = int(rng.integers(0, 160)), int(rng.integers(0, 96))
            both(lambda w: w.camera.ray_cast_at(x, y) if hasattr(w.camera, "ray_cast_at") else None)
        compare(f"step {step} op {op}")
    assert "refit mesh trees (device)" in a.context.timings()


@pytest.mark.parametrize("devices", [0, [0, 0]])
@pytest.mark.parametrize("call", ["rebuild_trees", "update_instances"])
def test_a_refused_in_place_change_leaves_no_scene_behind(built, monkeypatch, devices, call):
    """hiprz_rebuild_trees and hiprz_update_instances rewrite node tables, triangle order and instance roots IN PLACE.  When the host then
    refuses to prove the device's output (here: on request — HIPRZ_TEST_REFUSE_TREES stands in for a tree the proof rejects), the call fails
    AND the context has no scene any more: the next render returns HIPRZ_ERR_STATE instead of walking tables nobody proved terminating — on
    every stream that shares the device's scene copy — and a fresh upload brings the context back to the frame it rendered before."""
    from rayzath_amd._lib import HiprzError
    world = scenes.living_room(128, 80, 16)
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(LightSampling(1, 1), Tracing(5, 4)).struct()
    ctx = Context(devices)
    ctx.set_tree(DEVICE_SAH)
    ctx.upload_scene(flat), ctx.upload_camera(cam), ctx.set_config(cfg)
    ctx.render(1), ctx.render(4)
    before = ctx.read_accum()
    monkeypatch.setenv("HIPRZ_TEST_REFUSE_TREES", "1")
    with pytest.raises(HiprzError) as e:
        ctx.rebuild_trees(DEVICE_SAH) if call == "rebuild_trees" else ctx.update_instances(flat.instances)
    assert e.value.code == _abi.ERR_DEVICE and "refused" in str(e.value)
    monkeypatch.delenv("HIPRZ_TEST_REFUSE_TREES")
    for attempt in (lambda: ctx.render(1), lambda: ctx.ray_cast(10, 10), lambda: ctx.update_instances(flat.instances), lambda: ctx.rebuild_trees(DEVICE_SAH)):
        with pytest.raises(HiprzError) as e:
            attempt()
        assert e.value.code == _abi.ERR_STATE, str(e.value)
    ctx.upload_scene(flat)
    ctx.render(1), ctx.render(4)
    assert np.array_equal(ctx.read_accum(), before)
    ctx.close()


def _canonical_trees(nodes, roots, refpos):
    """The mesh trees as a walk sees them, free of slot numbers: per distinct root the depth-first sequence of (box, leaf?, the leaf's
    triangles by uploaded position in leaf order)."""
    out = []
    for root in sorted(set(int(r) for r in roots)):
        seq, stack = [], [root]
        while stack:
            n = nodes[stack.pop()]
            meta, begin = int(n["meta"]), int(n["begin"])
            box = (tuple(n["bb_min"].tolist()), tuple(n["bb_max"].tolist()))
            if meta & 0x80000000:
                seq.append((box, True, tuple(refpos[begin:begin + (meta & 0x1FFFFFFF)].tolist())))
            else:
                seq.append((box, False, (meta >> 29) & 3))
                stack += [begin + 1, begin]
        out.append(seq)
    return sorted(out, key=lambda q: (len(q), q[0][0]))


@pytest.mark.parametrize("name", ["textured", "living room"])
def test_device_sah_trees_are_the_same_from_run_to_run(built, name):
    """The binned surface-area build places triangles with per-wave atomic cursors, so the order inside a node's run depends on wave
    scheduling; the bottom phase puts every small root's run into triangle order before it reads it (rz_sah_small_kernel), which makes the
    emitted trees — topology, boxes, child order, leaf order — a function of the mesh.  (The SLOT a node gets is still handed out by an
    atomic counter: two builds number their nodes differently and hold the same trees.)  Two builds walk the same boxes in the same order:
    the executed-work counters under device SAH trees are reproducible."""
    build, samples = _worlds(name)
    world = build()
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(LightSampling(*samples), Tracing(6, 4)).struct()
    trees, counters = [], []
    for _ in range(3):
        c = _render(flat, cam, cfg, DEVICE_SAH, passes=(1,))
        nodes, root, order, roots, refpos = c.download_trees(len(flat.instances), len(flat.tris), len(flat.tlas_order))
        trees.append(_canonical_trees(nodes, roots[order] if len(order) else roots, refpos))
        c.set_walk_order(2)
        counters.append(c.render_counted(2))
        c.close()
    assert trees[0] == trees[1] == trees[2]
    assert counters[0] == counters[1] == counters[2]
