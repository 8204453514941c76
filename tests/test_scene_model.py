"""Host scene model (rayzath_amd/scene.py): procedural meshes follow RayZath/world.cpp, setters
clamp like the reference's, flattening produces a consistent snapshot."""
import math

import numpy as np
import pytest

from rayzath_amd import _abi, scenes
from rayzath_amd.scene import (Camera, DirectLight, Instance, Material, SpotLight, TextureBuffer, World, camera_struct,
                               flatten, generate_cube, generate_plane, generate_sphere)


def test_procedural_mesh_sizes_follow_the_reference():
    cube = generate_cube()
    assert len(cube.vertices) == 8 and len(cube.texcrds) == 4 and len(cube.tri_vertices) == 12   # world.cpp:129-166
    for sides in (3, 4, 7):
        p = generate_plane(sides, 2.0, 3.0)
        assert len(p.vertices) == sides and len(p.tri_vertices) == sides - 2                      # world.cpp:168-200
        assert np.allclose(p.vertices[:, 1], 0)
    for r in (4, 8, 80):
        s = generate_sphere(r)
        assert len(s.vertices) == r * (r // 2 - 1) + 2 and len(s.tri_vertices) == 2 * r + 2 * r * (r // 2 - 2)  # world.cpp:202-341
        assert np.allclose(np.linalg.norm(s.vertices, axis=1), 1.0, atol=1e-5)
        assert s.tri_texcrds.max() < len(s.texcrds) and s.tri_normals.max() < len(s.normals)
    assert len(generate_sphere(80).tri_vertices) == 6240      # config C stand-in
    # cube winding: every face normal points outwards
    v = cube.vertices[cube.tri_vertices]
    n = np.cross(v[:, 1] - v[:, 2], v[:, 1] - v[:, 0])        # cross(v2 - v3, v2 - v1), mesh_component.cpp:19-26
    assert (np.einsum("ij,ij->i", n, v.mean(axis=1)) > 0).all()


def test_setters_clamp_like_the_reference():
    m = Material(metalness=2, roughness=-1, emission=-5, ior=0.5, scattering=-1)
    assert (m.metalness, m.roughness, m.emission, m.ior, m.scattering) == (1.0, 0.0, 0.0, 1.0, 0.0)   # material.cpp:32-61
    assert SpotLight(beam_angle=10).beam_angle == 3.14159 and SpotLight(size=0).size > 0                 # spot_light.cpp:38-52
    assert DirectLight(angular_size=10).angular_size == pytest.approx(math.pi)
    c = Camera(fov=0, near_far=(0, 0), aperture=0)
    assert c.fov > 0 and c.near_far[1] > c.near_far[0] > 0 and c.aperture > 0                            # camera.cpp:100-152


def test_flatten_cornell(built):
    world = scenes.cornell_box(320, 200)
    flat = flatten(world)
    assert len(flat.instances) == 8 and len(flat.tris) == 2 + 2 + 12          # 3 shared meshes
    assert len(flat.materials) == 2 + 5 and flat.materials["emission"].max() == 50
    assert (flat.inst_materials >= 2).all()
    # instance boxes: walls are flat slabs of the room, boxes sit on the floor
    by_name = {inst.name: flat.instances[i] for i, inst in enumerate(world.instances)}
    assert np.allclose(by_name["floor"]["bb_min"][1], -1) and np.allclose(by_name["floor"]["bb_max"][1], -1, atol=1e-6)
    assert np.allclose(by_name["tall box"]["bb_min"][1], -1, atol=1e-5) and np.allclose(by_name["tall box"]["bb_max"][1], 1.4, atol=1e-5)
    cam = camera_struct(world.camera)
    assert cam.width == 320 and cam.aspect_ratio == pytest.approx(1.6) and cam.tan_half_fov == pytest.approx(1.0, abs=1e-6)
    assert list(cam.z_axis) == [0.0, 0.0, 1.0]


def test_flatten_textures_and_lights(built):
    world = scenes.textured_sphere_scene(64, 64, resolution=16, map_size=32)
    flat = flatten(world)
    assert len(flat.textures) == 3 and sorted(flat.textures["kind"].tolist()) == [_abi.TEX_RGBA8, _abi.TEX_RGBA8, _abi.TEX_R8]
    assert flat.texels.nbytes == 32 * 32 * 4 * 2 + 32 * 32 and (flat.textures["offset"] % 4 == 0).all()
    world = scenes.living_room(64, 64, 10)
    flat = flatten(world)
    assert len(flat.spot_lights) == 3 and len(flat.direct_lights) == 1
    assert np.allclose(np.linalg.norm(flat.spot_lights["direction"], axis=1), 1, atol=1e-6)
    assert np.allclose(flat.direct_lights["cos_angular_size"], np.cos(np.float32(0.05)))


def test_shared_texture_is_stored_once(built):
    world = World()
    tex = TextureBuffer(np.zeros((4, 4, 4), np.uint8))
    a, b = world.add(Material(texture=tex)), world.add(Material(texture=tex))
    world.add(Instance(world.add(generate_cube()), [a, b]))
    flat = flatten(world)
    assert len(flat.textures) == 1 and flat.materials["texture"][2] == flat.materials["texture"][3] == 0
    with pytest.raises(ValueError):
        w2 = World()
        w2.add(Instance(w2.add(generate_cube()), [Material()]))  # material never added to the world
        flatten(w2)


def test_group_transformations_compose_like_the_reference():
    """Transformation::operator*= (render_parts.cpp:75-82) along the chain of groups: position rotated by the group's axes and moved
    by its position, axes rotated, scales multiplied; the box always comes from the composed transformation (instance.cpp:125-155),
    the ray transformation only in "cuda" mode (cuda_instance.cu:244) — "cpu" keeps the instance's own (cpu_engine_kernel.cpp:308)."""
    from rayzath_amd.scene import Group, HostBackend
    backend = HostBackend()
    def world_with(mode):
        w = World()
        m = w.add(Material())
        inst = w.add(Instance(w.add(generate_cube()), [m], position=(0.3, 0.1, -0.2), rotation=(0.1, 0.4, -0.2), scale=(0.5, 0.7, 0.9)))
        inner = w.add(Group(position=(0.2, 0.6, 0.0), rotation=(0.0, 0.0, 0.3), scale=(1.0, 1.2, 1.0), objects=[inst]))
        w.add(Group(position=(0.0, 0.3, 0.4), rotation=(0.0, 0.5, 0.0), scale=(1.1, 1.0, 0.9), groups=[inner]))
        w.group_transforms = mode
        return w, inst
    f32 = np.float32
    w, inst = world_with("cuda")
    rec = flatten(w).instances[0]
    p, s = inst.position.astype(f32), inst.scale.astype(f32)
    axes = list(backend.axes(inst.rotation))
    g = inst.group
    while g is not None:
        gx, gy, gz = backend.axes(g.rotation)
        fwd = lambda v: ((gx * v[0] + gy * v[1]) + gz * v[2]).astype(f32)
        p = (fwd(p) + g.position).astype(f32)
        axes = [fwd(a) for a in axes]
        s = (s * g.scale).astype(f32)
        g = g.group
    assert np.array_equal(rec["position"], p) and np.array_equal(rec["scale"], s)
    assert np.array_equal(rec["x_axis"], axes[0]) and np.array_equal(rec["y_axis"], axes[1]) and np.array_equal(rec["z_axis"], axes[2])
    w_cpu, inst = world_with("cpu")
    rec_cpu = flatten(w_cpu).instances[0]
    assert np.array_equal(rec_cpu["position"], inst.position) and np.array_equal(rec_cpu["scale"], inst.scale)     # own transformation ...
    assert np.array_equal(rec_cpu["bb_min"], rec["bb_min"]) and np.array_equal(rec_cpu["bb_max"], rec["bb_max"])  # ... composed box
