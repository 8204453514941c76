"""Synthetic scenes for the BASELINE.json configs (the reference ships no scene assets;
SURVEY.md §8d defines these stand-ins).  All generators are deterministic.

  cornell_box            configs A / B: 5 wall quads + 2 boxes + 1 emissive quad (36 triangles)
  cornell_sphere         config C: Cornell box + a UV sphere of `resolution` 80 (6 240 triangles,
                         per-vertex normals + texcrds) standing in for the teapot
  textured_sphere_scene  config D: ~300 k-triangle displaced sphere with texture / normal /
                         roughness maps standing in for the Bugatti
  living_room            config E: instanced meshes, 3 spot + 1 direct light, glass + scattering
"""
import math

import numpy as np

from .scene import (Camera, DirectLight, Instance, Material, SpotLight, TextureBuffer, World, generate_cube,
                    generate_plane, generate_sphere)

HALF = 2.0  # the room is the cube [-2,2] x [-1,3] x [-2,2]


def _room(world, with_boxes=True):
    white = world.add(Material((230, 230, 230, 255), 0.0, 1.0, name="white"))
    red = world.add(Material((200, 40, 40, 255), 0.0, 1.0, name="red"))
    green = world.add(Material((40, 200, 40, 255), 0.0, 1.0, name="green"))
    light = world.add(Material((255, 255, 255, 255), 0.0, 1.0, emission=50.0, name="light"))
    mirror = world.add(Material.mirror())

    w = HALF * math.sqrt(2.0)  # generate_plane's `width` is the half-diagonal
    wall = world.add(generate_plane(4, w, w))
    lamp = world.add(generate_plane(4, 0.7, 0.7))
    cube = world.add(generate_cube())

    hp = math.pi / 2
    world.add(Instance(wall, [white], position=(0, -1, 0), name="floor"))
    world.add(Instance(wall, [white], position=(0, 3, 0), name="ceiling"))
    world.add(Instance(wall, [white], position=(0, 1, HALF), rotation=(hp, 0, 0), name="back"))
    world.add(Instance(wall, [red], position=(-HALF, 1, 0), rotation=(0, 0, hp), name="left"))
    world.add(Instance(wall, [green], position=(HALF, 1, 0), rotation=(0, 0, hp), name="right"))
    world.add(Instance(lamp, [light], position=(0, 2.99, 0), name="lamp"))
    if with_boxes:
        world.add(Instance(cube, [mirror], position=(-0.7, 0.2, 0.6), rotation=(0, 0.3, 0), scale=(1.2, 2.4, 1.2), name="tall box"))
        world.add(Instance(cube, [white], position=(0.7, -0.4, -0.5), rotation=(0, -0.3, 0), scale=(1.2, 1.2, 1.2), name="short box"))
    return dict(white=white, red=red, green=green, light=light, mirror=mirror, wall=wall, cube=cube)


def _camera(width, height):
    return Camera(position=(0, 1, -3.5), rotation=(0, 0, 0), resolution=(width, height), fov=math.pi / 2,
                  near_far=(1.0e-2, 1.0e3), focal_distance=4.0, aperture=0.02, exposure_time=1.0 / 60.0)


def cornell_box(width=256, height=256):
    world = World()
    _room(world)
    world.camera = _camera(width, height)
    return world


def cornell_sphere(width=1920, height=1080, resolution=80):
    world = World()
    parts = _room(world, with_boxes=False)
    sphere = world.add(generate_sphere(resolution, normals=True, texture_coordinates=True))
    gold = world.add(Material.gold())
    world.add(Instance(sphere, [gold], position=(0.0, 0.1, 0.2), rotation=(0.2, 0.4, 0.0), scale=(1.1, 1.1, 1.1), name="sphere"))
    world.add(Instance(parts["cube"], [parts["white"]], position=(1.2, -0.6, -0.9), rotation=(0, 0.5, 0), scale=(0.8, 0.8, 0.8), name="box"))
    world.camera = _camera(width, height)
    return world


def _noise_maps(size, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(size // 8, size // 8, 3), dtype=np.uint8)
    rgb = np.kron(base, np.ones((8, 8, 1), dtype=np.uint8))
    tex = np.concatenate([rgb // 2 + 96, np.full((size, size, 1), 255, np.uint8)], axis=-1).astype(np.uint8)
    n = rng.integers(-24, 25, size=(size, size, 2))
    nrm = np.stack([128 + n[..., 0], 128 + n[..., 1], np.full((size, size), 255), np.full((size, size), 255)], axis=-1).astype(np.uint8)
    rough = rng.integers(20, 200, size=(size, size), dtype=np.uint8)
    return tex, nrm, rough


def textured_sphere_scene(width=1920, height=1080, resolution=550, map_size=2048, seed=1234):
    world = World()
    parts = _room(world, with_boxes=False)
    tex, nrm, rough = _noise_maps(map_size, seed)
    mat = world.add(Material((255, 255, 255, 255), 0.2, 0.4, texture=TextureBuffer(tex), normal_map=TextureBuffer(nrm),
                             roughness_map=TextureBuffer(rough), name="bugatti stand-in"))
    mesh = world.add(generate_sphere(resolution, normals=True, texture_coordinates=True, displace=0.02, displace_seed=seed))
    world.add(Instance(mesh, [mat], position=(0.0, 0.4, 0.0), rotation=(0.3, 0.7, 0.1), scale=(1.4, 1.4, 1.4), name="bugatti stand-in"))
    world.add(Instance(parts["cube"], [parts["mirror"]], position=(1.3, -0.6, -1.0), rotation=(0, 0.5, 0), scale=(0.8, 0.8, 0.8), name="box"))
    world.camera = _camera(width, height)
    return world


def living_room(width=3840, height=2160, n_instances=40, seed=7):
    """Config E stand-in: instancing (6 meshes, `n_instances` instances), several lights so NEE/MIS
    and shadow rays run, glass (transmission) and a scattering material."""
    rng = np.random.default_rng(seed)
    world = World()
    parts = _room(world, with_boxes=False)
    meshes = [parts["cube"], world.add(generate_sphere(24)), world.add(generate_sphere(12, normals=False)),
              world.add(generate_plane(6, 0.5, 0.5)), world.add(generate_sphere(40)), world.add(generate_plane(3, 0.6, 0.6))]
    glass = world.add(Material.glass())
    fog = world.add(Material((255, 255, 255, 0x40), 0.0, 0.3, 0.0, 1.1, 0.5, name="scattering"))
    mats = [parts["white"], parts["red"], parts["green"], parts["mirror"], glass, fog, world.add(Material.gold()),
            world.add(Material((90, 110, 220, 255), 0.1, 0.3, name="blue"))]
    for i in range(n_instances):
        mesh = meshes[i % len(meshes)]
        pos = (rng.uniform(-1.6, 1.6), rng.uniform(-0.8, 2.2), rng.uniform(-1.4, 1.6))
        rot = tuple(rng.uniform(-1.0, 1.0, 3))
        sc = float(rng.uniform(0.25, 0.55))
        world.add(Instance(mesh, [mats[(i * 3) % len(mats)]], position=pos, rotation=rot, scale=(sc, sc * rng.uniform(0.7, 1.3), sc), name=f"object {i}"))
    world.add(SpotLight(position=(-1.2, 2.7, -1.0), direction=(0.4, -1.0, 0.5), color=(255, 240, 220, 255), size=0.15, emission=120.0, beam_angle=0.9))
    world.add(SpotLight(position=(1.2, 2.7, 0.5), direction=(-0.3, -1.0, 0.0), color=(220, 230, 255, 255), size=0.1, emission=90.0, beam_angle=0.7))
    world.add(SpotLight(position=(0.0, 2.5, 1.6), direction=(0.0, -0.7, -1.0), color=(255, 255, 255, 255), size=0.2, emission=60.0, beam_angle=1.2))
    world.add(DirectLight(direction=(0.3, -1.0, 0.8), color=(255, 250, 240, 255), emission=30.0, angular_size=0.05))
    world.camera = _camera(width, height)
    return world


CONFIGS = {
    "A": dict(build=lambda: cornell_box(256, 256), max_depth=4, note="Cornell box 256x256 depth 4 (plumbing / goldens)"),
    "B": dict(build=lambda: cornell_box(1920, 1080), max_depth=8, note="Cornell box 1920x1080 depth 8"),
    "C": dict(build=lambda: cornell_sphere(1920, 1080, 80), max_depth=8, note="Cornell + 6 240-tri sphere, 1920x1080 depth 8"),
    "D": dict(build=lambda: textured_sphere_scene(1920, 1080, 550), max_depth=8, note="301 400-tri textured sphere, 1920x1080 depth 8"),
    "E": dict(build=lambda: living_room(3840, 2160), max_depth=8, note="living room, lights + glass + scattering, 3840x2160"),
    # not a BASELINE config: D's scene with ten times the triangles, so that the geometry (3.06 M triangles: 147 MB of intersection records,
    # 49 MB of walk records) no longer lives in the 32 MB of L2 — what the walks do when nodes and triangles really stream
    "F": dict(build=lambda: textured_sphere_scene(1920, 1080, 1750), max_depth=8, note="3.06 M-tri textured sphere, 1920x1080 depth 8 (geometry beyond L2; not a BASELINE config)"),
}


def shading_inputs_scene(width=160, height=96, seed=11, lights=True):
    """Every shading input the CPU kernel reads that the BASELINE stand-ins do not use: a metalness map (R8),
    an emission map (R32F, cpu_engine_kernel.cpp:523-528), textures with rotation / scale / translation other than the identity
    (render_parts.hpp:209-221), a texture whose alpha channel makes parts of a surface transmissive (:505-512), an emissive AND
    textured world material (the sky a ray meets on a miss, :292-295), instances whose material slot is unset or missing
    (the default material, :359-360), under a spot and a direct light.  Open to the sky: no ceiling, no walls."""
    rng = np.random.default_rng(seed)

    def rgba(h, w, alpha=None, lo=40):
        c = rng.integers(lo, 256, size=(h, w, 4), dtype=np.uint8)
        c[..., 3] = 255 if alpha is None else alpha
        return c

    world = World()
    sky_em = (rng.uniform(0.0, 1.0, size=(16, 32)) ** 3 * 4.0).astype(np.float32)
    sky_em[rng.uniform(size=sky_em.shape) < 0.3] = 0.0          # patches of dark sky: the emission test `> 0` goes both ways
    world.material = Material((255, 255, 255, 0), 0.0, 0.0, 0.0, 1.0, 0.0,
                              texture=TextureBuffer(rgba(32, 64, alpha=0), scale=(2.0, 1.0), rotation=0.35, translation=(0.1, 0.2)),
                              emission_map=TextureBuffer(sky_em, scale=(1.0, 3.0), rotation=-0.2, translation=(0.3, 0.0)), name="sky")

    floor_mat = world.add(Material((255, 255, 255, 255), 0.1, 0.8, name="tiles",
                                   texture=TextureBuffer(rgba(24, 40), scale=(3.0, 2.0), rotation=0.6, translation=(0.25, 0.4)),
                                   roughness_map=TextureBuffer(rng.integers(0, 256, size=(16, 16), dtype=np.uint8), scale=(5.0, 5.0), rotation=1.1)))
    em = np.zeros((12, 12), dtype=np.float32)
    em[2:5, 3:9] = 6.0
    em[8:10, 1:4] = 0.75
    panel = world.add(Material((200, 180, 160, 255), 0.5, 0.3, name="panel",
                               metalness_map=TextureBuffer(rng.integers(0, 256, size=(20, 28), dtype=np.uint8), scale=(2.0, 2.0), translation=(0.5, 0.25)),
                               emission_map=TextureBuffer(em, rotation=0.25)))
    stained = rgba(16, 16)
    stained[..., 3] = np.where(rng.uniform(size=(16, 16)) < 0.5, 255, rng.integers(0, 200, size=(16, 16))).astype(np.uint8)
    window = world.add(Material((255, 255, 255, 255), 0.0, 0.05, 0.0, 1.3, 0.0, name="stained glass",
                                texture=TextureBuffer(stained, scale=(1.5, 1.5), rotation=-0.4, translation=(0.2, 0.7))))
    plain = world.add(Material((90, 110, 220, 255), 0.1, 0.3, name="blue"))

    ground = world.add(generate_plane(4, 6.0, 6.0))
    cube = world.add(generate_cube())
    ball = world.add(generate_sphere(24, normals=True, texture_coordinates=True))
    two_slots = generate_cube()
    two_slots.tri_materials[:] = np.arange(12, dtype=np.uint32) % 3      # slots 0, 1, 2: the instance below fills only 0 and 1
    two_slots = world.add(two_slots)
    world.add(Instance(ground, [floor_mat], position=(0, -1, 0), name="floor"))
    world.add(Instance(cube, [panel], position=(-1.1, -0.2, 0.8), rotation=(0.1, 0.5, 0.0), scale=(1.3, 1.6, 1.3), name="panel box"))
    world.add(Instance(ball, [None], position=(0.9, -0.3, 0.3), scale=(0.7, 0.7, 0.7), name="unset slot"))          # slot present, no material
    world.add(Instance(cube, [], position=(0.2, -0.6, -0.9), rotation=(0.0, 0.3, 0.0), scale=(0.6, 0.8, 0.6), name="no slots"))
    world.add(Instance(two_slots, [plain, None], position=(-0.3, 0.9, 1.6), rotation=(0.4, 0.2, 0.1), scale=(0.9, 0.9, 0.9), name="slot beyond the table"))
    world.add(Instance(ball, [window], position=(1.6, 0.4, 1.5), rotation=(0.0, 1.0, 0.3), scale=(0.9, 0.9, 0.9), name="window ball"))
    if lights:
        world.add(SpotLight(position=(-1.5, 3.0, -1.5), direction=(0.5, -1.0, 0.6), color=(255, 240, 220, 255), size=0.2, emission=150.0, beam_angle=0.8))
        world.add(DirectLight(direction=(0.4, -1.0, 0.5), color=(255, 250, 240, 255), emission=25.0, angular_size=0.06))
    world.camera = Camera(position=(0.0, 0.6, -3.6), rotation=(0.05, 0.0, 0.0), resolution=(width, height), fov=math.pi / 2,
                          near_far=(1.0e-2, 1.0e3), focal_distance=4.0, aperture=0.02, exposure_time=1.0 / 60.0)
    return world
