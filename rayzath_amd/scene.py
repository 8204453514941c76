"""Host scene model + flattening into the C-ABI snapshot (include/hiprz.h: hiprz_scene).

This is the Python twin of the part of RayZath's host `World` the render path consumes
(RayZath/world.hpp:64-76): materials, meshes, instances, lights, camera, with the setters'
clamping rules and the procedural meshes of RayZath/world.cpp:129-341.  It exists so the
tests and bench.py can build scenes without the reference's un-vendored host library; a
RayZath-side adapter fills the same `hiprz_scene` from the real `World`.

Tree building / bounds / axes are done by a *backend* object exposing the host entry
points of the C-ABI (`hiprz_build_mesh_tree`, ...).  The default backend is libhiprz.so;
tests pass the oracle library instead to cross-check the two builders.
"""
import ctypes as C
import math

import numpy as np

from . import _abi

F32 = np.float32
PI = F32(math.pi)


def _f3(v):
    return np.asarray(v, dtype=F32).reshape(3)


# ------------------------------------------------------------------------------------------
# World objects (names and defaults follow the reference's ConStruct<...> definitions)
# ------------------------------------------------------------------------------------------
class TextureBuffer:
    """RayZath/render_parts.hpp:113-222. bitmap: (H,W,4) u8 | (H,W) u8 | (H,W) f32, top row first."""

    FILTERS = {"point": 0, "linear": 1}
    ADDRESS_MODES = {"wrap": 0, "clamp": 1, "mirror": 2, "border": 3}

    def __init__(self, bitmap, scale=(1.0, 1.0), rotation=0.0, translation=(0.0, 0.0), filter_mode="point", address_mode="wrap"):
        bitmap = np.ascontiguousarray(bitmap)
        if bitmap.dtype == np.uint8 and bitmap.ndim == 3 and bitmap.shape[2] == 4:
            self.kind = _abi.TEX_RGBA8
        elif bitmap.dtype == np.uint8 and bitmap.ndim == 2:
            self.kind = _abi.TEX_R8
        elif bitmap.dtype == np.float32 and bitmap.ndim == 2:
            self.kind = _abi.TEX_R32F
        else:
            raise ValueError("bitmap must be (H,W,4) u8, (H,W) u8 or (H,W) f32")
        self.bitmap = bitmap
        self.scale = (F32(scale[0]), F32(scale[1]))
        self.rotation = F32(rotation)
        self.translation = (F32(translation[0]), F32(translation[1]))
        # read only in CUDA-compat mode (hiprz_set_mode): the CPU kernel point-samples with wrap-around (render_parts.hpp:209-221)
        self.filter_mode, self.address_mode = filter_mode, address_mode


class Material:
    """RayZath/material.hpp:119-160, setters material.cpp:32-61 (clamps)."""

    def __init__(self, color=(0xC0, 0xC0, 0xC0, 0xFF), metalness=0.0, roughness=0.0, emission=0.0, ior=1.5,
                 scattering=0.0, texture=None, normal_map=None, metalness_map=None, roughness_map=None,
                 emission_map=None, name="material name"):
        self.name = name
        c = tuple(int(x) for x in color)
        self.color = c if len(c) == 4 else c + (255,)
        self.metalness = min(max(float(metalness), 0.0), 1.0)
        self.roughness = min(max(float(roughness), 0.0), 1.0)
        self.emission = max(float(emission), 0.0)
        self.ior = max(float(ior), 1.0)
        self.scattering = max(float(scattering), 0.0)
        self.texture, self.normal_map = texture, normal_map
        self.metalness_map, self.roughness_map, self.emission_map = metalness_map, roughness_map, emission_map

    # presets of RayZath/material.cpp:92-198 that the configs use
    @staticmethod
    def mirror():
        return Material((0xF0, 0xF0, 0xF0, 0xFF), 0.9, 0.0, 0.0, 1.0, 0.0, name="generated_mirror")

    @staticmethod
    def glass():
        return Material((0xFF, 0xFF, 0xFF, 0x00), 0.0, 0.0, 0.0, 1.45, 0.0, name="generated_glass")

    @staticmethod
    def gold():
        return Material((0xFF, 0xD7, 0x00, 0xFF), 1.0, 0.001, 0.0, 1.0, 0.0, name="generated_gold")


class Mesh:
    """RayZath/mesh.hpp: vertices / texcrds / normals + indexed triangles with a material id."""

    def __init__(self, vertices, tri_vertices, texcrds=None, tri_texcrds=None, normals=None, tri_normals=None,
                 tri_materials=None, name="mesh"):
        self.name = name
        self.vertices = np.ascontiguousarray(vertices, dtype=F32).reshape(-1, 3)
        self.texcrds = np.ascontiguousarray(texcrds if texcrds is not None else np.zeros((0, 2)), dtype=F32).reshape(-1, 2)
        self.normals = np.ascontiguousarray(normals if normals is not None else np.zeros((0, 3)), dtype=F32).reshape(-1, 3)
        self.tri_vertices = np.ascontiguousarray(tri_vertices, dtype=np.uint32).reshape(-1, 3)
        T = len(self.tri_vertices)
        unused = np.full((T, 3), _abi.IDS_UNUSED, dtype=np.uint32)
        self.tri_texcrds = np.ascontiguousarray(tri_texcrds if tri_texcrds is not None else unused, dtype=np.uint32).reshape(-1, 3)
        self.tri_normals = np.ascontiguousarray(tri_normals if tri_normals is not None else unused, dtype=np.uint32).reshape(-1, 3)
        self.tri_materials = np.ascontiguousarray(tri_materials if tri_materials is not None else np.zeros(T), dtype=np.uint32).reshape(-1)
        assert len(self.tri_texcrds) == T and len(self.tri_normals) == T and len(self.tri_materials) == T

    def desc(self):
        d = _abi.MeshDesc()
        d.n_vertices, d.vertices = len(self.vertices), self.vertices.ctypes.data
        d.n_texcrds, d.texcrds = len(self.texcrds), self.texcrds.ctypes.data
        d.n_normals, d.normals = len(self.normals), self.normals.ctypes.data
        d.n_triangles = len(self.tri_vertices)
        d.tri_vertices, d.tri_texcrds = self.tri_vertices.ctypes.data, self.tri_texcrds.ctypes.data
        d.tri_normals, d.tri_materials = self.tri_normals.ctypes.data, self.tri_materials.ctypes.data
        return d


class Instance:
    """RayZath/instance.hpp:9-60: transformation + mesh + up to 64 material slots."""

    MATERIAL_CAPACITY = 64

    def __init__(self, mesh, materials=(), position=(0, 0, 0), rotation=(0, 0, 0), scale=(1, 1, 1), name="instance"):
        self.name = name
        self.mesh = mesh
        self.materials = list(materials) if isinstance(materials, (list, tuple)) else [materials]
        assert len(self.materials) <= self.MATERIAL_CAPACITY
        self.position, self.rotation, self.scale = _f3(position), _f3(rotation), _f3(scale)
        self.group = None  # Groupable::group()


class Group:
    """RayZath/group.hpp: a transformation over the instances and sub-groups linked to it (Group::link)."""

    def __init__(self, position=(0, 0, 0), rotation=(0, 0, 0), scale=(1, 1, 1), objects=(), groups=(), name="group"):
        self.name = name
        self.position, self.rotation, self.scale = _f3(position), _f3(rotation), _f3(scale)
        self.group = None
        for o in objects:
            o.group = self
        for g in groups:
            g.group = self


class SpotLight:
    """RayZath/spot_light.cpp:5-52 (clamps)."""

    def __init__(self, position=(0, 0, 0), direction=(0, -1, 0), color=(255, 255, 255, 255), size=0.5, emission=100.0,
                 beam_angle=1.0):
        self.position = _f3(position)
        self.direction = _f3(direction)
        self.color = tuple(int(x) for x in color)
        self.size = max(float(size), float(np.finfo(np.float32).tiny))
        self.emission = max(float(emission), 0.0)
        self.beam_angle = min(max(float(beam_angle), 0.0), 3.14159)


class DirectLight:
    """RayZath/direct_light.cpp:8-45 (clamps)."""

    def __init__(self, direction=(0, -1, 0), color=(255, 255, 255, 255), emission=100.0, angular_size=0.1):
        self.direction = _f3(direction)
        self.color = tuple(int(x) for x in color)
        self.emission = max(float(emission), 0.0)
        self.angular_size = min(max(float(angular_size), 0.0), float(PI))


class Camera:
    """RayZath/camera.hpp:127-161 defaults, camera.cpp setters (clamps)."""

    def __init__(self, position=(0, 0, -10), rotation=(0, 0, 0), resolution=(1280, 720), fov=math.pi / 2,
                 near_far=(1.0e-2, 1.0e3), focal_distance=10.0, aperture=0.02, exposure_time=1.0 / 60.0, enabled=True, temporal_blend=0.75):
        eps = float(np.finfo(np.float32).eps)
        self.enabled = bool(enabled)
        self.temporal_blend = min(max(float(temporal_blend), 0.0), 1.0)   # camera.cpp:154-156; COMPAT_REPROJECTION only
        self.position, self.rotation = _f3(position), _f3(rotation)
        self.width, self.height = max(int(resolution[0]), 1), max(int(resolution[1]), 1)
        self.fov = min(max(float(fov), eps), math.pi - eps)
        near = max(float(near_far[0]), eps)
        self.near_far = (near, max(float(near_far[1]), near + eps))
        self.focal_distance = max(float(focal_distance), eps)
        self.aperture = max(float(aperture), eps)
        self.exposure_time = max(float(exposure_time), eps)
        self.ray_cast_pixel = (0, 0)   # Camera::getRayCastPixel; the engine fills raycasted_instance / raycasted_material after each frame

    def ray_cast_at(self, x, y):
        """Camera::rayCastPixel (camera.cpp:159-165): clamped to the frame.  Accumulation goes on (MakeModified, not RequestUpdate)."""
        self.ray_cast_pixel = (min(max(int(x), 0), self.width - 1), min(max(int(y), 0), self.height - 1))

    def look_at(self, point):
        """Camera::lookAtPoint / lookInDirection, camera.cpp:68-80."""
        d = _f3(point) - self.position
        d = d / F32(np.sqrt(np.sum(d * d, dtype=F32)))
        self.rotation = np.array([math.asin(float(d[1])), -math.atan2(float(d[0]), float(d[2])), 0.0], dtype=F32)


class World:
    """RayZath/world.hpp:64-76 reduced to what the render path reads."""

    def __init__(self):
        self.materials, self.meshes, self.instances = [], [], []
        self.spot_lights, self.direct_lights = [], []
        self.camera = Camera()   # the first camera ...
        self.cameras = []        # ... and the others; every enabled one is rendered per call (cpu_engine_renderer.cpp:97-117)
        self.groups = []
        # "cpu" (default): like the CPU engine — box from the transformation composed through the groups (instance.cpp:125-155), rays
        # into the instance's own transformation (cpu_engine_kernel.cpp:308); "cuda": the composed one for both (cuda_instance.cu:244)
        self.group_transforms = "cpu"
        # world.cpp:33-43; Palette::LightGrey comes from the un-vendored Graphics library, value assumed
        self.material = Material((0xFF, 0xFF, 0xFF, 0x00), 0.0, 0.0, 0.0, 1.0, 0.0, name="world_material")
        self.default_material = Material((0xC0, 0xC0, 0xC0, 0xFF), name="world_default_material")

    def mark_moved(self):
        """Vertices of meshes and / or transformations of instances moved — the same meshes with the same triangles in the same instances
        (an animation frame).  An Engine whose context holds device-built trees refits them on the device and rebuilds the world tree there
        (Context.update_triangles / update_instances) instead of building every tree again on the host, as the reference does at any change
        (component_container.hpp:259-363); otherwise this is `_dirty = True`."""
        self._moved = True

    def add(self, obj):
        {Material: self.materials, Mesh: self.meshes, Instance: self.instances, SpotLight: self.spot_lights,
         DirectLight: self.direct_lights, Group: self.groups}[type(obj)].append(obj)
        return obj


# ------------------------------------------------------------------------------------------
# Procedural meshes: RayZath/world.cpp:129-341 (fp32 arithmetic kept)
# ------------------------------------------------------------------------------------------
def generate_cube():
    """world.cpp:129-166: unit cube, 8 vertices, 4 texcrds, 12 triangles."""
    v = [(-.5, .5, -.5), (-.5, .5, .5), (.5, .5, .5), (.5, .5, -.5), (-.5, -.5, -.5), (-.5, -.5, .5), (.5, -.5, .5), (.5, -.5, -.5)]
    t = [(0, 0), (0, 1), (1, 1), (1, 0)]
    tv = [(1, 2, 0), (3, 0, 2), (4, 7, 5), (6, 5, 7), (0, 3, 4), (7, 4, 3), (2, 1, 6), (5, 6, 1), (3, 2, 7), (6, 7, 2), (1, 0, 5), (4, 5, 0)]
    tt = [(1, 2, 0), (3, 0, 2)] * 6
    return Mesh(v, tv, texcrds=t, tri_texcrds=tt, name="default cube")


def generate_plane(sides=4, width=1.0, height=1.0):
    """world.cpp:168-200: regular polygon in the XZ plane, fan-triangulated."""
    assert sides >= 3
    delta = PI * F32(2.0) / F32(sides)
    offset = delta * F32(0.5)
    verts, uvs = [], []
    for i in range(sides):
        a = delta * F32(i) + offset
        s, c = F32(np.sin(a)), F32(np.cos(a))
        px, py = F32(1.0) * c - F32(0.0) * s, F32(1.0) * s + F32(0.0) * c  # vec2(1,0).Rotate(angle)
        verts.append((px * F32(width), F32(0.0), py * F32(height)))
        uvs.append((px * F32(0.5) + F32(0.5), py * F32(0.5) + F32(0.5)))
    tv = [(0, i + 2, i + 1) for i in range(sides - 2)]
    return Mesh(verts, tv, texcrds=uvs, tri_texcrds=tv, name="generated plane")


def _rot_x(v, a):
    s, c = F32(np.sin(a)), F32(np.cos(a))
    return (v[0], v[1] * c + v[2] * s, v[1] * -s + v[2] * c)


def _rot_y(v, a):
    s, c = F32(np.sin(a)), F32(np.cos(a))
    return (v[0] * c - v[2] * s, v[1], v[0] * s + v[2] * c)


def generate_sphere(resolution=16, normals=True, texture_coordinates=True, displace=0.0, displace_seed=0):
    """world.cpp:202-341 UV sphere: 2r + 2r(r/2-2) triangles.  `displace` (ours, for the
    Bugatti-class stand-in of SURVEY.md §8d) pushes every vertex radially by hash noise."""
    r = int(resolution)
    assert r >= 4
    half = r // 2
    d_theta = PI / F32(half)
    d_phi = F32(2.0) * PI / F32(r)
    th = (d_theta * (np.arange(half - 1, dtype=F32) + F32(1.0))).astype(F32)
    ph = (d_phi * np.arange(r, dtype=F32)).astype(F32)
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    # v = (0,1,0).RotateX(a_theta).RotateY(a_phi)
    y1 = np.cos(TH).astype(F32)           # y*c + z*s with y=1, z=0
    z1 = (-np.sin(TH)).astype(F32)        # y*-s + z*c
    x2 = (F32(0.0) * np.cos(PH).astype(F32) - z1 * np.sin(PH).astype(F32)).astype(F32)
    z2 = (F32(0.0) * np.sin(PH).astype(F32) + z1 * np.cos(PH).astype(F32)).astype(F32)
    ring = np.stack([x2, y1, z2], axis=-1).reshape(-1, 3).astype(F32)
    verts = np.concatenate([ring, np.array([[0, 1, 0], [0, -1, 0]], dtype=F32)])
    top_v, bottom_v = len(ring), len(ring) + 1
    nrm = verts.copy() if normals else None
    if displace:
        rng = np.random.default_rng(displace_seed)
        verts = (verts * (F32(1.0) + F32(displace) * rng.uniform(-1, 1, size=(len(verts), 1)).astype(F32))).astype(F32)

    uvs = None
    top_t = bottom_t = 0
    r_pi = F32(1.0 / math.pi)
    if texture_coordinates:
        rows = []
        for t in range(half - 1):
            a_theta = d_theta * F32(t + 1)
            u = (ph * F32(0.5) * r_pi).astype(F32)
            vv = np.full(r, F32(1.0) - a_theta * r_pi, dtype=F32)
            rows.append(np.stack([u, vv], axis=-1))
            rows.append(np.array([[F32(1.0), F32(1.0) - (d_theta * F32(t + 1)) * r_pi]], dtype=F32))
        base = sum(len(x) for x in rows)
        cap_u = (np.arange(r, dtype=F32) / F32(r) + F32(0.5) / F32(r)).astype(F32)
        top_t = base
        rows.append(np.stack([cap_u, np.ones(r, dtype=F32)], axis=-1))
        bottom_t = base + r
        rows.append(np.stack([cap_u, np.zeros(r, dtype=F32)], axis=-1))
        uvs = np.concatenate(rows).astype(F32)

    tv, tt = [], []
    for i in range(r):  # top and bottom fans
        tv.append((top_v, (i + 1) % r, i))
        tt.append((top_t + i, i + 1, i))
        tv.append((bottom_v, top_v - r + i, top_v - r + (i + 1) % r))
        tt.append((bottom_t + i, (top_t - r + i - 1) & 0xFFFFFFFF, top_t - r + i))
    for t in range(half - 2):  # middle layers
        for p in range(r):
            tv.append((t * r + p, t * r + (p + 1) % r, (t + 1) * r + (p + 1) % r))
            tt.append((t * (r + 1) + p, t * (r + 1) + (p + 1), (t + 1) * (r + 1) + (p + 1)))
            tv.append((t * r + p, (t + 1) * r + (p + 1) % r, (t + 1) * r + p))
            tt.append((t * (r + 1) + p, (t + 1) * (r + 1) + (p + 1), (t + 1) * (r + 1) + p))
    tv = np.array(tv, dtype=np.uint32)
    tt = np.array(tt, dtype=np.int64).astype(np.uint32) if texture_coordinates else None
    if texture_coordinates:
        # the reference's bottom-fan texcrd ids (world.cpp:297-300) reach below the last ring row
        # for i = 0; keep them in range (the reference would index out of bounds there)
        tt = np.minimum(tt, np.uint32(len(uvs) - 1))
    return Mesh(verts, tv, texcrds=uvs, tri_texcrds=tt if texture_coordinates else None, normals=nrm,
                tri_normals=tv if normals else None, name="generated sphere")


# ------------------------------------------------------------------------------------------
# Flattening
# ------------------------------------------------------------------------------------------
class HostBackend:
    """Host entry points of libhiprz.so (default) — or any library exporting the same six
    functions under another prefix (tests pass prefix='rzo_' with the oracle library)."""

    def __init__(self, lib=None, prefix="hiprz_"):
        if lib is None:
            from . import _lib
            lib = _lib.load()
        self.lib, self.prefix = lib, prefix
        if prefix != "hiprz_":
            for name in ("build_mesh_tree", "build_world_tree", "instance_bounds", "axes_from_rotation", "axes_look_at"):
                restype, argtypes = _abi.ENTRY_POINTS["hiprz_" + name]
                fn = getattr(lib, prefix + name)
                fn.restype, fn.argtypes = restype, argtypes

    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def axes(self, rotation, look_at=False):
        r = _f3(rotation)
        x, y, z = (np.zeros(3, dtype=F32) for _ in range(3))
        self._fn("axes_look_at" if look_at else "axes_from_rotation")(r.ctypes.data, x.ctypes.data, y.ctypes.data, z.ctypes.data)
        return x, y, z

    def mesh_tree(self, mesh):
        T = len(mesh.tri_vertices)
        max_nodes = 2 * T + 1
        nodes = np.zeros(max_nodes, dtype=_abi.node_dtype)
        tris = np.zeros(max(T, 1), dtype=_abi.tri_dtype)
        attrs = np.zeros(max(T, 1), dtype=_abi.tri_attr_dtype)
        n = C.c_uint32(0)
        d = mesh.desc()
        rc = self._fn("build_mesh_tree")(C.byref(d), nodes.ctypes.data, max_nodes, C.byref(n), tris.ctypes.data, attrs.ctypes.data)
        if rc != 0:
            raise ValueError(f"{self.prefix}build_mesh_tree failed ({rc}) for mesh {mesh.name!r}")
        return nodes[:n.value].copy(), tris[:T].copy(), attrs[:T].copy()

    def world_tree(self, instances, has_mesh):
        n_inst = len(instances)
        max_nodes = 2 * n_inst + 1
        nodes = np.zeros(max_nodes, dtype=_abi.node_dtype)
        order = np.zeros(max(n_inst, 1), dtype=np.uint32)
        n, n_order = C.c_uint32(0), C.c_uint32(0)
        has = np.ascontiguousarray(has_mesh, dtype=np.uint8)
        rc = self._fn("build_world_tree")(instances.ctypes.data, has.ctypes.data, n_inst, nodes.ctypes.data, max_nodes,
                                          C.byref(n), order.ctypes.data, C.byref(n_order))
        if rc != 0:
            raise ValueError(f"{self.prefix}build_world_tree failed ({rc})")
        return nodes[:n.value].copy(), order[:n_order.value].copy()

    def fill_triangles(self, mesh, order):
        """The mesh's triangles `order` (indices into the mesh) as device records, in that order (hiprz_fill_triangles)."""
        order = np.ascontiguousarray(order, dtype=np.uint32)
        tris, attrs = np.zeros(len(order), dtype=_abi.tri_dtype), np.zeros(len(order), dtype=_abi.tri_attr_dtype)
        d = mesh.desc()
        rc = self._fn("fill_triangles")(C.byref(d), order.ctypes.data, len(order), tris.ctypes.data, attrs.ctypes.data)
        if rc != 0:
            raise ValueError(f"{self.prefix}fill_triangles failed ({rc}) for mesh {mesh.name!r}")
        return tris, attrs

    def instance_bounds(self, vertices, inst_record):
        self._fn("instance_bounds")(vertices.ctypes.data, len(vertices), inst_record.ctypes.data)


class FlatScene:
    """Owns the numpy arrays a hiprz_scene points into."""

    FIELDS = ("nodes", "tlas_order", "tris", "tri_attrs", "instances", "inst_materials", "materials", "textures",
              "texels", "spot_lights", "direct_lights")

    def __init__(self, **arrays):
        for k in self.FIELDS:
            setattr(self, k, np.ascontiguousarray(arrays[k]))
        self.tlas_root = int(arrays.get("tlas_root", 0))
        # the map objects behind `textures`, by identity, in texture-index order (None: unknown, e.g. a snapshot loaded from a file)
        self.map_ids = arrays.get("map_ids")
        s = _abi.Scene()
        s.n_nodes, s.nodes = len(self.nodes), self.nodes.ctypes.data
        s.tlas_root = self.tlas_root
        s.n_tlas_order, s.tlas_order = len(self.tlas_order), self.tlas_order.ctypes.data
        s.n_tris, s.tris, s.tri_attrs = len(self.tris), self.tris.ctypes.data, self.tri_attrs.ctypes.data
        s.n_instances, s.instances = len(self.instances), self.instances.ctypes.data
        s.n_inst_materials, s.inst_materials = len(self.inst_materials), self.inst_materials.ctypes.data
        s.n_materials, s.materials = len(self.materials), self.materials.ctypes.data
        s.n_textures, s.textures = len(self.textures), self.textures.ctypes.data
        s.texel_bytes, s.texels = self.texels.nbytes, self.texels.ctypes.data
        s.n_spot_lights, s.spot_lights = len(self.spot_lights), self.spot_lights.ctypes.data
        s.n_direct_lights, s.direct_lights = len(self.direct_lights), self.direct_lights.ctypes.data
        self.struct = s

    def to_npz_dict(self):
        d = {k: getattr(self, k) for k in self.FIELDS}
        d["tlas_root"] = np.uint32(self.tlas_root)
        return d

    @staticmethod
    def from_npz_dict(d):
        """Inverse of to_npz_dict (np.savez keeps the structured dtypes)."""
        return FlatScene(tlas_root=int(d["tlas_root"]), **{k: d[k] for k in FlatScene.FIELDS})


def _normalize3(v):
    v = _f3(v)
    return (v * (F32(1.0) / F32(np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])))).astype(F32)


def _forward(axes, v):
    """CoordSystem::transformForward (render_parts.cpp:40-43): x_axis * v.x + y_axis * v.y + z_axis * v.z in fp32."""
    x, y, z = axes
    v = np.asarray(v, dtype=F32)
    return ((x * v[0] + y * v[1]) + z * v[2]).astype(F32)


def _in_group(inst, backend):
    """Transformation of an instance composed through its groups, Transformation::operator*= (render_parts.cpp:75-82)."""
    p, s = inst.position.astype(F32), inst.scale.astype(F32)
    x, y, z = backend.axes(inst.rotation)
    g = getattr(inst, "group", None)
    while g is not None:
        axes = backend.axes(g.rotation)
        p = (_forward(axes, p) + g.position.astype(F32)).astype(F32)
        x, y, z = _forward(axes, x), _forward(axes, y), _forward(axes, z)
        s = (s * g.scale.astype(F32)).astype(F32)
        g = g.group
    return p, s, x, y, z


def flatten_motion(world, uploaded_sources, backend=None):
    """Triangles and instance records only, for Context.update_triangles / update_instances: every mesh's triangles in the order of an
    earlier flatten() (`uploaded_sources` = its tris["source_index"]), no tree is built.  None when the world no longer matches that order."""
    backend = backend or HostBackend()
    seen, cursor, tri_parts, attr_parts = set(), 0, [], []
    for inst in world.instances:
        if inst.mesh is None or id(inst.mesh) in seen:
            continue
        seen.add(id(inst.mesh))
        T = len(inst.mesh.tri_vertices)
        if cursor + T > len(uploaded_sources) or (T and int(uploaded_sources[cursor:cursor + T].max()) >= T):
            return None
        tris, attrs = backend.fill_triangles(inst.mesh, uploaded_sources[cursor:cursor + T])
        tri_parts.append(tris), attr_parts.append(attrs)
        cursor += T
    if cursor != len(uploaded_sources):
        return None
    instances = np.zeros(len(world.instances), dtype=_abi.instance_dtype)
    for i, inst in enumerate(world.instances):
        r = instances[i:i + 1]
        own = (inst.position, inst.scale) + backend.axes(inst.rotation)
        r["position"], r["scale"], r["x_axis"], r["y_axis"], r["z_axis"] = _in_group(inst, backend)
        if inst.mesh is not None:
            backend.instance_bounds(inst.mesh.vertices, r)
        if world.group_transforms == "cpu":
            r["position"], r["scale"], r["x_axis"], r["y_axis"], r["z_axis"] = own
    tris = np.concatenate(tri_parts) if tri_parts else np.zeros(0, _abi.tri_dtype)
    attrs = np.concatenate(attr_parts) if attr_parts else np.zeros(0, _abi.tri_attr_dtype)
    return tris, attrs, instances


def flatten(world, backend=None):
    """World -> FlatScene (the arrays `hiprz_upload_scene` copies)."""
    backend = backend or HostBackend()
    # textures, deduplicated by identity
    tex_index, tex_records, pool, map_ids = {}, [], bytearray(), []

    def tex_id(t):
        if t is None:
            return -1
        if id(t) not in tex_index:
            while len(pool) % 4:
                pool.append(0)
            rec = np.zeros(1, dtype=_abi.texture_dtype)[0]
            rec["kind"], rec["height"], rec["width"] = t.kind, t.bitmap.shape[0], t.bitmap.shape[1]
            rec["offset"] = len(pool)
            rec["scale"], rec["translation"], rec["rotation"] = t.scale, t.translation, t.rotation
            rec["cos_rotation"], rec["sin_rotation"] = _cosf(F32(t.rotation)), _sinf(F32(t.rotation))
            rec["sampling"] = TextureBuffer.FILTERS[t.filter_mode] | (TextureBuffer.ADDRESS_MODES[t.address_mode] << 8)
            pool.extend(t.bitmap.tobytes())
            tex_index[id(t)] = len(tex_records)
            tex_records.append(rec)
            map_ids.append(id(t))
        return tex_index[id(t)]

    mats = [world.material, world.default_material] + list(world.materials)
    mat_index = {id(m): i for i, m in enumerate(mats)}
    materials = np.zeros(len(mats), dtype=_abi.material_dtype)
    for i, m in enumerate(mats):
        r = materials[i]
        r["color"] = m.color
        r["metalness"], r["roughness"], r["emission"], r["ior"], r["scattering"] = m.metalness, m.roughness, m.emission, m.ior, m.scattering
        r["texture"], r["normal_map"] = tex_id(m.texture), tex_id(m.normal_map)
        r["metalness_map"], r["roughness_map"], r["emission_map"] = tex_id(m.metalness_map), tex_id(m.roughness_map), tex_id(m.emission_map)

    # meshes (one tree each), concatenated after the world tree
    mesh_trees, mesh_slot = [], {}
    for inst in world.instances:
        if inst.mesh is not None and id(inst.mesh) not in mesh_slot:
            mesh_slot[id(inst.mesh)] = len(mesh_trees)
            mesh_trees.append(backend.mesh_tree(inst.mesh))

    instances = np.zeros(len(world.instances), dtype=_abi.instance_dtype)
    inst_materials = []
    has_mesh = np.zeros(len(world.instances), dtype=np.uint8)
    for i, inst in enumerate(world.instances):
        r = instances[i:i + 1]
        own = (inst.position, inst.scale) + backend.axes(inst.rotation)
        grouped = _in_group(inst, backend)   # the bounding box always comes from the composed transformation (instance.cpp:125-155)
        r["position"], r["scale"], r["x_axis"], r["y_axis"], r["z_axis"] = grouped
        r["material_base"], r["material_count"] = len(inst_materials), len(inst.materials)
        for m in inst.materials:
            if m is not None and id(m) not in mat_index:
                raise ValueError(f"instance {inst.name!r} uses a material that was not added to the world")
            inst_materials.append(-1 if m is None else mat_index[id(m)])
        if inst.mesh is not None:
            has_mesh[i] = 1
            backend.instance_bounds(inst.mesh.vertices, r)
        if world.group_transforms == "cpu":  # ... but the CPU kernel takes rays into the instance's OWN transformation
            r["position"], r["scale"], r["x_axis"], r["y_axis"], r["z_axis"] = own

    world_nodes, order = backend.world_tree(instances, has_mesh) if len(instances) else (np.zeros(0, _abi.node_dtype), np.zeros(0, np.uint32))
    node_parts, tri_parts, attr_parts = [world_nodes], [], []
    node_base, tri_base, roots = len(world_nodes), 0, []
    for nodes, tris, attrs in mesh_trees:
        nodes = nodes.copy()
        leaf = (nodes["meta"] & _abi.NODE_LEAF) != 0
        nodes["begin"] = np.where(leaf, nodes["begin"] + np.uint32(tri_base), nodes["begin"] + np.uint32(node_base))
        roots.append(node_base)
        node_parts.append(nodes)
        tri_parts.append(tris)
        attr_parts.append(attrs)
        node_base += len(nodes)
        tri_base += len(tris)
    for i, inst in enumerate(world.instances):
        if inst.mesh is not None:
            instances[i]["blas_root"] = roots[mesh_slot[id(inst.mesh)]]

    spots = np.zeros(len(world.spot_lights), dtype=_abi.spot_light_dtype)
    for i, l in enumerate(world.spot_lights):
        r = spots[i]
        r["position"], r["size"], r["direction"], r["emission"] = l.position, l.size, _normalize3(l.direction), l.emission
        r["color"], r["angle"], r["cos_angle"] = l.color, l.beam_angle, _cosf(F32(l.beam_angle))
    directs = np.zeros(len(world.direct_lights), dtype=_abi.direct_light_dtype)
    for i, l in enumerate(world.direct_lights):
        r = directs[i]
        r["direction"], r["emission"], r["color"] = _normalize3(l.direction), l.emission, l.color
        r["angular_size"], r["cos_angular_size"] = l.angular_size, _cosf(F32(l.angular_size))

    return FlatScene(
        nodes=np.concatenate(node_parts) if node_parts else np.zeros(0, _abi.node_dtype), tlas_root=0, tlas_order=order,
        tris=np.concatenate(tri_parts) if tri_parts else np.zeros(0, _abi.tri_dtype),
        tri_attrs=np.concatenate(attr_parts) if attr_parts else np.zeros(0, _abi.tri_attr_dtype),
        instances=instances, inst_materials=np.array(inst_materials, dtype=np.int32), materials=materials,
        textures=np.array(tex_records, dtype=_abi.texture_dtype) if tex_records else np.zeros(0, _abi.texture_dtype),
        texels=np.frombuffer(bytes(pool), dtype=np.uint8).copy() if pool else np.zeros(0, np.uint8),
        spot_lights=spots, direct_lights=directs, map_ids=tuple(map_ids))


def camera_struct(cam, backend=None):
    """Camera -> hiprz_camera (tan(fov/2) hoisted with fp32 libm, as the kernel would compute it)."""
    backend = backend or HostBackend()
    c = _abi.Camera()
    x, y, z = backend.axes(cam.rotation, look_at=True)
    c.position[:], c.x_axis[:], c.y_axis[:], c.z_axis[:] = cam.position.tolist(), x.tolist(), y.tolist(), z.tolist()
    c.width, c.height = cam.width, cam.height
    c.fov = cam.fov
    c.tan_half_fov = float(_tanf(F32(F32(cam.fov) * F32(0.5))))
    c.aspect_ratio = float(F32(cam.width) / F32(cam.height))
    c.near_far[:] = cam.near_far
    c.focal_distance, c.aperture, c.exposure_time = cam.focal_distance, cam.aperture, cam.exposure_time
    return c


_libm = C.CDLL("libm.so.6")
_libm.tanf.restype, _libm.tanf.argtypes = C.c_float, [C.c_float]
_libm.cosf.restype, _libm.cosf.argtypes = C.c_float, [C.c_float]
_libm.sinf.restype, _libm.sinf.argtypes = C.c_float, [C.c_float]


def _tanf(x):
    return F32(_libm.tanf(float(x)))


def _cosf(x):  # the C++ host side hoists the same values with std::cos / std::sin on floats: both hosts give the same bits
    return F32(_libm.cosf(float(x)))


def _sinf(x):
    return F32(_libm.sinf(float(x)))
