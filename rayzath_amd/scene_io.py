"""Scene files: RayZath's `.json` scene schema, `.obj` and `.mtl`.

Reading is done by the C++ host library (rayzath_amd/csrc/scene_io.cpp behind include/hiprz_io.h, after
RayZath/json_loader.cpp and RayZath/loader.cpp); `load_scene_file` returns the flattened snapshot and the camera
record the engine uploads.  `save_scene_json` / `save_obj` write the Python World model of rayzath_amd.scene in the
same schema (after RayZath/json_saver.cpp, saver.cpp) — that is how the BASELINE configs, which the reference
phrases as "a .json scene" / "an .obj", exist as files."""
import ctypes as C
import json
import os

import numpy as np

from . import _abi, _lib
from .scene import FlatScene

_HOST = None


def host_lib():
    """libhiprz_host.so (C++ host side + scene files).  Fails loudly when it has not been built."""
    global _HOST
    if _HOST is None:
        _lib.load()  # libhiprz.so first: the host library links against it
        path = os.environ.get("HIPRZ_HOST_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libhiprz_host.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: run `make -C rayzath_amd/csrc` (or __graft_entry__.build())")
        lib = C.CDLL(path)
        P = C.c_void_p
        lib.hiprz_scene_file_load.restype, lib.hiprz_scene_file_load.argtypes = C.c_int, [C.c_char_p, C.POINTER(P)]
        lib.hiprz_scene_file_free.restype, lib.hiprz_scene_file_free.argtypes = None, [P]
        lib.hiprz_scene_file_scene.restype, lib.hiprz_scene_file_scene.argtypes = C.POINTER(_abi.Scene), [P]
        lib.hiprz_scene_file_camera.restype, lib.hiprz_scene_file_camera.argtypes = C.POINTER(_abi.Camera), [P]
        lib.hiprz_scene_file_log.restype, lib.hiprz_scene_file_log.argtypes = C.c_char_p, [P]
        lib.hiprz_scene_file_error_count.restype, lib.hiprz_scene_file_error_count.argtypes = C.c_uint32, [P]
        lib.hiprz_scene_file_warning_count.restype, lib.hiprz_scene_file_warning_count.argtypes = C.c_uint32, [P]
        lib.hiprz_scene_file_save.restype, lib.hiprz_scene_file_save.argtypes = C.c_int, [P, C.c_char_p, C.c_int]
        lib.hiprz_io_last_error.restype, lib.hiprz_io_last_error.argtypes = C.c_char_p, []
        U32P = C.POINTER(C.c_uint32)
        lib.hiprz_image_read.restype, lib.hiprz_image_read.argtypes = C.c_int, [C.c_char_p, C.c_uint32, U32P, U32P, U32P, P, C.c_size_t]
        lib.hiprz_image_write_png.restype, lib.hiprz_image_write_png.argtypes = C.c_int, [C.c_char_p, P, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.hiprz_image_read_f32.restype, lib.hiprz_image_read_f32.argtypes = C.c_int, [C.c_char_p, U32P, U32P, P, C.c_size_t]
        lib.hiprz_image_write_hdr.restype, lib.hiprz_image_write_hdr.argtypes = C.c_int, [C.c_char_p, P, C.c_uint32, C.c_uint32]
        _HOST = lib
    return _HOST


IO_ENTRY_POINTS = ("hiprz_scene_file_load", "hiprz_scene_file_free", "hiprz_scene_file_scene", "hiprz_scene_file_camera",
                   "hiprz_scene_file_log", "hiprz_scene_file_error_count", "hiprz_scene_file_warning_count",
                   "hiprz_scene_file_save", "hiprz_io_last_error", "hiprz_image_read", "hiprz_image_write_png", "hiprz_image_read_f32",
                   "hiprz_image_write_hdr")


def read_image(path, channels=0):
    """Decode a PNG / BMP / TGA / binary PNM file with the C++ host library -> uint8 array (height, width, channels).
    `channels` = 0 keeps the file's own channel count, 1..4 converts the way stb_image does (include/hiprz_io.h)."""
    lib = host_lib()
    w, h, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    if lib.hiprz_image_read(os.fsencode(path), channels, C.byref(w), C.byref(h), C.byref(c), None, 0) != 0:
        raise _lib.HiprzError(-2, lib.hiprz_io_last_error().decode())
    out = np.zeros((h.value, w.value, c.value), dtype=np.uint8)
    if lib.hiprz_image_read(os.fsencode(path), channels, None, None, None, out.ctypes.data_as(C.c_void_p), out.nbytes) != 0:
        raise _lib.HiprzError(-2, lib.hiprz_io_last_error().decode())
    return out


def read_image_f32(path):
    """One float per pixel, as stbi_loadf(path, .., 1) gives an emission map (include/hiprz_io.h) -> float32 array (height, width)."""
    lib = host_lib()
    w, h = C.c_uint32(), C.c_uint32()
    if lib.hiprz_image_read_f32(os.fsencode(path), C.byref(w), C.byref(h), None, 0) != 0:
        raise _lib.HiprzError(-2, lib.hiprz_io_last_error().decode())
    out = np.zeros((h.value, w.value), dtype=np.float32)
    if lib.hiprz_image_read_f32(os.fsencode(path), None, None, out.ctypes.data_as(C.c_void_p), out.size) != 0:
        raise _lib.HiprzError(-2, lib.hiprz_io_last_error().decode())
    return out


def write_hdr(path, pixels):
    """float32 array (height, width) -> Radiance .hdr (RGBE), written by the C++ host library."""
    lib = host_lib()
    a = np.ascontiguousarray(pixels, dtype=np.float32)
    if lib.hiprz_image_write_hdr(os.fsencode(path), a.ctypes.data_as(C.c_void_p), a.shape[1], a.shape[0]) != 0:
        raise _lib.HiprzError(-2, lib.hiprz_io_last_error().decode())


_MAP_SLOTS = (("texture", "texture", "Texture"), ("normal_map", "normal map", "NormalMap"), ("metalness_map", "metalness map", "MetalnessMap"),
              ("roughness_map", "roughness map", "RoughnessMap"), ("emission_map", "emission map", "EmissionMap"))
_MAP_DIRS = ("texture", "normal", "metalness", "roughness", "emission")


def _save_maps(materials, scene_dir):
    """Every distinct map of `materials` as a file under <scene_dir>/maps/<kind>/ — the layout and conventions of the C++ saver
    (rayzath_amd/csrc/scene_io.cpp: SavedMaps): RGBA / grey PNG, Radiance .hdr for emission, a normal map's green negated back.
    Returns per kind {id(map): (name, relative file)} and the ordered lists."""
    names, order = [dict() for _ in range(5)], [[] for _ in range(5)]
    for m in materials:
        for k, (attr, _, _) in enumerate(_MAP_SLOTS):
            t = getattr(m, attr)
            if t is None or id(t) in names[k]:
                continue
            i = len(order[k])
            rel = f"maps/{_MAP_DIRS[k]}/{_MAP_DIRS[k]}_{i}" + (".hdr" if k == 4 else ".png")
            os.makedirs(os.path.join(scene_dir, "maps", _MAP_DIRS[k]), exist_ok=True)
            full = os.path.join(scene_dir, rel)
            if k == 4:
                write_hdr(full, t.bitmap)
            elif k == 1:
                px = t.bitmap.copy()
                px[..., 1] = (-px[..., 1].astype(np.int32)) & 0xFF
                write_png(full, px)
            else:
                write_png(full, t.bitmap)
            names[k][id(t)] = (f"{_MAP_DIRS[k]} {i}", rel)
            order[k].append(t)
    return names, order


def write_png(path, pixels):
    """uint8 array (height, width[, channels 1..4]) -> 8-bit PNG, written by the C++ host library."""
    lib = host_lib()
    a = np.ascontiguousarray(pixels, dtype=np.uint8)
    if a.ndim == 2:
        a = a[..., None]
    if lib.hiprz_image_write_png(os.fsencode(path), a.ctypes.data_as(C.c_void_p), a.shape[1], a.shape[0], a.shape[2]) != 0:
        raise _lib.HiprzError(-2, lib.hiprz_io_last_error().decode())


def _copy(ptr, count, dtype):
    if not ptr or count == 0:
        return np.zeros(0, dtype=dtype)
    n = int(count) * np.dtype(dtype).itemsize
    return np.frombuffer(C.string_at(ptr, n), dtype=dtype).copy()


class LoadedScene:
    """What a scene file flattens to: `.flat` (FlatScene), `.camera` (hiprz_camera), `.log`, `.errors`, `.warnings`."""

    def __init__(self, path):
        lib = host_lib()
        handle = C.c_void_p()
        if lib.hiprz_scene_file_load(os.fsencode(path), C.byref(handle)) != 0:
            raise _lib.HiprzError(-2, lib.hiprz_io_last_error().decode())
        try:
            s = lib.hiprz_scene_file_scene(handle).contents
            self.flat = FlatScene(
                nodes=_copy(s.nodes, s.n_nodes, _abi.node_dtype), tlas_order=_copy(s.tlas_order, s.n_tlas_order, np.uint32),
                tris=_copy(s.tris, s.n_tris, _abi.tri_dtype), tri_attrs=_copy(s.tri_attrs, s.n_tris, _abi.tri_attr_dtype),
                instances=_copy(s.instances, s.n_instances, _abi.instance_dtype),
                inst_materials=_copy(s.inst_materials, s.n_inst_materials, np.int32),
                materials=_copy(s.materials, s.n_materials, _abi.material_dtype), textures=_copy(s.textures, s.n_textures, _abi.texture_dtype),
                texels=_copy(s.texels, s.texel_bytes, np.uint8), spot_lights=_copy(s.spot_lights, s.n_spot_lights, _abi.spot_light_dtype),
                direct_lights=_copy(s.direct_lights, s.n_direct_lights, _abi.direct_light_dtype), tlas_root=s.tlas_root)
            cam = _abi.Camera()
            C.memmove(C.byref(cam), lib.hiprz_scene_file_camera(handle), C.sizeof(_abi.Camera))
            self.camera = cam
            self.log = lib.hiprz_scene_file_log(handle).decode()
            self.errors = int(lib.hiprz_scene_file_error_count(handle))
            self.warnings = int(lib.hiprz_scene_file_warning_count(handle))
        except Exception:
            lib.hiprz_scene_file_free(handle)
            raise
        self._handle = handle

    def save(self, path, kind="json"):
        """Write the loaded world back through the C++ writers (kind: "json" or "obj")."""
        lib = host_lib()
        if lib.hiprz_scene_file_save(self._handle, os.fsencode(path), 0 if kind == "json" else 1) != 0:
            raise _lib.HiprzError(-2, lib.hiprz_io_last_error().decode())

    def close(self):
        if self._handle:
            host_lib().hiprz_scene_file_free(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_scene_file(path):
    return LoadedScene(path)


# ------------------------------------------------------------------------------------------------
# writers for the Python World model
# ------------------------------------------------------------------------------------------------
def _f(v):
    return float(np.float32(v))  # json.dumps writes the shortest repr of the double: it reads back to the same float32


def _v(v):
    return [_f(x) for x in np.asarray(v, dtype=np.float32).reshape(-1)]


def _material(m, name=None):
    d = {} if name is None else {"name": name}
    d.update({"color": [int(c) for c in m.color], "metalness": _f(m.metalness), "roughness": _f(m.roughness),
              "emission": _f(m.emission), "ior": _f(m.ior), "scattering": _f(m.scattering)})
    return d


def save_scene_json(world, path):
    """The World as a RayZath .json scene with inline meshes (json_loader.cpp:538-662 reads them back); maps go to
    <dir>/maps/<kind>/ as PNG / .hdr files (JsonSaver::saveMap, json_saver.cpp:117-157)."""
    mats = list(world.materials)
    map_names, map_order = _save_maps(mats + [world.material, world.default_material], os.path.dirname(os.path.abspath(path)))

    def with_maps(body, m):
        for k, (attr, key, _) in enumerate(_MAP_SLOTS):
            t = getattr(m, attr)
            if t is not None:
                body[key] = map_names[k][id(t)][0]
        return body

    mat_name = {id(m): f"material {i}" for i, m in enumerate(mats)}
    meshes, mesh_name = [], {}
    for inst in world.instances:
        if inst.mesh is not None and id(inst.mesh) not in mesh_name:
            mesh_name[id(inst.mesh)] = f"mesh {len(meshes)}"
            meshes.append(inst.mesh)
    unused = 0xFFFFFFFF

    def mesh_json(m, name):
        tris = []
        for t in range(len(m.tri_vertices)):
            e = {"v": [int(x) for x in m.tri_vertices[t]]}
            if m.tri_texcrds[t][0] != unused:
                e["t"] = [int(x) for x in m.tri_texcrds[t]]
            if m.tri_normals[t][0] != unused:
                e["n"] = [int(x) for x in m.tri_normals[t]]
            e["m"] = int(m.tri_materials[t])
            tris.append(e)
        return {"name": name, "vertices": [_v(p) for p in m.vertices], "texcrds": [_v(p) for p in m.texcrds],
                "normals": [_v(p) for p in m.normals], "triangles": tris}

    cam = world.camera
    objects = {
        "Material": [with_maps(_material(m, mat_name[id(m)]), m) for m in mats],
        "Mesh": [mesh_json(m, mesh_name[id(m)]) for m in meshes],
        "Camera": [{"name": "camera", "position": _v(cam.position), "rotation": _v(cam.rotation),
                    "resolution": [int(cam.width), int(cam.height)], "fov": _f(cam.fov), "near plane": _f(cam.near_far[0]),
                    "far plane": _f(cam.near_far[1]), "focal distance": _f(cam.focal_distance), "aperture": _f(cam.aperture),
                    "exposure time": _f(cam.exposure_time), "temporal blend": _f(cam.temporal_blend), "enabled": True}],
        "SpotLight": [{"name": f"spot {i}", "position": _v(l.position), "direction": _v(l.direction), "color": [int(c) for c in l.color],
                       "size": _f(l.size), "emission": _f(l.emission), "angle": _f(l.beam_angle)} for i, l in enumerate(world.spot_lights)],
        "DirectLight": [{"name": f"direct {i}", "direction": _v(l.direction), "color": [int(c) for c in l.color], "emission": _f(l.emission),
                         "size": _f(l.angular_size)} for i, l in enumerate(world.direct_lights)],
        "Instance": [],
    }
    for i, inst in enumerate(world.instances):
        e = {"name": f"instance {i}", "position": _v(inst.position), "rotation": _v(inst.rotation), "scale": _v(inst.scale),
             "Material": [mat_name[id(m)] for m in inst.materials]}
        if inst.mesh is not None:
            e["Mesh"] = mesh_name[id(inst.mesh)]
        objects["Instance"].append(e)
    for k, (_, _, json_key) in enumerate(_MAP_SLOTS):
        if map_order[k]:
            objects[json_key] = [{"name": map_names[k][id(t)][0], "filter mode": t.filter_mode, "address mode": t.address_mode, "scale": [_f(t.scale[0]), _f(t.scale[1])],
                                  "rotation": _f(t.rotation), "translation": [_f(t.translation[0]), _f(t.translation[1])],
                                  "file": map_names[k][id(t)][1]} for t in map_order[k]]
    doc = {"Objects": objects, "Material": with_maps(_material(world.material), world.material),
           "DefaultMaterial": with_maps(_material(world.default_material), world.default_material)}
    with open(path, "w") as f:
        json.dump(doc, f)


def save_obj(world, path):
    """Every instance mesh as one `o` of a Wavefront .obj (+ <stem>.mtl): z negated, corners ordered so that the
    loader's fan (0, i+2, i+1) reads the original triangles back (loader.cpp:807, 1008-1016).  Instance transforms
    are not part of .obj."""
    assert path.endswith(".obj")
    mtl_path = path[:-4] + ".mtl"
    mats = list(world.materials)
    name = {id(m): f"material_{i}" for i, m in enumerate(mats)}
    r = lambda x: repr(_f(x))  # shortest decimal that reads back to the same float32
    map_names, _ = _save_maps(mats, os.path.dirname(os.path.abspath(path)))
    with open(mtl_path, "w") as f:
        for m in mats:
            c = [np.float32(x) / np.float32(255.0) for x in m.color]
            f.write(f"newmtl {name[id(m)]}\nKd {r(c[0])} {r(c[1])} {r(c[2])}\nd {r(c[3])}\nNi {r(m.ior)}\nPm {r(m.metalness)}\n"
                    f"Pr {r(m.roughness)}\nKe {r(m.emission)}\n")
            for k, statement in enumerate(("map_Kd", "norm", "map_Pm", "map_Pr", "map_Ke")):
                t = getattr(m, _MAP_SLOTS[k][0])
                if t is not None:
                    f.write(f'{statement} -o {r(t.translation[0])} {r(t.translation[1])} -s {r(t.scale[0])} {r(t.scale[1])} "{map_names[k][id(t)][1]}"\n')
            f.write("\n")
    unused = 0xFFFFFFFF
    with open(path, "w") as f:
        f.write(f"mtllib {os.path.basename(mtl_path)}\n")
        vb = tb = nb = 0
        for i, inst in enumerate(world.instances):
            m = inst.mesh
            if m is None:
                continue
            f.write(f"o instance_{i}\n")
            for p in m.vertices:
                f.write(f"v {r(p[0])} {r(p[1])} {r(-p[2])}\n")
            for p in m.texcrds:
                f.write(f"vt {r(p[0])} {r(p[1])}\n")
            for p in m.normals:
                f.write(f"vn {r(p[0])} {r(p[1])} {r(-p[2])}\n")
            current = None
            for t in range(len(m.tri_vertices)):
                mid = int(m.tri_materials[t])
                if mid != current:
                    current = mid
                    if mid < len(inst.materials):
                        f.write(f"usemtl {name[id(inst.materials[mid])]}\n")
                corners = []
                for c in (0, 2, 1):
                    s_ = str(vb + int(m.tri_vertices[t][c]) + 1)
                    has_t, has_n = m.tri_texcrds[t][c] != unused, m.tri_normals[t][c] != unused
                    if has_t or has_n:
                        s_ += "/"
                    if has_t:
                        s_ += str(tb + int(m.tri_texcrds[t][c]) + 1)
                    if has_n:
                        s_ += "/" + str(nb + int(m.tri_normals[t][c]) + 1)
                    corners.append(s_)
                f.write("f " + " ".join(corners) + "\n")
            vb += len(m.vertices)
            tb += len(m.texcrds)
            nb += len(m.normals)
