"""ctypes / numpy mirror of include/hiprz.h (POD records of the C-ABI).

Every record is declared twice on purpose: as a numpy structured dtype (to build arrays of
records) and as a ctypes Structure (for the by-pointer arguments).  tests/test_abi.py
checks the sizes and field offsets of both against the sizes the header states.
"""
import ctypes as C

import numpy as np

OK, ERR_INVALID, ERR_DEVICE, ERR_STATE = 0, 1, 2, 3

NODE_LEAF = 0x80000000
NODE_PTYPE_SHIFT = 29
NODE_COUNT_MASK = 0x1FFFFFFF
TRI_HAS_TEXCRDS = 0x40000000
TRI_HAS_NORMALS = 0x80000000
TRI_MATERIAL_MASK = 0x00FFFFFF
MATERIAL_WORLD, MATERIAL_DEFAULT = 0, 1
TEX_RGBA8, TEX_R8, TEX_R32F = 0, 1, 2
IDS_UNUSED = 0xFFFFFFFF

node_dtype = np.dtype([("bb_min", "<f4", 3), ("bb_max", "<f4", 3), ("begin", "<u4"), ("meta", "<u4")])
tri_dtype = np.dtype([("v1", "<f4", 3), ("material_flags", "<u4"), ("v2", "<f4", 3), ("source_index", "<u4"),
                      ("v3", "<f4", 3), ("pad0", "<u4")])
tri_attr_dtype = np.dtype([("n1", "<f4", 3), ("pad0", "<f4"), ("n2", "<f4", 3), ("pad1", "<f4"), ("n3", "<f4", 3),
                           ("pad2", "<f4"), ("face_normal", "<f4", 3), ("pad3", "<f4"), ("t1", "<f4", 2),
                           ("t2", "<f4", 2), ("t3", "<f4", 2), ("pad4", "<f4", 2)])
instance_dtype = np.dtype([("position", "<f4", 3), ("blas_root", "<u4"), ("scale", "<f4", 3), ("material_base", "<u4"),
                           ("x_axis", "<f4", 3), ("material_count", "<u4"), ("y_axis", "<f4", 3), ("pad0", "<u4"),
                           ("z_axis", "<f4", 3), ("pad1", "<u4"), ("bb_min", "<f4", 3), ("pad2", "<u4"),
                           ("bb_max", "<f4", 3), ("pad3", "<u4")])
material_dtype = np.dtype([("color", "u1", 4), ("metalness", "<f4"), ("roughness", "<f4"), ("emission", "<f4"),
                           ("ior", "<f4"), ("scattering", "<f4"), ("texture", "<i4"), ("normal_map", "<i4"),
                           ("metalness_map", "<i4"), ("roughness_map", "<i4"), ("emission_map", "<i4"), ("pad0", "<u4")])
texture_dtype = np.dtype([("kind", "<u4"), ("width", "<u4"), ("height", "<u4"), ("offset", "<u4"), ("scale", "<f4", 2),
                          ("translation", "<f4", 2), ("rotation", "<f4"), ("cos_rotation", "<f4"),
                          ("sin_rotation", "<f4"), ("sampling", "<u4")])
spot_light_dtype = np.dtype([("position", "<f4", 3), ("size", "<f4"), ("direction", "<f4", 3), ("emission", "<f4"),
                             ("color", "u1", 4), ("angle", "<f4"), ("cos_angle", "<f4"), ("pad0", "<u4")])
direct_light_dtype = np.dtype([("direction", "<f4", 3), ("emission", "<f4"), ("color", "u1", 4), ("angular_size", "<f4"),
                               ("cos_angular_size", "<f4"), ("pad0", "<u4")])

RECORD_SIZES = {"node": 32, "tri": 48, "tri_attr": 96, "instance": 112, "material": 48, "texture": 48,
                "spot_light": 48, "direct_light": 32}
RECORD_DTYPES = {"node": node_dtype, "tri": tri_dtype, "tri_attr": tri_attr_dtype, "instance": instance_dtype,
                 "material": material_dtype, "texture": texture_dtype, "spot_light": spot_light_dtype,
                 "direct_light": direct_light_dtype}


class Scene(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_uint32), ("nodes", C.c_void_p),
        ("tlas_root", C.c_uint32), ("n_tlas_order", C.c_uint32), ("tlas_order", C.c_void_p),
        ("n_tris", C.c_uint32), ("tris", C.c_void_p), ("tri_attrs", C.c_void_p),
        ("n_instances", C.c_uint32), ("instances", C.c_void_p),
        ("n_inst_materials", C.c_uint32), ("inst_materials", C.c_void_p),
        ("n_materials", C.c_uint32), ("materials", C.c_void_p),
        ("n_textures", C.c_uint32), ("textures", C.c_void_p),
        ("texel_bytes", C.c_size_t), ("texels", C.c_void_p),
        ("n_spot_lights", C.c_uint32), ("spot_lights", C.c_void_p),
        ("n_direct_lights", C.c_uint32), ("direct_lights", C.c_void_p),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3), ("x_axis", C.c_float * 3), ("y_axis", C.c_float * 3), ("z_axis", C.c_float * 3),
        ("width", C.c_uint32), ("height", C.c_uint32), ("fov", C.c_float), ("tan_half_fov", C.c_float),
        ("aspect_ratio", C.c_float), ("near_far", C.c_float * 2), ("focal_distance", C.c_float),
        ("aperture", C.c_float), ("exposure_time", C.c_float),
    ]


class Config(C.Structure):
    _fields_ = [("max_depth", C.c_uint32), ("rpp", C.c_uint32), ("spot_samples", C.c_uint32),
                ("direct_samples", C.c_uint32), ("seed", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("segments", "box_tests", "tri_tests", "hits", "shadow_rays", "light_samples",
                                          "texel_fetches", "finished", "shadow_box_tests", "shadow_tri_tests")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class MeshDesc(C.Structure):
    _fields_ = [
        ("n_vertices", C.c_uint32), ("vertices", C.c_void_p),
        ("n_texcrds", C.c_uint32), ("texcrds", C.c_void_p),
        ("n_normals", C.c_uint32), ("normals", C.c_void_p),
        ("n_triangles", C.c_uint32), ("tri_vertices", C.c_void_p), ("tri_texcrds", C.c_void_p),
        ("tri_normals", C.c_void_p), ("tri_materials", C.c_void_p),
    ]


class RayCast(C.Structure):  # hiprz_raycast
    _fields_ = [("instance", C.c_int32), ("material_slot", C.c_int32), ("material", C.c_int32), ("triangle", C.c_uint32)]


# name -> (restype, argtypes) of every entry point include/hiprz.h declares
P = C.c_void_p
U32, U64, SZ, I32 = C.c_uint32, C.c_uint64, C.c_size_t, C.c_int32
ENTRY_POINTS = {
    "hiprz_create": (C.c_int, [C.POINTER(P), C.c_int]),
    "hiprz_destroy": (C.c_int, [P]),
    "hiprz_last_error": (C.c_char_p, [P]),
    "hiprz_validate_scene": (C.c_int, [C.POINTER(Scene), C.c_char_p, SZ]),
    "hiprz_abi_sizes": (None, [P]),
    "hiprz_upload_scene": (C.c_int, [P, C.POINTER(Scene)]),
    "hiprz_upload_camera": (C.c_int, [P, C.POINTER(Camera)]),
    "hiprz_set_config": (C.c_int, [P, C.POINTER(Config)]),
    "hiprz_set_shard": (C.c_int, [P, U32, U32]),
    "hiprz_set_shard_mode": (C.c_int, [P, U32]),
    "hiprz_shard_mode": (C.c_int, [P, C.POINTER(U32)]),
    "hiprz_set_traversal_mode": (C.c_int, [P, C.c_int]),
    "hiprz_set_walk_order": (C.c_int, [P, C.c_int]),
    "hiprz_set_mode": (C.c_int, [P, U32]),
    "hiprz_set_temporal_blend": (C.c_int, [P, C.c_float]),
    "hiprz_create_multi": (C.c_int, [C.POINTER(P), C.POINTER(C.c_int), C.c_int]),
    "hiprz_device_count": (C.c_int, [P, C.POINTER(U32)]),
    "hiprz_set_camera_count": (C.c_int, [P, U32]),
    "hiprz_camera_count": (C.c_int, [P, C.POINTER(U32)]),
    "hiprz_select_camera": (C.c_int, [P, U32]),
    "hiprz_update_shading": (C.c_int, [P, P, U32, P, U32, P, U32]),
    "hiprz_set_tree": (C.c_int, [P, U32]),
    "hiprz_tree": (C.c_int, [P, C.POINTER(U32)]),
    "hiprz_rebuild_trees": (C.c_int, [P, U32]),
    "hiprz_rebuild_mesh_trees": (C.c_int, [C.POINTER(Scene), U32, P, U32, C.POINTER(U32), P, P, C.POINTER(U32)]),
    "hiprz_update_triangles": (C.c_int, [P, U32, U32, P, P]),
    "hiprz_update_instances": (C.c_int, [P, P, U32]),
    "hiprz_download_trees": (C.c_int, [P, P, U32, C.POINTER(U32), C.POINTER(U32), P, P, P]),
    "hiprz_set_lds_scene": (C.c_int, [P, C.c_int]),
    "hiprz_set_pipeline": (C.c_int, [P, C.c_int]),
    "hiprz_traversal_mode": (C.c_int, [P, C.POINTER(C.c_int)]),
    "hiprz_pipeline": (C.c_int, [P, C.POINTER(C.c_int)]),
    "hiprz_set_ray_sort": (C.c_int, [P, C.c_int]),
    "hiprz_set_xcd_swizzle": (C.c_int, [P, C.c_int]),
    "hiprz_set_graph": (C.c_int, [P, C.c_int]),
    "hiprz_graph_captures": (C.c_int, [P, C.POINTER(U32)]),
    "hiprz_reset": (C.c_int, [P]),
    "hiprz_render": (C.c_int, [P, U32]),
    "hiprz_render_counted": (C.c_int, [P, U32, C.POINTER(Counters)]),
    "hiprz_tonemap": (C.c_int, [P]),
    "hiprz_sync": (C.c_int, [P]),
    "hiprz_read_rgba8": (C.c_int, [P, P, SZ]),
    "hiprz_read_depth": (C.c_int, [P, P, SZ]),
    "hiprz_read_accum": (C.c_int, [P, P, SZ]),
    "hiprz_read_state": (C.c_int, [P, P, P, SZ]),
    "hiprz_ray_count": (C.c_int, [P, C.POINTER(U64)]),
    "hiprz_pass_count": (C.c_int, [P, C.POINTER(U32)]),
    "hiprz_local_pixel_capacity": (C.c_int, [P, C.POINTER(SZ)]),
    "hiprz_export_accum_tiles": (C.c_int, [P, P, SZ]),
    "hiprz_export_rgba8_tiles": (C.c_int, [P, P, SZ]),
    "hiprz_untile_gathered": (C.c_int, [P, C.c_void_p, C.c_uint32, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p]),
    "hiprz_untile_rgba8": (C.c_int, [P, P, U32, U32, P]),
    "hiprz_untile_accum": (C.c_int, [P, P, U32, U32, P]),
    "hiprz_tonemap_image": (C.c_int, [P, P, P]),
    "hiprz_tonemap_image_on": (C.c_int, [P, P, P, P]),
    "hiprz_stream": (P, [P]),
    "hiprz_pick": (C.c_int, [P, U32, U32, C.POINTER(I32), C.POINTER(I32)]),
    "hiprz_ray_cast": (C.c_int, [P, U32, U32, C.POINTER(RayCast)]),
    "hiprz_selftest": (C.c_int, [P, U32, U32, C.POINTER(U64), C.POINTER(U64)]),
    "hiprz_selftest_sort": (C.c_int, [P, C.POINTER(U32), U32, C.c_int, U32, C.POINTER(U64), C.POINTER(C.c_double)]),
    "hiprz_timings": (C.c_int, [P, C.c_char_p, SZ]),
    "hiprz_time_kernels": (C.c_int, [P, C.c_int]),
    "hiprz_kernel_breakdown_ms": (C.c_int, [P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(U32)]),
    "hiprz_kernel_time_ms": (C.c_int, [P, C.POINTER(C.c_double), C.POINTER(U64)]),
    "hiprz_build_mesh_tree": (C.c_int, [C.POINTER(MeshDesc), P, U32, C.POINTER(U32), P, P]),
    "hiprz_build_world_tree": (C.c_int, [P, P, U32, P, U32, C.POINTER(U32), P, C.POINTER(U32)]),
    "hiprz_fill_triangles": (C.c_int, [P, P, U32, P, P]),
    "hiprz_instance_bounds": (C.c_int, [P, U32, P]),
    "hiprz_axes_from_rotation": (None, [P, P, P, P]),
    "hiprz_axes_look_at": (None, [P, P, P, P]),
    "hiprz_seed_value": (C.c_float, [U32, U32, U32]),
    "hiprz_version": (C.c_char_p, []),
    "hiprz_kernel_count": (U32, []),
}


def bind(lib):
    """Attach restype/argtypes to every entry point; raises AttributeError if one is missing."""
    for name, (restype, argtypes) in ENTRY_POINTS.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return lib
