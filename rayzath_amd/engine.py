"""Python host side of the HIPGPU backend, above the C-ABI.

`Engine` mirrors the reference's backend interface — `renderWorld(World&, const
RenderConfig&, bool block, bool sync)` and `timingsString()` (RayZath/cpu_engine.hpp:17-22,
RayZath/cuda_engine.cuh:34-39) — and `RenderConfig/LightSampling/Tracing` mirror
RayZath/engine_parts.hpp:76-128.  `Context` is a thin handle around `hiprz_ctx`.
All compute happens in libhiprz.so; there is no CPU path here.
"""
import ctypes as C

import numpy as np

from . import _abi, _lib
from ._lib import HiprzError
from .scene import HostBackend, camera_struct, flatten, flatten_motion


class LightSampling:
    def __init__(self, spot_light=1, direct_light=1):
        self.spot_light, self.direct_light = int(spot_light), int(direct_light)


class Tracing:
    def __init__(self, max_depth=16, rpp=8):
        self.max_depth, self.rpp = int(max_depth), int(rpp)


class RenderConfig:
    def __init__(self, light_sampling=None, tracing=None, seed=20240501):
        self.light_sampling = light_sampling or LightSampling()
        self.tracing = tracing or Tracing()
        self.seed = int(seed)

    def struct(self):
        return _abi.Config(self.tracing.max_depth, self.tracing.rpp, self.light_sampling.spot_light,
                           self.light_sampling.direct_light, self.seed & 0xFFFFFFFF)


COMPAT_BEER_LAMBERT, COMPAT_SCATTERING, COMPAT_SHADOW_COLOR, COMPAT_TEXTURE_MULT, COMPAT_FILTERING, COMPAT_REPROJECTION = 1, 2, 4, 8, 16, 32  # hiprz_set_mode


TREE_REFERENCE, TREE_SAH, TREE_DEVICE, TREE_DEVICE_SAH, TREE_AUTO = range(5)   # hiprz_set_tree (include/hiprz.h)
SHARD_TILES, SHARD_SAMPLES = 0, 1   # hiprz_set_shard_mode


def default_streams(n_lights):
    """How many contexts-with-a-stream the hosts put on ONE GPU (hiprz_create_multi with the device named that often, tiles interleaved):
    two when the scene has no lights — one half's sorts, pass bookkeeping and kernel tails run beside the other half's walks (measured on
    MI355X: config B +6 %, C +12 %, D +5 %) — one when it has (the deferred shadow kernel and the two sorts of a pass already overlap on
    streams of their own, and halving their grids costs more than it hides: config E -10 %)."""
    return 1 if n_lights else 2


class Context:
    """Owns one hiprz_ctx: one GPU and one stream — or, given a list of device ids, one context over several GPUs
    (hiprz_create_multi: tiles interleaved over the devices, readbacks gather the peers' tiles over peer-to-peer copies)."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        self._ctx = C.c_void_p()
        if isinstance(device, (list, tuple)):
            ids = (C.c_int * len(device))(*[int(d) for d in device])
            rc = self.lib.hiprz_create_multi(C.byref(self._ctx), ids, len(device))
        else:
            rc = self.lib.hiprz_create(C.byref(self._ctx), int(device))
        if rc != _abi.OK:
            raise HiprzError(rc, (self.lib.hiprz_last_error(None) or b"").decode())
        self.width = self.height = 0
        self._sizes = {}

    def device_count(self):
        v = C.c_uint32()
        self._check(self.lib.hiprz_device_count(self._ctx, C.byref(v)))
        return v.value

    # --- cameras: one frame state each, the calls below address the selected one ---
    def set_camera_count(self, n):
        self._check(self.lib.hiprz_set_camera_count(self._ctx, n))

    def camera_count(self):
        v = C.c_uint32()
        self._check(self.lib.hiprz_camera_count(self._ctx, C.byref(v)))
        return v.value

    def select_camera(self, index):
        self._check(self.lib.hiprz_select_camera(self._ctx, index))
        self._sizes[getattr(self, "_camera", 0)] = (self.width, self.height)
        self._camera = index
        self.width, self.height = self._sizes.get(index, (0, 0))

    def update_shading(self, flat_scene):
        """Materials and lights of `flat_scene` replace those of the uploaded scene in place (same material count); no tree work.
        Map indices are positions in the UPLOADED scene's texture list: when `flat_scene` numbers other map objects, or the same ones in
        another order (a material re-pointed at another uploaded map changes the first-use order), the whole scene is uploaded instead."""
        f = flat_scene
        if getattr(f, "map_ids", None) != getattr(self, "_uploaded_map_ids", None) or f.map_ids is None:
            return self.upload_scene(f)
        self._check(self.lib.hiprz_update_shading(self._ctx, f.materials.ctypes.data, len(f.materials), f.spot_lights.ctypes.data, len(f.spot_lights),
                                                  f.direct_lights.ctypes.data, len(f.direct_lights)))

    def close(self):
        if self._ctx:
            self.lib.hiprz_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != _abi.OK:
            raise HiprzError(rc, (self.lib.hiprz_last_error(self._ctx) or b"").decode())

    # --- uploads ---
    def upload_scene(self, flat_scene):
        self._check(self.lib.hiprz_upload_scene(self._ctx, C.byref(flat_scene.struct)))
        self._uploaded_map_ids = getattr(flat_scene, "map_ids", None)

    def upload_camera(self, camera_struct_):
        self._check(self.lib.hiprz_upload_camera(self._ctx, C.byref(camera_struct_)))
        self.width, self.height = camera_struct_.width, camera_struct_.height

    def set_config(self, config_struct):
        self._check(self.lib.hiprz_set_config(self._ctx, C.byref(config_struct)))

    def set_shard(self, rank, world):
        self._check(self.lib.hiprz_set_shard(self._ctx, rank, world))

    def set_shard_mode(self, mode):
        """How the parts of a context over several devices / streams divide its share (hiprz_set_shard_mode): SHARD_TILES (default) —
        interleaved tiles, the one-device frame bit for bit; SHARD_SAMPLES — every part renders the whole share on the seed stream
        seed + part and the accumulators are summed wherever the frame leaves the context."""
        self._check(self.lib.hiprz_set_shard_mode(self._ctx, mode))

    def shard_mode(self):
        v = C.c_uint32()
        self._check(self.lib.hiprz_shard_mode(self._ctx, C.byref(v)))
        return v.value

    def set_traversal_mode(self, mode):
        self._check(self.lib.hiprz_set_traversal_mode(self._ctx, mode))

    def set_walk_order(self, order):
        """0 = mesh children in the reference's order (counters equal the CPU kernel's), 1 = front to back (default)."""
        self._check(self.lib.hiprz_set_walk_order(self._ctx, order))

    def set_mode(self, compat_flags):
        """0 = the CPU kernel (default, parity-checked); COMPAT_* flags add behaviours of the reference's CUDA engine (include/hiprz.h)."""
        self._check(self.lib.hiprz_set_mode(self._ctx, compat_flags))

    def set_temporal_blend(self, blend):
        """Camera::temporalBlend of the selected camera: the weight of the previous frame's history at a restart (COMPAT_REPROJECTION)."""
        self._check(self.lib.hiprz_set_temporal_blend(self._ctx, blend))

    def tree(self):
        """The trees of the uploaded scene (hiprz_tree): TREE_REFERENCE .. TREE_DEVICE_SAH."""
        out = C.c_uint32(0)
        self._check(self.lib.hiprz_tree(self._ctx, C.byref(out)))
        return out.value

    def set_tree(self, tree):
        """0 = the uploaded (reference) mesh trees, 1 = rebuilt with a binned SAH at the next upload_scene (same frames, fewer tests),
        2 = all trees built on the device at upload (same frames; update_triangles / update_instances afterwards), 3 = as 2 with the
        device's binned surface-area builder for the mesh trees, 4 (TREE_AUTO) = 0 for scenes staged in LDS, 3 for all others."""
        self._check(self.lib.hiprz_set_tree(self._ctx, tree))

    def update_triangles(self, first, tris, attrs):
        """New records (numpy arrays of _abi.tri_dtype / tri_attr_dtype) for triangles [first, first + len) of the uploaded order; the device
        refits the mesh trees that hold them.  Scenes uploaded under set_tree(2) only."""
        tris, attrs = np.ascontiguousarray(tris), np.ascontiguousarray(attrs)
        assert len(tris) == len(attrs)
        self._check(self.lib.hiprz_update_triangles(self._ctx, first, len(tris), tris.ctypes.data, attrs.ctypes.data))

    def rebuild_trees(self, tree=TREE_DEVICE_SAH):
        """Every mesh tree built again by the device over the vertices it holds now (after update_triangles deformed a mesh far from the
        shape its tree was built for); TREE_DEVICE or TREE_DEVICE_SAH.  Scenes with device-built trees only."""
        self._check(self.lib.hiprz_rebuild_trees(self._ctx, tree))

    def update_instances(self, instances):
        """New transformations and world boxes of ALL instances (array of _abi.instance_dtype); the device rebuilds the world tree."""
        instances = np.ascontiguousarray(instances)
        self._check(self.lib.hiprz_update_instances(self._ctx, instances.ctypes.data, len(instances)))

    def download_trees(self, n_instances, n_tris, n_tlas_order):
        """(nodes, tlas_root, tlas_order, blas_roots, tri_refpos) of the trees the context walks now."""
        n = C.c_uint32()
        self._check(self.lib.hiprz_download_trees(self._ctx, None, 0, C.byref(n), None, None, None, None))
        nodes = np.zeros(n.value, dtype=_abi.node_dtype)
        root = C.c_uint32()
        order, roots, refpos = np.zeros(max(n_tlas_order, 1), np.uint32), np.zeros(max(n_instances, 1), np.uint32), np.zeros(max(n_tris, 1), np.uint32)
        self._check(self.lib.hiprz_download_trees(self._ctx, nodes.ctypes.data, len(nodes), C.byref(n), C.byref(root), order.ctypes.data, roots.ctypes.data, refpos.ctypes.data))
        return nodes, root.value, order[:n_tlas_order], roots[:n_instances], refpos[:n_tris]

    def set_pipeline(self, pipeline):
        self._check(self.lib.hiprz_set_pipeline(self._ctx, pipeline))

    def pipeline(self):
        v = C.c_int()
        self._check(self.lib.hiprz_pipeline(self._ctx, C.byref(v)))
        return v.value

    def graph_captures(self):
        v = C.c_uint32()
        self._check(self.lib.hiprz_graph_captures(self._ctx, C.byref(v)))
        return v.value

    def traversal_mode(self):
        v = C.c_int()
        self._check(self.lib.hiprz_traversal_mode(self._ctx, C.byref(v)))
        return v.value

    def set_lds_scene(self, mode):
        self._check(self.lib.hiprz_set_lds_scene(self._ctx, mode))

    # --- rendering ---
    def reset(self):
        self._check(self.lib.hiprz_reset(self._ctx))

    def render(self, n_passes):
        self._check(self.lib.hiprz_render(self._ctx, n_passes))

    def render_counted(self, n_passes):
        out = _abi.Counters()
        self._check(self.lib.hiprz_render_counted(self._ctx, n_passes, C.byref(out)))
        return out.as_dict()

    def tonemap(self):
        self._check(self.lib.hiprz_tonemap(self._ctx))

    def sync(self):
        self._check(self.lib.hiprz_sync(self._ctx))

    # --- readback ---
    def read_rgba8(self):
        out = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        self._check(self.lib.hiprz_read_rgba8(self._ctx, out.ctypes.data, out.nbytes))
        return out

    def read_depth(self):
        out = np.zeros((self.height, self.width), dtype=np.float32)
        self._check(self.lib.hiprz_read_depth(self._ctx, out.ctypes.data, out.nbytes))
        return out

    def read_accum(self):
        out = np.zeros((self.height, self.width, 4), dtype=np.float32)
        self._check(self.lib.hiprz_read_accum(self._ctx, out.ctypes.data, out.nbytes))
        return out

    def read_state(self):
        n = self.width * self.height
        ray = np.zeros((self.height, self.width, 9), dtype=np.float32)
        md = np.zeros((self.height, self.width, 2), dtype=np.uint32)
        self._check(self.lib.hiprz_read_state(self._ctx, ray.ctypes.data, md.ctypes.data, n))
        return dict(origin=ray[..., 0:3], direction=ray[..., 3:6], color=ray[..., 6:9], material=md[..., 0], depth=md[..., 1])

    def ray_count(self):
        v = C.c_uint64()
        self._check(self.lib.hiprz_ray_count(self._ctx, C.byref(v)))
        return v.value

    def pass_count(self):
        v = C.c_uint32()
        self._check(self.lib.hiprz_pass_count(self._ctx, C.byref(v)))
        return v.value

    def pick(self, x, y):
        i, m = C.c_int32(), C.c_int32()
        self._check(self.lib.hiprz_pick(self._ctx, x, y, C.byref(i), C.byref(m)))
        return i.value, m.value

    def ray_cast(self, x, y):
        """Kernel::rayCast through pixel (x, y) of the selected camera's current frame: (instance, material slot, material, triangle's
        index in its mesh), -1 where nothing was met / the slot is unset."""
        r = _abi.RayCast()
        self._check(self.lib.hiprz_ray_cast(self._ctx, x, y, C.byref(r)))
        return r.instance, r.material_slot, r.material, r.triangle

    def selftest(self, cases_per_thread=64, seed=1):
        bad, n = C.c_uint64(), C.c_uint64()
        self._check(self.lib.hiprz_selftest(self._ctx, cases_per_thread, seed, C.byref(bad), C.byref(n)))
        return bad.value, n.value

    def selftest_sort(self, keys, key_bits=24, repeats=1):
        """The ray-order radix sort on `keys` (uint32 array): (violations of "stable permutation in key order", device microseconds)."""
        keys = np.ascontiguousarray(keys, dtype=np.uint32)
        bad, us = C.c_uint64(), C.c_double()
        self._check(self.lib.hiprz_selftest_sort(self._ctx, keys.ctypes.data_as(C.POINTER(C.c_uint32)), keys.size, int(key_bits), int(repeats),
                                                 C.byref(bad), C.byref(us)))
        return bad.value, us.value

    def timings(self):
        buf = C.create_string_buffer(4096)
        self._check(self.lib.hiprz_timings(self._ctx, buf, len(buf)))
        return buf.value.decode()

    def time_kernels(self, enabled=True):
        self._check(self.lib.hiprz_time_kernels(self._ctx, int(enabled)))

    def kernel_breakdown_ms(self):
        """(trace kernel ms, shade kernel ms, passes) of the last batch of cumulative passes (split pipeline)."""
        t, s_, n = C.c_double(), C.c_double(), C.c_uint32()
        self._check(self.lib.hiprz_kernel_breakdown_ms(self._ctx, C.byref(t), C.byref(s_), C.byref(n)))
        return t.value, s_.value, n.value

    def kernel_time_ms(self):
        """(total device ms, passes) of the pass kernel since the last call (hip events on the ctx stream)."""
        t, n = C.c_double(), C.c_uint64()
        self._check(self.lib.hiprz_kernel_time_ms(self._ctx, C.byref(t), C.byref(n)))
        return t.value, n.value

    # --- multi-GPU hand-off (device pointers) ---
    def local_pixel_capacity(self):
        v = C.c_size_t()
        self._check(self.lib.hiprz_local_pixel_capacity(self._ctx, C.byref(v)))
        return v.value

    def export_accum_tiles(self, dst_ptr, nbytes):
        self._check(self.lib.hiprz_export_accum_tiles(self._ctx, dst_ptr, nbytes))

    def export_rgba8_tiles(self, dst_ptr, nbytes):
        self._check(self.lib.hiprz_export_rgba8_tiles(self._ctx, dst_ptr, nbytes))

    def untile_gathered(self, src_ptr, world, part_stride_bytes, element_bytes, dst_ptr, stream=None):
        self._check(self.lib.hiprz_untile_gathered(self._ctx, src_ptr, world, part_stride_bytes, element_bytes, dst_ptr, stream))

    def untile_rgba8(self, src_ptr, rank, world, dst_ptr):
        self._check(self.lib.hiprz_untile_rgba8(self._ctx, src_ptr, rank, world, dst_ptr))

    def set_ray_sort(self, mode):
        self._check(self.lib.hiprz_set_ray_sort(self._ctx, mode))

    def set_xcd_swizzle(self, enabled):
        self._check(self.lib.hiprz_set_xcd_swizzle(self._ctx, int(enabled)))

    def set_graph(self, enabled):
        self._check(self.lib.hiprz_set_graph(self._ctx, int(enabled)))

    def untile_accum(self, src_ptr, rank, world, dst_ptr):
        self._check(self.lib.hiprz_untile_accum(self._ctx, src_ptr, rank, world, dst_ptr))

    def tonemap_image(self, src_ptr, dst_ptr, stream=None):
        self._check(self.lib.hiprz_tonemap_image_on(self._ctx, src_ptr, dst_ptr, stream))

    def stream(self):
        return self.lib.hiprz_stream(self._ctx)


class Engine:
    """HIPGPU peer of CPU::Engine / Cuda::Engine.  Writes its results into the world's camera
    like the reference backends do (imageBuffer / depthBuffer / rayCount, camera.hpp:50-56,113-119)."""

    REBUILD_EVERY = 16   # moved frames (World.mark_moved) between two device rebuilds of the refitted trees

    def __init__(self, device=0, streams=None):
        """`device`: a GPU id, or a list of ids (one context over several GPUs).  `streams` (single GPU only): how many contexts share
        the GPU, None = default_streams() of the first world rendered; asking for `engine.context` before that settles for one."""
        self._device, self._streams, self._context = device, streams, None
        self._tree = TREE_AUTO   # the hosts' default: the snapshot's trees for scenes staged in LDS, the device's surface-area trees otherwise
        self.backend = HostBackend(_lib.load())
        self._world_key = None
        self._camera_key, self._camera_ids = {}, None

    @property
    def context(self):
        if self._context is None:
            self._context = Context(self._device)
            self._context.set_tree(self._tree)
        return self._context

    def set_tree(self, tree):
        """Context.set_tree for the engine's context (default TREE_AUTO); takes effect at the next scene upload, which this forces."""
        self._tree, self._world_key = tree, None
        if self._context is not None:
            self._context.set_tree(tree)

    def set_mode(self, compat_flags):
        """Context.set_mode for the engine's context (COMPAT_REPROJECTION works over several streams / devices too: the context assembles
        the whole previous frame at a restart)."""
        self._mode = compat_flags
        if self._context is not None:
            self._context.set_mode(compat_flags)

    def renderWorld(self, world, render_config, block=True, sync=True):
        if self._context is None and not isinstance(self._device, (list, tuple)):
            k = self._streams or default_streams(len(world.spot_lights) + len(world.direct_lights))
            self._context = Context([self._device] * k) if k > 1 else Context(self._device)
            self._context.set_tree(self._tree)
            if getattr(self, "_mode", 0):
                self._context.set_mode(self._mode)
        ctx = self.context
        # the backend re-mirrors what changed and restarts accumulation then (cpu_engine_renderer.cpp:108-112)
        world_key = getattr(world, "_version", None), id(world)
        if getattr(world, "_moved", False):  # World.mark_moved(): an animation frame
            world._moved = False
            moved = None
            if self._world_key == world_key and not getattr(world, "_dirty", True) and ctx.tree() in (TREE_DEVICE, TREE_DEVICE_SAH) \
                    and len(world.instances) == len(self._flat.instances):
                moved = flatten_motion(world, self._flat.tris["source_index"], self.backend)
            if moved is None:
                world._dirty = True       # no device trees (or another topology): an ordinary modification
            else:                         # the device refits its trees and rebuilds the world tree; nothing is built on the host
                tris, attrs, instances = moved
                if len(tris):
                    ctx.update_triangles(0, tris, attrs)
                if len(instances):
                    ctx.update_instances(instances)
                # a refitted tree keeps the topology it was built with: every REBUILD_EVERY-th moved frame the device builds the trees
                # again over the vertices it holds
                self._moved_frames = getattr(self, "_moved_frames", 0) + 1
                if self._moved_frames % self.REBUILD_EVERY == 0:
                    ctx.rebuild_trees(ctx.tree())
        if self._world_key != world_key or getattr(world, "_dirty", True):
            self._flat = flatten(world, self.backend)
            self._moved_frames = 0
            ctx.upload_scene(self._flat)
            self._world_key = world_key
            world._dirty = False
        ctx.set_config(render_config.struct())
        cameras = [c for c in [world.camera] + list(getattr(world, "cameras", [])) if getattr(c, "enabled", True)]
        if [id(c) for c in cameras] != self._camera_ids:  # one frame state per enabled camera, in the reference's order
            ctx.set_camera_count(max(len(cameras), 1))
            self._camera_ids, self._camera_key = [id(c) for c in cameras], {}
        for k, cam in enumerate(cameras):
            ctx.select_camera(k)
            cam_key = (cam.width, cam.height, cam.position.tobytes(), cam.rotation.tobytes(), cam.fov, cam.near_far,
                       cam.focal_distance, cam.aperture, cam.exposure_time)
            if self._camera_key.get(k) != cam_key:
                ctx.upload_camera(camera_struct(cam, self.backend))
                ctx.set_temporal_blend(cam.temporal_blend)
                self._camera_key[k] = cam_key
            ctx.render(max(render_config.tracing.rpp, 1))
            ctx.tonemap()
            cam.image_buffer = ctx.read_rgba8()  # synchronises
            cam.depth_buffer = ctx.read_depth()
            cam.ray_count = ctx.ray_count()
            # Kernel::rayCast after every frame (cpu_engine_renderer.cpp:176): what the camera's ray-cast pixel looks at
            px, py = getattr(cam, "ray_cast_pixel", (0, 0))
            inst, slot, _, _ = ctx.ray_cast(int(px), int(py))
            cam.raycasted_instance = world.instances[inst] if 0 <= inst < len(world.instances) else None
            materials = getattr(cam.raycasted_instance, "materials", None) or []
            cam.raycasted_material = materials[slot] if cam.raycasted_instance is not None and 0 <= slot < len(materials) else None

    def timingsString(self):
        return self.context.timings()
