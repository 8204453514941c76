"""ctypes binding of the CPU oracle (oracle/librz_oracle.so) — TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
rayzath_amd package.
"""
import ctypes as C
import os

import numpy as np

from rayzath_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "librz_oracle.so")


class _Ctx(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("image", C.POINTER(C.c_float)),
                ("path_depth", C.POINTER(C.c_uint8)), ("ray_origin", C.POINTER(C.c_float)),
                ("ray_direction", C.POINTER(C.c_float)), ("ray_material", C.POINTER(C.c_uint32)),
                ("ray_color", C.POINTER(C.c_float)), ("depth", C.POINTER(C.c_float)),
                ("rgba8", C.POINTER(C.c_uint8)), ("passes", C.c_uint32), ("traced_rays", C.c_uint64)]


_lib = None


def load(path=LIB_PATH):
    global _lib
    if _lib is not None and path == LIB_PATH:
        return _lib
    lib = C.CDLL(path)
    P, U32, F = C.c_void_p, C.c_uint32, C.c_float
    lib.rzo_context_create.restype, lib.rzo_context_create.argtypes = C.POINTER(_Ctx), [U32, U32]
    lib.rzo_context_destroy.restype, lib.rzo_context_destroy.argtypes = None, [C.POINTER(_Ctx)]
    lib.rzo_context_reset.restype, lib.rzo_context_reset.argtypes = None, [C.POINTER(_Ctx)]
    lib.rzo_render_pass.restype = None
    lib.rzo_render_pass.argtypes = [C.POINTER(_abi.Scene), C.POINTER(_abi.Camera), C.POINTER(_abi.Config), C.POINTER(_Ctx),
                                    C.c_int, C.POINTER(_abi.Counters)]
    lib.rzo_pick.restype = None
    lib.rzo_pick.argtypes = [C.POINTER(_abi.Scene), C.POINTER(_abi.Camera), C.POINTER(_Ctx), U32, U32,
                             C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.rzo_seed_value.restype, lib.rzo_seed_value.argtypes = F, [U32, U32, U32]
    lib.rzo_rng_sequence.restype, lib.rzo_rng_sequence.argtypes = None, [F, F, F, U32, P]
    lib.rzo_box_test.restype, lib.rzo_box_test.argtypes = C.c_int, [P, P, P, P, F, F]
    lib.rzo_triangle_test.restype, lib.rzo_triangle_test.argtypes = C.c_int, [P, P, P, P, P, F, F, P]
    lib.rzo_fresnel.restype, lib.rzo_fresnel.argtypes = F, [P, P, F, F, P]
    lib.rzo_cosine_sample_hemisphere.restype, lib.rzo_cosine_sample_hemisphere.argtypes = None, [F, F, P, P]
    lib.rzo_sample_sphere.restype, lib.rzo_sample_sphere.argtypes = None, [F, F, P, P]
    lib.rzo_sample_disk.restype, lib.rzo_sample_disk.argtypes = None, [F, F, P, F, P]
    lib.rzo_tonemap_pixel.restype, lib.rzo_tonemap_pixel.argtypes = None, [P, F, F, P]
    lib.rzo_math_mode.restype, lib.rzo_math_mode.argtypes = C.c_char_p, []
    if path == LIB_PATH:
        _lib = lib
    return lib


class OracleRenderer:
    """CPU::Renderer + CPU::Kernel of the reference, restated (oracle/rz_oracle.c)."""

    def __init__(self, flat_scene, camera, config, lib=None):
        self.lib = lib or load()
        self.scene, self.camera, self.config = flat_scene, camera, config
        self.ctx = self.lib.rzo_context_create(camera.width, camera.height)
        self.w, self.h = camera.width, camera.height

    def close(self):
        if self.ctx:
            self.lib.rzo_context_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.lib.rzo_context_reset(self.ctx)

    def render(self, n_passes=1, threads=0, counted=False):
        total = {n: 0 for n, _ in _abi.Counters._fields_}
        cnt = _abi.Counters()
        for _ in range(n_passes):
            self.lib.rzo_render_pass(C.byref(self.scene.struct), C.byref(self.camera), C.byref(self.config), self.ctx,
                                     threads, C.byref(cnt) if counted else None)
            if counted:
                for k, v in cnt.as_dict().items():
                    total[k] += v
        return total

    def _arr(self, ptr, shape, dtype):
        n = int(np.prod(shape))
        return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).reshape(shape).copy()

    @property
    def accum(self):
        return self._arr(self.ctx.contents.image, (self.h, self.w, 4), np.float32)

    @property
    def depth(self):
        return self._arr(self.ctx.contents.depth, (self.h, self.w), np.float32)

    @property
    def rgba8(self):
        return self._arr(self.ctx.contents.rgba8, (self.h, self.w, 4), np.uint8)

    @property
    def state(self):
        c = self.ctx.contents
        return dict(origin=self._arr(c.ray_origin, (self.h, self.w, 3), np.float32),
                    direction=self._arr(c.ray_direction, (self.h, self.w, 3), np.float32),
                    color=self._arr(c.ray_color, (self.h, self.w, 4), np.float32)[..., :3],
                    material=self._arr(c.ray_material, (self.h, self.w), np.uint32),
                    depth=self._arr(c.path_depth, (self.h, self.w), np.uint8).astype(np.uint32))

    @property
    def passes(self):
        return int(self.ctx.contents.passes)

    @property
    def traced_rays(self):
        return int(self.ctx.contents.traced_rays)

    def pick(self, x, y):
        i, m = C.c_int32(), C.c_int32()
        self.lib.rzo_pick(C.byref(self.scene.struct), C.byref(self.camera), self.ctx, x, y, C.byref(i), C.byref(m))
        return i.value, m.value
