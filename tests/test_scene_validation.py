"""hiprz_validate_scene: the host check that runs before anything is sent to the GPU.  A wrong
index or a cyclic tree would fault or hang the device, so every such scene must be refused."""
import ctypes as C

import numpy as np
import pytest

from rayzath_amd import _abi, _lib, scenes
from rayzath_amd.scene import FlatScene, flatten


def _validate(flat):
    lib = _lib.load()
    msg = C.create_string_buffer(256)
    rc = lib.hiprz_validate_scene(C.byref(flat.struct), msg, 256)
    return rc, msg.value.decode()


def _mutated(base, **changes):
    d = {k: getattr(base, k).copy() for k in FlatScene.FIELDS}
    for k, fn in changes.items():
        fn(d[k])
    return FlatScene(tlas_root=base.tlas_root, **d)


@pytest.fixture(scope="module")
def base(built):
    return flatten(scenes.living_room(32, 32, n_instances=20))


def test_valid_scenes_pass(base):
    assert _validate(base) == (0, "")
    assert _validate(flatten(scenes.cornell_box(16, 16)))[0] == 0
    assert _validate(flatten(scenes.textured_sphere_scene(16, 16, resolution=16, map_size=32)))[0] == 0


def _set(field, index, value):
    def fn(a):
        a[field][index] = value
    return fn


CASES = {
    "child out of range": dict(nodes=_set("begin", 0, 10 ** 6)),
    "cycle in world tree": dict(nodes=_set("begin", 1, 0)),
    "instance id out of range": dict(tlas_order=lambda a: a.__setitem__(0, 10 ** 6)),
    "mesh root out of range": dict(instances=_set("blas_root", 0, 10 ** 6)),
    "material index out of range": dict(inst_materials=lambda a: a.__setitem__(0, 10 ** 6)),
    "material table out of range": dict(instances=_set("material_base", 0, 10 ** 6)),
    "too many material slots": dict(instances=_set("material_count", 0, 65)),
    "texture index out of range": dict(materials=_set("texture", 2, 5)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_broken_scenes_are_refused(base, name):
    rc, msg = _validate(_mutated(base, **CASES[name]))
    assert rc == _abi.ERR_INVALID and msg, name


def test_leaf_range_and_shared_subtrees(base):
    leaf = int(np.nonzero((base.nodes["meta"] & _abi.NODE_LEAF) != 0)[0][-1])
    rc, msg = _validate(_mutated(base, nodes=_set("begin", leaf, len(base.tris))))
    assert rc == _abi.ERR_INVALID and "leaf range" in msg
    # two inner nodes pointing at the same children = a DAG: refused (the walk would visit it twice)
    inner = np.nonzero((base.nodes["meta"] & _abi.NODE_LEAF) == 0)[0]
    a, b = int(inner[0]), int(inner[1])
    rc, msg = _validate(_mutated(base, nodes=_set("begin", b, int(base.nodes["begin"][a]))))
    assert rc == _abi.ERR_INVALID


def test_missing_world_materials_and_null_arrays(base):
    d = {k: getattr(base, k).copy() for k in FlatScene.FIELDS}
    d["materials"] = d["materials"][:1]
    d["inst_materials"][:] = -1
    assert _validate(FlatScene(tlas_root=0, **d))[0] == _abi.ERR_INVALID
    s = _abi.Scene.from_buffer_copy(base.struct)
    s.tris = None
    lib = _lib.load()
    assert lib.hiprz_validate_scene(C.byref(s), None, 0) == _abi.ERR_INVALID
    assert lib.hiprz_validate_scene(None, None, 0) == _abi.ERR_INVALID


def test_texture_descriptor_checks(built):
    flat = flatten(scenes.textured_sphere_scene(16, 16, resolution=16, map_size=32))
    assert _validate(_mutated(flat, textures=_set("offset", 0, 2)))[0] == _abi.ERR_INVALID          # unaligned
    assert _validate(_mutated(flat, textures=_set("width", 0, 10 ** 5)))[0] == _abi.ERR_INVALID     # beyond the pool
    assert _validate(_mutated(flat, textures=_set("kind", 0, _abi.TEX_R8)))[0] == _abi.ERR_INVALID  # colour map must be RGBA8
