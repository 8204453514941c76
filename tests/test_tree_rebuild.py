"""hiprz_rebuild_mesh_trees (opt-in SAH mesh trees) on the host: the rebuilt snapshot is a valid scene over the SAME
triangles, every leaf holds them in ascending reference order, every box encloses what is below it, and the
surface-area cost of the tree is lower than the reference builder's."""
import ctypes as C

import numpy as np

from rayzath_amd import _abi, _lib, scenes
from rayzath_amd.scene import FlatScene, flatten

LEAF, COUNT_MASK = _abi.NODE_LEAF, 0x1FFFFFFF


def rebuild(flat, tree=1):
    lib = _lib.load()
    max_nodes = len(flat.nodes) + 2 * len(flat.tris) + len(flat.instances) + 1
    nodes = np.zeros(max_nodes, dtype=_abi.node_dtype)
    order = np.zeros(max(len(flat.tris), 1), dtype=np.uint32)
    roots = np.zeros(max(len(flat.instances), 1), dtype=np.uint32)
    n, tlas = C.c_uint32(), C.c_uint32()
    rc = lib.hiprz_rebuild_mesh_trees(C.byref(flat.struct), tree, nodes.ctypes.data, max_nodes, C.byref(n), order.ctypes.data, roots.ctypes.data, C.byref(tlas))
    assert rc == 0
    return nodes[:n.value].copy(), order[:len(flat.tris)], roots[:len(flat.instances)], tlas.value


def sah_cost(nodes, root):
    """sum over the nodes of area / root area, leaves weighted by their triangle count"""
    def area(n):
        d = n["bb_max"] - n["bb_min"]
        return float(d[0] * d[1] + d[1] * d[2] + d[2] * d[0])
    root_area, cost, stack = area(nodes[root]), 0.0, [root]
    while stack:
        n = nodes[stack.pop()]
        if n["meta"] & LEAF:
            cost += area(n) / root_area * int(n["meta"] & COUNT_MASK)
        else:
            cost += 1.2 * area(n) / root_area
            stack += [int(n["begin"]), int(n["begin"]) + 1]
    return cost


def test_rebuilt_scene_is_valid_and_cheaper():
    for world in (scenes.cornell_sphere(64, 48, 40), scenes.living_room(64, 48, 12), scenes.textured_sphere_scene(64, 48, resolution=60, map_size=16)):
        flat = flatten(world)
        nodes, order, roots, tlas = rebuild(flat)
        assert sorted(order.tolist()) == list(range(len(flat.tris)))          # a permutation: the same triangles
        tris = flat.tris[order]
        inst = flat.instances.copy()
        inst["blas_root"] = roots
        new = FlatScene(nodes=nodes, tlas_root=tlas, tlas_order=flat.tlas_order, tris=tris, tri_attrs=flat.tri_attrs[order], instances=inst,
                        inst_materials=flat.inst_materials, materials=flat.materials, textures=flat.textures, texels=flat.texels,
                        spot_lights=flat.spot_lights, direct_lights=flat.direct_lights)
        msg = C.create_string_buffer(256)
        assert _lib.load().hiprz_validate_scene(C.byref(new.struct), msg, 256) == 0, msg.value
        for i in range(len(flat.instances)):
            old_root, new_root = int(flat.instances[i]["blas_root"]), int(roots[i])
            # every box encloses its subtree; leaves: ascending reference position
            stack, seen = [new_root], 0
            while stack:
                k = stack.pop()
                n = nodes[k]
                if n["meta"] & LEAF:
                    b, c = int(n["begin"]), int(n["meta"] & COUNT_MASK)
                    seen += c
                    ref = order[b:b + c]
                    assert (np.diff(ref.astype(np.int64)) > 0).all()
                    v = np.concatenate([tris["v1"][b:b + c], tris["v2"][b:b + c], tris["v3"][b:b + c]])
                    assert (v >= n["bb_min"]).all() and (v <= n["bb_max"]).all()
                else:
                    for ch in (int(n["begin"]), int(n["begin"]) + 1):
                        assert (nodes[ch]["bb_min"] >= n["bb_min"]).all() and (nodes[ch]["bb_max"] <= n["bb_max"]).all()
                        stack.append(ch)
            assert seen > 0
            if seen > 64:
                assert sah_cost(nodes, new_root) < sah_cost(flat.nodes, old_root), i


def test_instances_without_a_mesh_are_not_meshes():
    """An instance without a mesh is in no leaf of the world tree; the hosts leave its blas_root field 0 — a node of the WORLD tree, which a
    rebuild once took for a mesh root (and refused the scene: "leaves of a mesh must tile one range").  Both placeholder kinds of the
    rebuild must leave such instances alone (root = none) and give a valid scene."""
    from rayzath_amd.scene import Instance
    world = scenes.living_room(64, 48, 10)
    world.add(Instance(None, [], position=(0.3, 0.4, 0.5), name="no mesh"))
    world.instances.insert(2, Instance(None, [], position=(-0.3, 0.2, 0.1), name="no mesh either"))
    flat = flatten(world)
    meshless = [i for i, inst in enumerate(world.instances) if inst.mesh is None]
    assert len(meshless) == 2 and not set(meshless) & set(flat.tlas_order.tolist())
    for tree in (1, 2):
        nodes, order, roots, tlas = rebuild(flat, tree)
        assert all(int(roots[i]) == 0xFFFFFFFF for i in meshless)
        assert sorted(order.tolist()) == list(range(len(flat.tris)))
        inst = flat.instances.copy()
        inst["blas_root"] = roots
        new = FlatScene(nodes=nodes, tlas_root=tlas, tlas_order=flat.tlas_order, tris=flat.tris[order], tri_attrs=flat.tri_attrs[order], instances=inst,
                        inst_materials=flat.inst_materials, materials=flat.materials, textures=flat.textures, texels=flat.texels,
                        spot_lights=flat.spot_lights, direct_lights=flat.direct_lights)
        msg = C.create_string_buffer(256)
        assert _lib.load().hiprz_validate_scene(C.byref(new.struct), msg, 256) == 0, msg.value
