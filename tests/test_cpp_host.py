"""The C++ host side (rayzath_amd/csrc/hip_engine.{hpp,cpp}: RayZath::Hip::Engine, the class the
facade would own next to CPU::Engine / Cuda::Engine) against the Python host side on the same scene:
flattening on CPU, renderWorld on the GPU (sync and pipelined calls, error contract)."""
import math
import os
import struct
import subprocess

import numpy as np
import pytest

from rayzath_amd import _abi
from rayzath_amd.scene import Camera, Instance, Material, Mesh, World, camera_struct, flatten, generate_cube

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "rayzath_amd", "csrc", "hip_engine_test")


def read_dump(path):
    out, data = {}, open(path, "rb").read()
    i = 0
    while i < len(data):
        name = data[i:i + 16].split(b"\0")[0].decode()
        (n,) = struct.unpack_from("<Q", data, i + 16)
        out[name] = data[i + 24:i + 24 + n]
        i += 24 + n
    return out


def python_twin(width=96, height=64):
    """The scene hip_engine_test.cpp builds, through the Python scene model."""
    w = World()
    white = w.add(Material((230, 230, 230, 255), 0, 1)); red = w.add(Material((200, 40, 40, 255), 0, 1))
    green = w.add(Material((40, 200, 40, 255), 0, 1)); light = w.add(Material((255, 255, 255, 255), 0, 1, emission=50))
    mirror = w.add(Material((0xF0, 0xF0, 0xF0, 0xFF), 0.9, 0, 0, 1.0))

    def quad(v):
        return Mesh(v, [(0, 2, 1), (0, 3, 2)], texcrds=[(0, 0), (0, 1), (1, 1), (1, 0)], tri_texcrds=[(0, 2, 1), (0, 3, 2)])
    floor = quad([(-2, 0, -2), (-2, 0, 2), (2, 0, 2), (2, 0, -2)])
    cube = generate_cube()
    w.add(Instance(floor, [white], position=(0, -1, 0)))
    w.add(Instance(floor, [white], position=(0, 3, 0)))
    w.add(Instance(quad([(-2, -1, 2), (-2, 3, 2), (2, 3, 2), (2, -1, 2)]), [white]))
    w.add(Instance(quad([(-2, -1, -2), (-2, 3, -2), (-2, 3, 2), (-2, -1, 2)]), [red]))
    w.add(Instance(quad([(2, -1, -2), (2, 3, -2), (2, 3, 2), (2, -1, 2)]), [green]))
    w.add(Instance(quad([(-.5, 0, -.5), (-.5, 0, .5), (.5, 0, .5), (.5, 0, -.5)]), [light], position=(0, 2.99, 0)))
    w.add(Instance(cube, [mirror], position=(-0.7, 0.2, 0.6), rotation=(0, 0.3, 0), scale=(1.2, 2.4, 1.2)))
    w.add(Instance(cube, [white], position=(0.7, -0.4, -0.5), rotation=(0, -0.3, 0), scale=(1.2, 1.2, 1.2)))
    w.camera = Camera(position=(0, 1, -3.5), resolution=(width, height), fov=1.57079632679, focal_distance=4.0)
    return w


def test_cpp_flatten_equals_python_flatten(built, tmp_path):
    out = str(tmp_path / "flat.bin")
    subprocess.run([EXE, "flatten", out], check=True)
    d = read_dump(out)
    world = python_twin()
    flat = flatten(world)
    for k in ("nodes", "tlas_order", "tris", "tri_attrs", "instances", "inst_materials", "materials"):
        assert d[k] == getattr(flat, k).tobytes(), k
    assert d["camera"] == bytes(camera_struct(world.camera))


@pytest.mark.gpu
def test_cpp_engine_renders_like_the_python_host(built, tmp_path):
    from rayzath_amd.engine import Context, RenderConfig, Tracing
    out = str(tmp_path / "render.bin")
    calls = 3
    proc = subprocess.run([EXE, "render", out, str(calls)], check=True, capture_output=True, text=True)
    assert "render" in proc.stdout          # timingsString()
    d = read_dump(out)
    world = python_twin()
    ctx = Context(0)
    ctx.upload_scene(flatten(world)), ctx.upload_camera(camera_struct(world.camera))
    ctx.set_config(RenderConfig(tracing=Tracing(4, 3)).struct())
    ctx.render(3 * (calls + 1))
    ctx.tonemap()
    assert np.frombuffer(d["accum"], np.float32).tobytes() == ctx.read_accum().tobytes()
    assert d["image"] == ctx.read_rgba8().tobytes() and d["depth"] == ctx.read_depth().tobytes()
    assert struct.unpack("<Q", d["ray_count"])[0] == ctx.ray_count() == 3 * (calls + 1) * 96 * 64
    assert d["threw"] == b"\x01"            # broken world -> Hip::Exception(HIPRZ_ERR_INVALID), no device fault


@pytest.mark.gpu
def test_cpp_engine_refits_a_moved_world_on_the_device(built):
    """World::makeMoved() — vertices and transformations changed, nothing else: with device-built trees Hip::Engine calls
    hiprz_update_triangles / hiprz_update_instances (no host-side tree build, as the reference does at every change) and renders the
    frame a fresh engine renders from the moved world; on host trees the same call is an ordinary modification."""
    proc = subprocess.run([EXE, "moved", "-"], capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0 and "moved frame equal, refitted on the device, changed yes, on host trees equal" in proc.stdout, proc.stdout + proc.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_cpp_engines_agree_through_a_random_sequence_of_changes(built, seed):
    """Hip::Engine with device trees (moved frames through the refit, materials / lights in place, a rebuild now and then) against
    Hip::Engine on the snapshot's trees with a full upload at every change, twin worlds, 24 random steps: images, depths, ray counts."""
    proc = subprocess.run([EXE, "sequence", "-", str(seed)], capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0 and "sequence equal" in proc.stdout and "DIFFERENT" not in proc.stdout, proc.stdout + proc.stderr


@pytest.mark.gpu
def test_cpp_engine_reuploads_when_a_material_is_repointed_at_another_uploaded_map(built):
    """World::makeShadingModified() after two materials swapped their (already uploaded) textures: the map indices of the in-place path
    are positions in the uploaded texture list, so the engine has to notice that the first-use order changed and upload the scene again."""
    proc = subprocess.run([EXE, "swapmaps", "-"], capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0 and "swapped maps equal, colour change equal" in proc.stdout, proc.stdout + proc.stderr
