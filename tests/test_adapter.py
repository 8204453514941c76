"""rayzath_amd/csrc/rayzath_adapter.hpp — the mirror of a real `RayZath::Engine::World` into the backend's snapshot — executed against
the test double of tests/adapter_double.hpp (the reference's host library cannot be compiled in this image): the adapter's snapshot of
a scene with a deep mesh, all five map kinds, lights, groups and unset material slots equals Hip::flatten() of the same scene byte for
byte, and the reference's dirty flags (updatable.cpp:23-51) select nothing / a shading update / a full refresh as the CUDA backend's
`reconstruct` would (cuda_world.cu:69-75) — plus, where the context holds device-built trees and only vertices / transformations moved,
the records for a device-side refit in the uploaded order (Change::Moved)."""
import os
import subprocess

import pytest

from rayzath_amd import scenes
from rayzath_amd.scene import Group
from rayzath_amd.scene_io import save_scene_json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rayzath_amd", "csrc")


def _scene_and_exe(tmp_path):
    world = scenes.shading_inputs_scene(96, 64)
    inner = world.add(Group(position=(0.0, 0.3, 0.0), rotation=(0.1, 0.0, 0.2), objects=[world.instances[1]], name="inner"))
    world.add(Group(position=(0.2, 0.1, -0.3), rotation=(0.0, 0.4, 0.0), scale=(1.1, 1.0, 0.9), objects=[world.instances[5]], groups=[inner], name="outer"))
    for inst in world.instances:          # the scene file names its materials: unset trailing slots are simply left out
        while inst.materials and inst.materials[-1] is None:
            inst.materials.pop()
    scene = str(tmp_path / "scene.json")
    save_scene_json(world, scene)
    exe = str(tmp_path / "adapter_check")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-I", os.path.join(ROOT, "tests"),
                    os.path.join(ROOT, "tests", "adapter_check.cpp"), "-o", exe, "-L", CSRC, "-lhiprz_host", "-lhiprz",
                    "-Wl,-rpath," + CSRC], check=True)
    return scene, exe


def test_adapter_mirrors_the_world_like_flatten(built, tmp_path):
    scene, exe = _scene_and_exe(tmp_path)
    r = subprocess.run([exe, scene], capture_output=True, text=True)
    assert r.returncode == 0 and "ADAPTER OK" in r.stdout and "DIFFERENT" not in r.stdout, r.stdout + r.stderr
    # the scene really had what the docstring says
    for what in ("nodes", "texels", "spot_lights", "inst_materials"):
        assert any(line.startswith(what) and "(0 records)" not in line for line in r.stdout.splitlines()), what


@pytest.mark.gpu
def test_world_renderer_over_the_double_renders_like_the_engine_over_the_twin(built, tmp_path):
    """WorldRenderer (the renderWorld a RayZath build would call, over the adapter) against Hip::Engine on the twin: two cameras, a
    pipelined frame, a material change that goes through hiprz_update_shading — images, depth buffers and ray counts equal."""
    scene, exe = _scene_and_exe(tmp_path)
    r = subprocess.run([exe, scene, "render"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ADAPTER OK" in r.stdout and "DIFFERENT" not in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("camera 0 equal") == 4 and r.stdout.count("camera 1 equal") == 4
    assert r.stdout.count("ray cast equal") == 4 and "accumulation went on         yes" in r.stdout
    # ... and a moved world over device-built trees goes through hiprz_update_triangles / hiprz_update_instances (Change::Moved)
    assert "moved world, device trees    frame equal, refitted on the device" in r.stdout
