"""The shading inputs of SURVEY.md §8 a13 that BASELINE's stand-in scenes do not touch — metalness map (R8), emission map
(R32F), texture rotation / scale / translation, a texture's alpha channel, an emissive and textured world material (sky),
unset / missing material slots (default material) — rendered by the HIP kernels and by the CPU oracle on the same scene
(`rayzath_amd.scenes.shading_inputs_scene`), every packaging of the pass.

The CPU part proves that the scene really exercises each input: taking one away changes the oracle's frame.
Reference: cpu_engine_kernel.cpp:292-295 (sky texcrd), :359-360 (default material), :505-537 (fetch*),
render_parts.hpp:209-221 (TextureBuffer::fetch).
"""
import numpy as np
import pytest

import oracle
from rayzath_amd import scenes
from rayzath_amd.engine import LightSampling, RenderConfig, Tracing
from rayzath_amd.scene import TextureBuffer, camera_struct, flatten

W, H = 160, 96


def _oracle_frame(world, passes=3, samples=(1, 1)):
    flat, cam = flatten(world), camera_struct(world.camera)
    ref = oracle.OracleRenderer(flat, cam, RenderConfig(LightSampling(*samples), Tracing(5, passes)).struct())
    counters = ref.render(1, counted=True)
    ref.render(passes - 1)
    return ref.accum, counters


def _identity(t):
    return TextureBuffer(t.bitmap)


@pytest.mark.parametrize("what", ["sky emission map", "sky texture", "metalness map", "panel emission map", "texture transform",
                                  "texture alpha", "default material", "roughness map transform"])
def test_every_input_changes_the_oracle_frame(what):
    """Remove one input at a time: the frame must change (else the GPU comparison below would not be looking at it)."""
    full, full_counters = _oracle_frame(scenes.shading_inputs_scene(W, H))
    world = scenes.shading_inputs_scene(W, H)
    by_name = {m.name: m for m in world.materials}
    if what == "sky emission map":
        world.material.emission_map = None
    elif what == "sky texture":
        world.material.texture = None
    elif what == "metalness map":
        by_name["panel"].metalness_map = None
    elif what == "panel emission map":
        by_name["panel"].emission_map = None
    elif what == "texture transform":
        by_name["tiles"].texture = _identity(by_name["tiles"].texture)
    elif what == "roughness map transform":
        by_name["tiles"].roughness_map = _identity(by_name["tiles"].roughness_map)
    elif what == "texture alpha":
        t = by_name["stained glass"].texture
        opaque = t.bitmap.copy()
        opaque[..., 3] = 255
        by_name["stained glass"].texture = TextureBuffer(opaque, t.scale, t.rotation, t.translation)
    elif what == "default material":
        world.default_material.color = (255, 0, 0, 255)
    changed, counters = _oracle_frame(world)
    assert not np.array_equal(full, changed), what
    assert not np.isnan(full).any()
    if what.endswith("map") or what == "sky texture":
        assert counters["texel_fetches"] < full_counters["texel_fetches"]


def test_scene_reaches_the_sky_and_the_transmissive_branch():
    world = scenes.shading_inputs_scene(W, H)
    flat, cam = flatten(world), camera_struct(world.camera)
    ref = oracle.OracleRenderer(flat, cam, RenderConfig(LightSampling(3, 2), Tracing(6, 6)).struct())
    c = ref.render(6, counted=True)
    assert c["hits"] < c["segments"]                     # misses: sky segments
    assert c["light_samples"] > 0 and c["shadow_rays"] > 0
    stained = 2 + [m.name for m in world.materials].index("stained glass")
    assert (ref.state["material"] == stained).any()      # a path is travelling INSIDE the textured-alpha ball
