"""The oracle reproduces the committed end-to-end fixtures (tests/golden/*.npz, written by
tests/golden/make_golden.py): integer outputs bit for bit, float outputs bit for bit on the same
libm (same image), and independent of the thread count."""
import os

import numpy as np
import pytest

import oracle
from rayzath_amd import _abi
from rayzath_amd.scene import FlatScene

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ["cornell_128", "living_room_96x64", "sphere_160x90", "textured_80x48"]


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    flat = FlatScene.from_npz_dict({k[6:]: g[k] for k in g.files if k.startswith("scene_")})
    cam = _abi.Camera.from_buffer_copy(g["camera"].tobytes())
    cfg = _abi.Config.from_buffer_copy(g["config"].tobytes())
    return g, flat, cam, cfg


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(built, name):
    g, flat, cam, cfg = load_golden(name)
    passes = int(g["passes"])
    ref = oracle.OracleRenderer(flat, cam, cfg)
    first = ref.render(1, threads=0, counted=True)        # all hardware threads: result must not depend on them
    assert np.array_equal(ref.depth, g["depth"])
    assert list(first.values()) == g["counters_first"].tolist()
    rest = ref.render(passes - 1, threads=3, counted=True)
    assert list(rest.values()) == g["counters_rest"].tolist()
    assert np.array_equal(ref.accum[..., 3], g["accum"][..., 3])
    assert np.array_equal(ref.state["depth"], g["path_depth"]) and np.array_equal(ref.state["material"], g["ray_material"])
    assert np.array_equal(ref.accum, g["accum"]) and np.array_equal(ref.rgba8, g["rgba8"])
    assert ref.traced_rays == passes * cam.width * cam.height and ref.passes == passes


def test_golden_counters_are_consistent():
    for name in NAMES:
        g, flat, cam, cfg = load_golden(name)
        seg, box, tri, hits, shadow, lights, texels, finished, shadow_box, shadow_tri = (int(x) for x in g["counters_first"] + g["counters_rest"])
        assert shadow_box <= box and shadow_tri <= tri and (shadow_box > 0) == (shadow > 0)
        px = cam.width * cam.height
        assert seg == int(g["passes"]) * px
        assert finished == int(g["accum"][..., 3].sum())          # alpha counts finished paths
        assert hits <= seg and box >= seg * (1 if len(flat.instances) else 0)
        if len(flat.spot_lights) + len(flat.direct_lights) == 0:
            assert shadow == lights == 0
        else:
            assert 0 < shadow <= lights
        assert np.isfinite(g["accum"]).all() and (g["accum"][..., 3] >= 0).all()


def test_reset_restarts_accumulation(built):
    g, flat, cam, cfg = load_golden("cornell_128")
    ref = oracle.OracleRenderer(flat, cam, cfg)
    ref.render(3, threads=2)
    ref.reset()
    assert ref.passes == 0 and ref.traced_rays == 0 and not ref.accum.any()
    ref.render(int(g["passes"]), threads=2)
    assert np.array_equal(ref.accum, g["accum"])
