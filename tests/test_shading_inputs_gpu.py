"""GPU side of tests/test_shading_inputs.py: the HIP kernels against the CPU oracle on `shading_inputs_scene` — metalness map,
emission map (R32F), texture transforms, texture alpha, emissive + textured sky, default-material slots — in every
packaging of the pass (resident / fused / split; LDS-staged scene and global scene; with and without lights, i.e. the general
and the no-NEE instantiations), light samples (1,1) and (3,2).

Bars: first-hit depth, first-pass work counters (walk + shading), finished-path counts bit-exact; radiance within
rel 1e-3 for the stated fraction of pixels (measured value minus 0.2 points; the only arithmetic that differs is glibc-vs-ocml
atan2f / asinf of the sky texcrd and the samplers' sinf / cosf / powf / acosf, which can move a texel or an edge).
"""
import numpy as np
import pytest

import oracle
from rayzath_amd import scenes
from rayzath_amd.engine import Context, LightSampling, RenderConfig, Tracing
from rayzath_amd.scene import camera_struct, flatten

pytestmark = pytest.mark.gpu

W, H = 160, 96
PACKAGINGS = {
    "default": {},
    "split-global": dict(traversal_mode=3, lds_scene=0),
    "split-lds": dict(pipeline=1),
    "fused": dict(pipeline=0),
}


def _close(a, b):
    return (np.abs(a - b) <= 1e-3 * np.maximum(np.abs(b), 1.0)).all(-1)


@pytest.mark.parametrize("packaging", list(PACKAGINGS))
@pytest.mark.parametrize("lights,samples", [(True, (1, 1)), (True, (3, 2)), (False, (1, 1))])
def test_shading_inputs_against_the_oracle(built, packaging, lights, samples):
    world = scenes.shading_inputs_scene(W, H, lights=lights)
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(LightSampling(*samples), Tracing(6, 8)).struct()
    ctx = Context(0)
    for k, v in PACKAGINGS[packaging].items():
        getattr(ctx, "set_" + k)(v)
    ctx.upload_scene(flat), ctx.upload_camera(cam), ctx.set_config(cfg)
    ref = oracle.OracleRenderer(flat, cam, cfg)

    first, ref_first = ctx.render_counted(1), ref.render(1, counted=True)
    assert np.array_equal(ctx.read_depth(), ref.depth)
    for k in ("segments", "hits", "light_samples", "finished"):
        assert first[k] == ref_first[k], k
    for total, shadow in (("box_tests", "shadow_box_tests"), ("tri_tests", "shadow_tri_tests")):
        assert first[total] - first[shadow] == ref_first[total] - ref_first[shadow], total
    # one fetch per map per segment whatever the texel: the COUNT is exact even where an ulp of atan2f moves the texel
    assert first["texel_fetches"] == ref_first["texel_fetches"] and first["texel_fetches"] > 0
    if lights:
        assert abs(first["shadow_rays"] - ref_first["shadow_rays"]) <= 2
    else:
        assert first["shadow_rays"] == 0 == first["light_samples"]
    # first pass: without lights the accumulator holds the emission met by the primary ray (emission maps, sky) and nothing else
    # — no sampling routine involved; with lights the first hit's next-event estimation (expf, cosf, acosf) adds ulps
    acc, racc = ctx.read_accum(), ref.accum
    primary_exact = (acc == racc).all(-1).mean() if not lights else _close(acc[..., :3], racc[..., :3]).mean()
    assert np.array_equal(acc[..., 3], racc[..., 3])

    ctx.render(7), ref.render(7)
    acc, racc = ctx.read_accum(), ref.accum
    st, rst = ctx.read_state(), ref.state
    report = dict(primary_exact=float(primary_exact), alpha_equal=float((acc[..., 3] == racc[..., 3]).mean()),
                  rgb_close=float(_close(acc[..., :3], racc[..., :3]).mean()), rgb_exact=float((acc[..., :3] == racc[..., :3]).all(-1).mean()),
                  material_equal=float((st["material"] == rst["material"]).mean()), depth_equal=float((st["depth"] == rst["depth"]).mean()))
    print(f"shading inputs [{packaging}, lights={lights}, samples={samples}]", report)
    assert not np.isnan(acc).any()
    assert report["primary_exact"] >= THRESHOLDS["primary_exact"]
    assert report["alpha_equal"] >= THRESHOLDS["alpha_equal"] and report["material_equal"] >= THRESHOLDS["alpha_equal"]
    assert report["rgb_close"] >= (THRESHOLDS["rgb_close_lit"] if lights else THRESHOLDS["rgb_close_dark"])
    assert ctx.ray_count() == ref.traced_rays == 8 * W * H
    ctx.tonemap()
    img = ctx.read_rgba8()
    assert (np.abs(img.astype(int) - ref.rgba8.astype(int)).max(-1) <= 1).mean() >= THRESHOLDS["rgba8"]


# measured on MI355X (round 2) minus 0.2 points; see the printed reports
# round 2, every packaging: alpha / material / depth equal 1.0, radiance within 1e-3 on 1.0 of the pixels (0.94 bit-exact with
# lights, 1.0 bit-exact without), first pass 1.0
THRESHOLDS = dict(primary_exact=0.998, alpha_equal=0.998, rgb_close_lit=0.998, rgb_close_dark=0.998, rgba8=0.998)


def test_all_packagings_give_the_same_frame(built):
    """resident == fused == split on this scene too, bit for bit (every map fetch and the sky included)."""
    world = scenes.shading_inputs_scene(W, H)
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(LightSampling(2, 2), Tracing(6, 8)).struct()
    frames = []
    for name, settings in PACKAGINGS.items():
        ctx = Context(0)
        for k, v in settings.items():
            getattr(ctx, "set_" + k)(v)
        ctx.upload_scene(flat), ctx.upload_camera(cam), ctx.set_config(cfg)
        ctx.render(1), ctx.render(8)
        frames.append((name, ctx.read_accum(), ctx.read_state()))
        ctx.close()
    for name, acc, st in frames[1:]:
        assert np.array_equal(acc, frames[0][1]), name
        for k in st:
            assert np.array_equal(st[k], frames[0][2][k]), (name, k)
