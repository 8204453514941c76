"""CUDA-compat mode (hiprz_set_mode): behaviours of the reference's CUDA engine that its CPU engine — the parity
oracle — does not have.  They cannot be compared with the oracle, so each is checked against its ANALYTIC expectation on a scene
built for it, next to the same scene in the default (CPU) mode:

  Beer-Lambert            radiance through an absorbing slab = opacityColor * alpha^thickness x the CPU-mode radiance
  medium scattering       fraction of rays scattered before a wall at distance d = 1 - exp(-sigma d) (+1e-4)
  coloured shadows        light under a transparent sheet = unshadowed light * opacityColor.rgb * opacityColor.alpha
  texture x colour        first-pass emission = colour * texture * (emission * emission map)
  filter / address modes  an emission map sampled point / linear under wrap / clamp / mirror / border == a numpy restatement

and the default mode must not notice that the compat code exists (mode 0 after a compat render == a fresh context).
Reference: cuda_render_kernel.cu:146-237, cuda_material.cuh:75-159, cuda_world.cuh:91-100, cuda_instance.cuh:92-164, cuda_buffer.cuh:364-438.
"""
import math

import numpy as np
import pytest

from rayzath_amd import scenes
from rayzath_amd.engine import (COMPAT_BEER_LAMBERT, COMPAT_FILTERING, COMPAT_SCATTERING, COMPAT_SHADOW_COLOR, COMPAT_TEXTURE_MULT,
                                Context, LightSampling, RenderConfig, Tracing)
from rayzath_amd.scene import (Camera, Instance, Material, Mesh, SpotLight, TextureBuffer, World, camera_struct, flatten, generate_cube,
                               generate_plane)

pytestmark = pytest.mark.gpu


def _render(world, flags, passes, max_depth, samples=(1, 1)):
    flat, cam = flatten(world), camera_struct(world.camera)
    ctx = Context(0)
    ctx.set_mode(flags)
    ctx.upload_scene(flat), ctx.upload_camera(cam), ctx.set_config(RenderConfig(LightSampling(*samples), Tracing(max_depth, passes)).struct())
    ctx.render(passes)
    out = ctx.read_accum(), ctx.read_depth(), ctx.read_state()
    ctx.close()
    return out


def _quad(size, z=0.0):
    """Square [-size, size]^2 in the plane z, facing -z (towards a camera on the negative z axis), uv = ((x + size) / 2 size, (y + size) / 2 size)."""
    v = [(-size, -size, z), (size, -size, z), (size, size, z), (-size, size, z)]
    t = [(0, 0), (1, 0), (1, 1), (0, 1)]
    return Mesh(v, [(0, 2, 1), (0, 3, 2)], texcrds=t, tri_texcrds=[(0, 2, 1), (0, 3, 2)], name="quad")


def _narrow_camera(width=64, height=64, fov=0.2, z=-3.0):
    return Camera(position=(0, 0, z), rotation=(0, 0, 0), resolution=(width, height), fov=fov, near_far=(1e-2, 1e3), focal_distance=3.0,
                  aperture=1e-6, exposure_time=1.0 / 60.0)


def test_beer_lambert_through_a_slab():
    world = World()
    glow = world.add(Material((255, 255, 255, 255), 0.0, 1.0, emission=1.0, name="panel"))
    tinted = world.add(Material((200, 150, 100, 128), 0.0, 0.0, 0.0, 1.0, 0.0, name="absorbing glass"))   # ior 1: rays go straight through
    world.add(Instance(world.add(_quad(3.0)), [glow], position=(0, 0, 2.0), name="panel"))
    thickness = 0.5
    world.add(Instance(world.add(generate_cube()), [tinted], rotation=(0.0, 0.0, 0.37), scale=(4.0, 4.0, thickness), name="slab"))  # turned about z: no pixel centre on a face diagonal
    world.camera = _narrow_camera()
    # depth 3 = slab front, slab back, panel: exactly one emission term per path
    cpu, _, _ = _render(world, 0, 6, 3)
    compat, _, _ = _render(world, COMPAT_BEER_LAMBERT, 6, 3)
    assert np.array_equal(cpu[..., 3], compat[..., 3]) and cpu[..., 3].min() >= 1
    opacity = np.array([200, 150, 100]) / 255.0
    alpha = 1.0 - 128 / 255.0
    lit = cpu[..., :3].min(-1) > 0
    assert lit.mean() > 0.99
    ratio = compat[..., :3][lit] / cpu[..., :3][lit]
    for ch in range(3):
        expected = opacity[ch] * alpha ** thickness           # path length inside: thickness / cos(angle), cos >= 0.995 at this fov
        assert abs(ratio[:, ch].mean() / expected - 1.0) < 0.01, (ch, ratio[:, ch].mean(), expected)
        assert ratio[:, ch].std() < 0.01 * expected


def test_medium_scattering_follows_the_exponential_law():
    sigma, distance = 0.5, 4.0
    world = World()
    world.material = Material((255, 255, 255, 0), 0.0, 0.0, 0.0, 1.0, sigma, name="fog")
    wall = world.add(Material((255, 255, 255, 255), 0.0, 1.0, emission=1.0, name="wall"))
    world.add(Instance(world.add(_quad(3.0)), [wall], position=(0, 0, distance - 3.0), name="wall"))
    world.camera = _narrow_camera(128, 128)
    _, depth_cpu, _ = _render(world, 0, 1, 4)
    _, depth, state = _render(world, COMPAT_SCATTERING, 1, 4)
    assert depth_cpu.min() > distance - 1e-3                          # default mode: nothing scatters (cpu_engine_kernel.cpp:539-554 is never called)
    scattered = (depth < distance - 1e-3).mean()
    expected = 1.0 - math.exp(-sigma * distance) + 1e-4
    print(f"scattered {scattered:.4f}, expected {expected:.4f}")
    assert abs(scattered - expected) < 0.03
    # the free paths themselves: exponential with mean 1 / sigma, truncated at the wall
    free = depth[depth < distance - 1e-3]
    trunc_mean = 1 / sigma - distance * math.exp(-sigma * distance) / (1 - math.exp(-sigma * distance))
    assert abs(free.mean() - trunc_mean) < 0.05
    assert (state["material"][depth < distance - 1e-3] == 0).all()    # a scattered ray stays in the medium


def _shadow_scene(with_sheet):
    world = World()
    floor = world.add(Material((255, 255, 255, 255), 0.0, 1.0, name="floor"))
    world.add(Instance(world.add(generate_plane(4, 8.0, 8.0)), [floor], position=(0, -1, 0), name="floor"))
    if with_sheet:
        sheet = world.add(Material((255, 64, 64, 128), 0.0, 0.3, name="red sheet"))
        world.add(Instance(world.add(generate_plane(4, 1.5, 1.5)), [sheet], position=(0, 1.0, 0), name="sheet"))
    world.add(SpotLight(position=(0, 3.0, 0), direction=(0, -1, 0), color=(255, 255, 255, 255), size=0.05, emission=200.0, beam_angle=1.2))
    world.camera = Camera(position=(0, -0.2, -2.5), rotation=(-0.35, 0, 0), resolution=(96, 64), fov=1.0, near_far=(1e-2, 1e3),
                          focal_distance=3.0, aperture=1e-6, exposure_time=1.0 / 60.0)
    return world


@pytest.mark.parametrize("shadow_walk", ["1", "0"])
def test_coloured_shadows(monkeypatch, shadow_walk):
    """... through both shadow walks: by the wave (rz_shadow_packet_kernel<..., MASK>, what big frames get) and lane by lane (rz_shadow_coop_kernel<..., MASK>)."""
    monkeypatch.setenv("HIPRZ_SHADOW_PACKET", shadow_walk)
    open_, _, _ = _render(_shadow_scene(False), COMPAT_SHADOW_COLOR, 1, 1)
    opaque, _, _ = _render(_shadow_scene(True), 0, 1, 1)
    tinted, depth, _ = _render(_shadow_scene(True), COMPAT_SHADOW_COLOR, 1, 1)
    under = (opaque[..., :3].max(-1) == 0) & (open_[..., :3].min(-1) > 0) & (depth < 50)   # floor pixels the sheet shadows completely
    assert under.sum() > 200
    mask = np.array([255, 64, 64]) / 255.0 * (1.0 - 128 / 255.0)          # opacityColor.rgb * opacityColor.alpha (V_PL * V_PL.alpha)
    ratio = tinted[..., :3][under] / open_[..., :3][under]
    assert np.allclose(ratio, mask[None, :], rtol=2e-3), (ratio.mean(0), mask)
    lit = (opaque[..., :3].min(-1) > 0)
    assert np.allclose(tinted[..., :3][lit], opaque[..., :3][lit], rtol=1e-5)  # outside the shadow nothing changes


def _map_panel(emission_map, texture=None, color=(255, 255, 255, 255), emission=1.0):
    world = World()
    m = world.add(Material(color, 0.0, 1.0, emission=emission, texture=texture, emission_map=emission_map, name="panel"))
    world.add(Instance(world.add(_quad(1.0)), [m], name="panel"))
    world.camera = _narrow_camera(96, 96, fov=0.5)
    return world


def test_texture_and_emission_map_multiply():
    rng = np.random.default_rng(5)
    tex = rng.integers(30, 256, size=(8, 8, 4), dtype=np.uint8)
    tex[..., 3] = 255
    em = rng.uniform(0.5, 2.0, size=(4, 4)).astype(np.float32)
    world = _map_panel(TextureBuffer(em), TextureBuffer(tex), color=(128, 255, 64, 255), emission=2.0)
    cpu, depth, _ = _render(world, 0, 1, 2)
    compat, _, _ = _render(world, COMPAT_TEXTURE_MULT, 1, 2)
    on_panel = depth < 50
    assert on_panel.mean() > 0.3
    factor = np.array([128, 255, 64], dtype=np.float32) / np.float32(255) * np.float32(2.0)
    assert np.allclose(compat[..., :3][on_panel], cpu[..., :3][on_panel] * factor[None, :], rtol=1e-5)


def _sample_numpy(bitmap, u, v, scale, filter_mode, address_mode):
    """TextureBuffer::fetch of the CUDA engine on an R32F map without rotation / translation (cuda_buffer.cuh:427-438)."""
    h, w = bitmap.shape
    x, y = u * scale[0], 1.0 - v * scale[1]

    def texel(i, n):
        if address_mode == "clamp":
            return np.clip(i, 0, n - 1), np.ones_like(i, dtype=bool)
        if address_mode == "border":
            return np.clip(i, 0, n - 1), (i >= 0) & (i < n)
        if address_mode == "mirror":
            k = np.mod(i, 2 * n)
            return np.where(k < n, k, 2 * n - 1 - k), np.ones_like(i, dtype=bool)
        return np.mod(i, n), np.ones_like(i, dtype=bool)

    if filter_mode == "point":
        xi, okx = texel(np.floor(x * w).astype(int), w)
        yi, oky = texel(np.floor(y * h).astype(int), h)
        return np.where(okx & oky, bitmap[yi, xi], 0.0)
    fx, fy = x * w - 0.5, y * h - 0.5
    x0, y0 = np.floor(fx).astype(int), np.floor(fy).astype(int)
    ax, ay = fx - x0, fy - y0
    out = np.zeros_like(x)
    for k in range(4):
        xi, okx = texel(x0 + (k & 1), w)
        yi, oky = texel(y0 + (k >> 1), h)
        wgt = np.where(k & 1, ax, 1 - ax) * np.where(k >> 1, ay, 1 - ay)
        out += np.where(okx & oky, bitmap[yi, xi], 0.0) * wgt
    return out


@pytest.mark.parametrize("filter_mode", ["point", "linear"])
@pytest.mark.parametrize("address_mode", ["wrap", "clamp", "mirror", "border"])
def test_filter_and_address_modes(filter_mode, address_mode):
    em = np.array([[1.0, 2.0, 5.0], [3.0, 4.0, 7.0]], dtype=np.float32)
    scale = (2.5, -1.75)
    world = _map_panel(TextureBuffer(em, scale=scale, filter_mode=filter_mode, address_mode=address_mode))
    acc, depth, _ = _render(world, COMPAT_FILTERING, 1, 2)
    cam = world.camera
    px, py = np.meshgrid(np.arange(cam.width), np.arange(cam.height))
    tan = math.tan(cam.fov / 2)
    hit_x = ((px + 0.5) / cam.width - 0.5) * tan * 3.0                      # generateSimpleRay through the plane z = 0 from z = -3
    hit_y = ((py + 0.5) / cam.height - 0.5) * (-tan / (cam.width / cam.height)) * 3.0
    on_panel = depth < 50
    assert np.array_equal(on_panel, (np.abs(hit_x) < 1) & (np.abs(hit_y) < 1))
    expected = _sample_numpy(em.astype(np.float64), (hit_x + 1) / 2, (hit_y + 1) / 2, scale, filter_mode, address_mode)
    got = acc[..., 0]                                                       # white panel: radiance = the sampled emission
    # barycentric interpolation rounds the texcrd: a pixel on a texel boundary may fall on the other side
    ok = np.abs(got - expected) <= 1e-3 * np.maximum(np.abs(expected), 1.0)
    assert ok[on_panel].mean() > (0.97 if filter_mode == "point" else 0.999), ok[on_panel].mean()
    if address_mode == "border":
        assert (expected[on_panel] == 0).mean() > 0.3 and (got[on_panel][expected[on_panel] == 0] == 0).mean() > 0.97
    # the default mode ignores both settings: point sampling with wrap-around (render_parts.hpp:209-221)
    cpu, _, _ = _render(world, 0, 1, 2)
    plain = _map_panel(TextureBuffer(em, scale=scale))
    assert np.array_equal(cpu, _render(plain, 0, 1, 2)[0])


def test_default_mode_is_untouched_by_the_compat_code():
    """mode 0 after compat renders == a context that never left mode 0, bit for bit; and compat with no behaviour that the scene
    can trigger (no lights, opaque, no maps, no scattering medium) renders the Cornell box like the CPU mode within float noise."""
    world = scenes.cornell_box(96, 64)
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(tracing=Tracing(5, 4)).struct()
    a = Context(0)
    a.upload_scene(flat), a.upload_camera(cam), a.set_config(cfg)
    a.set_mode(31), a.render(4)
    compat = a.read_accum()
    a.set_mode(0), a.render(4)
    b = Context(0)
    b.upload_scene(flat), b.upload_camera(cam), b.set_config(cfg)
    b.render(4)
    assert np.array_equal(a.read_accum(), b.read_accum())
    assert np.array_equal(compat[..., 3], b.read_accum()[..., 3])
    assert np.allclose(compat[..., :3], b.read_accum()[..., :3], rtol=1e-4, atol=1e-6)   # Beer in a medium of alpha 1: x * pow(1, t)


# ---- spatio-temporal reprojection (COMPAT_REPROJECTION; cuda_camera.cuh:382-426, cuda_engine_renderer.cu:139-150) ----
def _host_reprojection(first, depth, prev_accum, prev_depth, cam_now, cam_prev, blend):
    """Camera::reproject restated with numpy float32 on row-major frames: returns the accumulator the first pass + history must give and
    the mask of pixels whose source pixel index is safely inside a pixel (not within 1e-3 of a pixel border, where one rounding decides)."""
    F = np.float32
    H, W = depth.shape
    ax = lambda v: np.array(list(v), dtype=F)
    def axes(c):
        return ax(c.x_axis), ax(c.y_axis), ax(c.z_axis)
    xs, ys = np.meshgrid(np.arange(W, dtype=F), np.arange(H, dtype=F))
    tana_now = F(cam_now.tan_half_fov)
    dx = ((xs + F(0.5)) / F(W) - F(0.5)) * tana_now
    dy = ((ys + F(0.5)) / F(H) - F(0.5)) * (-tana_now / F(cam_now.aspect_ratio))
    xa, ya, za = axes(cam_now)
    d = dx[..., None] * xa + dy[..., None] * ya + za
    d = d * (F(1) / np.sqrt((d * d).sum(-1, dtype=F)))[..., None]
    space = ax(cam_now.position) + d * depth[..., None]
    rel = space - ax(cam_prev.position)
    pxa, pya, pza = axes(cam_prev)
    lx, ly, lz = (rel * pxa).sum(-1, dtype=F), (rel * pya).sum(-1, dtype=F), (rel * pza).sum(-1, dtype=F)
    tana = F(cam_prev.tan_half_fov)
    with np.errstate(divide="ignore", invalid="ignore"):
        fx = ((lx / lz) / tana + F(0.5)) * F(W)
        fy = ((ly / lz) / (-tana / F(cam_prev.aspect_ratio)) + F(0.5)) * F(H)
    ok = (lz > 0) & (fx >= 0) & (fx < W) & (fy >= 0) & (fy < H)
    sx, sy = np.where(ok, fx, 0).astype(np.int64), np.where(ok, fy, 0).astype(np.int64)
    dist = np.sqrt((rel * rel).sum(-1, dtype=F))
    ok &= np.abs(dist - prev_depth[sy, sx]) < F(0.01) * dist
    out = first.copy()
    out[ok] = first[ok] + prev_accum[sy, sx][ok] * F(blend)
    safe = (np.abs(fx - np.round(fx)) > 1e-3) & (np.abs(fy - np.round(fy)) > 1e-3) | ~np.isfinite(fx)
    near_threshold = np.abs(np.abs(dist - prev_depth[sy, sx]) - F(0.01) * dist) < 1e-4 * dist
    return out, safe & ~near_threshold, ok


@pytest.mark.parametrize("pipeline", [-1, 1])
def test_reprojection_of_an_unchanged_view_adds_the_blended_history(pipeline):
    """A restart with nothing moved: every pixel finds itself, so the new accumulator is exactly first pass + blend x old accumulator
    (sample counts included) — in the resident and in the split pipeline; without the flag a restart starts from nothing."""
    from rayzath_amd.engine import COMPAT_REPROJECTION
    world = scenes.cornell_sphere(96, 64, 12)
    flat, cam = flatten(world), camera_struct(world.camera)
    cfg = RenderConfig(tracing=Tracing(5, 4)).struct()

    def context(flags):
        c = Context(0)
        c.set_pipeline(pipeline), c.set_mode(flags)
        c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
        return c

    fresh = context(0)
    fresh.render(1)
    first = fresh.read_accum()
    ctx = context(COMPAT_REPROJECTION)
    ctx.set_temporal_blend(0.6)
    ctx.render(1), ctx.render(7)
    old = ctx.read_accum()
    fresh.render(7)
    assert np.array_equal(old, fresh.read_accum())                      # the flag alone changes nothing while a frame accumulates
    ctx.reset()
    ctx.render(1)
    got = ctx.read_accum()
    assert np.array_equal(got, first + old * np.float32(0.6))
    assert got[..., 3].min() >= np.float32(0.6) * old[..., 3].min() and old[..., 3].max() > 0
    ctx.render(3)                                                        # accumulation goes on from there
    fresh2 = context(0)
    fresh2.render(1), fresh2.render(3)
    assert np.allclose(ctx.read_accum() - old * np.float32(0.6), fresh2.read_accum(), rtol=1e-5, atol=1e-4)
    plain = context(0)
    plain.render(8), plain.reset(), plain.render(1)
    assert np.array_equal(plain.read_accum(), first)


@pytest.mark.parametrize("devices", [0, [0, 0], [0, 0, 0]])
def test_reprojection_follows_a_moved_camera(devices):
    """(Also over several shards of one multi-device context: the history is the WHOLE previous frame, assembled from every device, so the
    result is that of one device bit for bit.)  The camera steps sideways and turns a little: history comes from where the surface WAS on the previous screen, and only where
    the previous depth buffer holds the same surface.  Against a numpy restatement of Camera::reproject on the frames read back."""
    from rayzath_amd.engine import COMPAT_REPROJECTION
    world = scenes.cornell_sphere(128, 96, 12)
    flat = flatten(world)
    cam_prev = camera_struct(world.camera)
    moved = Camera(position=tuple(np.asarray(world.camera.position) + np.array([0.25, 0.1, 0.05], dtype=np.float32)),
                   rotation=tuple(np.asarray(world.camera.rotation) + np.array([0.02, -0.06, 0.0], dtype=np.float32)),
                   resolution=(world.camera.width, world.camera.height), fov=world.camera.fov, near_far=world.camera.near_far,
                   focal_distance=world.camera.focal_distance, aperture=world.camera.aperture, exposure_time=world.camera.exposure_time)
    cam_now = camera_struct(moved)
    cfg = RenderConfig(tracing=Tracing(5, 4)).struct()
    ctx = Context(devices)
    ctx.set_mode(COMPAT_REPROJECTION)
    ctx.upload_scene(flat), ctx.upload_camera(cam_prev), ctx.set_config(cfg)
    ctx.render(1), ctx.render(5)
    prev_accum, prev_depth = ctx.read_accum(), ctx.read_depth()
    ctx.upload_camera(cam_now)                                           # restarts accumulation
    ctx.render(1)
    got, depth = ctx.read_accum(), ctx.read_depth()
    fresh = Context(0)
    fresh.upload_scene(flat), fresh.upload_camera(cam_now), fresh.set_config(cfg)
    fresh.render(1)
    first = fresh.read_accum()
    assert np.array_equal(depth, fresh.read_depth())
    want, safe, took = _host_reprojection(first, depth, prev_accum, prev_depth, cam_now, cam_prev, 0.75)
    carried = took.mean()
    print(f"reprojection: history carried over on {carried:.3f} of the pixels, {safe.mean():.3f} decided clear of a rounding")
    assert 0.5 < carried < 0.98                                          # most of the view is still there, the newly uncovered part is not
    close = np.abs(got - want) <= 1e-4 * np.maximum(np.abs(want), 1.0)
    assert close[safe].all(-1).mean() >= 0.999
    assert (got[took][..., 3] > first[took][..., 3]).mean() > 0.9 and (got[~took][..., 3] <= 1.0).all()   # sample counts travel with the colour
    if devices != 0:                                                     # the same frame as ONE device renders, history included
        one = Context(0)
        one.set_mode(COMPAT_REPROJECTION)
        one.upload_scene(flat), one.upload_camera(cam_prev), one.set_config(cfg)
        one.render(1), one.render(5)
        one.upload_camera(cam_now)
        one.render(1)
        assert np.array_equal(one.read_accum(), got)


@pytest.mark.parametrize("flags", [COMPAT_BEER_LAMBERT | COMPAT_SCATTERING, COMPAT_TEXTURE_MULT | COMPAT_FILTERING, 31, 31 & ~COMPAT_SHADOW_COLOR])
def test_compat_integrator_on_the_split_pipeline_equals_the_fused_kernel(flags):
    """The compat integrator runs by default through the packaging of the default mode — rays in sorted order, the cooperative front-to-back
    walk (with the medium's scattering distance drawn before it), shading, and — without the coloured shadow mask — the pass's shadow rays
    deferred to the lean cooperative kernel (with the coloured shadow mask: its mask-collecting instantiation).  hiprz_set_pipeline(0) keeps the
    one fused kernel per pass: the same frames, bit for bit (coloured masks: to rounding, see below)."""
    for world, samples in ((scenes.living_room(96, 64, 16), (2, 1)), (scenes.shading_inputs_scene(96, 64), (1, 2)), (scenes.cornell_box(80, 48), (1, 1))):
        for m in world.materials[:3]:                        # give the media something to scatter and absorb with
            m.scattering = max(m.scattering, 0.0)
        world.material.scattering = 0.15                     # the room's air scatters: every segment draws a distance first
        flat, cam = flatten(world), camera_struct(world.camera)
        cfg = RenderConfig(LightSampling(*samples), Tracing(5, 4)).struct()
        out = []
        for pipeline in (0, -1):
            c = Context(0)
            c.set_mode(flags), c.set_pipeline(pipeline)
            c.upload_scene(flat), c.upload_camera(cam), c.set_config(cfg)
            c.render(1), c.render(5), c.render(4)
            out.append((c.read_accum(), c.read_depth(), c.read_state(), c.pipeline()))
            c.close()
        assert [o[3] for o in out] == [0, 1]
        if flags & COMPAT_SHADOW_COLOR and len(flat.spot_lights) + len(flat.direct_lights):
            # coloured masks are PRODUCTS of the crossed triangles' opacity colours: the fused kernel multiplies them in its walk's order (the
            # reference's child order, one lane), the deferred cooperative kernel in its own (front to back, the testers of an entry over a
            # quad) — the same factors, so the sums agree to float rounding, not to the bit; everything that is not a mask stays equal
            assert np.array_equal(out[0][0][..., 3], out[1][0][..., 3])
            assert np.allclose(out[0][0][..., :3], out[1][0][..., :3], rtol=2e-5, atol=1e-6)
        else:
            assert np.array_equal(out[0][0], out[1][0])
        assert np.array_equal(out[0][1], out[1][1])
        for k in out[0][2]:
            assert np.array_equal(out[0][2][k], out[1][2][k]), k
        assert out[0][0][..., 3].sum() > 0
