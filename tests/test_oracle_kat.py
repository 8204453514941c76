"""Known-answer tests of the CPU oracle's building blocks.

Two kinds of pins: (1) the committed vectors of tests/golden/kat.npz (regression), and
(2) independent numpy float32 re-derivations written from the reference's formulas
(cpu_render_utils.cpp:8-27, render_parts.cpp:197-217, mesh_component.cpp:52-83), so the C
restatement is checked by something other than itself.  The one value that comes from the
real reference is the RNG pair recorded in SURVEY.md Appendix D."""
import os

import numpy as np

import oracle

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "kat.npz"))
f32 = np.float32


def np_rng(sx, sy, r, n):
    a, b = f32(f32(sx) + f32(sy)), f32(f32(r) * f32(245.310913))
    out = []
    for _ in range(n):
        af = f32(f32(a + f32(0.2311362)) * f32(b + f32(13.054377)))
        bf = f32(f32(a + f32(251.78431)) + f32(b - f32(73.054312)))
        a = f32(af - f32(np.trunc(af)))
        b = f32(bf - f32(np.trunc(bf)))
        out.append(abs(b))
    return np.array(out, dtype=f32)


def test_rng_matches_the_reference_pair_and_numpy(built):
    L = oracle.load()
    out = np.zeros(64, f32)
    L.rzo_rng_sequence(0.25, 0.5, 0.75, 64, out.ctypes.data)
    assert np.array_equal(out[:2], G["rng_reference_pair"])  # measured on the real reference
    for seed, want in zip(G["rng_seeds"], G["rng_sequences"]):
        L.rzo_rng_sequence(float(seed[0]), float(seed[1]), float(seed[2]), 64, out.ctypes.data)
        assert np.array_equal(out, want)
        assert np.array_equal(out, np_rng(seed[0], seed[1], seed[2], 64))
        assert (out >= 0).all() and (out < 1).all()


def np_mix32(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def test_seed_table_matches_spec_and_backend(built):
    from rayzath_amd import _lib
    L, H = oracle.load(), _lib.load()
    for p in range(3):
        for i in range(256):
            h = np_mix32(20240501 ^ np_mix32((p + 0x9E3779B9) & 0xFFFFFFFF))
            h = np_mix32(h ^ ((i * 0x85EBCA6B + 1) & 0xFFFFFFFF))
            want = f32(f32(h >> 8) * f32(20.0 / 16777216.0) - f32(10.0))
            assert L.rzo_seed_value(20240501, p, i) == want == G["seed_table"][p, i] == H.hiprz_seed_value(20240501, p, i)


def np_box(mn, mx, o, d, near, far):
    with np.errstate(divide="ignore", invalid="ignore"):
        t = [f32(f32(c - oo) / dd) for c, oo, dd in ((mn[0], o[0], d[0]), (mx[0], o[0], d[0]), (mn[1], o[1], d[1]),
                                                      (mx[1], o[1], d[1]), (mn[2], o[2], d[2]), (mx[2], o[2], d[2]))]
    lo = lambda a, b: a if a < b else b
    hi = lambda a, b: a if a > b else b
    tmin = hi(hi(lo(t[0], t[1]), lo(t[2], t[3])), lo(t[4], t[5]))
    tmax = lo(lo(hi(t[0], t[1]), hi(t[2], t[3])), hi(t[4], t[5]))
    return int(not (tmax < near or tmin > tmax or tmin > far))


def test_box_test(built):
    L = oracle.load()
    for i in range(len(G["box_hit"])):
        mn, mx, o, d = (np.ascontiguousarray(G[k][i]) for k in ("box_min", "box_max", "box_origin", "box_direction"))
        got = L.rzo_box_test(mn.ctypes.data, mx.ctypes.data, o.ctypes.data, d.ctypes.data, float(G["box_near"][i]), float(G["box_far"][i]))
        assert got == G["box_hit"][i] == np_box(mn, mx, o, d, G["box_near"][i], G["box_far"][i]), i
    assert 20 < G["box_hit"].sum() < 236  # both outcomes are exercised


def np_tri(v, o, d, near, far):
    dot = lambda a, b: f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))
    cross = lambda a, b: np.array([f32(f32(a[1] * b[2]) - f32(a[2] * b[1])), f32(f32(a[2] * b[0]) - f32(a[0] * b[2])),
                                   f32(f32(a[0] * b[1]) - f32(a[1] * b[0]))], dtype=f32)
    e1, e2 = (v[1] - v[0]).astype(f32), (v[2] - v[0]).astype(f32)
    p = cross(d, e2)
    det = dot(e1, p)
    det = f32(det + f32(f32(int(det > f32(-1e-7)) & int(det < f32(1e-7))) * f32(1e-7)))
    inv = f32(f32(1.0) / det)
    tv = (o - v[0]).astype(f32)
    b1 = f32(dot(tv, p) * inv)
    if b1 < 0 or b1 > 1:
        return None
    q = cross(tv, e1)
    b2 = f32(dot(d, q) * inv)
    if b2 < 0 or f32(b1 + b2) > 1:
        return None
    t = f32(dot(e2, q) * inv)
    if t <= near or t >= far:
        return None
    return np.array([t, b1, b2, float(det > 0)], dtype=f32)


def test_triangle_test(built):
    L = oracle.load()
    out = np.zeros(4, f32)
    hits = 0
    for i in range(len(G["tri_hit"])):
        v, o, d = (np.ascontiguousarray(G[k][i]) for k in ("tri_vertices", "tri_origin", "tri_direction"))
        got = L.rzo_triangle_test(v[0].ctypes.data, v[1].ctypes.data, v[2].ctypes.data, o.ctypes.data, d.ctypes.data, 0.0,
                                  float(G["tri_far"][i]), out.ctypes.data)
        want = np_tri(v, o, d, f32(0), G["tri_far"][i])
        assert got == G["tri_hit"][i] == int(want is not None), i
        if got:
            hits += 1
            assert np.array_equal(out, G["tri_result"][i]) and np.array_equal(out, want), i
    assert 30 < hits < 226


def test_helpers_against_committed_vectors(built):
    """fresnel is libm-free (bit-exact); the sampling helpers call sinf/cosf/acosf (tolerance 1e-6 abs)."""
    L = oracle.load()
    fact, v = np.zeros(2, f32), np.zeros(3, f32)
    for i in range(len(G["fresnel"])):
        n, inc = np.ascontiguousarray(G["helper_n"][i]), np.ascontiguousarray(G["helper_i"][i])
        fact[:] = 0
        r = L.rzo_fresnel(n.ctypes.data, inc.ctypes.data, float(G["helper_n1"][i]), float(G["helper_n2"][i]), fact.ctypes.data)
        assert r == G["fresnel"][i] and np.array_equal(fact, G["fresnel_factors"][i])
        assert 0.0 <= r <= 1.0
        r1, r2 = (float(x) for x in G["helper_r"][i])
        L.rzo_cosine_sample_hemisphere(r1, r2, n.ctypes.data, v.ctypes.data)
        # localCoordinate's axes are not unit length (cross products with a non-orthogonal helper axis,
        # cpu_render_utils.cpp:74-83), so the samples are only *nearly* unit: the reference renormalises
        # the direction when the ray is loaded again (cpu_render_utils.hpp:41-46).
        assert np.allclose(v, G["cosine_hemisphere"][i], atol=1e-6) and np.dot(v, n) >= -1e-6
        assert 0.5 < np.linalg.norm(v) < 1.0 + 1e-5
        L.rzo_sample_sphere(r1, r2, n.ctypes.data, v.ctypes.data)
        assert np.allclose(v, G["sample_sphere"][i], atol=1e-6) and 0.5 < np.linalg.norm(v) < 1.0 + 1e-5
        L.rzo_sample_disk(r1, r2, n.ctypes.data, 0.5, v.ctypes.data)
        assert np.allclose(v, G["sample_disk"][i], atol=1e-6) and np.linalg.norm(v) <= 0.5 + 1e-6 and abs(np.dot(v, n)) < 1e-6


def test_tonemap(built):
    """cpu_engine_renderer.cpp:224-235: rgb/alpha * pi*aperture^2 * exposure * 1e5, x/(x+1), truncate to u8."""
    L = oracle.load()
    out = np.zeros(4, np.uint8)
    for rgba, want in zip(G["tonemap_in"], G["tonemap_out"]):
        L.rzo_tonemap_pixel(np.ascontiguousarray(rgba).ctypes.data, 0.02, 1.0 / 60.0, out.ctypes.data)
        assert np.array_equal(out, want)
        c = rgba[:3].astype(np.float64) / (rgba[3] if rgba[3] else 1.0) * (0.02 ** 2 * np.pi) * (1 / 60) * 1e5
        approx = c / (c + 1) * 255
        assert (np.abs(out[:3].astype(float) - np.floor(approx)) <= 1).all() and out[3] == 255
