// hiprz_compat.hpp — what the reference's CUDA engine computes and its CPU engine does not, behind
// hiprz_set_mode().  The CPU kernel is the parity oracle, so none of this can be compared with it bit for bit: every feature has
// its own flag, the default mode (0) never reaches this file, and the GPU tests check each feature against its analytic
// expectation (tests/test_cuda_compat_gpu.py).
//
//   HIPRZ_COMPAT_BEER_LAMBERT  ray.color *= opacityColor(medium) * pow(opacityColor(medium).alpha, distance)
//                              (RayZath/cuda_render_kernel.cu:174-176)
//   HIPRZ_COMPAT_SCATTERING    the medium a ray travels in scatters it after -log(u + 1e-4) / sigma (Material::applyScattering,
//                              cuda_material.cuh:141-159; World::closestIntersection, cuda_world.cuh:91-100)
//   HIPRZ_COMPAT_SHADOW_COLOR  shadow rays go THROUGH triangles, the mask is multiplied by each one's opacityColor(uv)
//                              (cuda_instance.cuh:92-164; cuda_render_kernel.cu:282-288)
//   HIPRZ_COMPAT_TEXTURE_MULT  a texture multiplies the material colour, an emission map the emission (cuda_material.cuh:75-123)
//   HIPRZ_COMPAT_FILTERING     the maps' filter mode (point / linear) and address mode (wrap / clamp / mirror / border) are
//                              honoured (cuda_buffer.cuh:364-438: CUDA texture objects, normalised coordinates)
// HIPRZ_COMPAT_REPROJECTION (cuda_camera.cuh:390-426) is not an integrator flag: rz_reproject_kernel (hiprz_api.hip) acts on the frame
// state after the first pass of a restarted frame, whichever kernels render it.
#pragma once
#include "hiprz_device.hpp"

namespace hiprz {

// ---- maps as CUDA texture objects sample them ----
// texel index along one axis under an address mode; `inside` = false only for a border-mode access outside the image
RZ_DEV int compat_texel(int i, int n, uint32_t mode, bool& inside) {
    inside = true;
    if (mode == HIPRZ_TEX_ADDRESS_CLAMP) return i < 0 ? 0 : (i >= n ? n - 1 : i);
    if (mode == HIPRZ_TEX_ADDRESS_BORDER) {
        inside = i >= 0 && i < n;
        return i < 0 ? 0 : (i >= n ? n - 1 : i);
    }
    if (mode == HIPRZ_TEX_ADDRESS_MIRROR) {
        int k = i % (2 * n);
        if (k < 0) k += 2 * n;
        return k < n ? k : 2 * n - 1 - k;
    }
    int k = i % n;  // wrap
    return k < 0 ? k + n : k;
}
// one texel as four floats: RGBA8 and R8 normalised by 255 (cudaReadModeNormalizedFloat), R32F as is; missing channels are 0
RZ_DEV col4 compat_load(const DScene& s, uint32_t kind, size_t offset, uint32_t width, int x, int y) {
    const size_t at = size_t(y) * width + size_t(x);
    if (kind == HIPRZ_TEX_RGBA8) return from_u8(*reinterpret_cast<const uint32_t*>(s.texels + offset + 4u * at));
    if (kind == HIPRZ_TEX_R8) return col4{float(s.texels[offset + at]) / 255.0f, 0.0f, 0.0f, 0.0f};
    return col4{*reinterpret_cast<const float*>(s.texels + offset + 4u * at), 0.0f, 0.0f, 0.0f};
}
// TextureBuffer::fetch of the CUDA engine (cuda_buffer.cuh:427-438): transform, then tex2D(x, 1 - y) under the record's modes
template <bool COUNT>
RZ_DEV col4 compat_fetch(const DScene& s, int32_t tex, float u, float v, Counters& cnt) {
    const float4 a = s.textures[3 * tex], b = s.textures[3 * tex + 1], c = s.textures[3 * tex + 2];
    const uint32_t kind = __float_as_uint(a.x), width = __float_as_uint(a.y), height = __float_as_uint(a.z), offset = __float_as_uint(a.w);
    const uint32_t sampling = __float_as_uint(c.w), address = sampling & 0xFF00u;
    u += b.z, v += b.w;
    const float xx = u * c.y - v * c.z, yy = u * c.z + v * c.y;
    u = xx * b.x, v = 1.0f - yy * b.y;
    RZ_COUNT(texel_fetches);
    bool in_x, in_y;
    if ((sampling & 0xFFu) != HIPRZ_TEX_FILTER_LINEAR) {
        const int x = compat_texel(int(floorf(u * float(width))), int(width), address, in_x);
        const int y = compat_texel(int(floorf(v * float(height))), int(height), address, in_y);
        return in_x && in_y ? compat_load(s, kind, offset, width, x, y) : splat(0.0f);
    }
    const float fx = u * float(width) - 0.5f, fy = v * float(height) - 0.5f;
    const int x0 = int(floorf(fx)), y0 = int(floorf(fy));
    const float ax = fx - float(x0), ay = fy - float(y0);
    col4 sum = splat(0.0f);
    for (int k = 0; k < 4; ++k) {
        const int x = compat_texel(x0 + (k & 1), int(width), address, in_x), y = compat_texel(y0 + (k >> 1), int(height), address, in_y);
        const float w = ((k & 1) ? ax : 1.0f - ax) * ((k >> 1) ? ay : 1.0f - ay);
        if (in_x && in_y) sum = sum + compat_load(s, kind, offset, width, x, y) * w;
    }
    return sum;
}

// Material::opacityColor (cuda_material.cuh:80-95): colour with alpha turned into transparency, times the texture's likewise
template <bool COUNT>
RZ_DEV col4 compat_opacity_color(const DScene& s, const Material& m, float u, float v, bool textured, bool filtering, Counters& cnt) {
    col4 c = from_u8(m.color);
    c.a = 1.0f - c.a;
    if (!textured || m.texture < 0) return c;
    col4 t = filtering ? compat_fetch<COUNT>(s, m.texture, u, v, cnt) : fetch_rgba8<COUNT>(s, m.texture, u, v, cnt);
    t.a = 1.0f - t.a;
    return c * t;
}

// The factor a crossed triangle contributes to a shadow mask: opacityColor of its material at the crossing's texture coordinates
// (cuda_instance.cuh:105-112) — the tester's half of the cooperative mask walk (hiprz_device.hpp: any_hit_coop_mask).
template <bool COUNT>
RZ_DEV col4 compat_crossing_color(const DScene& s, uint32_t inst, uint32_t tri, uint32_t tri_flags, float b1, float b2, bool filtering, Counters& cnt) {
    const uint32_t material_base = __float_as_uint(s.instances[7 * inst + 1].w), material_count = __float_as_uint(s.instances[7 * inst + 2].w);
    float u = 0.0f, v = 0.0f;
    if (tri_flags & HIPRZ_TRI_HAS_TEXCRDS) {  // Triangle::texcrdFromBarycenter, mesh_component.cpp:115-123
        const float4 uv12 = s.tri_attrs[6 * size_t(tri) + 4], uv3 = s.tri_attrs[6 * size_t(tri) + 5];
        const float b3 = 1.0f - b1 - b2;
        u = uv12.x * b3 + uv12.z * b1 + uv3.x * b2;
        v = uv12.y * b3 + uv12.w * b1 + uv3.y * b2;
    }
    uint32_t slot = tri_flags & HIPRZ_TRI_MATERIAL_MASK;
    if (slot > 63u) slot = 63u;
    const int32_t mat = slot < material_count ? s.inst_materials[material_base + slot] : -1;
    const Material material = load_material(s, mat < 0 ? HIPRZ_MATERIAL_DEFAULT : uint32_t(mat));
    return compat_opacity_color<COUNT>(s, material, u, v, true, filtering, cnt);
}

// Shadow mask of the CUDA engine (cuda_bvh.cuh:172-232, cuda_instance.cuh:92-164, 215-229): starts white, every triangle the
// shadow ray crosses multiplies it by that triangle's opacityColor(uv); the walk ends early once the mask's alpha drops below 1e-4.
template <bool COUNT>
RZ_DEV col4 compat_shadow_mask(const DScene& s, const Ray& ray, bool filtering, Counters& cnt) {
    col4 mask = splat(1.0f);
    RZ_COUNT(shadow_rays);
    if (s.n_instances == 0u) return mask;
    const TopCache top{nullptr, nullptr, 0u};
    WalkRay g;
    g.o = ray.o, g.d = ray.d, g.near_ = ray.near_, g.far_ = ray.far_;
    prepare<false>(g, false);
    uint32_t n = s.tlas_root;
    while (n != RZ_END) {
        float4 n0, n1;
        uint32_t link;
        fetch_node(s, top, n, n0, n1, link);
        RZ_COUNT(box_tests);
        RZ_COUNT(shadow_box_tests);
        if (box_hit<false>(n0, n1, g)) {
            const uint32_t begin = __float_as_uint(n1.z), meta = __float_as_uint(n1.w);
            if (!(meta & HIPRZ_NODE_LEAF)) {
                n = begin;
                continue;
            }
            const uint32_t end = begin + (meta & HIPRZ_NODE_COUNT_MASK);
            for (uint32_t i = begin; i < end; ++i) {
                const uint32_t inst = s.tlas_order[i];
                float4 ib0, ib1;
                load_instance_box(s, inst, ib0, ib1);
                RZ_COUNT(box_tests);
                RZ_COUNT(shadow_box_tests);
                if (!box_hit<false>(ib0, ib1, g)) continue;
                const InstanceXform x = load_instance_xform(s, inst);
                WalkRay lr;
                to_local<false>(x, g, lr, false);
                uint32_t m = x.blas_root;
                while (m != RZ_END) {
                    float4 m0, m1;
                    uint32_t mlink;
                    fetch_node(s, top, m, m0, m1, mlink);
                    RZ_COUNT(box_tests);
                    RZ_COUNT(shadow_box_tests);
                    if (box_hit<false>(m0, m1, lr)) {
                        const uint32_t mbegin = __float_as_uint(m1.z), mmeta = __float_as_uint(m1.w);
                        if (!(mmeta & HIPRZ_NODE_LEAF)) {
                            m = mbegin;
                            continue;
                        }
                        for (uint32_t tj = mbegin; tj < mbegin + (mmeta & HIPRZ_NODE_COUNT_MASK); ++tj) {
                            const float4 ta = s.tris[3 * tj], tb = s.tris[3 * tj + 1], tc = s.tris[3 * tj + 2];
                            float t, b1, b2, det;
                            RZ_COUNT(tri_tests);
                            RZ_COUNT(shadow_tri_tests);
                            if (!tri_hit(xyz(ta), xyz(tb), xyz(tc), lr, t, b1, b2, det)) continue;
                            mask = mask * compat_crossing_color<COUNT>(s, inst, tj, __float_as_uint(ta.w), b1, b2, filtering, cnt);
                            if (mask.a < 1.0e-4f) return mask;
                        }
                    }
                    m = mlink;
                }
            }
        } else if (n == s.tlas_root) {
            return mask;  // root box missed
        }
        n = link;
    }
    return mask;
}

}  // namespace hiprz
