// hiprz_kernels.hpp — the pass kernels (templates).  Instantiated by the launch units hiprz_launch_*.hip.
//
// Replaces, for the HIPGPU backend, the reference's cuda_render_kernel.cu (renderFirstPass / renderCumulativePass,
// traceRay, directIllumination).  Written for gfx950 only: wave64, 256-thread workgroups = one 32x8-pixel tile, or
// single-wave workgroups of 64 rays in the walk kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "hiprz.h"
#include "hiprz_device.hpp"
#include "hiprz_compat.hpp"

namespace hiprz {

// One pass = one path segment per owned pixel: renderFirstPass (cpu_engine_kernel.cpp:15-57)
// when FIRST, else renderCumulativePass (:58-101), with traceRay (:113-178) inlined.
//
// The pass is written as three pieces — load_path, the closest-hit walk, shade_and_store — used by three
// pipelines that give identical results:
//   fused    (rz_pass_kernel):  all three in one kernel; state + accumulator cross HBM once (112 B/pixel).
//   resident (rz_batch_kernel): the fused pass in a loop over the passes of a render call; state stays on chip.
//   split  (rz_trace_kernel -> rz_shade_kernel): the walk runs in its own lean kernel (ray + hit only: no
//          register spills with the packed shared-reciprocal box test, higher occupancy) and hands a 20-B hit
//          record per pixel to the shading kernel through HBM (+88 B/pixel of traffic).
//
// LDS_SCENE: the workgroup first stages the scene's hot blob (geometry + shading records) into LDS and
// every traversal / shading fetch becomes a ds_read instead of a dependent global load.
struct PathState {
    Ray ray;
    col4 color;
    uint32_t material, depth;
};

template <bool LDS_SCENE>
RZ_DEV uint32_t stage_scene(DScene& s, unsigned char* lds) {
    if constexpr (LDS_SCENE) {
        float4* dst = reinterpret_cast<float4*>(lds);
        const uint32_t n16 = s.hot_bytes >> 4;
        for (uint32_t i = threadIdx.x; i < n16; i += 256u) dst[i] = s.hot[i];
        __syncthreads();
        repoint_hot(s, lds);
        return s.hot_bytes;
    }
    return 0u;
}

// the segment's ray: generateSimpleRay on the first pass, CameraContext::getRay afterwards
template <bool FIRST>
RZ_DEV void load_path(const DFrame& f, const DCamera& cam, const PixelId& p, PathState& ps) {
    ps.color = splat(1.0f);
    ps.material = HIPRZ_MATERIAL_WORLD, ps.depth = 0u;
    ps.ray.o = ps.ray.d = V3(0.0f, 0.0f, 1.0f), ps.ray.near_ = 0.0f, ps.ray.far_ = 0.0f;
    if (!p.active) return;
    if constexpr (FIRST) {
        generate_simple_ray(cam, ps.ray, p.x, p.y);
    } else {
        const float4 s0 = f.st0[p.local], s1 = f.st1[p.local];
        const float2 s2 = f.st2[p.local];
        const uint32_t bits = __float_as_uint(s2.y);
        ps.ray.o = V3(s0.x, s0.y, s0.z);
        ps.ray.d = normalized(V3(s0.w, s1.x, s1.y));  // SceneRay ctor normalises (cpu_render_utils.hpp:41-46)
        ps.ray.near_ = 0.0f, ps.ray.far_ = RZ_FLT_MAX;
        ps.color = col4{s1.z, s1.w, s2.x, 1.0f};
        ps.material = bits & 0xFFFFu;
        ps.depth = (bits >> 16) & 0xFFu;
        if (ps.depth == 0u) ps.ray.near_ = cam.near_, ps.ray.far_ = cam.far_;
    }
}

// closest hit of the segment with the selected walk; MODE 2 must be reached by all 256 threads
// MODE 4 = MODE 2 for a world tree of ONE leaf with at most 8 instances (the host checks it at upload): the instantiation without the
// general world walk, whose state would only cost registers (closest_hit_binned<..., FLAT>)
constexpr bool binned_mode(int mode) { return mode == 2 || mode == 4; }
template <int MODE, bool COUNT, bool RCP>
RZ_DEV int trace_path(const DScene& s, unsigned char* workspace, uint32_t* lds_column, bool active, Ray& ray, Hit& hit, Counters& cnt) {
    if constexpr (binned_mode(MODE)) {
        return closest_hit_binned<COUNT, RCP, MODE == 4>(s, workspace, active, ray, hit, cnt);
    } else {
        hit.instance = -1, hit.triangle = 0, hit.bx = hit.by = 0.0f, hit.external = true;
        return active ? closest_hit<MODE, COUNT, RCP>(s, lds_column, ray, hit, cnt) : 0;
    }
}

// everything of traceRay after the closest hit (active lanes only): returns the segment's radiance and whether the path
// goes on, and leaves the NEXT segment's ray / colour / material / depth in `ps` (TracingResult::repositionRay, or a fresh
// antialiased camera ray when the path ended).  ps.ray.far_ must hold the hit distance.
template <bool COUNT, int SHADOW = 1>
RZ_DEV void shade_segment(const DScene& s, const DCamera& cam, const DConfig& cfg, const PixelId& p, PathState& ps, uint32_t pass,
                          int found, const Hit& hit, const ShadowCtx& lds_column, Counters& cnt, col4& final_color, bool& path_continues,
                          Rng* shared_rng = nullptr) {
    constexpr bool COMPAT = shadow_mode_compat(SHADOW);  // the CUDA engine's behaviours by cfg.flags; `found == 3`: the medium scattered the ray
    Ray& ray = ps.ray;
    col4& ray_color = ps.color;
    uint32_t& ray_material = ps.material;
    uint32_t& depth = ps.depth;
    const uint32_t pixel_idx = p.y * cam.width + p.x;
    Rng own_rng(float(p.x) / float(cam.width), float(p.y) / float(cam.height), seed_value(cfg.seed, pass, (pixel_idx + depth) & 255u));
    Rng& rng = COMPAT ? *shared_rng : own_rng;  // compat: the kernel drew the scattering distance from this stream before the walk
    const bool filtering = COMPAT && (cfg.flags & HIPRZ_COMPAT_FILTERING) != 0u;

    final_color = splat(0.0f);
    Surface sf;
    sf.surface_material = sf.behind_material = HIPRZ_MATERIAL_WORLD;
    sf.u = sf.v = 0.0f;
    sf.normal = sf.mapped_normal = V3(0.0f, 0.0f, 0.0f);
    sf.fresnel = 1.0f, sf.reflectance = 0.0f, sf.tint_factor = 0.0f, sf.refr_x = sf.refr_y = 0.0f;
    sf.metalness = sf.roughness = 0.0f;

    constexpr bool TEX = SHADOW != RZ_SHADOW_PLAIN;  // PLAIN: the scene has no maps at all (every map index is -1)
    Material m;
    if (found == 2) {
        analyze_intersection<COUNT, TEX>(s, hit, sf, m, cnt, filtering);
    } else if (COMPAT && found == 3) {  // Material::applyScattering: the medium itself is the surface, its normal the ray's direction
        m = load_material(s, ray_material);
        sf.surface_material = sf.behind_material = ray_material;
        sf.normal = sf.mapped_normal = ray.d;
    } else {
        m = load_material(s, HIPRZ_MATERIAL_WORLD);
        if (TEX && found == 1) {  // texcrd of the sky sphere (cpu_engine_kernel.cpp:292-295); only a map reads it
            sf.u = -(0.5f + (RZ_ATAN2F(ray.d.z, ray.d.x) / (RZ_PI_F * 2.0f)));
            sf.v = 0.5f + (RZ_ASINF(ray.d.y) / RZ_PI_F);
        }
    }
    sf.surface_scattering = m.scattering;
    // fetchColor / fetchEmission (:505-512, 523-528)
    if (COMPAT && (cfg.flags & HIPRZ_COMPAT_TEXTURE_MULT)) {  // cuda_material.cuh:86-95, 118-123: maps multiply
        sf.color = compat_opacity_color<COUNT>(s, m, sf.u, sf.v, true, filtering, cnt);
        sf.emission = m.emission;
        if (m.emission_map >= 0) sf.emission *= filtering ? compat_fetch<COUNT>(s, m.emission_map, sf.u, sf.v, cnt).r : fetch_r32f<COUNT>(s, m.emission_map, sf.u, sf.v, cnt);
    } else {
        sf.color = from_u8(m.color);
        if (TEX && m.texture >= 0) sf.color = filtering ? compat_fetch<COUNT>(s, m.texture, sf.u, sf.v, cnt) : fetch_rgba8<COUNT>(s, m.texture, sf.u, sf.v, cnt);
        sf.color.a = 1.0f - sf.color.a;
        sf.emission = TEX && m.emission_map >= 0 ? (filtering ? compat_fetch<COUNT>(s, m.emission_map, sf.u, sf.v, cnt).r : fetch_r32f<COUNT>(s, m.emission_map, sf.u, sf.v, cnt)) : m.emission;
    }
    if (COMPAT && (cfg.flags & HIPRZ_COMPAT_BEER_LAMBERT)) {  // Beer's law in the medium the segment crossed: cuda_render_kernel.cu:158-176
        col4 medium = from_u8(load_material(s, ray_material).color);
        medium.a = 1.0f - medium.a;
        ray_color = ray_color * (medium * RZ_POWF(medium.a, ray.far_));
    }
    if (sf.emission > 0.0f) final_color = final_color + (ray_color * sf.color) * sf.emission;

    v3 point = V3(0.0f, 0.0f, 0.0f), next_direction = V3(0.0f, 0.0f, 0.0f);
    if (found != 2 && !(COMPAT && found == 3)) {
        depth = 255u;  // TracingState::endPath
    } else {
        RZ_COUNT(hits);
        depth += 1u;
        sf.metalness = TEX && m.metalness_map >= 0 ? (filtering ? compat_fetch<COUNT>(s, m.metalness_map, sf.u, sf.v, cnt).r : fetch_r8<COUNT>(s, m.metalness_map, sf.u, sf.v, cnt)) : m.metalness;
        sf.roughness = TEX && m.roughness_map >= 0 ? (filtering ? compat_fetch<COUNT>(s, m.roughness_map, sf.u, sf.v, cnt).r : fetch_r8<COUNT>(s, m.roughness_map, sf.u, sf.v, cnt)) : m.roughness;
        sf.fresnel = fresnel_specular_ratio(sf.mapped_normal, ray.d, material_ior(s, ray_material), material_ior(s, sf.behind_material),
                                            sf.refr_x, sf.refr_y);
        sf.reflectance = lerpf(sf.fresnel, 1.0f, sf.metalness);

        next_direction = sample_direction(ray.d, ray_material, sf, rng);
        point = (ray.o + ray.d * ray.far_) + sf.normal * (0.0001f * ray.far_);

        const col4 direct = direct_illumination<SHADOW, COUNT>(s, cfg, lds_column, ray.d, ray_material, point, next_direction, sf, rng, cnt);
        if constexpr (shadow_mode_defers(SHADOW)) {  // rz_shadow_kernel adds (direct * a) * b once it knows the shadow masks
            lds_column.defer_done = true;
            lds_column.defer_a = ray_color, lds_column.defer_b = lerp(splat(1.0f), sf.color, sf.metalness);
        } else if constexpr (SHADOW == RZ_SHADOW_NONE || SHADOW == RZ_SHADOW_PLAIN) {
            // direct == 0: (0 * ray_color) * lerp(..) is +0 for the finite, non-negative colours a path carries, and final_color
            // (+0 plus emission terms) is never -0, so the addition the lit variants perform leaves it unchanged
            (void)direct;
        } else {
            final_color = final_color + (direct * ray_color) * lerp(splat(1.0f), sf.color, sf.metalness);
        }
        ray_color = lerp(ray_color, ray_color * sf.color, sf.tint_factor);  // ColorF::Blend
    }
    path_continues = depth < cfg.max_depth;
    if (path_continues) {  // TracingResult::repositionRay
        ray.o = point;
        ray.d = next_direction;
    } else {
        RZ_COUNT(finished);
        generate_antialiased_ray(cam, ray, p.x, p.y, rng);
        ray_material = HIPRZ_MATERIAL_WORLD;
        ray_color = splat(1.0f);
        depth = 0u;
    }
}

// shade_segment + accumulation + next-segment state to HBM (renderFirstPass / renderCumulativePass after traceRay)
template <bool FIRST, bool COUNT, int SHADOW = 1>
RZ_DEV void shade_and_store(const DScene& s, const DCamera& cam, const DConfig& cfg, const DFrame& f, const PixelId& p, PathState& ps,
                            int found, const Hit& hit, const ShadowCtx& lds_column, Counters& cnt, Rng* shared_rng = nullptr) {
    const float hit_distance = ps.ray.far_;
    col4 final_color;
    bool path_continues;
    shade_segment<COUNT, SHADOW>(s, cam, cfg, p, ps, FIRST ? 0u : *f.pass, found, hit, lds_column, cnt, final_color, path_continues, shared_rng);
    const Ray& ray = ps.ray;
    const col4& ray_color = ps.color;
    const uint32_t ray_material = ps.material, depth = ps.depth;

    // ---- accumulate ----
    if constexpr (FIRST) f.depth[p.local] = hit_distance;
    if constexpr (shadow_mode_defers(SHADOW)) {
        // the radiance so far + what rz_shadow_kernel needs to finish it; it also does the accumulation
        const uint32_t bits = (path_continues ? 1u : 0u) | (lds_column.defer_done ? 2u : 0u) | (lds_column.defer_mask << 2);
        float4* rec = f.nee + size_t(p.local) * f.nee_quads;
        rec[0] = make_float4(final_color.r, final_color.g, final_color.b, __uint_as_float(bits));
        if (lds_column.defer_done) {
            rec[2] = make_float4(lds_column.defer_a.r, lds_column.defer_a.g, lds_column.defer_a.b, lds_column.defer_a.a);
            rec[3] = make_float4(lds_column.defer_b.r, lds_column.defer_b.g, lds_column.defer_b.b, lds_column.defer_b.a);
        }
    } else {
        col4 value;
        if constexpr (FIRST) {
            value = col4{final_color.r, final_color.g, final_color.b, float(!path_continues)};
        } else {
            const float4 acc = f.accum[p.local];
            value = col4{acc.x + final_color.r, acc.y + final_color.g, acc.z + final_color.b, acc.w + float(!path_continues)};
        }
        f.accum[p.local] = make_float4(value.r, value.g, value.b, value.a);
    }

    // ---- next segment ----
    f.st0[p.local] = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.d.x);
    f.st1[p.local] = make_float4(ray.d.y, ray.d.z, ray_color.r, ray_color.g);
    f.st2[p.local] = make_float2(ray_color.b, __uint_as_float((ray_material & 0xFFFFu) | (depth << 16)));
    if (f.sort_key) f.sort_key[p.local] = ray_sort_key(s, ray.o, ray.d, s.sort_variant);
    if constexpr (shadow_mode_defers(SHADOW)) {
        // the shadow rays of this pixel start at the hit point and point at the light the (last) sample chose: rays from one cell to
        // one light walk the same instances.  Pixels without a sample have nothing to walk and sort to the end.
        if (f.shadow_key) {
            uint32_t key = 0x00FFFFFEu;
            if (lds_column.defer_mask) {
                if (s.shadow_variant & 0x400u) {
                    // the LIGHT the ray goes to, then the origin's cell in a 64^3 grid: towards one light a point has one direction, so the
                    // light says what a direction code only approximates (round 4, E: 38.5 -> 37.8 ms per step against cell, then direction)
                    const float gx = fminf(fmaxf((lds_column.key_o[0] - s.bounds_min[0]) * s.bounds_scale[0] * 2.0f, 0.0f), 63.0f);
                    const float gy = fminf(fmaxf((lds_column.key_o[1] - s.bounds_min[1]) * s.bounds_scale[1] * 2.0f, 0.0f), 63.0f);
                    const float gz = fminf(fmaxf((lds_column.key_o[2] - s.bounds_min[2]) * s.bounds_scale[2] * 2.0f, 0.0f), 63.0f);
                    key = ((lds_column.key_light & 7u) << 21) | ((spread3(uint32_t(gx)) | (spread3(uint32_t(gy)) << 1) | (spread3(uint32_t(gz)) << 2)) << 3);
                } else {
                    key = ray_sort_key(s, V3(lds_column.key_o[0], lds_column.key_o[1], lds_column.key_o[2]),
                                       V3(lds_column.key_dir[0], lds_column.key_dir[1], lds_column.key_dir[2]), s.shadow_variant & 0xFFu);
                }
                // pixels with the same set of samples together: the shadow kernel's loop over the sample slots is wave-uniform, and a slot
                // that only a few of a wave's pixels hold costs the wave a whole walk
                if (s.shadow_variant & 0x100u) key = ((3u - (lds_column.defer_mask & 3u)) << 22) | (key >> 2);
            }
            f.shadow_key[p.local] = key;
        }
    }
}

template <bool COUNT>
RZ_DEV void flush_counters(const DFrame& f, uint32_t segments, const Counters& cnt) {
    if constexpr (COUNT) {
        uint32_t v[10] = {segments,        cnt.box_tests,     cnt.tri_tests,     cnt.hits,     cnt.shadow_rays,
                          cnt.light_samples, cnt.texel_fetches, cnt.finished, cnt.shadow_box_tests, cnt.shadow_tri_tests};
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            uint32_t x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
            if ((threadIdx.x & 63u) == 0u && x) atomicAdd(&f.counters[k], (unsigned long long)x);
        }
    }
}

// LDS carve-up shared by the kernels: [staged scene blob][walk workspace].  For MODE 2 the workspace is
// BinnedLds and its stack columns double as the LDS stack of the shadow rays; otherwise it is the stack.
template <int MODE>
RZ_DEV uint32_t* stack_column(unsigned char* workspace) {
    return reinterpret_cast<uint32_t*>(binned_mode(MODE) ? workspace + BinnedLds::kFixedBytes : workspace) + threadIdx.x;
}

// ---- fused pipeline ----
template <bool FIRST, bool COUNT, int MODE, bool LDS_SCENE>
__global__ void __launch_bounds__(256, RZ_MIN_WAVES) rz_pass_kernel(const DScene scene_in, const DCamera cam, const DConfig cfg, const DFrame f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rz_lds[];
    DScene s = scene_in;
    unsigned char* workspace = rz_lds + stage_scene<LDS_SCENE>(s, rz_lds);
    uint32_t* lds_column = stack_column<MODE>(workspace);
    const PixelId p = pixel_of_thread(f, cam, blockIdx.x, threadIdx.x);
    Counters cnt;
    PathState ps;
    load_path<FIRST>(f, cam, p, ps);
    Hit hit;
    int found;
    if constexpr (binned_mode(MODE)) {  // what the walk does not read is parked in LDS meanwhile
        // 4 KiB behind the binned walk's workspace (launch_pass adds them to the fused kernel's LDS size)
        uint32_t* park = reinterpret_cast<uint32_t*>(workspace + BinnedLds::kFixedBytes + (s.world_stack_entries + s.mesh_stack_entries) * 1024u);
        park[0 * 256 + threadIdx.x] = __float_as_uint(ps.color.r), park[1 * 256 + threadIdx.x] = __float_as_uint(ps.color.g);
        park[2 * 256 + threadIdx.x] = __float_as_uint(ps.color.b), park[3 * 256 + threadIdx.x] = ps.material | (ps.depth << 16);
        found = trace_path<MODE, COUNT, RZ_FUSED_SHARED_RCP != 0>(s, workspace, lds_column, p.active, ps.ray, hit, cnt);
        ps.color = col4{__uint_as_float(park[0 * 256 + threadIdx.x]), __uint_as_float(park[1 * 256 + threadIdx.x]),
                        __uint_as_float(park[2 * 256 + threadIdx.x]), 1.0f};
        const uint32_t bits = park[3 * 256 + threadIdx.x];
        ps.material = bits & 0xFFFFu, ps.depth = bits >> 16;
    } else {
        found = trace_path<MODE, COUNT, RZ_FUSED_SHARED_RCP != 0>(s, workspace, lds_column, p.active, ps.ray, hit, cnt);
    }
    if (p.active) shade_and_store<FIRST, COUNT>(s, cam, cfg, f, p, ps, found, hit, ShadowCtx{lds_column, TopCache{nullptr, nullptr, 0u}}, cnt);
    flush_counters<COUNT>(f, p.active ? 1u : 0u, cnt);
}

// ---- resident pipeline ----
// Pixels never interact, so a workgroup can take its tile through ALL the cumulative passes of a render batch in one
// launch: path state and accumulator stay in registers (parked in LDS during the binned walk) and cross HBM once per
// batch instead of once per pass, there is one launch per batch instead of two or three per pass, and the tone-mapped
// pixel is written on the way out.  Per pixel the arithmetic is that of n_passes launches of the fused kernel: the
// direction is re-normalised at the start of every segment as load_path does after reading it back, and the
// accumulator grows by the same sequence of additions.
// WAVES = waves per SIMD the register budget is cut for.  With 29 KB of LDS per workgroup (a Cornell-sized scene) five workgroups
// fit a CU, and when the grid oversubscribes the chip the 5-wave build of the plain instantiation wins although it spills more
// (96 VGPRs, 148 B of scratch: whole 1080p frame 2.15 -> 2.04 ms per step); a grid that fits the chip at once — an eighth of the
// frame on each of 8 GPUs — runs faster on the 4-wave build (0.326 against 0.350 ms), so launch_batch picks by grid size.
template <bool COUNT, int MODE, bool LDS_SCENE, int SHADING, int WAVES = RZ_MIN_WAVES>  // SHADING: 1 general, RZ_SHADOW_NONE (no lights), RZ_SHADOW_PLAIN (no lights, no maps)
__global__ void __launch_bounds__(256, WAVES) rz_batch_kernel(const DScene scene_in, const DCamera cam, const DConfig cfg, const DFrame f,
                                                                      uint32_t n_passes, uint32_t park_offset) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rz_lds[];
    const unsigned long long t_start = __builtin_readcyclecounter();
    const uint32_t unit = (f.launch_order && !f.xcd_swizzle) ? f.launch_order[blockIdx.x] : blockIdx.x;  // heaviest tiles first (DFrame::launch_order)
    const PixelId p = pixel_of_thread(f, cam, unit, threadIdx.x);
    if (!p.tile_inside) return;  // (the whole workgroup: the padding of the swizzled grid)
    DScene s = scene_in;
    unsigned char* workspace = rz_lds + stage_scene<LDS_SCENE>(s, rz_lds);
    uint32_t* lds_column = stack_column<MODE>(workspace);
    Counters cnt;
    PathState ps;
    load_path<false>(f, cam, p, ps);
    // the accumulator lives in LDS for the whole batch (touched once per pass); colour / material / depth join it
    // there while the binned walk runs
    uint32_t* park = reinterpret_cast<uint32_t*>(workspace + park_offset) + threadIdx.x;
    {
        const float4 acc = p.active ? f.accum[p.local] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        park[4 * 256] = __float_as_uint(acc.x), park[5 * 256] = __float_as_uint(acc.y);
        park[6 * 256] = __float_as_uint(acc.z), park[7 * 256] = __float_as_uint(acc.w);
    }
    const uint32_t pass0 = *f.pass;
    for (uint32_t i = 0; i < n_passes; ++i) {
        if (i != 0u && p.active) {  // what load_path does with the state the previous pass stored
            ps.ray.d = normalized(ps.ray.d);
            ps.ray.near_ = 0.0f, ps.ray.far_ = RZ_FLT_MAX;
            if (ps.depth == 0u) ps.ray.near_ = cam.near_, ps.ray.far_ = cam.far_;
        }
        Hit hit;
        int found;
        if constexpr (binned_mode(MODE)) {
            park[0 * 256] = __float_as_uint(ps.color.r), park[1 * 256] = __float_as_uint(ps.color.g);
            park[2 * 256] = __float_as_uint(ps.color.b), park[3 * 256] = ps.material | (ps.depth << 16);
            found = trace_path<MODE, COUNT, RZ_BATCH_SHARED_RCP != 0>(s, workspace, lds_column, p.active, ps.ray, hit, cnt);
            ps.color = col4{__uint_as_float(park[0 * 256]), __uint_as_float(park[1 * 256]), __uint_as_float(park[2 * 256]), 1.0f};
            const uint32_t bits = park[3 * 256];
            ps.material = bits & 0xFFFFu, ps.depth = bits >> 16;
        } else {
            found = trace_path<MODE, COUNT, RZ_BATCH_SHARED_RCP != 0>(s, workspace, lds_column, p.active, ps.ray, hit, cnt);
        }
        if (p.active) {
            col4 final_color;
            bool path_continues;
            shade_segment<COUNT, SHADING>(s, cam, cfg, p, ps, pass0 + i, found, hit, ShadowCtx{lds_column, TopCache{nullptr, nullptr, 0u}}, cnt, final_color, path_continues);
            park[4 * 256] = __float_as_uint(__uint_as_float(park[4 * 256]) + final_color.r);
            park[5 * 256] = __float_as_uint(__uint_as_float(park[5 * 256]) + final_color.g);
            park[6 * 256] = __float_as_uint(__uint_as_float(park[6 * 256]) + final_color.b);
            park[7 * 256] = __float_as_uint(__uint_as_float(park[7 * 256]) + float(!path_continues));
        }
    }
    if (p.active) {
        const float4 acc = make_float4(__uint_as_float(park[4 * 256]), __uint_as_float(park[5 * 256]), __uint_as_float(park[6 * 256]),
                                       __uint_as_float(park[7 * 256]));
        f.accum[p.local] = acc;
        f.st0[p.local] = make_float4(ps.ray.o.x, ps.ray.o.y, ps.ray.o.z, ps.ray.d.x);
        f.st1[p.local] = make_float4(ps.ray.d.y, ps.ray.d.z, ps.color.r, ps.color.g);
        f.st2[p.local] = make_float2(ps.color.b, __uint_as_float((ps.material & 0xFFFFu) | (ps.depth << 16)));
        f.rgba8[p.local] = tonemap(col4{acc.x, acc.y, acc.z, acc.w}, cam.aperture, cam.exposure_time);
    }
    if (f.unit_cost && threadIdx.x == 0u) {  // what this unit's batch cost: the next launches start the expensive units first
        const unsigned long long dt = (__builtin_readcyclecounter() - t_start) >> 4;
        f.unit_cost[unit] = dt < 0x00FFFFFFull ? uint32_t(dt) : 0x00FFFFFFu;
    }
    flush_counters<COUNT>(f, p.active ? n_passes : 0u, cnt);
}

// ---- resident pipeline for scenes that are not staged in LDS (no lights): one WAVE takes its 64 pixels through all the passes ----
// The split pipeline puts a barrier over the whole shard behind every kernel of every pass, and a launch lasts as long as its slowest
// wave: the one whose rays graze the big mesh.  While the grid oversubscribes the chip that tail hides behind other workgroups; an
// eighth of a frame on each of 8 GPUs is ONE round of waves, and every pass then costs its slowest wave (config D, shard of 8: 250-490 us
// per pass against 130 us of work).  Here the passes of a wave follow each other without any barrier — a slow pass of one wave runs
// beside the fast passes of the others, and the launch lasts as long as the slowest SUM of passes.  Per pixel the arithmetic is that of
// the split kernels (cooperative front-to-back walk, then shade_segment), the accumulator grows by the same additions in the same order.
template <bool COUNT, int SHADING, int MINW, bool ONE_LEAF_WORLD = false>
__global__ void __launch_bounds__(64, MINW) rz_wave_batch_kernel(const DScene s, const DCamera cam, const DConfig cfg, const DFrame f, uint32_t n_passes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rz_lds[];
    const unsigned long long t_start = __builtin_readcyclecounter();
    const uint32_t unit = f.launch_order ? f.launch_order[blockIdx.x] : blockIdx.x;  // heaviest waves first (DFrame::launch_order)
    const uint32_t slot = unit * 64u + threadIdx.x;
    const PixelId p = pixel_of_local(f, cam, slot);
    if (!p.tile_inside) return;  // (the whole wave)
    Counters cnt;
    PathState ps;
    load_path<false>(f, cam, p, ps);
    float4 acc = p.active ? f.accum[p.local] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const uint32_t pass0 = *f.pass;
    for (uint32_t i = 0; i < n_passes; ++i) {
        if (i != 0u && p.active) {  // what load_path does with the state the previous pass stored
            ps.ray.d = normalized(ps.ray.d);
            ps.ray.near_ = 0.0f, ps.ray.far_ = RZ_FLT_MAX;
            if (ps.depth == 0u) ps.ray.near_ = cam.near_, ps.ray.far_ = cam.far_;
        }
        Hit hit;
        hit.instance = -1, hit.triangle = 0u, hit.bx = hit.by = 0.0f, hit.external = true;
        int found = 0;
        if (s.n_instances != 0u) found = closest_hit_coop<COUNT, RZ_TRACE_SHARED_RCP != 0, ONE_LEAF_WORLD>(s, CoopLds(rz_lds), p.active, ps.ray, hit, cnt);
        if (p.active) {
            col4 final_color;
            bool path_continues;
            shade_segment<COUNT, SHADING>(s, cam, cfg, p, ps, pass0 + i, found, hit, ShadowCtx{nullptr, TopCache{nullptr, nullptr, 0u}}, cnt, final_color, path_continues);
            acc = make_float4(acc.x + final_color.r, acc.y + final_color.g, acc.z + final_color.b, acc.w + float(!path_continues));
        }
    }
    if (p.active) {
        f.accum[p.local] = acc;
        f.st0[p.local] = make_float4(ps.ray.o.x, ps.ray.o.y, ps.ray.o.z, ps.ray.d.x);
        f.st1[p.local] = make_float4(ps.ray.d.y, ps.ray.d.z, ps.color.r, ps.color.g);
        f.st2[p.local] = make_float2(ps.color.b, __uint_as_float((ps.material & 0xFFFFu) | (ps.depth << 16)));
        f.rgba8[p.local] = tonemap(col4{acc.x, acc.y, acc.z, acc.w}, cam.aperture, cam.exposure_time);
    }
    if (f.unit_cost && threadIdx.x == 0u) {  // what this unit's batch cost: the next launches start the expensive units first
        const unsigned long long dt = (__builtin_readcyclecounter() - t_start) >> 4;
        f.unit_cost[unit] = dt < 0x00FFFFFFull ? uint32_t(dt) : 0x00FFFFFFu;
    }
    flush_counters<COUNT>(f, p.active ? n_passes : 0u, cnt);
}

// ---- split pipeline ----
// hit record: hit0 = (far, b1, b2, bits(triangle)), hit1 = instance | found << 29 | external << 31
template <bool FIRST, bool COUNT, int MODE, bool LDS_SCENE>
__global__ void __launch_bounds__(256, RZ_TRACE_MIN_WAVES) rz_trace_kernel(const DScene scene_in, const DCamera cam, const DFrame f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rz_lds[];
    DScene s = scene_in;
    unsigned char* workspace = rz_lds + stage_scene<LDS_SCENE>(s, rz_lds);
    uint32_t* lds_column = stack_column<MODE>(workspace);
    const PixelId p = pixel_of_thread(f, cam, blockIdx.x, threadIdx.x);  // scenes staged in LDS are never reordered
    Counters cnt;
    Ray ray;
    {
        PathState ps;
        load_path<FIRST>(f, cam, p, ps);
        ray = ps.ray;
    }
    Hit hit;
    const int found = trace_path<MODE, COUNT, RZ_TRACE_SHARED_RCP != 0>(s, workspace, lds_column, p.active, ray, hit, cnt);
    if (p.active) {
        f.hit0[p.local] = make_float4(ray.far_, hit.bx, hit.by, __uint_as_float(hit.triangle));
        f.hit1[p.local] = (uint32_t(hit.instance) & 0x1FFFFFFFu) | (uint32_t(found) << 29) | (hit.external ? 0x80000000u : 0u);
    }
    flush_counters<COUNT>(f, 0u, cnt);
}

// The ray a thread of the single-wave trace kernels walks: in sorted order thread i takes the ray of local pixel perm[i] (the hit
// record still goes to that pixel's slot).  The three 16-byte gathers per ray hide well behind the walk: gathering the rays into a
// contiguous stream first (a kernel of its own after the sort) took 108 us per pass on config C and saved the walk 21.
template <bool FIRST>
RZ_DEV PixelId trace_ray_of_slot(const DFrame& f, const DCamera& cam, uint32_t slot, Ray& ray) {
    const PixelId p = pixel_of_local(f, cam, (!FIRST && f.perm) ? f.perm[slot] : slot);
    PathState ps;
    load_path<FIRST>(f, cam, p, ps);
    ray = ps.ray;
    return p;
}

// MODE 3 trace kernel, one wave per workgroup.  A workgroup's registers and LDS stay allocated until its LAST wave ends and
// a wave lasts as long as its slowest ray, so with heavy-tailed ray costs single-wave workgroups give their slots back sooner
// (config D 3 378 -> 3 093 us, C 974 -> 910 us against 256 threads); the price is a smaller share of LDS for the tree-top cache
// (top_n nodes per workgroup).  MINW = waves per SIMD the register budget is cut for: big trees are bound by the latency of
// their node fetches and want occupancy (D: 6 waves 2 959 us, 4 waves 3 370 us), trees that live in L2 / LDS want registers
// (C: 4 waves 879 us, 6 waves 984 us).
template <bool FIRST, bool COUNT, int MINW>
__global__ void __launch_bounds__(64, MINW) rz_trace_skip_kernel(const DScene s, const DCamera cam, const DFrame f, uint32_t top_n) {
    constexpr int WG = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char rz_lds[];
    float4* ln = reinterpret_cast<float4*>(rz_lds);
    uint32_t* ls = reinterpret_cast<uint32_t*>(rz_lds + top_n * 32u);
    for (uint32_t i = threadIdx.x; i < 2u * top_n; i += uint32_t(WG)) ln[i] = s.nodes[i];
    for (uint32_t i = threadIdx.x; i < top_n; i += uint32_t(WG)) ls[i] = s.node_skip[i];
    const uint32_t slot = blockIdx.x * uint32_t(WG) + threadIdx.x;
    Counters cnt;
    Ray ray;
    const PixelId p = trace_ray_of_slot<FIRST>(f, cam, slot, ray);
    Hit hit;
    hit.instance = -1, hit.triangle = 0u, hit.bx = hit.by = 0.0f, hit.external = true;
    int found = 0;
    if (p.active && s.n_instances != 0u) {
        const TopCache top{ln, ls, top_n};
        found = closest_hit_skip<COUNT, RZ_TRACE_SHARED_RCP != 0>(s, top, ray, hit, cnt);
    }
    if (p.active) {
        f.hit0[p.local] = make_float4(ray.far_, hit.bx, hit.by, __uint_as_float(hit.triangle));
        f.hit1[p.local] = (uint32_t(hit.instance) & 0x1FFFFFFFu) | (uint32_t(found) << 29) | (hit.external ? 0x80000000u : 0u);
    }
    flush_counters<COUNT>(f, 0u, cnt);
}

// The front-to-back walk with the cooperative triangle phase (hiprz_device.hpp: closest_hit_coop): one wave per workgroup, all
// 64 lanes go through the walk together (a lane without a ray only helps with other lanes' triangles).
// ONE_LEAF_WORLD: the world tree is one leaf (hiprz_device.hpp: closest_hit_coop<..., ONE_STEP>).
template <bool FIRST, bool COUNT, int MINW, bool ONE_LEAF_WORLD = false>
__global__ void __launch_bounds__(64, MINW) rz_trace_coop_kernel(const DScene s, const DCamera cam, const DFrame f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rz_lds[];
    const uint32_t slot = blockIdx.x * 64u + threadIdx.x;
    Counters cnt;
    Ray ray;
    const PixelId p = trace_ray_of_slot<FIRST>(f, cam, slot, ray);
    Hit hit;
    hit.instance = -1, hit.triangle = 0u, hit.bx = hit.by = 0.0f, hit.external = true;
    int found = 0;
    if (s.n_instances != 0u) found = closest_hit_coop<COUNT, RZ_TRACE_SHARED_RCP != 0, ONE_LEAF_WORLD>(s, CoopLds(rz_lds), p.active, ray, hit, cnt);
    if (p.active) {
        f.hit0[p.local] = make_float4(ray.far_, hit.bx, hit.by, __uint_as_float(hit.triangle));
        f.hit1[p.local] = (uint32_t(hit.instance) & 0x1FFFFFFFu) | (uint32_t(found) << 29) | (hit.external ? 0x80000000u : 0u);
    }
    flush_counters<COUNT>(f, 0u, cnt);
}

// The same walk for the CUDA-compat integrator (hiprz_set_mode): with HIPRZ_COMPAT_SCATTERING the medium the ray travels in may end the
// segment before any surface does (Material::applyScattering, cuda_material.cuh:141-159) — the distance is the FIRST draw of the
// segment's random stream (cuda_world.cuh:91-100), taken here as the walk's range and taken again by the shade kernel.
template <bool FIRST, bool COUNT>
__global__ void __launch_bounds__(64, 4) rz_trace_coop_compat_kernel(const DScene s, const DCamera cam, const DConfig cfg, const DFrame f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rz_lds[];
    const uint32_t slot = blockIdx.x * 64u + threadIdx.x;
    Counters cnt;
    const PixelId p = pixel_of_local(f, cam, (!FIRST && f.perm) ? f.perm[slot] : slot);
    PathState ps;
    load_path<FIRST>(f, cam, p, ps);
    bool scattered = false;
    if (p.active && (cfg.flags & HIPRZ_COMPAT_SCATTERING)) {
        const float sigma = material_scattering(s, ps.material);
        if (sigma > 1.0e-4f) {
            const uint32_t pass = FIRST ? 0u : *f.pass, pixel_idx = p.y * cam.width + p.x;
            Rng rng(float(p.x) / float(cam.width), float(p.y) / float(cam.height), seed_value(cfg.seed, pass, (pixel_idx + ps.depth) & 255u));
            const float distance = (-logf(rng.unsignedUniform() + 1.0e-4f)) / sigma;
            if (distance < ps.ray.far_) ps.ray.far_ = distance, scattered = true;
        }
    }
    Hit hit;
    hit.instance = -1, hit.triangle = 0u, hit.bx = hit.by = 0.0f, hit.external = true;
    int found = 0;
    if (s.n_instances != 0u) found = closest_hit_coop<COUNT, RZ_TRACE_SHARED_RCP != 0>(s, CoopLds(rz_lds), p.active, ps.ray, hit, cnt);
    if (scattered && found != 2) found = 3;
    if (p.active) {
        f.hit0[p.local] = make_float4(ps.ray.far_, hit.bx, hit.by, __uint_as_float(hit.triangle));
        f.hit1[p.local] = (uint32_t(hit.instance) & 0x1FFFFFFFu) | (uint32_t(found) << 29) | (hit.external ? 0x80000000u : 0u);
    }
    flush_counters<COUNT>(f, 0u, cnt);
}

// SHADOW: the shadow-ray walk — 1 = nested loops with the per-lane LDS stack (scenes staged in LDS), 3 = skip links with the
// tree tops staged in LDS instead of a stack (everything else; `top_n` nodes).
template <bool FIRST, bool COUNT, bool LDS_SCENE, int SHADOW>
__global__ void __launch_bounds__(256, RZ_MIN_WAVES) rz_shade_kernel(const DScene scene_in, const DCamera cam, const DConfig cfg, const DFrame f, uint32_t top_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rz_lds[];
    DScene s = scene_in;
    unsigned char* workspace = rz_lds + stage_scene<LDS_SCENE>(s, rz_lds);
    ShadowCtx shadow{stack_column<1>(workspace), TopCache{nullptr, nullptr, 0u}};
    if constexpr (shadow_mode_defers(SHADOW) || shadow_mode_compat(SHADOW)) {
        shadow.lds_column = nullptr;
    }
    if constexpr (SHADOW == 3) {
        float4* ln = reinterpret_cast<float4*>(workspace);
        uint32_t* ls = reinterpret_cast<uint32_t*>(workspace + top_n * 32u);
        for (uint32_t i = threadIdx.x; i < 2u * top_n; i += 256u) ln[i] = s.nodes[i];
        for (uint32_t i = threadIdx.x; i < top_n; i += 256u) ls[i] = s.node_skip[i];
        __syncthreads();
        shadow.lds_column = nullptr;
        shadow.top = TopCache{ln, ls, top_n};
    }
    const PixelId p = pixel_of_thread(f, cam, blockIdx.x, threadIdx.x);
    Counters cnt;
    if (p.active) {
        PathState ps;
        load_path<FIRST>(f, cam, p, ps);
        const float4 h0 = f.hit0[p.local];
        const uint32_t h1 = f.hit1[p.local];
        Hit hit;
        const int found = int((h1 >> 29) & 3u);  // 3 (compat): the medium scattered the ray before it met a surface
        ps.ray.far_ = h0.x;
        hit.bx = h0.y, hit.by = h0.z, hit.triangle = __float_as_uint(h0.w);
        hit.instance = found == 2 ? int32_t(h1 & 0x1FFFFFFFu) : -1;
        hit.external = (h1 & 0x80000000u) != 0u;
        shadow.nee = f.nee + size_t(p.local) * f.nee_quads;
        if constexpr (shadow_mode_compat(SHADOW)) {
            // the CUDA engine's stream of draws: the scattering distance of the medium first (the compat trace kernel drew the same number
            // for the same pixel, pass and depth), then whatever the shading draws
            const uint32_t pass = FIRST ? 0u : *f.pass, pixel_idx = p.y * cam.width + p.x;
            Rng rng(float(p.x) / float(cam.width), float(p.y) / float(cam.height), seed_value(cfg.seed, pass, (pixel_idx + ps.depth) & 255u));
            if ((cfg.flags & HIPRZ_COMPAT_SCATTERING) && material_scattering(s, ps.material) > 1.0e-4f) (void)rng.unsignedUniform();
            shade_and_store<FIRST, COUNT, SHADOW>(s, cam, cfg, f, p, ps, found, hit, shadow, cnt, &rng);
        } else {
            shade_and_store<FIRST, COUNT, SHADOW>(s, cam, cfg, f, p, ps, found, hit, shadow, cnt);
        }
    } else if (f.sort_key && p.local < f.n_local_tiles * 256u) {
        f.sort_key[p.local] = 0x00FFFFFFu;  // slots outside the frame sort to the end
        if (f.shadow_key) f.shadow_key[p.local] = 0x00FFFFFFu;
    }
    flush_counters<COUNT>(f, p.active ? 1u : 0u, cnt);
}

// The shadow rays of a pass, deferred by rz_shade_kernel<..., RZ_SHADOW_DEFER>: anyIntersection (cpu_engine_kernel.cpp:398-481)
// for every sample slot that holds a ray, then the sums of directLightSampling / spotLightSampling (:742-743, :789-790),
// `final += (direct * ray_color) * lerp(1, colour, metalness)` (:160-165) and the accumulation of renderFirstPass /
// renderCumulativePass (:42-45, :82-86), all in the order the inline path has them.  One wave per workgroup, tree tops in
// LDS, packed box test: the walk runs at the trace kernel's occupancy instead of the shading kernel's 128 VGPRs.
template <bool FIRST, bool COUNT, int MINW>
__global__ void __launch_bounds__(64, MINW) rz_shadow_kernel(const DScene s, const DCamera cam, const DConfig cfg, const DFrame f, uint32_t top_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rz_lds[];
    float4* ln = reinterpret_cast<float4*>(rz_lds);
    uint32_t* ls = reinterpret_cast<uint32_t*>(rz_lds + top_n * 32u);
    for (uint32_t i = threadIdx.x; i < 2u * top_n; i += 64u) ln[i] = s.nodes[i];
    for (uint32_t i = threadIdx.x; i < top_n; i += 64u) ls[i] = s.node_skip[i];
    const uint32_t slot = blockIdx.x * 64u + threadIdx.x;
    const uint32_t* order = f.shadow_perm ? f.shadow_perm : f.perm;
    const PixelId p = pixel_of_local(f, cam, order ? order[slot] : slot);
    Counters cnt;
    if (p.active) {
        const ShadowCtx sc{nullptr, TopCache{ln, ls, top_n}};
        const float4* rec = f.nee + size_t(p.local) * f.nee_quads;
        const float4 base = rec[0];
        const uint32_t bits = __float_as_uint(base.w);
        const bool path_continues = (bits & 1u) != 0u;
        col4 final_color{base.x, base.y, base.z, 0.0f};
        if (bits & 2u) {
            const uint32_t mask = bits >> 2;
            const float4 o = rec[1];
            auto shadowed_sum = [&](uint32_t first, uint32_t count) {
                col4 total = splat(0.0f);
                for (uint32_t k = first; k < first + count; ++k) {
                    if (!(mask & (1u << k))) continue;
                    const float4 d = rec[4u + 2u * k], t = rec[5u + 2u * k];
                    Ray sr;
                    sr.o = V3(o.x, o.y, o.z), sr.d = V3(d.x, d.y, d.z), sr.near_ = 0.0f, sr.far_ = d.w;
                    const col4 V_PL = splat(any_hit<3, COUNT>(s, sc, sr, cnt));
                    total = total + (col4{t.x, t.y, t.z, t.w} * V_PL) * V_PL.a;
                }
                return total;
            };
            col4 direct_total = splat(0.0f), spot_total = splat(0.0f);
            if (s.n_direct_lights != 0u) direct_total = div_scalar(shadowed_sum(0u, cfg.direct_samples), float(cfg.direct_samples) / float(s.n_direct_lights));
            if (s.n_spot_lights != 0u) spot_total = div_scalar(shadowed_sum(cfg.direct_samples, cfg.spot_samples), float(cfg.spot_samples) / float(s.n_spot_lights));
            const col4 direct = direct_total + spot_total;
            const float4 a = rec[2], b = rec[3];
            final_color = final_color + (direct * col4{a.x, a.y, a.z, a.w}) * col4{b.x, b.y, b.z, b.w};
        }
        col4 value;
        if constexpr (FIRST) {
            value = col4{final_color.r, final_color.g, final_color.b, float(!path_continues)};
        } else {
            const float4 acc = f.accum[p.local];
            value = col4{acc.x + final_color.r, acc.y + final_color.g, acc.z + final_color.b, acc.w + float(!path_continues)};
        }
        f.accum[p.local] = make_float4(value.r, value.g, value.b, value.a);
    }
    flush_counters<COUNT>(f, 0u, cnt);
}

// rz_shadow_kernel with the cooperative any-hit walk (hiprz_device.hpp: any_hit_coop): the sample loop is wave-uniform, a lane
// whose pixel has no shadow ray in slot k walks along as a helper.  Sums, order and accumulation are those of rz_shadow_kernel.
// MASK: the coloured shadow masks of the CUDA engine (HIPRZ_COMPAT_SHADOW_COLOR) — V_PL is the product of the crossed triangles' opacity
// colours instead of 0 / 1; everything else as above.
template <bool FIRST, bool COUNT, int MINW, bool MASK = false>
__global__ void __launch_bounds__(64, MINW) rz_shadow_coop_kernel(const DScene s, const DCamera cam, const DConfig cfg, const DFrame f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rz_lds[];
    const CoopLds lds(rz_lds);
    const uint32_t slot = blockIdx.x * 64u + threadIdx.x;
    const uint32_t* order = f.shadow_perm ? f.shadow_perm : f.perm;
    const PixelId p = pixel_of_local(f, cam, order ? order[slot] : slot);
    Counters cnt;
    float4 base = make_float4(0.0f, 0.0f, 0.0f, 0.0f), o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t bits = 0u;
    const float4* rec = f.nee + size_t(p.local) * f.nee_quads;
    if (p.active) {
        base = rec[0];
        bits = __float_as_uint(base.w);
        if (bits & 2u) o = rec[1];
    }
    const uint32_t mask = (bits & 2u) ? bits >> 2 : 0u;
    col4 direct_total = splat(0.0f), spot_total = splat(0.0f);
    const uint32_t n_samples = cfg.direct_samples + cfg.spot_samples;
    for (uint32_t k = 0u; k < n_samples; ++k) {  // wave-uniform
        const bool has = (mask & (1u << k)) != 0u;
        if (!__any(has)) continue;
        float4 d = make_float4(0.0f, 0.0f, 1.0f, 0.0f), t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (has) d = rec[4u + 2u * k], t = rec[5u + 2u * k];
        Ray sr;
        sr.o = V3(o.x, o.y, o.z), sr.d = V3(d.x, d.y, d.z), sr.near_ = 0.0f, sr.far_ = d.w;
        if (has) { RZ_COUNT(shadow_rays); }
        col4 V_PL = splat(MASK ? 1.0f : 0.0f);
        if (s.n_instances != 0u) V_PL = any_hit_coop_mask<COUNT, RZ_SHADE_SHARED_RCP != 0, MASK>(s, lds, has, sr, MASK && (cfg.flags & HIPRZ_COMPAT_FILTERING) != 0u, cnt);
        if (has) {
            const col4 term = (col4{t.x, t.y, t.z, t.w} * V_PL) * V_PL.a;
            if (k < cfg.direct_samples) direct_total = direct_total + term;
            else spot_total = spot_total + term;
        }
    }
    if (p.active) {
        const bool path_continues = (bits & 1u) != 0u;
        col4 final_color{base.x, base.y, base.z, 0.0f};
        if (bits & 2u) {
            col4 dt = splat(0.0f), st = splat(0.0f);
            if (s.n_direct_lights != 0u) dt = div_scalar(direct_total, float(cfg.direct_samples) / float(s.n_direct_lights));
            if (s.n_spot_lights != 0u) st = div_scalar(spot_total, float(cfg.spot_samples) / float(s.n_spot_lights));
            const col4 direct = dt + st;
            const float4 a = rec[2], b = rec[3];
            final_color = final_color + (direct * col4{a.x, a.y, a.z, a.w}) * col4{b.x, b.y, b.z, b.w};
        }
        col4 value;
        if constexpr (FIRST) {
            value = col4{final_color.r, final_color.g, final_color.b, float(!path_continues)};
        } else {
            const float4 acc = f.accum[p.local];
            value = col4{acc.x + final_color.r, acc.y + final_color.g, acc.z + final_color.b, acc.w + float(!path_continues)};
        }
        f.accum[p.local] = make_float4(value.r, value.g, value.b, value.a);
    }
    flush_counters<COUNT>(f, 0u, cnt);
}


// rz_shadow_coop_kernel with the wave-level walk (hiprz_device.hpp: any_hit_packet): the shadow rays' sorted order hands a wave 64 rays
// from one cell towards one light.  Sums, order and accumulation are those of rz_shadow_kernel.
template <bool FIRST, bool COUNT, int MINW, bool MASK = false>
__global__ void __launch_bounds__(64, MINW) rz_shadow_packet_kernel(const DScene s, const DCamera cam, const DConfig cfg, const DFrame f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rz_lds[];  // 2 KiB: the rays of a leaf's triangle phase (MASK: + 1 KiB, the crossed triangles' colours)
    const uint32_t slot = blockIdx.x * 64u + threadIdx.x;
    const uint32_t* order = f.shadow_perm ? f.shadow_perm : f.perm;
    const PixelId p = pixel_of_local(f, cam, order ? order[slot] : slot);
    Counters cnt;
    float4 base = make_float4(0.0f, 0.0f, 0.0f, 0.0f), o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t bits = 0u;
    const float4* rec = f.nee + size_t(p.local) * f.nee_quads;
    if (p.active) {
        base = rec[0];
        bits = __float_as_uint(base.w);
        if (bits & 2u) o = rec[1];
    }
    const uint32_t mask = (bits & 2u) ? bits >> 2 : 0u;
    col4 direct_total = splat(0.0f), spot_total = splat(0.0f);
    const uint32_t n_samples = cfg.direct_samples + cfg.spot_samples;
    for (uint32_t k = 0u; k < n_samples; ++k) {  // wave-uniform
        const bool has = (mask & (1u << k)) != 0u;
        if (!__any(has)) continue;
        float4 d = make_float4(0.0f, 0.0f, 1.0f, 0.0f), t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (has) d = rec[4u + 2u * k], t = rec[5u + 2u * k];
        Ray sr;
        sr.o = V3(o.x, o.y, o.z), sr.d = V3(d.x, d.y, d.z), sr.near_ = 0.0f, sr.far_ = d.w;
        col4 V_PL = splat(MASK ? 1.0f : 0.0f);
        if (s.n_instances != 0u) V_PL = any_hit_packet<COUNT, RZ_SHADE_SHARED_RCP != 0, MASK>(s, (RZ_LDS f4*)rz_lds, has, sr, MASK && (cfg.flags & HIPRZ_COMPAT_FILTERING) != 0u, cnt);
        else if (has) { RZ_COUNT(shadow_rays); }
        if (has) {
            const col4 term = (col4{t.x, t.y, t.z, t.w} * V_PL) * V_PL.a;
            if (k < cfg.direct_samples) direct_total = direct_total + term;
            else spot_total = spot_total + term;
        }
    }
    if (p.active) {
        const bool path_continues = (bits & 1u) != 0u;
        col4 final_color{base.x, base.y, base.z, 0.0f};
        if (bits & 2u) {
            col4 dt = splat(0.0f), st = splat(0.0f);
            if (s.n_direct_lights != 0u) dt = div_scalar(direct_total, float(cfg.direct_samples) / float(s.n_direct_lights));
            if (s.n_spot_lights != 0u) st = div_scalar(spot_total, float(cfg.spot_samples) / float(s.n_spot_lights));
            const col4 direct = dt + st;
            const float4 a = rec[2], b = rec[3];
            final_color = final_color + (direct * col4{a.x, a.y, a.z, a.w}) * col4{b.x, b.y, b.z, b.w};
        }
        col4 value;
        if constexpr (FIRST) {
            value = col4{final_color.r, final_color.g, final_color.b, float(!path_continues)};
        } else {
            const float4 acc = f.accum[p.local];
            value = col4{acc.x + final_color.r, acc.y + final_color.g, acc.z + final_color.b, acc.w + float(!path_continues)};
        }
        f.accum[p.local] = make_float4(value.r, value.g, value.b, value.a);
    }
    flush_counters<COUNT>(f, 0u, cnt);
}

// ---- CUDA-compat mode (hiprz_set_mode, hiprz_compat.hpp): one fused kernel per pass, scene in global memory ----
// Order of the random draws as in the CUDA engine: the medium's scattering distance first (cuda_material.cuh:146-148), then
// whatever the shading draws.
template <bool FIRST, bool COUNT>
__global__ void __launch_bounds__(256) rz_compat_pass_kernel(const DScene s, const DCamera cam, const DConfig cfg, const DFrame f) {
    const PixelId p = pixel_of_thread(f, cam, blockIdx.x, threadIdx.x);
    Counters cnt;
    if (p.active) {
        PathState ps;
        load_path<FIRST>(f, cam, p, ps);
        const uint32_t pass = FIRST ? 0u : *f.pass, pixel_idx = p.y * cam.width + p.x;
        Rng rng(float(p.x) / float(cam.width), float(p.y) / float(cam.height), seed_value(cfg.seed, pass, (pixel_idx + ps.depth) & 255u));
        bool scattered = false;
        if (cfg.flags & HIPRZ_COMPAT_SCATTERING) {  // Material::applyScattering of the medium the ray travels in
            const float sigma = material_scattering(s, ps.material);
            if (sigma > 1.0e-4f) {
                const float distance = (-logf(rng.unsignedUniform() + 1.0e-4f)) / sigma;
                if (distance < ps.ray.far_) ps.ray.far_ = distance, scattered = true;
            }
        }
        Hit hit;
        hit.instance = -1, hit.triangle = 0u, hit.bx = hit.by = 0.0f, hit.external = true;
        int found = 0;
        if (s.n_instances != 0u) found = closest_hit_skip<COUNT, false>(s, TopCache{nullptr, nullptr, 0u}, ps.ray, hit, cnt);
        if (scattered && found != 2) found = 3;
        shade_and_store<FIRST, COUNT, RZ_SHADOW_COMPAT>(s, cam, cfg, f, p, ps, found, hit, ShadowCtx{nullptr, TopCache{nullptr, nullptr, 0u}}, cnt, &rng);
    }
    flush_counters<COUNT>(f, p.active ? 1u : 0u, cnt);
}

}  // namespace hiprz
