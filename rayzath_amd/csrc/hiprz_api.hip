// hiprz_api.hip — kernels + the device half of the C-ABI declared in include/hiprz.h.
//
// Replaces, for the HIPGPU backend, what the reference's CUDA backend does in
// cuda_engine_core.cu (host<->device mirroring, readback), cuda_engine_renderer.cu
// (launch sequence) and cuda_render_kernel.cu / cuda_postprocess_kernel.cu (kernels).
// Written for gfx950 only: wave64, 256-thread workgroups = one 32x8-pixel tile.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "hiprz.h"
#include "hiprz_device.hpp"

using namespace hiprz;

// =======================================================================================
// Kernels
// =======================================================================================

// One pass = one path segment per owned pixel: renderFirstPass (cpu_engine_kernel.cpp:15-57)
// when FIRST, else renderCumulativePass (:58-101), with traceRay (:113-178) inlined.
//
// The pass is written as three pieces — load_path, the closest-hit walk, shade_and_store — used by two
// pipelines that give identical results:
//   fused  (rz_pass_kernel):   all three in one kernel; state + accumulator cross HBM once (112 B/pixel).
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

// pool = local pixel slots [pool_begin, pool_end) (through f.perm when rays are sorted); results go to f.hit0/hit1.
template <bool FIRST, bool COUNT>
__device__ __forceinline__ void trace_persistent(const DScene& s, const WalkTop& top, const DCamera& cam, const DFrame& f,
                                                 uint32_t* pool_next, uint32_t pool_begin, uint32_t pool_end, Counters& cnt) {
    const bool scene_fast = s.fast_div != 0u;
    // per-ray state
    bool has_ray = false, pool_empty = false;
    uint32_t pixel = 0u;
    WalkRay cur;
    cur.o = cur.d = cur.y = V3(0.0f, 0.0f, 1.0f), cur.near_ = cur.far_ = 0.0f, cur.fast = false;
    v3 world_o = cur.o, world_d = cur.d;
    float world_near = 0.0f, world_far = 0.0f, len = 1.0f;
    uint32_t n = RZ_END, ret = RZ_END, inst = 0u;
    bool in_mesh = false, found_here = false, root_missed = false;
    Hit hit;
    hit.instance = -1, hit.triangle = 0u, hit.bx = hit.by = 0.0f, hit.external = true;
    uint32_t guard = 0u;

    while (true) {
        // ---- refill ----
        if (!has_ray && !pool_empty) {
            const uint32_t slot = pool_begin + atomicAdd(pool_next, 1u);
            if (slot >= pool_end) {
                pool_empty = true;
            } else {
                const PixelId p = pixel_of_local(f, cam, (!FIRST && f.perm) ? f.perm[slot] : slot);
                pixel = p.local;
                if (p.active) {
                    PathState ps;
                    load_path<FIRST>(f, cam, p, ps);
                    cur.o = ps.ray.o, cur.d = ps.ray.d, cur.near_ = ps.ray.near_, cur.far_ = ps.ray.far_;
                    prepare<true>(cur, scene_fast);
                    world_o = cur.o, world_d = cur.d, world_near = cur.near_, world_far = cur.far_;
                    hit.instance = -1, hit.triangle = 0u, hit.bx = hit.by = 0.0f, hit.external = true;
                    in_mesh = false, found_here = false, root_missed = false;
                    n = s.n_instances ? s.tlas_root : RZ_END;
                    if (s.n_instances == 0u) root_missed = true;
                    has_ray = true;
                }
            }
        }
        if (!__any(has_ray)) {
            if (__all(pool_empty)) break;
            continue;
        }
        RZ_GUARD(guard);

        // ---- node phase: step until this lane holds a leaf or its ray has ended ----
        uint32_t leaf_begin = 0u, leaf_end = 0u;
        while (has_ray && leaf_end == leaf_begin) {
            if (n == RZ_END) {
                if (in_mesh) {  // leave the instance: cpu_engine_kernel.cpp:320-329
                    if (found_here) {
                        hit.instance = int32_t(inst);
                        world_near = cur.near_ / len;
                        world_far = cur.far_ / len;
                    }
                    cur.o = world_o, cur.d = world_d, cur.near_ = world_near, cur.far_ = world_far;
                    prepare<true>(cur, scene_fast);
                    in_mesh = false;
                    n = ret;
                    continue;
                }
                // the ray's walk is complete: publish the hit record (rz_trace_kernel's layout)
                const int found = root_missed ? 0 : (hit.instance >= 0 ? 2 : 1);
                f.hit0[pixel] = make_float4(world_far, hit.bx, hit.by, __uint_as_float(hit.triangle));
                f.hit1[pixel] = (uint32_t(hit.instance) & 0x1FFFFFFFu) | (uint32_t(found) << 29) | (hit.external ? 0x80000000u : 0u);
                has_ray = false;
                break;
            }
            float4 n0, n1;
            uint32_t link;
            fetch_walk_node(s, top, n, n0, n1, link);
            RZ_COUNT(box_tests);
            if (box_hit<true>(n0, n1, cur)) {
                const uint32_t a = __float_as_uint(n1.z), meta = __float_as_uint(n1.w);
                const uint32_t type = meta >> RZ_WALK_TYPE_SHIFT;
                if (type == RZ_WALK_INNER || type == RZ_WALK_CHAIN) {
                    n = a;
                } else if (type == RZ_WALK_INSTANCE) {  // enter: cpu_engine_kernel.cpp:307-319
                    inst = a;
                    const InstanceXform x = load_instance_xform(s, inst);
                    world_o = cur.o, world_d = cur.d, world_near = cur.near_, world_far = cur.far_;
                    cur.o = transform_backward(x.xa, x.ya, x.za, cur.o - x.position);
                    cur.d = transform_backward(x.xa, x.ya, x.za, cur.d);
                    if (!x.unit_scale) {
                        cur.o = cur.o / x.scale;
                        cur.d = cur.d / x.scale;
                    }
                    len = magnitude(cur.d);
                    cur.near_ = cur.near_ * len;
                    cur.far_ = cur.far_ * len;
                    cur.d = cur.d * (1.0f / len);
                    prepare<true>(cur, scene_fast);
                    in_mesh = true, found_here = false;
                    ret = link;
                    n = x.blas_root;
                } else {  // triangle leaf: hold it
                    leaf_begin = a, leaf_end = a + (meta & RZ_WALK_COUNT_MASK);
                    n = link;
                }
            } else {
                if (n == s.tlas_root) root_missed = true;  // cpu_engine_kernel.cpp:283
                n = link;
            }
        }

        // ---- leaf phase ----
        for (uint32_t i = leaf_begin; i < leaf_end; ++i) {
            const float4 ta = s.tris[3 * i], tb = s.tris[3 * i + 1], tc = s.tris[3 * i + 2];
            float t, b1, b2, det;
            RZ_COUNT(tri_tests);
            if (tri_hit(xyz(ta), xyz(tb), xyz(tc), cur, t, b1, b2, det)) {
                cur.far_ = t;
                hit.triangle = i;
                hit.external = det > 0.0f;
                hit.bx = b1, hit.by = b2;
                found_here = true;
            }
        }
    }
}

// closest hit of the segment with the selected walk; MODE 2 must be reached by all 256 threads
template <int MODE, bool COUNT, bool RCP>
RZ_DEV int trace_path(const DScene& s, unsigned char* workspace, uint32_t* lds_column, bool active, Ray& ray, Hit& hit, Counters& cnt) {
    if constexpr (MODE == 2) {
        return closest_hit_binned<COUNT, RCP>(s, workspace, active, ray, hit, cnt);
    } else if constexpr (MODE == 3) {  // workspace = [top nodes][top links], staged here by the whole workgroup
        float4* ln = reinterpret_cast<float4*>(workspace);
        uint32_t* ls = reinterpret_cast<uint32_t*>(workspace + s.top_count * 32u);
        for (uint32_t i = threadIdx.x; i < 2u * s.top_count; i += 256u) ln[i] = s.nodes[i];
        for (uint32_t i = threadIdx.x; i < s.top_count; i += 256u) ls[i] = s.node_skip[i];
        __syncthreads();
        hit.instance = -1, hit.triangle = 0, hit.bx = hit.by = 0.0f, hit.external = true;
        if (!active || s.n_instances == 0) return 0;
        const TopCache top{ln, ls, s.top_count};
        return closest_hit_skip<COUNT, RCP>(s, top, ray, hit, cnt);
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
                          int found, const Hit& hit, const ShadowCtx& lds_column, Counters& cnt, col4& final_color, bool& path_continues) {
    Ray& ray = ps.ray;
    col4& ray_color = ps.color;
    uint32_t& ray_material = ps.material;
    uint32_t& depth = ps.depth;
    const uint32_t pixel_idx = p.y * cam.width + p.x;
    Rng rng(float(p.x) / float(cam.width), float(p.y) / float(cam.height), seed_value(cfg.seed, pass, (pixel_idx + depth) & 255u));

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
        analyze_intersection<COUNT, TEX>(s, hit, sf, m, cnt);
    } else {
        m = load_material(s, HIPRZ_MATERIAL_WORLD);
        if (TEX && found == 1) {  // texcrd of the sky sphere (cpu_engine_kernel.cpp:292-295); only a map reads it
            sf.u = -(0.5f + (RZ_ATAN2F(ray.d.z, ray.d.x) / (RZ_PI_F * 2.0f)));
            sf.v = 0.5f + (RZ_ASINF(ray.d.y) / RZ_PI_F);
        }
    }
    sf.surface_scattering = m.scattering;
    // fetchColor / fetchEmission (:505-512, 523-528)
    sf.color = from_u8(m.color);
    if (TEX && m.texture >= 0) sf.color = fetch_rgba8<COUNT>(s, m.texture, sf.u, sf.v, cnt);
    sf.color.a = 1.0f - sf.color.a;
    sf.emission = TEX && m.emission_map >= 0 ? fetch_r32f<COUNT>(s, m.emission_map, sf.u, sf.v, cnt) : m.emission;
    if (sf.emission > 0.0f) final_color = final_color + (ray_color * sf.color) * sf.emission;

    v3 point = V3(0.0f, 0.0f, 0.0f), next_direction = V3(0.0f, 0.0f, 0.0f);
    if (found != 2) {
        depth = 255u;  // TracingState::endPath
    } else {
        RZ_COUNT(hits);
        depth += 1u;
        sf.metalness = TEX && m.metalness_map >= 0 ? fetch_r8<COUNT>(s, m.metalness_map, sf.u, sf.v, cnt) : m.metalness;
        sf.roughness = TEX && m.roughness_map >= 0 ? fetch_r8<COUNT>(s, m.roughness_map, sf.u, sf.v, cnt) : m.roughness;
        sf.fresnel = fresnel_specular_ratio(sf.mapped_normal, ray.d, material_ior(s, ray_material), material_ior(s, sf.behind_material),
                                            sf.refr_x, sf.refr_y);
        sf.reflectance = lerpf(sf.fresnel, 1.0f, sf.metalness);

        next_direction = sample_direction(ray.d, ray_material, sf, rng);
        point = (ray.o + ray.d * ray.far_) + sf.normal * (0.0001f * ray.far_);

        const col4 direct = direct_illumination<SHADOW, COUNT>(s, cfg, lds_column, ray.d, ray_material, point, next_direction, sf, rng, cnt);
        if constexpr (SHADOW == RZ_SHADOW_DEFER) {  // rz_shadow_kernel adds (direct * a) * b once it knows the shadow masks
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
                            int found, const Hit& hit, const ShadowCtx& lds_column, Counters& cnt) {
    const float hit_distance = ps.ray.far_;
    col4 final_color;
    bool path_continues;
    shade_segment<COUNT, SHADOW>(s, cam, cfg, p, ps, FIRST ? 0u : *f.pass, found, hit, lds_column, cnt, final_color, path_continues);
    const Ray& ray = ps.ray;
    const col4& ray_color = ps.color;
    const uint32_t ray_material = ps.material, depth = ps.depth;

    // ---- accumulate ----
    if constexpr (FIRST) f.depth[p.local] = hit_distance;
    if constexpr (SHADOW == RZ_SHADOW_DEFER) {
        // the radiance so far + what rz_shadow_kernel needs to finish it; it also does the accumulation
        const uint32_t bits = (path_continues ? 1u : 0u) | (lds_column.defer_done ? 2u : 0u) | (lds_column.defer_mask << 2);
        f.nee_base[p.local] = make_float4(final_color.r, final_color.g, final_color.b, __uint_as_float(bits));
        if (lds_column.defer_done) {
            f.nee_a[p.local] = make_float4(lds_column.defer_a.r, lds_column.defer_a.g, lds_column.defer_a.b, lds_column.defer_a.a);
            f.nee_b[p.local] = make_float4(lds_column.defer_b.r, lds_column.defer_b.g, lds_column.defer_b.b, lds_column.defer_b.a);
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
    if constexpr (SHADOW == RZ_SHADOW_DEFER) {
        // the shadow rays of this pixel start at the hit point and point at the light the (last) sample chose: rays from one cell to
        // one light walk the same instances.  Pixels without a sample have nothing to walk and sort to the end.
        if (f.shadow_key)
            f.shadow_key[p.local] = lds_column.defer_mask ? ray_sort_key(s, V3(lds_column.key_o[0], lds_column.key_o[1], lds_column.key_o[2]),
                                                                         V3(lds_column.key_dir[0], lds_column.key_dir[1], lds_column.key_dir[2]), s.shadow_variant)
                                                          : 0x00FFFFFEu;
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
    return reinterpret_cast<uint32_t*>(MODE == 2 ? workspace + BinnedLds::kFixedBytes : workspace) + threadIdx.x;
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
    if constexpr (MODE == 2) {  // what the walk does not read is parked in LDS meanwhile
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
    DScene s = scene_in;
    unsigned char* workspace = rz_lds + stage_scene<LDS_SCENE>(s, rz_lds);
    uint32_t* lds_column = stack_column<MODE>(workspace);
    const PixelId p = pixel_of_thread(f, cam, blockIdx.x, threadIdx.x);
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
        if constexpr (MODE == 2) {
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
    // sorted order: thread i walks the ray of pixel perm[i] (the hit record still goes to that pixel's slot)
    // sorted order: thread i walks the ray of pixel perm[i] (the hit record still goes to that pixel's slot)
    if (f.wg_times && threadIdx.x == 0u) f.wg_times[2u * blockIdx.x] = wall_clock64();
    const uint32_t sorted_slot = blockIdx.x * 256u + threadIdx.x;
    const PixelId p = (!FIRST && f.perm) ? pixel_of_local(f, cam, f.perm[sorted_slot]) : pixel_of_thread(f, cam, blockIdx.x, threadIdx.x);
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
    if (f.wg_times) {
        __syncthreads();
        if (threadIdx.x == 0u) f.wg_times[2u * blockIdx.x + 1u] = wall_clock64();
    }
}

// MODE 3 trace kernel, one wave per workgroup.  A workgroup's registers and LDS stay allocated until its LAST wave ends and
// a wave lasts as long as its slowest ray, so with heavy-tailed ray costs single-wave workgroups give their slots back sooner
// (config D 3 378 -> 3 093 us, C 974 -> 910 us against 256 threads); the price is a smaller share of LDS for the tree-top cache
// (top_n nodes per workgroup).  MINW = waves per SIMD the register budget is cut for: big trees are bound by the latency of
// their node fetches and want occupancy (D: 6 waves 2 959 us, 4 waves 3 370 us), trees that live in L2 / LDS want registers
// (C: 4 waves 879 us, 6 waves 984 us).
#ifndef RZ_TRACE_PARK
#define RZ_TRACE_PARK 1
#endif
template <bool FIRST, bool COUNT, int MINW, bool ORDERED>
__global__ void __launch_bounds__(64, MINW) rz_trace_skip_kernel(const DScene s, const DCamera cam, const DFrame f, uint32_t top_n) {
    constexpr int WG = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char rz_lds[];
    float4* ln = reinterpret_cast<float4*>(rz_lds);
    uint32_t* ls = reinterpret_cast<uint32_t*>(rz_lds + top_n * 32u);
    if constexpr (!ORDERED) {  // the front-to-back walk reads its 64-B records straight from L1 / L2 (top_n = 0): LDS only parks state
        for (uint32_t i = threadIdx.x; i < 2u * top_n; i += uint32_t(WG)) ln[i] = s.nodes[i];
        for (uint32_t i = threadIdx.x; i < top_n; i += uint32_t(WG)) ls[i] = s.node_skip[i];
    }
    if constexpr (WG > 64) __syncthreads();
    const uint32_t slot = blockIdx.x * uint32_t(WG) + threadIdx.x;
    const PixelId p = pixel_of_local(f, cam, (!FIRST && f.perm) ? f.perm[slot] : slot);
    Counters cnt;
    Ray ray;
    {
        PathState ps;
        load_path<FIRST>(f, cam, p, ps);
        ray = ps.ray;
    }
    Hit hit;
    hit.instance = -1, hit.triangle = 0u, hit.bx = hit.by = 0.0f, hit.external = true;
    int found = 0;
    if (p.active && s.n_instances != 0u) {
        // MINW 6 (80 VGPRs): the world-space ray waits in LDS while a mesh is walked (closest_hit_skip<.., PARK>)
        constexpr bool PARK = ORDERED && MINW >= 6 && RZ_TRACE_PARK != 0;
        const TopCache top{ln, ls, top_n, reinterpret_cast<float*>(rz_lds)};
        found = closest_hit_skip<COUNT, RZ_TRACE_SHARED_RCP != 0, ORDERED, PARK>(s, top, ray, hit, cnt);
    }
    if (p.active) {
        f.hit0[p.local] = make_float4(ray.far_, hit.bx, hit.by, __uint_as_float(hit.triangle));
        f.hit1[p.local] = (uint32_t(hit.instance) & 0x1FFFFFFFu) | (uint32_t(found) << 29) | (hit.external ? 0x80000000u : 0u);
    }
    flush_counters<COUNT>(f, 0u, cnt);
}

// The front-to-back walk with the cooperative triangle phase (hiprz_device.hpp: closest_hit_coop): one wave per workgroup, all
// 64 lanes go through the walk together (a lane without a ray only helps with other lanes' triangles).
template <bool FIRST, bool COUNT, int MINW>
__global__ void __launch_bounds__(64, MINW) rz_trace_coop_kernel(const DScene s, const DCamera cam, const DFrame f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rz_lds[];
    const uint32_t slot = blockIdx.x * 64u + threadIdx.x;
    const PixelId p = pixel_of_local(f, cam, (!FIRST && f.perm) ? f.perm[slot] : slot);
    Counters cnt;
    Ray ray;
    {
        PathState ps;
        load_path<FIRST>(f, cam, p, ps);
        ray = ps.ray;
    }
    Hit hit;
    hit.instance = -1, hit.triangle = 0u, hit.bx = hit.by = 0.0f, hit.external = true;
    int found = 0;
    if (s.n_instances != 0u) found = closest_hit_coop<COUNT, RZ_TRACE_SHARED_RCP != 0>(s, CoopLds(rz_lds), p.active, ray, hit, cnt);
    if (p.active) {
        f.hit0[p.local] = make_float4(ray.far_, hit.bx, hit.by, __uint_as_float(hit.triangle));
        f.hit1[p.local] = (uint32_t(hit.instance) & 0x1FFFFFFFu) | (uint32_t(found) << 29) | (hit.external ? 0x80000000u : 0u);
    }
    flush_counters<COUNT>(f, 0u, cnt);
}

// MODE 6 trace kernel: "wave pool".  A wave lasts as long as its slowest ray, and with heavy-tailed ray costs (config D:
// the slowest of 64 rays costs ~12x the mean) the MODE 3 walk leaves 87 % of the lanes idle.  Here a 64-lane workgroup is
// persistent: it draws rays from a global counter and alternates two phases over in-register per-lane state —
//   A. lanes that hold a ray but are not inside a mesh advance through the world tree / instance boxes to their next
//      mesh (or finish the ray, write its hit record and free the lane);
//   B. lanes inside a mesh walk it ("while-while": node steps until a leaf is held, then the triangles), until fewer
//      than `threshold` lanes are left in meshes and somebody could join them — then idle lanes are refilled, phase A
//      brings the others to their next mesh, and phase B resumes with a dense wave while the stragglers simply
//      kept their state in their lanes.
// Per ray the boxes and triangles are tested in the reference's order; only the interleaving across lanes differs.
#ifndef RZ_POOL_MIN_WAVES
#define RZ_POOL_MIN_WAVES 5
#endif
template <bool FIRST, bool COUNT>
__global__ void __launch_bounds__(64, RZ_POOL_MIN_WAVES) rz_trace_pool_kernel(const DScene s, const DCamera cam, const DFrame f, uint32_t top_n,
                                                                              uint32_t* fresh_next, uint32_t threshold) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rz_lds[];
    constexpr bool RCP = RZ_TRACE_SHARED_RCP != 0;
    float4* ln = reinterpret_cast<float4*>(rz_lds);
    uint32_t* ls = reinterpret_cast<uint32_t*>(rz_lds + top_n * 32u);
    for (uint32_t k = threadIdx.x; k < 2u * top_n; k += 64u) ln[k] = s.nodes[k];
    for (uint32_t k = threadIdx.x; k < top_n; k += 64u) ls[k] = s.node_skip[k];
    const TopCache top{ln, ls, top_n};
    const uint32_t n_slots = f.n_local_tiles * 256u, lane = threadIdx.x;
    const bool scene_fast = s.fast_div != 0u;

    bool has_ray = false, in_mesh = false, found = false, root_missed = false;
    bool fresh_left = true;  // wave-uniform
    uint32_t pixel = 0u, n = RZ_END, i = 0u, end = 0u, inst = 0u, m = RZ_END;
    uint32_t tj = 0u, tj_end = 0u;  // the held leaf's remaining triangles
    const uint32_t kmax = s.walk_k ? s.walk_k : 0xFFFFFFFFu, lmax = s.walk_l ? s.walk_l : 0xFFFFFFFFu;
    float len = 1.0f;
    WalkRay g, lr;
    g.o = g.d = g.y = V3(0.0f, 0.0f, 1.0f), g.near_ = g.far_ = 0.0f, g.fast = false;
    lr = g;
    Hit hit;
    hit.instance = -1, hit.triangle = 0u, hit.bx = hit.by = 0.0f, hit.external = true;
    Counters cnt;
    uint32_t guard = 0u;

    while (true) {
        RZ_GUARD(guard);
        // ---- refill the idle lanes from the global ray counter (one atomic per wave) ----
        if (fresh_left) {
            const unsigned long long idle = __ballot(!has_ray);
            if (idle) {
                const int leader = __ffsll((long long)idle) - 1;
                const uint32_t want = uint32_t(__popcll(idle));
                uint32_t base = 0u;
                if (int(lane) == leader) base = atomicAdd(fresh_next, want);
                base = __shfl(base, leader);
                if (base + want >= n_slots) fresh_left = false;
                if (!has_ray) {
                    const uint32_t slot = base + uint32_t(__popcll(idle & ((1ull << lane) - 1ull)));
                    if (slot < n_slots) {
                        const PixelId p = pixel_of_local(f, cam, (!FIRST && f.perm) ? f.perm[slot] : slot);
                        if (p.active) {
                            PathState ps;
                            load_path<FIRST>(f, cam, p, ps);
                            pixel = p.local;
                            g.o = ps.ray.o, g.d = ps.ray.d, g.near_ = ps.ray.near_, g.far_ = ps.ray.far_;
                            prepare<RCP>(g, scene_fast);
                            hit.instance = -1, hit.triangle = 0u, hit.bx = hit.by = 0.0f, hit.external = true;
                            in_mesh = false, found = false, i = end = 0u;
                            root_missed = s.n_instances == 0u;  // no instances: the reference returns at once (:282)
                            n = s.n_instances ? s.tlas_root : RZ_END;
                            has_ray = true;
                        }
                    }
                }
            }
        }
        if (!__any(has_ray)) {
            if (!fresh_left) break;
            continue;  // the slots drawn were all outside the frame: draw again (the counter only grows)
        }

        // ---- phase A: to the next mesh, or to the end of the ray ----
        while (has_ray && !in_mesh) {
            RZ_GUARD(guard);
            if (i < end) {  // instances of the current world leaf (cpu_engine_kernel.cpp:268-275, 299-306)
                inst = s.tlas_order[i];
                i += 1u;
                float4 ib0, ib1;
                load_instance_box(s, inst, ib0, ib1);
                RZ_COUNT(box_tests);
                if (box_hit<RCP>(ib0, ib1, g)) {
                    const InstanceXform x = load_instance_xform(s, inst);
                    len = to_local<RCP>(x, g, lr, scene_fast);
                    in_mesh = true, found = false;
                    m = x.blas_root;
                    tj = tj_end = 0u;
                }
                continue;
            }
            if (n == RZ_END) {  // the ray is complete: publish its hit record (rz_trace_kernel's layout)
                const int code = root_missed ? 0 : (hit.instance >= 0 ? 2 : 1);
                f.hit0[pixel] = make_float4(g.far_, hit.bx, hit.by, __uint_as_float(hit.triangle));
                f.hit1[pixel] = (uint32_t(hit.instance) & 0x1FFFFFFFu) | (uint32_t(code) << 29) | (hit.external ? 0x80000000u : 0u);
                has_ray = false;
                break;
            }
            float4 n0, n1;
            uint32_t link;
            fetch_node(s, top, n, n0, n1, link);
            RZ_COUNT(box_tests);
            if (box_hit<RCP>(n0, n1, g)) {
                const uint32_t begin = __float_as_uint(n1.z), meta = __float_as_uint(n1.w);
                if (!(meta & HIPRZ_NODE_LEAF)) {
                    n = begin;
                } else {
                    i = begin, end = begin + (meta & HIPRZ_NODE_COUNT_MASK);
                    n = link;
                }
            } else {
                if (n == s.tlas_root) root_missed = true, link = RZ_END;  // root box missed (:283)
                n = link;
            }
        }

        // ---- phase B: the mesh walks (cpu_engine_kernel.cpp:331-352) ----
        while (true) {
            RZ_GUARD(guard);
            const uint32_t walking = uint32_t(__popcll(__ballot(in_mesh)));
            if (walking == 0u) break;
            if (walking < threshold && (__any(has_ray && !in_mesh) || (fresh_left && __any(!has_ray)))) break;
            if (in_mesh) {
                // one bounded round: up to walk_k node steps for a lane that holds no leaf, then up to walk_l triangles of the held leaf
                uint32_t k = 0u;
                while (tj == tj_end && m != RZ_END && k < kmax) {
                    RZ_GUARD(guard);
                    k += 1u;
                    float4 m0, m1;
                    uint32_t mlink;
                    fetch_node(s, top, m, m0, m1, mlink);
                    RZ_COUNT(box_tests);
                    if (box_hit<RCP>(m0, m1, lr)) {
                        const uint32_t mbegin = __float_as_uint(m1.z), mmeta = __float_as_uint(m1.w);
                        if (!(mmeta & HIPRZ_NODE_LEAF)) {
                            m = mbegin;
                            continue;
                        }
                        tj = mbegin, tj_end = mbegin + (mmeta & HIPRZ_NODE_COUNT_MASK);
                    }
                    m = mlink;
                }
                if (tj == tj_end && m == RZ_END) {  // the mesh is done (:320-329)
                    if (found) {
                        hit.instance = int32_t(inst);
                        g.near_ = lr.near_ / len;
                        g.far_ = lr.far_ / len;
                    }
                    in_mesh = false;
                } else {
                    uint32_t l = 0u;
                    for (; tj < tj_end && l < lmax; ++tj, ++l) {
                        const float4 a = s.tris[3 * tj], b = s.tris[3 * tj + 1], c = s.tris[3 * tj + 2];
                        float t, b1, b2, det;
                        RZ_COUNT(tri_tests);
                        if (tri_hit(xyz(a), xyz(b), xyz(c), lr, t, b1, b2, det)) {
                            lr.far_ = t;
                            hit.triangle = tj;
                            hit.external = det > 0.0f;
                            hit.bx = b1, hit.by = b2;
                            found = true;
                        }
                    }
                }
            }
        }
    }
    flush_counters<COUNT>(f, 0u, cnt);
}

// MODE 5 trace kernels: round 0 walks every owned pixel's ray, round r > 0 the rays round r-1 left unfinished.
// Queue record r of a round (48 B, SoA): q0 = (pixel slot, world leaf, tlas_order slot, mesh node), q1 = (near, far,
// mesh-space far, bits(found | external << 1)), q2 = (b1, b2, bits(triangle), bits(instance)).
struct DRequeue {
    const uint4* in0;
    const float4* in1;
    const float4* in2;
    uint4* out0;
    float4* out1;
    float4* out2;
    uint32_t* counts;    // counts[r] = rays queued FOR round r (counts[0] unused)
    uint32_t round;
    uint32_t threshold;  // lanes that must remain in a mesh walk for it to go on (this round)
};
template <bool FIRST, bool COUNT, bool ROUND0, bool CAN_BAIL>
__global__ void __launch_bounds__(256, RZ_TRACE_MIN_WAVES) rz_trace_requeue_kernel(const DScene s, const DCamera cam, const DFrame f, const DRequeue q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rz_lds[];
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    uint32_t n_in = 0u;
    if constexpr (!ROUND0) {
        n_in = q.counts[q.round];
        if (blockIdx.x * 256u >= n_in) return;  // whole workgroup: nothing queued for it
    }
    float4* ln = reinterpret_cast<float4*>(rz_lds);
    uint32_t* ls = reinterpret_cast<uint32_t*>(rz_lds + s.top_count * 32u);
    for (uint32_t i = threadIdx.x; i < 2u * s.top_count; i += 256u) ln[i] = s.nodes[i];
    for (uint32_t i = threadIdx.x; i < s.top_count; i += 256u) ls[i] = s.node_skip[i];
    __syncthreads();
    const TopCache top{ln, ls, s.top_count};

    Counters cnt;
    PixelId p;
    Ray ray;
    Hit hit;
    hit.instance = -1, hit.triangle = 0u, hit.bx = hit.by = 0.0f, hit.external = true;
    WalkResume rs;
    rs.n = s.tlas_root, rs.i = 0u, rs.m = RZ_END, rs.in_mesh = false, rs.found = false, rs.lr_far = 0.0f;
    if constexpr (ROUND0) {
        p = (!FIRST && f.perm) ? pixel_of_local(f, cam, f.perm[slot]) : pixel_of_thread(f, cam, blockIdx.x, threadIdx.x);
        PathState ps;
        load_path<FIRST>(f, cam, p, ps);
        ray = ps.ray;
    } else {
        const bool queued = slot < n_in;
        const uint4 r0 = queued ? q.in0[slot] : make_uint4(0u, 0u, 0u, 0u);
        p = pixel_of_local(f, cam, r0.x);
        if (!queued) p.active = false;
        PathState ps;
        load_path<FIRST>(f, cam, p, ps);
        ray = ps.ray;
        if (queued) {
            const float4 r1 = q.in1[slot], r2 = q.in2[slot];
            const uint32_t bits = __float_as_uint(r1.w);
            ray.near_ = r1.x, ray.far_ = r1.y;
            rs.n = r0.y, rs.i = r0.z, rs.m = r0.w, rs.in_mesh = true, rs.found = (bits & 1u) != 0u, rs.lr_far = r1.z;
            hit.bx = r2.x, hit.by = r2.y, hit.triangle = __float_as_uint(r2.z), hit.instance = int32_t(__float_as_uint(r2.w));
            hit.external = (bits & 2u) != 0u;
        }
    }
    int found = 0;
    if (p.active && s.n_instances != 0u) found = closest_hit_requeue<COUNT, RZ_TRACE_SHARED_RCP != 0, CAN_BAIL>(s, top, ray, hit, rs, q.threshold, cnt);
    if constexpr (CAN_BAIL) {
        // unfinished rays -> next round's queue, one atomic per wave
        const bool bailed = found == 3;
        const unsigned long long mask = __ballot(bailed);
        if (mask) {
            const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            uint32_t base = 0u;
            if (lane == uint32_t(__ffsll((long long)mask) - 1)) base = atomicAdd(&q.counts[q.round + 1u], uint32_t(__popcll(mask)));
            base = __shfl(base, __ffsll((long long)mask) - 1);
            if (bailed) {
                const uint32_t o = base + uint32_t(__popcll(mask & ((1ull << lane) - 1ull)));
                q.out0[o] = make_uint4(p.local, rs.n, rs.i, rs.m);
                q.out1[o] = make_float4(ray.near_, ray.far_, rs.lr_far, __uint_as_float((rs.found ? 1u : 0u) | (hit.external ? 2u : 0u)));
                q.out2[o] = make_float4(hit.bx, hit.by, __uint_as_float(hit.triangle), __uint_as_float(uint32_t(hit.instance)));
            }
        }
    }
    if (p.active && found != 3) {
        f.hit0[p.local] = make_float4(ray.far_, hit.bx, hit.by, __uint_as_float(hit.triangle));
        f.hit1[p.local] = (uint32_t(hit.instance) & 0x1FFFFFFFu) | (uint32_t(found) << 29) | (hit.external ? 0x80000000u : 0u);
    }
    flush_counters<COUNT>(f, 0u, cnt);
}

// MODE 4 trace kernel: persistent lanes over a per-workgroup pool of RZ_POOL_FACTOR * 256 rays
template <bool FIRST, bool COUNT>
__global__ void __launch_bounds__(256, RZ_TRACE_MIN_WAVES) rz_trace_persistent_kernel(const DScene s, const DCamera cam, const DFrame f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rz_lds[];
    float4* ln = reinterpret_cast<float4*>(rz_lds);
    uint32_t* ls = reinterpret_cast<uint32_t*>(rz_lds + s.wtop_count * 32u);
    uint32_t* pool_next = ls + s.wtop_count;
    for (uint32_t i = threadIdx.x; i < 2u * s.wtop_count; i += 256u) ln[i] = s.wnodes[i];
    for (uint32_t i = threadIdx.x; i < s.wtop_count; i += 256u) ls[i] = s.wskip[i];
    if (threadIdx.x == 0u) *pool_next = 0u;
    __syncthreads();
    const uint32_t pool_rays = 256u * RZ_POOL_FACTOR, n_slots = f.n_local_tiles * 256u;
    const uint32_t pool_begin = blockIdx.x * pool_rays, pool_end = pool_begin + pool_rays < n_slots ? pool_begin + pool_rays : n_slots;
    Counters cnt;
    const WalkTop top{ln, ls, s.wtop_count};
    trace_persistent<FIRST, COUNT>(s, top, cam, f, pool_next, pool_begin, pool_end, cnt);
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
    if constexpr (SHADOW == RZ_SHADOW_DEFER) {
        shadow.lds_column = nullptr;
        shadow.nee_point = f.nee_point, shadow.nee_dir = f.nee_dir, shadow.nee_term = f.nee_term;
        shadow.nee_stride = f.n_local_tiles * 256u;
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
        const int found = int((h1 >> 29) & 3u);
        ps.ray.far_ = h0.x;
        hit.bx = h0.y, hit.by = h0.z, hit.triangle = __float_as_uint(h0.w);
        hit.instance = found == 2 ? int32_t(h1 & 0x1FFFFFFFu) : -1;
        hit.external = (h1 & 0x80000000u) != 0u;
        shadow.pixel = p.local;
        shade_and_store<FIRST, COUNT, SHADOW>(s, cam, cfg, f, p, ps, found, hit, shadow, cnt);
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
template <bool FIRST, bool COUNT, int MINW, bool ORDERED>
__global__ void __launch_bounds__(64, MINW) rz_shadow_kernel(const DScene s, const DCamera cam, const DConfig cfg, const DFrame f, uint32_t top_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rz_lds[];
    float4* ln = reinterpret_cast<float4*>(rz_lds);
    uint32_t* ls = reinterpret_cast<uint32_t*>(rz_lds + top_n * 32u);
    if constexpr (!ORDERED)
        for (uint32_t i = threadIdx.x; i < 2u * top_n; i += 64u) ln[i] = s.nodes[i];
    if constexpr (!ORDERED)
        for (uint32_t i = threadIdx.x; i < top_n; i += 64u) ls[i] = s.node_skip[i];
    const uint32_t slot = blockIdx.x * 64u + threadIdx.x;
    const uint32_t* order = f.shadow_perm ? f.shadow_perm : f.perm;
    const PixelId p = pixel_of_local(f, cam, order ? order[slot] : slot);
    Counters cnt;
    if (p.active) {
        const ShadowCtx sc{nullptr, TopCache{ln, ls, top_n}};
        const float4 base = f.nee_base[p.local];
        const uint32_t bits = __float_as_uint(base.w);
        const bool path_continues = (bits & 1u) != 0u;
        col4 final_color{base.x, base.y, base.z, 0.0f};
        if (bits & 2u) {
            const uint32_t mask = bits >> 2, stride = f.n_local_tiles * 256u;
            const float4 o = f.nee_point[p.local];
            auto shadowed_sum = [&](uint32_t first, uint32_t count) {
                col4 total = splat(0.0f);
                for (uint32_t k = first; k < first + count; ++k) {
                    if (!(mask & (1u << k))) continue;
                    const float4 d = f.nee_dir[size_t(k) * stride + p.local], t = f.nee_term[size_t(k) * stride + p.local];
                    Ray sr;
                    sr.o = V3(o.x, o.y, o.z), sr.d = V3(d.x, d.y, d.z), sr.near_ = 0.0f, sr.far_ = d.w;
                    const col4 V_PL = splat(any_hit<ORDERED ? 7 : 3, COUNT>(s, sc, sr, cnt));
                    total = total + (col4{t.x, t.y, t.z, t.w} * V_PL) * V_PL.a;
                }
                return total;
            };
            col4 direct_total = splat(0.0f), spot_total = splat(0.0f);
            if (s.n_direct_lights != 0u) direct_total = div_scalar(shadowed_sum(0u, cfg.direct_samples), float(cfg.direct_samples) / float(s.n_direct_lights));
            if (s.n_spot_lights != 0u) spot_total = div_scalar(shadowed_sum(cfg.direct_samples, cfg.spot_samples), float(cfg.spot_samples) / float(s.n_spot_lights));
            const col4 direct = direct_total + spot_total;
            const float4 a = f.nee_a[p.local], b = f.nee_b[p.local];
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
template <bool FIRST, bool COUNT, int MINW>
__global__ void __launch_bounds__(64, MINW) rz_shadow_coop_kernel(const DScene s, const DCamera cam, const DConfig cfg, const DFrame f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rz_lds[];
    const CoopLds lds(rz_lds);
    const uint32_t slot = blockIdx.x * 64u + threadIdx.x;
    const uint32_t* order = f.shadow_perm ? f.shadow_perm : f.perm;
    const PixelId p = pixel_of_local(f, cam, order ? order[slot] : slot);
    Counters cnt;
    float4 base = make_float4(0.0f, 0.0f, 0.0f, 0.0f), o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t bits = 0u;
    if (p.active) {
        base = f.nee_base[p.local];
        bits = __float_as_uint(base.w);
        if (bits & 2u) o = f.nee_point[p.local];
    }
    const uint32_t mask = (bits & 2u) ? bits >> 2 : 0u, stride = f.n_local_tiles * 256u;
    col4 direct_total = splat(0.0f), spot_total = splat(0.0f);
    const uint32_t n_samples = cfg.direct_samples + cfg.spot_samples;
    for (uint32_t k = 0u; k < n_samples; ++k) {  // wave-uniform
        const bool has = (mask & (1u << k)) != 0u;
        if (!__any(has)) continue;
        float4 d = make_float4(0.0f, 0.0f, 1.0f, 0.0f), t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (has) d = f.nee_dir[size_t(k) * stride + p.local], t = f.nee_term[size_t(k) * stride + p.local];
        Ray sr;
        sr.o = V3(o.x, o.y, o.z), sr.d = V3(d.x, d.y, d.z), sr.near_ = 0.0f, sr.far_ = d.w;
        if (has) { RZ_COUNT(shadow_rays); }
        float v = 0.0f;
        if (s.n_instances != 0u) v = any_hit_coop<COUNT, RZ_SHADE_SHARED_RCP != 0>(s, lds, has, sr, cnt);
        if (has) {
            const col4 V_PL = splat(v);
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
            const float4 a = f.nee_a[p.local], b = f.nee_b[p.local];
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

// passUpdate / segmentUpdate (cuda_postprocess_kernel.cu:95-104, cuda_render_kernel.cu:122-129):
// the pass index lives on the device so a captured graph replays without new arguments.
__global__ void rz_pass_update_kernel(uint32_t* pass) { *pass += 1u; }
__global__ void rz_pass_reset_kernel(uint32_t* pass) { *pass = 0u; }
__global__ void rz_pass_add_kernel(uint32_t* pass, uint32_t n) { *pass += n; }

// toneMap (cuda_postprocess_kernel.cu:38-93; CPU: cpu_engine_renderer.cpp:224-235)
__global__ void __launch_bounds__(256) rz_tonemap_tiles_kernel(const float4* accum, uint32_t* rgba8, uint32_t n, float aperture,
                                                               float exposure_time) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 a = accum[i];
    rgba8[i] = tonemap(col4{a.x, a.y, a.z, a.w}, aperture, exposure_time);
}
__global__ void __launch_bounds__(256) rz_tonemap_image_kernel(const float4* image, uint32_t* rgba8, uint32_t n, float aperture,
                                                               float exposure_time) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 a = image[i];
    rgba8[i] = tonemap(col4{a.x, a.y, a.z, a.w}, aperture, exposure_time);
}

// tile-major (owned tiles of shard rank/world) -> row-major full frame
template <typename T>
__global__ void __launch_bounds__(256) rz_untile_kernel(const T* tiles, T* image, uint32_t width, uint32_t height,
                                                        uint32_t tiles_x, uint32_t rank, uint32_t world) {
    const uint32_t tile = blockIdx.x * world + rank;
    const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t x = tx * 32u + wave * 8u + (lane & 7u), y = ty * 8u + (lane >> 3);
    if (x < width && y < height) image[size_t(y) * width + x] = tiles[size_t(blockIdx.x) * 256u + threadIdx.x];
}
// the gathered tiles of ALL shards (shard r at tiles + r * part_stride elements) -> row-major full frame, one launch
template <typename T>
__global__ void __launch_bounds__(256) rz_untile_gathered_kernel(const T* tiles, size_t part_stride, T* image, uint32_t width, uint32_t height,
                                                                 uint32_t tiles_x, uint32_t n_tiles, uint32_t world) {
    const uint32_t rank = blockIdx.y, tile = blockIdx.x * world + rank;
    if (tile >= n_tiles) return;
    const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t x = tx * 32u + wave * 8u + (lane & 7u), y = ty * 8u + (lane >> 3);
    if (x < width && y < height) image[size_t(y) * width + x] = tiles[rank * part_stride + size_t(blockIdx.x) * 256u + threadIdx.x];
}
__global__ void __launch_bounds__(256) rz_untile_state_kernel(const float4* st0, const float4* st1, const float2* st2, float* ray9,
                                                              uint32_t* md2, uint32_t width, uint32_t height, uint32_t tiles_x,
                                                              uint32_t rank, uint32_t world) {
    const uint32_t tile = blockIdx.x * world + rank;
    const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t x = tx * 32u + wave * 8u + (lane & 7u), y = ty * 8u + (lane >> 3);
    if (x >= width || y >= height) return;
    const size_t i = size_t(blockIdx.x) * 256u + threadIdx.x, o = size_t(y) * width + x;
    const float4 a = st0[i], b = st1[i];
    const float2 c = st2[i];
    float* r = ray9 + 9 * o;
    r[0] = a.x, r[1] = a.y, r[2] = a.z, r[3] = a.w, r[4] = b.x, r[5] = b.y, r[6] = b.z, r[7] = b.w, r[8] = c.x;
    const uint32_t bits = __float_as_uint(c.y);
    md2[2 * o] = bits & 0xFFFFu;
    md2[2 * o + 1] = (bits >> 16) & 0xFFu;
}

// Kernel::rayCast (cpu_engine_kernel.cpp:102-111, 483-501): one thread.
__global__ void rz_pick_kernel(const DScene s, const DCamera cam, uint32_t x, uint32_t y, float depth, int32_t* out2) {
    uint32_t* rz_lds = nullptr;
    Ray ray;
    generate_simple_ray(cam, ray, x, y);
    ray.near_ = depth * 0.99f;
    ray.far_ = depth * 1.01f;
    Hit hit;
    Counters cnt;
    out2[0] = out2[1] = -1;
    if (closest_hit<0, false, false>(s, rz_lds, ray, hit, cnt) == 2) {
        const uint32_t inst = uint32_t(hit.instance);
        const uint32_t material_base = __float_as_uint(s.instances[7 * inst + 1].w);
        const uint32_t material_count = __float_as_uint(s.instances[7 * inst + 2].w);
        uint32_t slot = __float_as_uint(s.tris[3 * hit.triangle].w) & HIPRZ_TRI_MATERIAL_MASK;
        if (slot > 63u) slot = 63u;
        out2[0] = hit.instance;
        out2[1] = slot < material_count ? s.inst_materials[material_base + slot] : -1;
    }
}

// Device self-test: div_shared() must equal the correctly rounded `/` bit for bit over its
// whole stated operand range.  out[0] = mismatches, out[1] = cases tested.
__global__ void __launch_bounds__(256) rz_selftest_div_kernel(uint32_t n_per_thread, uint32_t seed, unsigned long long* out) {
    uint32_t h = mix32(seed ^ (blockIdx.x * 256u + threadIdx.x) * 0x9E3779B9u);
    uint32_t bad = 0, tested = 0;
    for (uint32_t i = 0; i < n_per_thread; ++i) {
        h = mix32(h + i);
        // d: random sign/mantissa, exponent in [-40, 2); n: zero or exponent in [-84, 41)
        const uint32_t de = 127u - 40u + (h >> 8) % 42u;
        const float d = __uint_as_float((h & 0x80000000u) | (de << 23) | (mix32(h) & 0x7FFFFFu));
        const uint32_t g = mix32(h ^ 0xA5A5A5A5u);
        const uint32_t ne = 127u - 84u + (g >> 8) % 125u;
        float n = __uint_as_float((g & 0x80000000u) | (ne << 23) | (mix32(g) & 0x7FFFFFu));
        if ((g & 0xFFu) == 0u) n = 0.0f;
        const float y = refined_rcp(d);
        const float n2 = n == 0.0f ? 0.0f : __uint_as_float(__float_as_uint(n) ^ (mix32(g + 7u) & 0x007FFFFFu));  // second numerator, same exponent
        const f2 fast = div_shared2(f2{n, n2}, d, y);
        const float exact = n / d, exact2 = n2 / d;
        tested += 2;
        if (__float_as_uint(fast.x) != __float_as_uint(exact) && !(fast.x == 0.0f && exact == 0.0f)) bad += 1;
        if (__float_as_uint(fast.y) != __float_as_uint(exact2) && !(fast.y == 0.0f && exact2 == 0.0f)) bad += 1;
        if (__float_as_uint(div_shared(n, d, y)) != __float_as_uint(fast.x)) bad += 1;
        // sincosf must return what sinf and cosf return: the samplers' angles are in [0, 2*pi], test a wider range
        const float angle = __uint_as_float((g & 0x80000000u) | ((118u + (h >> 11) % 16u) << 23) | (mix32(h + 3u) & 0x7FFFFFu));
        float sn, cs;
        sincosf(angle, &sn, &cs);
        if (__float_as_uint(sn) != __float_as_uint(sinf(angle)) || __float_as_uint(cs) != __float_as_uint(cosf(angle))) bad += 1;
    }
    atomicAdd(&out[0], (unsigned long long)bad);
    atomicAdd(&out[1], (unsigned long long)tested);
}

// =======================================================================================
// Host side of the context
// =======================================================================================
namespace {

thread_local std::string g_create_error;

// Timer/TimeTable of the reference (engine_parts.hpp:34-74): last + EMA(0.05) per stage.
struct TimeTable {
    struct Entry {
        std::string name;
        double last_ms = 0, avg_ms = 0;
        bool seen = false;
    };
    std::vector<Entry> entries;
    void set(const char* name, double ms) {
        for (auto& e : entries)
            if (e.name == name) {
                e.last_ms = ms;
                e.avg_ms = e.seen ? e.avg_ms + (ms - e.avg_ms) * 0.05 : ms;
                e.seen = true;
                return;
            }
        entries.push_back({name, ms, ms, true});
    }
    std::string str() const {
        std::string out;
        char line[160];
        for (const auto& e : entries) {
            std::snprintf(line, sizeof line, "%-22s %9.3fms (avg %9.3fms)\n", e.name.c_str(), e.last_ms, e.avg_ms);
            out += line;
        }
        return out;
    }
};
struct StageTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

template <typename T>
struct DeviceArray {
    T* ptr = nullptr;
    size_t count = 0;
    hipError_t assign(const T* src, size_t n, hipStream_t stream) {
        if (n > count || !ptr) {
            if (ptr) (void)hipFree(ptr);
            ptr = nullptr;
            count = 0;
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&ptr), sizeof(T) * (n ? n : 1));
            if (e != hipSuccess) return e;
            count = n ? n : 1;
        }
        if (n) return hipMemcpyAsync(ptr, src, sizeof(T) * n, hipMemcpyHostToDevice, stream);
        return hipSuccess;
    }
    hipError_t resize(size_t n) {
        if (n <= count && ptr) return hipSuccess;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&ptr), sizeof(T) * (n ? n : 1));
        if (e == hipSuccess) count = n ? n : 1;
        return e;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
    }
};

}  // namespace

struct hiprz_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;
    TimeTable timings;

    // scene mirror
    DeviceArray<uint8_t> hot;  // nodes | tlas_order | instances | tris | tri_attrs | materials | inst_materials
    DeviceArray<hiprz_node> wnodes;
    DeviceArray<uint32_t> wskip;
    DeviceArray<uint32_t> node_skip;
    DeviceArray<uint32_t> nodes64;
    int walk_order = 1;  // 0 = meshes in the reference's child order, 1 = front-to-back, 2 = also when counting (hiprz_set_walk_order)
    DeviceArray<hiprz_texture> textures;
    DeviceArray<uint8_t> texels;
    DeviceArray<hiprz_spot_light> spot_lights;
    DeviceArray<hiprz_direct_light> direct_lights;
    DScene dscene{};
    bool have_scene = false;
    uint32_t stack_entries = 2;  // LDS stack entries per lane the trees need (MODE 1)
    bool lds_scene = false;      // hot blob is staged into LDS by every workgroup
    int lds_scene_override = -1; // -1 auto, 0 never, 1 always (if it fits at all)

    // camera + per-pixel state
    hiprz_camera camera{};
    DCamera dcamera{};
    bool have_camera = false;
    uint32_t rank = 0, world = 1;
    uint32_t tiles_x = 0, tiles_y = 0, n_local_tiles = 0;
    uint64_t owned_pixels = 0;
    DeviceArray<float4> st0, st1, accum, hit0;
    DeviceArray<uint32_t> hit1;
    // 0 fused (one kernel per pass), 1 split (trace kernel -> shade kernel per pass), 2 resident (one kernel per batch
    // of passes).  -1: resident when the scene is staged in LDS (config B: as fast as split on a whole frame, 2.26 ms per
    // 8 passes, and 0.34 vs 0.45 ms on an eighth of it — per-pass launch/ramp/tail costs vanish), else split (10-20 % faster
    // than fused on configs C, D; the resident kernel has no LDS room for the tree-top cache).
    int pipeline_setting = -1;
    int pipeline = 1;  // resolved by resolve_pipeline() at upload / set time
    bool rgba8_valid = false;  // the resident kernel tone-maps on its way out: hiprz_tonemap has nothing to do
    DeviceArray<float2> st2;
    DeviceArray<float> depth;
    DeviceArray<uint32_t> rgba8;
    DeviceArray<float4> image_f4;  // row-major staging for readback
    DeviceArray<uint32_t> state_md;
    DeviceArray<float> state_ray;
    DeviceArray<uint32_t> pass_dev;
    DeviceArray<unsigned long long> counters_dev;
    DeviceArray<int32_t> pick_dev;

    hiprz_config config{8u, 8u, 1u, 1u, 20240501u};
    bool reset_pending = true;
    uint32_t passes = 0;
    uint64_t ray_count = 0;
    int traversal_mode = -1;  // -1 = choose per scene (effective_mode)
    // MODE 5: two ping-pong ray queues (48 B per entry, one entry per owned pixel each), per-round counters and the
    // schedule: requeue_thresholds[r] = lanes that must remain in a mesh walk during round r (the final round never bails)
    DeviceArray<uint4> rq0[2];
    DeviceArray<float4> rq1[2], rq2[2];
    DeviceArray<uint32_t> rq_counts;
    std::vector<uint32_t> requeue_thresholds{40u, 40u, 32u, 32u, 24u, 16u};

    // hipGraph of one batch of cumulative passes ([pass kernel, pass update] x n): replayed while nothing that
    // the captured kernel arguments depend on has changed (scene, camera, config, shard, variants)
    hipGraphExec_t graph_exec = nullptr;
    uint32_t graph_passes = 0;
    bool graph_valid = false;
    // ray reordering between passes (split pipeline): keys from the shade kernel -> radix sort -> permutation
    DeviceArray<uint32_t> sort_keys, sort_keys_out, sort_iota, sort_perm;
    DeviceArray<uint32_t> shadow_keys, shadow_perm;  // deferred shadow rays follow their own order (hiprz_device.hpp: DFrame::shadow_key)
    int coop_walk = 1;    // front-to-back walk with the cooperative triangle phase (rz_trace_coop_kernel); HIPRZ_COOP=0: rz_trace_skip_kernel
    int coop_shadow = 1;  // deferred shadow rays in rz_shadow_coop_kernel (HIPRZ_COOP_SHADOW=0: rz_shadow_kernel)
    uint32_t n_textures = 0;  // of the uploaded scene
    int batch_waves = 0;  // HIPRZ_BATCH_WAVES=4: never the 5-wave build of the plain batch kernel
    int nolight_kernels = 1;  // scenes without lights use the instantiations without next-event estimation (HIPRZ_NOLIGHT_KERNELS=0: the general ones)
    int sort_bits = 0;    // most significant key bits the radix sorts look at; 0 = by frame size (HIPRZ_SORT_BITS)
    int shadow_sort = 1;  // HIPRZ_SHADOW_SORT=0: the shadow kernel follows the next pass's ray order instead
    DeviceArray<uint8_t> sort_temp;
    size_t sort_temp_bytes = 0;
    int sort_rays = -1;  // -1 auto (on for scenes walked with MODE 3), 0 off, 1 on
    DeviceArray<unsigned long long> wg_times;  // diagnostics: start / end clock of every trace-kernel workgroup of the last pass
    bool wg_timing = false;
    uint32_t pool_threshold = 32u;  // MODE 6: lanes that must be inside meshes for the mesh phase to go on while others could join
    bool sorted_this_pass = false;  // the deferred shadow kernel wants the NEXT pass's ray order: the sort then runs before it
    bool defer_shadow_rays = true;  // HIPRZ_DEFER_SHADOWS=0: walk them inside the shade kernel
    DeviceArray<float4> nee_base, nee_a, nee_b, nee_point, nee_dir, nee_term;
    int shade_shadow_walk = 3;  // shade kernel of scenes not staged in LDS: 3 = skip links + staged tree tops, 1 = LDS stack (HIPRZ_SHADOW_WALK)
    int trace_wg = 64;    // MODE 3 trace kernel: 64 = one wave per workgroup (rz_trace_skip_kernel); HIPRZ_TRACE_WG=256: the 256-thread kernel
    int trace_waves = 0;  // 0 = by tree size; HIPRZ_TRACE_WAVES = 4 | 6 forces the register budget
    uint32_t n_nodes = 0;
    bool time_kernels = false;  // record events around the trace and the shade kernel of every pass of a batch
    std::vector<hipEvent_t> kernel_events;
    uint32_t kernel_event_passes = 0;
    bool use_graph = true;
    bool xcd_swizzle = false;  // measured: banding the image per XCD concentrates the expensive region on few XCDs (D: 4.3 -> 5.1 ms)

    // kernel timing (hip events on `stream` around each render batch)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_events;
    std::vector<uint32_t> pending_launches;
    std::vector<hipEvent_t> event_pool;
};

namespace {

int fail(hiprz_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->error = msg;
    else g_create_error = msg;
    return code;
}

#define RZ_HIP(ctx, call)                                                                                       \
    do {                                                                                                        \
        hipError_t rz_e = (call);                                                                               \
        if (rz_e != hipSuccess)                                                                                 \
            return fail(ctx, HIPRZ_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(rz_e));           \
    } while (0)

struct TreeCheck {
    const hiprz_scene* sc;
    std::vector<uint32_t>& skip;
    std::vector<uint8_t> visited;
    uint32_t max_depth = 0;
    std::string error;
    std::vector<uint32_t> world_leaves;

    // Walks one tree from `root`, verifies every index it will make the kernel follow, fills
    // the skip links, returns false on the first violation.  `is_world`: leaves index tlas_order.
    bool walk(uint32_t root, bool is_world) {
        struct Item {
            uint32_t node, skip, depth;
        };
        std::vector<Item> stack{{root, RZ_END, 1u}};
        while (!stack.empty()) {
            const Item it = stack.back();
            stack.pop_back();
            if (it.node >= sc->n_nodes) return err("node index out of range");
            if (visited[it.node]) return err("node reachable twice (trees must be disjoint and acyclic)");
            visited[it.node] = 1;
            skip[it.node] = it.skip;
            if (it.depth > max_depth) max_depth = it.depth;
            if (it.depth > 64u) return err("tree deeper than 64 levels");
            const hiprz_node& n = sc->nodes[it.node];
            if (n.meta & HIPRZ_NODE_LEAF) {
                const uint64_t end = uint64_t(n.begin) + (n.meta & HIPRZ_NODE_COUNT_MASK);
                if (end > (is_world ? sc->n_tlas_order : sc->n_tris)) return err("leaf range out of bounds");
                if (is_world) world_leaves.push_back(it.node);
            } else {
                if (uint64_t(n.begin) + 1 >= sc->n_nodes) return err("child index out of range");
                stack.push_back({n.begin + 1, it.skip, it.depth + 1});
                stack.push_back({n.begin, n.begin + 1, it.depth + 1});
            }
        }
        return true;
    }
    bool err(const char* m) {
        error = m;
        return false;
    }
};

void release_frame(hiprz_ctx* c) {
    c->st0.release(), c->st1.release(), c->st2.release(), c->accum.release(), c->depth.release(), c->rgba8.release();
    c->hit0.release(), c->hit1.release();
    for (int k = 0; k < 2; ++k) c->rq0[k].release(), c->rq1[k].release(), c->rq2[k].release();
    c->nee_base.release(), c->nee_a.release(), c->nee_b.release(), c->nee_point.release(), c->nee_dir.release(), c->nee_term.release();
    c->sort_keys.release(), c->sort_keys_out.release(), c->sort_iota.release(), c->sort_perm.release(), c->sort_temp.release(), c->shadow_keys.release(), c->shadow_perm.release();
    c->image_f4.release(), c->state_md.release(), c->state_ray.release();
}

int allocate_frame(hiprz_ctx* c) {
    const uint32_t W = c->camera.width, H = c->camera.height;
    c->tiles_x = (W + 31u) / 32u;
    c->tiles_y = (H + 7u) / 8u;
    const uint32_t n_tiles = c->tiles_x * c->tiles_y;
    c->n_local_tiles = c->rank < n_tiles ? (n_tiles - c->rank + c->world - 1u) / c->world : 0u;
    // owned active pixels (ray counter of this shard)
    uint64_t owned = 0;
    for (uint32_t lt = 0; lt < c->n_local_tiles; ++lt) {
        const uint32_t t = lt * c->world + c->rank, tx = t % c->tiles_x, ty = t / c->tiles_x;
        const uint32_t w = std::min(32u, W - tx * 32u), h = std::min(8u, H - ty * 8u);
        owned += uint64_t(w) * h;
    }
    c->owned_pixels = owned;
    const size_t n = size_t(c->n_local_tiles) * 256u;
    RZ_HIP(c, c->st0.resize(n));
    RZ_HIP(c, c->st1.resize(n));
    RZ_HIP(c, c->st2.resize(n));
    RZ_HIP(c, c->accum.resize(n));
    RZ_HIP(c, c->hit0.resize(n));
    RZ_HIP(c, c->hit1.resize(n));
    RZ_HIP(c, c->sort_keys.resize(n));
    RZ_HIP(c, c->sort_keys_out.resize(n));
    RZ_HIP(c, c->sort_perm.resize(n));
    RZ_HIP(c, c->sort_iota.resize(n));
    RZ_HIP(c, c->shadow_keys.resize(n));
    RZ_HIP(c, c->shadow_perm.resize(n));
    if (n) {
        std::vector<uint32_t> iota(n);
        for (size_t i = 0; i < n; ++i) iota[i] = uint32_t(i);
        RZ_HIP(c, hipMemcpyAsync(c->sort_iota.ptr, iota.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        RZ_HIP(c, hipMemcpyAsync(c->sort_perm.ptr, iota.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));  // identity until the first sort
        RZ_HIP(c, hipMemcpyAsync(c->shadow_perm.ptr, iota.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        RZ_HIP(c, hipMemsetAsync(c->sort_keys.ptr, 0, n * sizeof(uint32_t), c->stream));
        RZ_HIP(c, hipMemsetAsync(c->shadow_keys.ptr, 0, n * sizeof(uint32_t), c->stream));
        RZ_HIP(c, hipStreamSynchronize(c->stream));
        size_t bytes = 0;
        RZ_HIP(c, hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, c->sort_keys.ptr, c->sort_keys_out.ptr, c->sort_iota.ptr,
                                                     c->sort_perm.ptr, int(n), 0, 24, c->stream));
        RZ_HIP(c, c->sort_temp.resize(bytes));
        c->sort_temp_bytes = bytes;
    }
    RZ_HIP(c, c->depth.resize(n));
    RZ_HIP(c, c->rgba8.resize(n));
    RZ_HIP(c, c->image_f4.resize(size_t(W) * H));
    if (n) {
        RZ_HIP(c, hipMemsetAsync(c->accum.ptr, 0, n * sizeof(float4), c->stream));
        RZ_HIP(c, hipMemsetAsync(c->depth.ptr, 0, n * sizeof(float), c->stream));
        RZ_HIP(c, hipMemsetAsync(c->rgba8.ptr, 0, n * sizeof(uint32_t), c->stream));
        RZ_HIP(c, hipMemsetAsync(c->st0.ptr, 0, n * sizeof(float4), c->stream));
        RZ_HIP(c, hipMemsetAsync(c->st1.ptr, 0, n * sizeof(float4), c->stream));
        RZ_HIP(c, hipMemsetAsync(c->st2.ptr, 0, n * sizeof(float2), c->stream));
    }
    return HIPRZ_OK;
}

constexpr uint32_t kLatencyBoundNodes = 32768u;  // trees beyond ~1 MiB of nodes: fetches come from L2 / HBM, occupancy hides them
int effective_mode(const hiprz_ctx* c);
void resolve_pipeline(hiprz_ctx* c);
bool defer_shadows(const hiprz_ctx* c);
bool use_lds_scene(const hiprz_ctx* c);
bool resident_active(const hiprz_ctx* c) { return c->pipeline == 2; }
// rays are reordered where the walk is bound by scattered fetches: scenes not staged in LDS, split pipeline
bool sort_enabled(const hiprz_ctx* c) {
    if (c->pipeline != 1 || c->sort_rays == 0) return false;
    if (c->sort_rays == 1) return true;
    // measured (1920x1080+, MODE 3; the sort itself costs ~0.12 ms per pass at 1080p): many small instances (config E, 46) 33.9 ->
    // 26.5 ms per pass; one mid-size mesh (config C, 12 k nodes) trace kernel 853 -> 645 us, step 7.70 -> 6.98 ms; one big mesh
    // (config D, 600 k nodes) trace kernel 2 656 -> 2 586 us but step 22.5 -> 23.2 ms.  So: on for many instances, and for trees
    // small enough that a coherent wave finds its nodes in LDS / L2.
    return !use_lds_scene(c) && effective_mode(c) >= 3 && (c->dscene.n_instances >= 16u || c->n_nodes <= kLatencyBoundNodes);
}

DFrame make_frame(hiprz_ctx* c, bool counted) {
    DFrame f{};
    f.st0 = c->st0.ptr, f.st1 = c->st1.ptr, f.st2 = c->st2.ptr;
    f.accum = c->accum.ptr, f.depth = c->depth.ptr, f.rgba8 = c->rgba8.ptr;
    f.hit0 = c->hit0.ptr, f.hit1 = c->hit1.ptr;
    f.pass = c->pass_dev.ptr;
    f.counters = counted ? c->counters_dev.ptr : nullptr;
    f.tiles_x = c->tiles_x, f.rank = c->rank, f.world = c->world, f.n_local_tiles = c->n_local_tiles;
    f.xcd_swizzle = c->xcd_swizzle ? 1u : 0u;
    f.wg_times = c->wg_timing ? c->wg_times.ptr : nullptr;
    f.nee_base = c->nee_base.ptr, f.nee_a = c->nee_a.ptr, f.nee_b = c->nee_b.ptr;
    f.nee_point = c->nee_point.ptr, f.nee_dir = c->nee_dir.ptr, f.nee_term = c->nee_term.ptr;
    const bool sorting = sort_enabled(c);
    f.sort_key = sorting ? c->sort_keys.ptr : nullptr;
    f.perm = sorting ? c->sort_perm.ptr : nullptr;  // always a valid permutation (identity until the first sort)
    const bool shadow_sorting = sorting && c->shadow_sort != 0 && c->pipeline == 1 && defer_shadows(c);
    f.shadow_key = shadow_sorting ? c->shadow_keys.ptr : nullptr;
    f.shadow_perm = shadow_sorting ? c->shadow_perm.ptr : nullptr;
    return f;
}

DConfig make_config(const hiprz_ctx* c) {
    return DConfig{c->config.max_depth, c->config.spot_samples, c->config.direct_samples, c->config.seed};
}

// The binned walk pays off when a mesh visit is short and uniform (every mesh tree is a single leaf, e.g. the
// Cornell configs: 329 vs 370 us per pass); with deep mesh trees a round lasts as long as its slowest item
// and the nested walk is faster (config C: 1 668 vs 2 450 us).
int effective_mode(const hiprz_ctx* c) {
    if (c->traversal_mode >= 0) return c->traversal_mode;
    // records do not fit LDS: nested skip-link walk with cached tree tops.  (MODE 4, persistent lanes on the flat walk
    // graph, is selectable but measured slower — config D 6.1 vs 3.5 ms: once lanes are desynchronised every loop
    // iteration pays for the instance-entry / exit / refill blocks.)
    if (!c->lds_scene && c->pipeline == 1) return 3;
    return c->dscene.mesh_stack_entries <= 2u ? 2 : 1;
}

// Shadow rays get their own kernel when the scene has lights, is not staged in LDS (split pipeline) and the sample slots of a
// segment fit the 30-bit mask of the hand-over record.
bool defer_shadows(const hiprz_ctx* c) {
    return c->defer_shadow_rays && c->pipeline == 1 && !use_lds_scene(c) && c->dscene.n_spot_lights + c->dscene.n_direct_lights != 0u &&
           c->config.spot_samples + c->config.direct_samples <= 30u;
}

void resolve_pipeline(hiprz_ctx* c) {
    if (c->pipeline_setting >= 0) c->pipeline = c->pipeline_setting;
    else {
        // resident needs blob + walk workspace + 8 KiB of parked state per workgroup, four workgroups per CU
        const size_t lds = size_t(c->dscene.hot_bytes) + size_t(c->stack_entries) * 1024u + BinnedLds::kFixedBytes + 8u * 1024u;
        const bool mode_ok = c->traversal_mode == -1 || c->traversal_mode == 1 || c->traversal_mode == 2;  // walks the batch kernel has
        c->pipeline = (c->have_scene && c->lds_scene && c->lds_scene_override != 0 && mode_ok && lds <= 40u * 1024u) ? 2 : 1;
    }
}

constexpr uint32_t kTopCacheNodes = 682u;  // 682 x 36 B = 24 KiB per workgroup: ~9 levels of every tree, 5 workgroups per CU

constexpr size_t kLdsSceneLimit = 52u * 1024u;  // per workgroup: 3 x 52 KiB < 160 KiB per CU

bool use_lds_scene(const hiprz_ctx* c) {
    if (c->lds_scene_override == 0) return false;
    if (c->lds_scene_override == 1) return size_t(c->dscene.hot_bytes) + size_t(c->stack_entries) * 1024u <= 160u * 1024u;
    return c->lds_scene;
}

void launch_sort(hiprz_ctx* c);
void launch_shadow_sort(hiprz_ctx* c);
template <bool FIRST, bool COUNT>
void launch_pass(hiprz_ctx* c, const DFrame& f, hipEvent_t between_trace_and_shade = nullptr) {
    c->sorted_this_pass = false;
    // with the XCD swizzle the grid is padded to a multiple of 8 workgroups (the extra ones find no tile)
    const dim3 grid(c->xcd_swizzle ? ((c->n_local_tiles + 7u) / 8u) * 8u : c->n_local_tiles), block(256);
    const DConfig cfg = make_config(c);
    const bool lds_scene = use_lds_scene(c);
    const size_t blob = lds_scene ? c->dscene.hot_bytes : 0u;
    int mode = effective_mode(c);
    if (mode >= 3 && (lds_scene || c->pipeline != 1)) mode = 1;  // the top cache is for scenes that are not staged whole, in the trace kernel
    const size_t stack_lds = size_t(c->stack_entries) * 256u * sizeof(uint32_t);
    const size_t walk_lds = mode == 2 ? size_t(BinnedLds::bytes_host(c->dscene.world_stack_entries, c->dscene.mesh_stack_entries))
                            : mode == 1 ? stack_lds : 0u;
#define RZ_LAUNCH(kernel_lds, kernel_global, lds_bytes, ...)                                                        \
    do {                                                                                                            \
        if (lds_scene) hipLaunchKernelGGL(kernel_lds, grid, block, blob + (lds_bytes), c->stream, __VA_ARGS__);     \
        else hipLaunchKernelGGL(kernel_global, grid, block, (lds_bytes), c->stream, __VA_ARGS__);                   \
    } while (0)
    if (c->pipeline == 1) {
        if (mode == 4) {
            const uint32_t pools = (c->n_local_tiles + RZ_POOL_FACTOR - 1u) / RZ_POOL_FACTOR;
            hipLaunchKernelGGL((rz_trace_persistent_kernel<FIRST, COUNT>), dim3(pools), block, c->dscene.wtop_count * 36u + 16u, c->stream, c->dscene, c->dcamera, f);
        } else if (mode == 5) {
            // round 0 over all owned pixels, then one launch per scheduled round over the rays the previous one queued
            // (the grid is sized for the worst case; workgroups past the queue's end return at once)
            const uint32_t rounds = uint32_t(c->requeue_thresholds.size());
            const size_t top_lds = TopCache::bytes_host(c->dscene.top_count);
            (void)hipMemsetAsync(c->rq_counts.ptr, 0, (rounds + 2u) * sizeof(uint32_t), c->stream);
            for (uint32_t r = 0; r <= rounds; ++r) {
                const int in = int((r + 1u) & 1u), out = int(r & 1u);
                DRequeue q{c->rq0[in].ptr, c->rq1[in].ptr, c->rq2[in].ptr, c->rq0[out].ptr, c->rq1[out].ptr, c->rq2[out].ptr,
                           c->rq_counts.ptr, r, r < rounds ? c->requeue_thresholds[r] : 0u};
                if (r == 0 && rounds > 0) hipLaunchKernelGGL((rz_trace_requeue_kernel<FIRST, COUNT, true, true>), grid, block, top_lds, c->stream, c->dscene, c->dcamera, f, q);
                else if (r == 0) hipLaunchKernelGGL((rz_trace_requeue_kernel<FIRST, COUNT, true, false>), grid, block, top_lds, c->stream, c->dscene, c->dcamera, f, q);
                else if (r < rounds) hipLaunchKernelGGL((rz_trace_requeue_kernel<FIRST, COUNT, false, true>), grid, block, top_lds, c->stream, c->dscene, c->dcamera, f, q);
                else hipLaunchKernelGGL((rz_trace_requeue_kernel<FIRST, COUNT, false, false>), grid, block, top_lds, c->stream, c->dscene, c->dcamera, f, q);
            }
        } else if (mode == 6) {
            // persistent 64-lane workgroups: as many as the chip holds at once (256 CUs x 4 SIMDs x RZ_POOL_MIN_WAVES), but no
            // more than there are waves of rays
            const uint32_t top_n = std::min<uint32_t>(c->dscene.top_count, kTopCacheNodes / 4u);
            const uint32_t waves = std::min<uint32_t>(c->n_local_tiles * 4u, 256u * 4u * RZ_POOL_MIN_WAVES);
            (void)hipMemsetAsync(c->rq_counts.ptr, 0, sizeof(uint32_t), c->stream);
            hipLaunchKernelGGL((rz_trace_pool_kernel<FIRST, COUNT>), dim3(waves), dim3(64), TopCache::bytes_host(top_n), c->stream, c->dscene,
                               c->dcamera, f, top_n, c->rq_counts.ptr, c->pool_threshold);
        } else if (mode == 3 && c->trace_wg != 256) {
            const uint32_t n_wg = c->n_local_tiles * 4u;
            const bool big_trees = c->trace_waves > 0 ? c->trace_waves >= 6 : c->n_nodes > kLatencyBoundNodes;
            // tree-top cache: 160 KiB of LDS over 24 (6 waves per SIMD) or 16 (4) single-wave workgroups per CU
            if (COUNT ? c->walk_order == 2 : c->walk_order != 0) {  // front-to-back mesh walks: 48 B of LDS per cached node instead of 36
                if (c->coop_walk) {
                    // cooperative triangle phase; 4 waves per SIMD for every tree size (D: 1 037 us against 1 131 us with 6 waves)
                    if (c->trace_waves >= 6) hipLaunchKernelGGL((rz_trace_coop_kernel<FIRST, COUNT, 6>), dim3(n_wg), dim3(64), CoopLds::kBytes, c->stream, c->dscene, c->dcamera, f);
                    else hipLaunchKernelGGL((rz_trace_coop_kernel<FIRST, COUNT, 4>), dim3(n_wg), dim3(64), CoopLds::kBytes, c->stream, c->dscene, c->dcamera, f);
                } else {
                const size_t park = TopCache::park_bytes_host();  // no tree-top cache: see fetch_node_ordered
                if (big_trees) hipLaunchKernelGGL((rz_trace_skip_kernel<FIRST, COUNT, 6, true>), dim3(n_wg), dim3(64), park, c->stream, c->dscene, c->dcamera, f, 0u);
                else hipLaunchKernelGGL((rz_trace_skip_kernel<FIRST, COUNT, 4, true>), dim3(n_wg), dim3(64), park, c->stream, c->dscene, c->dcamera, f, 0u);
                }
            } else {
                const uint32_t top_n = std::min<uint32_t>(c->dscene.top_count, big_trees ? 170u : 272u);
                if (big_trees) hipLaunchKernelGGL((rz_trace_skip_kernel<FIRST, COUNT, 6, false>), dim3(n_wg), dim3(64), TopCache::bytes_host(top_n), c->stream, c->dscene, c->dcamera, f, top_n);
                else hipLaunchKernelGGL((rz_trace_skip_kernel<FIRST, COUNT, 4, false>), dim3(n_wg), dim3(64), TopCache::bytes_host(top_n), c->stream, c->dscene, c->dcamera, f, top_n);
            }
        } else if (mode == 3) hipLaunchKernelGGL((rz_trace_kernel<FIRST, COUNT, 3, false>), grid, block, TopCache::bytes_host(c->dscene.top_count), c->stream, c->dscene, c->dcamera, f);
        else if (mode == 2) RZ_LAUNCH((rz_trace_kernel<FIRST, COUNT, 2, true>), (rz_trace_kernel<FIRST, COUNT, 2, false>), walk_lds, c->dscene, c->dcamera, f);
        else if (mode == 1) RZ_LAUNCH((rz_trace_kernel<FIRST, COUNT, 1, true>), (rz_trace_kernel<FIRST, COUNT, 1, false>), walk_lds, c->dscene, c->dcamera, f);
        else RZ_LAUNCH((rz_trace_kernel<FIRST, COUNT, 0, true>), (rz_trace_kernel<FIRST, COUNT, 0, false>), walk_lds, c->dscene, c->dcamera, f);
        if (between_trace_and_shade) (void)hipEventRecord(between_trace_and_shade, c->stream);
        // shadow rays: LDS-stack walk on a staged scene, skip-link walk with staged tree tops otherwise (no lights: no walk at all)
        const bool lights = c->dscene.n_spot_lights + c->dscene.n_direct_lights != 0u;
        const uint32_t shade_top = std::min<uint32_t>(c->dscene.top_count, kTopCacheNodes);
        if (!lights && c->nolight_kernels && c->n_textures == 0u) {  // no lights, no maps
            if (lds_scene) hipLaunchKernelGGL((rz_shade_kernel<FIRST, COUNT, true, RZ_SHADOW_PLAIN>), grid, block, blob, c->stream, c->dscene, c->dcamera, cfg, f, 0u);
            else hipLaunchKernelGGL((rz_shade_kernel<FIRST, COUNT, false, RZ_SHADOW_PLAIN>), grid, block, 0, c->stream, c->dscene, c->dcamera, cfg, f, 0u);
        } else if (!lights && c->nolight_kernels) {  // no next-event estimation: the instantiation without it (no shadow walk, no LDS stack)
            if (lds_scene) hipLaunchKernelGGL((rz_shade_kernel<FIRST, COUNT, true, RZ_SHADOW_NONE>), grid, block, blob, c->stream, c->dscene, c->dcamera, cfg, f, 0u);
            else hipLaunchKernelGGL((rz_shade_kernel<FIRST, COUNT, false, RZ_SHADOW_NONE>), grid, block, 0, c->stream, c->dscene, c->dcamera, cfg, f, 0u);
        } else if (lds_scene) hipLaunchKernelGGL((rz_shade_kernel<FIRST, COUNT, true, 1>), grid, block, blob + stack_lds, c->stream, c->dscene, c->dcamera, cfg, f, 0u);
        else if (lights && defer_shadows(c)) {
            // shading without shadow walks, then every shadow ray of the pass in a lean single-wave kernel
            hipLaunchKernelGGL((rz_shade_kernel<FIRST, COUNT, false, RZ_SHADOW_DEFER>), grid, block, 0, c->stream, c->dscene, c->dcamera, cfg, f, 0u);
            // the shadow rays start where the next segment's rays start: walk them in the order the next trace kernel will use
            // (origin cell + direction of the next ray), so that a wave's rays meet the same instances
            launch_sort(c);
            if (f.shadow_key) launch_shadow_sort(c);
            const bool big_trees = c->trace_waves > 0 ? c->trace_waves >= 6 : c->n_nodes > kLatencyBoundNodes;
            const dim3 sgrid(c->n_local_tiles * 4u), sblock(64);
            if ((COUNT ? c->walk_order == 2 : c->walk_order != 0) && c->coop_shadow) {
                hipLaunchKernelGGL((rz_shadow_coop_kernel<FIRST, COUNT, 4>), sgrid, sblock, CoopLds::kBytes, c->stream, c->dscene, c->dcamera, cfg, f);
            } else if (COUNT ? c->walk_order == 2 : c->walk_order != 0) {
                if (big_trees) hipLaunchKernelGGL((rz_shadow_kernel<FIRST, COUNT, 6, true>), sgrid, sblock, 0, c->stream, c->dscene, c->dcamera, cfg, f, 0u);
                else hipLaunchKernelGGL((rz_shadow_kernel<FIRST, COUNT, 4, true>), sgrid, sblock, 0, c->stream, c->dscene, c->dcamera, cfg, f, 0u);
            } else {
                const uint32_t top_n = std::min<uint32_t>(c->dscene.top_count, big_trees ? 170u : 272u);
                if (big_trees) hipLaunchKernelGGL((rz_shadow_kernel<FIRST, COUNT, 6, false>), sgrid, sblock, TopCache::bytes_host(top_n), c->stream, c->dscene, c->dcamera, cfg, f, top_n);
                else hipLaunchKernelGGL((rz_shadow_kernel<FIRST, COUNT, 4, false>), sgrid, sblock, TopCache::bytes_host(top_n), c->stream, c->dscene, c->dcamera, cfg, f, top_n);
            }
        } else if (lights && c->shade_shadow_walk == 3) hipLaunchKernelGGL((rz_shade_kernel<FIRST, COUNT, false, 3>), grid, block, TopCache::bytes_host(shade_top), c->stream, c->dscene, c->dcamera, cfg, f, shade_top);
        else hipLaunchKernelGGL((rz_shade_kernel<FIRST, COUNT, false, 1>), grid, block, stack_lds, c->stream, c->dscene, c->dcamera, cfg, f, 0u);
    } else {
        // the fused kernel's shadow rays use the stack walk: its columns must exist in every mode
        const size_t fused_lds = mode == 0 ? stack_lds : mode == 2 ? walk_lds + 4096u : walk_lds;  // mode 2: + the parked path state
        if (mode == 2) RZ_LAUNCH((rz_pass_kernel<FIRST, COUNT, 2, true>), (rz_pass_kernel<FIRST, COUNT, 2, false>), fused_lds, c->dscene, c->dcamera, cfg, f);
        else if (mode == 1) RZ_LAUNCH((rz_pass_kernel<FIRST, COUNT, 1, true>), (rz_pass_kernel<FIRST, COUNT, 1, false>), fused_lds, c->dscene, c->dcamera, cfg, f);
        else RZ_LAUNCH((rz_pass_kernel<FIRST, COUNT, 0, true>), (rz_pass_kernel<FIRST, COUNT, 0, false>), fused_lds, c->dscene, c->dcamera, cfg, f);
    }
#undef RZ_LAUNCH
}

// resident pipeline: all `n` cumulative passes of the batch in one launch (+ one launch that advances the pass index)
template <bool COUNT>
void launch_batch(hiprz_ctx* c, const DFrame& f, uint32_t n, hipEvent_t before = nullptr, hipEvent_t after = nullptr) {
    const dim3 grid(c->xcd_swizzle ? ((c->n_local_tiles + 7u) / 8u) * 8u : c->n_local_tiles), block(256);
    const DConfig cfg = make_config(c);
    const bool lds_scene = use_lds_scene(c);
    const size_t blob = lds_scene ? c->dscene.hot_bytes : 0u;
    int mode = effective_mode(c);
    if (mode != 2) mode = 1;
    const size_t stack_lds = size_t(c->stack_entries) * 256u * sizeof(uint32_t);
    const size_t walk_lds = mode == 2 ? size_t(BinnedLds::bytes_host(c->dscene.world_stack_entries, c->dscene.mesh_stack_entries)) : stack_lds;
    const size_t park = 8u * 1024u;
    const size_t lds = blob + walk_lds + park;
    const uint32_t park_offset = uint32_t(walk_lds);
    if (before) (void)hipEventRecord(before, c->stream);
    // scenes without lights run the instantiation whose next-event-estimation code is compiled out (RZ_SHADOW_NONE), scenes that
    // have no maps either the one without texture fetches and normal mapping (RZ_SHADOW_PLAIN)
    const bool dark = c->dscene.n_spot_lights + c->dscene.n_direct_lights == 0u && c->nolight_kernels;
    const bool plain = dark && c->n_textures == 0u;
    // 5 workgroups per CU must fit LDS, and the grid must be more than two full loads of the chip (256 CUs x 5)
    const bool five = lds * 5u <= 160u * 1024u && grid.x > 2u * 5u * 256u && c->batch_waves != 4;
#define RZ_BATCH(M, L)                                                                                                                     \
    do {                                                                                                                                   \
        if (plain && five) hipLaunchKernelGGL((rz_batch_kernel<COUNT, M, L, RZ_SHADOW_PLAIN, 5>), grid, block, lds, c->stream, c->dscene, c->dcamera, cfg, f, n, park_offset); \
        else if (plain) hipLaunchKernelGGL((rz_batch_kernel<COUNT, M, L, RZ_SHADOW_PLAIN>), grid, block, lds, c->stream, c->dscene, c->dcamera, cfg, f, n, park_offset); \
        else if (dark) hipLaunchKernelGGL((rz_batch_kernel<COUNT, M, L, RZ_SHADOW_NONE>), grid, block, lds, c->stream, c->dscene, c->dcamera, cfg, f, n, park_offset); \
        else hipLaunchKernelGGL((rz_batch_kernel<COUNT, M, L, 1>), grid, block, lds, c->stream, c->dscene, c->dcamera, cfg, f, n, park_offset);     \
    } while (0)
    if (mode == 2) {
        if (lds_scene) RZ_BATCH(2, true);
        else RZ_BATCH(2, false);
    } else {
        if (lds_scene) RZ_BATCH(1, true);
        else RZ_BATCH(1, false);
    }
#undef RZ_BATCH
    if (after) (void)hipEventRecord(after, c->stream);
    hipLaunchKernelGGL(rz_pass_add_kernel, dim3(1), dim3(1), 0, c->stream, c->pass_dev.ptr, n);
    c->rgba8_valid = true;
}

hipEvent_t take_event(hiprz_ctx* c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void drop_graph(hiprz_ctx* c) {
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    c->graph_exec = nullptr;
    c->graph_valid = false;
}

// Two radix passes (16 key bits) are enough while a bin of the coarser order still holds a wave's worth of rays: up to ~2 M owned
// pixels (config C: step 4.53 -> 4.33 ms, the sort 88 -> 59 us per pass).  Bigger frames and scenes with lights (whose shadow rays
// follow a sorted order of their own) keep all 24 bits (config E: 16 bits 55.9 ms per step against 51.0).
int effective_sort_bits(const hiprz_ctx* c) {
    if (c->sort_bits > 0) return c->sort_bits;
    const bool lights = c->dscene.n_spot_lights + c->dscene.n_direct_lights != 0u;
    return (!lights && size_t(c->n_local_tiles) * 256u <= (size_t(32) << 16)) ? 16 : 24;
}

// radix sort of the keys the shade kernel just wrote -> permutation the next trace kernel follows
void launch_sort(hiprz_ctx* c) {
    if (!sort_enabled(c) || c->n_local_tiles == 0 || c->sorted_this_pass) return;
    c->sorted_this_pass = true;
    size_t bytes = c->sort_temp_bytes;
    (void)hipcub::DeviceRadixSort::SortPairs(c->sort_temp.ptr, bytes, c->sort_keys.ptr, c->sort_keys_out.ptr, c->sort_iota.ptr,
                                             c->sort_perm.ptr, int(c->n_local_tiles * 256u), 24 - effective_sort_bits(c), 24, c->stream);
}

// the same for the keys of the pass's shadow rays -> the order rz_shadow_kernel follows
void launch_shadow_sort(hiprz_ctx* c) {
    if (c->n_local_tiles == 0) return;
    size_t bytes = c->sort_temp_bytes;
    (void)hipcub::DeviceRadixSort::SortPairs(c->sort_temp.ptr, bytes, c->shadow_keys.ptr, c->sort_keys_out.ptr, c->sort_iota.ptr,
                                             c->shadow_perm.ptr, int(c->n_local_tiles * 256u), 24 - effective_sort_bits(c), 24, c->stream);
}

// [cumulative pass, sort, pass update] x n on the stream — eagerly, or into a capture
void enqueue_cumulative(hiprz_ctx* c, const DFrame& f, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) {
        launch_pass<false, false>(c, f);
        launch_sort(c);
        hipLaunchKernelGGL(rz_pass_update_kernel, dim3(1), dim3(1), 0, c->stream, c->pass_dev.ptr);
    }
}

int finish_batch(hiprz_ctx* c, hipEvent_t e0, hipEvent_t e1, uint32_t n_passes, const StageTimer& timer) {
    RZ_HIP(c, hipEventRecord(e1, c->stream));
    RZ_HIP(c, hipGetLastError());
    if (c->pending_events.size() >= 4096) {  // nobody is collecting timings: recycle the oldest pair
        c->event_pool.push_back(c->pending_events.front().first);
        c->event_pool.push_back(c->pending_events.front().second);
        c->pending_events.erase(c->pending_events.begin());
        c->pending_launches.erase(c->pending_launches.begin());
    }
    c->pending_events.emplace_back(e0, e1);
    c->pending_launches.push_back(n_passes);
    c->timings.set("render (enqueue)", timer.ms());
    return HIPRZ_OK;
}

int render_impl(hiprz_ctx* c, uint32_t n_passes, bool counted) {
    if (!c->have_scene || !c->have_camera) return fail(c, HIPRZ_ERR_STATE, "render before scene and camera upload");
    if (n_passes == 0 || c->n_local_tiles == 0) return HIPRZ_OK;
    StageTimer timer;
    if (defer_shadows(c)) {  // hand-over buffers of the deferred shadow rays: (4 + 2 * samples) float4 per owned pixel
        const size_t n = size_t(c->n_local_tiles) * 256u, k = c->config.spot_samples + c->config.direct_samples;
        if (c->nee_dir.count < n * k || c->nee_base.count < n) c->graph_valid = false;
        RZ_HIP(c, c->nee_base.resize(n));
        RZ_HIP(c, c->nee_a.resize(n));
        RZ_HIP(c, c->nee_b.resize(n));
        RZ_HIP(c, c->nee_point.resize(n));
        RZ_HIP(c, c->nee_dir.resize(n * k));
        RZ_HIP(c, c->nee_term.resize(n * k));
    }
    if (effective_mode(c) == 6) RZ_HIP(c, c->rq_counts.resize(32));
    if (effective_mode(c) == 5 && c->pipeline == 1 && !use_lds_scene(c)) {
        const size_t n = size_t(c->n_local_tiles) * 256u;
        for (int k = 0; k < 2; ++k) {
            RZ_HIP(c, c->rq0[k].resize(n));
            RZ_HIP(c, c->rq1[k].resize(n));
            RZ_HIP(c, c->rq2[k].resize(n));
        }
        RZ_HIP(c, c->rq_counts.resize(32));
    }
    const DFrame f = make_frame(c, counted);
    hipEvent_t e0 = take_event(c), e1 = take_event(c);
    RZ_HIP(c, hipEventRecord(e0, c->stream));
    c->rgba8_valid = false;
    if (c->pipeline == 2) {
        uint32_t remaining = n_passes;
        if (c->reset_pending) {  // renderFirstPass: the fused kernel
            hipLaunchKernelGGL(rz_pass_reset_kernel, dim3(1), dim3(1), 0, c->stream, c->pass_dev.ptr);
            if (counted) launch_pass<true, true>(c, f);
            else launch_pass<true, false>(c, f);
            hipLaunchKernelGGL(rz_pass_update_kernel, dim3(1), dim3(1), 0, c->stream, c->pass_dev.ptr);
            c->reset_pending = false;
            c->passes = 1;
            c->ray_count = c->owned_pixels;
            remaining -= 1u;
        }
        if (remaining) {
            c->kernel_event_passes = 0;
            if (counted) launch_batch<true>(c, f, remaining);
            else if (c->time_kernels) {  // bench.py's roofline: the batch kernel's own duration
                while (c->kernel_events.size() < 3u) {
                    hipEvent_t e = nullptr;
                    (void)hipEventCreate(&e);
                    c->kernel_events.push_back(e);
                }
                launch_batch<false>(c, f, remaining, c->kernel_events[0], c->kernel_events[1]);
                c->kernel_event_passes = remaining;
            } else launch_batch<false>(c, f, remaining);
            c->passes += remaining;
            c->ray_count += uint64_t(remaining) * c->owned_pixels;
        }
        return finish_batch(c, e0, e1, n_passes, timer);
    }
    if (c->use_graph && !c->time_kernels && !counted && !c->reset_pending && n_passes >= 2) {
        // steady state: one graph launch instead of 2 * n_passes kernel launches
        if (!c->graph_valid || c->graph_passes != n_passes) {
            drop_graph(c);
            hipGraph_t graph = nullptr;
            RZ_HIP(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            enqueue_cumulative(c, f, n_passes);
            RZ_HIP(c, hipStreamEndCapture(c->stream, &graph));
            const hipError_t ie = hipGraphInstantiate(&c->graph_exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ie != hipSuccess) return fail(c, HIPRZ_ERR_DEVICE, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie));
            c->graph_passes = n_passes;
            c->graph_valid = true;
        }
        RZ_HIP(c, hipGraphLaunch(c->graph_exec, c->stream));
        c->passes += n_passes;
        c->ray_count += uint64_t(n_passes) * c->owned_pixels;
        return finish_batch(c, e0, e1, n_passes, timer);
    }
    // kernel-level timing (bench.py's roofline): events before the trace kernel, between the two kernels and after the
    // shade kernel of every pass.  Event timing does not work from inside a captured graph, so a timed batch is launched
    // eagerly.
    const bool timed = c->time_kernels && c->pipeline == 1 && !counted && !c->reset_pending;
    if (timed) {
        while (c->kernel_events.size() < size_t(3 * n_passes)) {
            hipEvent_t e = nullptr;
            (void)hipEventCreate(&e);
            c->kernel_events.push_back(e);
        }
        c->kernel_event_passes = n_passes;
    }
    for (uint32_t i = 0; i < n_passes; ++i) {
        if (timed) {
            (void)hipEventRecord(c->kernel_events[3 * i], c->stream);
            launch_pass<false, false>(c, f, c->kernel_events[3 * i + 1]);
            (void)hipEventRecord(c->kernel_events[3 * i + 2], c->stream);
            launch_sort(c);
            hipLaunchKernelGGL(rz_pass_update_kernel, dim3(1), dim3(1), 0, c->stream, c->pass_dev.ptr);
            c->passes += 1;
            c->ray_count += c->owned_pixels;
            continue;
        }
        if (c->reset_pending) {
            hipLaunchKernelGGL(rz_pass_reset_kernel, dim3(1), dim3(1), 0, c->stream, c->pass_dev.ptr);
            if (counted) launch_pass<true, true>(c, f);
            else launch_pass<true, false>(c, f);
            c->reset_pending = false;
            c->passes = 0;
            c->ray_count = 0;
        } else {
            if (counted) launch_pass<false, true>(c, f);
            else launch_pass<false, false>(c, f);
        }
        launch_sort(c);
        hipLaunchKernelGGL(rz_pass_update_kernel, dim3(1), dim3(1), 0, c->stream, c->pass_dev.ptr);
        c->passes += 1;
        c->ray_count += c->owned_pixels;  // traced_rays += W*H per pass (cpu_engine_renderer.cpp:173), per shard
    }
    return finish_batch(c, e0, e1, n_passes, timer);
}

template <typename T>
int read_untiled(hiprz_ctx* c, const T* tiles, T* dst, size_t bytes, const char* what) {
    if (!c->have_camera) return fail(c, HIPRZ_ERR_STATE, "readback before camera upload");
    const size_t n = size_t(c->camera.width) * c->camera.height;
    if (!dst || bytes != n * sizeof(T)) return fail(c, HIPRZ_ERR_INVALID, std::string(what) + ": destination size mismatch");
    StageTimer timer;
    T* image = reinterpret_cast<T*>(c->image_f4.ptr);
    RZ_HIP(c, hipMemsetAsync(image, 0, bytes, c->stream));
    if (c->n_local_tiles)
        hipLaunchKernelGGL((rz_untile_kernel<T>), dim3(c->n_local_tiles), dim3(256), 0, c->stream, tiles, image,
                           c->camera.width, c->camera.height, c->tiles_x, c->rank, c->world);
    RZ_HIP(c, hipMemcpyAsync(dst, image, bytes, hipMemcpyDeviceToHost, c->stream));
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    c->timings.set(what, timer.ms());
    return HIPRZ_OK;
}

// Everything the kernels will dereference is checked here, on the host, before any launch: a
// bad index or a cyclic tree would otherwise fault or hang the GPU.  Also derives the skip links,
// the world-tree leaves and the tree depths the upload needs.
struct SceneCheck {
    std::string error;
    std::vector<uint32_t> skip;
    std::vector<uint32_t> world_leaves;
    uint32_t world_depth = 0, mesh_depth = 0;
};
int check_scene(const hiprz_scene* sc, SceneCheck& out) {
    auto bad = [&out](const std::string& m) {
        out.error = m;
        return HIPRZ_ERR_INVALID;
    };
    if (!sc) return bad("scene is null");
    // ---- validate everything the kernels will dereference, on the host, before any launch ----
    if (sc->n_materials < 2 || !sc->materials) return bad("scene needs materials[0]=world, [1]=default");
    if (sc->n_materials > 65536u) return bad("more than 65536 materials");
    if ((sc->n_nodes && !sc->nodes) || (sc->n_tris && (!sc->tris || !sc->tri_attrs)) || (sc->n_instances && !sc->instances) ||
        (sc->n_tlas_order && !sc->tlas_order) || (sc->n_inst_materials && !sc->inst_materials) ||
        (sc->n_textures && !sc->textures) || (sc->texel_bytes && !sc->texels) || (sc->n_spot_lights && !sc->spot_lights) ||
        (sc->n_direct_lights && !sc->direct_lights))
        return bad("upload_scene: null array with non-zero count");
    for (uint32_t i = 0; i < sc->n_textures; ++i) {
        const hiprz_texture& t = sc->textures[i];
        const uint64_t texel = t.kind == HIPRZ_TEX_R8 ? 1u : 4u;
        if (t.kind > HIPRZ_TEX_R32F || t.width == 0 || t.height == 0 || (t.offset & 3u) ||
            uint64_t(t.offset) + texel * t.width * t.height > sc->texel_bytes)
            return bad("texture " + std::to_string(i) + ": bad kind/size/offset");
    }
    auto tex_ok = [&](int32_t t, uint32_t kind) { return t < 0 || (uint32_t(t) < sc->n_textures && sc->textures[t].kind == kind); };
    for (uint32_t i = 0; i < sc->n_materials; ++i) {
        const hiprz_material& m = sc->materials[i];
        if (!tex_ok(m.texture, HIPRZ_TEX_RGBA8) || !tex_ok(m.normal_map, HIPRZ_TEX_RGBA8) ||
            !tex_ok(m.metalness_map, HIPRZ_TEX_R8) || !tex_ok(m.roughness_map, HIPRZ_TEX_R8) ||
            !tex_ok(m.emission_map, HIPRZ_TEX_R32F))
            return bad("material " + std::to_string(i) + ": map index/kind invalid");
    }
    for (uint32_t i = 0; i < sc->n_inst_materials; ++i)
        if (sc->inst_materials[i] >= int32_t(sc->n_materials))
            return bad("inst_materials[" + std::to_string(i) + "] out of range");
    for (uint32_t i = 0; i < sc->n_tlas_order; ++i)
        if (sc->tlas_order[i] >= sc->n_instances) return bad("tlas_order entry out of range");
    for (uint32_t i = 0; i < sc->n_instances; ++i) {
        const hiprz_instance& in = sc->instances[i];
        if (in.material_count > 64u || uint64_t(in.material_base) + in.material_count > sc->n_inst_materials)
            return bad("instance " + std::to_string(i) + ": material table out of range");
    }
    std::vector<uint32_t> skip(sc->n_nodes ? sc->n_nodes : 1, RZ_END);
    TreeCheck check{sc, skip, std::vector<uint8_t>(sc->n_nodes ? sc->n_nodes : 1, 0)};
    uint32_t world_depth = 0, mesh_depth = 0;
    if (sc->n_instances) {
        if (!check.walk(sc->tlas_root, true)) return bad("world tree: " + check.error);
        world_depth = check.max_depth;
        std::vector<uint8_t> root_seen(sc->n_nodes, 0);
        for (uint32_t i = 0; i < sc->n_tlas_order; ++i) {
            const uint32_t root = sc->instances[sc->tlas_order[i]].blas_root;
            if (root >= sc->n_nodes) return bad("instance mesh root out of range");
            if (root_seen[root]) continue;
            root_seen[root] = 1;
            check.max_depth = 0;
            if (!check.walk(root, false)) return bad("mesh tree: " + check.error);
            mesh_depth = std::max(mesh_depth, check.max_depth);
        }
    }
    out.skip = std::move(skip);
    out.world_leaves = std::move(check.world_leaves);
    out.world_depth = world_depth, out.mesh_depth = mesh_depth;
    return HIPRZ_OK;
}

// Device-side tables derived from a validated scene (pure host): relayouted nodes + links, walk graph.
struct DerivedTables {
    std::vector<uint32_t> new_index;
    std::vector<hiprz_node> dnodes, wnodes;
    std::vector<uint32_t> dskip, wskip;
    std::vector<uint32_t> dskip8;  // [node][octant]: skip links of the front-to-back mesh walk (hiprz_device.hpp: fetch_node_ordered)
};
int derive_tables(const hiprz_scene* sc, SceneCheck& chk, DerivedTables& out) {
    // Relayout: breadth-first over ALL trees at once (world root, then every distinct mesh root, then their child
    // pairs, ...), children staying adjacent.  The levels nearest the roots become a prefix of the array (the part
    // MODE 3 caches in LDS) and siblings/cousins share cache lines.  Leaf ranges are untouched.
    std::vector<uint32_t>& new_index = out.new_index;
    new_index.assign(sc->n_nodes, RZ_END);
    std::vector<uint32_t> bfs;
    bfs.reserve(sc->n_nodes);
    auto enqueue = [&](uint32_t old) {
        if (old < sc->n_nodes && new_index[old] == RZ_END) {
            new_index[old] = uint32_t(bfs.size());
            bfs.push_back(old);
        }
    };
    if (sc->n_instances) enqueue(sc->tlas_root);
    for (uint32_t i = 0; i < sc->n_tlas_order; ++i) enqueue(sc->instances[sc->tlas_order[i]].blas_root);
    for (size_t q = 0; q < bfs.size(); ++q) {
        const hiprz_node& n = sc->nodes[bfs[q]];
        if (!(n.meta & HIPRZ_NODE_LEAF)) enqueue(n.begin), enqueue(n.begin + 1);
    }
    for (uint32_t old = 0; old < sc->n_nodes; ++old) enqueue(old);  // nodes no instance reaches keep a slot
    std::vector<hiprz_node>& dnodes = out.dnodes;
    std::vector<uint32_t>& dskip = out.dskip;
    dnodes.assign(sc->n_nodes, hiprz_node{});
    dskip.assign(sc->n_nodes ? sc->n_nodes : 1, RZ_END);
    for (uint32_t old = 0; old < sc->n_nodes; ++old) {
        hiprz_node n = sc->nodes[old];
        if (!(n.meta & HIPRZ_NODE_LEAF)) n.begin = new_index[n.begin];
        dnodes[new_index[old]] = n;
        dskip[new_index[old]] = chk.skip[old] == RZ_END ? RZ_END : new_index[chk.skip[old]];
    }

    // ---- skip links per ray octant (front-to-back walk) ----
    // Under octant o an inner node with partition type p (X=2, Y=1, Z=0) is left towards its SECOND child first when bit p of o is
    // set; a size split (type 3) is never flipped.  The child visited first links to its sibling, the other one inherits the
    // parent's link.  Parents precede their children in the breadth-first numbering, so one ascending sweep fills all tables;
    // roots end their walks (RZ_END).  Octant 0 reproduces dskip.
    std::vector<uint32_t>& dskip8 = out.dskip8;
    dskip8.assign(size_t(sc->n_nodes ? sc->n_nodes : 1) * 8u, RZ_END);
    for (uint32_t n = 0; n < sc->n_nodes; ++n) {
        const hiprz_node& nd = dnodes[n];
        if (nd.meta & HIPRZ_NODE_LEAF) continue;
        const uint32_t ptype = (nd.meta >> HIPRZ_NODE_PTYPE_SHIFT) & 3u, c0 = nd.begin;
        if (c0 <= n || size_t(c0) + 1 >= sc->n_nodes) continue;  // cannot happen after check_scene + the BFS relayout; keeps the sweep safe
        for (uint32_t o = 0; o < 8u; ++o) {
            const uint32_t flip = (o >> ptype) & 1u;  // ptype 3 reads bit 3 = 0
            dskip8[size_t(c0 + flip) * 8u + o] = c0 + 1u - flip;
            dskip8[size_t(c0 + 1u - flip) * 8u + o] = dskip8[size_t(n) * 8u + o];
        }
    }

    // ---- walk graph of the threaded traversal (hiprz_device.hpp: walk_threaded), in the relayouted numbering ----
    // every world-tree leaf becomes a CHAIN node whose `begin` points at a run of INSTANCE pseudo-nodes
    // (box = instance box, begin = instance id) appended after the real nodes and linked by skip; the last one
    // links to whatever follows the leaf.
    std::vector<hiprz_node>& wnodes = out.wnodes;
    std::vector<uint32_t>& wskip = out.wskip;
    wnodes = dnodes;
    wskip.assign(dskip.begin(), dskip.begin() + sc->n_nodes);
    for (auto& n : wnodes) {
        const bool leaf = (n.meta & HIPRZ_NODE_LEAF) != 0;
        n.meta = ((leaf ? RZ_WALK_TRIS : RZ_WALK_INNER) << RZ_WALK_TYPE_SHIFT) | (leaf ? (n.meta & HIPRZ_NODE_COUNT_MASK) : 0u);
    }
    for (uint32_t old_leaf : chk.world_leaves) {
        const uint32_t leaf = new_index[old_leaf];
        const hiprz_node src = dnodes[leaf];
        const uint32_t count = src.meta & HIPRZ_NODE_COUNT_MASK;
        if (count == 0) continue;  // stays an empty TRIS leaf: box test, then follow the link
        const uint32_t chain = uint32_t(wnodes.size());
        for (uint32_t k = 0; k < count; ++k) {
            const uint32_t inst = sc->tlas_order[src.begin + k];
            hiprz_node p{};
            std::memcpy(p.bb_min, sc->instances[inst].bb_min, 12);
            std::memcpy(p.bb_max, sc->instances[inst].bb_max, 12);
            p.begin = inst;
            p.meta = RZ_WALK_INSTANCE << RZ_WALK_TYPE_SHIFT;
            wnodes.push_back(p);
            wskip.push_back(k + 1 < count ? chain + k + 1 : wskip[leaf]);
        }
        wnodes[leaf].begin = chain;
        wnodes[leaf].meta = (RZ_WALK_CHAIN << RZ_WALK_TYPE_SHIFT) | count;
    }
    // The kernels follow these derived tables blindly: prove on the host that every walk over them terminates
    // (each step moves strictly forward in depth-first order, so a walk may take at most one step per node).
    {
        auto terminates = [](const std::vector<hiprz_node>& nodes, const std::vector<uint32_t>& links, uint32_t root, bool walk_graph) {
            uint32_t n = root;
            for (size_t steps = 0; steps <= nodes.size(); ++steps) {
                if (n == RZ_END) return true;
                if (n >= nodes.size()) return false;
                const hiprz_node& nd = nodes[n];
                bool descend;
                if (walk_graph) {
                    const uint32_t type = nd.meta >> RZ_WALK_TYPE_SHIFT;
                    descend = type == RZ_WALK_INNER || type == RZ_WALK_CHAIN;
                } else {
                    descend = !(nd.meta & HIPRZ_NODE_LEAF);
                }
                n = descend ? nd.begin : links[n];
            }
            return false;
        };
        bool ok = true;
        for (uint32_t old = 0; ok && old < sc->n_nodes; ++old)  // the stack walks reach the second child as first + 1
            if (!(sc->nodes[old].meta & HIPRZ_NODE_LEAF)) ok = new_index[sc->nodes[old].begin + 1] == new_index[sc->nodes[old].begin] + 1u;
        if (ok && sc->n_instances) ok = terminates(dnodes, dskip, new_index[sc->tlas_root], false) && terminates(wnodes, wskip, new_index[sc->tlas_root], true);
        for (uint32_t i = 0; ok && i < sc->n_tlas_order; ++i) {
            const uint32_t root = new_index[sc->instances[sc->tlas_order[i]].blas_root];
            ok = terminates(dnodes, dskip, root, false) && terminates(wnodes, wskip, root, true);
        }
        // the same for every octant's links: a walk that enters every box takes exactly one step per node of the tree it walks
        auto terminates8 = [&](uint32_t root, uint32_t o) {
            uint32_t n = root;
            for (size_t steps = 0; steps <= dnodes.size(); ++steps) {
                if (n == RZ_END) return true;
                if (n >= dnodes.size()) return false;
                const hiprz_node& nd = dnodes[n];
                if (!(nd.meta & HIPRZ_NODE_LEAF)) n = nd.begin + ((o >> ((nd.meta >> HIPRZ_NODE_PTYPE_SHIFT) & 3u)) & 1u);
                else n = dskip8[size_t(n) * 8u + o];
            }
            return false;
        };
        for (uint32_t n = 0; ok && n < sc->n_nodes; ++n) ok = dskip8[size_t(n) * 8u] == dskip[n];
        if (ok && sc->n_instances) ok = terminates8(new_index[sc->tlas_root], 0u);
        {
            std::vector<uint8_t> seen(sc->n_nodes ? sc->n_nodes : 1, 0);
            for (uint32_t i = 0; ok && i < sc->n_tlas_order; ++i) {
                const uint32_t root = new_index[sc->instances[sc->tlas_order[i]].blas_root];
                if (seen[root]) continue;
                seen[root] = 1;
                for (uint32_t o = 0; ok && o < 8u; ++o) ok = terminates8(root, o);
            }
        }
        if (!ok) {
            chk.error = "internal: derived walk tables are inconsistent (refusing to launch)";
            return HIPRZ_ERR_INVALID;
        }
    }
    return HIPRZ_OK;
}

}  // namespace

extern "C" {

int hiprz_validate_scene(const hiprz_scene* scene, char* message, size_t len) {
    SceneCheck chk;
    int rc = check_scene(scene, chk);
    if (rc == HIPRZ_OK) {  // also prove that the tables the kernels will follow can be derived and terminate
        DerivedTables derived;
        rc = derive_tables(scene, chk, derived);
    }
    if (message && len) std::snprintf(message, len, "%s", chk.error.c_str());
    return rc;
}

int hiprz_create(hiprz_ctx** out, int device_id) {
    if (!out) return fail(nullptr, HIPRZ_ERR_INVALID, "hiprz_create: out is null");
    *out = nullptr;
    int n_devices = 0;
    hipError_t e = hipGetDeviceCount(&n_devices);
    if (e != hipSuccess || n_devices == 0)
        return fail(nullptr, HIPRZ_ERR_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
    if (device_id < 0 || device_id >= n_devices) return fail(nullptr, HIPRZ_ERR_INVALID, "hiprz_create: device id out of range");
    e = hipSetDevice(device_id);
    if (e != hipSuccess) return fail(nullptr, HIPRZ_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) return fail(nullptr, HIPRZ_ERR_DEVICE, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, HIPRZ_ERR_DEVICE, std::string("hiprz is built for gfx950 only, device is ") + prop.gcnArchName);
    auto* c = new hiprz_ctx();
    if (const char* t = std::getenv("HIPRZ_POOL_THRESHOLD")) c->pool_threshold = uint32_t(std::atoi(t));
    if (const char* wg = std::getenv("HIPRZ_TRACE_WG")) {
        const int v = std::atoi(wg);
        if (v == 64 || v == 256) c->trace_wg = v;
    }
    if (const char* w = std::getenv("HIPRZ_TRACE_WAVES")) c->trace_waves = std::atoi(w);
    if (const char* w = std::getenv("HIPRZ_DEFER_SHADOWS")) c->defer_shadow_rays = std::atoi(w) != 0;
    if (const char* w = std::getenv("HIPRZ_COOP")) c->coop_walk = std::atoi(w) != 0;
    if (const char* w = std::getenv("HIPRZ_COOP_SHADOW")) c->coop_shadow = std::atoi(w) != 0;
    if (const char* w = std::getenv("HIPRZ_BATCH_WAVES")) c->batch_waves = std::atoi(w);
    if (const char* w = std::getenv("HIPRZ_NOLIGHT_KERNELS")) c->nolight_kernels = std::atoi(w) != 0;
    if (const char* w = std::getenv("HIPRZ_SORT_BITS")) c->sort_bits = std::min(24, std::max(0, std::atoi(w)));
    if (const char* w = std::getenv("HIPRZ_SHADOW_SORT")) c->shadow_sort = std::atoi(w) != 0;
    if (const char* w = std::getenv("HIPRZ_SHADOW_WALK")) c->shade_shadow_walk = std::atoi(w) == 1 ? 1 : 3;
    c->device = device_id;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = c->pass_dev.resize(1);
    if (e == hipSuccess) e = c->counters_dev.resize(16);
    if (e == hipSuccess) e = c->pick_dev.resize(2);
    if (e == hipSuccess) e = hipMemsetAsync(c->pass_dev.ptr, 0, sizeof(uint32_t), c->stream);
    if (e != hipSuccess) {
        const std::string msg = std::string("context setup: ") + hipGetErrorString(e);
        hiprz_destroy(c);
        return fail(nullptr, HIPRZ_ERR_DEVICE, msg);
    }
    *out = c;
    return HIPRZ_OK;
}

int hiprz_destroy(hiprz_ctx* c) {
    if (!c) return HIPRZ_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    drop_graph(c);
    for (auto& p : c->pending_events) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    for (auto e : c->kernel_events) (void)hipEventDestroy(e);
    c->hot.release(), c->wnodes.release(), c->wskip.release(), c->node_skip.release(), c->nodes64.release(), c->textures.release();
    c->texels.release(), c->spot_lights.release(), c->direct_lights.release();
    release_frame(c);
    c->pass_dev.release(), c->counters_dev.release(), c->pick_dev.release(), c->rq_counts.release(), c->wg_times.release();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return HIPRZ_OK;
}

const char* hiprz_last_error(const hiprz_ctx* c) { return c ? c->error.c_str() : g_create_error.c_str(); }

int hiprz_upload_scene(hiprz_ctx* c, const hiprz_scene* sc) {
    if (!c) return HIPRZ_ERR_INVALID;
    c->graph_valid = false;
    StageTimer timer;
    SceneCheck chk;
    if (check_scene(sc, chk) != HIPRZ_OK) return fail(c, HIPRZ_ERR_INVALID, "upload_scene: " + chk.error);
    const uint32_t world_depth = chk.world_depth, mesh_depth = chk.mesh_depth;
    c->stack_entries = world_depth + mesh_depth + 2u;
    c->dscene.world_stack_entries = world_depth + 1u;
    c->dscene.mesh_stack_entries = mesh_depth + 1u;

    DerivedTables derived;
    if (derive_tables(sc, chk, derived) != HIPRZ_OK) return fail(c, HIPRZ_ERR_INVALID, "upload_scene: " + chk.error);
    std::vector<uint32_t>& new_index = derived.new_index;
    std::vector<hiprz_node>& dnodes = derived.dnodes;
    std::vector<hiprz_node>& wnodes = derived.wnodes;
    std::vector<uint32_t>& dskip = derived.dskip;
    std::vector<uint32_t>& wskip = derived.wskip;
    // shared-reciprocal division is exact only for coordinates that are 0 or in [2^-60, 2^40)
    auto coord_ok = [](float x) {
        uint32_t b;
        std::memcpy(&b, &x, 4);
        const uint32_t e = (b >> 23) & 0xFFu;
        return (b & 0x7FFFFFFFu) == 0u || (e >= 127u - 60u && e < 127u + 40u);
    };
    bool fast_div = true;
    for (const auto& n : wnodes)
        for (int a = 0; a < 3; ++a) fast_div = fast_div && coord_ok(n.bb_min[a]) && coord_ok(n.bb_max[a]);

    (void)hipSetDevice(c->device);
    // hot blob: one buffer, 16-B aligned sections
    std::vector<uint8_t> blob;
    auto append = [&blob](const void* src, size_t bytes) {
        const uint32_t off = uint32_t(blob.size());
        blob.resize(blob.size() + ((bytes + 15u) & ~size_t(15)), 0);
        if (bytes) std::memcpy(blob.data() + off, src, bytes);
        return off;
    };
    DScene& d = c->dscene;
    // device copies keep every box interleaved, (min.x, max.x, min.y, max.y, min.z, max.z), so one axis' two
    // plane distances are one packed operand of the box test (hiprz_device.hpp: box_hit)
    auto interleave = [](float* mn, float* mx) {
        const float v[6] = {mn[0], mx[0], mn[1], mx[1], mn[2], mx[2]};
        mn[0] = v[0], mn[1] = v[1], mn[2] = v[2], mx[0] = v[3], mx[1] = v[4], mx[2] = v[5];
    };
    for (auto& n : dnodes) interleave(n.bb_min, n.bb_max);  // bb_min[3] and bb_max[3] are contiguous
    for (auto& n : wnodes) interleave(n.bb_min, n.bb_max);
    std::vector<hiprz_instance> dinstances(sc->instances, sc->instances + sc->n_instances);
    for (auto& in : dinstances) {
        if (in.blas_root < sc->n_nodes) in.blas_root = new_index[in.blas_root];
        in.pad0 = (in.scale[0] == 1.0f && in.scale[1] == 1.0f && in.scale[2] == 1.0f) ? 1u : 0u;  // x / 1.0f == x: the walk skips it
        const float v[6] = {in.bb_min[0], in.bb_max[0], in.bb_min[1], in.bb_max[1], in.bb_min[2], in.bb_max[2]};
        in.bb_min[0] = v[0], in.bb_min[1] = v[1], in.bb_min[2] = v[2];
        std::memcpy(&in.pad2, &v[3], 4);
        in.bb_max[0] = v[4], in.bb_max[1] = v[5], in.bb_max[2] = 0.0f;
    }
    d.off_nodes = append(dnodes.data(), sizeof(hiprz_node) * dnodes.size());
    d.off_tlas_order = append(sc->tlas_order, sizeof(uint32_t) * sc->n_tlas_order);
    d.off_instances = append(dinstances.data(), sizeof(hiprz_instance) * dinstances.size());
    // device triangles hold v1 and the edges v2 - v1, v3 - v1; v2 and v3 themselves (normal mapping only) move into
    // the padding words of the attribute record
    std::vector<hiprz_tri> dtris(sc->tris, sc->tris + sc->n_tris);
    std::vector<hiprz_tri_attr> dattrs(sc->tri_attrs, sc->tri_attrs + sc->n_tris);
    for (uint32_t i = 0; i < sc->n_tris; ++i) {
        hiprz_tri& t = dtris[i];
        hiprz_tri_attr& a = dattrs[i];
        a.pad0 = t.v2[0], a.pad1 = t.v2[1], a.pad2 = t.v2[2], a.pad3 = t.v3[0], a.pad4[0] = t.v3[1], a.pad4[1] = t.v3[2];
        for (int k = 0; k < 3; ++k) {
            const float v2 = t.v2[k], v3 = t.v3[k];
            t.v2[k] = v2 - t.v1[k];
            t.v3[k] = v3 - t.v1[k];
        }
    }
    d.off_tris = append(dtris.data(), sizeof(hiprz_tri) * dtris.size());
    d.off_tri_attrs = append(dattrs.data(), sizeof(hiprz_tri_attr) * dattrs.size());
    d.off_materials = append(sc->materials, sizeof(hiprz_material) * sc->n_materials);
    d.off_inst_materials = append(sc->inst_materials, sizeof(int32_t) * sc->n_inst_materials);
    if (blob.size() > 0xFFFFFFF0ull) return fail(c, HIPRZ_ERR_INVALID, "scene geometry exceeds 4 GiB");
    d.hot_bytes = uint32_t(blob.size());

    RZ_HIP(c, c->hot.assign(blob.data(), blob.size(), c->stream));
    RZ_HIP(c, c->node_skip.assign(dskip.data(), dskip.size(), c->stream));
    // front-to-back walk: 64-B records, node (interleaved box) + the 8 octant links
    std::vector<uint32_t> nodes64(size_t(dnodes.size() ? dnodes.size() : 1) * 16u, RZ_END);
    for (size_t n = 0; n < dnodes.size(); ++n) {
        std::memcpy(&nodes64[n * 16u], &dnodes[n], sizeof(hiprz_node));
        std::memcpy(&nodes64[n * 16u + 8u], &derived.dskip8[n * 8u], 32);
    }
    RZ_HIP(c, c->nodes64.assign(nodes64.data(), nodes64.size(), c->stream));
    RZ_HIP(c, c->wnodes.assign(wnodes.data(), wnodes.size(), c->stream));
    RZ_HIP(c, c->wskip.assign(wskip.data(), wskip.size(), c->stream));
    RZ_HIP(c, c->textures.assign(sc->textures, sc->n_textures, c->stream));
    RZ_HIP(c, c->texels.assign(sc->texels, sc->texel_bytes, c->stream));
    RZ_HIP(c, c->spot_lights.assign(sc->spot_lights, sc->n_spot_lights, c->stream));
    RZ_HIP(c, c->direct_lights.assign(sc->direct_lights, sc->n_direct_lights, c->stream));
    RZ_HIP(c, hipStreamSynchronize(c->stream));  // host staging vectors and the caller's arrays may go away after return

    d.hot = reinterpret_cast<const float4*>(c->hot.ptr);
    d.nodes = reinterpret_cast<const float4*>(c->hot.ptr + d.off_nodes);
    d.tlas_order = reinterpret_cast<const uint32_t*>(c->hot.ptr + d.off_tlas_order);
    d.instances = reinterpret_cast<const float4*>(c->hot.ptr + d.off_instances);
    d.tris = reinterpret_cast<const float4*>(c->hot.ptr + d.off_tris);
    d.tri_attrs = reinterpret_cast<const float4*>(c->hot.ptr + d.off_tri_attrs);
    d.materials = reinterpret_cast<const float4*>(c->hot.ptr + d.off_materials);
    d.inst_materials = reinterpret_cast<const int32_t*>(c->hot.ptr + d.off_inst_materials);
    d.wnodes = reinterpret_cast<const float4*>(c->wnodes.ptr);
    d.wskip = c->wskip.ptr;
    d.fast_div = fast_div ? 1u : 0u;
    d.textures = reinterpret_cast<const float4*>(c->textures.ptr);
    d.texels = c->texels.ptr;
    d.spot_lights = reinterpret_cast<const float4*>(c->spot_lights.ptr);
    d.direct_lights = reinterpret_cast<const float4*>(c->direct_lights.ptr);
    d.n_instances = sc->n_instances;
    d.tlas_root = sc->n_instances ? new_index[sc->tlas_root] : 0u;
    d.node_skip = c->node_skip.ptr;
    for (int a = 0; a < 3; ++a) {
        const float lo = sc->n_instances ? sc->nodes[sc->tlas_root].bb_min[a] : 0.0f, hi = sc->n_instances ? sc->nodes[sc->tlas_root].bb_max[a] : 0.0f;
        d.bounds_min[a] = lo;
        d.bounds_scale[a] = hi > lo ? 32.0f / (hi - lo) : 0.0f;
    }
    d.top_count = std::min<uint32_t>(sc->n_nodes, kTopCacheNodes);
    d.nodes64 = reinterpret_cast<const float4*>(c->nodes64.ptr);
    c->n_nodes = sc->n_nodes;
    c->n_textures = sc->n_textures;
    // mesh walk rounds of at most 4 node steps and 8 triangles per lane (measured: D 3 163 -> 2 891 us, C 935 -> 892 us)
    d.walk_k = 4u, d.walk_l = 8u;
    // ray reordering key: without lights only the closest-hit walk follows the sorted order and the interleaved origin/direction
    // code groups best (config C: trace kernel 583 -> 503 us); with lights the deferred shadow rays follow it too and they fan out
    // from the origin cell, so the origin leads (config E: 86.5 ms per step against 92.6)
    d.sort_variant = (sc->n_spot_lights + sc->n_direct_lights) && c->shadow_sort == 0 ? 0u : 2u;
    d.shadow_variant = 0u;
    if (const char* v = std::getenv("HIPRZ_SHADOW_KEY")) d.shadow_variant = uint32_t(std::atoi(v));
    if (const char* v = std::getenv("HIPRZ_SORT_KEY")) d.sort_variant = uint32_t(std::atoi(v));
    if (const char* v = std::getenv("HIPRZ_WALK_K")) d.walk_k = uint32_t(std::atoi(v));
    if (const char* v = std::getenv("HIPRZ_WALK_L")) d.walk_l = uint32_t(std::atoi(v));
    d.wtop_count = std::min<uint32_t>(uint32_t(wnodes.size()), kTopCacheNodes);
    d.n_spot_lights = sc->n_spot_lights;
    d.n_direct_lights = sc->n_direct_lights;
    // Stage the blob in LDS when three workgroups per CU (the kernel's register-limited residency)
    // still fit into the CU's 160 KiB together with their traversal stacks.
    c->lds_scene = size_t(d.hot_bytes) + size_t(c->stack_entries) * 1024u + BinnedLds::kFixedBytes <= kLdsSceneLimit;
    c->have_scene = true;
    resolve_pipeline(c);
    c->reset_pending = true;  // world changed => accumulation restarts (cpu_engine_renderer.cpp:108-112)
    c->timings.set("upload scene", timer.ms());
    return HIPRZ_OK;
}

int hiprz_upload_camera(hiprz_ctx* c, const hiprz_camera* cam) {
    if (!c) return HIPRZ_ERR_INVALID;
    c->graph_valid = false;
    if (!cam) return fail(c, HIPRZ_ERR_INVALID, "upload_camera: camera is null");
    if (cam->width == 0 || cam->height == 0 || cam->width > 32768u || cam->height > 32768u)
        return fail(c, HIPRZ_ERR_INVALID, "upload_camera: resolution must be 1..32768");
    StageTimer timer;
    (void)hipSetDevice(c->device);
    const bool resized = !c->have_camera || cam->width != c->camera.width || cam->height != c->camera.height;
    c->camera = *cam;
    DCamera& d = c->dcamera;
    std::memcpy(d.position, cam->position, 12);
    std::memcpy(d.x_axis, cam->x_axis, 12);
    std::memcpy(d.y_axis, cam->y_axis, 12);
    std::memcpy(d.z_axis, cam->z_axis, 12);
    d.width = cam->width, d.height = cam->height;
    d.tan_half_fov = cam->tan_half_fov, d.aspect_ratio = cam->aspect_ratio;
    d.near_ = cam->near_far[0], d.far_ = cam->near_far[1];
    d.focal_distance = cam->focal_distance, d.aperture = cam->aperture, d.exposure_time = cam->exposure_time;
    c->have_camera = true;
    if (resized) {
        const int rc = allocate_frame(c);
        if (rc != HIPRZ_OK) return rc;
    }
    c->reset_pending = true;  // camera changed => context.reset (cpu_engine_renderer.cpp:108-112)
    c->timings.set("upload camera", timer.ms());
    return HIPRZ_OK;
}

int hiprz_set_config(hiprz_ctx* c, const hiprz_config* cfg) {
    if (!c) return HIPRZ_ERR_INVALID;
    c->graph_valid = false;
    if (!cfg) return fail(c, HIPRZ_ERR_INVALID, "set_config: config is null");
    if (cfg->max_depth == 0 || cfg->max_depth > 254u) return fail(c, HIPRZ_ERR_INVALID, "max_depth must be 1..254 (u8, 255 = path ended)");
    // The CPU kernel divides by sample_count/light_count and yields NaN for 0 samples
    // (cpu_engine_kernel.cpp:742-743, 789-790); the CUDA backend clamps to >= 1 (cuda_kernel_data.cu:23-31).
    if (cfg->spot_samples == 0 || cfg->direct_samples == 0 || cfg->spot_samples > 255u || cfg->direct_samples > 255u)
        return fail(c, HIPRZ_ERR_INVALID, "light sample counts must be 1..255");
    c->config = *cfg;
    return HIPRZ_OK;
}

int hiprz_set_shard(hiprz_ctx* c, uint32_t rank, uint32_t world) {
    if (!c) return HIPRZ_ERR_INVALID;
    c->graph_valid = false;
    if (world == 0 || rank >= world) return fail(c, HIPRZ_ERR_INVALID, "set_shard: need rank < world");
    const bool changed = rank != c->rank || world != c->world;
    c->rank = rank, c->world = world;
    if (changed && c->have_camera) {
        (void)hipSetDevice(c->device);
        const int rc = allocate_frame(c);
        if (rc != HIPRZ_OK) return rc;
        c->reset_pending = true;
    }
    return HIPRZ_OK;
}

int hiprz_set_traversal_mode(hiprz_ctx* c, int mode) {
    if (!c) return HIPRZ_ERR_INVALID;
    c->graph_valid = false;
    if (mode < -1 || mode > 6) return fail(c, HIPRZ_ERR_INVALID, "traversal mode: -1 = auto, 0 = threaded, 1 = LDS stack, 2 = workgroup-binned, 3 = skip links + LDS-cached tree tops, 4 = persistent lanes on the flat walk graph, 5 = mode 3 in rounds with ray requeueing, 6 = wave pool (persistent waves, phased world / mesh walk)");
    c->traversal_mode = mode;
    resolve_pipeline(c);
    return HIPRZ_OK;
}

int hiprz_set_walk_order(hiprz_ctx* c, int order) {
    if (!c) return HIPRZ_ERR_INVALID;
    c->graph_valid = false;
    if (order < 0 || order > 2) return fail(c, HIPRZ_ERR_INVALID, "walk order: 0 = the reference's child order, 1 = front-to-back, 2 = front-to-back also in counted renders");
    c->walk_order = order;
    return HIPRZ_OK;
}

int hiprz_set_requeue_schedule(hiprz_ctx* c, const uint32_t* thresholds, uint32_t n_rounds) {
    if (!c) return HIPRZ_ERR_INVALID;
    c->graph_valid = false;
    if (n_rounds > 30u || (n_rounds && !thresholds)) return fail(c, HIPRZ_ERR_INVALID, "requeue schedule: at most 30 bailing rounds");
    for (uint32_t r = 0; r < n_rounds; ++r)
        if (thresholds[r] > 64u) return fail(c, HIPRZ_ERR_INVALID, "requeue schedule: a threshold is a lane count (0..64)");
    c->requeue_thresholds.assign(thresholds, thresholds + n_rounds);
    return HIPRZ_OK;
}

int hiprz_set_workgroup_timing(hiprz_ctx* c, int enabled) {
    if (!c) return HIPRZ_ERR_INVALID;
    c->graph_valid = false;
    c->wg_timing = enabled != 0;
    if (c->wg_timing) RZ_HIP(c, c->wg_times.resize(size_t(c->n_local_tiles + 8u) * 2u));
    return HIPRZ_OK;
}

int hiprz_read_workgroup_times(hiprz_ctx* c, uint64_t* start_end_out, uint32_t n_workgroups) {
    if (!c || !start_end_out) return HIPRZ_ERR_INVALID;
    if (!c->wg_timing || !c->wg_times.ptr || n_workgroups > c->n_local_tiles) return fail(c, HIPRZ_ERR_STATE, "workgroup timing is not enabled for this frame");
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    RZ_HIP(c, hipMemcpy(start_end_out, c->wg_times.ptr, sizeof(uint64_t) * 2u * n_workgroups, hipMemcpyDeviceToHost));
    return HIPRZ_OK;
}

int hiprz_requeue_counts(hiprz_ctx* c, uint32_t* counts_out, uint32_t n) {
    if (!c || !counts_out) return HIPRZ_ERR_INVALID;
    for (uint32_t i = 0; i < n; ++i) counts_out[i] = 0u;
    if (!c->rq_counts.ptr) return HIPRZ_OK;
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    RZ_HIP(c, hipMemcpy(counts_out, c->rq_counts.ptr, sizeof(uint32_t) * std::min<uint32_t>(n, 32u), hipMemcpyDeviceToHost));
    return HIPRZ_OK;
}

int hiprz_set_lds_scene(hiprz_ctx* c, int mode) {
    if (!c) return HIPRZ_ERR_INVALID;
    c->graph_valid = false;
    if (mode < -1 || mode > 1) return fail(c, HIPRZ_ERR_INVALID, "lds scene: -1 auto, 0 off, 1 on");
    c->lds_scene_override = mode;
    resolve_pipeline(c);
    return HIPRZ_OK;
}

int hiprz_set_pipeline(hiprz_ctx* c, int pipeline) {
    if (!c) return HIPRZ_ERR_INVALID;
    c->graph_valid = false;
    if (pipeline < -1 || pipeline > 2) return fail(c, HIPRZ_ERR_INVALID, "pipeline: -1 = per scene, 0 = fused pass kernel, 1 = trace kernel + shade kernel, 2 = resident (one launch per batch of passes)");
    c->pipeline_setting = pipeline;
    resolve_pipeline(c);
    return HIPRZ_OK;
}

int hiprz_pipeline(hiprz_ctx* c, int* out) {
    if (!c || !out) return HIPRZ_ERR_INVALID;
    *out = c->pipeline;
    return HIPRZ_OK;
}

int hiprz_traversal_mode(hiprz_ctx* c, int* out) {
    if (!c || !out) return HIPRZ_ERR_INVALID;
    *out = effective_mode(c);
    return HIPRZ_OK;
}

int hiprz_set_ray_sort(hiprz_ctx* c, int mode) {
    if (!c) return HIPRZ_ERR_INVALID;
    c->graph_valid = false;
    if (mode < -1 || mode > 1) return fail(c, HIPRZ_ERR_INVALID, "ray sort: -1 auto, 0 off, 1 on");
    c->sort_rays = mode;
    return HIPRZ_OK;
}

int hiprz_set_xcd_swizzle(hiprz_ctx* c, int enabled) {
    if (!c) return HIPRZ_ERR_INVALID;
    c->graph_valid = false;
    c->xcd_swizzle = enabled != 0;
    return HIPRZ_OK;
}

int hiprz_set_graph(hiprz_ctx* c, int enabled) {
    if (!c) return HIPRZ_ERR_INVALID;
    c->use_graph = enabled != 0;
    return HIPRZ_OK;
}

int hiprz_reset(hiprz_ctx* c) {
    if (!c) return HIPRZ_ERR_INVALID;
    c->reset_pending = true;
    return HIPRZ_OK;
}

int hiprz_render(hiprz_ctx* c, uint32_t n_passes) {
    if (!c) return HIPRZ_ERR_INVALID;
    (void)hipSetDevice(c->device);
    return render_impl(c, n_passes, false);
}

int hiprz_render_counted(hiprz_ctx* c, uint32_t n_passes, hiprz_counters* out) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (!out) return fail(c, HIPRZ_ERR_INVALID, "render_counted: out is null");
    (void)hipSetDevice(c->device);
    RZ_HIP(c, hipMemsetAsync(c->counters_dev.ptr, 0, 16 * sizeof(unsigned long long), c->stream));
    const int rc = render_impl(c, n_passes, true);
    if (rc != HIPRZ_OK) return rc;
    unsigned long long v[10];
    RZ_HIP(c, hipMemcpyAsync(v, c->counters_dev.ptr, sizeof v, hipMemcpyDeviceToHost, c->stream));
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    out->segments = v[0], out->box_tests = v[1], out->tri_tests = v[2], out->hits = v[3];
    out->shadow_rays = v[4], out->light_samples = v[5], out->texel_fetches = v[6], out->finished = v[7];
    out->shadow_box_tests = v[8], out->shadow_tri_tests = v[9];
    return HIPRZ_OK;
}

int hiprz_tonemap(hiprz_ctx* c) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (!c->have_camera) return fail(c, HIPRZ_ERR_STATE, "tonemap before camera upload");
    (void)hipSetDevice(c->device);
    if (c->rgba8_valid && !c->reset_pending) return HIPRZ_OK;  // the resident kernel already wrote this frame's pixels
    const uint32_t n = c->n_local_tiles * 256u;
    if (n)
        hipLaunchKernelGGL(rz_tonemap_tiles_kernel, dim3(c->n_local_tiles), dim3(256), 0, c->stream, c->accum.ptr, c->rgba8.ptr, n,
                           c->camera.aperture, c->camera.exposure_time);
    RZ_HIP(c, hipGetLastError());
    return HIPRZ_OK;
}

int hiprz_sync(hiprz_ctx* c) {
    if (!c) return HIPRZ_ERR_INVALID;
    (void)hipSetDevice(c->device);
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    return HIPRZ_OK;
}

int hiprz_read_rgba8(hiprz_ctx* c, uint8_t* dst, size_t bytes) {
    if (!c) return HIPRZ_ERR_INVALID;
    (void)hipSetDevice(c->device);
    return read_untiled<uint32_t>(c, c->rgba8.ptr, reinterpret_cast<uint32_t*>(dst), bytes, "read rgba8");
}
int hiprz_read_depth(hiprz_ctx* c, float* dst, size_t bytes) {
    if (!c) return HIPRZ_ERR_INVALID;
    (void)hipSetDevice(c->device);
    return read_untiled<float>(c, c->depth.ptr, dst, bytes, "read depth");
}
int hiprz_read_accum(hiprz_ctx* c, float* dst, size_t bytes) {
    if (!c) return HIPRZ_ERR_INVALID;
    (void)hipSetDevice(c->device);
    return read_untiled<float4>(c, c->accum.ptr, reinterpret_cast<float4*>(dst), bytes, "read accum");
}

int hiprz_read_state(hiprz_ctx* c, float* ray9, uint32_t* md2, size_t n_pixels) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (!c->have_camera) return fail(c, HIPRZ_ERR_STATE, "readback before camera upload");
    const size_t n = size_t(c->camera.width) * c->camera.height;
    if (!ray9 || !md2 || n_pixels != n) return fail(c, HIPRZ_ERR_INVALID, "read_state: destination size mismatch");
    (void)hipSetDevice(c->device);
    RZ_HIP(c, c->state_ray.resize(9 * n));
    RZ_HIP(c, c->state_md.resize(2 * n));
    RZ_HIP(c, hipMemsetAsync(c->state_ray.ptr, 0, 9 * n * sizeof(float), c->stream));
    RZ_HIP(c, hipMemsetAsync(c->state_md.ptr, 0, 2 * n * sizeof(uint32_t), c->stream));
    if (c->n_local_tiles)
        hipLaunchKernelGGL(rz_untile_state_kernel, dim3(c->n_local_tiles), dim3(256), 0, c->stream, c->st0.ptr, c->st1.ptr,
                           c->st2.ptr, c->state_ray.ptr, c->state_md.ptr, c->camera.width, c->camera.height, c->tiles_x, c->rank,
                           c->world);
    RZ_HIP(c, hipMemcpyAsync(ray9, c->state_ray.ptr, 9 * n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    RZ_HIP(c, hipMemcpyAsync(md2, c->state_md.ptr, 2 * n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    return HIPRZ_OK;
}

int hiprz_ray_count(hiprz_ctx* c, uint64_t* out) {
    if (!c || !out) return HIPRZ_ERR_INVALID;
    *out = c->ray_count;
    return HIPRZ_OK;
}
int hiprz_pass_count(hiprz_ctx* c, uint32_t* out) {
    if (!c || !out) return HIPRZ_ERR_INVALID;
    *out = c->passes;
    return HIPRZ_OK;
}

int hiprz_local_pixel_capacity(hiprz_ctx* c, size_t* out) {
    if (!c || !out) return HIPRZ_ERR_INVALID;
    *out = size_t(c->n_local_tiles) * 256u;
    return HIPRZ_OK;
}
int hiprz_export_accum_tiles(hiprz_ctx* c, void* dst_device, size_t bytes) {
    if (!c) return HIPRZ_ERR_INVALID;
    const size_t need = size_t(c->n_local_tiles) * 256u * sizeof(float4);
    if (!dst_device || bytes < need) return fail(c, HIPRZ_ERR_INVALID, "export_accum_tiles: destination too small");
    (void)hipSetDevice(c->device);
    if (need) RZ_HIP(c, hipMemcpyAsync(dst_device, c->accum.ptr, need, hipMemcpyDeviceToDevice, c->stream));
    return HIPRZ_OK;
}
int hiprz_export_rgba8_tiles(hiprz_ctx* c, void* dst_device, size_t bytes) {
    if (!c) return HIPRZ_ERR_INVALID;
    const size_t need = size_t(c->n_local_tiles) * 256u * sizeof(uint32_t);
    if (!dst_device || bytes < need) return fail(c, HIPRZ_ERR_INVALID, "export_rgba8_tiles: destination too small");
    (void)hipSetDevice(c->device);
    if (need) RZ_HIP(c, hipMemcpyAsync(dst_device, c->rgba8.ptr, need, hipMemcpyDeviceToDevice, c->stream));
    return HIPRZ_OK;
}
int hiprz_untile_rgba8(hiprz_ctx* c, const void* src_tiles, uint32_t rank, uint32_t world, void* dst_image) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (!c->have_camera) return fail(c, HIPRZ_ERR_STATE, "untile before camera upload");
    if (!src_tiles || !dst_image || world == 0 || rank >= world) return fail(c, HIPRZ_ERR_INVALID, "untile_rgba8: bad arguments");
    (void)hipSetDevice(c->device);
    const uint32_t n_tiles = c->tiles_x * c->tiles_y;
    const uint32_t n_local = rank < n_tiles ? (n_tiles - rank + world - 1u) / world : 0u;
    if (n_local)
        hipLaunchKernelGGL((rz_untile_kernel<uint32_t>), dim3(n_local), dim3(256), 0, c->stream,
                           reinterpret_cast<const uint32_t*>(src_tiles), reinterpret_cast<uint32_t*>(dst_image), c->camera.width,
                           c->camera.height, c->tiles_x, rank, world);
    RZ_HIP(c, hipGetLastError());
    return HIPRZ_OK;
}
int hiprz_untile_gathered(hiprz_ctx* c, const void* src_parts, uint32_t world, size_t part_stride_bytes, uint32_t element_bytes,
                          void* dst_image, void* stream) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (!c->have_camera) return fail(c, HIPRZ_ERR_STATE, "untile before camera upload");
    if (!src_parts || !dst_image || world == 0 || world > 65535u || (element_bytes != 4u && element_bytes != 16u) || part_stride_bytes % element_bytes)
        return fail(c, HIPRZ_ERR_INVALID, "untile_gathered: bad arguments");
    (void)hipSetDevice(c->device);
    const uint32_t n_tiles = c->tiles_x * c->tiles_y;
    const uint32_t per_rank = (n_tiles + world - 1u) / world;
    if (part_stride_bytes < size_t(per_rank) * 256u * element_bytes) return fail(c, HIPRZ_ERR_INVALID, "untile_gathered: part stride smaller than a shard");
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : c->stream;
    if (n_tiles) {
        const dim3 grid(per_rank, world);
        if (element_bytes == 4u)
            hipLaunchKernelGGL((rz_untile_gathered_kernel<uint32_t>), grid, dim3(256), 0, st, reinterpret_cast<const uint32_t*>(src_parts),
                               part_stride_bytes / 4u, reinterpret_cast<uint32_t*>(dst_image), c->camera.width, c->camera.height, c->tiles_x, n_tiles, world);
        else
            hipLaunchKernelGGL((rz_untile_gathered_kernel<float4>), grid, dim3(256), 0, st, reinterpret_cast<const float4*>(src_parts),
                               part_stride_bytes / 16u, reinterpret_cast<float4*>(dst_image), c->camera.width, c->camera.height, c->tiles_x, n_tiles, world);
    }
    RZ_HIP(c, hipGetLastError());
    return HIPRZ_OK;
}
int hiprz_untile_accum(hiprz_ctx* c, const void* src_tiles, uint32_t rank, uint32_t world, void* dst_image) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (!c->have_camera) return fail(c, HIPRZ_ERR_STATE, "untile before camera upload");
    if (!src_tiles || !dst_image || world == 0 || rank >= world) return fail(c, HIPRZ_ERR_INVALID, "untile_accum: bad arguments");
    (void)hipSetDevice(c->device);
    const uint32_t n_tiles = c->tiles_x * c->tiles_y;
    const uint32_t n_local = rank < n_tiles ? (n_tiles - rank + world - 1u) / world : 0u;
    if (n_local)
        hipLaunchKernelGGL((rz_untile_kernel<float4>), dim3(n_local), dim3(256), 0, c->stream,
                           reinterpret_cast<const float4*>(src_tiles), reinterpret_cast<float4*>(dst_image), c->camera.width,
                           c->camera.height, c->tiles_x, rank, world);
    RZ_HIP(c, hipGetLastError());
    return HIPRZ_OK;
}
int hiprz_tonemap_image(hiprz_ctx* c, const void* src_image, void* dst_rgba8) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (!c->have_camera) return fail(c, HIPRZ_ERR_STATE, "tonemap before camera upload");
    if (!src_image || !dst_rgba8) return fail(c, HIPRZ_ERR_INVALID, "tonemap_image: null pointer");
    (void)hipSetDevice(c->device);
    const uint32_t n = c->camera.width * c->camera.height;
    hipLaunchKernelGGL(rz_tonemap_image_kernel, dim3((n + 255u) / 256u), dim3(256), 0, c->stream,
                       reinterpret_cast<const float4*>(src_image), reinterpret_cast<uint32_t*>(dst_rgba8), n, c->camera.aperture,
                       c->camera.exposure_time);
    RZ_HIP(c, hipGetLastError());
    return HIPRZ_OK;
}
void* hiprz_stream(hiprz_ctx* c) { return c ? reinterpret_cast<void*>(c->stream) : nullptr; }

int hiprz_pick(hiprz_ctx* c, uint32_t x, uint32_t y, int32_t* instance_out, int32_t* material_out) {
    if (!c) return HIPRZ_ERR_INVALID;
    if (!c->have_scene || !c->have_camera) return fail(c, HIPRZ_ERR_STATE, "pick before scene and camera upload");
    if (!instance_out || !material_out) return fail(c, HIPRZ_ERR_INVALID, "pick: null output");
    // Camera::rayCastPixel clamps (camera.cpp:161-167)
    if (x >= c->camera.width) x = c->camera.width - 1;
    if (y >= c->camera.height) y = c->camera.height - 1;
    (void)hipSetDevice(c->device);
    // depth of the pixel: only the shard that owns it can answer
    const uint32_t tile = (y / 8u) * c->tiles_x + (x / 32u);
    *instance_out = *material_out = -1;
    if (tile % c->world != c->rank) return HIPRZ_OK;
    const uint32_t lt = tile / c->world, in_tile = ((x % 32u) / 8u) * 64u + (y % 8u) * 8u + (x % 8u);
    float depth = 0.0f;
    RZ_HIP(c, hipMemcpyAsync(&depth, c->depth.ptr + size_t(lt) * 256u + in_tile, sizeof(float), hipMemcpyDeviceToHost, c->stream));
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    hipLaunchKernelGGL(rz_pick_kernel, dim3(1), dim3(1), 0, c->stream, c->dscene, c->dcamera, x, y, depth, c->pick_dev.ptr);
    int32_t out2[2] = {-1, -1};
    RZ_HIP(c, hipMemcpyAsync(out2, c->pick_dev.ptr, sizeof out2, hipMemcpyDeviceToHost, c->stream));
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    *instance_out = out2[0], *material_out = out2[1];
    return HIPRZ_OK;
}

int hiprz_selftest(hiprz_ctx* c, uint32_t cases_per_thread, uint32_t seed, uint64_t* mismatches, uint64_t* tested) {
    if (!c || !mismatches || !tested) return HIPRZ_ERR_INVALID;
    (void)hipSetDevice(c->device);
    RZ_HIP(c, hipMemsetAsync(c->counters_dev.ptr, 0, 8 * sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(rz_selftest_div_kernel, dim3(1024), dim3(256), 0, c->stream, cases_per_thread, seed, c->counters_dev.ptr);
    unsigned long long v[2];
    RZ_HIP(c, hipMemcpyAsync(v, c->counters_dev.ptr, sizeof v, hipMemcpyDeviceToHost, c->stream));
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    *mismatches = v[0], *tested = v[1];
    return HIPRZ_OK;
}

#ifdef RZ_BOXPATH_STATS
int hiprz_read_boxpath(unsigned long long out[4]) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(hiprz::rz_boxpath), 32);
    unsigned long long zero[4] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(hiprz::rz_boxpath), zero, 32);
    return 0;
}
#endif
#ifdef RZ_PHASE_STATS
extern "C" int hiprz_read_phase_stats(unsigned long long out[16]) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(hiprz::rz_phase), 128);
    unsigned long long zero[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(hiprz::rz_phase), zero, 128);
    return 0;
}
#endif
#ifdef RZ_STAMP
int hiprz_read_stamps(unsigned long long out[8]) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(hiprz::rz_stamp_sums), 64);
    unsigned long long zero[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(hiprz::rz_stamp_sums), zero, 64);
    return 0;
}
#endif

int hiprz_timings(hiprz_ctx* c, char* buf, size_t len) {
    if (!c || !buf || len == 0) return HIPRZ_ERR_INVALID;
    const std::string s = c->timings.str();
    std::snprintf(buf, len, "%s", s.c_str());
    return HIPRZ_OK;
}

int hiprz_time_kernels(hiprz_ctx* c, int enabled) {
    if (!c) return HIPRZ_ERR_INVALID;
    c->graph_valid = false;
    c->time_kernels = enabled != 0;
    return HIPRZ_OK;
}

int hiprz_kernel_breakdown_ms(hiprz_ctx* c, double* trace_ms, double* shade_ms, uint32_t* passes) {
    if (!c || !trace_ms || !shade_ms || !passes) return HIPRZ_ERR_INVALID;
    (void)hipSetDevice(c->device);
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    *trace_ms = *shade_ms = 0.0;
    *passes = 0;
    if (resident_active(c)) {  // one launch for all the passes of the batch: reported as "trace"
        if (c->kernel_event_passes) {
            float a = 0;
            RZ_HIP(c, hipEventElapsedTime(&a, c->kernel_events[0], c->kernel_events[1]));
            *trace_ms = a;
            *passes = c->kernel_event_passes;
        }
        return HIPRZ_OK;
    }
    if (c->pipeline != 1) return HIPRZ_OK;
    for (uint32_t i = 0; i < c->kernel_event_passes; ++i) {
        float a = 0, b = 0;
        RZ_HIP(c, hipEventElapsedTime(&a, c->kernel_events[3 * i], c->kernel_events[3 * i + 1]));
        RZ_HIP(c, hipEventElapsedTime(&b, c->kernel_events[3 * i + 1], c->kernel_events[3 * i + 2]));
        *trace_ms += a, *shade_ms += b;
    }
    *passes = c->kernel_event_passes;
    return HIPRZ_OK;
}

int hiprz_kernel_time_ms(hiprz_ctx* c, double* total_ms, uint64_t* launches) {
    if (!c || !total_ms || !launches) return HIPRZ_ERR_INVALID;
    (void)hipSetDevice(c->device);
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    double total = 0;
    uint64_t n = 0;
    for (size_t i = 0; i < c->pending_events.size(); ++i) {
        float ms = 0;
        RZ_HIP(c, hipEventElapsedTime(&ms, c->pending_events[i].first, c->pending_events[i].second));
        total += ms;
        n += c->pending_launches[i];
        c->event_pool.push_back(c->pending_events[i].first);
        c->event_pool.push_back(c->pending_events[i].second);
    }
    c->pending_events.clear();
    c->pending_launches.clear();
    *total_ms = total, *launches = n;
    if (n) c->timings.set("pass kernel (device)", total / double(n));
    return HIPRZ_OK;
}

}  // extern "C"
