// hiprz_device.hpp — gfx950 device code of the path-tracing pass.
//
// One thread = one pixel = one path segment per pass, as in the reference
// (RayZath/cpu_engine_kernel.cpp:15-101; RayZath/cuda_render_kernel.cu:7-121), but on a
// flattened SoA scene (include/hiprz.h) and with stack-free walks of the two trees on skip links
// (or, for scenes staged in LDS, an LDS stack / a workgroup-binned walk).  All arithmetic is fp32 and is spelled
// operation by operation in the order of the CPU reference, compiled with
// -ffp-contract=off, so that everything except libm-vs-ocml transcendentals is bit-equal
// to the CPU result.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hiprz.h"
#include "hiprz_shard.hpp"

namespace hiprz {

#define RZ_DEV __device__ __forceinline__
// experiment knobs (tools/ab_variants.sh builds one library per setting)
#ifndef RZ_FUSED_SHARED_RCP   // packed shared-reciprocal box test in the fused pass kernel's closest-hit walk
#define RZ_FUSED_SHARED_RCP 0
#endif
#ifndef RZ_TRACE_SHARED_RCP   // ... in the split pipeline's trace kernel (lean enough not to spill with it)
#define RZ_TRACE_SHARED_RCP 1
#endif
#ifndef RZ_BATCH_SHARED_RCP   // ... in the resident pipeline's batch kernel
#define RZ_BATCH_SHARED_RCP 1
#endif
#ifndef RZ_MIN_WAVES
#define RZ_MIN_WAVES 4
#endif
#ifndef RZ_TRACE_MIN_WAVES
#define RZ_TRACE_MIN_WAVES 5
#endif
#define RZ_PI_F 3.14159265358979323846f
#define RZ_END 0xFFFFFFFFu
// Termination of every walk is proven on the host before anything is launched (hiprz_api.hip: check_scene walks the
// uploaded trees, derive_tables walks the derived links), so the loops need no step budget.  -DRZ_WALK_GUARD=1 adds
// one anyway (debug builds: a wrong table then gives wrong pixels instead of a hung wave); it costs 9 % on config B.
#define RZ_GUARD_LIMIT (1u << 26)
#ifndef RZ_WALK_GUARD
#define RZ_WALK_GUARD 0
#endif
#if RZ_WALK_GUARD
#define RZ_GUARD(counter) \
    if (++(counter) > RZ_GUARD_LIMIT) break
#else
#define RZ_GUARD(counter) (void)(counter)
#endif
#define RZ_FLT_MAX 3.402823466e+38f

#define RZ_SINF(x) sinf(x)
#define RZ_COSF(x) cosf(x)
// sine and cosine of one angle with ONE argument reduction; ocml's sincosf returns bit-for-bit what sinf and cosf
// return separately (checked on the device by hiprz_selftest)
#define RZ_SINCOSF(x, s, c) sincosf((x), &(s), &(c))
#define RZ_ACOSF(x) acosf(x)
#define RZ_ASINF(x) asinf(x)
#define RZ_ATAN2F(y, x) atan2f(y, x)
#define RZ_POWF(x, y) powf(x, y)
#define RZ_EXPF(x) expf(x)

// ---------------------------------------------------------------------------------------
// Device-side views.  Every record array is addressed as float4 so a record is fetched
// with 16-byte loads (node = 2, triangle = 3, triangle attributes = 6, instance = 7,
// material = 3, texture = 3, spot light = 3, direct light = 2 float4).
// ---------------------------------------------------------------------------------------
struct DScene {
    const float4* nodes;         // hiprz_node records, breadth-first over all trees, boxes interleaved (min.x, max.x, ...)
    const uint32_t* tlas_order;
    const float4* tris;
    const float4* tri_attrs;
    const float4* instances;
    const int32_t* inst_materials;
    const float4* materials;
    const float4* textures;
    const uint8_t* texels;
    const float4* spot_lights;
    const float4* direct_lights;
    uint32_t n_instances;
    uint32_t tlas_root;
    uint32_t n_spot_lights;
    uint32_t n_direct_lights;
    uint32_t fast_div;  // every node / instance box coordinate is 0 or in [2^-60, 2^40): shared-reciprocal division is exact
    // The geometry + shading records live in ONE device buffer ("hot blob": nodes | tlas_order |
    // instances | tris | tri_attrs | materials | inst_materials, each section 16-B aligned) so a
    // workgroup can stage it into LDS with one strided copy when it is small enough.
    uint32_t hot_bytes;
    const float4* hot;  // start of the blob
    uint32_t off_nodes, off_tlas_order, off_instances, off_tris, off_tri_attrs, off_materials, off_inst_materials;
    uint32_t world_stack_entries;  // LDS stack entries per lane the world tree needs / the deepest mesh tree needs
    uint32_t mesh_stack_entries;
    float bounds_min[3];           // world box (root of the world tree) and 32 / extent per axis: cells of the ray sort key
    float bounds_scale[3];
    uint32_t top_count;            // MODE 3: the first top_count nodes (+ their links) are staged in LDS by every workgroup
    const uint32_t* node_skip;     // link to the node that follows a node's subtree (same numbering as `nodes`)
    uint32_t walk_k, walk_l;       // MODE 3 mesh walk: node steps / triangle tests per lane per round (0 = unbounded)
    uint32_t walk_advance;         // cooperative walks: further instance boxes a lane may test in one round while it has found none to enter
    uint32_t world_advance;        // ... and further world-tree nodes a lane may step through while it holds no leaf
    uint32_t walk_h;               // cooperative walks: the node phase of a round ends as soon as this many lanes hold a leaf
    // front-to-back mesh walk (hiprz_set_walk_order): 64-B records = node (32 B) + its skip link under each of the 8
    // ray-direction octants
    const float4* nodes64;
    uint32_t shadow_variant;  // the same for the shadow rays' key (HIPRZ_SHADOW_KEY); + 0x100: the pixel's set of sample slots leads the key; + 0x400: the light the ray goes to, then the origin's cell in a 64^3 grid, instead of a layout
    // the shadow rays' own world tree (hiprz_api.hip: build_shadow_world_tree): 64-byte walk records, the instance ids its leaves index, its root
    // (record 0) or RZ_END: none — the walks then take the reference's world tree
    const float4* shadow_nodes64;
    const uint32_t* shadow_order;
    uint32_t shadow_root;
    uint32_t sort_variant;  // ray_sort_key layout (HIPRZ_SORT_KEY): 0 origin cell then direction, 1 direction then cell, 2 interleaved, 3 octahedral direction interleaved with the cell, 4 origin cell interleaved with the cell where the ray leaves the world box
};

// Re-point the blob sections at a staged copy (LDS).
RZ_DEV void repoint_hot(DScene& v, const unsigned char* base) {
    v.nodes = reinterpret_cast<const float4*>(base + v.off_nodes);
    v.tlas_order = reinterpret_cast<const uint32_t*>(base + v.off_tlas_order);
    v.instances = reinterpret_cast<const float4*>(base + v.off_instances);
    v.tris = reinterpret_cast<const float4*>(base + v.off_tris);
    v.tri_attrs = reinterpret_cast<const float4*>(base + v.off_tri_attrs);
    v.materials = reinterpret_cast<const float4*>(base + v.off_materials);
    v.inst_materials = reinterpret_cast<const int32_t*>(base + v.off_inst_materials);
}

struct DCamera {
    float position[3];
    float x_axis[3], y_axis[3], z_axis[3];
    uint32_t width, height;
    float tan_half_fov, aspect_ratio;
    float near_, far_;
    float focal_distance, aperture, exposure_time;
};

struct DConfig {
    uint32_t max_depth, spot_samples, direct_samples, seed;
    uint32_t flags;  // HIPRZ_COMPAT_* (hiprz_set_mode); 0 = the CPU kernel's behaviour.  Only the compat instantiations read it.
};

// Per-pixel persistent state (CameraContext, cpu_engine_kernel.hpp:29-51), tile-major:
// local pixel i = owned_tile * 256 + thread.  40 B of path state + 16 B accumulator.
struct DFrame {
    float4* st0;      // origin.xyz, direction.x
    float4* st1;      // direction.yz, color.rg
    float2* st2;      // color.b, bits(material | depth << 16)
    float4* accum;    // RGBA32F, alpha = finished paths
    float4* hit0;     // split pipeline: far, b1, b2, bits(triangle)
    uint32_t* hit1;   // split pipeline: instance | found << 29 | external << 31
    float* depth;     // first-hit distance (first pass)
    uint32_t* rgba8;  // tone-mapped output
    const uint32_t* pass;  // device-resident pass index
    unsigned long long* counters;  // 8 x u64 (hiprz_counters) or nullptr
    uint32_t tiles_x;      // 32x8-pixel tiles per row
    uint32_t rank, world;  // tile sharding
    uint32_t n_local_tiles;
    uint32_t* sort_key;    // shade kernel: sort key of the pixel's NEXT ray (nullptr = ray reordering off)
    const uint32_t* perm;  // trace kernel: thread i walks the ray of local pixel perm[i] (nullptr = pixel order)
    uint32_t xcd_swizzle;  // 1: workgroup b works on owned tile (b % 8) * ceil(n/8) + b / 8 (see pixel_of_thread)
    // deferred shadow rays (rz_shade_kernel<..., RZ_SHADOW_DEFER> -> rz_shadow_kernel); null when shadow rays are walked inline
    // hand-over record of the deferred shadow rays, ONE contiguous record of nee_quads float4 per pixel (the shadow kernels follow
    // their own sorted order, so every array they read costs a scattered cache line per pixel: nine arrays were 5.3 GB per launch
    // on config E).  [0] radiance before next-event estimation, bits(path continues | NEE ran << 1 | sample mask << 2);
    // [1] shadow-ray origin; [2], [3] a, b of final += (direct * a) * b; [4 + 2k] direction + far of sample k, [5 + 2k] its
    // unshadowed contribution
    float4* nee;
    uint32_t nee_quads;
    uint32_t* shadow_key;         // deferred shadow rays: sort key of the pixel's shadow rays (origin cell + direction towards the light)
    const uint32_t* shadow_perm;  // rz_shadow_kernel: thread i finishes local pixel shadow_perm[i] (nullptr: follow `perm`)
    // Resident kernels, heaviest first (round 4): a launch lasts until its slowest unit (a tile's / a wave's chain of passes) is done, and a
    // unit that starts late ends late.  Every unit writes what its batch cost (clock ticks) into unit_cost; the host sorts the units by
    // falling cost now and then, and workgroup b of the next launches works on unit launch_order[b] — the long chains start first, the
    // short ones fill the end.  Execution order only: storage stays indexed by the unit.  nullptr = in unit order.
    const uint32_t* launch_order;
    uint32_t* unit_cost;
};

struct v3 {
    float x, y, z;
};
struct col4 {
    float r, g, b, a;
};

RZ_DEV v3 V3(float x, float y, float z) { return v3{x, y, z}; }
RZ_DEV v3 ld3(const float* p) { return v3{p[0], p[1], p[2]}; }
RZ_DEV v3 xyz(float4 f) { return v3{f.x, f.y, f.z}; }
RZ_DEV v3 operator+(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RZ_DEV v3 operator-(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RZ_DEV v3 operator*(v3 a, v3 b) { return v3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RZ_DEV v3 operator/(v3 a, v3 b) { return v3{a.x / b.x, a.y / b.y, a.z / b.z}; }
RZ_DEV v3 operator*(v3 a, float s) { return v3{a.x * s, a.y * s, a.z * s}; }
RZ_DEV v3 operator/(v3 a, float s) { return v3{a.x / s, a.y / s, a.z / s}; }
RZ_DEV v3 operator-(v3 a) { return v3{-a.x, -a.y, -a.z}; }
RZ_DEV float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RZ_DEV v3 cross(v3 a, v3 b) { return v3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
RZ_DEV float magnitude(v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
RZ_DEV float rcp_magnitude(v3 a) { return 1.0f / magnitude(a); }
RZ_DEV v3 normalized(v3 a) { return a * rcp_magnitude(a); }
RZ_DEV float similarity(v3 a, v3 b) { return dot(a, b) * (rcp_magnitude(a) * rcp_magnitude(b)); }

RZ_DEV col4 splat(float v) { return col4{v, v, v, v}; }
RZ_DEV col4 from_u8(uint32_t rgba) {
    return col4{float(rgba & 255u) / 255.0f, float((rgba >> 8) & 255u) / 255.0f, float((rgba >> 16) & 255u) / 255.0f,
                float(rgba >> 24) / 255.0f};
}
RZ_DEV col4 operator+(col4 a, col4 b) { return col4{a.r + b.r, a.g + b.g, a.b + b.b, a.a + b.a}; }
RZ_DEV col4 operator-(col4 a, col4 b) { return col4{a.r - b.r, a.g - b.g, a.b - b.b, a.a - b.a}; }
RZ_DEV col4 operator*(col4 a, col4 b) { return col4{a.r * b.r, a.g * b.g, a.b * b.b, a.a * b.a}; }
RZ_DEV col4 operator*(col4 a, float s) { return col4{a.r * s, a.g * s, a.b * s, a.a * s}; }
RZ_DEV col4 div_scalar(col4 a, float s) { return a * (1.0f / s); }  // ColorF / float: reciprocal multiply
RZ_DEV col4 operator/(col4 a, col4 b) { return col4{a.r / b.r, a.g / b.g, a.b / b.b, a.a / b.a}; }
RZ_DEV float lerpf(float a, float b, float t) { return a + (b - a) * t; }
RZ_DEV col4 lerp(col4 a, col4 b, float t) { return a + (b - a) * t; }

// --- RNG: cpu_render_utils.cpp:8-27 -----------------------------------------------------
struct Rng {
    float a, b;
    RZ_DEV Rng(float sx, float sy, float r) : a(sx + sy), b(r * 245.310913f) {}
    RZ_DEV float unsignedUniform() {
        const float af = (a + 0.2311362f) * (b + 13.054377f);
        const float bf = (a + 251.78431f) + (b - 73.054312f);
        a = af - float(int32_t(af));
        b = bf - float(int32_t(bf));
        return fabsf(b);
    }
    RZ_DEV float signedUniform() { return unsignedUniform() * 2.0f - 1.0f; }
};
RZ_DEV uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
// entry i of the pass's 256-entry seed table (hiprz.h: hiprz_seed_value), computed in place
RZ_DEV float seed_value(uint32_t seed, uint32_t pass, uint32_t i) {
    uint32_t h = mix32(seed ^ mix32(pass + 0x9E3779B9u));
    h = mix32(h ^ (i * 0x85EBCA6Bu + 1u));
    return float(h >> 8) * (20.0f / 16777216.0f) - 10.0f;
}

struct Ray {
    v3 o, d;
    float near_, far_;
};

struct Counters {
    uint32_t box_tests = 0, tri_tests = 0, hits = 0, shadow_rays = 0, light_samples = 0, texel_fetches = 0, finished = 0;
    uint32_t shadow_box_tests = 0, shadow_tri_tests = 0;
};
#define RZ_COUNT(field) \
    if constexpr (COUNT) cnt.field++

// --- BoundingBox::rayIntersection: render_parts.cpp:197-217 -----------------------------
RZ_DEV float min_lt(float a, float b) { return a < b ? a : b; }
RZ_DEV float max_gt(float a, float b) { return a > b ? a : b; }
RZ_DEV bool box_hit(v3 mn, v3 mx, const Ray& r) {
    const float t1 = (mn.x - r.o.x) / r.d.x;
    const float t2 = (mx.x - r.o.x) / r.d.x;
    const float t3 = (mn.y - r.o.y) / r.d.y;
    const float t4 = (mx.y - r.o.y) / r.d.y;
    const float t5 = (mn.z - r.o.z) / r.d.z;
    const float t6 = (mx.z - r.o.z) / r.d.z;
    const float tmin = max_gt(max_gt(min_lt(t1, t2), min_lt(t3, t4)), min_lt(t5, t6));
    const float tmax = min_lt(min_lt(max_gt(t1, t2), max_gt(t3, t4)), max_gt(t5, t6));
    return !(tmax < r.near_ || tmin > tmax || tmin > r.far_);
}

// --- Moller-Trumbore: mesh_component.cpp:52-114 -----------------------------------------
// The device triangle record carries v1 and the two edges v2 - v1, v3 - v1 (computed on upload with the same fp32
// subtraction the reference performs per test, mesh_component.cpp:56-57).
RZ_DEV bool tri_hit(v3 v1, v3 edge1, v3 edge2, const Ray& r, float& t_out, float& b1_out, float& b2_out, float& det_out) {
    const v3 pvec = cross(r.d, edge2);
    float det = dot(edge1, pvec);
    det += float(uint32_t(det > -1.0e-7f) & uint32_t(det < 1.0e-7f)) * 1.0e-7f;
    const float inv_det = 1.0f / det;
    const v3 tvec = r.o - v1;
    const float b1 = dot(tvec, pvec) * inv_det;
    if (b1 < 0.0f || b1 > 1.0f) return false;
    const v3 qvec = cross(tvec, edge1);
    const float b2 = dot(r.d, qvec) * inv_det;
    if (b2 < 0.0f || b1 + b2 > 1.0f) return false;
    const float t = dot(edge2, qvec) * inv_det;
    if (t <= r.near_ || t >= r.far_) return false;
    t_out = t, b1_out = b1, b2_out = b2, det_out = det;
    return true;
}

struct Instance {
    v3 position, scale, xa, ya, za, bb_min, bb_max;
    uint32_t blas_root, material_base, material_count;
};
// device instance records keep the box interleaved like nodes: [5] = (min.x, max.x, min.y, max.y), [6].xy = (min.z, max.z)
RZ_DEV void load_instance_box(const DScene& s, uint32_t i, float4& b0, float4& b1) {
    b0 = s.instances[7 * i + 5];
    b1 = s.instances[7 * i + 6];
}
struct InstanceXform {
    v3 position, scale, xa, ya, za;
    uint32_t blas_root;
    bool unit_scale;  // scale == (1,1,1): dividing by it is the identity and is skipped
};
RZ_DEV InstanceXform load_instance_xform(const DScene& s, uint32_t i) {
    const float4 a = s.instances[7 * i + 0], b = s.instances[7 * i + 1], c = s.instances[7 * i + 2],
                 d = s.instances[7 * i + 3], e = s.instances[7 * i + 4];
    InstanceXform x;
    x.position = xyz(a), x.blas_root = __float_as_uint(a.w);
    x.scale = xyz(b);
    x.unit_scale = __float_as_uint(d.w) != 0u;  // pad0 of the device copy (set on upload)
    x.xa = xyz(c), x.ya = xyz(d), x.za = xyz(e);
    return x;
}
// CoordSystem::transformForward / transformBackward: render_parts.cpp:42-50
RZ_DEV v3 transform_forward(v3 xa, v3 ya, v3 za, v3 v) { return xa * v.x + ya * v.y + za * v.z; }
RZ_DEV v3 transform_backward(v3 xa, v3 ya, v3 za, v3 v) {
    return v3{xa.x * v.x + xa.y * v.y + xa.z * v.z, ya.x * v.x + ya.y * v.y + ya.z * v.z,
              za.x * v.x + za.y * v.y + za.z * v.z};
}
// Transformation::transformG2L (render_parts.cpp:117-125) + the range rescale of
// cpu_engine_kernel.cpp:307-312 / :442-445.  Returns the length factor.
RZ_DEV float to_local(const InstanceXform& x, const Ray& g, Ray& l) {
    l.o = g.o - x.position;
    l.o = transform_backward(x.xa, x.ya, x.za, l.o);
    l.o = l.o / x.scale;
    l.d = transform_backward(x.xa, x.ya, x.za, g.d);
    l.d = l.d / x.scale;
    const float len = magnitude(l.d);
    l.near_ = g.near_ * len;
    l.far_ = g.far_ * len;
    l.d = normalized(l.d);
    return len;
}

// ---------------------------------------------------------------------------------------
// Tree walk.  The reference descends depth-first, first child then second, testing a
// node's box when it is entered (cpu_engine_kernel.cpp:254-277, 331-352) — a FIXED order.
// Three walks visit the same boxes and triangles in that same order (and a fourth, the front-to-back
// cooperative walk, reaches the same hits with fewer tests):
//
//  MODE 1 "LDS stack": nested loops (world tree -> instances of a leaf -> mesh tree) with an
//  explicit per-lane stack in LDS (level-major columns: lanes on one level hit distinct banks).
//  MODE 2 "workgroup-binned": the (ray, instance) visits of a workgroup's 256 rays are binned by
//  instance in LDS and processed by dense waves (scenes staged in LDS whose meshes are single leaves).
//  MODE 3 "skip links": stack-free — every node carries a link to whatever follows its subtree.
//  Measured and removed (DESIGN.md §5 keeps the numbers): a threaded flat walk graph with instance
//  pseudo-nodes, persistent lanes on it, requeue rounds, a persistent wave pool.
// ---------------------------------------------------------------------------------------
struct Hit {
    int32_t instance;  // -1 = none
    uint32_t triangle;
    float bx, by;
    bool external;
};


// Correctly rounded fp32 division with a reciprocal shared between numerators.  This is the
// instruction sequence hipcc emits for `n / d` (v_rcp_f32, two fma to refine it, then
// mul + fma,fma + fma,fma on the quotient) WITHOUT the v_div_scale / v_div_fixup wrapping,
// which is an identity when no operand or result needs rescaling: |d| in [2^-40, 4) and
// n == 0 or |n| in [2^-84, 2^41) (see prepare() and the upload-time check of box coordinates).
// Outside that range the plain `/` is used.  The two quotients of one axis (box min and max over the
// same ray component) run as ONE packed sequence (v_pk_mul_f32 + 4 v_pk_fma_f32: measured 1.05 ns
// per fma-lane against 1.8 ns for v_fma_f32).  Proven bit-equal to `/` by hiprz_selftest().
typedef float f2 __attribute__((ext_vector_type(2)));
RZ_DEV float refined_rcp(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
RZ_DEV float div_shared(float n, float d, float y) {
    float q = n * y;
    float r = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(r, y, q);
    r = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(r, y, q);
}
RZ_DEV f2 div_shared2(f2 n, float d, float y) {
    const f2 nd = {-d, -d}, yy = {y, y};
    f2 q = n * yy;
    f2 r = __builtin_elementwise_fma(nd, q, n);
    q = __builtin_elementwise_fma(r, yy, q);
    r = __builtin_elementwise_fma(nd, q, n);
    return __builtin_elementwise_fma(r, yy, q);
}
// |x| in [2^lo, 2^hi) tested on the exponent field
RZ_DEV bool exponent_in(float x, int lo, int hi) {
    const uint32_t e = (__float_as_uint(x) >> 23) & 0xFFu;
    return (e - uint32_t(lo + 127)) < uint32_t(hi - lo);
}
RZ_DEV bool zero_or_exponent_in(float x, int lo, int hi) {
    return (__float_as_uint(x) & 0x7FFFFFFFu) == 0u || exponent_in(x, lo, hi);
}
#ifdef RZ_PHASE_STATS  // diagnostic build: wave-level executions and active lanes of the MODE 3 walk's steps (tools/phase_stats.py)
static __device__ unsigned long long rz_phase[16];  // one copy per translation unit: read through that unit's hiprz_read_phase_stats
#define RZ_PHASE(k)                                                                                        \
    do {                                                                                                   \
        const unsigned long long rz_a = __ballot(1);                                                       \
        if (int(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))) == __ffsll((long long)rz_a) - 1) { \
            atomicAdd(&rz_phase[2 * (k)], 1ull);                                                           \
            atomicAdd(&rz_phase[2 * (k) + 1], (unsigned long long)__popcll(rz_a));                         \
        }                                                                                                  \
    } while (0)
#else
#define RZ_PHASE(k)
#endif
struct WalkRay {
    v3 o, d, y;  // y = refined reciprocal of d (valid when `fast`)
    float near_, far_;
    bool fast;
};
template <bool SHARED_RCP>
RZ_DEV void prepare(WalkRay& r, bool scene_fast) {
    if constexpr (!SHARED_RCP) {
        r.fast = false;
        return;
    }
    r.fast = scene_fast && exponent_in(r.d.x, -40, 2) && exponent_in(r.d.y, -40, 2) && exponent_in(r.d.z, -40, 2) &&
             zero_or_exponent_in(r.o.x, -60, 40) && zero_or_exponent_in(r.o.y, -60, 40) && zero_or_exponent_in(r.o.z, -60, 40);
    r.y = v3{refined_rcp(r.d.x), refined_rcp(r.d.y), refined_rcp(r.d.z)};
}
// min / max without NaN handling (operands are finite on the path that uses them)
RZ_DEV float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
RZ_DEV float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
RZ_DEV float vmin3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
RZ_DEV float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// BoundingBox::rayIntersection (render_parts.cpp:197-217) on a prepared ray.  The box comes as the
// device stores it: b0 = (min.x, max.x, min.y, max.y), b1.xy = (min.z, max.z).
// The same test with the six quotients computed one by one (the same instruction sequence per quotient, unpacked): for the
// world-level tests of the cooperative walk — a handful per ray — where keeping the ray's components as register PAIRS for the
// packed form costs more (18 VGPRs live across the whole kernel) than the few extra instructions.
template <bool SHARED_RCP>
RZ_DEV void box_range_unpacked(float4 b0, float4 b1, const WalkRay& r, float& tmin, float& tmax) {
    if (SHARED_RCP && __all(r.fast)) {  // wave-uniform branch
        const float t1 = div_shared(b0.x - r.o.x, r.d.x, r.y.x), t2 = div_shared(b0.y - r.o.x, r.d.x, r.y.x);
        const float t3 = div_shared(b0.z - r.o.y, r.d.y, r.y.y), t4 = div_shared(b0.w - r.o.y, r.d.y, r.y.y);
        const float t5 = div_shared(b1.x - r.o.z, r.d.z, r.y.z), t6 = div_shared(b1.y - r.o.z, r.d.z, r.y.z);
        tmin = vmax3(vmin(t1, t2), vmin(t3, t4), vmin(t5, t6));
        tmax = vmin3(vmax(t1, t2), vmax(t3, t4), vmax(t5, t6));
        return;
    }
    const float t1 = (b0.x - r.o.x) / r.d.x;
    const float t2 = (b0.y - r.o.x) / r.d.x;
    const float t3 = (b0.z - r.o.y) / r.d.y;
    const float t4 = (b0.w - r.o.y) / r.d.y;
    const float t5 = (b1.x - r.o.z) / r.d.z;
    const float t6 = (b1.y - r.o.z) / r.d.z;
    tmin = max_gt(max_gt(min_lt(t1, t2), min_lt(t3, t4)), min_lt(t5, t6));
    tmax = min_lt(min_lt(max_gt(t1, t2), max_gt(t3, t4)), max_gt(t5, t6));
}
template <bool SHARED_RCP>
RZ_DEV bool box_hit_unpacked(float4 b0, float4 b1, const WalkRay& r) {
    float tmin, tmax;
    box_range_unpacked<SHARED_RCP>(b0, b1, r, tmin, tmax);
    return !(tmax < r.near_ || tmin > tmax || tmin > r.far_);
}
// The same verdict for less arithmetic (the mesh-level node test of the cooperative walks, their most frequent operation).  The six
// quotients only matter through three comparisons.  One multiplication by the ray's refined reciprocal gives each quotient to within
// 2^-21 of its correctly rounded value (the numerator is the exact path's own: one rounded subtraction; the reciprocal is within an ulp,
// the product adds a rounding), and max / min of values that are each within a relative band are within the same band (x -> x +- e|x|
// is monotone).  So bounds that are 2^-20 wide decide the verdict whenever no comparison falls inside them — certainly hit, or
// certainly missed — and only when a lane of the wave cannot tell does the wave run the exact sequence (hiprz_selftest compares the
// verdicts on random boxes and counts how often that happens).
#ifndef RZ_FILTERED_BOX_TEST
#define RZ_FILTERED_BOX_TEST 1
#endif
RZ_DEV void box_filter(float4 b0, float4 b1, const WalkRay& r, bool& missed, bool& hit) {  // r.fast lanes only
    const float a1 = (b0.x - r.o.x) * r.y.x, a2 = (b0.y - r.o.x) * r.y.x;
    const float a3 = (b0.z - r.o.y) * r.y.y, a4 = (b0.w - r.o.y) * r.y.y;
    const float a5 = (b1.x - r.o.z) * r.y.z, a6 = (b1.y - r.o.z) * r.y.z;
    const float tmin = vmax3(vmin(a1, a2), vmin(a3, a4), vmin(a5, a6));
    const float tmax = vmin3(vmax(a1, a2), vmax(a3, a4), vmax(a5, a6));
    const float e = 9.5367431640625e-07f;  // 2^-20
    const float dmin = e * fabsf(tmin), dmax = e * fabsf(tmax);
    const float lo_min = tmin - dmin, hi_min = tmin + dmin, lo_max = tmax - dmax, hi_max = tmax + dmax;
    missed = hi_max < r.near_ || lo_min > hi_max || lo_min > r.far_;
    hit = lo_max >= r.near_ && hi_min <= lo_max && hi_min <= r.far_;
}
template <bool SHARED_RCP>
RZ_DEV bool box_hit_filtered(float4 b0, float4 b1, const WalkRay& r) {
#if RZ_FILTERED_BOX_TEST
    if (SHARED_RCP && __all(r.fast)) {  // wave-uniform branch
        bool missed, hit;
        box_filter(b0, b1, r, missed, hit);
        if (__all(missed || hit)) return hit;
    }
#endif
    return box_hit_unpacked<SHARED_RCP>(b0, b1, r);
}
template <bool SHARED_RCP>
RZ_DEV bool box_hit(float4 b0, float4 b1, const WalkRay& r) {
    if (SHARED_RCP && __all(r.fast)) {  // wave-uniform branch
        const f2 tx = div_shared2(f2{b0.x, b0.y} - f2{r.o.x, r.o.x}, r.d.x, r.y.x);
        const f2 ty = div_shared2(f2{b0.z, b0.w} - f2{r.o.y, r.o.y}, r.d.y, r.y.y);
        const f2 tz = div_shared2(f2{b1.x, b1.y} - f2{r.o.z, r.o.z}, r.d.z, r.y.z);
        // every t is finite here, so `a < b ? a : b` is the plain minimum (equal values may differ in the sign of zero only)
        const float tmin = vmax3(vmin(tx.x, tx.y), vmin(ty.x, ty.y), vmin(tz.x, tz.y));
        const float tmax = vmin3(vmax(tx.x, tx.y), vmax(ty.x, ty.y), vmax(tz.x, tz.y));
        return !(tmax < r.near_ || tmin > tmax || tmin > r.far_);
    }
    const float t1 = (b0.x - r.o.x) / r.d.x;
    const float t2 = (b0.y - r.o.x) / r.d.x;
    const float t3 = (b0.z - r.o.y) / r.d.y;
    const float t4 = (b0.w - r.o.y) / r.d.y;
    const float t5 = (b1.x - r.o.z) / r.d.z;
    const float t6 = (b1.y - r.o.z) / r.d.z;
    const float tmin = max_gt(max_gt(min_lt(t1, t2), min_lt(t3, t4)), min_lt(t5, t6));
    const float tmax = min_lt(min_lt(max_gt(t1, t2), max_gt(t3, t4)), max_gt(t5, t6));
    return !(tmax < r.near_ || tmin > tmax || tmin > r.far_);
}
RZ_DEV bool tri_hit(v3 v1, v3 edge1, v3 edge2, const WalkRay& r, float& t, float& b1, float& b2, float& det) {
    Ray q;
    q.o = r.o, q.d = r.d, q.near_ = r.near_, q.far_ = r.far_;
    return tri_hit(v1, edge1, edge2, q, t, b1, b2, det);
}

// ---- MODE 1: nested loops with an LDS stack ----
struct LdsStack {
    uint32_t* column;  // this lane's column: entry k lives at column[k * blockDim.x]
    uint32_t sp;
    RZ_DEV explicit LdsStack(uint32_t* lds_column) : column(lds_column), sp(0) {}
    RZ_DEV uint32_t mark() const { return sp; }
    RZ_DEV uint32_t next(uint32_t base) {  // subtree finished: pop the pending second child
        if (sp == base) return RZ_END;
        sp -= 1;
        return column[sp * blockDim.x];
    }
    RZ_DEV uint32_t descend(uint32_t first_child) {  // enter inner node: remember the second child
        column[sp * blockDim.x] = first_child + 1u;
        sp += 1;
        return first_child;
    }
};

// Instance entry: Transformation::transformG2L (render_parts.cpp:117-125) + the range rescale of
// cpu_engine_kernel.cpp:307-312 / :442-445 on a prepared ray.  Returns the length factor.
template <bool RCP>
RZ_DEV float to_local(const InstanceXform& x, const WalkRay& g, WalkRay& l, bool scene_fast) {
    l.o = transform_backward(x.xa, x.ya, x.za, g.o - x.position);
    l.d = transform_backward(x.xa, x.ya, x.za, g.d);
    if (!x.unit_scale) {
        l.o = l.o / x.scale;
        l.d = l.d / x.scale;
    }
    const float len = magnitude(l.d);
    l.near_ = g.near_ * len;
    l.far_ = g.far_ * len;
    l.d = l.d * (1.0f / len);
    prepare<RCP>(l, scene_fast);
    return len;
}

// closestIntersection(const Mesh&, ...): cpu_engine_kernel.cpp:331-352
template <bool COUNT, bool RCP, bool PACKED = true>
RZ_DEV bool closest_in_mesh_stack(const DScene& s, LdsStack& w, uint32_t root, WalkRay& lr, Hit& hit, Counters& cnt) {
    bool found = false;
    const uint32_t base = w.mark();
    uint32_t n = root, guard = 0u;
    while (n != RZ_END) {
        RZ_GUARD(guard);
        const float4 n0 = s.nodes[2 * n], n1 = s.nodes[2 * n + 1];
        RZ_PHASE(3);
        RZ_COUNT(box_tests);
        if (PACKED ? box_hit<RCP>(n0, n1, lr) : box_hit_unpacked<RCP>(n0, n1, lr)) {
            const uint32_t begin = __float_as_uint(n1.z), meta = __float_as_uint(n1.w);
            if (!(meta & HIPRZ_NODE_LEAF)) {
                n = w.descend(begin);
                continue;
            }
            const uint32_t end = begin + (meta & HIPRZ_NODE_COUNT_MASK);
            for (uint32_t i = begin; i < end; ++i) {
                const float4 a = s.tris[3 * i], b = s.tris[3 * i + 1], c = s.tris[3 * i + 2];
                float t, b1, b2, det;
                RZ_PHASE(4);
                RZ_COUNT(tri_tests);
                if (tri_hit(xyz(a), xyz(b), xyz(c), lr, t, b1, b2, det)) {
                    lr.far_ = t;
                    hit.triangle = i;
                    hit.external = det > 0.0f;
                    hit.bx = b1, hit.by = b2;
                    found = true;
                }
            }
        }
        n = w.next(base);
    }
    return found;
}

// traverseWorld + closestIntersection(instance): cpu_engine_kernel.cpp:254-330
template <bool COUNT, bool RCP>
RZ_DEV int closest_hit_stack(const DScene& s, uint32_t* lds_column, Ray& ray, Hit& hit, Counters& cnt) {
    LdsStack w(lds_column);
    const bool scene_fast = s.fast_div != 0u;
    WalkRay g;
    g.o = ray.o, g.d = ray.d, g.near_ = ray.near_, g.far_ = ray.far_;
    prepare<RCP>(g, scene_fast);
    uint32_t n = s.tlas_root, guard = 0u;
    while (n != RZ_END) {
        RZ_GUARD(guard);
        const float4 n0 = s.nodes[2 * n], n1 = s.nodes[2 * n + 1];
        RZ_COUNT(box_tests);
        if (box_hit<RCP>(n0, n1, g)) {
            const uint32_t begin = __float_as_uint(n1.z), meta = __float_as_uint(n1.w);
            if (!(meta & HIPRZ_NODE_LEAF)) {
                n = w.descend(begin);
                continue;
            }
            const uint32_t end = begin + (meta & HIPRZ_NODE_COUNT_MASK);
            for (uint32_t i = begin; i < end; ++i) {
                const uint32_t inst = s.tlas_order[i];
                float4 ib0, ib1;
                load_instance_box(s, inst, ib0, ib1);
                RZ_COUNT(box_tests);
                if (!box_hit<RCP>(ib0, ib1, g)) continue;
                const InstanceXform x = load_instance_xform(s, inst);
                WalkRay lr;
                const float len = to_local<RCP>(x, g, lr, scene_fast);
                if (closest_in_mesh_stack<COUNT, RCP>(s, w, x.blas_root, lr, hit, cnt)) {
                    hit.instance = int32_t(inst);
                    g.near_ = lr.near_ / len;
                    g.far_ = lr.far_ / len;
                }
            }
        } else if (n == s.tlas_root) {
            return 0;  // root box missed (cpu_engine_kernel.cpp:283)
        }
        n = w.next(0u);
    }
    ray.near_ = g.near_, ray.far_ = g.far_;
    return hit.instance >= 0 ? 2 : 1;
}

// anyIntersection: cpu_engine_kernel.cpp:398-481
template <bool COUNT>
RZ_DEV float any_hit_stack(const DScene& s, uint32_t* lds_column, const Ray& ray, Counters& cnt) {
    LdsStack w(lds_column);
    const bool scene_fast = s.fast_div != 0u;
    WalkRay g;
    g.o = ray.o, g.d = ray.d, g.near_ = ray.near_, g.far_ = ray.far_;
    prepare<false>(g, scene_fast);
    uint32_t n = s.tlas_root, guard = 0u;
    while (n != RZ_END) {
        RZ_GUARD(guard);
        const float4 n0 = s.nodes[2 * n], n1 = s.nodes[2 * n + 1];
        RZ_COUNT(box_tests);
        RZ_COUNT(shadow_box_tests);
        if (box_hit<false>(n0, n1, g)) {
            const uint32_t begin = __float_as_uint(n1.z), meta = __float_as_uint(n1.w);
            if (!(meta & HIPRZ_NODE_LEAF)) {
                n = w.descend(begin);
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
                to_local<false>(x, g, lr, scene_fast);
                // anyIntersection(const Mesh&, ...) :450-481
                const uint32_t base = w.mark();
                uint32_t m = x.blas_root;
                while (m != RZ_END) {
                    RZ_GUARD(guard);
                    const float4 m0 = s.nodes[2 * m], m1 = s.nodes[2 * m + 1];
                    RZ_COUNT(box_tests);
        RZ_COUNT(shadow_box_tests);
                    if (box_hit<false>(m0, m1, lr)) {
                        const uint32_t mbegin = __float_as_uint(m1.z), mmeta = __float_as_uint(m1.w);
                        if (!(mmeta & HIPRZ_NODE_LEAF)) {
                            m = w.descend(mbegin);
                            continue;
                        }
                        const uint32_t mend = mbegin + (mmeta & HIPRZ_NODE_COUNT_MASK);
                        for (uint32_t j = mbegin; j < mend; ++j) {
                            const float4 a = s.tris[3 * j], b = s.tris[3 * j + 1], c = s.tris[3 * j + 2];
                            float t, b1, b2, det;
                            RZ_COUNT(tri_tests);
                            RZ_COUNT(shadow_tri_tests);
                            if (tri_hit(xyz(a), xyz(b), xyz(c), lr, t, b1, b2, det)) return 0.0f;
                        }
                    }
                    m = w.next(base);
                }
            }
        } else if (n == s.tlas_root) {
            return 1.0f;  // root box missed (:402)
        }
        n = w.next(0u);
    }
    return 1.0f;
}


// ---- MODE 2: workgroup-binned closest hit ("wave64 ballot/prefix compaction of live paths") ----
// With one thread = one ray, only the lanes whose ray enters an instance do the expensive part
// (instance transform, mesh-tree walk, triangle tests): ~30 % of a wave on the Cornell scene.  Here
// the 256 rays of a workgroup advance in rounds.  In a round every ray walks the world tree to ITS
// next instance whose box it hits (its own order, with its current range — the reference's sequence);
// the (ray, instance) items are then counted per instance, prefix-summed and scattered into a dense
// list in LDS, and lanes 0..n_items-1 each process ONE item: fetch that ray from LDS, enter the
// instance, walk the mesh tree, and write a closer hit back to the ray's slot.  Idle lanes become idle
// WAVES, which cost nothing, and a wave's items are (nearly) all the same instance, so its leaf loops
// have equal trip counts.  Every ray still sees its instances one after another with an updated
// range, so results equal the sequential walk bit for bit.
#define RZ_BIN_NONE 0xFFFFFFFFu
#define RZ_BIN_WIDE 0x80000000u  // item flag: one of the 8 lanes that share a visit of a single-leaf mesh with more than 4 triangles
RZ_DEV uint32_t octet_min(uint32_t v);  // (defined with the cooperative walk's helpers below)
struct BinnedLds {  // per-workgroup workspace carved from dynamic LDS (256 lanes)
    float* ray;         // [8][256]  o.xyz d.xyz near far
    uint32_t* hit;      // [5][256]  triangle, external, b1, b2, instance
    uint32_t* items;    // [256]     instance << 8 | source lane
    uint32_t* bins;     // [2][64] per-instance item counts, double-buffered by round
    uint32_t* stacks;   // [(world + mesh entries)][256] level-major stack columns
    static constexpr uint32_t kFixedBytes = 15u * 1024u;
    static __host__ uint32_t bytes_host(uint32_t world_entries, uint32_t mesh_entries) {
        return kFixedBytes + (world_entries + mesh_entries) * 1024u;
    }
    __device__ __forceinline__ explicit BinnedLds(unsigned char* base) {
        ray = reinterpret_cast<float*>(base);
        hit = reinterpret_cast<uint32_t*>(base + 8u * 1024u);
        items = reinterpret_cast<uint32_t*>(base + 13u * 1024u);
        bins = reinterpret_cast<uint32_t*>(base + 14u * 1024u);
        stacks = reinterpret_cast<uint32_t*>(base + kFixedBytes);
    }
};

// Worlds whose tree is ONE leaf (up to 8 instances: a Cornell box).  The reference tests every instance box of a leaf, one after the
// other, each against the range as it is when the walk gets there (cpu_engine_kernel.cpp:299-305).  Only `tmin > far` depends on
// what was hit before, and `tmax < near` on a near end that moves by a rounding at most (a hit rescales it through the instance's
// length factor and back): so all the boxes are tested ONCE, up front, at full lane utilisation — bit k of the mask = "tmax >= near
// and tmin <= tmax", tm[k] = tmin — and a round only compares tm[k] with the far end as it is then.  The same verdicts as the
// one-by-one walk, the same count of box tests; the divergent per-lane search of the next candidate (a wave iterates as often as
// its slowest lane) becomes eight uniform tests and a few compares per round.
template <bool COUNT, bool RCP>
RZ_DEV uint32_t pretest_leaf_instances(const DScene& s, uint32_t begin, uint32_t count, uint32_t from_k, const WalkRay& g, float (&tm)[8], bool counting, Counters& cnt) {
    uint32_t mask = 0u;
#pragma unroll
    for (uint32_t k = 0; k < 8u; ++k) {
        if (k < count && k >= from_k) {
            float4 ib0, ib1;
            load_instance_box(s, s.tlas_order[begin + k], ib0, ib1);
            float tmin, tmax;
            box_range_unpacked<RCP>(ib0, ib1, g, tmin, tmax);
            if (counting) { RZ_PHASE(1); RZ_COUNT(box_tests); }
            if (!(tmax < g.near_ || tmin > tmax)) mask |= 1u << k, tm[k] = tmin;
        }
    }
    return mask;
}

// Must be called by ALL 256 threads of the workgroup (it contains barriers); `active` = this lane
// carries a ray.  Returns 0 / 1 / 2 like closest_hit().
template <bool COUNT, bool RCP, bool FLAT = false>  // FLAT: the host guarantees a one-leaf world tree (<= 8 instances)
__device__ __forceinline__ int closest_hit_binned(const DScene& s, unsigned char* workspace, bool active, Ray& ray, Hit& hit,
                                                  Counters& cnt) {
    const uint32_t tid = threadIdx.x;
    BinnedLds lds(workspace);
    hit.instance = -1, hit.triangle = 0, hit.bx = hit.by = 0.0f, hit.external = true;
    if (s.n_instances == 0) return 0;  // uniform

    lds.ray[0 * 256 + tid] = ray.o.x, lds.ray[1 * 256 + tid] = ray.o.y, lds.ray[2 * 256 + tid] = ray.o.z;
    lds.ray[3 * 256 + tid] = ray.d.x, lds.ray[4 * 256 + tid] = ray.d.y, lds.ray[5 * 256 + tid] = ray.d.z;
    lds.ray[6 * 256 + tid] = ray.near_, lds.ray[7 * 256 + tid] = ray.far_;
    lds.hit[4 * 256 + tid] = 0xFFFFFFFFu;

    LdsStack world(lds.stacks + tid);
    uint32_t* mesh_column = lds.stacks + s.world_stack_entries * 256u + tid;
    const bool sorted = s.n_instances <= 64u;
    WalkRay g;
    g.o = ray.o, g.d = ray.d, g.near_ = ray.near_, g.far_ = ray.far_;
    const bool scene_fast = s.fast_div != 0u;
    prepare<RCP>(g, scene_fast);
    uint32_t n = active ? s.tlas_root : RZ_END;  // world-tree cursor of this lane's ray
    uint32_t leaf_i = 0, leaf_end = 0;
    bool root_missed = false;
    uint32_t round = 0u;
    // a world of one leaf: its instance boxes are tested up front (pretest_leaf_instances)
    const float4 root1 = s.nodes[2 * s.tlas_root + 1];
    const uint32_t flat_begin = __float_as_uint(root1.z), flat_count = __float_as_uint(root1.w) & HIPRZ_NODE_COUNT_MASK;
    const bool flat_world = FLAT;  // compile-time: the general world walk below drops out of the FLAT instantiation
    float tm[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    uint32_t flat_mask = 0u, flat_next = 0u;
    float flat_near = g.near_;
    if (flat_world && active) {
        const float4 root0 = s.nodes[2 * s.tlas_root];
        RZ_PHASE(0);
        RZ_COUNT(box_tests);
        if (box_hit_unpacked<RCP>(root0, root1, g)) flat_mask = pretest_leaf_instances<COUNT, RCP>(s, flat_begin, flat_count, 0u, g, tm, true, cnt);
        else root_missed = true;
    }
    if (tid < 128u) lds.bins[tid] = 0u;
    // Visits of a mesh that is ONE leaf of more than 4 triangles (a Cornell cube: 12) are shared by 8 lanes: a wave's dense items are a
    // mix of such visits and of 2-triangle walls, and with one lane per visit the wave's triangle loop ran as long as its longest item
    // (config B: 10.6 iterations per wave and pass at 21 of 64 lanes).  Bit k of wide_mask: instance k (of the first 64) has such a mesh.
    unsigned long long wide_mask = 0ull;
    if (sorted) {
        const uint32_t k = tid & 63u;
        bool wide = false;
        if (k < s.n_instances) {
            const uint32_t root = __float_as_uint(s.instances[7 * k].w);
            const uint32_t meta = __float_as_uint(s.nodes[2 * root + 1].w);
            wide = (meta & HIPRZ_NODE_LEAF) != 0u && (meta & HIPRZ_NODE_COUNT_MASK) > 4u;
        }
        wide_mask = __ballot(wide);
    }
    __syncthreads();

    uint32_t guard = 0u;
    while (round < (1u << 20)) {  // workgroup-uniform bound: a ray enters each instance at most once
        // A. advance this ray to its next candidate instance (traverseWorld, cpu_engine_kernel.cpp:254-277, 305)
        RZ_PHASE(5);
        uint32_t cand = RZ_BIN_NONE;
        if (flat_world) {
            if (__any(flat_mask != 0u && __float_as_uint(g.near_) != __float_as_uint(flat_near))) {
                // rare: a hit moved the near end by a rounding — the boxes not yet visited are tested again with it (the ray comes
                // back from its LDS slot; the tests were counted the first time)
                if (flat_mask != 0u && __float_as_uint(g.near_) != __float_as_uint(flat_near)) {
                    WalkRay t;
                    t.o = V3(lds.ray[0 * 256 + tid], lds.ray[1 * 256 + tid], lds.ray[2 * 256 + tid]);
                    t.d = V3(lds.ray[3 * 256 + tid], lds.ray[4 * 256 + tid], lds.ray[5 * 256 + tid]);
                    t.near_ = g.near_, t.far_ = g.far_;
                    prepare<RCP>(t, scene_fast);
                    flat_mask = pretest_leaf_instances<COUNT, RCP>(s, flat_begin, flat_count, flat_next, t, tm, false, cnt);
                    flat_near = g.near_;
                }
            }
#pragma unroll
            for (uint32_t k = 0; k < 8u; ++k) {
                if (cand == RZ_BIN_NONE && k >= flat_next && ((flat_mask >> k) & 1u)) {
                    flat_next = k + 1u;
                    if (!(tm[k] > g.far_)) cand = s.tlas_order[flat_begin + k];
                }
            }
            if (cand == RZ_BIN_NONE) flat_mask = 0u;
        } else
        while (true) {
            RZ_GUARD(guard);
            if (leaf_i < leaf_end) {
                const uint32_t inst = s.tlas_order[leaf_i++];
                float4 ib0, ib1;
                load_instance_box(s, inst, ib0, ib1);
                RZ_PHASE(1);
                RZ_COUNT(box_tests);
                if (box_hit_unpacked<RCP>(ib0, ib1, g)) {
                    cand = inst;
                    break;
                }
                continue;
            }
            if (n == RZ_END) break;
            const float4 n0 = s.nodes[2 * n], n1 = s.nodes[2 * n + 1];
            RZ_PHASE(0);
            RZ_COUNT(box_tests);
            if (box_hit_unpacked<RCP>(n0, n1, g)) {
                const uint32_t begin = __float_as_uint(n1.z), meta = __float_as_uint(n1.w);
                if (!(meta & HIPRZ_NODE_LEAF)) {
                    n = world.descend(begin);
                    continue;
                }
                leaf_i = begin, leaf_end = begin + (meta & HIPRZ_NODE_COUNT_MASK);
            } else if (n == s.tlas_root) {
                root_missed = true;
            }
            n = world.next(0u);
        }

        // B. bin the items by instance: count (LDS atomics), exclusive prefix over the 64 bins computed
        //    redundantly by every wave (no barrier between scan and scatter), scatter.  The bins are
        //    double-buffered: this round's were zeroed during the previous round.
        uint32_t* bins = lds.bins + (round & 1u) * 64u;
        uint32_t rank = 0u;
        const uint32_t bin = sorted ? cand : 0u;
        const bool wide_item = cand != RZ_BIN_NONE && cand < 64u && ((wide_mask >> cand) & 1ull) != 0ull;
        if (cand != RZ_BIN_NONE) rank = atomicAdd(&bins[bin], wide_item ? 8u : 1u);  // in lanes: 8 per visit of a wide instance
        __syncthreads();
        const uint32_t lane = tid & 63u;
        const uint32_t c = bins[lane];
        // ONE scan for three prefix sums, packed: lanes of the wide bins (bits 0-11: they come first, so every visit's 8 lanes are an aligned
        // octet), lanes of the other bins (12-20), visits of all bins (21-29: the order when the lanes would not fit the workgroup)
        const bool wide_bin = ((wide_mask >> lane) & 1ull) != 0ull;
        uint32_t incl = wide_bin ? (c | ((c >> 3) << 21)) : ((c << 12) | (c << 21));
        const uint32_t own = incl;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off);
            if (int(lane) >= off) incl += v;
        }
        const uint32_t totals = __shfl(incl, 63);
        const uint32_t n_visits = totals >> 21;
        if (n_visits == 0u) break;  // workgroup-uniform
        const uint32_t wide_lanes = totals & 0xFFFu, narrow_lanes = (totals >> 12) & 0x1FFu;
        const bool split = wide_lanes != 0u && wide_lanes + narrow_lanes <= 256u;  // workgroup-uniform: else one lane per visit, as before
        const uint32_t n_items = split ? wide_lanes + narrow_lanes : n_visits;
        const uint32_t before = __shfl(incl - own, int(bin & 63u));
        if (cand != RZ_BIN_NONE) {
            if (split && wide_item) {
                const uint32_t at = (before & 0xFFFu) + rank;
#pragma unroll
                for (uint32_t j = 0; j < 8u; ++j) lds.items[at + j] = RZ_BIN_WIDE | (cand << 8) | tid;
            } else if (split) {
                lds.items[wide_lanes + ((before >> 12) & 0x1FFu) + rank] = (cand << 8) | tid;
            } else {
                lds.items[(before >> 21) + (wide_item ? rank >> 3 : rank)] = (cand << 8) | tid;
            }
        }
        if (tid < 64u) lds.bins[((round + 1u) & 1u) * 64u + tid] = 0u;
        __syncthreads();

        // C. dense: one lane per item (closestIntersection(instance) + (mesh), :299-352) — eight per visit of a wide instance, lane j
        //    of the octet testing triangles j, j + 8, ... of the mesh's one leaf.  The lane that takes item i rotates with the round and
        //    the workgroup, so the busy waves — and with them the SIMDs they live on — change from round to round instead of always
        //    being waves 0..1.
        const uint32_t slot = (tid - ((blockIdx.x + round) & 3u) * 64u) & 255u;
        if (slot < n_items) {
            RZ_PHASE(2);
            const uint32_t item = lds.items[slot], inst = (item & 0x7FFFFFFFu) >> 8, src = item & 255u;
            WalkRay w;
            w.o = V3(lds.ray[0 * 256 + src], lds.ray[1 * 256 + src], lds.ray[2 * 256 + src]);
            w.d = V3(lds.ray[3 * 256 + src], lds.ray[4 * 256 + src], lds.ray[5 * 256 + src]);
            w.near_ = lds.ray[6 * 256 + src], w.far_ = lds.ray[7 * 256 + src];
            const InstanceXform x = load_instance_xform(s, inst);
            WalkRay lr;
            const float len = to_local<RCP>(x, w, lr, scene_fast);
            if (item & RZ_BIN_WIDE) {
                // the mesh is one leaf: its box once per visit (counted by lane 0 of the octet), then this lane's share of its triangles —
                // each against the range the visit started with, shortened by this lane's own earlier hits; the octet's minimum below is
                // what the one-by-one loop ends with (the nearest hit, the first in leaf order among equal distances)
                const uint32_t j = slot & 7u;
                const float4 n0 = s.nodes[2 * x.blas_root], n1 = s.nodes[2 * x.blas_root + 1];
                RZ_PHASE(3);
                if (j == 0u) { RZ_COUNT(box_tests); }
                uint32_t wide_t = 0xFFFFFFFFu, wide_tri = 0xFFFFFFFFu;  // this lane's best triangle of the shared visit
                float wide_b1 = 0.0f, wide_b2 = 0.0f;
                bool wide_external = false;
                const float visit_near = lr.near_;
                if (box_hit_unpacked<RCP>(n0, n1, lr)) {
                    const uint32_t begin = __float_as_uint(n1.z), end = begin + (__float_as_uint(n1.w) & HIPRZ_NODE_COUNT_MASK);
                    for (uint32_t i = begin + j; i < end; i += 8u) {
                        const float4 a = s.tris[3 * i], b = s.tris[3 * i + 1], cc = s.tris[3 * i + 2];
                        float t, b1, b2, det;
                        RZ_PHASE(4);
                        RZ_COUNT(tri_tests);
                        if (tri_hit(xyz(a), xyz(b), xyz(cc), lr, t, b1, b2, det)) {
                            lr.far_ = t;
                            wide_t = __float_as_uint(t), wide_tri = i, wide_b1 = b1, wide_b2 = b2, wide_external = det > 0.0f;
                        }
                    }
                }
                // the octet's winner hands the visit's hit to the ray's slot (the 8 lanes of a visit take this branch together; distances
                // are positive: their bits order like they do)
                const uint32_t t_min = octet_min(wide_t);
                const bool nearest = wide_t != 0xFFFFFFFFu && wide_t == t_min;
                const uint32_t first = octet_min(nearest ? wide_tri : 0xFFFFFFFFu);
                if (nearest && wide_tri == first) {
                    lds.ray[6 * 256 + src] = visit_near / len;
                    lds.ray[7 * 256 + src] = __uint_as_float(wide_t) / len;
                    lds.hit[0 * 256 + src] = wide_tri;
                    lds.hit[1 * 256 + src] = wide_external ? 1u : 0u;
                    lds.hit[2 * 256 + src] = __float_as_uint(wide_b1);
                    lds.hit[3 * 256 + src] = __float_as_uint(wide_b2);
                    lds.hit[4 * 256 + src] = inst;
                }
            } else {
                LdsStack mesh(mesh_column);
                Hit h;
                if (closest_in_mesh_stack<COUNT, RCP, false>(s, mesh, x.blas_root, lr, h, cnt)) {  // single-leaf meshes: one box test per visit
                    lds.ray[6 * 256 + src] = lr.near_ / len;
                    lds.ray[7 * 256 + src] = lr.far_ / len;
                    lds.hit[0 * 256 + src] = h.triangle;
                    lds.hit[1 * 256 + src] = h.external ? 1u : 0u;
                    lds.hit[2 * 256 + src] = __float_as_uint(h.bx);
                    lds.hit[3 * 256 + src] = __float_as_uint(h.by);
                    lds.hit[4 * 256 + src] = inst;
                }
            }
        }
        __syncthreads();
        round += 1u;
        // D. the ray's owner picks up its (possibly shortened) range
        g.near_ = lds.ray[6 * 256 + tid];
        g.far_ = lds.ray[7 * 256 + tid];
    }
    // origin and direction were not kept in registers across the rounds: take them back from the slot
    ray.o = V3(lds.ray[0 * 256 + tid], lds.ray[1 * 256 + tid], lds.ray[2 * 256 + tid]);
    ray.d = V3(lds.ray[3 * 256 + tid], lds.ray[4 * 256 + tid], lds.ray[5 * 256 + tid]);
    ray.near_ = g.near_, ray.far_ = g.far_;
    const uint32_t inst = lds.hit[4 * 256 + tid];
    if (inst != 0xFFFFFFFFu) {
        hit.instance = int32_t(inst);
        hit.triangle = lds.hit[0 * 256 + tid];
        hit.external = lds.hit[1 * 256 + tid] != 0u;
        hit.bx = __uint_as_float(lds.hit[2 * 256 + tid]);
        hit.by = __uint_as_float(lds.hit[3 * 256 + tid]);
    }
    if (root_missed) return 0;
    return hit.instance >= 0 ? 2 : 1;
}


// ---- MODE 3: nested walk on skip links with the top of every tree cached in LDS ----
// For scenes whose records do not fit LDS (configs C, D) a segment is a chain of ~60-90 dependent node
// fetches served by L2.  The device copy of the nodes is laid out breadth-first over ALL trees
// (hiprz_api.hip: relayout), so the levels nearest the roots — the ones every ray visits — form a prefix;
// each workgroup stages that prefix (nodes + links) into LDS.  Following skip links instead of popping a
// stack means the walk needs no LDS stack at all, which is what frees the space for the cache.
struct TopCache {
    const float4* nodes;   // LDS: top_count x 2 float4
    const uint32_t* skip;  // LDS: top_count
    uint32_t count;
    static __host__ uint32_t bytes_host(uint32_t top_count) { return top_count * 36u; }
};
// Front-to-back mesh walk.  The reference visits a node's first child, then its second (cpu_engine_kernel.cpp:331-352);
// its builder puts the centroids BELOW the split plane into the first child (bvh_tree_node.hpp:150-215), so that fixed order is
// front-to-back only for rays that travel up the split axis.  Which child a ray should enter first depends on nothing but the
// sign of its direction along the node's split axis, i.e. on the ray's octant: per octant the whole visiting order is fixed, and
// so are the skip links.  The upload derives the links of all 8 octants (64-B records: node + 8 links); a lane picks its table
// once per mesh and the walk stays stack-free (and LDS-free).  The closest hit is the same: among equal distances the triangle the reference
// would have met first (lower index — triangles are stored in the reference's visiting order) wins (tri_hit_ordered).
// octant bit p = the direction component along the axis of partition type p is negative (X=2, Y=1, Z=0; type 3 = split by size:
// never flipped).
RZ_DEV uint32_t octant_of(v3 d) { return uint32_t(d.z < 0.0f) | (uint32_t(d.y < 0.0f) << 1) | (uint32_t(d.x < 0.0f) << 2); }
RZ_DEV void fetch_node_ordered(const DScene& s, uint32_t n, uint32_t oct, float4& n0, float4& n1, uint32_t& link) {
    // straight from the 64-B records: with rays in sorted order and every lane entering the nearer child first the top levels stay
    // in L1, and an LDS copy of them costs more than it saves (staging 48 B per node per 64-ray workgroup + a branch per fetch:
    // config C trace kernel 504 -> 430 us, D 1 454 -> 1 434 us without it)
    const float4* rec = s.nodes64 + 4 * size_t(n);
    n0 = rec[0], n1 = rec[1];
    link = reinterpret_cast<const uint32_t*>(rec + 2)[oct];
}
// Triangle::closestIntersection for a walk in another order than the reference's: a hit at exactly the current `far` replaces the
// held one when the reference would have met it first (`tie_ok`: a hit of THIS mesh is held and this triangle's index is lower).
RZ_DEV bool tri_hit_ordered(v3 v1, v3 edge1, v3 edge2, const WalkRay& r, bool tie_ok, float& t_out, float& b1_out, float& b2_out, float& det_out) {
    const v3 pvec = cross(r.d, edge2);
    float det = dot(edge1, pvec);
    det += float(uint32_t(det > -1.0e-7f) & uint32_t(det < 1.0e-7f)) * 1.0e-7f;
    const float inv_det = 1.0f / det;
    const v3 tvec = r.o - v1;
    const float b1 = dot(tvec, pvec) * inv_det;
    if (b1 < 0.0f || b1 > 1.0f) return false;
    const v3 qvec = cross(tvec, edge1);
    const float b2 = dot(r.d, qvec) * inv_det;
    if (b2 < 0.0f || b1 + b2 > 1.0f) return false;
    const float t = dot(edge2, qvec) * inv_det;
    if (t <= r.near_ || t > r.far_ || (t == r.far_ && !tie_ok)) return false;
    t_out = t, b1_out = b1, b2_out = b2, det_out = det;
    return true;
}
RZ_DEV void fetch_node(const DScene& s, const TopCache& top, uint32_t n, float4& n0, float4& n1, uint32_t& link) {
    if (n < top.count) {
        n0 = top.nodes[2 * n], n1 = top.nodes[2 * n + 1], link = top.skip[n];
    } else {
        n0 = s.nodes[2 * n], n1 = s.nodes[2 * n + 1], link = s.node_skip[n];
    }
}
// TIES: the trees may be in another order than the reference's (rebuilt or device-built): among equally distant triangles of a mesh the
// one the reference meets first wins (tri_hit_ordered on the triangles' reference positions), as in the cooperative walks.
template <bool COUNT, bool RCP, bool TIES = false>
RZ_DEV int closest_hit_skip(const DScene& s, const TopCache& top, Ray& ray, Hit& hit, Counters& cnt) {
    const bool scene_fast = s.fast_div != 0u;
    WalkRay g;
    g.o = ray.o, g.d = ray.d, g.near_ = ray.near_, g.far_ = ray.far_;
    prepare<RCP>(g, scene_fast);
    uint32_t n = s.tlas_root, guard = 0u;
    while (n != RZ_END) {
        RZ_GUARD(guard);
        float4 n0, n1;
        uint32_t link;
        fetch_node(s, top, n, n0, n1, link);
        RZ_PHASE(0);
        RZ_COUNT(box_tests);
        if (box_hit<RCP>(n0, n1, g)) {
            const uint32_t begin = __float_as_uint(n1.z), meta = __float_as_uint(n1.w);
            if (!(meta & HIPRZ_NODE_LEAF)) {
                n = begin;
                continue;
            }
            uint32_t end = begin + (meta & HIPRZ_NODE_COUNT_MASK);
            for (uint32_t i = begin; i < end; ++i) {
                uint32_t inst = s.tlas_order[i];
                float4 ib0, ib1;
                load_instance_box(s, inst, ib0, ib1);
                RZ_PHASE(1);
                RZ_COUNT(box_tests);
                if (!box_hit<RCP>(ib0, ib1, g)) continue;
                RZ_PHASE(2);
                const InstanceXform x = load_instance_xform(s, inst);
                WalkRay lr;
                float len = to_local<RCP>(x, g, lr, scene_fast);
                bool found = false;
                uint32_t held_refpos = 0u;
                uint32_t m = x.blas_root;
                // closestIntersection(const Mesh&, ...): cpu_engine_kernel.cpp:331-352, as a "while-while" walk in bounded
                // rounds: lanes without a leaf step through nodes until they HOLD one (at most walk_k steps per round), then the
                // lanes that hold a leaf test its triangles (at most walk_l per round), together.  With the leaf loop nested in
                // the node loop some lane is at a leaf in almost every step and the whole wave waits through its triangles (11 %
                // lane utilisation on the 301 k-triangle mesh); with unbounded phases the wave waits for the lane with the longest
                // search, then for the one with the fullest leaf.  0 = unbounded.
                uint32_t tj = 0u, tj_end = 0u;  // the held leaf's remaining triangles
                const uint32_t kmax = s.walk_k ? s.walk_k : 0xFFFFFFFFu, lmax = s.walk_l ? s.walk_l : 0xFFFFFFFFu;
                while (true) {
                    RZ_PHASE(5);
                    uint32_t k = 0u;
                    while (tj == tj_end && m != RZ_END && k < kmax) {
                        RZ_GUARD(guard);
                        k += 1u;
                        float4 m0, m1;
                        uint32_t mlink;
                        fetch_node(s, top, m, m0, m1, mlink);
                        RZ_PHASE(3);
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
                    if (tj == tj_end && m == RZ_END) break;  // walk finished
                    uint32_t l = 0u;
                    for (; tj < tj_end && l < lmax; ++tj, ++l) {
                        const float4 a = s.tris[3 * tj], b = s.tris[3 * tj + 1], c = s.tris[3 * tj + 2];
                        float t, b1, b2, det;
                        RZ_PHASE(4);
                        RZ_COUNT(tri_tests);
                        const uint32_t refpos = __float_as_uint(c.w);
                        if (TIES ? tri_hit_ordered(xyz(a), xyz(b), xyz(c), lr, found && refpos < held_refpos, t, b1, b2, det)
                                 : tri_hit(xyz(a), xyz(b), xyz(c), lr, t, b1, b2, det)) {
                            lr.far_ = t;
                            hit.triangle = tj;
                            hit.external = det > 0.0f;
                            hit.bx = b1, hit.by = b2;
                            found = true, held_refpos = refpos;
                        }
                    }
                }
                if (found) {
                    hit.instance = int32_t(inst);
                    g.near_ = lr.near_ / len;
                    g.far_ = lr.far_ / len;
                }
            }
        } else if (n == s.tlas_root) {
            return 0;  // root box missed (cpu_engine_kernel.cpp:283)
        }
        n = link;
    }
    ray.near_ = g.near_, ray.far_ = g.far_;
    return hit.instance >= 0 ? 2 : 1;
}

// ---- the front-to-back walk with a COOPERATIVE triangle phase ----
// After front to back the thin phase of the walk is the triangle test: a lane that holds a leaf tests its (up to 8) triangles one
// after the other while the lanes without a leaf wait — config D: 133 wave-level triangle steps per wave at 7.9 active lanes.
// Here every loop of the walk is wave-uniform (conditions are ballots, lanes carry predicates), so ALL 64 lanes reach the triangle
// phase, and the (ray, triangle) pairs of the lanes that hold leaves are dealt out over the whole wave in groups of 8 lanes:
//   a holder's (up to 8) triangles form one or two ENTRIES of at most 4; entry positions come from two ballots + mbcnt (no scan), and
//   the holder writes a 48-byte record per entry — mesh-space ray, range, first triangle, count — to LDS; the four lanes of quad e of
//   a step read record e (three ds_read_b128, broadcast within the quad) and test triangle j = lane & 3 of that entry; the quad's
//   closest hit is a 2-step DPP minimum over bits(t) (t > near >= 0, so the bit pattern orders like the number), the lowest lane
//   holding the minimum — the lowest triangle index, i.e. the reference's "first found wins" — writes (t, triangle, barycentrics,
//   side) to the entry's result slot, and the holder takes the better of its entries.  No atomics, one hand-over each way.
// The node phase of a round ends after walk_k steps, or as soon as walk_h lanes hold a leaf (a full step of 64 slots).  A leaf's triangles tested against the range the lane held when it reached the leaf, then reduced by
// (t, index), give what testing them one by one in index order gives (each accepted t is strictly smaller, or equal with a lower
// index than a hit of another leaf): the same hits as a per-lane front-to-back walk with tri_hit_ordered.  The caller brings all
// 64 lanes (`active` = has a ray).  LDS per wave: CoopLds::kBytes.
#define RZ_LDS __attribute__((address_space(3)))
typedef float f4 __attribute__((ext_vector_type(4)));  // a native vector: loads / stores through address-space-qualified pointers
RZ_DEV f4 F4(float x, float y, float z, float w) { return f4{x, y, z, w}; }
struct CoopLds {  // LDS-qualified pointers: ds_read / ds_write, not flat accesses
    RZ_LDS f4* rec;       // [3][128] by entry: (o.xyz, near), (d.xyz, far), bits(first triangle, count <= 4, held triangle + 1 or 0, R)
    RZ_LDS f4* res;       // [128]    by entry: bits(t) or ~0 = no hit, bits(triangle | external << 31), b1, b2
    // R: the winner's position in the reference's leaf order, written by the winning tester after the entry's four lanes have read the
    // record (one wave: LDS operations complete in issue order).  8 KiB per wave: twenty single-wave workgroups fill a CU's 160 KiB.
    static constexpr uint32_t kEntries = 128u, kBytes = 4u * kEntries * 16u;
    RZ_DEV explicit CoopLds(unsigned char* base) : rec((RZ_LDS f4*)base), res((RZ_LDS f4*)(base + 3u * kEntries * 16u)) {}
    RZ_DEV RZ_LDS uint32_t* ref(uint32_t e) const { return (RZ_LDS uint32_t*)(rec + 2u * kEntries + e) + 3; }
};
// Hand-over points of the cooperative phase: the LDS unit executes a wave's instructions in issue order, so a fence that keeps the
// compiler from moving LDS accesses across it is all one wave needs (the workgroup IS one wave).
RZ_DEV void rz_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
RZ_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// number of set bits of `mask` below this lane
RZ_DEV uint32_t rank_in(unsigned long long mask) { return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u)); }
// minimum of an unsigned value over each quad (aligned group of 4 lanes), in every lane of the quad: two quad_perm exchanges
RZ_DEV uint32_t quad_min(uint32_t v) {
    uint32_t o = uint32_t(__builtin_amdgcn_update_dpp(int(v), int(v), 0xB1, 0xF, 0xF, false));  // quad_perm:[1,0,3,2]
    v = o < v ? o : v;
    o = uint32_t(__builtin_amdgcn_update_dpp(int(v), int(v), 0x4E, 0xF, 0xF, false));           // quad_perm:[2,3,0,1]
    return o < v ? o : v;
}
// ... and over each octet (aligned group of 8 lanes): the quads' minima exchanged across the half row (row_half_mirror: lane i <-> 7 - i)
RZ_DEV uint32_t octet_min(uint32_t v) {
    v = quad_min(v);
    const uint32_t o = uint32_t(__builtin_amdgcn_update_dpp(int(v), int(v), 0x141, 0xF, 0xF, false));
    return o < v ? o : v;
}
// How the triangles of the lanes that hold a leaf are dealt out: a holder of c <= 4 triangles fills ONE entry (a quad of lanes), a
// holder of 5..8 two — positions come from two ballots (no scan): entries before mine = holders before me + two-entry holders before me.
struct CoopDeal {
    uint32_t n_entries, pos;
    bool big;
};
RZ_DEV CoopDeal coop_deal(bool holding, uint32_t c) {
    const unsigned long long hmask = __ballot(holding), bmask = __ballot(holding && c > 4u);
    return CoopDeal{uint32_t(__popcll(hmask)) + uint32_t(__popcll(bmask)), rank_in(hmask) + rank_in(bmask), c > 4u};
}
// One cooperative triangle step for the closest-hit walk.  Every lane calls it; `holding` lanes own a leaf [tj, tj + c), c <= 8.
template <bool COUNT>
RZ_DEV void coop_closest_triangles(const DScene& s, const CoopLds& lds, bool holding, uint32_t c, v3 lr_o, v3 lr_d, float lr_near, uint32_t tj,
                                   bool& found, uint32_t& held_ref, Hit& hit, float& far_, Counters& cnt) {
    const uint32_t lane = lane_id();
    const CoopDeal deal = coop_deal(holding, c);
    if (holding) {
        const f4 r0 = F4(lr_o.x, lr_o.y, lr_o.z, lr_near), r1 = F4(lr_d.x, lr_d.y, lr_d.z, far_);
        const uint32_t held = found ? held_ref + 1u : 0u;  // triangles are ranked by their position in the REFERENCE's leaf order
        lds.rec[deal.pos] = r0, lds.rec[CoopLds::kEntries + deal.pos] = r1;
        lds.rec[2u * CoopLds::kEntries + deal.pos] = F4(__uint_as_float(tj), __uint_as_float(c < 4u ? c : 4u), __uint_as_float(held), 0.0f);
        lds.res[deal.pos] = F4(__uint_as_float(0xFFFFFFFFu), 0.0f, 0.0f, 0.0f);
        if (deal.big) {
            lds.rec[deal.pos + 1u] = r0, lds.rec[CoopLds::kEntries + deal.pos + 1u] = r1;
            lds.rec[2u * CoopLds::kEntries + deal.pos + 1u] = F4(__uint_as_float(tj + 4u), __uint_as_float(c - 4u), __uint_as_float(held), 0.0f);
            lds.res[deal.pos + 1u] = F4(__uint_as_float(0xFFFFFFFFu), 0.0f, 0.0f, 0.0f);
        }
    }
    rz_wave_sync();
    for (uint32_t base = 0u; base < deal.n_entries * 4u; base += 64u) {  // wave-uniform
        const uint32_t e = (base + lane) >> 2, j = lane & 3u;
        uint32_t tbits = 0xFFFFFFFFu, tri = 0u, refpos = 0xFFFFFFFFu;
        float b1 = 0.0f, b2 = 0.0f, det = 0.0f;
        if (e < deal.n_entries) {
            const f4 r2 = lds.rec[2u * CoopLds::kEntries + e];
            if (j < __float_as_uint(r2.y)) {
                const f4 r0 = lds.rec[e], r1 = lds.rec[CoopLds::kEntries + e];
                WalkRay hr;
                hr.o = V3(r0.x, r0.y, r0.z), hr.d = V3(r1.x, r1.y, r1.z), hr.near_ = r0.w, hr.far_ = r1.w;
                tri = __float_as_uint(r2.x) + j;
                const float4 a = s.tris[3 * tri], b = s.tris[3 * tri + 1], cc = s.tris[3 * tri + 2];
                float t;
                RZ_PHASE(4);
                RZ_COUNT(tri_tests);
                refpos = __float_as_uint(cc.w);
                // a hit at exactly the held distance replaces the held one when the reference would have met it first: lower position
                if (tri_hit_ordered(xyz(a), xyz(b), xyz(cc), hr, refpos + 1u < __float_as_uint(r2.z), t, b1, b2, det)) tbits = __float_as_uint(t);
            }
        }
        const uint32_t tmin = quad_min(tbits);
        const bool candidate = tbits != 0xFFFFFFFFu && tbits == tmin;
        const uint32_t first = quad_min(candidate ? refpos : 0xFFFFFFFFu);  // among equal distances: the one the reference meets first
        if (candidate && refpos == first) {
            lds.res[e] = F4(__uint_as_float(tbits), __uint_as_float(tri | (det > 0.0f ? 0x80000000u : 0u)), b1, b2);
            *lds.ref(e) = refpos;
        }
    }
    rz_wave_sync();
    if (holding) {
        f4 best = lds.res[deal.pos];
        uint32_t best_ref = *lds.ref(deal.pos);
        if (deal.big) {
            const f4 other = lds.res[deal.pos + 1u];
            const uint32_t other_ref = *lds.ref(deal.pos + 1u);
            if (__float_as_uint(other.x) < __float_as_uint(best.x) || (__float_as_uint(other.x) == __float_as_uint(best.x) && __float_as_uint(other.x) != 0xFFFFFFFFu && other_ref < best_ref))
                best = other, best_ref = other_ref;
        }
        if (__float_as_uint(best.x) != 0xFFFFFFFFu) {
            far_ = best.x;
            held_ref = best_ref;
            hit.triangle = __float_as_uint(best.y) & 0x7FFFFFFFu;
            hit.external = (__float_as_uint(best.y) & 0x80000000u) != 0u;
            hit.bx = best.z, hit.by = best.w;
            found = true;
        }
    }
    rz_wave_sync();  // the next phase rewrites rec / res
}

// ONE_STEP: the world level as one node step per round and nothing else (what a world that is one leaf needs; the register budget of the
// 5-wave trace kernel is tight enough that the general form's extra live values cost config D 3.5 %: 799 -> 827 us)
template <bool COUNT, bool RCP, bool ONE_STEP = false>
RZ_DEV int closest_hit_coop(const DScene& s, const CoopLds& lds, bool active, Ray& ray, Hit& hit, Counters& cnt) {
    const bool scene_fast = s.fast_div != 0u;
    WalkRay g;
    g.o = ray.o, g.d = ray.d, g.near_ = ray.near_, g.far_ = ray.far_;
    prepare<RCP>(g, scene_fast);
    const uint32_t kmax = s.walk_k ? s.walk_k : 0xFFFFFFFFu, lmax = s.walk_l ? (s.walk_l < 8u ? s.walk_l : 8u) : 8u, hmin = s.walk_h;
    uint32_t n = active ? s.tlas_root : RZ_END, guard = 0u;
    bool root_missed = false;
    while (__any(n != RZ_END)) {
        RZ_GUARD(guard);
        // world level, in the reference's order: a lane steps through the nodes of the world tree until it HOLDS a leaf with instances
        // (at most 1 + world_advance steps per round; 0: one step, then the wave turns to the lanes that hold a leaf), `after` = where
        // it goes on behind that leaf
        uint32_t i = 0u, end = 0u, after = RZ_END;
        bool descended = false;  // (ONE_STEP)
        if constexpr (ONE_STEP) {
            if (n != RZ_END) {
                float4 n0, n1;
                fetch_node_ordered(s, n, 0u, n0, n1, after);
                RZ_PHASE(0);
                RZ_COUNT(box_tests);
                if (box_hit_unpacked<RCP>(n0, n1, g)) {
                    const uint32_t begin = __float_as_uint(n1.z), meta = __float_as_uint(n1.w);
                    if (!(meta & HIPRZ_NODE_LEAF)) n = begin, descended = true;
                    else i = begin, end = begin + (meta & HIPRZ_NODE_COUNT_MASK);
                } else if (n == s.tlas_root) {
                    root_missed = true, after = RZ_END;  // root box missed (cpu_engine_kernel.cpp:283): this lane's walk is over
                }
            }
        } else {
            for (uint32_t r = 0u;; ++r) {
                if (n != RZ_END && i == end) {
                    float4 n0, n1;
                    uint32_t link;
                    fetch_node_ordered(s, n, 0u, n0, n1, link);  // the world tree keeps the reference's order
                    RZ_PHASE(0);
                    RZ_COUNT(box_tests);
                    if (box_hit_unpacked<RCP>(n0, n1, g)) {
                        const uint32_t begin = __float_as_uint(n1.z), meta = __float_as_uint(n1.w);
                        if (!(meta & HIPRZ_NODE_LEAF)) n = begin;
                        else {
                            i = begin, end = begin + (meta & HIPRZ_NODE_COUNT_MASK), after = link;
                            if (i == end) n = link;  // (an empty leaf)
                        }
                    } else {
                        if (n == s.tlas_root) root_missed = true, link = RZ_END;  // root box missed (cpu_engine_kernel.cpp:283): this lane's walk is over
                        n = link;
                    }
                }
                if (r >= s.world_advance || !__any(n != RZ_END && i == end)) break;
            }
        }
        const bool held = !ONE_STEP && i < end;
        while (__any(i < end)) {
            bool enter = false;
            uint32_t inst = 0u;
            // a lane tests the boxes of its leaf's instances, in the reference's order, until it meets one it enters — at most
            // 1 + walk_advance of them before the wave goes on (0: one box per lane and round, every lane at the same instance)
            for (uint32_t r = 0u;; ++r) {
                if (!enter && i < end) {
                    const uint32_t candidate = s.tlas_order[i];
                    float4 ib0, ib1;
                    load_instance_box(s, candidate, ib0, ib1);
                    RZ_PHASE(1);
                    RZ_COUNT(box_tests);
                    if (box_hit_unpacked<RCP>(ib0, ib1, g)) enter = true, inst = candidate;
                    i += 1u;
                }
                if (r >= s.walk_advance || !__any(!enter && i < end)) break;
            }
            if (!__any(enter)) continue;
            WalkRay lr;
            lr.o = lr.d = lr.y = V3(0.0f, 0.0f, 0.0f), lr.near_ = lr.far_ = 0.0f, lr.fast = true;
            float len = 1.0f;
            bool found = false;
            uint32_t held_ref = 0u;  // reference position of the held hit of THIS mesh
            uint32_t m = RZ_END, oct = 0u;
            if (enter) {
                RZ_PHASE(2);
                const InstanceXform x = load_instance_xform(s, inst);
                len = to_local<RCP>(x, g, lr, scene_fast);
                m = x.blas_root;
                oct = octant_of(lr.d);
            }
            uint32_t tj = 0u, tj_end = 0u;  // the held leaf's remaining triangles
            while (__any(tj != tj_end || m != RZ_END)) {
                RZ_PHASE(5);
                // node phase: lanes without a leaf step, the others wait — until a full triangle step's worth of lanes hold a leaf
                for (uint32_t k = 0u; k < kmax && __any(tj == tj_end && m != RZ_END); ++k) {
                    if (k != 0u && uint32_t(__popcll(__ballot(tj != tj_end))) >= hmin) break;
                    if (tj == tj_end && m != RZ_END) {
                        RZ_GUARD(guard);
                        float4 m0, m1;
                        uint32_t mlink;
                        fetch_node_ordered(s, m, oct, m0, m1, mlink);
                        RZ_PHASE(3);
                        RZ_COUNT(box_tests);
                        const uint32_t mbegin = __float_as_uint(m1.z), mmeta = __float_as_uint(m1.w);
                        const bool mleaf = (mmeta & HIPRZ_NODE_LEAF) != 0u;
                        const uint32_t near_child = mbegin + ((oct >> (mmeta >> HIPRZ_NODE_PTYPE_SHIFT)) & 1u);  // the nearer child is entered first
                        if (box_hit_filtered<RCP>(m0, m1, lr)) {
                            if (!mleaf) mlink = near_child;
                            else tj = mbegin, tj_end = mbegin + (mmeta & HIPRZ_NODE_COUNT_MASK);
                        }
                        m = mlink;
                    }
                }
                if (!__any(tj != tj_end)) continue;
                // triangle phase, all 64 lanes
                const uint32_t c = tj_end - tj < lmax ? tj_end - tj : lmax;
                coop_closest_triangles<COUNT>(s, lds, c != 0u, c, lr.o, lr.d, lr.near_, tj, found, held_ref, hit, lr.far_, cnt);
                tj += c;
            }
            if (found) {
                hit.instance = int32_t(inst);
                g.near_ = lr.near_ / len;
                g.far_ = lr.far_ / len;
            }
        }
        if constexpr (ONE_STEP) {
            if (n != RZ_END && !descended) n = after;
        } else {
            if (held) n = after;
        }
    }
    ray.near_ = g.near_, ray.far_ = g.far_;
    if (!active || root_missed) return 0;
    return hit.instance >= 0 ? 2 : 1;
}

// anyIntersection (cpu_engine_kernel.cpp:398-481) on skip links, with the tree tops from LDS: the shadow-ray walk of the
// kernels that have a TopCache.  Same tests in the same order as any_hit_stack; returns the mask's alpha (0 or 1).
template <bool COUNT, bool RCP>
RZ_DEV float any_hit_skip(const DScene& s, const TopCache& top, const Ray& ray, Counters& cnt) {
    const bool scene_fast = s.fast_div != 0u;
    WalkRay g;
    g.o = ray.o, g.d = ray.d, g.near_ = ray.near_, g.far_ = ray.far_;
    prepare<RCP>(g, scene_fast);
    uint32_t n = s.tlas_root, guard = 0u;
    while (n != RZ_END) {
        RZ_GUARD(guard);
        float4 n0, n1;
        uint32_t link;
        fetch_node(s, top, n, n0, n1, link);
        RZ_COUNT(box_tests);
        RZ_COUNT(shadow_box_tests);
        if (box_hit<RCP>(n0, n1, g)) {
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
                if (!box_hit<RCP>(ib0, ib1, g)) continue;
                const InstanceXform x = load_instance_xform(s, inst);
                WalkRay lr;
                to_local<RCP>(x, g, lr, scene_fast);
                uint32_t m = x.blas_root;
                // anyIntersection(const Mesh&, ...) :450-481, in the bounded while-while rounds of closest_hit_skip
                uint32_t tj = 0u, tj_end = 0u;
                const uint32_t kmax = s.walk_k ? s.walk_k : 0xFFFFFFFFu, lmax = s.walk_l ? s.walk_l : 0xFFFFFFFFu;
                while (true) {
                    uint32_t k = 0u;
                    while (tj == tj_end && m != RZ_END && k < kmax) {
                        RZ_GUARD(guard);
                        k += 1u;
                        float4 m0, m1;
                        uint32_t mlink;
                        fetch_node(s, top, m, m0, m1, mlink);
                        RZ_COUNT(box_tests);
                        RZ_COUNT(shadow_box_tests);
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
                    if (tj == tj_end && m == RZ_END) break;
                    uint32_t l = 0u;
                    for (; tj < tj_end && l < lmax; ++tj, ++l) {
                        const float4 a = s.tris[3 * tj], b = s.tris[3 * tj + 1], c = s.tris[3 * tj + 2];
                        float t, b1, b2, det;
                        RZ_COUNT(tri_tests);
                        RZ_COUNT(shadow_tri_tests);
                        if (tri_hit(xyz(a), xyz(b), xyz(c), lr, t, b1, b2, det)) return 0.0f;
                    }
                }
            }
        } else if (n == s.tlas_root) {
            return 1.0f;  // root box missed (:402)
        }
        n = link;
    }
    return 1.0f;
}

// anyIntersection with the cooperative triangle phase of closest_hit_coop: all 64 lanes walk together (`active` = has a shadow
// ray), the triangles of the lanes that hold a leaf are dealt out over the wave in groups of 8, and a holder is occluded as soon as
// ANY of its items hits (a ballot per group instead of the minimum).  A shadow ray's range is fixed, so the answer does not depend
// on the order of the tests.
// MASK (HIPRZ_COMPAT_SHADOW_COLOR, cuda_instance.cuh:92-164): the ray goes THROUGH the triangles it crosses and the result is the product
// of their opacity colours (white where nothing is crossed); a tester that hits fetches its triangle's colour (hiprz_compat.hpp:
// compat_crossing_color), the four testers of an entry multiply theirs over the quad (two DPP exchanges, a fixed order), the holder
// multiplies its mask by its entries' products and stops once the mask's alpha is below 1e-4, as the CUDA engine's walk does.  The order of
// the factors is this walk's (front to back, entries in leaf order), not the inline walk's: products agree to rounding, not to the bit.
template <bool COUNT>
RZ_DEV col4 compat_crossing_color(const DScene& s, uint32_t inst, uint32_t tri, uint32_t tri_flags, float b1, float b2, bool filtering, Counters& cnt);  // hiprz_compat.hpp
RZ_DEV float quad_product(float v) {
    float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0xB1, 0xF, 0xF, false));  // quad_perm:[1,0,3,2]
    v = v * o;
    o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x4E, 0xF, 0xF, false));        // quad_perm:[2,3,0,1]
    return v * o;
}
template <bool COUNT, bool RCP, bool MASK = false>
RZ_DEV col4 any_hit_coop_mask(const DScene& s, const CoopLds& lds, bool active, const Ray& ray, bool filtering, Counters& cnt) {
    const bool scene_fast = s.fast_div != 0u;
    const uint32_t lane = lane_id();
    WalkRay g;
    g.o = ray.o, g.d = ray.d, g.near_ = ray.near_, g.far_ = ray.far_;
    prepare<RCP>(g, scene_fast);
    const uint32_t kmax = s.walk_k ? s.walk_k : 0xFFFFFFFFu, lmax = s.walk_l ? (s.walk_l < 8u ? s.walk_l : 8u) : 8u, hmin = s.walk_h;
    uint32_t n = active ? s.tlas_root : RZ_END, guard = 0u;
    bool occluded = false;
    col4 shadow = splat(1.0f);  // (MASK)
    while (__any(n != RZ_END)) {
        RZ_GUARD(guard);
        uint32_t i = 0u, end = 0u, after = RZ_END;  // (the world level as in closest_hit_coop)
        for (uint32_t r = 0u;; ++r) {
            if (n != RZ_END && i == end) {
                float4 n0, n1;
                uint32_t link;
                fetch_node_ordered(s, n, 0u, n0, n1, link);
                RZ_COUNT(box_tests);
                RZ_COUNT(shadow_box_tests);
                if (box_hit_unpacked<RCP>(n0, n1, g)) {
                    const uint32_t begin = __float_as_uint(n1.z), meta = __float_as_uint(n1.w);
                    if (!(meta & HIPRZ_NODE_LEAF)) n = begin;
                    else {
                        i = begin, end = begin + (meta & HIPRZ_NODE_COUNT_MASK), after = link;
                        if (i == end) n = link;
                    }
                } else {
                    n = n == s.tlas_root ? RZ_END : link;  // root box missed (:402): clear
                }
            }
            if (r >= s.world_advance || !__any(n != RZ_END && i == end)) break;
        }
        bool held = i < end;
        while (__any(i < end)) {
            bool enter = false;
            uint32_t inst = 0u;
            for (uint32_t r = 0u;; ++r) {  // (as in closest_hit_coop)
                if (!enter && i < end) {
                    const uint32_t candidate = s.tlas_order[i];
                    float4 ib0, ib1;
                    load_instance_box(s, candidate, ib0, ib1);
                    RZ_COUNT(box_tests);
                    RZ_COUNT(shadow_box_tests);
                    if (box_hit_unpacked<RCP>(ib0, ib1, g)) enter = true, inst = candidate;
                    i += 1u;
                }
                if (r >= s.walk_advance || !__any(!enter && i < end)) break;
            }
            if (!__any(enter)) continue;
            WalkRay lr;
            lr.o = lr.d = lr.y = V3(0.0f, 0.0f, 0.0f), lr.near_ = lr.far_ = 0.0f, lr.fast = true;
            uint32_t m = RZ_END, oct = 0u;
            if (enter) {
                const InstanceXform x = load_instance_xform(s, inst);
                to_local<RCP>(x, g, lr, scene_fast);
                m = x.blas_root;
                oct = octant_of(lr.d);
            }
            uint32_t tj = 0u, tj_end = 0u;
            while (__any(tj != tj_end || m != RZ_END)) {
                for (uint32_t k = 0u; k < kmax && __any(tj == tj_end && m != RZ_END); ++k) {
                    if (k != 0u && uint32_t(__popcll(__ballot(tj != tj_end))) >= hmin) break;
                    if (tj == tj_end && m != RZ_END) {
                        RZ_GUARD(guard);
                        float4 m0, m1;
                        uint32_t mlink;
                        fetch_node_ordered(s, m, oct, m0, m1, mlink);
                        RZ_COUNT(box_tests);
                        RZ_COUNT(shadow_box_tests);
                        const uint32_t mbegin = __float_as_uint(m1.z), mmeta = __float_as_uint(m1.w);
                        const bool mleaf = (mmeta & HIPRZ_NODE_LEAF) != 0u;
                        const uint32_t near_child = mbegin + ((oct >> (mmeta >> HIPRZ_NODE_PTYPE_SHIFT)) & 1u);
                        if (box_hit_filtered<RCP>(m0, m1, lr)) {
                            if (!mleaf) mlink = near_child;
                            else tj = mbegin, tj_end = mbegin + (mmeta & HIPRZ_NODE_COUNT_MASK);
                        }
                        m = mlink;
                    }
                }
                if (!__any(tj != tj_end)) continue;
                const uint32_t c = tj_end - tj < lmax ? tj_end - tj : lmax;
                const bool holding = c != 0u;
                const CoopDeal deal = coop_deal(holding, c);
                if (holding) {
                    const f4 r0 = F4(lr.o.x, lr.o.y, lr.o.z, lr.near_), r1 = F4(lr.d.x, lr.d.y, lr.d.z, lr.far_);
                    lds.rec[deal.pos] = r0, lds.rec[CoopLds::kEntries + deal.pos] = r1;
                    lds.rec[2u * CoopLds::kEntries + deal.pos] = F4(__uint_as_float(tj), __uint_as_float(c < 4u ? c : 4u), __uint_as_float(inst), 0.0f);
                    lds.res[deal.pos] = MASK ? F4(1.0f, 1.0f, 1.0f, 1.0f) : F4(0.0f, 0.0f, 0.0f, 0.0f);  // .x != 0: one of the entry's triangles was hit (MASK: the entry's product)
                    if (deal.big) {
                        lds.rec[deal.pos + 1u] = r0, lds.rec[CoopLds::kEntries + deal.pos + 1u] = r1;
                        lds.rec[2u * CoopLds::kEntries + deal.pos + 1u] = F4(__uint_as_float(tj + 4u), __uint_as_float(c - 4u), __uint_as_float(inst), 0.0f);
                        lds.res[deal.pos + 1u] = MASK ? F4(1.0f, 1.0f, 1.0f, 1.0f) : F4(0.0f, 0.0f, 0.0f, 0.0f);
                    }
                }
                rz_wave_sync();
                for (uint32_t base = 0u; base < deal.n_entries * 4u; base += 64u) {
                    const uint32_t e = (base + lane) >> 2, j = lane & 3u;
                    bool item_hit = false;
                    col4 crossing = splat(1.0f);  // (MASK) this tester's factor
                    if (e < deal.n_entries) {
                        const f4 r2 = lds.rec[2u * CoopLds::kEntries + e];
                        if (j < __float_as_uint(r2.y)) {
                            const f4 r0 = lds.rec[e], r1 = lds.rec[CoopLds::kEntries + e];
                            WalkRay hr;
                            hr.o = V3(r0.x, r0.y, r0.z), hr.d = V3(r1.x, r1.y, r1.z), hr.near_ = r0.w, hr.far_ = r1.w;
                            const uint32_t tri = __float_as_uint(r2.x) + j;
                            const float4 a = s.tris[3 * tri], b = s.tris[3 * tri + 1], cc = s.tris[3 * tri + 2];
                            float t, b1, b2, det;
                            RZ_COUNT(tri_tests);
                            RZ_COUNT(shadow_tri_tests);
                            item_hit = tri_hit(xyz(a), xyz(b), xyz(cc), hr, t, b1, b2, det);
                            if constexpr (MASK) {
                                if (item_hit) crossing = compat_crossing_color<COUNT>(s, __float_as_uint(r2.z), tri, __float_as_uint(a.w), b1, b2, filtering, cnt);
                            }
                        }
                    }
                    if constexpr (MASK) {  // the entry's product, in every lane of its quad; lane 0 of the quad hands it over
                        crossing = col4{quad_product(crossing.r), quad_product(crossing.g), quad_product(crossing.b), quad_product(crossing.a)};
                        if (e < deal.n_entries && j == 0u) lds.res[e] = F4(crossing.r, crossing.g, crossing.b, crossing.a);
                    } else {
                        if (item_hit) lds.res[e] = F4(1.0f, 0.0f, 0.0f, 0.0f);  // every writer of a slot writes the same value
                    }
                }
                rz_wave_sync();
                if (holding) {
                    tj += c;
                    bool done;
                    if constexpr (MASK) {
                        const f4 p0 = lds.res[deal.pos];
                        shadow = shadow * col4{p0.x, p0.y, p0.z, p0.w};
                        if (deal.big) {
                            const f4 p1 = lds.res[deal.pos + 1u];
                            shadow = shadow * col4{p1.x, p1.y, p1.z, p1.w};
                        }
                        done = shadow.a < 1.0e-4f;  // nothing gets through any more (cuda_instance.cuh: the walk returns)
                    } else {
                        done = lds.res[deal.pos].x != 0.0f || (deal.big && lds.res[deal.pos + 1u].x != 0.0f);  // occluded (:465 "TODO: texture fetch" -> mask 0)
                    }
                    if (done) {  // this lane's walk is over
                        occluded = true;
                        tj = tj_end = 0u, m = RZ_END, i = end = 0u, n = RZ_END, held = false;
                    }
                }
                rz_wave_sync();
            }
        }
        if (held) n = after;
    }
    if constexpr (MASK) return shadow;
    return splat(occluded ? 0.0f : 1.0f);
}
template <bool COUNT, bool RCP>
RZ_DEV float any_hit_coop(const DScene& s, const CoopLds& lds, bool active, const Ray& ray, Counters& cnt) {
    return any_hit_coop_mask<COUNT, RCP, false>(s, lds, active, ray, false, cnt).a;
}

// anyIntersection for a wave whose 64 shadow rays form a BEAM (the shadow rays' sorted order: one origin cell, one light): the WAVE walks the
// trees, not the lanes.  One node at a time, in the trees' own order on the skip links, fetched once for all lanes (a uniform address: scalar
// loads, the box in scalar registers); every lane that is still undecided and has passed all of the node's ancestors tests the box
// against its own ray; the wave enters the node if any lane does, and a lane that misses it sits out until the walk reaches the node's skip
// link — the next node outside the subtree — so a lane tests exactly the boxes and triangles the per-lane walk in that order tests for its
// ray: the same verdicts (a shadow ray's range is fixed, the answer does not depend on the order or on other rays).  No stack, no
// divergence: loops and branches are wave-uniform, lanes only differ in predicates; 2 KiB of LDS per wave (`rays`: [2][64] f4; MASK: 3 KiB) for the triangle phase.  What it costs is the UNION of the lanes' walks —
// close to one lane's walk for a beam, far more for rays that have nothing in common.
// MASK (HIPRZ_COMPAT_SHADOW_COLOR): the ray goes THROUGH the triangles it crosses and the result is the product of their opacity colours, as in
// any_hit_coop_mask; a tester that hits leaves its triangle's colour in LDS (`rays` + 128: one f4 per pair of a step), the ray's lane multiplies its
// pairs' colours in leaf order and stops below alpha 1e-4.  The factors are those of the other walks, their order is this walk's: products agree
// to rounding.
template <bool COUNT, bool RCP, bool MASK = false>
RZ_DEV col4 any_hit_packet(const DScene& s, RZ_LDS f4* rays, bool active, const Ray& ray, bool filtering, Counters& cnt) {
    const bool scene_fast = s.fast_div != 0u;
    WalkRay g;
    g.o = ray.o, g.d = ray.d, g.near_ = ray.near_, g.far_ = ray.far_;
    prepare<RCP>(g, scene_fast);
    bool live = active, occluded = false, wblocked = false;
    col4 shadow = splat(1.0f);  // (MASK)
    uint32_t wresume = 0u, guard = 0u;
    if (active) { RZ_COUNT(shadow_rays); }
    // The world level: a shadow ray's answer does not depend on the order in which it meets the instances, so where the context has
    // built one the walk takes the shadow rays' OWN world tree — a surface-area tree over the instances' boxes with one instance per leaf,
    // whose leaf box IS the instance's box (the leaf's test is the instance's test) — instead of the reference's.
    const bool own = s.shadow_root != RZ_END;
    const float4* world = own ? s.shadow_nodes64 : s.nodes64;
    const uint32_t* world_order = own ? s.shadow_order : s.tlas_order;
    uint32_t n = own ? s.shadow_root : s.tlas_root;
    while (n != RZ_END && __any(live)) {  // wave-uniform
        RZ_GUARD(guard);
        if (wblocked && wresume == n) wblocked = false;
        const float4* rec = world + 4 * size_t(n);
        const float4 n0 = rec[0], n1 = rec[1];
        const uint32_t link = uint32_t(__builtin_amdgcn_readfirstlane(int(reinterpret_cast<const uint32_t*>(rec + 2)[0])));
        bool hit = false;
        if (live && !wblocked) {
            RZ_PHASE(0);
            RZ_COUNT(box_tests);
            RZ_COUNT(shadow_box_tests);
            hit = box_hit_unpacked<RCP>(n0, n1, g);
            if (!hit) wblocked = true, wresume = link;  // (the root's link is RZ_END: a ray that misses the root box is clear, :402)
        }
        if (!__any(hit)) {
            n = link;
            continue;
        }
        const uint32_t begin = uint32_t(__builtin_amdgcn_readfirstlane(int(__float_as_uint(n1.z)))), meta = uint32_t(__builtin_amdgcn_readfirstlane(int(__float_as_uint(n1.w))));
        if (!(meta & HIPRZ_NODE_LEAF)) {
            n = begin;
            continue;
        }
        const uint32_t end = begin + (meta & HIPRZ_NODE_COUNT_MASK);
        for (uint32_t i = begin; i < end && __any(live); ++i) {  // the leaf's instances, in the reference's order
            const uint32_t inst = uint32_t(__builtin_amdgcn_readfirstlane(int(world_order[i])));
            bool enter = live && hit;  // (own tree: the leaf's box was the instance's)
            if (!own && enter) {
                float4 ib0, ib1;
                load_instance_box(s, inst, ib0, ib1);
                RZ_PHASE(1);
                RZ_COUNT(box_tests);
                RZ_COUNT(shadow_box_tests);
                enter = box_hit_unpacked<RCP>(ib0, ib1, g);
            }
            if (!__any(enter)) continue;
            const InstanceXform x = load_instance_xform(s, inst);
            WalkRay lr;
            lr.o = lr.d = lr.y = V3(0.0f, 0.0f, 0.0f), lr.near_ = lr.far_ = 0.0f, lr.fast = true;
            if (enter) {
                RZ_PHASE(2);
                to_local<RCP>(x, g, lr, scene_fast);
            }
            uint32_t m = uint32_t(__builtin_amdgcn_readfirstlane(int(x.blas_root))), mresume = 0u;
            bool mblocked = false;
            while (m != RZ_END && __any(enter && live)) {  // wave-uniform
                RZ_GUARD(guard);
                if (mblocked && mresume == m) mblocked = false;
                float4 m0, m1;
                uint32_t mlink;
                fetch_node_ordered(s, m, 0u, m0, m1, mlink);
                mlink = uint32_t(__builtin_amdgcn_readfirstlane(int(mlink)));
                bool mhit = false;
                if (enter && live && !mblocked) {
                    RZ_PHASE(3);
                    RZ_COUNT(box_tests);
                    RZ_COUNT(shadow_box_tests);
                    mhit = box_hit_filtered<RCP>(m0, m1, lr);
                    if (!mhit) mblocked = true, mresume = mlink;
                }
                if (!__any(mhit)) {
                    m = mlink;
                    continue;
                }
                const uint32_t mbegin = uint32_t(__builtin_amdgcn_readfirstlane(int(__float_as_uint(m1.z)))), mmeta = uint32_t(__builtin_amdgcn_readfirstlane(int(__float_as_uint(m1.w))));
                if (!(mmeta & HIPRZ_NODE_LEAF)) {
                    m = mbegin;  // (the order of octant 0: the lower child first)
                    continue;
                }
                // The leaf's triangles: (ray, triangle) pairs dealt over all 64 lanes.  The lanes that passed the leaf's box put their mesh-space
                // rays into LDS by rank; lane q of a step tests triangle q mod c for the ray of rank q div c; a ballot brings the verdicts
                // back, and a ray is occluded if any of its c testers hit.  (One triangle at a time for the lanes of the leaf ran at 15 lanes.)
                const uint32_t c = mmeta & HIPRZ_NODE_COUNT_MASK;
                const bool owner = mhit && live;
                const unsigned long long owners = __ballot(owner);
                const uint32_t pairs = uint32_t(__popcll(owners)) * c, r = rank_in(owners);
                if (owner) {
                    RZ_PHASE(4);
                    rays[r] = F4(lr.o.x, lr.o.y, lr.o.z, lr.near_), rays[64u + r] = F4(lr.d.x, lr.d.y, lr.d.z, lr.far_);
                }
                rz_wave_sync();
                const float rcp_c = __builtin_amdgcn_rcpf(float(c));
                const uint32_t lane = lane_id();
                for (uint32_t base = 0u; base < pairs; base += 64u) {  // wave-uniform
                    const uint32_t q = base + lane;
                    bool pair_hit = false;
                    if (q < pairs) {
                        // q div c, q mod c: by reciprocal where that is exact beyond doubt ((q + 0.5) / c is at least 0.5 / c away from an integer)
                        const uint32_t of = c <= 256u ? uint32_t((float(q) + 0.5f) * rcp_c) : q / c, tri = mbegin + (q - of * c);
                        const f4 r0 = rays[of], r1 = rays[64u + of];
                        WalkRay hr;
                        hr.o = V3(r0.x, r0.y, r0.z), hr.d = V3(r1.x, r1.y, r1.z), hr.near_ = r0.w, hr.far_ = r1.w;
                        const float4 ta = s.tris[3 * tri], tb = s.tris[3 * tri + 1], tc = s.tris[3 * tri + 2];
                        float t, b1, b2, det;
                        RZ_COUNT(tri_tests);
                        RZ_COUNT(shadow_tri_tests);
                        pair_hit = tri_hit(xyz(ta), xyz(tb), xyz(tc), hr, t, b1, b2, det);
                        if constexpr (MASK) {
                            if (pair_hit) {
                                const col4 crossing = compat_crossing_color<COUNT>(s, inst, tri, __float_as_uint(ta.w), b1, b2, filtering, cnt);
                                rays[128u + lane] = F4(crossing.r, crossing.g, crossing.b, crossing.a);
                            }
                        }
                    }
                    if constexpr (MASK) rz_wave_sync();
                    const unsigned long long verdicts = __ballot(pair_hit);
                    if (owner) {  // my testers are pairs r * c .. r * c + c - 1; those of this step sit in bits first - base .. of `verdicts`
                        const uint32_t first = r * c, last = first + c;
                        const uint32_t lo = first > base ? first - base : 0u, hi = last < base + 64u ? (last > base ? last - base : 0u) : 64u;
                        if constexpr (MASK) {
                            for (uint32_t k = lo; k < hi; ++k)
                                if ((verdicts >> k) & 1ull) {
                                    const f4 cr = rays[128u + k];
                                    shadow = shadow * col4{cr.x, cr.y, cr.z, cr.w};
                                }
                            if (shadow.a < 1.0e-4f) live = false;  // nothing gets through any more (cuda_instance.cuh: the walk returns)
                        } else {
                            if (lo < hi && ((verdicts >> lo) & (hi - lo >= 64u ? ~0ull : ((1ull << (hi - lo)) - 1ull))) != 0ull) live = false, occluded = true;  // (:465: any hit occludes)
                        }
                    }
                    if constexpr (MASK) rz_wave_sync();  // the next step rewrites the colours
                }
                rz_wave_sync();
                m = mlink;
            }
        }
        n = link;
    }
    if constexpr (MASK) return shadow;
    return splat(occluded ? 0.0f : 1.0f);
}

template <int MODE, bool COUNT, bool RCP>
RZ_DEV int closest_hit(const DScene& s, uint32_t* lds_column, Ray& ray, Hit& hit, Counters& cnt) {
    hit.instance = -1;
    hit.triangle = 0;
    hit.bx = hit.by = 0.0f;
    hit.external = true;
    if (s.n_instances == 0) return 0;
    return closest_hit_stack<COUNT, RCP>(s, lds_column, ray, hit, cnt);  // MODE 2 calls closest_hit_binned directly
}
// what a shadow-ray walk needs besides the scene: the lane's LDS stack column (MODE 1) or the staged tree tops (MODE 3)
// MODE 4 ("defer"): no walk here — the sample's shadow ray and its unshadowed radiance term are written out for
// rz_shadow_kernel, which walks the rays of all pixels in a lean kernel of its own and finishes the sums in this order.
#define RZ_SHADOW_DEFER 4
// MODE 5 ("none"): instantiation for scenes WITHOUT lights — directIllumination returns 0 there before it evaluates anything
// (cpu_engine_kernel.cpp:703, :758), so the whole next-event-estimation code (and its registers) is compiled out.
#define RZ_SHADOW_NONE 5
// MODE 6 ("plain"): no lights AND no maps of any kind in the scene — texture fetches, normal mapping and the sky's texture
// coordinates are compiled out as well.
#define RZ_SHADOW_PLAIN 6
// MODE 8 ("compat"): the CUDA engine's behaviours selected by DConfig::flags (hiprz_compat.hpp): shadow rays inline on skip links,
// opaque as in the CPU kernel or — HIPRZ_COMPAT_SHADOW_COLOR — through triangles with a coloured mask.
#define RZ_SHADOW_COMPAT 8
// MODE 9: the compat shading with the shadow rays DEFERRED like MODE 4 (the opaque shadow rays of the CPU kernel; the coloured mask of
// HIPRZ_COMPAT_SHADOW_COLOR needs a texture fetch per crossed triangle and stays inline, MODE 8)
#define RZ_SHADOW_COMPAT_DEFER 9
constexpr bool shadow_mode_defers(int mode) { return mode == RZ_SHADOW_DEFER || mode == RZ_SHADOW_COMPAT_DEFER; }
constexpr bool shadow_mode_compat(int mode) { return mode == RZ_SHADOW_COMPAT || mode == RZ_SHADOW_COMPAT_DEFER; }
template <bool COUNT>
RZ_DEV col4 compat_shadow_mask(const DScene& s, const Ray& ray, bool filtering, Counters& cnt);  // hiprz_compat.hpp
template <bool COUNT>
RZ_DEV col4 compat_fetch(const DScene& s, int32_t tex, float u, float v, Counters& cnt);            // hiprz_compat.hpp
struct ShadowCtx {
    uint32_t* lds_column;
    TopCache top;
    // deferred shadow rays (split pipeline, scenes with lights that are not staged in LDS)
    float4* nee = nullptr;  // this pixel's hand-over record (DFrame::nee)
    mutable uint32_t defer_mask = 0u;  // bit k: sample slot k holds a shadow ray
    mutable float key_dir[3] = {0.0f, 0.0f, 0.0f};  // direction of the shadow ray in the highest slot (spot samples come last) and the
    mutable float key_o[3] = {0.0f, 0.0f, 0.0f};    // rays' common origin: what the pixel's shadow sort key is made of
    mutable uint32_t key_light = 0u;                // ... and the light that ray goes to (spot light i: i, direct light i: 4 + i)
    mutable bool defer_done = false;   // the segment went through directIllumination
    mutable col4 defer_a{0.0f, 0.0f, 0.0f, 0.0f}, defer_b{0.0f, 0.0f, 0.0f, 0.0f};  // final += (direct * a) * b
};
#ifndef RZ_SHADE_SHARED_RCP   // packed shared-reciprocal box test in the shade kernel's shadow-ray walk (MODE 3)
#define RZ_SHADE_SHARED_RCP 1
#endif
// anyIntersection(const RangedRay&): returns the shadow mask's alpha (0 or 1)
template <int MODE, bool COUNT>
RZ_DEV float any_hit(const DScene& s, const ShadowCtx& sc, const Ray& ray, Counters& cnt) {
    uint32_t* lds_column = sc.lds_column;
    RZ_COUNT(shadow_rays);
    if (s.n_instances == 0) return 0.0f;
    if constexpr (MODE == 3) {
        return any_hit_skip<COUNT, RZ_SHADE_SHARED_RCP != 0>(s, sc.top, ray, cnt);
    } else {
        return any_hit_stack<COUNT>(s, lds_column, ray, cnt);
    }
}

// --- materials and textures -------------------------------------------------------------
struct Material {
    uint32_t color;
    float metalness, roughness, emission, ior, scattering;
    int32_t texture, normal_map, metalness_map, roughness_map, emission_map;
};
RZ_DEV Material load_material(const DScene& s, uint32_t i) {
    const float4 a = s.materials[3 * i], b = s.materials[3 * i + 1], c = s.materials[3 * i + 2];
    Material m;
    m.color = __float_as_uint(a.x), m.metalness = a.y, m.roughness = a.z, m.emission = a.w;
    m.ior = b.x, m.scattering = b.y, m.texture = __float_as_int(b.z), m.normal_map = __float_as_int(b.w);
    m.metalness_map = __float_as_int(c.x), m.roughness_map = __float_as_int(c.y), m.emission_map = __float_as_int(c.z);
    return m;
}
RZ_DEV float material_ior(const DScene& s, uint32_t i) { return s.materials[3 * i + 1].x; }
RZ_DEV float material_scattering(const DScene& s, uint32_t i) { return s.materials[3 * i + 1].y; }

// TextureBuffer::fetch: render_parts.hpp:209-221 — returns the byte offset of the texel
template <bool COUNT>
RZ_DEV size_t texel_offset(const DScene& s, int32_t tex, float u, float v, uint32_t texel_size, Counters& cnt) {
    const float4 a = s.textures[3 * tex], b = s.textures[3 * tex + 1], c = s.textures[3 * tex + 2];
    const uint32_t width = __float_as_uint(a.y), height = __float_as_uint(a.z), offset = __float_as_uint(a.w);
    u += b.z;  // translation
    v += b.w;
    {
        const float xx = u * c.y - v * c.z;  // cos, sin of the rotation (hoisted)
        const float yy = u * c.z + v * c.y;
        u = xx, v = yy;
    }
    u *= b.x;  // scale
    v *= b.y;
    u = fmodf(fmodf(u, 1.0f) + 1.0f, 1.0f);
    v = 1.0f - fmodf(fmodf(v, 1.0f) + 1.0f, 1.0f);
    uint32_t x = uint32_t(u * float(width));
    uint32_t y = uint32_t(v * float(height));
    if (x > width - 1u) x = width - 1u;
    if (y > height - 1u) y = height - 1u;
    RZ_COUNT(texel_fetches);
    return size_t(offset) + size_t(texel_size) * (size_t(y) * width + x);
}
template <bool COUNT>
RZ_DEV col4 fetch_rgba8(const DScene& s, int32_t tex, float u, float v, Counters& cnt) {
    return from_u8(*reinterpret_cast<const uint32_t*>(s.texels + texel_offset<COUNT>(s, tex, u, v, 4u, cnt)));
}
template <bool COUNT>
RZ_DEV float fetch_r8(const DScene& s, int32_t tex, float u, float v, Counters& cnt) {
    return float(s.texels[texel_offset<COUNT>(s, tex, u, v, 1u, cnt)]) / 255.0f;
}
template <bool COUNT>
RZ_DEV float fetch_r32f(const DScene& s, int32_t tex, float u, float v, Counters& cnt) {
    return *reinterpret_cast<const float*>(s.texels + texel_offset<COUNT>(s, tex, u, v, 4u, cnt));
}

struct Surface {
    uint32_t surface_material, behind_material;
    float u, v;
    v3 normal, mapped_normal;
    col4 color;
    float metalness, roughness, emission;
    float fresnel, reflectance, tint_factor;
    float refr_x, refr_y;
    float surface_scattering;  // surface_material->scattering(), read with the material
};

// analyzeIntersection: cpu_engine_kernel.cpp:354-395; mesh_component.cpp:115-167
template <bool COUNT, bool TEX = true>
RZ_DEV void analyze_intersection(const DScene& s, const Hit& hit, Surface& sf, Material& m, Counters& cnt, bool compat_filtering = false) {
    const uint32_t inst = uint32_t(hit.instance);
    const float4 i1 = s.instances[7 * inst + 1], i2 = s.instances[7 * inst + 2],
                 i3 = s.instances[7 * inst + 3], i4 = s.instances[7 * inst + 4];
    const v3 scale = xyz(i1), xa = xyz(i2), ya = xyz(i3), za = xyz(i4);
    const uint32_t material_base = __float_as_uint(i1.w), material_count = __float_as_uint(i2.w);

    const float4 ta = s.tris[3 * hit.triangle];
    const uint32_t flags = __float_as_uint(ta.w);
    const float4* at = s.tri_attrs + 6 * size_t(hit.triangle);

    uint32_t slot = flags & HIPRZ_TRI_MATERIAL_MASK;
    if (slot > 63u) slot = 63u;
    int32_t mat = -1;
    if (slot < material_count) mat = s.inst_materials[material_base + slot];
    sf.surface_material = mat < 0 ? HIPRZ_MATERIAL_DEFAULT : uint32_t(mat);
    sf.behind_material = hit.external ? sf.surface_material : HIPRZ_MATERIAL_WORLD;
    m = load_material(s, sf.surface_material);

    const bool has_texcrds = (flags & HIPRZ_TRI_HAS_TEXCRDS) != 0;
    float4 uv12 = make_float4(0, 0, 0, 0), uv3 = make_float4(0, 0, 0, 0);
    if (has_texcrds) {
        uv12 = at[4], uv3 = at[5];
        const float b3 = 1.0f - hit.bx - hit.by;
        sf.u = uv12.x * b3 + uv12.z * hit.bx + uv3.x * hit.by;
        sf.v = uv12.y * b3 + uv12.w * hit.bx + uv3.y * hit.by;
    }
    const float external_factor = float(hit.external) * 2.0f - 1.0f;
    const v3 face_normal = xyz(at[3]);
    if (flags & HIPRZ_TRI_HAS_NORMALS) {
        const v3 n1 = xyz(at[0]), n2 = xyz(at[1]), n3 = xyz(at[2]);
        sf.mapped_normal = normalized(n1 * (1.0f - hit.bx - hit.by) + n2 * hit.bx + n3 * hit.by);
    } else {
        sf.mapped_normal = face_normal;
    }
    if (TEX && m.normal_map >= 0 && has_texcrds) {  // Triangle::mapNormal, mesh_component.cpp:132-167
        const col4 map_color = compat_filtering ? compat_fetch<COUNT>(s, m.normal_map, sf.u, sf.v, cnt) : fetch_rgba8<COUNT>(s, m.normal_map, sf.u, sf.v, cnt);
        // v2 and v3 ride in the padding of the device attribute record (the triangle record holds edges)
        const v3 v1 = xyz(ta), v2 = V3(at[0].w, at[1].w, at[2].w), vv3 = V3(at[3].w, uv3.z, uv3.w);
        const v3 edge1 = (v2 - v1) * scale;
        const v3 edge2 = (vv3 - v1) * scale;
        const float duv1x = uv12.z - uv12.x, duv1y = uv12.w - uv12.y;
        const float duv2x = uv3.x - uv12.x, duv2y = uv3.y - uv12.y;
        v3 mn = sf.mapped_normal / scale;
        const float f = 1.0f / (duv1x * duv2y - duv2x * duv1y);
        v3 tangent = normalized((edge1 * duv2y - edge2 * duv1y) * f);
        tangent = normalized(tangent - mn * dot(tangent, mn));
        const v3 bitangent = cross(tangent, mn);
        const v3 map_n = V3(map_color.r, map_color.g, map_color.b) * 2.0f - V3(1.0f, 1.0f, 1.0f);
        mn = mn * map_n.z + tangent * map_n.x + bitangent * map_n.y;
        sf.mapped_normal = transform_forward(xa, ya, za, mn);  // transformL2GNoScale
    } else {
        sf.mapped_normal = transform_forward(xa, ya, za, sf.mapped_normal / scale);  // transformL2G
    }
    sf.mapped_normal = normalized(sf.mapped_normal);
    sf.mapped_normal = sf.mapped_normal * external_factor;

    sf.normal = face_normal * external_factor;
    sf.normal = transform_forward(xa, ya, za, sf.normal / scale);
    sf.normal = normalized(sf.normal);
}

// --- helpers: cpu_render_utils.cpp:29-170 -----------------------------------------------
RZ_DEV v3 reflect_vector(v3 vI, v3 vN) { return (vN * -2.0f) * dot(vN, vI) + vI; }
RZ_DEV v3 halfway_vector(v3 vI, v3 vR) { return normalized((-vI) + vR); }
RZ_DEV void local_coordinate(v3 vN, v3& vX, v3& vY) {
    const bool b = fabsf(vN.x) > fabsf(vN.y);
    vX = v3{float(!b), float(b), 0.0f};
    vY = cross(vN, vX);
    vX = cross(vN, vY);
}
RZ_DEV v3 cosine_sample_hemisphere(float r1, float r2, v3 vN) {
    v3 vX, vY;
    local_coordinate(vN, vX, vY);
    const float phi = r1 * 6.283185f;
    const float theta = r2;
    const float sqrt_theta = sqrtf(theta);
    float sin_phi, cos_phi;
    RZ_SINCOSF(phi, sin_phi, cos_phi);
    const v3 a = (vX * sqrt_theta) * cos_phi;
    const v3 b = (vY * sqrt_theta) * sin_phi;
    const v3 c = vN * sqrtf(1.0f - theta);
    return (a + b) + c;
}
RZ_DEV v3 sample_sphere(float r1, float r2, v3 vN) {
    v3 vX, vY;
    local_coordinate(vN, vX, vY);
    const float phi = r1 * 6.283185f;
    const float theta = RZ_ACOSF(1.0f - 2.0f * r2);
    float sin_theta, cos_theta, sin_phi, cos_phi;
    RZ_SINCOSF(theta, sin_theta, cos_theta);
    RZ_SINCOSF(phi, sin_phi, cos_phi);
    const v3 a = (vX * sin_theta) * cos_phi;
    const v3 b = (vY * sin_theta) * sin_phi;
    const v3 c = vN * cos_theta;
    return (a + b) + c;
}
RZ_DEV v3 sample_hemisphere(float r1, float r2, v3 vN) { return sample_sphere(r1, r2 * 0.5f, vN); }
RZ_DEV v3 sample_disk(float r1, float r2, v3 vN, float radius) {
    v3 vX, vY;
    local_coordinate(vN, vX, vY);
    const float phi = r1 * 2.0f * RZ_PI_F;
    const float mag = sqrtf(r2);
    float sin_phi, cos_phi;
    RZ_SINCOSF(phi, sin_phi, cos_phi);
    return ((vX * sin_phi + vY * cos_phi) * mag) * radius;
}
RZ_DEV float fresnel_specular_ratio(v3 vN, v3 vI, float n1, float n2, float& fx, float& fy) {
    const float ratio = n1 / n2;
    const float cosi = fabsf(dot(vI, vN));
    const float sin2_t = ratio * ratio * (1.0f - cosi * cosi);
    if (sin2_t >= 1.0f) return 1.0f;
    const float cost = sqrtf(1.0f - sin2_t);
    const float Rp = ((n1 * cosi) - (n2 * cost)) / ((n1 * cosi) + (n2 * cost));
    const float Rs = ((n2 * cosi) - (n1 * cost)) / ((n2 * cosi) + (n1 * cost));
    fx = ratio;
    fy = ratio * cosi - cost;
    return (Rs * Rs + Rp * Rp) / 2.0f;
}

// --- BRDF: cpu_engine_kernel.cpp:556-594 ------------------------------------------------
RZ_DEV float ndf(v3 vN, v3 vH, float roughness) {
    const float d = dot(vN, vH);
    const float b = (d * d) * (roughness - 1.0f) + 1.0001f;
    return (roughness + 1.0e-5f) / (b * b);
}
RZ_DEV float attenuation(float cos_angle, float roughness) {
    return cos_angle / ((cos_angle * (1.0f - roughness)) + roughness);
}
RZ_DEV float brdf(v3 ray_d, const Surface& sf, v3 vPL) {
    if (sf.surface_scattering > 0.0f) return 1.0f;
    const float vN_dot_vO = dot(sf.mapped_normal, vPL);
    if (vN_dot_vO <= 0.0f) return 0.0f;
    const float vN_dot_vI = dot(sf.mapped_normal, -ray_d);
    if (vN_dot_vI <= 0.0f) return 0.0f;
    const v3 vH = halfway_vector(ray_d, vPL);
    const float nd = ndf(sf.mapped_normal, vH, sf.roughness);
    const float atten_i = attenuation(vN_dot_vI, sf.roughness);
    const float atten_o = attenuation(vN_dot_vO, sf.roughness);
    const float atten = atten_i * atten_o;
    const float diffuse = vN_dot_vO * float(sf.color.a == 0.0f);
    const float specular = nd * atten / (vN_dot_vI * vN_dot_vO);
    return lerpf(diffuse, specular * vN_dot_vO, sf.reflectance);
}
RZ_DEV col4 brdf_color(const Surface& sf) { return lerp(sf.color, splat(1.0f), sf.reflectance); }

// --- direction sampling: cpu_engine_kernel.cpp:596-687 ----------------------------------
RZ_DEV v3 sample_direction(v3 ray_d, uint32_t& ray_material, Surface& sf, Rng& rng) {
    if (sf.color.a > 0.0f) {
        if (sf.surface_scattering > 0.0f) {
            const float u1 = rng.unsignedUniform();
            const float u2 = rng.unsignedUniform();
            const v3 vO = sample_sphere(u1, u2, ray_d);
            sf.tint_factor = sf.metalness;
            return vO;
        }
        if (sf.fresnel < rng.unsignedUniform()) {
            const v3 vO = ray_d * sf.refr_x + sf.mapped_normal * sf.refr_y;
            ray_material = sf.behind_material;
            sf.normal = -sf.normal;
            sf.tint_factor = 1.0f;
            return vO;
        }
        v3 vO = reflect_vector(ray_d, sf.mapped_normal);
        const float d = dot(vO, sf.normal);
        if (d < 0.0f) vO = vO + (sf.normal * -2.0f) * d;
        sf.tint_factor = sf.metalness;
        return vO;
    }
    // diffuse (cosine_sample_hemisphere) or glossy (sample_hemisphere around the normal, then a mirror reflection about that half
    // vector): a wave nearly always holds both kinds (a dielectric reflects a few percent of its rays), so what the two have in
    // common — the draws, the tangent frame, sincosf(phi), the combination, the flip above the surface — is spelled ONCE, outside the
    // branch; each lane's operations and their order are those of its own formula (cpu_render_utils.cpp:45-104).
    const bool diffuse = rng.unsignedUniform() > sf.reflectance;
    const float u1 = rng.unsignedUniform();
    const float u2 = rng.unsignedUniform();
    float s_xy, c_z;
    if (diffuse) {
        s_xy = sqrtf(u2), c_z = sqrtf(1.0f - u2);
    } else {
        const float theta = RZ_ACOSF(1.0f - 2.0f * ((1.0f - RZ_POWF(u2 + 1.0e-5f, sf.roughness)) * 0.5f));
        RZ_SINCOSF(theta, s_xy, c_z);
    }
    v3 vX, vY;
    local_coordinate(sf.mapped_normal, vX, vY);
    const float phi = u1 * 6.283185f;
    float sin_phi, cos_phi;
    RZ_SINCOSF(phi, sin_phi, cos_phi);
    const v3 around = (((vX * s_xy) * cos_phi) + ((vY * s_xy) * sin_phi)) + sf.mapped_normal * c_z;
    v3 vO = diffuse ? around : reflect_vector(ray_d, around);
    const float d = similarity(vO, sf.normal);
    if (d < 0.0f) vO = vO + (sf.normal * -2.0f) * d;
    sf.tint_factor = diffuse ? 1.0f : sf.metalness;
    return vO;
}

RZ_DEV void defer_sample(const ShadowCtx& sc, uint32_t slot, const Ray& sr, col4 term, uint32_t light = 0u) {
    sc.nee[4u + 2u * slot] = make_float4(sr.d.x, sr.d.y, sr.d.z, sr.far_);
    sc.nee[5u + 2u * slot] = make_float4(term.r, term.g, term.b, term.a);
    sc.nee[1] = make_float4(sr.o.x, sr.o.y, sr.o.z, 0.0f);
    if ((1u << slot) > sc.defer_mask) {
        sc.key_dir[0] = sr.d.x, sc.key_dir[1] = sr.d.y, sc.key_dir[2] = sr.d.z;
        sc.key_o[0] = sr.o.x, sc.key_o[1] = sr.o.y, sc.key_o[2] = sr.o.z;
        sc.key_light = light;
    }
    sc.defer_mask |= 1u << slot;
}

// --- next-event estimation: cpu_engine_kernel.cpp:690-865 -------------------------------
// The two halves of a sample's contribution — (light colour * brdf colour) * radiance, and the shadow mask it is multiplied
// with twice (V_PL, then V_PL.alpha: the CPU kernel's mask is a colour) — are kept apart so that MODE RZ_SHADOW_DEFER can
// hand the first half and the shadow ray to rz_shadow_kernel.
template <int MODE, bool COUNT>
RZ_DEV col4 direct_illumination(const DScene& s, const DConfig& cfg, const ShadowCtx& lds_column, v3 ray_d, uint32_t ray_material,
                                v3 point, v3 next_dir, const Surface& sf, Rng& rng, Counters& cnt) {
    if constexpr (MODE == RZ_SHADOW_NONE || MODE == RZ_SHADOW_PLAIN) return splat(0.0f);
    if (s.n_direct_lights == 0u && s.n_spot_lights == 0u) return splat(0.0f);  // both samplers return 0 before they read vS_pdf (:703, :758)
    const float vS_pdf = brdf(ray_d, sf, next_dir);
    col4 direct_total = splat(0.0f);
    if (s.n_direct_lights != 0) {  // directLightSampling :745-791
        for (uint32_t i = 0; i < cfg.direct_samples; ++i) {
            uint32_t li = uint32_t(rng.unsignedUniform() * float(s.n_direct_lights));
            if (li >= s.n_direct_lights) li = s.n_direct_lights - 1u;
            const float4 l0 = s.direct_lights[2 * li], l1 = s.direct_lights[2 * li + 1];
            RZ_COUNT(light_samples);
            const v3 ldir = xyz(l0);
            const float emission = l0.w, cos_angle = l1.z;
            float Se = 0.0f;
            v3 vPL;
            const float d = dot(next_dir, -ldir);
            if (d > cos_angle) {
                Se = emission;
                vPL = next_dir;
            } else {
                const float u1 = rng.unsignedUniform();
                const float u2 = rng.unsignedUniform();
                vPL = sample_sphere(u1, u2 * 0.5f * (1.0f - cos_angle), -ldir);
            }
            const float b = brdf(ray_d, sf, normalized(vPL));
            const col4 bc = brdf_color(sf);
            const float solid_angle = 2.0f * RZ_PI_F * (1.0f - cos_angle);
            const float L_pdf = 1.0f / solid_angle;
            const float vSw = vS_pdf / (vS_pdf + L_pdf);
            const float Lw = 1.0f - vSw;
            const float Le = emission * solid_angle * b;
            const float radiance = (Le * Lw + Se * vSw);
            if (radiance < 1.0e-4f) continue;
            Ray sr;
            sr.o = point, sr.d = normalized(vPL), sr.near_ = 0.0f, sr.far_ = RZ_FLT_MAX;
            const col4 term = (from_u8(__float_as_uint(l1.x)) * bc) * radiance;
            if constexpr (shadow_mode_defers(MODE)) {
                defer_sample(lds_column, i, sr, term, 4u + li);
            } else {
                col4 V_PL;
                if constexpr (MODE == RZ_SHADOW_COMPAT)
                    V_PL = (cfg.flags & HIPRZ_COMPAT_SHADOW_COLOR) ? compat_shadow_mask<COUNT>(s, sr, (cfg.flags & HIPRZ_COMPAT_FILTERING) != 0u, cnt) : splat(any_hit<3, COUNT>(s, lds_column, sr, cnt));
                else V_PL = splat(any_hit<MODE, COUNT>(s, lds_column, sr, cnt));
                direct_total = direct_total + (term * V_PL) * V_PL.a;
            }
        }
        const float pdf = float(cfg.direct_samples) / float(s.n_direct_lights);
        direct_total = div_scalar(direct_total, pdf);
    }
    col4 spot_total = splat(0.0f);
    if (s.n_spot_lights != 0) {  // spotLightSampling :690-744
        for (uint32_t i = 0; i < cfg.spot_samples; ++i) {
            uint32_t li = uint32_t(rng.unsignedUniform() * float(s.n_spot_lights));
            if (li >= s.n_spot_lights) li = s.n_spot_lights - 1u;
            const float4 l0 = s.spot_lights[3 * li], l1 = s.spot_lights[3 * li + 1], l2 = s.spot_lights[3 * li + 2];
            RZ_COUNT(light_samples);
            const v3 lpos = xyz(l0), ldir = xyz(l1);
            const float size = l0.w, emission = l1.w, cos_angle = l2.z;
            float Se = 0.0f;
            v3 vPL;
            {  // spotLightSampleDirection :805-828, rayPointCalculation cpu_render_utils.cpp:48-72
                const v3 rd = normalized(next_dir);
                const v3 vOP = lpos - point;
                const float dOP = magnitude(vOP);
                const float vOP_dot_vD = dot(vOP, rd);
                const float dPQ = sqrtf(dOP * dOP - vOP_dot_vD * vOP_dot_vD);
                if (dPQ < size && vOP_dot_vD > 0.0f) {
                    Se = emission;
                    const float dOQ = sqrtf(dOP * dOP - dPQ * dPQ);
                    vPL = next_dir * fmaxf(dOQ, 1.0e-4f);
                } else {
                    const float u1 = rng.unsignedUniform();
                    const float u2 = rng.unsignedUniform();
                    vPL = (sample_disk(u1, u2, vOP / dOP, size) + lpos) - point;
                }
            }
            const float dPL = magnitude(vPL);
            const float b = brdf(ray_d, sf, vPL / dPL);
            if (b < 1.0e-4f) continue;
            const col4 bc = brdf_color(sf);
            const float A = size * size * RZ_PI_F;
            const float d1 = dPL + 1.0f;
            const float solid_angle = A / (d1 * d1);
            const float sctr_factor = RZ_EXPF(-dPL * material_scattering(s, ray_material));
            const float beam = float(cos_angle < similarity(-vPL, ldir));
            if (beam < 1.0e-4f) continue;
            const float L_pdf = 1.0f / solid_angle;
            const float vSw = vS_pdf / (vS_pdf + L_pdf);
            const float Lw = 1.0f - vSw;
            const float Le = emission * solid_angle * b;
            const float radiance = (Le * Lw + Se * vSw) * sctr_factor * beam;
            if (radiance < 1.0e-4f) continue;
            Ray sr;
            sr.o = point, sr.d = normalized(vPL), sr.near_ = 0.0f, sr.far_ = dPL;
            const col4 term = (from_u8(__float_as_uint(l2.x)) * bc) * radiance;
            if constexpr (shadow_mode_defers(MODE)) {
                defer_sample(lds_column, cfg.direct_samples + i, sr, term, li);
            } else {
                col4 V_PL;
                if constexpr (MODE == RZ_SHADOW_COMPAT)
                    V_PL = (cfg.flags & HIPRZ_COMPAT_SHADOW_COLOR) ? compat_shadow_mask<COUNT>(s, sr, (cfg.flags & HIPRZ_COMPAT_FILTERING) != 0u, cnt) : splat(any_hit<3, COUNT>(s, lds_column, sr, cnt));
                else V_PL = splat(any_hit<MODE, COUNT>(s, lds_column, sr, cnt));
                spot_total = spot_total + (term * V_PL) * V_PL.a;
            }
        }
        const float pdf = float(cfg.spot_samples) / float(s.n_spot_lights);
        spot_total = div_scalar(spot_total, pdf);
    }
    return direct_total + spot_total;
}

// --- camera rays: cpu_engine_kernel.cpp:180-252 -----------------------------------------
RZ_DEV void screen_direction(const DCamera& c, uint32_t px, uint32_t py, float& dx, float& dy) {
    const float tana = c.tan_half_fov;
    dx = (((float(px) + 0.5f) / float(c.width)) - 0.5f) * tana;
    dy = (((float(py) + 0.5f) / float(c.height)) - 0.5f) * (-tana / c.aspect_ratio);
}
RZ_DEV void generate_simple_ray(const DCamera& c, Ray& ray, uint32_t px, uint32_t py) {
    float dx, dy;
    screen_direction(c, px, py, dx, dy);
    const v3 xa = ld3(c.x_axis), ya = ld3(c.y_axis), za = ld3(c.z_axis);
    ray.o = transform_forward(xa, ya, za, V3(0.0f, 0.0f, 0.0f)) + ld3(c.position);
    ray.d = normalized(transform_forward(xa, ya, za, V3(dx, dy, 1.0f)));
    ray.near_ = c.near_, ray.far_ = c.far_;
}
RZ_DEV void generate_antialiased_ray(const DCamera& c, Ray& ray, uint32_t px, uint32_t py, Rng& rng) {
    float dx, dy;
    screen_direction(c, px, py, dx, dy);
    v3 dir = V3(dx, dy, 1.0f);
    dir.x += ((0.5f / float(c.width)) * rng.signedUniform());
    dir.y += ((0.5f / float(c.width)) * rng.signedUniform());  // (sic) cpu_engine_kernel.cpp:227-228
    const v3 focal_point = dir * c.focal_distance;
    const float aperture_angle = rng.unsignedUniform() * 2.0f * RZ_PI_F;
    const float aperture_sample = sqrtf(rng.unsignedUniform()) * c.aperture;
    float sin_a, cos_a;
    RZ_SINCOSF(aperture_angle, sin_a, cos_a);
    const v3 origin = V3(aperture_sample * sin_a, aperture_sample * cos_a, 0.0f);
    dir = focal_point - origin;
    const v3 xa = ld3(c.x_axis), ya = ld3(c.y_axis), za = ld3(c.z_axis);
    ray.o = transform_forward(xa, ya, za, origin) + ld3(c.position);
    ray.d = normalized(transform_forward(xa, ya, za, dir));
    ray.near_ = c.near_, ray.far_ = c.far_;
}

// Tone map: cpu_engine_renderer.cpp:224-235
RZ_DEV uint32_t tonemap(col4 color, float aperture, float exposure_time) {
    const float aperture_area = aperture * aperture * RZ_PI_F;
    color = div_scalar(color, color.a == 0.0f ? 1.0f : color.a);
    color = color * aperture_area;
    color = color * exposure_time;
    color = color * 1.0e5f;
    color = color / (color + splat(1.0f));
    const uint32_t r = uint32_t(uint8_t(color.r * 255.0f)), g = uint32_t(uint8_t(color.g * 255.0f)),
                   b = uint32_t(uint8_t(color.b * 255.0f));
    return r | (g << 8) | (b << 16) | 0xFF000000u;
}

// thread -> pixel.  A block is one 32x8 tile (4 waves of 8x8 pixels); where owned tile `lt` of
// shard (rank, world) lies in the frame: hiprz_shard.hpp.
struct PixelId {
    uint32_t x, y, local;
    bool active;
    bool tile_inside;  // the thread's workgroup has a tile (the swizzled grid is padded to a multiple of 8 workgroups)
};
RZ_DEV PixelId pixel_of_thread(const DFrame& f, const DCamera& c, uint32_t block, uint32_t tid) {
    // Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one, MI355X_MICROARCH.md), each with its
    // own 4 MiB L2.  With the swizzle, XCD k works through the k-th contiguous eighth of the owned tiles — one band of
    // the image — so the tree nodes its rays visit are shared within ONE L2 instead of being replicated in all eight.
    // Only the execution order changes: storage stays indexed by the owned-tile number.
    uint32_t lt = block;
    if (f.xcd_swizzle) {
        const uint32_t per_xcd = (f.n_local_tiles + 7u) >> 3;
        lt = (block & 7u) * per_xcd + (block >> 3);
    }
    uint32_t tx, ty;
    shard_tile(lt, f.tiles_x, f.rank, f.world, tx, ty);
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    PixelId p;
    p.x = tx * 32u + wave * 8u + (lane & 7u);
    p.y = ty * 8u + (lane >> 3);
    p.local = lt * 256u + tid;
    p.tile_inside = lt < f.n_local_tiles;
    p.active = lt < f.n_local_tiles && p.x < c.width && p.y < c.height;
    return p;
}

// PixelId of a local (tile-major) pixel index — the trace kernel's mapping when rays are walked in sorted order
RZ_DEV PixelId pixel_of_local(const DFrame& f, const DCamera& c, uint32_t local) {
    const uint32_t lt = local >> 8, tid = local & 255u;
    uint32_t tx, ty;
    shard_tile(lt, f.tiles_x, f.rank, f.world, tx, ty);
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    PixelId p;
    p.x = tx * 32u + wave * 8u + (lane & 7u);
    p.y = ty * 8u + (lane >> 3);
    p.local = local;
    p.tile_inside = lt < f.n_local_tiles;
    p.active = lt < f.n_local_tiles && p.x < c.width && p.y < c.height;
    return p;
}

// Sort key of a ray: 15-bit Morton code of the origin's cell in a 32^3 grid over the world box, then 3 bits per
// direction component.  Rays with equal keys start close together and point the same way, so a wave that takes 64
// consecutive rays of the sorted order walks the same nodes (coherent fetches, less divergence).  The key only
// decides WHICH THREAD walks a ray; results do not depend on it.
RZ_DEV uint32_t spread3(uint32_t v) {  // 5 bits -> every third bit
    v = (v | (v << 8)) & 0x0000F00Fu;
    v = (v | (v << 4)) & 0x000C30C3u;
    v = (v | (v << 2)) & 0x00249249u;
    return v;
}
// groups of 3 bits at 3k -> 6k (k = 0..3): the levels of a 12-bit Morton code, three bits apart
RZ_DEV uint32_t spread_levels(uint32_t v) {
    v = (v & 0x03Fu) | ((v & 0xFC0u) << 6);
    return (v & 0x007007u) | ((v & 0x038038u) << 3);
}
RZ_DEV uint32_t ray_sort_key(const DScene& s, v3 o, v3 d, uint32_t variant) {
    if (variant == 4u) {
        // two points instead of a point and a direction: the origin's cell and the cell where the ray LEAVES the world box, 16^3 each,
        // interleaved level by level (3 origin bits, 3 exit bits, four times).  Rays from one cell to one cell form a thin beam whatever
        // the distance between the two, while equal direction codes fan out with it — in a closed room the exit cell is close to what the
        // ray will hit.  Round 4, trace kernel: E 2 556 -> 2 400 us (layout 3), C 316 -> 299; exit bits first inside a level 2 467, bit by
        // bit 2 437, one origin level on top 2 484, 15 + 9 or 9 + 15 bits 2 408 / 2 440 (profiles/r04/ab_sort.txt).
        // In grid units (the box is [0, 32]^3); approximate reciprocals are good enough — the key only decides WHICH THREAD walks a ray.
        const float og[3] = {(o.x - s.bounds_min[0]) * s.bounds_scale[0], (o.y - s.bounds_min[1]) * s.bounds_scale[1], (o.z - s.bounds_min[2]) * s.bounds_scale[2]};
        const float dg[3] = {d.x * s.bounds_scale[0], d.y * s.bounds_scale[1], d.z * s.bounds_scale[2]};
        float t_exit = 3.0e38f;
        for (int a = 0; a < 3; ++a) {
            const float t = ((dg[a] > 0.0f ? 32.0f : 0.0f) - og[a]) * __builtin_amdgcn_rcpf(dg[a]);
            t_exit = fminf(t_exit, dg[a] != 0.0f ? fmaxf(t, 0.0f) : 3.0e38f);
        }
        if (!(t_exit < 1.0e37f)) t_exit = 0.0f;
        uint32_t oi[3], ei[3];
        for (int a = 0; a < 3; ++a) {
            oi[a] = uint32_t(fminf(fmaxf(og[a], 0.0f), 31.0f)) >> 1;
            ei[a] = uint32_t(fminf(fmaxf(og[a] + dg[a] * t_exit, 0.0f), 31.0f)) >> 1;
        }
        // x, y, z of both points side by side (exit in bits 0..3, origin in 4..7: spread3 takes 8 bits), then the levels three bits apart
        const uint32_t mx = spread3((oi[0] << 4) | ei[0]), my = spread3((oi[1] << 4) | ei[1]), mz = spread3((oi[2] << 4) | ei[2]);
        const uint32_t both = mx | (my << 1) | (mz << 2);  // exit code in bits 0..11, origin code in bits 12..23
        return (spread_levels(both >> 12) << 3) | spread_levels(both & 0xFFFu);
    }
    const float cx = fminf(fmaxf((o.x - s.bounds_min[0]) * s.bounds_scale[0], 0.0f), 31.0f);
    const float cy = fminf(fmaxf((o.y - s.bounds_min[1]) * s.bounds_scale[1], 0.0f), 31.0f);
    const float cz = fminf(fmaxf((o.z - s.bounds_min[2]) * s.bounds_scale[2], 0.0f), 31.0f);
    const uint32_t morton = spread3(uint32_t(cx)) | (spread3(uint32_t(cy)) << 1) | (spread3(uint32_t(cz)) << 2);
    const float inv = __builtin_amdgcn_rcpf(fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), 1.0e-30f)));  // (approximate: the key only decides which thread walks a ray)
    const uint32_t qx = uint32_t(fminf(fmaxf(d.x * inv * 3.99f + 4.0f, 0.0f), 7.0f));
    const uint32_t qy = uint32_t(fminf(fmaxf(d.y * inv * 3.99f + 4.0f, 0.0f), 7.0f));
    const uint32_t qz = uint32_t(fminf(fmaxf(d.z * inv * 3.99f + 4.0f, 0.0f), 7.0f));
    if (variant == 1u) return (((qx << 6) | (qy << 3) | qz) << 15) | morton;  // direction-major
    if (variant == 3u) {
        // the direction on the octahedron — two coordinates of 6 bits in Z-order instead of three of which one is saturated — interleaved
        // with the origin's cell in a 16^3 grid: 2 direction bits, 3 cell bits, four times, then the direction's last 4 bits.  (Round 4, config E:
        // trace kernel 2 664 -> 2 558 us against the 6-D code; the same bits direction-first 3 561, cell-first within a level 2 592.)
        const float l1 = 1.0f / fmaxf(fabsf(d.x) + fabsf(d.y) + fabsf(d.z), 1.0e-30f);
        float u = d.x * l1, v = d.y * l1;
        if (d.z < 0.0f) {
            const float uu = (1.0f - fabsf(v)) * (u < 0.0f ? -1.0f : 1.0f), vv = (1.0f - fabsf(u)) * (v < 0.0f ? -1.0f : 1.0f);
            u = uu, v = vv;
        }
        const uint32_t iu = uint32_t(fminf(fmaxf((u + 1.0f) * 32.0f, 0.0f), 63.0f)), iv = uint32_t(fminf(fmaxf((v + 1.0f) * 32.0f, 0.0f), 63.0f));
        uint32_t dir = 0u;
        for (uint32_t b = 0u; b < 6u; ++b) dir |= (((iu >> b) & 1u) << (2u * b)) | (((iv >> b) & 1u) << (2u * b + 1u));
        const uint32_t coarse = spread3(uint32_t(cx) >> 1) | (spread3(uint32_t(cy) >> 1) << 1) | (spread3(uint32_t(cz) >> 1) << 2);
        uint32_t key = 0u;
        for (int b = 3; b >= 0; --b) key = (key << 5) | (((dir >> (2 * b + 4)) & 3u) << 3) | ((coarse >> (3 * b)) & 7u);
        return (key << 4) | (dir & 15u);
    }
    if (variant == 2u) {  // 6-D Morton code: 4 bits of each origin cell coordinate and of each direction component, interleaved
        const uint32_t px = uint32_t(cx) >> 1, py = uint32_t(cy) >> 1, pz = uint32_t(cz) >> 1;
        const uint32_t dx = uint32_t(fminf(fmaxf(d.x * inv * 7.99f + 8.0f, 0.0f), 15.0f)), dy = uint32_t(fminf(fmaxf(d.y * inv * 7.99f + 8.0f, 0.0f), 15.0f)),
                       dz = uint32_t(fminf(fmaxf(d.z * inv * 7.99f + 8.0f, 0.0f), 15.0f));
        uint32_t key = 0u;
        for (int b = 3; b >= 0; --b)
            key = (key << 6) | (((dx >> b) & 1u) << 5) | (((dy >> b) & 1u) << 4) | (((dz >> b) & 1u) << 3) | (((px >> b) & 1u) << 2) | (((py >> b) & 1u) << 1) | ((pz >> b) & 1u);
        return key;
    }
    return (morton << 9) | (qx << 6) | (qy << 3) | qz;
}

}  // namespace hiprz
