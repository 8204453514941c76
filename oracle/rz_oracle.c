/*
 * rz_oracle.c — TEST INFRASTRUCTURE ONLY (see rz_oracle.h: "PARITY UNPINNED").
 *
 * Plain-C restatement of the reference CPU path tracer.  Every function cites the
 * reference lines it follows (paths relative to /root/reference/RayZath/).  Arithmetic
 * is written out operation by operation in the order the reference's expressions
 * evaluate (left to right; function arguments and `+` operands left to right, which is
 * what clang does — SURVEY.md Appendix D), and the file must be compiled with
 * -ffp-contract=off so no multiply-add is fused.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp -shared)
 */
#include "rz_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* All transcendental calls go through these (glibc's libm on the CPU side; the device build uses ocml). */
#define RZ_SINF(x) sinf(x)
#define RZ_COSF(x) cosf(x)
#define RZ_ACOSF(x) acosf(x)
#define RZ_ASINF(x) asinf(x)
#define RZ_ATAN2F(y, x) atan2f(y, x)
#define RZ_POWF(x, y) powf(x, y)
#define RZ_EXPF(x) expf(x)
const char* rzo_math_mode(void) { return "libm"; }

#define RZ_PI 3.14159265358979323846f /* std::numbers::pi_v<float> */

/* ------------------------------------------------------------------------------------
 * Math::vec3f / Math::vec2f / Graphics::ColorF — the un-vendored types.  Conventions
 * follow the reference's CUDA restatement (cuda_render_parts.cuh:15-330, 520-700):
 *   dot = x*x' + y*y' + z*z';  Magnitude = sqrtf(dot);  Normalize multiplies by
 *   1.0f/Magnitude;  Similarity = dot * (rcp|a| * rcp|b|);  v/float and v/v divide
 *   componentwise;  ColorF/float multiplies by the reciprocal;  Color->ColorF = /255.0f.
 * ---------------------------------------------------------------------------------- */
typedef struct {
    float x, y, z;
} v3;
typedef struct {
    float r, g, b, a;
} col;

static inline v3 V3(float x, float y, float z) {
    v3 v = {x, y, z};
    return v;
}
static inline v3 v3_from(const float* p) { return V3(p[0], p[1], p[2]); }
static inline v3 v3_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_div(v3 a, v3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 v3_scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_divs(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 v3_neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float v3_mag(v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
static inline float v3_rcp_mag(v3 a) { return 1.0f / v3_mag(a); }
static inline v3 v3_normalized(v3 a) { return v3_scale(a, v3_rcp_mag(a)); }
static inline float v3_similarity(v3 a, v3 b) { return v3_dot(a, b) * (v3_rcp_mag(a) * v3_rcp_mag(b)); }

static inline col COL(float r, float g, float b, float a) {
    col c = {r, g, b, a};
    return c;
}
static inline col col_splat(float v) { return COL(v, v, v, v); }
static inline col col_from_u8(const uint8_t* c) {
    return COL(c[0] / 255.0f, c[1] / 255.0f, c[2] / 255.0f, c[3] / 255.0f);
}
static inline col col_add(col a, col b) { return COL(a.r + b.r, a.g + b.g, a.b + b.b, a.a + b.a); }
static inline col col_sub(col a, col b) { return COL(a.r - b.r, a.g - b.g, a.b - b.b, a.a - b.a); }
static inline col col_mul(col a, col b) { return COL(a.r * b.r, a.g * b.g, a.b * b.b, a.a * b.a); }
static inline col col_scale(col a, float s) { return COL(a.r * s, a.g * s, a.b * s, a.a * s); }
static inline col col_divs(col a, float s) { return col_scale(a, 1.0f / s); }
static inline col col_div(col a, col b) { return COL(a.r / b.r, a.g / b.g, a.b / b.b, a.a / b.a); }
/* template lerp, cpu_render_utils.hpp:203-207 */
static inline float lerpf(float a, float b, float t) { return a + (b - a) * t; }
static inline col col_lerp(col a, col b, float t) { return col_add(a, col_scale(col_sub(b, a), t)); }

/* ------------------------------------------------------------------------------------
 * RNG — cpu_render_utils.cpp:8-27
 * ---------------------------------------------------------------------------------- */
typedef struct {
    float a, b;
} rng_t;

static inline rng_t rng_make(float seed_x, float seed_y, float r) {
    rng_t g;
    g.a = seed_x + seed_y;
    g.b = r * 245.310913f;
    return g;
}
static inline float rng_unsigned(rng_t* g) {
    const float af = (g->a + 0.2311362f) * (g->b + 13.054377f);
    const float bf = (g->a + 251.78431f) + (g->b - 73.054312f);
    g->a = af - (float)((int32_t)af);
    g->b = bf - (float)((int32_t)bf);
    return fabsf(g->b);
}
static inline float rng_signed(rng_t* g) { return rng_unsigned(g) * 2.0f - 1.0f; }

/* Deterministic stand-in for Seeds::reconstruct (cuda_kernel_data.cu:10-18): entry i of
 * the 256-entry table of pass `pass`, uniform in [-10,10).  Spec shared with the backend
 * (hiprz.h: hiprz_seed_value), implemented independently. */
static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
float rzo_seed_value(uint32_t seed, uint32_t pass, uint32_t i) {
    uint32_t h = mix32(seed ^ mix32(pass + 0x9E3779B9u));
    h = mix32(h ^ (i * 0x85EBCA6Bu + 1u));
    return (float)(h >> 8) * (20.0f / 16777216.0f) - 10.0f;
}
void rzo_rng_sequence(float seed_x, float seed_y, float r, uint32_t n, float* out) {
    rng_t g = rng_make(seed_x, seed_y, r);
    for (uint32_t i = 0; i < n; ++i) out[i] = rng_unsigned(&g);
}

/* ------------------------------------------------------------------------------------
 * Rays and per-segment records — cpu_render_utils.hpp:33-170
 * ---------------------------------------------------------------------------------- */
typedef struct {
    v3 origin, direction;
    float near_, far_; /* near_far.x / .y */
    uint32_t material; /* const Material* */
    col color;
} ray_t;

typedef struct {
    int32_t closest_instance; /* -1 = nullptr */
    int32_t closest_triangle; /* global index into scene->tris, -1 = nullptr */
    float bx, by;
    int external;
} traversal_t;

typedef struct {
    uint32_t surface_material, behind_material;
    float u, v; /* texcrd */
    v3 normal, mapped_normal;
    col color;
    float metalness, roughness, emission;
    float fresnel, reflectance, tint_factor;
    float refr_x, refr_y;
} surface_t;

typedef struct {
    const hiprz_scene* s;
    const hiprz_camera* cam;
    const hiprz_config* cfg;
    hiprz_counters* cnt; /* per-thread, may be NULL */
} kctx;

#define COUNT(k, field, n) \
    do {                   \
        if ((k)->cnt) (k)->cnt->field += (n); \
    } while (0)

/* ------------------------------------------------------------------------------------
 * BoundingBox::rayIntersection — render_parts.cpp:197-217
 * ---------------------------------------------------------------------------------- */
static inline float my_min(float a, float b) { return a < b ? a : b; }
static inline float my_max(float a, float b) { return a > b ? a : b; }
static int box_hit(const float* mn, const float* mx, const ray_t* ray) {
    float t1 = (mn[0] - ray->origin.x) / ray->direction.x;
    float t2 = (mx[0] - ray->origin.x) / ray->direction.x;
    float t3 = (mn[1] - ray->origin.y) / ray->direction.y;
    float t4 = (mx[1] - ray->origin.y) / ray->direction.y;
    float t5 = (mn[2] - ray->origin.z) / ray->direction.z;
    float t6 = (mx[2] - ray->origin.z) / ray->direction.z;
    float tmin = my_max(my_max(my_min(t1, t2), my_min(t3, t4)), my_min(t5, t6));
    float tmax = my_min(my_min(my_max(t1, t2), my_max(t3, t4)), my_max(t5, t6));
    return !(tmax < ray->near_ || tmin > tmax || tmin > ray->far_);
}
int rzo_box_test(const float bb_min[3], const float bb_max[3], const float origin[3], const float direction[3],
                 float near_, float far_) {
    ray_t r;
    r.origin = v3_from(origin);
    r.direction = v3_from(direction);
    r.near_ = near_;
    r.far_ = far_;
    return box_hit(bb_min, bb_max, &r);
}

/* ------------------------------------------------------------------------------------
 * Triangle::closestIntersection / anyIntersection — mesh_component.cpp:52-83 / 84-114
 * Shared Moller-Trumbore core: returns 1 and writes t,b1,b2,det on acceptance.
 * ---------------------------------------------------------------------------------- */
static int tri_hit(const hiprz_tri* tri, const ray_t* ray, float* t_out, float* b1_out, float* b2_out,
                   float* det_out) {
    const v3 v1 = v3_from(tri->v1), v2 = v3_from(tri->v2), vv3 = v3_from(tri->v3);
    const v3 edge1 = v3_sub(v2, v1);
    const v3 edge2 = v3_sub(vv3, v1);
    const v3 pvec = v3_cross(ray->direction, edge2);

    float det = v3_dot(edge1, pvec);
    det += (float)((uint8_t)(det > -1.0e-7f) & (uint8_t)(det < 1.0e-7f)) * 1.0e-7f;
    const float inv_det = 1.0f / det;

    const v3 tvec = v3_sub(ray->origin, v1);
    const float b1 = v3_dot(tvec, pvec) * inv_det;
    if (b1 < 0.0f || b1 > 1.0f) return 0;

    const v3 qvec = v3_cross(tvec, edge1);
    const float b2 = v3_dot(ray->direction, qvec) * inv_det;
    if (b2 < 0.0f || b1 + b2 > 1.0f) return 0;

    const float t = v3_dot(edge2, qvec) * inv_det;
    if (t <= ray->near_ || t >= ray->far_) return 0;

    *t_out = t;
    *b1_out = b1;
    *b2_out = b2;
    *det_out = det;
    return 1;
}
int rzo_triangle_test(const float v1[3], const float v2[3], const float vv3[3], const float origin[3],
                      const float direction[3], float near_, float far_, float out4[4]) {
    hiprz_tri tri;
    memset(&tri, 0, sizeof tri);
    memcpy(tri.v1, v1, 12);
    memcpy(tri.v2, v2, 12);
    memcpy(tri.v3, vv3, 12);
    ray_t r;
    r.origin = v3_from(origin);
    r.direction = v3_from(direction);
    r.near_ = near_;
    r.far_ = far_;
    float t, b1, b2, det;
    if (!tri_hit(&tri, &r, &t, &b1, &b2, &det)) return 0;
    out4[0] = t;
    out4[1] = b1;
    out4[2] = b2;
    out4[3] = det > 0.0f ? 1.0f : 0.0f;
    return 1;
}

/* ------------------------------------------------------------------------------------
 * Transformation — render_parts.cpp:42-50, 117-135
 * ---------------------------------------------------------------------------------- */
static inline v3 transform_forward(const float* xa, const float* ya, const float* za, v3 v) {
    /* x_axis * v.x + y_axis * v.y + z_axis * v.z */
    return v3_add(v3_add(v3_scale(v3_from(xa), v.x), v3_scale(v3_from(ya), v.y)), v3_scale(v3_from(za), v.z));
}
static inline v3 transform_backward(const float* xa, const float* ya, const float* za, v3 v) {
    return V3(xa[0] * v.x + xa[1] * v.y + xa[2] * v.z, ya[0] * v.x + ya[1] * v.y + ya[2] * v.z,
              za[0] * v.x + za[1] * v.y + za[2] * v.z);
}
static void transform_g2l(const hiprz_instance* in, ray_t* ray) {
    ray->origin = v3_sub(ray->origin, v3_from(in->position));
    ray->origin = transform_backward(in->x_axis, in->y_axis, in->z_axis, ray->origin);
    ray->origin = v3_div(ray->origin, v3_from(in->scale));
    ray->direction = transform_backward(in->x_axis, in->y_axis, in->z_axis, ray->direction);
    ray->direction = v3_div(ray->direction, v3_from(in->scale));
}
static inline v3 transform_l2g(const hiprz_instance* in, v3 v) {
    v = v3_div(v, v3_from(in->scale));
    return transform_forward(in->x_axis, in->y_axis, in->z_axis, v);
}
static inline v3 transform_l2g_noscale(const hiprz_instance* in, v3 v) {
    return transform_forward(in->x_axis, in->y_axis, in->z_axis, v);
}

/* ------------------------------------------------------------------------------------
 * Closest-hit search — cpu_engine_kernel.cpp:254-352
 * ---------------------------------------------------------------------------------- */
static inline int node_is_leaf(const hiprz_node* n) { return (n->meta & HIPRZ_NODE_LEAF) != 0; }
static inline uint32_t node_count(const hiprz_node* n) { return n->meta & HIPRZ_NODE_COUNT_MASK; }

/* closestIntersection(const Mesh&, ...) :331-352 */
static void closest_mesh(const kctx* k, uint32_t node_idx, ray_t* ray, traversal_t* tr) {
    const hiprz_node* node = &k->s->nodes[node_idx];
    COUNT(k, box_tests, 1);
    if (!box_hit(node->bb_min, node->bb_max, ray)) return;
    if (node_is_leaf(node)) {
        const uint32_t end = node->begin + node_count(node);
        for (uint32_t i = node->begin; i < end; ++i) {
            float t, b1, b2, det;
            COUNT(k, tri_tests, 1);
            if (tri_hit(&k->s->tris[i], ray, &t, &b1, &b2, &det)) {
                ray->far_ = t;
                tr->closest_triangle = (int32_t)i;
                tr->external = det > 0.0f;
                tr->bx = b1;
                tr->by = b2;
            }
        }
    } else {
        closest_mesh(k, node->begin, ray, tr);
        closest_mesh(k, node->begin + 1, ray, tr);
    }
}

/* closestIntersection(const Handle<Instance>&, ...) :299-330 */
static void closest_instance(const kctx* k, uint32_t inst_idx, ray_t* ray, traversal_t* tr) {
    const hiprz_instance* in = &k->s->instances[inst_idx];
    COUNT(k, box_tests, 1);
    if (!box_hit(in->bb_min, in->bb_max, ray)) return;

    ray_t local = *ray;
    transform_g2l(in, &local);

    const float length_factor = v3_mag(local.direction);
    local.near_ *= length_factor;
    local.far_ *= length_factor;
    local.direction = v3_normalized(local.direction);

    const int32_t closest_triangle = tr->closest_triangle;
    tr->closest_triangle = -1;

    closest_mesh(k, in->blas_root, &local, tr);
    if (tr->closest_triangle >= 0) {
        tr->closest_instance = (int32_t)inst_idx;
        ray->near_ = local.near_ / length_factor;
        ray->far_ = local.far_ / length_factor;
    } else {
        tr->closest_triangle = closest_triangle;
    }
}

/* traverseWorld :254-277 */
static void traverse_world(const kctx* k, uint32_t node_idx, ray_t* ray, traversal_t* tr) {
    const hiprz_node* node = &k->s->nodes[node_idx];
    if (node_is_leaf(node)) {
        const uint32_t end = node->begin + node_count(node);
        for (uint32_t i = node->begin; i < end; ++i) closest_instance(k, k->s->tlas_order[i], ray, tr);
    } else {
        const hiprz_node* first = &k->s->nodes[node->begin];
        COUNT(k, box_tests, 1);
        if (box_hit(first->bb_min, first->bb_max, ray)) traverse_world(k, node->begin, ray, tr);
        const hiprz_node* second = &k->s->nodes[node->begin + 1];
        COUNT(k, box_tests, 1);
        if (box_hit(second->bb_min, second->bb_max, ray)) traverse_world(k, node->begin + 1, ray, tr);
    }
}

/* ------------------------------------------------------------------------------------
 * TextureBuffer::fetch — render_parts.hpp:209-221 (point sampling, wrap, v flipped)
 * ---------------------------------------------------------------------------------- */
static void texel_coords(const hiprz_texture* tex, float u, float v, uint32_t* px, uint32_t* py) {
    u += tex->translation[0];
    v += tex->translation[1];
    { /* vec2::Rotate (cuda_render_parts.cuh:373-381), sin/cos hoisted into the descriptor */
        const float xx = u * tex->cos_rotation - v * tex->sin_rotation;
        const float yy = u * tex->sin_rotation + v * tex->cos_rotation;
        u = xx;
        v = yy;
    }
    u *= tex->scale[0];
    v *= tex->scale[1];
    u = fmodf(fmodf(u, 1.0f) + 1.0f, 1.0f);
    v = 1.0f - fmodf(fmodf(v, 1.0f) + 1.0f, 1.0f);
    uint32_t x = (uint32_t)(u * (float)tex->width);
    uint32_t y = (uint32_t)(v * (float)tex->height);
    if (x > tex->width - 1u) x = tex->width - 1u;
    if (y > tex->height - 1u) y = tex->height - 1u;
    *px = x;
    *py = y;
}
static col fetch_rgba8(const kctx* k, int32_t tex_idx, float u, float v) {
    const hiprz_texture* tex = &k->s->textures[tex_idx];
    uint32_t x, y;
    texel_coords(tex, u, v, &x, &y);
    COUNT(k, texel_fetches, 1);
    return col_from_u8(k->s->texels + tex->offset + 4u * ((size_t)y * tex->width + x));
}
static uint8_t fetch_r8(const kctx* k, int32_t tex_idx, float u, float v) {
    const hiprz_texture* tex = &k->s->textures[tex_idx];
    uint32_t x, y;
    texel_coords(tex, u, v, &x, &y);
    COUNT(k, texel_fetches, 1);
    return k->s->texels[tex->offset + ((size_t)y * tex->width + x)];
}
static float fetch_r32f(const kctx* k, int32_t tex_idx, float u, float v) {
    const hiprz_texture* tex = &k->s->textures[tex_idx];
    uint32_t x, y;
    texel_coords(tex, u, v, &x, &y);
    COUNT(k, texel_fetches, 1);
    float f;
    memcpy(&f, k->s->texels + tex->offset + 4u * ((size_t)y * tex->width + x), 4);
    return f;
}

/* fetchColor / fetchMetalness / fetchEmission / fetchRoughness — cpu_engine_kernel.cpp:505-537 */
static col fetch_color(const kctx* k, const hiprz_material* m, float u, float v) {
    col c = col_from_u8(m->color);
    if (m->texture >= 0) c = fetch_rgba8(k, m->texture, u, v);
    c.a = 1.0f - c.a;
    return c;
}
static float fetch_metalness(const kctx* k, const hiprz_material* m, float u, float v) {
    if (m->metalness_map >= 0) return (float)fetch_r8(k, m->metalness_map, u, v) / 255.0f;
    return m->metalness;
}
static float fetch_emission(const kctx* k, const hiprz_material* m, float u, float v) {
    if (m->emission_map >= 0) return fetch_r32f(k, m->emission_map, u, v);
    return m->emission;
}
static float fetch_roughness(const kctx* k, const hiprz_material* m, float u, float v) {
    if (m->roughness_map >= 0) return (float)fetch_r8(k, m->roughness_map, u, v) / 255.0f;
    return m->roughness;
}

/* ------------------------------------------------------------------------------------
 * analyzeIntersection — cpu_engine_kernel.cpp:354-395; Triangle helpers
 * mesh_component.cpp:115-167
 * ---------------------------------------------------------------------------------- */
static void map_normal(const hiprz_tri* tri, const hiprz_tri_attr* at, col map_color, v3* mapped_normal, v3 scale) {
    const v3 v1 = v3_from(tri->v1), v2 = v3_from(tri->v2), vv3 = v3_from(tri->v3);
    const v3 edge1 = v3_mul(v3_sub(v2, v1), scale);
    const v3 edge2 = v3_mul(v3_sub(vv3, v1), scale);
    const float duv1x = at->t2[0] - at->t1[0], duv1y = at->t2[1] - at->t1[1];
    const float duv2x = at->t3[0] - at->t1[0], duv2y = at->t3[1] - at->t1[1];
    *mapped_normal = v3_div(*mapped_normal, scale);

    const float f = 1.0f / (duv1x * duv2y - duv2x * duv1y);
    v3 tangent = v3_normalized(v3_scale(v3_sub(v3_scale(edge1, duv2y), v3_scale(edge2, duv1y)), f));
    tangent = v3_normalized(v3_sub(tangent, v3_scale(*mapped_normal, v3_dot(tangent, *mapped_normal))));
    const v3 bitangent = v3_cross(tangent, *mapped_normal);

    const v3 map_n = v3_sub(v3_scale(V3(map_color.r, map_color.g, map_color.b), 2.0f), V3(1.0f, 1.0f, 1.0f));
    *mapped_normal = v3_add(v3_add(v3_scale(*mapped_normal, map_n.z), v3_scale(tangent, map_n.x)),
                            v3_scale(bitangent, map_n.y));
}

static void analyze_intersection(const kctx* k, const traversal_t* tr, surface_t* sf) {
    const hiprz_scene* s = k->s;
    const hiprz_instance* in = &s->instances[tr->closest_instance];
    const hiprz_tri* tri = &s->tris[tr->closest_triangle];
    const hiprz_tri_attr* at = &s->tri_attrs[tr->closest_triangle];

    /* instance.material(id): m_materials[min(id, 63)], unset -> default material */
    uint32_t mat_slot = tri->material_flags & HIPRZ_TRI_MATERIAL_MASK;
    if (mat_slot > 63u) mat_slot = 63u;
    int32_t mat = -1;
    if (mat_slot < in->material_count) mat = s->inst_materials[in->material_base + mat_slot];
    sf->surface_material = mat < 0 ? HIPRZ_MATERIAL_DEFAULT : (uint32_t)mat;
    sf->behind_material = tr->external ? sf->surface_material : HIPRZ_MATERIAL_WORLD;

    const int has_texcrds = (tri->material_flags & HIPRZ_TRI_HAS_TEXCRDS) != 0;
    if (has_texcrds) { /* texcrdFromBarycenter :115-124 */
        const float b3 = 1.0f - tr->bx - tr->by;
        sf->u = at->t1[0] * b3 + at->t2[0] * tr->bx + at->t3[0] * tr->by;
        sf->v = at->t1[1] * b3 + at->t2[1] * tr->bx + at->t3[1] * tr->by;
    }

    const float external_factor = (float)tr->external * 2.0f - 1.0f;

    if (tri->material_flags & HIPRZ_TRI_HAS_NORMALS) { /* averageNormal :125-131 */
        const v3 n1 = v3_from(at->n1), n2 = v3_from(at->n2), n3 = v3_from(at->n3);
        sf->mapped_normal = v3_normalized(
            v3_add(v3_add(v3_scale(n1, 1.0f - tr->bx - tr->by), v3_scale(n2, tr->bx)), v3_scale(n3, tr->by)));
    } else {
        sf->mapped_normal = v3_from(at->face_normal);
    }
    const hiprz_material* m = &s->materials[sf->surface_material];
    /* The reference indexes mesh.texcrds() unconditionally inside mapNormal (UB for a
     * triangle without texcrds); such triangles are shaded here as if unmapped. */
    if (m->normal_map >= 0 && has_texcrds) {
        map_normal(tri, at, fetch_rgba8(k, m->normal_map, sf->u, sf->v), &sf->mapped_normal, v3_from(in->scale));
        sf->mapped_normal = transform_l2g_noscale(in, sf->mapped_normal);
    } else {
        sf->mapped_normal = transform_l2g(in, sf->mapped_normal);
    }
    sf->mapped_normal = v3_normalized(sf->mapped_normal);
    sf->mapped_normal = v3_scale(sf->mapped_normal, external_factor);

    sf->normal = v3_scale(v3_from(at->face_normal), external_factor);
    sf->normal = transform_l2g(in, sf->normal);
    sf->normal = v3_normalized(sf->normal);
}

/* closestIntersection(RangedRay&, SurfaceProperties&) — :279-298 */
static int closest_intersection(const kctx* k, ray_t* ray, surface_t* sf, traversal_t* tr_out) {
    const hiprz_scene* s = k->s;
    traversal_t tr = {-1, -1, 0.0f, 0.0f, 1};
    if (tr_out) *tr_out = tr;
    if (s->n_instances == 0) return 0;
    const hiprz_node* root = &s->nodes[s->tlas_root];
    COUNT(k, box_tests, 1);
    if (!box_hit(root->bb_min, root->bb_max, ray)) return 0;

    traverse_world(k, s->tlas_root, ray, &tr);
    if (tr_out) *tr_out = tr;

    const int found = tr.closest_instance >= 0;
    if (found) {
        if (sf) analyze_intersection(k, &tr, sf);
    } else if (sf) { /* texcrd of the sky sphere */
        sf->u = -(0.5f + (RZ_ATAN2F(ray->direction.z, ray->direction.x) / (RZ_PI * 2.0f)));
        sf->v = 0.5f + (RZ_ASINF(ray->direction.y) / RZ_PI);
    }
    return found;
}

/* ------------------------------------------------------------------------------------
 * Shadow rays — cpu_engine_kernel.cpp:398-481.  The shadow mask is a ColorF whose four
 * channels are always equal here (1 or 0: "TODO: texture fetch" :465), so only its
 * alpha is carried; V_PL * V_PL.alpha is rebuilt by the caller.
 * ---------------------------------------------------------------------------------- */
static void any_mesh(const kctx* k, uint32_t node_idx, const ray_t* ray, float* mask) {
    if (*mask < 1.0e-4f) return;
    const hiprz_node* node = &k->s->nodes[node_idx];
    COUNT(k, box_tests, 1);
    COUNT(k, shadow_box_tests, 1);
    if (!box_hit(node->bb_min, node->bb_max, ray)) return;
    if (node_is_leaf(node)) {
        const uint32_t end = node->begin + node_count(node);
        for (uint32_t i = node->begin; i < end; ++i) {
            float t, b1, b2, det;
            COUNT(k, tri_tests, 1);
            COUNT(k, shadow_tri_tests, 1);
            if (tri_hit(&k->s->tris[i], ray, &t, &b1, &b2, &det)) {
                *mask *= 0.0f;
                return;
            }
        }
    } else {
        any_mesh(k, node->begin, ray, mask);
        any_mesh(k, node->begin + 1, ray, mask);
    }
}
static float any_instance(const kctx* k, uint32_t inst_idx, const ray_t* ray) {
    const hiprz_instance* in = &k->s->instances[inst_idx];
    COUNT(k, box_tests, 1);
    COUNT(k, shadow_box_tests, 1);
    if (!box_hit(in->bb_min, in->bb_max, ray)) return 1.0f;
    ray_t local = *ray;
    transform_g2l(in, &local);
    const float len = v3_mag(local.direction);
    local.near_ *= len;
    local.far_ *= len;
    local.direction = v3_normalized(local.direction);
    float mask = 1.0f;
    any_mesh(k, in->blas_root, &local, &mask);
    return mask;
}
static void any_world(const kctx* k, uint32_t node_idx, const ray_t* ray, float* mask) {
    const hiprz_node* node = &k->s->nodes[node_idx];
    if (node_is_leaf(node)) {
        const uint32_t end = node->begin + node_count(node);
        for (uint32_t i = node->begin; i < end; ++i) {
            *mask *= any_instance(k, k->s->tlas_order[i], ray);
            if (*mask < 1.0e-4f) return;
        }
    } else {
        const hiprz_node* first = &k->s->nodes[node->begin];
        COUNT(k, box_tests, 1);
    COUNT(k, shadow_box_tests, 1);
        if (box_hit(first->bb_min, first->bb_max, ray)) {
            any_world(k, node->begin, ray, mask);
            if (*mask < 1.0e-4f) return;
        }
        const hiprz_node* second = &k->s->nodes[node->begin + 1];
        COUNT(k, box_tests, 1);
    COUNT(k, shadow_box_tests, 1);
        if (box_hit(second->bb_min, second->bb_max, ray)) any_world(k, node->begin + 1, ray, mask);
    }
}
static float any_intersection(const kctx* k, const ray_t* ray) {
    const hiprz_scene* s = k->s;
    COUNT(k, shadow_rays, 1);
    if (s->n_instances == 0) return 0.0f;
    const hiprz_node* root = &s->nodes[s->tlas_root];
    COUNT(k, box_tests, 1);
    COUNT(k, shadow_box_tests, 1);
    if (!box_hit(root->bb_min, root->bb_max, ray)) return 1.0f;
    float mask = 1.0f;
    any_world(k, s->tlas_root, ray, &mask);
    return mask;
}

/* ------------------------------------------------------------------------------------
 * Helper functions — cpu_render_utils.cpp:29-170
 * ---------------------------------------------------------------------------------- */
static inline v3 reflect_vector(v3 vI, v3 vN) { /* :29-32 */
    return v3_add(v3_scale(v3_scale(vN, -2.0f), v3_dot(vN, vI)), vI);
}
static inline v3 halfway_vector(v3 vI, v3 vR) { /* :33-36 */
    return v3_normalized(v3_add(v3_neg(vI), vR));
}
static void local_coordinate(v3 vN, v3* vX, v3* vY) { /* :74-83 */
    const int b = fabsf(vN.x) > fabsf(vN.y);
    vX->x = (float)(!b);
    vX->y = (float)(b);
    vX->z = 0.0f;
    *vY = v3_cross(vN, *vX);
    *vX = v3_cross(vN, *vY);
}
static v3 cosine_sample_hemisphere(float r1, float r2, v3 vN) { /* :85-101 */
    v3 vX, vY;
    local_coordinate(vN, &vX, &vY);
    const float phi = r1 * 6.283185f;
    const float theta = r2;
    const float sqrt_theta = sqrtf(theta);
    const v3 a = v3_scale(v3_scale(vX, sqrt_theta), RZ_COSF(phi));
    const v3 b = v3_scale(v3_scale(vY, sqrt_theta), RZ_SINF(phi));
    const v3 c = v3_scale(vN, sqrtf(1.0f - theta));
    return v3_add(v3_add(a, b), c);
}
static v3 sample_sphere(float r1, float r2, v3 vN) { /* :102-119 */
    v3 vX, vY;
    local_coordinate(vN, &vX, &vY);
    const float phi = r1 * 6.283185f;
    const float theta = RZ_ACOSF(1.0f - 2.0f * r2);
    const float sin_theta = RZ_SINF(theta);
    const v3 a = v3_scale(v3_scale(vX, sin_theta), RZ_COSF(phi));
    const v3 b = v3_scale(v3_scale(vY, sin_theta), RZ_SINF(phi));
    const v3 c = v3_scale(vN, RZ_COSF(theta));
    return v3_add(v3_add(a, b), c);
}
static inline v3 sample_hemisphere(float r1, float r2, v3 vN) { /* :120-126 */
    return sample_sphere(r1, r2 * 0.5f, vN);
}
static v3 sample_disk(float r1, float r2, v3 vN, float radius) { /* :127-138 */
    v3 vX, vY;
    local_coordinate(vN, &vX, &vY);
    const float phi = r1 * 2.0f * RZ_PI;
    const float mag = sqrtf(r2);
    return v3_scale(v3_scale(v3_add(v3_scale(vX, RZ_SINF(phi)), v3_scale(vY, RZ_COSF(phi))), mag), radius);
}
static float fresnel_specular_ratio(v3 vN, v3 vI, float n1, float n2, float* fx, float* fy) { /* :141-159 */
    const float ratio = n1 / n2;
    const float cosi = fabsf(v3_dot(vI, vN));
    const float sin2_t = ratio * ratio * (1.0f - cosi * cosi);
    if (sin2_t >= 1.0f) return 1.0f;
    const float cost = sqrtf(1.0f - sin2_t);
    const float Rp = ((n1 * cosi) - (n2 * cost)) / ((n1 * cosi) + (n2 * cost));
    const float Rs = ((n2 * cosi) - (n1 * cost)) / ((n2 * cosi) + (n1 * cost));
    *fx = ratio;
    *fy = ratio * cosi - cost;
    return (Rs * Rs + Rp * Rp) / 2.0f;
}
float rzo_fresnel(const float n[3], const float i[3], float n1, float n2, float factors[2]) {
    return fresnel_specular_ratio(v3_from(n), v3_from(i), n1, n2, &factors[0], &factors[1]);
}
void rzo_cosine_sample_hemisphere(float r1, float r2, const float n[3], float out[3]) {
    v3 r = cosine_sample_hemisphere(r1, r2, v3_from(n));
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
void rzo_sample_sphere(float r1, float r2, const float n[3], float out[3]) {
    v3 r = sample_sphere(r1, r2, v3_from(n));
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
void rzo_sample_disk(float r1, float r2, const float n[3], float radius, float out[3]) {
    v3 r = sample_disk(r1, r2, v3_from(n), radius);
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}

/* ------------------------------------------------------------------------------------
 * BRDF — cpu_engine_kernel.cpp:556-594
 * ---------------------------------------------------------------------------------- */
static float ndf(v3 vN, v3 vH, float roughness) { /* :585-590 */
    const float d = v3_dot(vN, vH);
    const float b = (d * d) * (roughness - 1.0f) + 1.0001f;
    return (roughness + 1.0e-5f) / (b * b);
}
static float attenuation(float cos_angle, float roughness) { /* :591-594 */
    return cos_angle / ((cos_angle * (1.0f - roughness)) + roughness);
}
static float brdf(const kctx* k, const ray_t* ray, const surface_t* sf, v3 vPL) { /* :556-579 */
    if (k->s->materials[sf->surface_material].scattering > 0.0f) return 1.0f;
    const float vN_dot_vO = v3_dot(sf->mapped_normal, vPL);
    if (vN_dot_vO <= 0.0f) return 0.0f;
    const float vN_dot_vI = v3_dot(sf->mapped_normal, v3_neg(ray->direction));
    if (vN_dot_vI <= 0.0f) return 0.0f;

    const v3 vH = halfway_vector(ray->direction, vPL);
    const float nd = ndf(sf->mapped_normal, vH, sf->roughness);
    const float atten_i = attenuation(vN_dot_vI, sf->roughness);
    const float atten_o = attenuation(vN_dot_vO, sf->roughness);
    const float atten = atten_i * atten_o;

    const float diffuse = vN_dot_vO * (float)(sf->color.a == 0.0f);
    const float specular = nd * atten / (vN_dot_vI * vN_dot_vO);
    return lerpf(diffuse, specular * vN_dot_vO, sf->reflectance);
}
static inline col brdf_color(const surface_t* sf) { /* :580-583 */
    return col_lerp(sf->color, col_splat(1.0f), sf->reflectance);
}

/* ------------------------------------------------------------------------------------
 * Direction sampling — cpu_engine_kernel.cpp:596-687
 * ---------------------------------------------------------------------------------- */
static v3 sample_direction(const kctx* k, ray_t* ray, surface_t* sf, rng_t* rng) {
    if (sf->color.a > 0.0f) {
        if (k->s->materials[sf->surface_material].scattering > 0.0f) { /* sampleScatteringDirection :681-687 */
            const float u1 = rng_unsigned(rng);
            const float u2 = rng_unsigned(rng);
            const v3 vO = sample_sphere(u1, u2, ray->direction);
            sf->tint_factor = sf->metalness;
            return vO;
        }
        /* sampleTransmissionDirection :655-680 */
        if (sf->fresnel < rng_unsigned(rng)) {
            const v3 vO = v3_add(v3_scale(ray->direction, sf->refr_x), v3_scale(sf->mapped_normal, sf->refr_y));
            ray->material = sf->behind_material;
            sf->normal = v3_neg(sf->normal);
            sf->tint_factor = 1.0f;
            return vO;
        } else {
            v3 vO = reflect_vector(ray->direction, sf->mapped_normal);
            const float d = v3_dot(vO, sf->normal);
            if (d < 0.0f) vO = v3_add(vO, v3_scale(v3_scale(sf->normal, -2.0f), d));
            sf->tint_factor = sf->metalness;
            return vO;
        }
    }
    if (rng_unsigned(rng) > sf->reflectance) { /* sampleDiffuseDirection :622-635 */
        const float u1 = rng_unsigned(rng);
        const float u2 = rng_unsigned(rng);
        v3 vO = cosine_sample_hemisphere(u1, u2, sf->mapped_normal);
        const float d = v3_similarity(vO, sf->normal);
        if (d < 0.0f) vO = v3_add(vO, v3_scale(v3_scale(sf->normal, -2.0f), d));
        sf->tint_factor = 1.0f;
        return vO;
    } else { /* sampleGlossyDirection :636-654 */
        const float u1 = rng_unsigned(rng);
        const float u2 = rng_unsigned(rng);
        const v3 vH = sample_hemisphere(u1, 1.0f - RZ_POWF(u2 + 1.0e-5f, sf->roughness), sf->mapped_normal);
        v3 vO = reflect_vector(ray->direction, vH);
        const float d = v3_similarity(vO, sf->normal);
        if (d < 0.0f) vO = v3_add(vO, v3_scale(v3_scale(sf->normal, -2.0f), d));
        sf->tint_factor = sf->metalness;
        return vO;
    }
}

/* ------------------------------------------------------------------------------------
 * Next-event estimation — cpu_engine_kernel.cpp:690-865
 * ---------------------------------------------------------------------------------- */
static ray_t shadow_ray(v3 origin, v3 direction, float near_, float far_) {
    ray_t r;
    r.origin = origin;
    r.direction = v3_normalized(direction); /* Ray ctor normalises, cpu_render_utils.hpp:41-46 */
    r.near_ = near_;
    r.far_ = far_;
    r.material = 0;
    r.color = col_splat(1.0f);
    return r;
}

/* directLightSampling :745-791 (with directLightSampleDirection :840-861, SolidAngle :862-865) */
static col direct_light_sampling(const kctx* k, const ray_t* ray, v3 point, v3 next_dir, const surface_t* sf,
                                 float vS_pdf, rng_t* rng) {
    const hiprz_scene* s = k->s;
    const uint32_t light_count = s->n_direct_lights;
    const uint32_t sample_count = k->cfg->direct_samples;
    col total = col_splat(0.0f);
    if (light_count == 0) return total;
    for (uint32_t i = 0; i < sample_count; ++i) {
        uint32_t li = (uint32_t)(rng_unsigned(rng) * (float)light_count);
        if (li >= light_count) li = light_count - 1u;
        const hiprz_direct_light* light = &s->direct_lights[li];
        COUNT(k, light_samples, 1);

        float Se = 0.0f;
        v3 vPL;
        {
            const v3 ldir = v3_from(light->direction);
            const float dot = v3_dot(next_dir, v3_neg(ldir));
            const float cos_angle = light->cos_angular_size;
            if (dot > cos_angle) {
                Se = light->emission;
                vPL = next_dir;
            } else {
                const float u1 = rng_unsigned(rng);
                const float u2 = rng_unsigned(rng);
                vPL = sample_sphere(u1, u2 * 0.5f * (1.0f - cos_angle), v3_neg(ldir));
            }
        }
        const float b = brdf(k, ray, sf, v3_normalized(vPL));
        const col bc = brdf_color(sf);
        const float solid_angle = 2.0f * RZ_PI * (1.0f - light->cos_angular_size);

        const float L_pdf = 1.0f / solid_angle;
        const float vSw = vS_pdf / (vS_pdf + L_pdf);
        const float Lw = 1.0f - vSw;
        const float Le = light->emission * solid_angle * b;
        const float radiance = (Le * Lw + Se * vSw);
        if (radiance < 1.0e-4f) continue;

        const ray_t sr = shadow_ray(point, vPL, 0.0f, FLT_MAX);
        const float V = any_intersection(k, &sr);
        const col V_PL = col_splat(V);
        total = col_add(total,
                        col_scale(col_mul(col_scale(col_mul(col_from_u8(light->color), bc), radiance), V_PL), V_PL.a));
    }
    const float pdf = (float)sample_count / (float)light_count;
    return col_divs(total, pdf);
}

/* spotLightSampling :690-744 (with spotLightSampleDirection :805-828, SolidAngle :829-834,
 * BeamIllumination :835-838; rayPointCalculation cpu_render_utils.cpp:48-72) */
static col spot_light_sampling(const kctx* k, const ray_t* ray, v3 point, v3 next_dir, const surface_t* sf,
                               float vS_pdf, rng_t* rng) {
    const hiprz_scene* s = k->s;
    const uint32_t light_count = s->n_spot_lights;
    const uint32_t sample_count = k->cfg->spot_samples;
    col total = col_splat(0.0f);
    if (light_count == 0) return total;
    for (uint32_t i = 0; i < sample_count; ++i) {
        uint32_t li = (uint32_t)(rng_unsigned(rng) * (float)light_count);
        if (li >= light_count) li = light_count - 1u;
        const hiprz_spot_light* light = &s->spot_lights[li];
        COUNT(k, light_samples, 1);
        const v3 lpos = v3_from(light->position);

        float Se = 0.0f;
        v3 vPL;
        {
            const v3 rd = v3_normalized(next_dir); /* Ray(point, vS) normalises */
            const v3 vOP = v3_sub(lpos, point);
            const float dOP = v3_mag(vOP);
            const float vOP_dot_vD = v3_dot(vOP, rd);
            const float dPQ = sqrtf(dOP * dOP - vOP_dot_vD * vOP_dot_vD);
            if (dPQ < light->size && vOP_dot_vD > 0.0f) {
                Se = light->emission;
                const float dOQ = sqrtf(dOP * dOP - dPQ * dPQ);
                vPL = v3_scale(next_dir, fmaxf(dOQ, 1.0e-4f));
            } else {
                const float u1 = rng_unsigned(rng);
                const float u2 = rng_unsigned(rng);
                vPL = v3_sub(v3_add(sample_disk(u1, u2, v3_divs(vOP, dOP), light->size), lpos), point);
            }
        }
        const float dPL = v3_mag(vPL);

        const float b = brdf(k, ray, sf, v3_divs(vPL, dPL));
        if (b < 1.0e-4f) continue;
        const col bc = brdf_color(sf);
        const float A = light->size * light->size * RZ_PI;
        const float d1 = dPL + 1.0f;
        const float solid_angle = A / (d1 * d1);
        const float sctr_factor = RZ_EXPF(-dPL * s->materials[ray->material].scattering);

        const float beam = (float)(light->cos_angle < v3_similarity(v3_neg(vPL), v3_from(light->direction)));
        if (beam < 1.0e-4f) continue;

        const float L_pdf = 1.0f / solid_angle;
        const float vSw = vS_pdf / (vS_pdf + L_pdf);
        const float Lw = 1.0f - vSw;
        const float Le = light->emission * solid_angle * b;
        const float radiance = (Le * Lw + Se * vSw) * sctr_factor * beam;
        if (radiance < 1.0e-4f) continue;

        const ray_t sr = shadow_ray(point, vPL, 0.0f, dPL);
        const float V = any_intersection(k, &sr);
        const col V_PL = col_splat(V);
        total = col_add(total,
                        col_scale(col_mul(col_scale(col_mul(col_from_u8(light->color), bc), radiance), V_PL), V_PL.a));
    }
    const float pdf = (float)sample_count / (float)light_count;
    return col_divs(total, pdf);
}

/* directIllumination :792-803 */
static col direct_illumination(const kctx* k, const ray_t* ray, v3 point, v3 next_dir, const surface_t* sf,
                               rng_t* rng) {
    const float vS_pdf = brdf(k, ray, sf, next_dir);
    const col d = direct_light_sampling(k, ray, point, next_dir, sf, vS_pdf, rng);
    const col sp = spot_light_sampling(k, ray, point, next_dir, sf, vS_pdf, rng);
    return col_add(d, sp);
}

/* ------------------------------------------------------------------------------------
 * traceRay — cpu_engine_kernel.cpp:113-178
 * ---------------------------------------------------------------------------------- */
typedef struct {
    col final_color;
    uint8_t path_depth;
} tracing_state;
typedef struct {
    v3 point, next_direction;
} tracing_result;

static tracing_result trace_ray(const kctx* k, tracing_state* ts, ray_t* ray, rng_t* rng) {
    tracing_result result;
    memset(&result, 0, sizeof result);
    surface_t sf;
    memset(&sf, 0, sizeof sf);
    sf.surface_material = sf.behind_material = HIPRZ_MATERIAL_WORLD;
    sf.fresnel = 1.0f;

    COUNT(k, segments, 1);
    const int any_hit = closest_intersection(k, ray, &sf, NULL);
    const hiprz_material* sm = &k->s->materials[sf.surface_material];

    sf.color = fetch_color(k, sm, sf.u, sf.v);
    sf.emission = fetch_emission(k, sm, sf.u, sf.v);

    if (sf.emission > 0.0f) ts->final_color = col_add(ts->final_color, col_scale(col_mul(ray->color, sf.color), sf.emission));

    if (!any_hit) {
        ts->path_depth = 255; /* endPath, cpu_engine_kernel.hpp:24-27 */
        return result;
    }
    COUNT(k, hits, 1);
    ++ts->path_depth;

    sf.metalness = fetch_metalness(k, sm, sf.u, sf.v);
    sf.roughness = fetch_roughness(k, sm, sf.u, sf.v);

    sf.fresnel = fresnel_specular_ratio(sf.mapped_normal, ray->direction, k->s->materials[ray->material].ior,
                                        k->s->materials[sf.behind_material].ior, &sf.refr_x, &sf.refr_y);
    sf.reflectance = lerpf(sf.fresnel, 1.0f, sf.metalness);

    result.next_direction = sample_direction(k, ray, &sf, rng);
    result.point = v3_add(v3_add(ray->origin, v3_scale(ray->direction, ray->far_)),
                          v3_scale(sf.normal, 0.0001f * ray->far_));

    {
        const col direct = direct_illumination(k, ray, result.point, result.next_direction, &sf, rng);
        ts->final_color = col_add(
            ts->final_color, col_mul(col_mul(direct, ray->color), col_lerp(col_splat(1.0f), sf.color, sf.metalness)));
    }

    ray->color = col_lerp(ray->color, col_mul(ray->color, sf.color), sf.tint_factor); /* ColorF::Blend */
    return result;
}

/* ------------------------------------------------------------------------------------
 * Camera rays — cpu_engine_kernel.cpp:180-252
 * ---------------------------------------------------------------------------------- */
static void screen_direction(const hiprz_camera* c, uint32_t px, uint32_t py, float* dx, float* dy) {
    const float tana = c->tan_half_fov; /* std::tanf(fov * 0.5f), hoisted */
    *dx = ((((float)px + 0.5f) / (float)c->width) - 0.5f) * tana;
    *dy = ((((float)py + 0.5f) / (float)c->height) - 0.5f) * (-tana / c->aspect_ratio);
}
static void generate_simple_ray(const hiprz_camera* c, ray_t* ray, uint32_t px, uint32_t py) {
    ray->origin = V3(0.0f, 0.0f, 0.0f);
    float dx, dy;
    screen_direction(c, px, py, &dx, &dy);
    ray->direction = V3(dx, dy, 1.0f);

    ray->origin = transform_forward(c->x_axis, c->y_axis, c->z_axis, ray->origin);
    ray->origin = v3_add(ray->origin, v3_from(c->position));
    ray->direction = transform_forward(c->x_axis, c->y_axis, c->z_axis, ray->direction);
    ray->direction = v3_normalized(ray->direction);

    ray->near_ = c->near_far[0];
    ray->far_ = c->near_far[1];
}
static void generate_antialiased_ray(const hiprz_camera* c, ray_t* ray, uint32_t px, uint32_t py, rng_t* rng) {
    float dx, dy;
    screen_direction(c, px, py, &dx, &dy);
    ray->direction = V3(dx, dy, 1.0f);

    ray->direction.x += ((0.5f / (float)c->width) * rng_signed(rng));
    ray->direction.y += ((0.5f / (float)c->width) * rng_signed(rng)); /* (sic) x resolution, :227-228 */

    const v3 focal_point = v3_scale(ray->direction, c->focal_distance);

    const float aperture_angle = rng_unsigned(rng) * 2.0f * RZ_PI;
    const float aperture_sample = sqrtf(rng_unsigned(rng)) * c->aperture;
    ray->origin = V3(aperture_sample * RZ_SINF(aperture_angle), aperture_sample * RZ_COSF(aperture_angle), 0.0f);

    ray->direction = v3_sub(focal_point, ray->origin);

    ray->origin = transform_forward(c->x_axis, c->y_axis, c->z_axis, ray->origin);
    ray->origin = v3_add(ray->origin, v3_from(c->position));
    ray->direction = transform_forward(c->x_axis, c->y_axis, c->z_axis, ray->direction);
    ray->direction = v3_normalized(ray->direction);

    ray->near_ = c->near_far[0];
    ray->far_ = c->near_far[1];
}

/* ------------------------------------------------------------------------------------
 * CameraContext::setRay / getRay — cpu_engine_renderer.cpp:40-53
 * ---------------------------------------------------------------------------------- */
static void set_ray(rzo_context* ctx, size_t p, const ray_t* ray) {
    ctx->ray_origin[3 * p + 0] = ray->origin.x;
    ctx->ray_origin[3 * p + 1] = ray->origin.y;
    ctx->ray_origin[3 * p + 2] = ray->origin.z;
    ctx->ray_direction[3 * p + 0] = ray->direction.x;
    ctx->ray_direction[3 * p + 1] = ray->direction.y;
    ctx->ray_direction[3 * p + 2] = ray->direction.z;
    ctx->ray_material[p] = ray->material;
    ctx->ray_color[4 * p + 0] = ray->color.r;
    ctx->ray_color[4 * p + 1] = ray->color.g;
    ctx->ray_color[4 * p + 2] = ray->color.b;
    ctx->ray_color[4 * p + 3] = ray->color.a;
}
static ray_t get_ray(const rzo_context* ctx, size_t p) {
    ray_t r;
    r.origin = v3_from(&ctx->ray_origin[3 * p]);
    r.direction = v3_normalized(v3_from(&ctx->ray_direction[3 * p])); /* SceneRay ctor -> Ray ctor normalises */
    r.near_ = 0.0f;
    r.far_ = FLT_MAX;
    r.material = ctx->ray_material[p];
    r.color = COL(ctx->ray_color[4 * p], ctx->ray_color[4 * p + 1], ctx->ray_color[4 * p + 2], ctx->ray_color[4 * p + 3]);
    return r;
}

/* Tone map of one pixel — cpu_engine_renderer.cpp:224-235 */
static void tonemap(col color, float aperture, float exposure_time, uint8_t* out) {
    const float aperture_area = aperture * aperture * RZ_PI;
    color = col_divs(color, color.a == 0.0f ? 1.0f : color.a);
    color = col_scale(color, aperture_area);
    color = col_scale(color, exposure_time);
    color = col_scale(color, 1.0e5f);
    color = col_div(color, col_add(color, col_splat(1.0f)));
    out[0] = (uint8_t)(color.r * 255.0f);
    out[1] = (uint8_t)(color.g * 255.0f);
    out[2] = (uint8_t)(color.b * 255.0f);
    out[3] = 255;
}
void rzo_tonemap_pixel(const float rgba[4], float aperture, float exposure_time, uint8_t out[4]) {
    tonemap(COL(rgba[0], rgba[1], rgba[2], rgba[3]), aperture, exposure_time, out);
}

/* Harness seeding convention (after cuda_render_kernel.cu:24-28, 86-90):
 * RNG(vec2(x/W, y/H), seeds_of_pass[(pixel_idx + depth) % 256]). */
static rng_t pixel_rng(const kctx* k, uint32_t pass, uint32_t x, uint32_t y, uint32_t depth) {
    const uint32_t idx = y * k->cam->width + x;
    const float seed = rzo_seed_value(k->cfg->seed, pass, (idx + depth) & 255u);
    return rng_make((float)x / (float)k->cam->width, (float)y / (float)k->cam->height, seed);
}

/* renderFirstPass — cpu_engine_kernel.cpp:15-57 */
static col render_first_pass(const kctx* k, rzo_context* ctx, uint32_t x, uint32_t y) {
    const size_t p = (size_t)y * ctx->width + x;
    rng_t rng = pixel_rng(k, 0u, x, y, 0u);

    ray_t ray;
    memset(&ray, 0, sizeof ray);
    ray.color = col_splat(1.0f);
    ray.material = HIPRZ_MATERIAL_WORLD;
    generate_simple_ray(k->cam, &ray, x, y);

    tracing_state ts = {col_splat(0.0f), 0u};
    const tracing_result result = trace_ray(k, &ts, &ray, &rng);
    const int path_continues = ts.path_depth < k->cfg->max_depth;

    ctx->depth[p] = ray.far_;

    ts.final_color.a = (float)(!path_continues);
    ctx->image[4 * p + 0] = ts.final_color.r;
    ctx->image[4 * p + 1] = ts.final_color.g;
    ctx->image[4 * p + 2] = ts.final_color.b;
    ctx->image[4 * p + 3] = ts.final_color.a;

    if (path_continues) { /* TracingResult::repositionRay, cpu_render_utils.hpp:152-157 */
        ray.origin = result.point;
        ray.direction = result.next_direction;
        ray.near_ = 0.0f;
        ray.far_ = FLT_MAX;
    } else {
        COUNT(k, finished, 1);
        generate_antialiased_ray(k->cam, &ray, x, y, &rng);
        ray.material = HIPRZ_MATERIAL_WORLD;
        ray.color = col_splat(1.0f);
    }
    set_ray(ctx, p, &ray);
    ctx->path_depth[p] = path_continues ? ts.path_depth : 0u;
    return ts.final_color;
}

/* renderCumulativePass — cpu_engine_kernel.cpp:58-101 */
static col render_cumulative_pass(const kctx* k, rzo_context* ctx, uint32_t pass, uint32_t x, uint32_t y) {
    const size_t p = (size_t)y * ctx->width + x;
    tracing_state ts = {col_splat(0.0f), ctx->path_depth[p]};
    rng_t rng = pixel_rng(k, pass, x, y, ts.path_depth);

    ray_t ray = get_ray(ctx, p);
    if (ts.path_depth == 0) {
        ray.near_ = k->cam->near_far[0];
        ray.far_ = k->cam->near_far[1];
    }
    const tracing_result result = trace_ray(k, &ts, &ray, &rng);
    const int path_continues = ts.path_depth < k->cfg->max_depth;

    col value = COL(ctx->image[4 * p], ctx->image[4 * p + 1], ctx->image[4 * p + 2], ctx->image[4 * p + 3]);
    value.r += ts.final_color.r;
    value.g += ts.final_color.g;
    value.b += ts.final_color.b;
    value.a += (float)(!path_continues);
    ctx->image[4 * p + 0] = value.r;
    ctx->image[4 * p + 1] = value.g;
    ctx->image[4 * p + 2] = value.b;
    ctx->image[4 * p + 3] = value.a;

    if (path_continues) {
        ray.origin = result.point;
        ray.direction = result.next_direction;
        ray.near_ = 0.0f;
        ray.far_ = FLT_MAX;
    } else {
        COUNT(k, finished, 1);
        generate_antialiased_ray(k->cam, &ray, x, y, &rng);
        ray.material = HIPRZ_MATERIAL_WORLD;
        ray.color = col_splat(1.0f);
    }
    set_ray(ctx, p, &ray);
    ctx->path_depth[p] = path_continues ? ts.path_depth : 0u;
    return value;
}

/* ------------------------------------------------------------------------------------
 * Renderer::renderCameraView — cpu_engine_renderer.cpp:186-279: 128x128 tiles pulled from
 * an atomic counter by one worker per hardware thread (OpenMP dynamic schedule here).
 * ---------------------------------------------------------------------------------- */
static void counters_add(hiprz_counters* a, const hiprz_counters* b) {
    a->segments += b->segments;
    a->box_tests += b->box_tests;
    a->tri_tests += b->tri_tests;
    a->hits += b->hits;
    a->shadow_rays += b->shadow_rays;
    a->light_samples += b->light_samples;
    a->texel_fetches += b->texel_fetches;
    a->finished += b->finished;
    a->shadow_box_tests += b->shadow_box_tests;
    a->shadow_tri_tests += b->shadow_tri_tests;
}

void rzo_render_pass(const hiprz_scene* scene, const hiprz_camera* camera, const hiprz_config* config,
                     rzo_context* ctx, int threads, hiprz_counters* counters) {
    const uint32_t W = ctx->width, H = ctx->height;
    const uint32_t x_blocks = ((W - 1) / 128u) + 1, y_blocks = ((H - 1) / 128u) + 1;
    const int block_count = (int)(x_blocks * y_blocks);
    const int first = ctx->passes == 0;
    const uint32_t pass = ctx->passes;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_num_procs();
#else
    threads = 1;
#endif
    if (counters) memset(counters, 0, sizeof *counters);

#pragma omp parallel num_threads(threads)
    {
        hiprz_counters local;
        memset(&local, 0, sizeof local);
        kctx k = {scene, camera, config, counters ? &local : NULL};
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < block_count; ++b) {
            const uint32_t by = (uint32_t)b / x_blocks, bx = (uint32_t)b % x_blocks;
            const uint32_t x0 = bx * 128u, y0 = by * 128u;
            const uint32_t x1 = x0 + 128u < W ? x0 + 128u : W, y1 = y0 + 128u < H ? y0 + 128u : H;
            for (uint32_t y = y0; y != y1; ++y) {
                for (uint32_t x = x0; x != x1; ++x) {
                    const col c = first ? render_first_pass(&k, ctx, x, y) : render_cumulative_pass(&k, ctx, pass, x, y);
                    tonemap(c, camera->aperture, camera->exposure_time, &ctx->rgba8[4 * ((size_t)y * W + x)]);
                }
            }
        }
        if (counters) {
#pragma omp critical
            counters_add(counters, &local);
        }
    }
    ctx->passes += 1;
    ctx->traced_rays += (uint64_t)W * H; /* cpu_engine_renderer.cpp:173 */
}

/* Kernel::rayCast / worldRayCast — cpu_engine_kernel.cpp:102-111, 483-501 */
void rzo_pick(const hiprz_scene* scene, const hiprz_camera* camera, const rzo_context* ctx, uint32_t x,
              uint32_t y, int32_t* instance_out, int32_t* material_out) {
    hiprz_config cfg;
    memset(&cfg, 0, sizeof cfg);
    kctx k = {scene, camera, &cfg, NULL};
    ray_t ray;
    memset(&ray, 0, sizeof ray);
    generate_simple_ray(camera, &ray, x, y);
    const float depth = ctx->depth[(size_t)y * ctx->width + x];
    ray.near_ = depth * 0.99f;
    ray.far_ = depth * 1.01f;
    traversal_t tr;
    *instance_out = -1;
    *material_out = -1;
    closest_intersection(&k, &ray, NULL, &tr);
    if (tr.closest_instance >= 0) {
        const hiprz_instance* in = &scene->instances[tr.closest_instance];
        uint32_t slot = scene->tris[tr.closest_triangle].material_flags & HIPRZ_TRI_MATERIAL_MASK;
        if (slot > 63u) slot = 63u;
        *instance_out = tr.closest_instance;
        *material_out = slot < in->material_count ? scene->inst_materials[in->material_base + slot] : -1;
    }
}

/* ------------------------------------------------------------------------------------
 * Context
 * ---------------------------------------------------------------------------------- */
rzo_context* rzo_context_create(uint32_t width, uint32_t height) {
    rzo_context* c = (rzo_context*)calloc(1, sizeof *c);
    const size_t n = (size_t)width * height;
    c->width = width;
    c->height = height;
    c->image = (float*)calloc(n * 4, sizeof(float));
    c->path_depth = (uint8_t*)calloc(n, 1);
    c->ray_origin = (float*)calloc(n * 3, sizeof(float));
    c->ray_direction = (float*)calloc(n * 3, sizeof(float));
    c->ray_material = (uint32_t*)calloc(n, sizeof(uint32_t));
    c->ray_color = (float*)calloc(n * 4, sizeof(float));
    c->depth = (float*)calloc(n, sizeof(float));
    c->rgba8 = (uint8_t*)calloc(n * 4, 1);
    return c;
}
void rzo_context_destroy(rzo_context* c) {
    if (!c) return;
    free(c->image);
    free(c->path_depth);
    free(c->ray_origin);
    free(c->ray_direction);
    free(c->ray_material);
    free(c->ray_color);
    free(c->depth);
    free(c->rgba8);
    free(c);
}
void rzo_context_reset(rzo_context* c) {
    memset(c->image, 0, (size_t)c->width * c->height * 4 * sizeof(float));
    c->passes = 0;
    c->traced_rays = 0;
}

/* ====================================================================================
 * Host tree builders — TreeNode::construct (bvh_tree_node.hpp:117-215) and
 * ComponentTreeNode::construct (component_container.hpp:259-363).  The two are the same
 * algorithm with different leaf sizes (4 / 8) and root-stays-a-leaf thresholds (8 / 32),
 * over "items" that only expose a bounding box.
 * ================================================================================== */
typedef struct {
    float mn[3], mx[3];
} bbox;

typedef struct bnode {
    struct bnode *first, *second; /* NULL,NULL = leaf */
    uint32_t ptype;
    uint32_t* items; /* leaf */
    uint32_t n_items;
    bbox bb;
} bnode;

/* BoundingBox(p1,p2) uses std::min/std::max (render_parts.cpp:166-177); extendBy uses
 * strict compares (:190-197). */
static inline float std_min(float a, float b) { return b < a ? b : a; }
static inline float std_max(float a, float b) { return a < b ? b : a; }
static bbox bbox_from2(const float* p1, const float* p2) {
    bbox b;
    for (int i = 0; i < 3; ++i) {
        b.mn[i] = std_min(p1[i], p2[i]);
        b.mx[i] = std_max(p1[i], p2[i]);
    }
    return b;
}
static void bbox_extend_point(bbox* b, const float* p) {
    for (int i = 0; i < 3; ++i) {
        if (b->mn[i] > p[i]) b->mn[i] = p[i];
        if (b->mx[i] < p[i]) b->mx[i] = p[i];
    }
}
static void bbox_extend(bbox* b, const bbox* o) {
    for (int i = 0; i < 3; ++i) {
        if (b->mn[i] > o->mn[i]) b->mn[i] = o->mn[i];
        if (b->mx[i] < o->mx[i]) b->mx[i] = o->mx[i];
    }
}
static inline void bbox_centroid(const bbox* b, float* c) { /* (min + max) * 0.5f */
    for (int i = 0; i < 3; ++i) c[i] = (b->mn[i] + b->mx[i]) * 0.5f;
}

/* std::partition on bidirectional iterators (libstdc++ __partition, bidirectional form;
 * MSVC's is the same scan-from-both-ends-and-swap algorithm). */
typedef int (*pred_fn)(const bbox* item_bb, const void* arg);
static uint32_t* partition_items(uint32_t* first, uint32_t* last, const bbox* bbs, pred_fn pred, const void* arg) {
    for (;;) {
        for (;;) {
            if (first == last) return first;
            else if (pred(&bbs[*first], arg)) ++first;
            else break;
        }
        --last;
        for (;;) {
            if (first == last) return first;
            else if (!pred(&bbs[*last], arg)) --last;
            else break;
        }
        uint32_t t = *first;
        *first = *last;
        *last = t;
        ++first;
    }
}
static int pred_smaller(const bbox* b, const void* arg) {
    const float* node_size = (const float*)arg;
    const float sx = b->mx[0] - b->mn[0], sy = b->mx[1] - b->mn[1], sz = b->mx[2] - b->mn[2];
    return sx < node_size[0] && sy < node_size[1] && sz < node_size[2];
}
typedef struct {
    int axis;
    float plane;
} plane_arg;
static int pred_below(const bbox* b, const void* arg) {
    const plane_arg* pa = (const plane_arg*)arg;
    float c[3];
    bbox_centroid(b, c);
    return c[pa->axis] < pa->plane;
}

typedef struct {
    const bbox* bbs;
    uint32_t leaf_size, root_leaf_size, max_depth;
} build_cfg;

static bnode* bnode_build(const build_cfg* cfg, bbox bb, uint32_t* begin, uint32_t* end, uint32_t depth);

static void bnode_make_leaf(bnode* n, uint32_t* begin, uint32_t* end) {
    n->n_items = (uint32_t)(end - begin);
    n->items = (uint32_t*)malloc(sizeof(uint32_t) * (n->n_items ? n->n_items : 1));
    memcpy(n->items, begin, sizeof(uint32_t) * n->n_items);
}
static void bnode_construct(const build_cfg* cfg, bnode* n, uint32_t* begin, uint32_t* end, uint32_t depth) {
    const ptrdiff_t count = end - begin;
    if (depth > cfg->max_depth || count <= (ptrdiff_t)cfg->leaf_size ||
        (depth == 0 && count <= (ptrdiff_t)cfg->root_leaf_size)) {
        bnode_make_leaf(n, begin, end);
        return;
    }
    const float node_size[3] = {n->bb.mx[0] - n->bb.mn[0], n->bb.mx[1] - n->bb.mn[1], n->bb.mx[2] - n->bb.mn[2]};
    uint32_t* size_split = partition_items(begin, end, cfg->bbs, pred_smaller, node_size);
    const ptrdiff_t to_split_count = size_split - begin;
    const ptrdiff_t too_large_count = end - size_split;
    if (to_split_count != 0 && too_large_count != 0) {
        n->first = bnode_build(cfg, n->bb, begin, size_split, depth + 1);
        n->second = bnode_build(cfg, n->bb, size_split, end, depth + 1);
        n->ptype = 3; /* Size */
        return;
    } else if (to_split_count == 0) {
        bnode_make_leaf(n, size_split, end);
        return;
    }

    float split_point[3] = {0.0f, 0.0f, 0.0f};
    for (ptrdiff_t i = 0; i < to_split_count; ++i) {
        float c[3];
        bbox_centroid(&cfg->bbs[begin[i]], c);
        for (int a = 0; a < 3; ++a) split_point[a] += (c[a] - split_point[a]) / (float)(i + 1);
    }
    float variance_sum[3] = {0.0f, 0.0f, 0.0f};
    uint32_t split_count[3] = {0, 0, 0};
    for (ptrdiff_t i = 0; i < to_split_count; ++i) {
        float c[3];
        bbox_centroid(&cfg->bbs[begin[i]], c);
        for (int a = 0; a < 3; ++a) {
            const float diff = c[a] - split_point[a];
            variance_sum[a] += diff * diff;
            split_count[a] += (uint32_t)(c[a] < split_point[a]);
        }
    }
    if (split_count[0] == 0 && split_count[1] == 0 && split_count[2] == 0) {
        bnode_make_leaf(n, begin, size_split);
        return;
    }
    const float score[3] = {variance_sum[0] / (float)to_split_count, variance_sum[1] / (float)to_split_count,
                            variance_sum[2] / (float)to_split_count};
    int axis;
    if (score[0] >= score[1] && score[0] >= score[2] && split_count[0]) axis = 0;
    else if (score[1] >= score[0] && score[1] >= score[2] && split_count[1]) axis = 1;
    else axis = 2;

    plane_arg pa = {axis, split_point[axis]};
    uint32_t* split_plane = partition_items(begin, size_split, cfg->bbs, pred_below, &pa);
    float mx[3] = {n->bb.mx[0], n->bb.mx[1], n->bb.mx[2]};
    float mn[3] = {n->bb.mn[0], n->bb.mn[1], n->bb.mn[2]};
    mx[axis] = mn[axis] = split_point[axis];
    n->first = bnode_build(cfg, bbox_from2(n->bb.mn, mx), begin, split_plane, depth + 1);
    n->second = bnode_build(cfg, bbox_from2(mn, n->bb.mx), split_plane, size_split, depth + 1);
    n->ptype = axis == 0 ? 2u : axis == 1 ? 1u : 0u; /* X=2, Y=1, Z=0 */
}
static void bnode_fit(const build_cfg* cfg, bnode* n) { /* fitBoundingBox */
    memset(&n->bb, 0, sizeof n->bb);
    if (!n->first) {
        if (n->n_items) {
            n->bb = cfg->bbs[n->items[0]];
            for (uint32_t i = 1; i < n->n_items; ++i) bbox_extend(&n->bb, &cfg->bbs[n->items[i]]);
        }
    } else {
        n->bb = n->first->bb;
        bbox_extend(&n->bb, &n->second->bb);
    }
}
static bnode* bnode_build(const build_cfg* cfg, bbox bb, uint32_t* begin, uint32_t* end, uint32_t depth) {
    bnode* n = (bnode*)calloc(1, sizeof *n);
    n->bb = bb;
    bnode_construct(cfg, n, begin, end, depth);
    bnode_fit(cfg, n);
    return n;
}
static void bnode_free(bnode* n) {
    if (!n) return;
    bnode_free(n->first);
    bnode_free(n->second);
    free(n->items);
    free(n);
}

/* Flattened layout (hiprz.h): root in slot 0; an inner node reserves two adjacent slots for
 * its children when it is emitted, then the first subtree is emitted completely before the
 * second; leaf primitives are appended in that same depth-first order. */
typedef struct {
    hiprz_node* nodes;
    uint32_t max_nodes, n_nodes;
    uint32_t* order;
    uint32_t n_order;
    int overflow;
} flat_t;
static void flatten(flat_t* f, const bnode* n, uint32_t slot) {
    hiprz_node* o = &f->nodes[slot];
    memcpy(o->bb_min, n->bb.mn, 12);
    memcpy(o->bb_max, n->bb.mx, 12);
    if (!n->first) {
        o->begin = f->n_order;
        o->meta = n->n_items | HIPRZ_NODE_LEAF;
        for (uint32_t i = 0; i < n->n_items; ++i) f->order[f->n_order++] = n->items[i];
    } else {
        if (f->n_nodes + 2 > f->max_nodes) {
            f->overflow = 1;
            o->begin = 0;
            o->meta = HIPRZ_NODE_LEAF;
            return;
        }
        const uint32_t c = f->n_nodes;
        f->n_nodes += 2;
        o->begin = c;
        o->meta = n->ptype << HIPRZ_NODE_PTYPE_SHIFT;
        flatten(f, n->first, c);
        flatten(f, n->second, c + 1);
    }
}

int rzo_build_mesh_tree(const hiprz_mesh_desc* mesh, hiprz_node* nodes_out, uint32_t max_nodes,
                        uint32_t* n_nodes_out, hiprz_tri* tris_out, hiprz_tri_attr* attrs_out) {
    const uint32_t T = mesh->n_triangles;
    if (max_nodes < 1) return HIPRZ_ERR_INVALID;
    bbox* bbs = (bbox*)malloc(sizeof(bbox) * (T ? T : 1));
    uint32_t* items = (uint32_t*)malloc(sizeof(uint32_t) * (T ? T : 1));
    for (uint32_t t = 0; t < T; ++t) { /* Triangle::boundingBox, mesh_component.cpp:27-33 */
        const float* p1 = &mesh->vertices[3 * mesh->tri_vertices[3 * t + 0]];
        const float* p2 = &mesh->vertices[3 * mesh->tri_vertices[3 * t + 1]];
        const float* p3 = &mesh->vertices[3 * mesh->tri_vertices[3 * t + 2]];
        bbs[t] = bbox_from2(p1, p2);
        bbox_extend_point(&bbs[t], p3);
        items[t] = t;
    }
    /* ComponentTreeNode(mesh, components): bb of components[0] extended by all */
    bbox root_bb;
    memset(&root_bb, 0, sizeof root_bb);
    if (T) root_bb = bbs[0];
    for (uint32_t t = 0; t < T; ++t) bbox_extend(&root_bb, &bbs[t]);

    build_cfg cfg = {bbs, 8u, 32u, 31u};
    bnode* root = bnode_build(&cfg, root_bb, items, items + T, 0);

    uint32_t* order = (uint32_t*)malloc(sizeof(uint32_t) * (T ? T : 1));
    flat_t f = {nodes_out, max_nodes, 1, order, 0, 0};
    flatten(&f, root, 0);
    bnode_free(root);
    int rc = f.overflow ? HIPRZ_ERR_INVALID : HIPRZ_OK;
    if (rc == HIPRZ_OK) {
        *n_nodes_out = f.n_nodes;
        for (uint32_t i = 0; i < f.n_order; ++i) {
            const uint32_t t = order[i];
            hiprz_tri* o = &tris_out[i];
            hiprz_tri_attr* a = &attrs_out[i];
            memset(o, 0, sizeof *o);
            memset(a, 0, sizeof *a);
            const float* p1 = &mesh->vertices[3 * mesh->tri_vertices[3 * t + 0]];
            const float* p2 = &mesh->vertices[3 * mesh->tri_vertices[3 * t + 1]];
            const float* p3 = &mesh->vertices[3 * mesh->tri_vertices[3 * t + 2]];
            memcpy(o->v1, p1, 12);
            memcpy(o->v2, p2, 12);
            memcpy(o->v3, p3, 12);
            o->source_index = t;
            uint32_t flags = mesh->tri_materials ? (mesh->tri_materials[t] & HIPRZ_TRI_MATERIAL_MASK) : 0u;
            const uint32_t* tt = mesh->tri_texcrds ? &mesh->tri_texcrds[3 * t] : NULL;
            const uint32_t* tn = mesh->tri_normals ? &mesh->tri_normals[3 * t] : NULL;
            /* `texcrds != Mesh::ids_unused` (cpu_engine_kernel.cpp:367,373): any index set */
            if (tt && !(tt[0] == 0xFFFFFFFFu && tt[1] == 0xFFFFFFFFu && tt[2] == 0xFFFFFFFFu)) {
                flags |= HIPRZ_TRI_HAS_TEXCRDS;
                memcpy(a->t1, &mesh->texcrds[2 * tt[0]], 8);
                memcpy(a->t2, &mesh->texcrds[2 * tt[1]], 8);
                memcpy(a->t3, &mesh->texcrds[2 * tt[2]], 8);
            }
            if (tn && !(tn[0] == 0xFFFFFFFFu && tn[1] == 0xFFFFFFFFu && tn[2] == 0xFFFFFFFFu)) {
                flags |= HIPRZ_TRI_HAS_NORMALS;
                memcpy(a->n1, &mesh->normals[3 * tn[0]], 12);
                memcpy(a->n2, &mesh->normals[3 * tn[1]], 12);
                memcpy(a->n3, &mesh->normals[3 * tn[2]], 12);
            }
            o->material_flags = flags;
            /* Triangle::calculateNormal, mesh_component.cpp:19-26 */
            const v3 v1 = v3_from(p1), v2 = v3_from(p2), vv3 = v3_from(p3);
            const v3 nrm = v3_normalized(v3_cross(v3_sub(v2, vv3), v3_sub(v2, v1)));
            a->face_normal[0] = nrm.x;
            a->face_normal[1] = nrm.y;
            a->face_normal[2] = nrm.z;
        }
    }
    free(order);
    free(items);
    free(bbs);
    return rc;
}

int rzo_build_world_tree(const hiprz_instance* instances, const uint8_t* has_mesh, uint32_t n_instances,
                         hiprz_node* nodes_out, uint32_t max_nodes, uint32_t* n_nodes_out,
                         uint32_t* order_out, uint32_t* n_order_out) {
    if (max_nodes < 1) return HIPRZ_ERR_INVALID;
    /* ObjectContainerWithBVH::update, bvh.hpp:29-53 */
    bbox* bbs = (bbox*)malloc(sizeof(bbox) * (n_instances ? n_instances : 1));
    uint32_t* items = (uint32_t*)malloc(sizeof(uint32_t) * (n_instances ? n_instances : 1));
    uint32_t n_items = 0;
    bbox bb;
    memset(&bb, 0, sizeof bb);
    for (uint32_t i = 0; i < n_instances; ++i) {
        memcpy(bbs[i].mn, instances[i].bb_min, 12);
        memcpy(bbs[i].mx, instances[i].bb_max, 12);
    }
    if (n_instances) bb = bbs[0];
    for (uint32_t i = 0; i < n_instances; ++i) {
        if (has_mesh[i]) {
            bbox_extend(&bb, &bbs[i]);
            items[n_items++] = i;
        }
    }
    build_cfg cfg = {bbs, 4u, 8u, 31u};
    bnode* root = bnode_build(&cfg, bb, items, items + n_items, 0);
    flat_t f = {nodes_out, max_nodes, 1, order_out, 0, 0};
    flatten(&f, root, 0);
    bnode_free(root);
    free(items);
    free(bbs);
    if (f.overflow) return HIPRZ_ERR_INVALID;
    *n_nodes_out = f.n_nodes;
    *n_order_out = f.n_order;
    return HIPRZ_OK;
}

/* Instance::calculateBoundingBox — instance.cpp:117-155 (no group) */
void rzo_instance_bounds(const float* vertices, uint32_t n_vertices, hiprz_instance* inst) {
    memset(inst->bb_min, 0, 12);
    memset(inst->bb_max, 0, 12);
    if (n_vertices == 0) return;
    bbox b;
    for (uint32_t i = 0; i < n_vertices; ++i) {
        v3 v = v3_mul(v3_from(&vertices[3 * i]), v3_from(inst->scale));
        v = transform_forward(inst->x_axis, inst->y_axis, inst->z_axis, v);
        const float p[3] = {v.x, v.y, v.z};
        if (i == 0) b = bbox_from2(p, p);
        else bbox_extend_point(&b, p);
    }
    for (int a = 0; a < 3; ++a) {
        inst->bb_min[a] = b.mn[a] + inst->position[a];
        inst->bb_max[a] = b.mx[a] + inst->position[a];
    }
}

/* Math::vec3::RotateX/Y/Z as restated in cuda_render_parts.cuh:116-139. */
static v3 rot_x(v3 v, float a) {
    const float s = sinf(a), c = cosf(a);
    return V3(v.x, v.y * c + v.z * s, v.y * -s + v.z * c);
}
static v3 rot_y(v3 v, float a) {
    const float s = sinf(a), c = cosf(a);
    return V3(v.x * c - v.z * s, v.y, v.x * s + v.z * c);
}
static v3 rot_z(v3 v, float a) {
    const float s = sinf(a), c = cosf(a);
    return V3(v.x * c + v.y * s, v.x * -s + v.y * c, v.z);
}
static void store3(float* o, v3 v) { o[0] = v.x, o[1] = v.y, o[2] = v.z; }
/* CoordSystem::applyRotation — render_parts.cpp:51-56 (RotatedXYZ = X then Y then Z) */
void rzo_axes_from_rotation(const float r[3], float xa[3], float ya[3], float za[3]) {
    store3(xa, rot_z(rot_y(rot_x(V3(1, 0, 0), r[0]), r[1]), r[2]));
    store3(ya, rot_z(rot_y(rot_x(V3(0, 1, 0), r[0]), r[1]), r[2]));
    store3(za, rot_z(rot_y(rot_x(V3(0, 0, 1), r[0]), r[1]), r[2]));
}
/* CoordSystem::lookAt — render_parts.cpp:57-62 (Z then X then Y) */
void rzo_axes_look_at(const float r[3], float xa[3], float ya[3], float za[3]) {
    store3(xa, rot_y(rot_x(rot_z(V3(1, 0, 0), r[2]), r[0]), r[1]));
    store3(ya, rot_y(rot_x(rot_z(V3(0, 1, 0), r[2]), r[0]), r[1]));
    store3(za, rot_y(rot_x(rot_z(V3(0, 0, 1), r[2]), r[0]), r[1]));
}
