/*
 * rz_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of RayZath's CPU path tracer (`/root/reference/RayZath/
 * cpu_engine_kernel.cpp` and the helpers it calls).  It is the checker the HIP backend
 * is compared against; nothing under rayzath_amd/ may include, link or call it.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * PARITY UNPINNED: the reference holds no golden vector, known-answer test or fixture
 * for this path (the four files under Tests/ cover Args, index_of, static_dictionary, text_utils only),
 * and the reference's CPU engine cannot be built in this image without writing
 * stand-ins for headers it lacks (vec2.h vec3.h angle.h constants.h color.h bitmap.h
 * point.h from the un-vendored Greketrotny/Math and Greketrotny/Graphics repos, no
 * pinned version: RayZath/RayZath.vcxproj:49,68-69; README.md:49).  This restatement is
 * therefore anchored on the reference's source text alone; where that text calls into
 * the missing headers (vector normalise / similarity / rotate, ColorF arithmetic,
 * Color->ColorF) it follows the reference's own CUDA restatement of the same types
 * (RayZath/cuda_render_parts.cuh:15-330, 520-700).  See DESIGN.md §Oracle.
 */
#ifndef RZ_ORACLE_H
#define RZ_ORACLE_H

#include "../include/hiprz.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Per-pixel persistent state = CPU::CameraContext (cpu_engine_kernel.hpp:29-51),
 * row-major W*H arrays. */
typedef struct rzo_context {
    uint32_t width, height;
    float* image;         /* RGBA32F accumulator, alpha = finished paths   (m_image)      */
    uint8_t* path_depth;  /*                                               (m_path_depth) */
    float* ray_origin;    /* xyz                                           (m_ray_origin) */
    float* ray_direction; /* xyz                                           (m_ray_direction) */
    uint32_t* ray_material; /* material index instead of const Material*   (m_ray_material) */
    float* ray_color;     /* RGBA                                          (m_ray_color)  */
    float* depth;         /* Camera::depthBuffer()                                        */
    uint8_t* rgba8;       /* Camera::imageBuffer()                                        */
    uint32_t passes;      /* passes rendered since reset                                  */
    uint64_t traced_rays; /* m_traced_rays                                                */
} rzo_context;

rzo_context* rzo_context_create(uint32_t width, uint32_t height);
void rzo_context_destroy(rzo_context* ctx);
void rzo_context_reset(rzo_context* ctx); /* CameraContext::reset, cpu_engine_renderer.cpp:32-39 */

/* One pass over every pixel, tiled 128x128 over `threads` workers pulling tiles from an
 * atomic counter (Renderer::renderCameraView, cpu_engine_renderer.cpp:186-279).  The
 * first pass after a reset is renderFirstPass, later ones renderCumulativePass; each
 * pass also tone-maps into ctx->rgba8 like the reference does inline.  counters may be
 * NULL.  threads <= 0 means all hardware threads. */
void rzo_render_pass(const hiprz_scene* scene, const hiprz_camera* camera, const hiprz_config* config,
                     rzo_context* ctx, int threads, hiprz_counters* counters);

/* Kernel::rayCast (cpu_engine_kernel.cpp:102-111, 483-501). */
void rzo_pick(const hiprz_scene* scene, const hiprz_camera* camera, const rzo_context* ctx, uint32_t x,
              uint32_t y, int32_t* instance_out, int32_t* material_out);

/* Independent restatement of the host tree builders (bvh_tree_node.hpp:117-215,
 * component_container.hpp:259-363) with the flattened output layout of hiprz.h — used to
 * check hiprz_build_mesh_tree / hiprz_build_world_tree node-for-node. */
int rzo_build_mesh_tree(const hiprz_mesh_desc* mesh, hiprz_node* nodes_out, uint32_t max_nodes,
                        uint32_t* n_nodes_out, hiprz_tri* tris_out, hiprz_tri_attr* attrs_out);
int rzo_build_world_tree(const hiprz_instance* instances, const uint8_t* has_mesh, uint32_t n_instances,
                         hiprz_node* nodes_out, uint32_t max_nodes, uint32_t* n_nodes_out,
                         uint32_t* order_out, uint32_t* n_order_out);
void rzo_instance_bounds(const float* vertices, uint32_t n_vertices, hiprz_instance* inst);
void rzo_axes_from_rotation(const float rotation[3], float x_axis[3], float y_axis[3], float z_axis[3]);
void rzo_axes_look_at(const float rotation[3], float x_axis[3], float y_axis[3], float z_axis[3]);

/* --- known-answer entry points for unit tests of single functions --- */
float rzo_seed_value(uint32_t seed, uint32_t pass, uint32_t i);
void rzo_rng_sequence(float seed_x, float seed_y, float r, uint32_t n, float* out);
int rzo_box_test(const float bb_min[3], const float bb_max[3], const float origin[3], const float direction[3],
                 float near_, float far_);
/* returns 1 on hit and writes t, b1, b2, external */
int rzo_triangle_test(const float v1[3], const float v2[3], const float v3[3], const float origin[3],
                      const float direction[3], float near_, float far_, float out4[4]);
float rzo_fresnel(const float n[3], const float i[3], float n1, float n2, float factors[2]);
void rzo_cosine_sample_hemisphere(float r1, float r2, const float n[3], float out[3]);
void rzo_sample_sphere(float r1, float r2, const float n[3], float out[3]);
void rzo_sample_disk(float r1, float r2, const float n[3], float radius, float out[3]);
void rzo_tonemap_pixel(const float rgba[4], float aperture, float exposure_time, uint8_t out[4]);

const char* rzo_math_mode(void); /* "libm" or "portable" */

#ifdef __cplusplus
}
#endif
#endif
