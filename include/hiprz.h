/*
 * hiprz.h — C-ABI of the MI355X (gfx950) path-tracing backend for RayZath.
 *
 * This is the drop-in boundary for ONE path of the reference: what
 * `RayZath::Engine::Engine::renderWorld` (RayZath/rayzath.cpp:64-94) dispatches to a
 * backend — `CPU::Engine::renderWorld` (RayZath/cpu_engine.hpp:17-22) /
 * `Cuda::Engine::renderWorld` (RayZath/cuda_engine.cuh:34-39) — i.e. the per-pixel
 * integration loop of RayZath/cpu_engine_kernel.cpp, plus the device-side scene
 * mirror the CUDA backend keeps (RayZath/cuda_world.cuh, cuda_bvh.cuh,
 * cuda_instance.cuh, cuda_material.cuh, cuda_camera.cuh).
 *
 * Everything crossing this boundary is plain-old-data: pointers, sizes, fixed-layout
 * structs.  No C++ types, no torch types, no exceptions.  Every call returns an int
 * (HIPRZ_OK == 0); the message of the last failure is available from
 * hiprz_last_error().  A RayZath-side adapter walks the host
 * `World`, fills a `hiprz_scene`, and converts non-zero returns into
 * `RayZath::Exception` (RayZath/rzexception.hpp:11-26), including the deferred-throw
 * contract of the CUDA backend (RayZath/cuda_engine_core.cu:41).
 *
 * The context is NOT re-entrant: callers serialise calls per context exactly as the
 * reference serialises `renderWorld` with a mutex (RayZath/cpu_engine_core.cpp:15,
 * RayZath/cuda_engine_core.cu:38).
 */
#ifndef HIPRZ_H
#define HIPRZ_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPRZ_OK 0
#define HIPRZ_ERR_INVALID 1   /* bad argument / inconsistent scene (checked on the host before any launch) */
#define HIPRZ_ERR_DEVICE 2    /* a hip* call failed */
#define HIPRZ_ERR_STATE 3     /* call order violated (e.g. render before scene+camera upload) */

/* ---------------------------------------------------------------------------------
 * Flattened scene snapshot (SoA of fixed-size POD records; all float = IEEE fp32).
 * Replaces the CUDA backend's AoS device mirror (RayZath/cuda_bvh_tree_node.cuh:8-54
 * 48-B nodes, cuda_render_parts.cuh:996-1003 144-B triangles, cuda_instance.cuh:167-176
 * ~640-B instances with 64 material pointers).
 * ------------------------------------------------------------------------------- */

#define HIPRZ_NODE_LEAF 0x80000000u       /* meta bit 31: node is a leaf                  */
#define HIPRZ_NODE_PTYPE_SHIFT 29         /* meta bits 29..30: partition type of an inner */
#define HIPRZ_NODE_COUNT_MASK 0x1FFFFFFFu /* node (X=2,Y=1,Z=0,Size=3; bvh_tree_node.hpp:22-28) */

/* One BVH node, 32 B.  The world-level tree over instances (RayZath/bvh.hpp:29-53) and
 * every per-mesh tree over triangles (RayZath/component_container.hpp:145-363) live in
 * ONE array.  Inner node: children are nodes[begin] (first) and nodes[begin+1] (second).
 * Leaf: primitives [begin, begin+count) — indices into tlas_order[] for the world tree,
 * into tris[] (already in leaf order, de-indexed) for a mesh tree. */
typedef struct hiprz_node {
    float bb_min[3];
    float bb_max[3];
    uint32_t begin;
    uint32_t meta; /* count | ptype << 29 | HIPRZ_NODE_LEAF */
} hiprz_node;

#define HIPRZ_TRI_HAS_TEXCRDS 0x40000000u
#define HIPRZ_TRI_HAS_NORMALS 0x80000000u
#define HIPRZ_TRI_MATERIAL_MASK 0x00FFFFFFu

/* Intersection record of one triangle, 48 B (three 16-B loads).  Vertices are stored
 * de-indexed in the order the reference reads them, v1 v2 v3
 * (RayZath/mesh_component.cpp:52-57). */
typedef struct hiprz_tri {
    float v1[3];
    uint32_t material_flags; /* material_id | HIPRZ_TRI_HAS_* (mesh_component.hpp:27-33) */
    float v2[3];
    uint32_t source_index; /* index of the triangle in its mesh before leaf reordering */
    float v3[3];
    uint32_t pad0; /* ignored on upload (the device copy keeps the triangle's position in the reference's leaf order here) */
} hiprz_tri;

/* Shading record of one triangle, 96 B, read once per hit
 * (RayZath/cpu_engine_kernel.cpp:354-395). */
typedef struct hiprz_tri_attr {
    float n1[3];
    float pad0;
    float n2[3];
    float pad1;
    float n3[3];
    float pad2;
    float face_normal[3]; /* normalize(cross(v2-v3, v2-v1)), mesh_component.cpp:19-26 */
    float pad3;
    float t1[2];
    float t2[2];
    float t3[2];
    float pad4[2];
} hiprz_tri_attr;

/* One instance, 112 B (RayZath/instance.hpp:17-24, render_parts.hpp:39-69). */
typedef struct hiprz_instance {
    float position[3];
    uint32_t blas_root; /* node index of the mesh tree's root */
    float scale[3];
    uint32_t material_base; /* first entry in inst_materials[] */
    float x_axis[3];
    uint32_t material_count; /* entries in inst_materials[] (<= 64, instance.hpp:17) */
    float y_axis[3];
    uint32_t pad0;
    float z_axis[3];
    uint32_t pad1;
    float bb_min[3]; /* world-space box, instance.cpp:117-155 */
    uint32_t pad2;
    float bb_max[3];
    uint32_t pad3;
} hiprz_instance;

#define HIPRZ_MATERIAL_WORLD 0u   /* materials[0] = World::material()        (world.cpp:33-38) */
#define HIPRZ_MATERIAL_DEFAULT 1u /* materials[1] = World::defaultMaterial() (world.cpp:39-43) */

/* One material, 48 B (RayZath/material.hpp:119-160).  Map fields index textures[], -1 = none. */
typedef struct hiprz_material {
    uint8_t color[4]; /* Graphics::Color r,g,b,a (a = opacity, 255 opaque) */
    float metalness;
    float roughness;
    float emission;
    float ior;
    float scattering;
    int32_t texture;
    int32_t normal_map;
    int32_t metalness_map;
    int32_t roughness_map;
    int32_t emission_map;
    uint32_t pad0;
} hiprz_material;

#define HIPRZ_TEX_RGBA8 0u /* Texture / NormalMap (Graphics::Color)   */
#define HIPRZ_TEX_R8 1u    /* MetalnessMap / RoughnessMap (uint8_t)   */
#define HIPRZ_TEX_R32F 2u  /* EmissionMap (float)                     */
/* hiprz_texture.sampling (RayZath/render_parts.hpp:116-130 FilterMode / AddressMode; cuda_buffer.cuh:364-401) */
#define HIPRZ_TEX_FILTER_POINT 0u
#define HIPRZ_TEX_FILTER_LINEAR 1u
#define HIPRZ_TEX_ADDRESS_WRAP (0u << 8)
#define HIPRZ_TEX_ADDRESS_CLAMP (1u << 8)
#define HIPRZ_TEX_ADDRESS_MIRROR (2u << 8)
#define HIPRZ_TEX_ADDRESS_BORDER (3u << 8)

/* One texture descriptor, 48 B (RayZath/render_parts.hpp:113-222).  Texels are
 * row-major, top row first, at texels + offset. cos/sin of the rotation are hoisted to
 * the host (the reference evaluates them inside vec2::Rotate on every fetch). */
typedef struct hiprz_texture {
    uint32_t kind;
    uint32_t width;
    uint32_t height;
    uint32_t offset; /* byte offset into the texel pool, 4-B aligned */
    float scale[2];
    float translation[2];
    float rotation;
    float cos_rotation;
    float sin_rotation;
    uint32_t sampling; /* HIPRZ_TEX_FILTER_* | HIPRZ_TEX_ADDRESS_*: read only in CUDA-compat mode (HIPRZ_COMPAT_FILTERING); the CPU
                          kernel point-samples with wrap-around whatever the scene file says (render_parts.hpp:209-221) */
} hiprz_texture;

/* Spot light, 48 B (RayZath/spot_light.hpp). cos_angle = cosf(angle) hoisted to the host. */
typedef struct hiprz_spot_light {
    float position[3];
    float size;
    float direction[3]; /* normalised, spot_light.cpp:27-32 */
    float emission;
    uint8_t color[4];
    float angle;
    float cos_angle;
    uint32_t pad0;
} hiprz_spot_light;

/* Direct light, 32 B (RayZath/direct_light.hpp). */
typedef struct hiprz_direct_light {
    float direction[3]; /* normalised, direct_light.cpp:22-27 */
    float emission;
    uint8_t color[4];
    float angular_size;
    float cos_angular_size; /* cosf(angular_size), hoisted to the host */
    uint32_t pad0;
} hiprz_direct_light;

typedef struct hiprz_scene {
    uint32_t n_nodes;
    const hiprz_node* nodes;
    uint32_t tlas_root; /* node index of the world tree's root; ignored if n_tlas_order == 0 */
    uint32_t n_tlas_order;
    const uint32_t* tlas_order; /* instance ids in leaf order */
    uint32_t n_tris;
    const hiprz_tri* tris;
    const hiprz_tri_attr* tri_attrs;
    uint32_t n_instances;
    const hiprz_instance* instances;
    uint32_t n_inst_materials;
    const int32_t* inst_materials; /* material index, -1 = unset (=> default material) */
    uint32_t n_materials;          /* >= 2: [0] world, [1] default */
    const hiprz_material* materials;
    uint32_t n_textures;
    const hiprz_texture* textures;
    size_t texel_bytes;
    const uint8_t* texels;
    uint32_t n_spot_lights;
    const hiprz_spot_light* spot_lights;
    uint32_t n_direct_lights;
    const hiprz_direct_light* direct_lights;
} hiprz_scene;

/* Camera (RayZath/camera.hpp:127-161, camera.cpp). tan_half_fov = tanf(fov*0.5f) is
 * hoisted to the host (cpu_engine_kernel.cpp:186,214 evaluates it per pixel). */
typedef struct hiprz_camera {
    float position[3];
    float x_axis[3];
    float y_axis[3];
    float z_axis[3];
    uint32_t width;
    uint32_t height;
    float fov;
    float tan_half_fov;
    float aspect_ratio; /* float(width) / float(height), camera.cpp:53 */
    float near_far[2];
    float focal_distance;
    float aperture;
    float exposure_time;
} hiprz_camera;

/* RenderConfig (RayZath/engine_parts.hpp:76-128) + the deterministic seeding the
 * harness adds (the CPU renderer seeds from std::random_device,
 * cpu_engine_renderer.cpp:143-145, so a fixed convention is ours: SURVEY.md §8 a1). */
typedef struct hiprz_config {
    uint32_t max_depth;      /* Tracing::maxDepth, u8 in the reference (1..254) */
    uint32_t rpp;            /* Tracing::rpp: cumulative passes per render call */
    uint32_t spot_samples;   /* LightSampling::spotLight   (>= 1) */
    uint32_t direct_samples; /* LightSampling::directLight (>= 1) */
    uint32_t seed;           /* base of the per-pass 256-entry seed table */
} hiprz_config;

/* Work counters of the traversal (the figures the algorithmic byte
 * count is built from).  Filled only by hiprz_render_counted(). */
typedef struct hiprz_counters {
    uint64_t segments;      /* path segments traced (= rays, cpu_engine_renderer.cpp:173) */
    uint64_t box_tests;     /* BoundingBox::rayIntersection calls, closest + shadow */
    uint64_t tri_tests;     /* Triangle::closest/anyIntersection calls */
    uint64_t hits;          /* segments whose closest-hit search found a surface */
    uint64_t shadow_rays;   /* anyIntersection(ray) calls */
    uint64_t light_samples; /* light samples evaluated (NEE loop iterations) */
    uint64_t texel_fetches; /* TextureBuffer::fetch calls */
    uint64_t finished;      /* paths finished (alpha increments, cpu_engine_kernel.cpp:82) */
    uint64_t shadow_box_tests; /* the part of box_tests done for shadow rays (anyIntersection) */
    uint64_t shadow_tri_tests; /* the part of tri_tests done for shadow rays */
} hiprz_counters;

typedef struct hiprz_ctx hiprz_ctx;

/* --- lifecycle (replaces Cuda::Engine ctor/dtor, RayZath/cuda_engine.cu:8-21) --- */
int hiprz_create(hiprz_ctx** out, int device_id);
/* One context over several GPUs of a node (replaces the device selection of Cuda::EngineCore, cuda_engine_core.cu:17, which is pinned
 * to device 0): device r of n renders shard r of n of the interleaved tiles (hiprz_set_shard, which then splits the context's share once
 * more), the scene is mirrored to every device, and hiprz_read_* / hiprz_pick / hiprz_ray_count return the whole frame — the peers'
 * tiles cross xGMI in peer-to-peer copies on the head's stream.  Global pixel ids and seeds are unchanged: the frame equals the
 * single-device frame bit for bit.  The same id may be listed more than once: several contexts-with-a-stream on one GPU, each rendering
 * its interleaved share — one share's sorts, pass bookkeeping and kernel tails then run beside another share's walks (measured on
 * MI355X, two streams: +6 % on a Cornell box, +12 % / +5 % with a 6 k / 301 k triangle mesh in it, -10 % on a scene with lights, whose
 * shadow kernel and sorts already overlap on streams of their own; the Engine hosts choose by that: Hip::Engine::defaultStreams). */
int hiprz_create_multi(hiprz_ctx** out, const int* device_ids, int n_devices);
int hiprz_device_count(hiprz_ctx* ctx, uint32_t* out);
int hiprz_destroy(hiprz_ctx* ctx);
/* message of the last non-OK return on this context (ctx may be NULL for create failures) */
const char* hiprz_last_error(const hiprz_ctx* ctx);

/* Pure host check of a scene snapshot (the same one hiprz_upload_scene runs before it touches the
 * device): every index in range, trees disjoint and acyclic, map kinds consistent.  Writes the
 * reason into message on failure. */
int hiprz_validate_scene(const hiprz_scene* scene, char* message, size_t len);
/* sizeof() of the POD records, in the order node, tri, tri_attr, instance, material, texture,
 * spot_light, direct_light, scene, camera, config, counters, mesh_desc (for binding generators). */
void hiprz_abi_sizes(uint32_t out[13]);

/* --- host→device mirroring (replaces Cuda::World::reconstruct*, cuda_world.cu:28-57) --- */
int hiprz_upload_scene(hiprz_ctx* ctx, const hiprz_scene* scene);   /* validates, copies; caller keeps ownership */
int hiprz_upload_camera(hiprz_ctx* ctx, const hiprz_camera* camera); /* (re)allocates per-pixel state on resize */
/* Materials and lights changed, geometry did not (the reference tracks modifications per container, updatable.cpp:23-51, and the
 * CUDA backend re-mirrors only what changed, cuda_world.cu:28-57): replaces those records of the uploaded scene in place — no tree is
 * rebuilt or re-derived.  n_materials must equal the uploaded scene's; light counts may differ.  Restarts accumulation. */
int hiprz_update_shading(hiprz_ctx* ctx, const hiprz_material* materials, uint32_t n_materials, const hiprz_spot_light* spot_lights,
                         uint32_t n_spot_lights, const hiprz_direct_light* direct_lights, uint32_t n_direct_lights);
int hiprz_set_config(hiprz_ctx* ctx, const hiprz_config* config);
/* Cameras.  The reference renders every enabled camera of the world per call, each with its own accumulation state
 * (cpu_engine_renderer.cpp:97-117, CameraContext).  A context keeps one frame state per camera: hiprz_set_camera_count(n), then
 * hiprz_select_camera(k) decides which camera hiprz_upload_camera / hiprz_reset / hiprz_render* / hiprz_tonemap / hiprz_read_* /
 * hiprz_pick / hiprz_ray_count / hiprz_pass_count address.  Scene, config and every setting are shared.  Default: one camera. */
int hiprz_set_camera_count(hiprz_ctx* ctx, uint32_t n);
int hiprz_camera_count(hiprz_ctx* ctx, uint32_t* out);
int hiprz_select_camera(hiprz_ctx* ctx, uint32_t index);
/* Own only shard `rank` of `world` of the frame's 32x8-pixel tiles: the tiles are numbered row by row, shard `rank` owns the numbers t with
 * t % world == rank and keeps them in order (local tile lt = t / world; the shards' counts differ by at most one).  Within tile row r the
 * numbers are rotated by a few columns so that column c goes to shard (c + offset(r)) % world, offset() running through a permutation of
 * 0 .. world - 1 every `world` rows: a column of tiles, a row or a slanted line of the image is dealt to all shards in turn — a thin
 * expensive feature (a lamp post, the edge of a wall) does not land on one or two of them, as it does with unrotated numbers whenever
 * `world` divides the tiles per row (rayzath_amd/csrc/hiprz_shard.hpp; rayzath_amd/distributed.py mirrors it).  Global pixel ids and
 * seeds are unchanged, so results are identical for any world size.  Default rank 0 of 1.  A context over n devices / streams
 * (hiprz_create_multi) renders the sub-shards rank * n + k of world * n, k = 0 .. n - 1 — which tile for tile is another set than shard
 * `rank` of `world` of a one-part context: all contexts of a job must have the same number of parts. */
int hiprz_set_shard(hiprz_ctx* ctx, uint32_t rank, uint32_t world);
/* How the n parts (devices / streams) of a hiprz_create_multi context divide the context's share of the frame (tiles with a
 * gather, "or per-sample reduce").  The reference has nothing to mirror (one device: cuda_engine_core.cu:17).
 * HIPRZ_SHARD_TILES (default): part k renders sub-shard rank * n + k of world * n of the interleaved tiles; frames are the one-device frame
 * bit for bit whatever n is, a frame of `rpp` passes arrives sooner — but a part's step lasts as long as the longest per-tile chain of
 * passes, which caps the kernel-side speed-up of 8 parts at 5.1 - 5.6 (2.7 with one hot image region).
 * HIPRZ_SHARD_SAMPLES: EVERY part renders the context's whole share, part k on the seed stream `config.seed + k` (hiprz_seed_value), and
 * the parts' accumulators are SUMMED (colour sums and finished-path counts alike, in part order: deterministic) wherever the frame leaves
 * the context: hiprz_tonemap tone-maps the sum, hiprz_read_accum / hiprz_export_accum_tiles return it (ONE tile-major slice of the share),
 * hiprz_read_rgba8 / hiprz_export_rgba8_tiles the tone-mapped sum, hiprz_ray_count the rays of all parts (n * passes * pixels);
 * hiprz_read_depth, hiprz_read_state, hiprz_ray_cast and hiprz_pass_count answer for part 0 (the first-pass depth is every part's).
 * A call of hiprz_render(p) therefore adds n * p samples per pixel.  The frame equals the sum of n one-part frames rendered with the
 * seeds seed .. seed + n - 1 — the same for a given n, another one for another n.  Changing the mode restarts accumulation.
 * HIPRZ_COMPAT_REPROJECTION under this mode: every part keeps the history of its own frame (each holds the whole share), so a restarted
 * frame starts from the sum of the parts' blended histories — what one part's history is to one part's frame.
 * Processes that sample-shard a frame between them give each context its own `config.seed` (rank * n apart) and reduce the exported
 * accumulators (rayzath_amd/distributed.py: ShardedFrame.reduce, one ncclReduce(sum) per readback). */
#define HIPRZ_SHARD_TILES 0u
#define HIPRZ_SHARD_SAMPLES 1u
int hiprz_set_shard_mode(hiprz_ctx* ctx, uint32_t mode);
int hiprz_shard_mode(hiprz_ctx* ctx, uint32_t* mode_out);

/* Tree-walk variant of the pass kernels.  -1 (default) = chosen per scene; 1 = nested loops with a per-lane stack in LDS;
 * 2 = workgroup-binned (rays advance in rounds, the (ray, instance) visits of a round are compacted and sorted by instance
 * in LDS and processed by dense waves; scenes staged in LDS); 3 = skip links, one wave per workgroup, for scenes that are not
 * staged in LDS (see hiprz_set_walk_order: front to back with a cooperative triangle phase by default; in the reference's
 * child order with the tree tops cached in LDS).  All reach the same hits and give identical results; 1, 2 and 3 in the
 * reference's order also execute the same box / triangle tests in the same per-ray order. */
int hiprz_set_traversal_mode(hiprz_ctx* ctx, int mode);
/* Order in which the single-wave skip-link walk (mode 3) enters the two children of a MESH-tree node.  0 = the reference's
 * fixed order, first child then second (cpu_engine_kernel.cpp:331-352): box / triangle test counters equal the CPU kernel's.
 * 1 (default) = front to back: the child on the side the ray comes from first (decided by the sign of the ray direction along
 * the node's split axis; per ray octant the order is fixed, so the walk stays stack-free on per-octant skip links).  The closest
 * hit is the same triangle — among equal distances the one the reference meets first wins — while fewer boxes and triangles are
 * tested (config D: -22 % / -30 %).  World trees and instance lists are always walked in the reference's order.  In this
 * order the leaf triangles of all lanes of a wave are tested cooperatively (rz_trace_coop_kernel) and shadow rays likewise.
 * hiprz_render_counted() walks in the reference's order under 0 and 1, so its counters are the work of the reference's
 * algorithm (what the roofline's algorithmic bytes are made of); 2 = front to back there too: counters = tests executed. */
int hiprz_set_walk_order(hiprz_ctx* ctx, int order);
int hiprz_traversal_mode(hiprz_ctx* ctx, int* effective_mode_out); /* valid after hiprz_upload_scene */

/* Behaviours of the reference's CUDA engine that its CPU engine — the parity oracle — does not have.  mode = 0
 * (default): the CPU kernel, bit-comparable with the oracle.  Any HIPRZ_COMPAT_* flag routes the passes through one fused kernel
 * (no LDS staging, no ray reordering) that adds the selected behaviours; they are validated against their analytic expectations,
 * not against the oracle.  Changing the mode restarts accumulation. */
#define HIPRZ_COMPAT_BEER_LAMBERT 1u  /* ray.color *= opacityColor(medium) * pow(its alpha, distance): cuda_render_kernel.cu:174-176   */
#define HIPRZ_COMPAT_SCATTERING 2u    /* medium scattering, Material::applyScattering: cuda_material.cuh:141-159, cuda_world.cuh:91-100 */
#define HIPRZ_COMPAT_SHADOW_COLOR 4u  /* shadow rays pass through triangles, mask *= opacityColor(uv): cuda_instance.cuh:92-164         */
#define HIPRZ_COMPAT_TEXTURE_MULT 8u  /* texture x colour, emission map x emission: cuda_material.cuh:75-123                           */
#define HIPRZ_COMPAT_FILTERING 16u    /* hiprz_texture.sampling honoured (linear filter, clamp / mirror / border): cuda_buffer.cuh:364-438 */
#define HIPRZ_COMPAT_REPROJECTION 32u  /* spatio-temporal reprojection at a restart: cuda_camera.cuh:382-426, cuda_engine_renderer.cu:139-150 */
#define HIPRZ_MODE_CPU 0u
#define HIPRZ_MODE_CUDA_COMPAT 63u
int hiprz_set_mode(hiprz_ctx* ctx, uint32_t compat_flags);
/* HIPRZ_COMPAT_REPROJECTION: when accumulation restarts (camera, scene or config changed, hiprz_reset) the frame rendered so far is
 * kept as history; after the new first pass every pixel's first hit point is projected into the previous camera and, where the
 * previous depth buffer agrees within 1 %, the previous accumulator * temporal blend is appended — colour sum and sample count alike,
 * so the tone map sees a frame that starts with blend * (old sample count) samples.  It changes what a restart starts from, not the
 * integrator: it combines with any other flag, also with none.  History is per camera and is dropped when the resolution or the
 * shard changes.  A context over several devices (hiprz_create_multi) assembles the whole previous frame from all of them at the restart,
 * so the source pixel of a moved camera may come from any device; a frame the caller sharded over several CONTEXTS (hiprz_set_shard)
 * carries over history from the tiles each context owns only.
 * hiprz_set_temporal_blend: Camera::temporalBlend of the selected camera (camera.cpp:154-156: clamped to [0, 1]; default 0.75). */
int hiprz_set_temporal_blend(hiprz_ctx* ctx, float blend);

/* Which mesh trees the walks use.  HIPRZ_TREE_REFERENCE (default): the trees of the uploaded snapshot — the reference's builder
 * (bvh_tree_node.hpp:117-215), the anchor of the work counters.  HIPRZ_TREE_SAH: at upload every mesh tree is rebuilt over the same
 * triangles with a binned surface-area heuristic (leaves of at most 8 triangles).  A tree decides which boxes and triangles a ray
 * meets, not what it hits: frames are identical (equally distant triangles are ranked by their position in the reference's visiting
 * order), while the tests per segment drop.  One caveat, and it is the reference's arithmetic: Moeller-Trumbore bumps a determinant
 * below 1e-7 to 1e-7 (mesh_component.cpp:52-83), so a triangle with an area around 1e-7 and a ray nearly in its plane can report a
 * hit at a distance that has nothing to do with it — a hit that exists only if that triangle is TESTED, i.e. if the ray meets the
 * box of the leaf it sits in, and every tree groups leaves differently.  BASELINE's configs A - E have no such triangles (frames equal
 * under all trees at full size); a 3 M-triangle version of D differs in 39 of 2 073 600 first hits (profiles/r04/tree_dependence_F.txt).
 * A host that needs the reference's frames on such meshes keeps HIPRZ_TREE_REFERENCE.  With a rebuilt tree every walk — counted ones too — runs front to back on skip links
 * (the scene is not staged in LDS), and the work counters are those of the rebuilt tree.  Takes effect at the next hiprz_upload_scene. */
#define HIPRZ_TREE_REFERENCE 0u
#define HIPRZ_TREE_SAH 1u
/* HIPRZ_TREE_DEVICE: the trees are built ON THE DEVICE at upload (the reference rebuilds them on the host at every change,
 * bvh_tree_node.hpp:117-215, component_container.hpp:259-363, cuda_engine_core.cu:58-60): every mesh tree by kernels — Morton order of
 * the triangle centroids, a binary radix tree, boxes bottom-up, leaves of at most 4 triangles, the skip links of all 8 ray octants —
 * and the world tree over the instances by the reference's own top-down builder run on the device (the order in which a ray meets the
 * instances is part of its arithmetic, so that tree must be the reference's: it is, node for node).  Frames are those of the reference
 * trees bit for bit.  Afterwards hiprz_update_triangles and hiprz_update_instances change geometry without a host-side tree build. */
#define HIPRZ_TREE_DEVICE 2u
/* HIPRZ_TREE_DEVICE_SAH: as HIPRZ_TREE_DEVICE, the mesh trees built by the device with HIPRZ_TREE_SAH's binned surface-area heuristic
 * (16 bins per axis, leaves of at most 8 triangles; level by level for large nodes, a thread per subtree below 32 triangles) instead
 * of Morton order: a few more milliseconds of build for walks as short as the host-built surface-area trees'.  Same frames, same
 * hiprz_update_triangles / hiprz_update_instances afterwards. */
#define HIPRZ_TREE_DEVICE_SAH 3u
/* HIPRZ_TREE_AUTO: per uploaded scene — HIPRZ_TREE_REFERENCE when the scene's records are small enough to be staged in LDS (the resident
 * kernels walk the snapshot's trees there: BASELINE config B), HIPRZ_TREE_DEVICE_SAH otherwise (configs C, D, E: 3 - 9 % more rays per
 * second, a device build of 1 - 4 ms).  What the Engine hosts and bench.py set; a bare context keeps HIPRZ_TREE_REFERENCE, the anchor of
 * the work counters. */
#define HIPRZ_TREE_AUTO 4u
int hiprz_set_tree(hiprz_ctx* ctx, uint32_t tree);
int hiprz_tree(hiprz_ctx* ctx, uint32_t* tree_out); /* the trees of the uploaded scene: HIPRZ_TREE_REFERENCE .. HIPRZ_TREE_DEVICE_SAH */
/* Scenes uploaded under HIPRZ_TREE_DEVICE only.  New records for the triangles [first, first + n) of the uploaded snapshot's order
 * (vertices moved, normals / texture coordinates / face normals as the caller computed them; materials and source indices are taken
 * from the new records): the device copies are rewritten and the boxes of the mesh trees that hold them are fitted again, bottom-up,
 * on the device — the topology of the trees stays.  Restarts accumulation.  Follow with hiprz_update_instances when a mesh's extent
 * changed (the instances' world-space boxes are the caller's, Instance::calculateBoundingBox, instance.cpp:117-155). */
int hiprz_update_triangles(hiprz_ctx* ctx, uint32_t first, uint32_t n, const hiprz_tri* tris, const hiprz_tri_attr* attrs);
/* Scenes uploaded under HIPRZ_TREE_DEVICE only.  New transformations and world-space boxes for ALL instances (n = the uploaded count;
 * position, scale, axes and box are taken from the records, mesh and material tables stay as uploaded): the world tree is rebuilt on
 * the device.  Restarts accumulation. */
int hiprz_update_instances(hiprz_ctx* ctx, const hiprz_instance* instances, uint32_t n);
/* Scenes with device-built trees only.  Every mesh tree built again by the device over the vertices it holds NOW (after
 * hiprz_update_triangles calls deformed a mesh far from the shape its tree was built for, a refitted tree keeps its topology and its
 * boxes grow into each other): HIPRZ_TREE_DEVICE (Morton order, the fastest build) or HIPRZ_TREE_DEVICE_SAH.  Nothing crosses the bus;
 * the uploaded order of the triangles — what hiprz_update_triangles addresses — stays.  Restarts accumulation. */
int hiprz_rebuild_trees(hiprz_ctx* ctx, uint32_t tree);
/* The trees the context walks now, as a snapshot would hold them (for validation and tests): node records with plain boxes, the world
 * tree's root and leaf order, every instance's mesh root, and for every triangle of the device order its position in the uploaded
 * order.  Any output may be NULL. */
int hiprz_download_trees(hiprz_ctx* ctx, hiprz_node* nodes_out, uint32_t max_nodes, uint32_t* n_nodes_out, uint32_t* tlas_root_out,
                         uint32_t* tlas_order_out, uint32_t* blas_roots_out, uint32_t* tri_refpos_out);
/* The rebuild itself (pure host): new node array (world tree copied, mesh trees rebuilt; at most max_nodes = n_nodes + 2 * n_tris +
 * n_instances), tri_order_out[new index] = index in scene->tris, blas_roots_out[instance] = its mesh root in the new array. */
int hiprz_rebuild_mesh_trees(const hiprz_scene* scene, uint32_t tree, hiprz_node* nodes_out, uint32_t max_nodes, uint32_t* n_nodes_out,
                             uint32_t* tri_order_out, uint32_t* blas_roots_out, uint32_t* tlas_root_out);

/* Stage the scene's geometry + shading records into LDS in every workgroup (ds_read instead of
 * dependent global loads): -1 = automatic (when the records fit three workgroups per CU), 0 = never,
 * 1 = whenever they fit one workgroup.  Results are identical either way. */
int hiprz_set_lds_scene(hiprz_ctx* ctx, int mode);

/* How the passes of hiprz_render are packaged into kernels.  1 = per pass a lean trace kernel followed by a shade
 * kernel, with a 20-byte hit record per pixel passed through device memory; 0 = per pass one fused kernel;
 * 2 = resident: ONE kernel per hiprz_render call — a workgroup takes its 32x8 tile through all the passes, path state
 * and accumulator stay on chip, the tone-mapped pixel is written on the way out (hiprz_tonemap then has nothing to do);
 * -1 (default) = 2 for scenes staged in LDS and for scenes without lights (those run per-wave chains of passes on the cooperative
 * walk), else 1.  Identical results. */
int hiprz_set_pipeline(hiprz_ctx* ctx, int pipeline);
int hiprz_pipeline(hiprz_ctx* ctx, int* effective_pipeline_out); /* valid after hiprz_upload_scene */
/* Reorder rays between passes (split pipeline): the shade kernel emits a sort key per pixel (cell of the next
 * ray's origin interleaved with the cell where that ray leaves the world box), a device radix sort turns the keys into a permutation, and the trace kernel
 * walks the rays in that order so that a wave's rays visit the same nodes.  -1 = automatic (on for scenes that are
 * not staged in LDS and have many instances), 0 = off, 1 = on.  Only the assignment of rays to threads changes; results are identical. */
int hiprz_set_ray_sort(hiprz_ctx* ctx, int mode);
/* XCD-aware workgroup -> tile mapping of the pass kernels (default off — it unbalances scenes whose cost is
 * concentrated in one image region): each of the 8 XCDs works through one
 * contiguous band of the owned tiles, so its L2 holds that band's part of the trees.  Execution order only. */
int hiprz_set_xcd_swizzle(hiprz_ctx* ctx, int enabled);
/* Replay the cumulative passes of a render call from a captured hipGraph (default on). */
int hiprz_set_graph(hiprz_ctx* ctx, int enabled);
/* How many times this context captured and instantiated a graph so far: steady-state frames with unchanged settings replay
 * the existing one (setters invalidate it only when a value really changes). */
int hiprz_graph_captures(hiprz_ctx* ctx, uint32_t* out);

/* --- rendering (replaces Renderer::renderFunction, cuda_engine_renderer.cu:73-262) --- */
/* Restart accumulation: the next hiprz_render starts with renderFirstPass
 * (CameraContext::reset, cpu_engine_renderer.cpp:32-39). */
int hiprz_reset(hiprz_ctx* ctx);
/* Trace n_passes passes (one path segment per owned pixel each): the first after a
 * reset is renderFirstPass (cpu_engine_kernel.cpp:15-57), the rest renderCumulativePass
 * (:58-101).  Asynchronous on the context's stream. */
int hiprz_render(hiprz_ctx* ctx, uint32_t n_passes);
/* Same, with the work counters enabled (instrumented kernel; synchronous). */
int hiprz_render_counted(hiprz_ctx* ctx, uint32_t n_passes, hiprz_counters* out);
/* Tone-map the accumulator into the RGBA8 image (cpu_engine_renderer.cpp:224-235). */
int hiprz_tonemap(hiprz_ctx* ctx);
int hiprz_sync(hiprz_ctx* ctx);

/* --- device→host readback (replaces EngineCore::CopyRenderToHost, cuda_engine_core.cu:129-242).
 * Row-major W*H images of the full frame; pixels not owned by this shard are zero. --- */
int hiprz_read_rgba8(hiprz_ctx* ctx, uint8_t* dst, size_t bytes);    /* W*H*4      */
int hiprz_read_depth(hiprz_ctx* ctx, float* dst, size_t bytes);      /* W*H*4      */
int hiprz_read_accum(hiprz_ctx* ctx, float* dst, size_t bytes);      /* W*H*16 RGBA32F, alpha = finished paths */
/* Per-pixel path state (CameraContext, cpu_engine_kernel.hpp:29-51) for parity tests:
 * origin[3], direction[3], color[3] as W*H*9 floats; material id u16 and depth u8 widened to u32 pairs. */
int hiprz_read_state(hiprz_ctx* ctx, float* ray9, uint32_t* material_depth2, size_t n_pixels);
int hiprz_ray_count(hiprz_ctx* ctx, uint64_t* out); /* Camera::rayCount, camera.hpp:113-114 */
int hiprz_pass_count(hiprz_ctx* ctx, uint32_t* out);

/* --- multi-GPU hand-off: tile-major device buffers for an RCCL gather.
 * Layout: local tile lt of the shard (hiprz_set_shard), 256 pixels each (4 waves of 8x8).  A context over n devices / streams
 * (hiprz_create_multi) hands out n slices of hiprz_local_pixel_capacity / n pixels each: slice r = the tiles of sub-shard rank * n + r of
 * world * n.  The slices of all ranks laid end to end are the sub-shards 0 .. world * n - 1: hiprz_untile_gathered(world * n parts). --- */
int hiprz_local_pixel_capacity(hiprz_ctx* ctx, size_t* out);          /* owned tiles * 256 (n slices of the job's largest sub-shard on a multi context) */
int hiprz_export_accum_tiles(hiprz_ctx* ctx, void* dst_device, size_t bytes);  /* float4 per local pixel, D2D on ctx stream */
/* The same for the tone-mapped output: u32 RGBA8 per local pixel (after hiprz_tonemap). */
int hiprz_export_rgba8_tiles(hiprz_ctx* ctx, void* dst_device, size_t bytes);
int hiprz_untile_rgba8(hiprz_ctx* ctx, const void* src_device_tiles, uint32_t rank, uint32_t world, void* dst_device_image);
/* Scatter tile-major float4 tiles of shard (rank, world) into a row-major W*H*16 device image. */
int hiprz_untile_accum(hiprz_ctx* ctx, const void* src_device_tiles, uint32_t rank, uint32_t world,
                       void* dst_device_image);
/* The gathered tiles of all `world` shards (shard r starts at src_parts + r * part_stride_bytes; element_bytes 4 =
 * RGBA8, 16 = float4) -> row-major W*H device image, in one launch on `stream` (a hipStream_t; NULL = the context's
 * stream) — the stream the gather itself ran on. */
int hiprz_untile_gathered(hiprz_ctx* ctx, const void* src_parts, uint32_t world, size_t part_stride_bytes, uint32_t element_bytes,
                          void* dst_device_image, void* stream);
/* Tone-map a row-major W*H float4 device image into W*H RGBA8 (device pointers). */
int hiprz_tonemap_image(hiprz_ctx* ctx, const void* src_device_image, void* dst_device_rgba8);
/* The same on `stream` (a hipStream_t; NULL = the context's stream): the stream a reduce of the accumulators ran on. */
int hiprz_tonemap_image_on(hiprz_ctx* ctx, const void* src_device_image, void* dst_device_rgba8, void* stream);
void* hiprz_stream(hiprz_ctx* ctx); /* the hipStream_t all of the above are enqueued on */

/* --- picking (Kernel::rayCast, cpu_engine_kernel.cpp:102-111, 483-501; Cuda: cuda_render_kernel.cu:130-144) ---
 * The reference casts one ray per camera and frame through the camera's ray-cast pixel (Camera::getRayCastPixel), in a thin shell
 * around the first-hit depth of that pixel, and stores the instance and the instance's material slot it met in the host camera
 * (m_raycasted_instance / m_raycasted_material, camera.hpp:55-56; writers cpu_engine_renderer.cpp:176, cuda_engine_core.cu:164-181).
 * hiprz_ray_cast is that query for the selected camera's current frame; both engine hosts call it after every frame. */
typedef struct hiprz_raycast {
    int32_t instance;      /* index into hiprz_scene::instances, -1 = nothing met */
    int32_t material_slot; /* the triangle's material id clamped to 0..63 = the slot Instance::material(slot) is asked for, -1 = nothing met */
    int32_t material;      /* index into hiprz_scene::materials of that slot, -1 = slot unset (or nothing met) */
    uint32_t triangle;     /* the triangle's source_index within its mesh */
} hiprz_raycast;
int hiprz_ray_cast(hiprz_ctx* ctx, uint32_t x, uint32_t y, hiprz_raycast* out);
int hiprz_pick(hiprz_ctx* ctx, uint32_t x, uint32_t y, int32_t* instance_out, int32_t* material_out); /* instance + material of hiprz_ray_cast */

/* Device self-test of the kernels' exact-arithmetic shortcuts (shared-reciprocal division must
 * equal the correctly rounded quotient): runs 262144 * cases_per_thread random cases. */
int hiprz_selftest(hiprz_ctx* ctx, uint32_t cases_per_thread, uint32_t seed, uint64_t* mismatches, uint64_t* tested);
/* Device self-test of the ray-order radix sort (hiprz_sort.hip; no reference counterpart — the reference traces pixels in tile order):
 * sorts the caller's n keys (host array, 1..32 significant bits from bit 0; the sort looks at whole bytes) `repeats` times and checks
 * on the host that the result is a STABLE permutation in key order and that the sorted keys are the keys in that order.
 * *errors = violations found, *sort_us = device time of the fastest repeat (HIP events on the context's stream). */
int hiprz_selftest_sort(hiprz_ctx* ctx, const uint32_t* keys_host, uint32_t n, int key_bits, uint32_t repeats, uint64_t* errors, double* sort_us);

/* --- timing (TimeTable, engine_parts.hpp:34-74; Engine::debugInfo, rayzath.cpp:96-113) --- */
int hiprz_timings(hiprz_ctx* ctx, char* buf, size_t len);
/* Kernel-level timing of the split pipeline: while hiprz_time_kernels(ctx, 1) is set, a batch of cumulative passes is
 * launched eagerly (not from the captured graph) with events before the trace kernel, between the two kernels and
 * after the shade kernel of each pass; hiprz_kernel_breakdown_ms returns the summed durations of the LAST such batch
 * and its pass count. */
int hiprz_time_kernels(hiprz_ctx* ctx, int enabled);
int hiprz_kernel_breakdown_ms(hiprz_ctx* ctx, double* trace_ms, double* shade_ms, uint32_t* passes);
/* Average device time of the pass kernel since the last call, from hip events recorded on the
 * context's stream around every hiprz_render() batch. */
int hiprz_kernel_time_ms(hiprz_ctx* ctx, double* total_ms, uint64_t* launches);

/* ---------------------------------------------------------------------------------
 * Host-side tree builders (pure CPU; no device needed).  They restate the reference's
 * builders — TreeNode::construct (RayZath/bvh_tree_node.hpp:117-215) and
 * ComponentTreeNode::construct (RayZath/component_container.hpp:259-363) — and emit the
 * flattened layout above.  A RayZath-side adapter would instead flatten the trees the
 * host World already built; these exist so the backend is usable
 * without the un-vendored host library.
 * ------------------------------------------------------------------------------- */
typedef struct hiprz_mesh_desc {
    uint32_t n_vertices;
    const float* vertices; /* xyz */
    uint32_t n_texcrds;
    const float* texcrds; /* uv */
    uint32_t n_normals;
    const float* normals; /* xyz */
    uint32_t n_triangles;
    const uint32_t* tri_vertices;  /* 3 per triangle */
    const uint32_t* tri_texcrds;   /* 3 per triangle, 0xFFFFFFFF = unused; may be NULL */
    const uint32_t* tri_normals;   /* 3 per triangle, 0xFFFFFFFF = unused; may be NULL */
    const uint32_t* tri_materials; /* 1 per triangle; may be NULL (=0) */
} hiprz_mesh_desc;

/* Builds the per-mesh tree.  Writes at most max_nodes nodes (node.begin of inner nodes is
 * relative to nodes_out[0]; leaf begin is relative to tris_out[0]) and exactly n_triangles
 * tris/attrs in leaf order.  Returns HIPRZ_OK and *n_nodes_out, or HIPRZ_ERR_INVALID if
 * max_nodes is too small (2*n_triangles+1 always suffices). */
int hiprz_build_mesh_tree(const hiprz_mesh_desc* mesh, hiprz_node* nodes_out, uint32_t max_nodes,
                          uint32_t* n_nodes_out, hiprz_tri* tris_out, hiprz_tri_attr* attrs_out);
/* The intersection + shading records of the triangles order[0..n) of a mesh (de-indexed vertices, texcrds, normals, face normal as
 * Triangle::calculateNormal, mesh_component.cpp:19-26): what hiprz_build_mesh_tree writes for its own leaf order.  For callers that
 * bring a tree of their own — the adapter that mirrors the reference's ComponentBVH (rayzath_adapter.hpp). */
int hiprz_fill_triangles(const hiprz_mesh_desc* mesh, const uint32_t* order, uint32_t n, hiprz_tri* tris_out, hiprz_tri_attr* attrs_out);
/* Builds the world tree over instance boxes (bb_min/bb_max of each instance; has_mesh[i]==0
 * instances are left out, bvh.hpp:40-47).  order_out receives instance ids in leaf order. */
int hiprz_build_world_tree(const hiprz_instance* instances, const uint8_t* has_mesh, uint32_t n_instances,
                           hiprz_node* nodes_out, uint32_t max_nodes, uint32_t* n_nodes_out,
                           uint32_t* order_out, uint32_t* n_order_out);
/* Instance::calculateBoundingBox (instance.cpp:117-155) for an instance without group. */
int hiprz_instance_bounds(const float* vertices, uint32_t n_vertices, hiprz_instance* inst);
/* CoordSystem::applyRotation (RotatedXYZ; render_parts.cpp:51-56) and ::lookAt (:57-62). */
void hiprz_axes_from_rotation(const float rotation[3], float x_axis[3], float y_axis[3], float z_axis[3]);
void hiprz_axes_look_at(const float rotation[3], float x_axis[3], float y_axis[3], float z_axis[3]);

/* seed table entry i (0..255) of pass `pass` for base seed `seed`: uniform in [-10, 10)
 * (distribution of RayZath/cuda_kernel_data.cu:10-18, made deterministic). */
float hiprz_seed_value(uint32_t seed, uint32_t pass, uint32_t i);

const char* hiprz_version(void);
/* How many kernel instantiations the library's launchers can select; each registered its host stub when the library was loaded.  The first
 * hiprz_create on a device resolves every one of them in the loaded gfx950 code objects and fails with HIPRZ_ERR_DEVICE, naming the kernel,
 * when one is missing (a launch of such a kernel would end the process inside the HIP runtime).  tools/check_kernels.py proves the same
 * for the built files; its count is this one. */
uint32_t hiprz_kernel_count(void);

#ifdef __cplusplus
}
#endif
#endif /* HIPRZ_H */
