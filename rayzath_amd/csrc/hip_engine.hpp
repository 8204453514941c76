// hip_engine.hpp — C++ host side of the HIPGPU backend, above the C-ABI (include/hiprz.h).
//
// `RayZath::Hip::Engine` has exactly the interface the facade calls on its backends
// (RayZath/cpu_engine.hpp:17-22, RayZath/cuda_engine.cuh:34-39):
//
//     void renderWorld(World&, const RenderConfig&, bool block = true, bool sync = true);
//     std::string timingsString();
//
// The reference's host `World` cannot be compiled in this image (un-vendored Math/Graphics
// headers), so this header carries a minimal stand-alone twin of the parts of it
// the render path reads — same object kinds, member names, defaults and clamping rules
// (world.hpp:64-76, material.hpp, mesh.hpp, instance.hpp, camera.hpp, spot_light.hpp,
// direct_light.hpp, engine_parts.hpp:76-128).  INTEGRATION.md shows the adapter that fills the
// same `hiprz_scene` from the real `RayZath::Engine::World` instead.
#pragma once

#include <array>
#include <cstdint>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "hiprz.h"

namespace RayZath::Hip {

// Cuda::Exception's peer (cuda_exception.hpp:9-19).  Built inside RayZath (-DHIPRZ_RAYZATH_BUILD) it derives from the
// facade's RayZath::Exception (rzexception.hpp:11-18), so the facade's `catch (RayZath::Exception&)` paths see it like a CUDA
// backend error; stand-alone (that header drags in the un-vendored CUDA driver types) it derives from the same std::runtime_error.
#ifdef HIPRZ_RAYZATH_BUILD
}  // namespace RayZath::Hip
#include "rzexception.hpp"
namespace RayZath::Hip {
using ExceptionBase = RayZath::Exception;
#else
using ExceptionBase = std::runtime_error;
#endif
struct Exception : public ExceptionBase {
    int code;
    Exception(int code_, const std::string& message) : ExceptionBase(message), code(code_) {}
};

struct vec3f {
    float x = 0, y = 0, z = 0;
};
struct Color {
    uint8_t red = 255, green = 255, blue = 255, alpha = 255;
};

// dirty-flag base, after Updatable/StateRegister (updatable.cpp:23-51)
class Updatable {
public:
    bool isModified() const { return m_modified; }
    void makeModified() { m_modified = true; }
    void makeUnmodified() { m_modified = false; }

private:
    bool m_modified = true;
};

struct TextureBuffer {  // render_parts.hpp:113-222
    uint32_t kind = HIPRZ_TEX_RGBA8, width = 0, height = 0;
    std::vector<uint8_t> bitmap;  // row-major, top row first
    float scale[2] = {1, 1}, rotation = 0, translation[2] = {0, 0};
    uint32_t sampling = HIPRZ_TEX_FILTER_POINT | HIPRZ_TEX_ADDRESS_WRAP;  // "filter mode" / "address mode" of the scene file: CUDA-compat mode only
};

struct Material {  // material.hpp:119-160; setters clamp as material.cpp:32-61
    Color color{0xC0, 0xC0, 0xC0, 0xFF};
    std::shared_ptr<TextureBuffer> texture, normal_map, metalness_map, roughness_map, emission_map;
    void metalness(float v) { m_metalness = v < 0 ? 0 : v > 1 ? 1 : v; }
    void roughness(float v) { m_roughness = v < 0 ? 0 : v > 1 ? 1 : v; }
    void emission(float v) { m_emission = v < 0 ? 0 : v; }
    void ior(float v) { m_ior = v < 1 ? 1 : v; }
    void scattering(float v) { m_scattering = v < 0 ? 0 : v; }
    float metalness() const { return m_metalness; }
    float roughness() const { return m_roughness; }
    float emission() const { return m_emission; }
    float ior() const { return m_ior; }
    float scattering() const { return m_scattering; }

private:
    float m_metalness = 0, m_roughness = 0, m_emission = 0, m_ior = 1.5f, m_scattering = 0;
};

struct Mesh {  // mesh.hpp
    static constexpr uint32_t ids_unused = 0xFFFFFFFFu;
    std::vector<float> vertices, texcrds, normals;                // xyz / uv / xyz
    std::vector<uint32_t> tri_vertices, tri_texcrds, tri_normals; // 3 per triangle
    std::vector<uint32_t> tri_materials;                          // 1 per triangle
    uint32_t createVertex(float x, float y, float z);
    uint32_t createTexcrd(float u, float v);
    uint32_t createTriangle(std::array<uint32_t, 3> vs, std::array<uint32_t, 3> ts = {ids_unused, ids_unused, ids_unused},
                            std::array<uint32_t, 3> ns = {ids_unused, ids_unused, ids_unused}, uint32_t material_id = 0);
    static std::shared_ptr<Mesh> generateCube();  // world.cpp:129-166
};

struct Group {  // group.hpp: a transformation over instances and sub-groups (groupable.hpp)
    vec3f position, rotation, scale{1, 1, 1};
    std::shared_ptr<Group> group;  // the group this one belongs to
};

struct Instance {  // instance.hpp:9-60
    static constexpr uint32_t materialCapacity() { return 64; }
    vec3f position, rotation, scale{1, 1, 1};
    std::shared_ptr<Mesh> mesh;
    std::array<std::shared_ptr<Material>, 64> materials;
    std::shared_ptr<Group> group;  // Groupable::group()
};

struct SpotLight {  // spot_light.hpp
    vec3f position, direction{0, -1, 0};
    Color color;
    float size = 0.5f, emission = 100.0f, beam_angle = 1.0f;
};
struct DirectLight {  // direct_light.hpp
    vec3f direction{0, -1, 0};
    Color color;
    float emission = 100.0f, angular_size = 0.1f;
};

struct Camera : public Updatable {  // camera.hpp:127-161
    vec3f position{0, 0, -10}, rotation;
    uint32_t width = 1280, height = 720;
    float fov = 1.57079632679f, near_plane = 1.0e-2f, far_plane = 1.0e3f;
    float focal_distance = 10.0f, aperture = 0.02f, exposure_time = 1.0f / 60.0f;
    float temporal_blend = 0.75f;  // camera.hpp:135; read with HIPRZ_COMPAT_REPROJECTION only
    bool enabled = true;  // camera.hpp:150; disabled cameras are skipped by the renderers (cpu_engine_renderer.cpp:99)
    // outputs the backend writes (camera.hpp:50-56, 113-119)
    std::vector<uint8_t> image_buffer;  // RGBA8 W*H
    std::vector<float> depth_buffer;    // W*H
    uint64_t ray_count = 0;
    // Kernel::rayCast (cpu_engine_kernel.cpp:102-111): after every frame, what the ray through this pixel meets — the editor's picking
    void rayCastPixel(uint32_t x, uint32_t y) {  // camera.cpp:159-165: clamped, marks the camera modified (accumulation goes on: the record the
                                                 // backend mirrors is unchanged, and neither reference engine restarts for MakeModified alone)
        ray_cast_pixel[0] = x >= width ? width - 1 : x, ray_cast_pixel[1] = y >= height ? height - 1 : y;
        makeModified();
    }
    uint32_t ray_cast_pixel[2] = {0, 0};
    std::shared_ptr<Instance> raycasted_instance;  // camera.hpp:55-56 (m_raycasted_instance / m_raycasted_material)
    std::shared_ptr<Material> raycasted_material;
};

struct World : public Updatable {  // world.hpp:64-76
    World();
    std::vector<std::shared_ptr<Material>> materials;
    std::vector<std::shared_ptr<Mesh>> meshes;
    std::vector<std::shared_ptr<Instance>> instances;
    std::vector<std::shared_ptr<SpotLight>> spot_lights;
    std::vector<std::shared_ptr<DirectLight>> direct_lights;
    Camera camera;                                  // the first camera ...
    std::vector<std::shared_ptr<Camera>> cameras;   // ... and the others: every enabled one is rendered per call (cpu_engine_renderer.cpp:97-117)
    std::vector<std::shared_ptr<Group>> groups;
    Material material;          // world / sky medium (world.cpp:33-38)
    Material default_material;  // world.cpp:39-43
    // How an instance inside groups is mirrored.  Cpu (default): what the CPU engine does — the bounding box comes from the
    // transformation composed through its groups (Instance::calculateBoundingBox, instance.cpp:125-155) while rays are taken into
    // the instance's OWN transformation (cpu_engine_kernel.cpp:308); the two agree only outside groups.  Cuda: the composed
    // transformation for both (cuda_instance.cu:244, transformationInGroup()).
    enum class GroupTransforms { Cpu, Cuda } group_transforms = GroupTransforms::Cpu;
    // materials / lights changed but no geometry: the engine replaces those records only (updatable.cpp:23-51 per container)
    void makeShadingModified() { m_shading_modified = true; }
    bool isShadingModified() const { return m_shading_modified; }
    void makeShadingUnmodified() { m_shading_modified = false; }
    // vertices of meshes and / or transformations of instances moved; the same meshes with the same triangles in the same instances
    // (an animation frame).  Where the context holds device-built trees the engine then refits them on the device and rebuilds the
    // world tree there (hiprz_update_triangles / hiprz_update_instances) instead of building every tree again on the host, as the
    // reference does at any change (component_container.hpp:259-363); otherwise it is an ordinary modification.
    void makeMoved() { m_moved = true; }
    bool isMoved() const { return m_moved; }
    void makeUnmoved() { m_moved = false; }

private:
    bool m_shading_modified = false, m_moved = false;
};

struct LightSampling {  // engine_parts.hpp:76-94
    uint8_t spot_light = 1, direct_light = 1;
};
struct Tracing {  // engine_parts.hpp:95-113
    uint8_t max_depth = 16;
    uint32_t rpp = 8;
};
struct RenderConfig {
    LightSampling light_sampling;
    Tracing tracing;
    uint32_t seed = 20240501u;
};

// The flattened snapshot with owning storage (what hiprz_upload_scene copies).
struct FlatScene {
    std::vector<hiprz_node> nodes;
    std::vector<uint32_t> tlas_order;
    std::vector<hiprz_tri> tris;
    std::vector<hiprz_tri_attr> tri_attrs;
    std::vector<hiprz_instance> instances;
    std::vector<int32_t> inst_materials;
    std::vector<hiprz_material> materials;
    std::vector<hiprz_texture> textures;
    std::vector<uint8_t> texels;
    std::vector<hiprz_spot_light> spot_lights;
    std::vector<hiprz_direct_light> direct_lights;
    std::vector<const void*> maps;  // the map objects behind `textures`, by identity, in texture-index order (host-side bookkeeping only)
    hiprz_scene view() const;
};
FlatScene flatten(const World& world);           // pure host
FlatScene flattenShading(const World& world);    // materials + lights only (for hiprz_update_shading): pure host
// triangles + instances only (for hiprz_update_triangles / hiprz_update_instances): the meshes' triangles in the order of an earlier
// flatten() (`uploaded_sources` = its tris[k].source_index), no tree is built.  Empty when the world no longer matches that order.
FlatScene flattenMotion(const World& world, const std::vector<uint32_t>& uploaded_sources);
hiprz_camera cameraRecord(const Camera& camera); // pure host

class Engine {
public:
    // throws Hip::Exception: the facade then falls back to CPU (rayzath.cpp:21-28).  `streams`: how many contexts-with-a-stream share
    // the GPU (hiprz_create_multi with the device named that often: one share's sorts, bookkeeping and kernel tails run beside another
    // share's walks); 0 = chosen when the first world is rendered — two for a world without lights, one otherwise (defaultStreams).
    explicit Engine(int device = 0, int streams = 0);
    explicit Engine(const std::vector<int>& devices);  // one context over several GPUs: tiles interleaved, peer-to-peer gather
    static int defaultStreams(const World& world) { return world.spot_lights.empty() && world.direct_lights.empty() ? 2 : 1; }
    void mode(uint32_t compat_flags);  // hiprz_set_mode: behaviours of the CUDA engine (default 0 = the CPU kernel)
    void tree(uint32_t tree);          // hiprz_set_tree, applied at the next scene upload
    // How the devices of Engine(devices) divide a frame (hiprz_set_shard_mode; no counterpart in the reference, which drives one device:
    // cuda_engine_core.cu:17).  Tiles (default): interleaved 32x8 tiles — the one-device frame bit for bit, a frame of `rpp` passes arrives
    // sooner: what an interactive host wants.  Samples: every device renders the whole frame on its own seed stream and the accumulators are
    // summed — a renderWorld call adds devices * rpp samples per pixel and rayCount() grows by as many rays; aggregate rays per second scale
    // with the devices where tile sharding is held back by its slowest tile: what a converging (headless / offline) host
    // wants — hiprz_headless --devices picks it.  Restarts accumulation.
    enum class ShardMode : uint32_t { Tiles = HIPRZ_SHARD_TILES, Samples = HIPRZ_SHARD_SAMPLES };
    void shardMode(ShardMode mode);
    ~Engine();
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;

    // `block` is accepted and ignored like in both reference backends.  sync = true: the camera buffers hold
    // THIS call's frame on return.  sync = false: the call only enqueues; the buffers are filled by the next call
    // (the CUDA backend's pipelined readback, cuda_engine_core.cu:115-120).  A device error met after a
    // non-sync call returned is stored and thrown by the next call (cuda_engine_core.cu:41).
    void renderWorld(World& world, const RenderConfig& render_config, bool block = true, bool sync = true);
    std::string timingsString();

    hiprz_ctx* context() {  // for tests: settles the stream count as it is
        m_streams_pending = false;
        return m_ctx;
    }

private:
    void check(int rc);
    void readback(Camera& camera, const World& world);
    std::vector<Camera*> enabledCameras(World& world) const;

    hiprz_ctx* m_ctx = nullptr;
    int m_device = 0;
    bool m_streams_pending = false;  // the context is still the single one of the constructor: the first world decides
    uint32_t m_mode = 0, m_tree = HIPRZ_TREE_AUTO;  // what mode() / tree() set (the hosts' default trees: per scene), for the context that replaces it
    std::mutex m_mutex;  // renderWorld is serialised (cpu_engine_core.cpp:15)
    bool m_pending_readback = false;
    std::unique_ptr<Exception> m_deferred;
    const World* m_last_world = nullptr;
    std::vector<const Camera*> m_camera_slots;  // camera k of the context mirrors this camera
    std::vector<hiprz_camera> m_camera_records; // ... as this record (a modified camera whose record is unchanged is not uploaded again)
    std::vector<const void*> m_uploaded_maps;   // the maps of the uploaded scene: hiprz_update_shading may only refer to these, by these indices
    std::vector<uint32_t> m_uploaded_sources;   // source_index of every uploaded triangle, in upload order (World::makeMoved: the order new vertices go up in)
    size_t m_uploaded_instances = 0;
    uint32_t m_moved_frames = 0;                // consecutive frames that went through the refit (World::makeMoved)
    static constexpr uint32_t kRebuildEvery = 16u;
};

}  // namespace RayZath::Hip
