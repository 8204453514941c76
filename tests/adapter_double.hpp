// adapter_double.hpp — TEST DOUBLE for rayzath_amd/csrc/rayzath_adapter.hpp.
//
// The adapter is written against the public interface of RayZath's host library (world.hpp, object_container.hpp, material.hpp,
// mesh.hpp, bvh_tree_node.hpp, component_container.hpp, instance.hpp ...), which cannot be compiled in this image (its Math / Graphics
// headers are not vendored).  This file is NOT that library and not a build of it: it is a minimal object model that merely answers to
// the same member names, filled from a Hip::World twin and its flattened snapshot, so that the adapter's own logic — renumbering the
// pointer trees, map / material / instance indexing, dirty-flag handling — can be executed and compared with Hip::flatten().
// Deliberate differences from what the adapter might wrongly assume: colours are stored blue-first in memory (only member names are
// valid), containers keep their objects in separately allocated nodes, instance indices in the world tree come through handles.
#pragma once

#include <array>
#include <cstdint>
#include <map>
#include <memory>
#include <tuple>
#include <vector>

#include "hip_engine.hpp"

namespace Double {

enum class ObjectType { Texture, NormalMap, MetalnessMap, RoughnessMap, EmissionMap, Material, Mesh, Camera, SpotLight, DirectLight, Instance, Group };

struct StateRegister {
    bool modified = true;
    StateRegister* parent = nullptr;
    bool IsModified() const { return modified; }
    void MakeModified() {
        modified = true;
        if (parent) parent->MakeModified();
    }
    void MakeUnmodified() { modified = false; }
};
struct Updatable {
    StateRegister m_register;
    StateRegister& stateRegister() { return m_register; }
    const StateRegister& stateRegister() const { return m_register; }
};

template <class T>
struct Handle {
    std::shared_ptr<T> p;
    explicit operator bool() const noexcept { return bool(p); }
    T* operator->() const noexcept { return p.get(); }
    T& operator*() const noexcept { return *p; }
    void release() { p.reset(); }  // roho.hpp:201
};

struct vec3f { float x = 0, y = 0, z = 0; };
struct vec2f { float x = 0, y = 0; };
struct angle_radf {
    float v = 0;
    float value() const { return v; }
};
struct Color { uint8_t blue = 0, green = 0, red = 0, alpha = 0; };
struct BoundingBox { vec3f min, max; };

template <class T>
struct Buffer2D {
    size_t w = 0, h = 0;
    std::vector<T> data;
    size_t GetWidth() const { return w; }
    size_t GetHeight() const { return h; }
    const T& Value(size_t x, size_t y) const { return data[y * w + x]; }
    T& Value(size_t x, size_t y) { return data[y * w + x]; }
};
struct TextureBufferBase {
    enum class FilterMode { Point, Linear };
    enum class AddressMode { Wrap, Clamp, Mirror, Border };
};
template <class T>
struct TextureBuffer : Updatable, TextureBufferBase {
    Buffer2D<T> m_bitmap;
    vec2f m_scale, m_translation;
    angle_radf m_rotation;
    FilterMode m_filter = FilterMode::Point;
    AddressMode m_address = AddressMode::Wrap;
    const Buffer2D<T>& bitmap() const { return m_bitmap; }
    vec2f scale() const { return m_scale; }
    vec2f translation() const { return m_translation; }
    angle_radf rotation() const { return m_rotation; }
    FilterMode filterMode() const { return m_filter; }
    AddressMode addressMode() const { return m_address; }
};
using Texture = TextureBuffer<Color>;
using NormalMap = TextureBuffer<Color>;
using MetalnessMap = TextureBuffer<uint8_t>;
using RoughnessMap = TextureBuffer<uint8_t>;
using EmissionMap = TextureBuffer<float>;

struct Material : Updatable {
    Color m_color;
    float m_metalness = 0, m_roughness = 0, m_emission = 0, m_ior = 1, m_scattering = 0;
    std::tuple<Handle<Texture>, Handle<NormalMap>, Handle<MetalnessMap>, Handle<RoughnessMap>, Handle<EmissionMap>> m_maps;
    const Color& color() const noexcept { return m_color; }
    float metalness() const noexcept { return m_metalness; }
    float roughness() const noexcept { return m_roughness; }
    float emission() const noexcept { return m_emission; }
    float ior() const noexcept { return m_ior; }
    float scattering() const noexcept { return m_scattering; }
    template <ObjectType K>
    decltype(auto) map() const {
        return std::get<size_t(K)>(m_maps);
    }
};

struct Triangle {
    std::array<uint32_t, 3> vertices, texcrds, normals;
    uint32_t material_id = 0;
};
enum class PartitionType : uint8_t { X = 2, Y = 1, Z = 0, Size = 3 };
template <class Object>
struct TreeNode {
    struct Children {
        TreeNode first, second;
        PartitionType type;
    };
    std::unique_ptr<Children> m_children;
    std::vector<Object> m_objects;
    BoundingBox m_bb;
    const std::unique_ptr<Children>& children() const { return m_children; }
    const std::vector<Object>& objects() const { return m_objects; }
    const BoundingBox& boundingBox() const { return m_bb; }
    bool isLeaf() const { return !m_children; }
};
template <class T>
struct ComponentContainer {
    std::vector<T> items;
    uint32_t count() const { return uint32_t(items.size()); }
    const T& operator[](uint32_t i) const { return items[i]; }
};
struct TriangleBVH {
    TreeNode<const Triangle*> m_root;
    const TreeNode<const Triangle*>& rootNode() const { return m_root; }
};
struct TriangleContainer : ComponentContainer<Triangle> {
    TriangleBVH m_bvh;
    const TriangleBVH& getBVH() const { return m_bvh; }
};
struct Mesh : Updatable {
    ComponentContainer<vec3f> m_vertices, m_normals;
    ComponentContainer<vec2f> m_texcrds;
    TriangleContainer m_triangles;
    const ComponentContainer<vec3f>& vertices() const { return m_vertices; }
    const ComponentContainer<vec2f>& texcrds() const { return m_texcrds; }
    const ComponentContainer<vec3f>& normals() const { return m_normals; }
    const TriangleContainer& triangles() const { return m_triangles; }
};

struct CoordSystem {
    vec3f x, y, z;
    const vec3f xAxis() const { return x; }
    const vec3f yAxis() const { return y; }
    const vec3f zAxis() const { return z; }
};
struct Transformation {
    vec3f m_position, m_rotation, m_scale;
    CoordSystem m_coord_system;
    const vec3f& position() const { return m_position; }
    const vec3f& rotation() const { return m_rotation; }
    const vec3f& scale() const { return m_scale; }
    const CoordSystem& coordSystem() const { return m_coord_system; }
};
struct Instance : Updatable {
    Transformation m_transformation, m_transformation_in_group;
    BoundingBox m_bb;
    Handle<Mesh> m_mesh;
    std::array<Handle<Material>, 64> m_materials;
    const Transformation& transformation() const { return m_transformation; }
    const Transformation& transformationInGroup() const { return m_transformation_in_group; }
    const BoundingBox& boundingBox() const { return m_bb; }
    const Handle<Mesh>& mesh() const { return m_mesh; }
    const Handle<Material>& material(uint32_t i) const { return m_materials[i]; }
    static constexpr uint32_t materialCapacity() { return 64; }
};
struct Group : Updatable {};
struct SpotLight : Updatable {
    vec3f m_position, m_direction;
    Color m_color;
    float m_size = 0, m_emission = 0, m_angle = 0;
    const vec3f& position() const noexcept { return m_position; }
    const vec3f& direction() const noexcept { return m_direction; }
    const Color& color() const noexcept { return m_color; }
    float size() const noexcept { return m_size; }
    float emission() const noexcept { return m_emission; }
    float GetBeamAngle() const noexcept { return m_angle; }
};
struct DirectLight : Updatable {
    vec3f m_direction;
    Color m_color;
    float m_emission = 0, m_angular_size = 0;
    const vec3f direction() const noexcept { return m_direction; }
    const Color color() const noexcept { return m_color; }
    float emission() const noexcept { return m_emission; }
    float angularSize() const noexcept { return m_angular_size; }
};
struct Camera : Updatable {
    vec3f m_position;
    CoordSystem m_coord_system;
    uint32_t m_width = 0, m_height = 0;
    angle_radf m_fov;
    float m_near = 0, m_far = 0, m_focal = 0, m_aperture = 0, m_exposure = 0;
    bool m_enabled = true;
    float m_temporal_blend = 0.75f;
    uint64_t m_ray_count = 0;
    Buffer2D<Color> m_image;
    Buffer2D<float> m_depth;
    bool enabled() const { return m_enabled; }
    Buffer2D<Color>& imageBuffer() { return m_image; }
    Buffer2D<float>& depthBuffer() { return m_depth; }
    uint64_t rayCount() const { return m_ray_count; }
    void rayCount(uint64_t n) { m_ray_count = n; }
    uint32_t width() const { return m_width; }
    uint32_t height() const { return m_height; }
    const vec3f& position() const { return m_position; }
    const CoordSystem& coordSystem() const { return m_coord_system; }
    const angle_radf& fov() const { return m_fov; }
    const float& nearDistance() const { return m_near; }
    const float& farDistance() const { return m_far; }
    float focalDistance() const { return m_focal; }
    float aperture() const { return m_aperture; }
    float exposureTime() const { return m_exposure; }
    float temporalBlend() const { return m_temporal_blend; }
    // picking (camera.hpp:53-56, 90, 109)
    struct vec2ui32 { uint32_t x = 0, y = 0; };
    vec2ui32 m_ray_cast_pixel;
    vec2ui32 getRayCastPixel() const { return m_ray_cast_pixel; }
    void rayCastPixel(vec2ui32 pixel) {
        if (pixel.x >= m_width) pixel.x = m_width - 1;
        if (pixel.y >= m_height) pixel.y = m_height - 1;
        m_ray_cast_pixel = pixel;
        stateRegister().MakeModified();
    }
    Handle<Instance> m_raycasted_instance;
    Handle<Material> m_raycasted_material;
};

template <class T>
struct ObjectContainer : Updatable {
    std::vector<Handle<T>> m_objects;
    uint32_t count() const { return uint32_t(m_objects.size()); }
    const Handle<T>& operator[](uint32_t i) const { return m_objects[i]; }
    Handle<T> create() {
        m_objects.push_back(Handle<T>{std::make_shared<T>()});
        m_objects.back()->stateRegister().parent = &stateRegister();
        return m_objects.back();
    }
};
template <class T>
struct ObjectContainerWithBVH : ObjectContainer<T> {
    TreeNode<Handle<T>> m_root;
    const TreeNode<Handle<T>>& root() const { return m_root; }
};

struct World : Updatable {
    std::tuple<ObjectContainer<Texture>, ObjectContainer<NormalMap>, ObjectContainer<MetalnessMap>, ObjectContainer<RoughnessMap>,
               ObjectContainer<EmissionMap>, ObjectContainer<Material>, ObjectContainer<Mesh>, ObjectContainer<Camera>,
               ObjectContainer<SpotLight>, ObjectContainer<DirectLight>, ObjectContainerWithBVH<Instance>, ObjectContainer<Group>>
        m_containers;
    Material m_material, m_default_material;
    World() {
        std::apply([this](auto&... c) { ((c.stateRegister().parent = &stateRegister()), ...); }, m_containers);
        m_material.stateRegister().parent = &stateRegister(), m_default_material.stateRegister().parent = &stateRegister();
    }
    template <ObjectType K>
    auto& container() {
        return std::get<size_t(K)>(m_containers);
    }
    Material& material() { return m_material; }
    Material& defaultMaterial() { return m_default_material; }
};

struct Api {
    using ObjectType = Double::ObjectType;
};

struct LightSampling {  // engine_parts.hpp:76-94
    uint8_t m_spot = 1, m_direct = 1;
    uint8_t spotLight() const { return m_spot; }
    uint8_t directLight() const { return m_direct; }
};
struct Tracing {  // engine_parts.hpp:95-113
    uint8_t m_max_depth = 16;
    uint32_t m_rpp = 8;
    uint8_t maxDepth() const { return m_max_depth; }
    uint32_t rpp() const { return m_rpp; }
};
struct RenderConfig {
    LightSampling m_light_sampling;
    Tracing m_tracing;
    const LightSampling& lightSampling() const { return m_light_sampling; }
    const Tracing& tracing() const { return m_tracing; }
};

// ---- filled from a Hip::World twin + its flattened snapshot ----
namespace detail {
inline vec3f v3(const float* p) { return vec3f{p[0], p[1], p[2]}; }
inline Color color(const RayZath::Hip::Color& c) {
    Color o;
    o.red = c.red, o.green = c.green, o.blue = c.blue, o.alpha = c.alpha;
    return o;
}
template <class T>
void fill_map(TextureBuffer<T>& m, const RayZath::Hip::TextureBuffer& t) {
    m.m_bitmap.w = t.width, m.m_bitmap.h = t.height;
    m.m_scale = vec2f{t.scale[0], t.scale[1]}, m.m_translation = vec2f{t.translation[0], t.translation[1]}, m.m_rotation.v = t.rotation;
    m.m_filter = (t.sampling & 0xFFu) == HIPRZ_TEX_FILTER_LINEAR ? TextureBufferBase::FilterMode::Linear : TextureBufferBase::FilterMode::Point;
    m.m_address = TextureBufferBase::AddressMode(t.sampling >> 8);
}
// pointer tree from flat nodes (children adjacent at begin / begin + 1, leaves hold [begin, begin + count))
template <class Object, class LeafObject>
void tree_from_flat(const std::vector<hiprz_node>& nodes, size_t base, uint32_t slot, TreeNode<Object>& out, LeafObject&& object_of) {
    const hiprz_node& n = nodes[base + slot];
    out.m_bb.min = v3(n.bb_min), out.m_bb.max = v3(n.bb_max);
    if (n.meta & HIPRZ_NODE_LEAF) {
        for (uint32_t k = 0; k < (n.meta & HIPRZ_NODE_COUNT_MASK); ++k) out.m_objects.push_back(object_of(n.begin + k));
        return;
    }
    out.m_children.reset(new typename TreeNode<Object>::Children{});
    out.m_children->type = PartitionType((n.meta >> HIPRZ_NODE_PTYPE_SHIFT) & 3u);
    tree_from_flat(nodes, base, n.begin - uint32_t(base), out.m_children->first, object_of);
    tree_from_flat(nodes, base, n.begin + 1 - uint32_t(base), out.m_children->second, object_of);
}
}  // namespace detail

// `flat` = RayZath::Hip::flatten(twin) with twin.group_transforms == Cpu; `flat_cuda` the same with Cuda (for transformationInGroup)
inline std::unique_ptr<World> from_twin(const RayZath::Hip::World& twin, const RayZath::Hip::FlatScene& flat, const RayZath::Hip::FlatScene& flat_cuda) {
    namespace H = RayZath::Hip;
    auto w = std::make_unique<World>();
    std::map<const H::TextureBuffer*, Handle<Texture>> textures;
    std::map<const H::TextureBuffer*, Handle<NormalMap>> normal_maps;
    std::map<const H::TextureBuffer*, Handle<MetalnessMap>> r8_maps_m;
    std::map<const H::TextureBuffer*, Handle<RoughnessMap>> r8_maps_r;
    std::map<const H::TextureBuffer*, Handle<EmissionMap>> emission_maps;
    auto color_map = [&](const std::shared_ptr<H::TextureBuffer>& t, auto& cache, auto& container) {
        using HandleT = typename std::decay_t<decltype(cache)>::mapped_type;
        if (!t) return HandleT{};
        auto it = cache.find(t.get());
        if (it != cache.end()) return it->second;
        auto h = container.create();
        detail::fill_map(*h, *t);
        using Texel = std::decay_t<decltype(h->m_bitmap.data[0])>;
        const size_t n = size_t(t->width) * t->height;
        h->m_bitmap.data.resize(n);
        for (size_t i = 0; i < n; ++i) {
            if constexpr (std::is_same_v<Texel, Color>) {
                Color c;
                c.red = t->bitmap[4 * i], c.green = t->bitmap[4 * i + 1], c.blue = t->bitmap[4 * i + 2], c.alpha = t->bitmap[4 * i + 3];
                h->m_bitmap.data[i] = c;
            } else if constexpr (std::is_same_v<Texel, uint8_t>) {
                h->m_bitmap.data[i] = t->bitmap[i];
            } else {
                std::memcpy(&h->m_bitmap.data[i], &t->bitmap[4 * i], 4);
            }
        }
        return cache[t.get()] = h;
    };
    auto fill_material = [&](Material& m, const H::Material& s) {
        m.m_color = detail::color(s.color);
        m.m_metalness = s.metalness(), m.m_roughness = s.roughness(), m.m_emission = s.emission(), m.m_ior = s.ior(), m.m_scattering = s.scattering();
        std::get<0>(m.m_maps) = color_map(s.texture, textures, w->container<ObjectType::Texture>());
        std::get<1>(m.m_maps) = color_map(s.normal_map, normal_maps, w->container<ObjectType::NormalMap>());
        std::get<2>(m.m_maps) = color_map(s.metalness_map, r8_maps_m, w->container<ObjectType::MetalnessMap>());
        std::get<3>(m.m_maps) = color_map(s.roughness_map, r8_maps_r, w->container<ObjectType::RoughnessMap>());
        std::get<4>(m.m_maps) = color_map(s.emission_map, emission_maps, w->container<ObjectType::EmissionMap>());
    };
    fill_material(w->material(), twin.material);
    fill_material(w->defaultMaterial(), twin.default_material);
    std::map<const H::Material*, Handle<Material>> materials;
    for (const auto& m : twin.materials) {
        auto h = w->container<ObjectType::Material>().create();
        fill_material(*h, *m);
        materials[m.get()] = h;
    }
    std::map<const H::Mesh*, Handle<Mesh>> meshes;
    auto& instances = w->container<ObjectType::Instance>();
    for (size_t i = 0; i < twin.instances.size(); ++i) {
        const auto& src = *twin.instances[i];
        auto inst = instances.create();
        const hiprz_instance &r = flat.instances[i], &rc = flat_cuda.instances[i];
        auto xform = [](const hiprz_instance& q) {
            Transformation t;
            t.m_position = detail::v3(q.position), t.m_scale = detail::v3(q.scale);
            t.m_coord_system = CoordSystem{detail::v3(q.x_axis), detail::v3(q.y_axis), detail::v3(q.z_axis)};
            return t;
        };
        inst->m_transformation = xform(r), inst->m_transformation_in_group = xform(rc);
        inst->m_bb = BoundingBox{detail::v3(r.bb_min), detail::v3(r.bb_max)};
        for (uint32_t k = 0; k < 64; ++k)
            if (src.materials[k]) inst->m_materials[k] = materials.at(src.materials[k].get());
        if (!src.mesh) continue;
        auto it = meshes.find(src.mesh.get());
        if (it == meshes.end()) {
            auto mesh = w->container<ObjectType::Mesh>().create();
            const H::Mesh& m = *src.mesh;
            for (size_t k = 0; k + 2 < m.vertices.size(); k += 3) mesh->m_vertices.items.push_back(vec3f{m.vertices[k], m.vertices[k + 1], m.vertices[k + 2]});
            for (size_t k = 0; k + 1 < m.texcrds.size(); k += 2) mesh->m_texcrds.items.push_back(vec2f{m.texcrds[k], m.texcrds[k + 1]});
            for (size_t k = 0; k + 2 < m.normals.size(); k += 3) mesh->m_normals.items.push_back(vec3f{m.normals[k], m.normals[k + 1], m.normals[k + 2]});
            for (size_t k = 0; k < m.tri_materials.size(); ++k) {
                Triangle t;
                for (int j = 0; j < 3; ++j) t.vertices[j] = m.tri_vertices[3 * k + j], t.texcrds[j] = m.tri_texcrds[3 * k + j], t.normals[j] = m.tri_normals[3 * k + j];
                t.material_id = m.tri_materials[k];
                mesh->m_triangles.items.push_back(t);
            }
            // the mesh's tree, from the flat snapshot: nodes [blas_root, ...) with leaf begins relative to the global triangle array
            const uint32_t root = r.blas_root;
            const Triangle* first = mesh->m_triangles.items.data();
            detail::tree_from_flat(flat.nodes, root, 0u, mesh->m_triangles.m_bvh.m_root,
                                   [&](uint32_t tri) { return first + flat.tris[tri].source_index; });
            it = meshes.emplace(src.mesh.get(), mesh).first;
        }
        inst->m_mesh = it->second;
    }
    if (!flat.nodes.empty() && !twin.instances.empty())
        detail::tree_from_flat(flat.nodes, 0, 0u, instances.m_root, [&](uint32_t k) { return instances[flat.tlas_order[k]]; });
    for (const auto& l : twin.spot_lights) {
        auto h = w->container<ObjectType::SpotLight>().create();
        h->m_position = vec3f{l->position.x, l->position.y, l->position.z}, h->m_direction = vec3f{l->direction.x, l->direction.y, l->direction.z};
        h->m_color = detail::color(l->color), h->m_size = l->size, h->m_emission = l->emission, h->m_angle = l->beam_angle;
    }
    for (const auto& l : twin.direct_lights) {
        auto h = w->container<ObjectType::DirectLight>().create();
        h->m_direction = vec3f{l->direction.x, l->direction.y, l->direction.z};
        h->m_color = detail::color(l->color), h->m_emission = l->emission, h->m_angular_size = l->angular_size;
    }
    return w;
}

inline Handle<Camera> add_camera(World& w, const RayZath::Hip::Camera& src) {
    auto cam = w.container<ObjectType::Camera>().create();
    const hiprz_camera ref = RayZath::Hip::cameraRecord(src);
    cam->m_position = detail::v3(ref.position);
    cam->m_coord_system = CoordSystem{detail::v3(ref.x_axis), detail::v3(ref.y_axis), detail::v3(ref.z_axis)};
    cam->m_width = src.width, cam->m_height = src.height, cam->m_fov.v = src.fov;
    cam->m_near = src.near_plane, cam->m_far = src.far_plane, cam->m_focal = src.focal_distance;
    cam->m_aperture = src.aperture, cam->m_exposure = src.exposure_time, cam->m_enabled = src.enabled;
    cam->m_image.w = cam->m_depth.w = src.width, cam->m_image.h = cam->m_depth.h = src.height;
    cam->m_image.data.resize(size_t(src.width) * src.height), cam->m_depth.data.resize(size_t(src.width) * src.height);
    return cam;
}

}  // namespace Double
