// rayzath_adapter.hpp — mirrors a real `RayZath::Engine::World` into the snapshot the HIP backend uploads (include/hiprz.h), the way
// `Cuda::World::reconstruct` mirrors it into device objects (cuda_world.cu:28-75, cuda_object_container.cuh:40-161).
//
// Everything is read through the host library's PUBLIC interface, named as the reference names it:
//
//   world.container<ObjectType::X>()          world.hpp:93-114       count(), operator[] -> Handle<T>   object_container.hpp:65-72
//   world.material(), world.defaultMaterial() world.hpp:116-119      Handle: operator bool, operator->   roho.hpp:186-197
//   x.stateRegister().IsModified() / MakeUnmodified()                updatable.cpp:23-51
//   Material: color() metalness() roughness() emission() ior() scattering() map<ObjectType::Texture>() ...   material.hpp:80-108
//   TextureBuffer<T>: bitmap() (GetWidth, GetHeight, Value(x, y)) scale() rotation().value() translation() filterMode() addressMode()
//                                                                    render_parts.hpp:112-222
//   Mesh: vertices() texcrds() normals() triangles() (count(), operator[]), triangles().getBVH().rootNode()   mesh.hpp:37-44
//   Triangle: vertices texcrds normals (std::array<uint32_t, 3>), material_id                                   mesh_component.hpp:27-33
//   tree nodes: isLeaf() objects() children()->first / second / type, boundingBox().min / max   bvh_tree_node.hpp:60-90, component_container.hpp:206-231
//   Instance: transformation() transformationInGroup() (position() scale() coordSystem().xAxis() ...) boundingBox() mesh() material(i)
//                                                                    instance.hpp:46-60, groupable.hpp:26-27, render_parts.hpp:40-69
//   SpotLight / DirectLight / Camera getters                         spot_light.hpp:40-45, direct_light.hpp:36-39, camera.hpp:72-113
//
// The trees are the REFERENCE's own (ObjectContainerWithBVH::root(), ComponentBVH::rootNode()): the adapter only renumbers them into
// the flat layout (children adjacent, leaves in visiting order) — the same numbering hiprz_build_mesh_tree / hiprz_build_world_tree
// produce, so a world mirrored through the adapter and the same world flattened by Hip::flatten() give byte-identical snapshots
// (tests/test_adapter.py does exactly that with a test double that has the reference's member names).
//
// Incremental updates follow the reference's dirty flags per container (what the CUDA backend does, cuda_world.cu:69-75): nothing
// modified -> Change::None; only materials / lights (and no new map) -> Change::Shading, for hiprz_update_shading; anything else ->
// Change::Scene.  Flags are cleared as `reconstruct` clears them.
//
// This header compiles inside RayZath (-DHIPRZ_RAYZATH_BUILD): the host library's Math / Graphics headers are not
// vendored with the reference, so it cannot be compiled against the real world in this repository's image.
#pragma once

#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <vector>

#include "hip_engine.hpp"

namespace RayZath::Hip {

// Api::ObjectType is the host library's object-kind enumeration (typedefs.hpp:8-27).
template <class Api>
class WorldAdapter {
    using OT = typename Api::ObjectType;

public:
    // Moved: only vertices / transformations changed (same meshes, same triangles, same instances) AND the caller holds device-built
    // trees: scene() then has the new triangle records in the UPLOADED order and the new instance records — what
    // hiprz_update_triangles / hiprz_update_instances take; its nodes and leaf orders stay those of the uploaded snapshot.
    enum class Change { None, Shading, Scene, Moved };
    World::GroupTransforms group_transforms = World::GroupTransforms::Cpu;  // see hip_engine.hpp: what the CPU engine does / the CUDA engine

    template <class RZWorld>
    Change refresh(RZWorld& world, bool device_trees = false) {
        if (m_valid && !world.stateRegister().IsModified()) return Change::None;
        const bool maps_modified = modified<OT::Texture>(world) || modified<OT::NormalMap>(world) || modified<OT::MetalnessMap>(world) ||
                                   modified<OT::RoughnessMap>(world) || modified<OT::EmissionMap>(world);
        const bool geometry = !m_valid || modified<OT::Mesh>(world) || modified<OT::Instance>(world) || modified<OT::Group>(world) || maps_modified;
        // the world's flag is the OR of all its containers' (updatable.cpp:23-27): a camera that moved its ray-cast pixel, say, sets it
        // too, and is none of the scene mirror's business
        const bool shading_modified = modified<OT::Material>(world) || modified<OT::SpotLight>(world) || modified<OT::DirectLight>(world);
        if (!geometry && !shading_modified) {
            world.stateRegister().MakeUnmodified();
            return Change::None;
        }
        Change change = Change::Scene;
        if (m_valid && device_trees && geometry && !maps_modified && !shading_modified) {
            // meshes / instances / groups only: if the world still has the uploaded shape — the same instances using the same meshes and
            // material tables, every mesh with its triangle count — the new vertices and transformations go up in the uploaded order
            FlatScene s;
            std::vector<const void*> maps;
            std::vector<std::pair<uint32_t, uint32_t>> ranges;
            shading(world, s, maps, /*texels=*/false);
            geometry_of(world, s, ranges);
            if (maps == m_maps && same_shape(s, ranges)) {
                std::vector<hiprz_tri> tris(s.tris.size());
                std::vector<hiprz_tri_attr> attrs(s.tris.size());
                bool permutation = true;
                for (const auto& [first, count] : ranges) {
                    std::vector<uint32_t> where(count, 0xFFFFFFFFu);  // triangle of the mesh -> its place in the NEW leaf order
                    for (uint32_t k = 0; k < count; ++k)
                        if (s.tris[first + k].source_index < count) where[s.tris[first + k].source_index] = k;
                    for (uint32_t j = 0; j < count && permutation; ++j) {
                        const uint32_t source = m_scene.tris[first + j].source_index;
                        permutation = source < count && where[source] != 0xFFFFFFFFu;
                        if (permutation) tris[first + j] = s.tris[first + where[source]], attrs[first + j] = s.tri_attrs[first + where[source]];
                    }
                }
                if (permutation) {
                    for (size_t i = 0; i < s.instances.size(); ++i) s.instances[i].blas_root = m_scene.instances[i].blas_root;
                    m_scene.tris = std::move(tris), m_scene.tri_attrs = std::move(attrs), m_scene.instances = std::move(s.instances);
                    change = Change::Moved;
                }
            }
        }
        if (change == Change::Scene && !geometry) {
            // materials and lights only — unless a material now points at a map the uploaded scene does not hold
            FlatScene s;
            std::vector<const void*> maps;
            shading(world, s, maps, /*texels=*/false);
            if (maps == m_maps && s.materials.size() == m_scene.materials.size()) {
                m_scene.materials = s.materials, m_scene.spot_lights = s.spot_lights, m_scene.direct_lights = s.direct_lights;
                change = Change::Shading;
            }
        }
        if (change == Change::Scene) {
            FlatScene s;
            std::vector<const void*> maps;
            shading(world, s, maps, /*texels=*/true);
            geometry_of(world, s, m_mesh_ranges);
            m_scene = std::move(s), m_maps = std::move(maps), m_valid = true;
        }
        clear<OT::Texture>(world), clear<OT::NormalMap>(world), clear<OT::MetalnessMap>(world), clear<OT::RoughnessMap>(world);
        clear<OT::EmissionMap>(world), clear<OT::Material>(world), clear<OT::Mesh>(world), clear<OT::SpotLight>(world);
        clear<OT::DirectLight>(world), clear<OT::Instance>(world), clear<OT::Group>(world);
        world.stateRegister().MakeUnmodified();
        return change;
    }
    const FlatScene& scene() const { return m_scene; }

    // camera.hpp:93-111; the same clamping as Hip::cameraRecord (camera.cpp:95-110 applies it in the setters already)
    template <class RZCamera>
    static hiprz_camera cameraRecord(const RZCamera& cam) {
        const float eps = std::numeric_limits<float>::epsilon();
        hiprz_camera c{};
        put(c.position, cam.position());
        put(c.x_axis, cam.coordSystem().xAxis()), put(c.y_axis, cam.coordSystem().yAxis()), put(c.z_axis, cam.coordSystem().zAxis());
        c.width = std::max<uint32_t>(cam.width(), 1u), c.height = std::max<uint32_t>(cam.height(), 1u);
        c.fov = std::min(std::max(float(cam.fov().value()), eps), 3.14159265358979f - eps);
        c.tan_half_fov = std::tan(c.fov * 0.5f);
        c.aspect_ratio = float(c.width) / float(c.height);
        c.near_far[0] = std::max(float(cam.nearDistance()), eps);
        c.near_far[1] = std::max(float(cam.farDistance()), c.near_far[0] + eps);
        c.focal_distance = std::max(float(cam.focalDistance()), eps);
        c.aperture = std::max(float(cam.aperture()), eps);
        c.exposure_time = std::max(float(cam.exposureTime()), eps);
        return c;
    }

private:
    template <OT K, class RZWorld>
    static bool modified(RZWorld& world) {
        return world.template container<K>().stateRegister().IsModified();
    }
    template <OT K, class RZWorld>
    static void clear(RZWorld& world) {
        world.template container<K>().stateRegister().MakeUnmodified();
    }
    template <class V>
    static void put(float* dst, const V& v) {
        dst[0] = float(v.x), dst[1] = float(v.y), dst[2] = float(v.z);
    }
    template <class C>
    static void put_color(uint8_t* dst, const C& c) {
        dst[0] = c.red, dst[1] = c.green, dst[2] = c.blue, dst[3] = c.alpha;
    }
    static void normalize3(float* v) {
        const float s = 1.0f / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        v[0] *= s, v[1] *= s, v[2] *= s;
    }

    // ---- maps: numbered in first-use order over world material, default material, the world's materials (as Hip::flatten) ----
    static void texel(std::vector<uint8_t>& out, uint8_t v) { out.push_back(v); }
    static void texel(std::vector<uint8_t>& out, float v) {
        uint8_t b[4];
        std::memcpy(b, &v, 4);
        out.insert(out.end(), b, b + 4);
    }
    template <class C>
    static auto texel(std::vector<uint8_t>& out, const C& c) -> decltype(c.red, void()) {
        out.insert(out.end(), {uint8_t(c.red), uint8_t(c.green), uint8_t(c.blue), uint8_t(c.alpha)});
    }
    template <class Map>
    static int32_t map_id(const Map& map, uint32_t kind, FlatScene& s, std::vector<const void*>& maps, bool texels) {
        if (!map) return -1;
        const void* key = static_cast<const void*>(&*map);
        for (size_t i = 0; i < maps.size(); ++i)
            if (maps[i] == key) return int32_t(i);
        maps.push_back(key);
        if (texels) {
            while (s.texels.size() % 4) s.texels.push_back(0);
            hiprz_texture rec{};
            const auto& bitmap = map->bitmap();
            rec.kind = kind, rec.width = uint32_t(bitmap.GetWidth()), rec.height = uint32_t(bitmap.GetHeight()), rec.offset = uint32_t(s.texels.size());
            rec.scale[0] = float(map->scale().x), rec.scale[1] = float(map->scale().y);
            rec.translation[0] = float(map->translation().x), rec.translation[1] = float(map->translation().y);
            rec.rotation = float(map->rotation().value()), rec.cos_rotation = std::cos(rec.rotation), rec.sin_rotation = std::sin(rec.rotation);
            // FilterMode {Point, Linear}, AddressMode {Wrap, Clamp, Mirror, Border} (render_parts.hpp:97-108) in the order of HIPRZ_TEX_*
            rec.sampling = (uint32_t(map->filterMode()) == 1u ? HIPRZ_TEX_FILTER_LINEAR : HIPRZ_TEX_FILTER_POINT) | address_bits(uint32_t(map->addressMode()));
            for (size_t y = 0; y < bitmap.GetHeight(); ++y)
                for (size_t x = 0; x < bitmap.GetWidth(); ++x) texel(s.texels, bitmap.Value(x, y));
            s.textures.push_back(rec);
        }
        return int32_t(maps.size() - 1);
    }
    static uint32_t address_bits(uint32_t mode) {
        switch (mode) {
            case 1: return HIPRZ_TEX_ADDRESS_CLAMP;
            case 2: return HIPRZ_TEX_ADDRESS_MIRROR;
            case 3: return HIPRZ_TEX_ADDRESS_BORDER;
            default: return HIPRZ_TEX_ADDRESS_WRAP;
        }
    }

    template <class RZWorld>
    void shading(RZWorld& world, FlatScene& s, std::vector<const void*>& maps, bool texels) {
        auto add = [&](const auto& m) {
            hiprz_material r{};
            put_color(r.color, m.color());
            r.metalness = m.metalness(), r.roughness = m.roughness(), r.emission = m.emission(), r.ior = m.ior(), r.scattering = m.scattering();
            r.texture = map_id(m.template map<OT::Texture>(), HIPRZ_TEX_RGBA8, s, maps, texels);
            r.normal_map = map_id(m.template map<OT::NormalMap>(), HIPRZ_TEX_RGBA8, s, maps, texels);
            r.metalness_map = map_id(m.template map<OT::MetalnessMap>(), HIPRZ_TEX_R8, s, maps, texels);
            r.roughness_map = map_id(m.template map<OT::RoughnessMap>(), HIPRZ_TEX_R8, s, maps, texels);
            r.emission_map = map_id(m.template map<OT::EmissionMap>(), HIPRZ_TEX_R32F, s, maps, texels);
            s.materials.push_back(r);
        };
        add(world.material());
        add(world.defaultMaterial());
        auto& materials = world.template container<OT::Material>();
        m_material_index.clear();
        for (uint32_t i = 0; i < materials.count(); ++i) {
            m_material_index[static_cast<const void*>(&*materials[i])] = int32_t(s.materials.size());
            add(*materials[i]);
        }
        auto& spots = world.template container<OT::SpotLight>();
        for (uint32_t i = 0; i < spots.count(); ++i) {
            const auto& l = *spots[i];
            hiprz_spot_light r{};
            put(r.position, l.position()), put(r.direction, l.direction());
            normalize3(r.direction);
            r.size = std::max(float(l.size()), std::numeric_limits<float>::min()), r.emission = std::max(float(l.emission()), 0.0f);
            put_color(r.color, l.color());
            r.angle = std::min(std::max(float(l.GetBeamAngle()), 0.0f), 3.14159f), r.cos_angle = std::cos(r.angle);
            s.spot_lights.push_back(r);
        }
        auto& directs = world.template container<OT::DirectLight>();
        for (uint32_t i = 0; i < directs.count(); ++i) {
            const auto& l = *directs[i];
            hiprz_direct_light r{};
            put(r.direction, l.direction());
            normalize3(r.direction);
            r.emission = std::max(float(l.emission()), 0.0f);
            put_color(r.color, l.color());
            r.angular_size = std::min(std::max(float(l.angularSize()), 0.0f), 3.14159265358979f), r.cos_angular_size = std::cos(r.angular_size);
            s.direct_lights.push_back(r);
        }
    }

    // ---- trees: the reference's nodes renumbered depth first, the two children of a node adjacent (hiprz_node) ----
    template <class Node, class LeafFn>
    static void mirror_tree(const Node& node, uint32_t slot, std::vector<hiprz_node>& nodes, size_t base, uint32_t& n_order, LeafFn&& leaf) {
        hiprz_node n{};
        put(n.bb_min, node.boundingBox().min), put(n.bb_max, node.boundingBox().max);
        if (node.isLeaf()) {
            n.begin = n_order;
            n.meta = uint32_t(node.objects().size()) | HIPRZ_NODE_LEAF;
            for (const auto& object : node.objects()) leaf(object), ++n_order;
            nodes[base + slot] = n;
            return;
        }
        const uint32_t c = uint32_t(nodes.size() - base);
        nodes.resize(nodes.size() + 2);
        mirror_tree(node.children()->first, c, nodes, base, n_order, leaf);
        mirror_tree(node.children()->second, c + 1, nodes, base, n_order, leaf);
        n.begin = c;
        n.meta = uint32_t(node.children()->type) << HIPRZ_NODE_PTYPE_SHIFT;  // PartitionType X = 2, Y = 1, Z = 0, Size = 3 (bvh_tree_node.hpp:22-28)
        nodes[base + slot] = n;
    }

    // the freshly mirrored geometry against the uploaded one: instance for instance the same mesh (by its place in the first-use order),
    // the same material table; every mesh with the same number of triangles
    bool same_shape(const FlatScene& s, const std::vector<std::pair<uint32_t, uint32_t>>& ranges) const {
        if (ranges != m_mesh_ranges || s.tris.size() != m_scene.tris.size() || s.instances.size() != m_scene.instances.size() ||
            s.inst_materials != m_scene.inst_materials || s.materials.size() != m_scene.materials.size())
            return false;
        std::map<uint32_t, uint32_t> root_of;  // uploaded mesh root -> new mesh root
        for (size_t i = 0; i < s.instances.size(); ++i) {
            const hiprz_instance &a = m_scene.instances[i], &b = s.instances[i];
            if (a.material_base != b.material_base || a.material_count != b.material_count) return false;
            const auto it = root_of.find(a.blas_root);
            if (it == root_of.end()) root_of[a.blas_root] = b.blas_root;
            else if (it->second != b.blas_root) return false;
        }
        return true;
    }

    template <class RZWorld>
    void geometry_of(RZWorld& world, FlatScene& s, std::vector<std::pair<uint32_t, uint32_t>>& mesh_ranges) {
        mesh_ranges.clear();
        auto& instances = world.template container<OT::Instance>();
        const uint32_t n_inst = instances.count();
        std::map<const void*, uint32_t> instance_index;
        for (uint32_t i = 0; i < n_inst; ++i) instance_index[static_cast<const void*>(&*instances[i])] = i;

        // world tree (ObjectContainerWithBVH::root(), bvh.hpp:24): instances without a mesh are not in it (bvh.hpp:40-47)
        if (n_inst) {
            s.nodes.resize(1);
            uint32_t n_order = 0;
            mirror_tree(instances.root(), 0u, s.nodes, 0, n_order,
                        [&](const auto& handle) { s.tlas_order.push_back(instance_index.at(static_cast<const void*>(&*handle))); });
        }

        // one tree per distinct mesh, in the order the instances first use them
        std::map<const void*, uint32_t> mesh_root;
        for (uint32_t i = 0; i < n_inst; ++i) {
            const auto& inst = *instances[i];
            if (!inst.mesh() || mesh_root.count(static_cast<const void*>(&*inst.mesh()))) continue;
            const auto& mesh = *inst.mesh();
            std::vector<float> vertices, texcrds, normals;
            for (uint32_t k = 0; k < mesh.vertices().count(); ++k) { const auto& v = mesh.vertices()[k]; vertices.insert(vertices.end(), {float(v.x), float(v.y), float(v.z)}); }
            for (uint32_t k = 0; k < mesh.texcrds().count(); ++k) { const auto& t = mesh.texcrds()[k]; texcrds.insert(texcrds.end(), {float(t.x), float(t.y)}); }
            for (uint32_t k = 0; k < mesh.normals().count(); ++k) { const auto& v = mesh.normals()[k]; normals.insert(normals.end(), {float(v.x), float(v.y), float(v.z)}); }
            const uint32_t T = mesh.triangles().count();
            std::vector<uint32_t> tv(3 * size_t(T)), tt(3 * size_t(T)), tn(3 * size_t(T)), tm(T);
            for (uint32_t k = 0; k < T; ++k) {
                const auto& t = mesh.triangles()[k];
                for (int j = 0; j < 3; ++j) tv[3 * k + j] = t.vertices[j], tt[3 * k + j] = t.texcrds[j], tn[3 * k + j] = t.normals[j];
                tm[k] = t.material_id;
            }
            hiprz_mesh_desc d{};
            d.n_vertices = uint32_t(vertices.size() / 3), d.vertices = vertices.data();
            d.n_texcrds = uint32_t(texcrds.size() / 2), d.texcrds = texcrds.data();
            d.n_normals = uint32_t(normals.size() / 3), d.normals = normals.data();
            d.n_triangles = T, d.tri_vertices = tv.data(), d.tri_texcrds = tt.data(), d.tri_normals = tn.data(), d.tri_materials = tm.data();

            const size_t node_base = s.nodes.size(), tri_base = s.tris.size();
            mesh_root[static_cast<const void*>(&mesh)] = uint32_t(node_base);
            s.nodes.resize(node_base + 1);
            std::vector<uint32_t> order;
            uint32_t n_order = 0;
            const auto* first_triangle = T ? &mesh.triangles()[0] : nullptr;
            mirror_tree(mesh.triangles().getBVH().rootNode(), 0u, s.nodes, node_base, n_order,
                        [&](const auto* triangle) { order.push_back(uint32_t(triangle - first_triangle)); });
            for (size_t k = node_base; k < s.nodes.size(); ++k)
                s.nodes[k].begin += (s.nodes[k].meta & HIPRZ_NODE_LEAF) ? uint32_t(tri_base) : uint32_t(node_base);
            mesh_ranges.emplace_back(uint32_t(tri_base), uint32_t(order.size()));
            s.tris.resize(tri_base + order.size()), s.tri_attrs.resize(tri_base + order.size());
            if (hiprz_fill_triangles(&d, order.data(), uint32_t(order.size()), s.tris.data() + tri_base, s.tri_attrs.data() + tri_base) != HIPRZ_OK)
                throw Exception(HIPRZ_ERR_INVALID, "mesh with out-of-range indices");
        }

        for (uint32_t i = 0; i < n_inst; ++i) {
            const auto& inst = *instances[i];
            hiprz_instance r{};
            const auto& t = group_transforms == World::GroupTransforms::Cuda ? inst.transformationInGroup() : inst.transformation();
            put(r.position, t.position()), put(r.scale, t.scale());
            put(r.x_axis, t.coordSystem().xAxis()), put(r.y_axis, t.coordSystem().yAxis()), put(r.z_axis, t.coordSystem().zAxis());
            put(r.bb_min, inst.boundingBox().min), put(r.bb_max, inst.boundingBox().max);  // Instance::calculateBoundingBox: composed through the groups
            r.material_base = uint32_t(s.inst_materials.size());
            uint32_t count = 0;
            for (uint32_t k = 0; k < inst.materialCapacity(); ++k)
                if (inst.material(k)) count = k + 1;
            r.material_count = count;
            for (uint32_t k = 0; k < count; ++k) {
                const auto& m = inst.material(k);
                s.inst_materials.push_back(m ? m_material_index.at(static_cast<const void*>(&*m)) : -1);
            }
            if (inst.mesh()) r.blas_root = mesh_root.at(static_cast<const void*>(&*inst.mesh()));
            s.instances.push_back(r);
        }
    }

    FlatScene m_scene;
    std::vector<std::pair<uint32_t, uint32_t>> m_mesh_ranges;  // (first triangle, count) of every distinct mesh in the uploaded snapshot
    std::vector<const void*> m_maps;                    // the uploaded maps, by identity, in texture-index order
    std::map<const void*, int32_t> m_material_index;    // material object -> index in hiprz_scene::materials
    bool m_valid = false;
};

// renderWorld over a real world: what `Hip::Engine::renderWorld` does for the stand-alone twin (hip_engine.cpp), with the adapter in the
// place of Hip::flatten().  Every enabled camera is rendered into its own frame state and read back into the camera's own buffers
// (camera.hpp:113-119), as cpu_engine_renderer.cpp:97-117 / cuda_engine_core.cu:60-120 do.  `RZConfig` is the reference's RenderConfig
// (engine_parts.hpp:76-128).  sync = false leaves the readback to the next call, like the CUDA backend's pipelined frames.
template <class Api>
class WorldRenderer {
    using OT = typename Api::ObjectType;

public:
    explicit WorldRenderer(hiprz_ctx* ctx) : m_ctx(ctx) {}
    uint32_t movedFrames() const { return m_moved_frames; }
    WorldAdapter<Api>& adapter() { return m_adapter; }
    uint32_t seed = 20240501u;

    template <class RZWorld, class RZConfig>
    void renderWorld(RZWorld& world, const RZConfig& config, bool /*block*/ = true, bool sync = true) {
        auto& cameras = world.template container<OT::Camera>();
        std::vector<uint32_t> enabled;
        for (uint32_t i = 0; i < cameras.count(); ++i)
            if (cameras[i] && cameras[i]->enabled()) enabled.push_back(i);
        if (m_pending_readback) {
            m_pending_readback = false;
            for (size_t k = 0; k < m_slots.size(); ++k)
                for (uint32_t i : enabled)
                    if (static_cast<const void*>(&*cameras[i]) == m_slots[k]) check(hiprz_select_camera(m_ctx, uint32_t(k))), readback(world, *cameras[i]);
        }
        uint32_t tree = HIPRZ_TREE_REFERENCE;
        const bool device_trees = hiprz_tree(m_ctx, &tree) == HIPRZ_OK && (tree == HIPRZ_TREE_DEVICE || tree == HIPRZ_TREE_DEVICE_SAH);
        switch (m_adapter.refresh(world, device_trees)) {
            case WorldAdapter<Api>::Change::Moved: {  // an animation frame: the device refits its trees and rebuilds the world tree
                const FlatScene& s = m_adapter.scene();
                if (!s.tris.empty()) check(hiprz_update_triangles(m_ctx, 0u, uint32_t(s.tris.size()), s.tris.data(), s.tri_attrs.data()));
                if (!s.instances.empty()) check(hiprz_update_instances(m_ctx, s.instances.data(), uint32_t(s.instances.size())));
                if (++m_moved_frames % 16u == 0u) check(hiprz_rebuild_trees(m_ctx, tree));  // refitted trees keep their topology: now and then build again
                break;
            }
            case WorldAdapter<Api>::Change::Scene: {
                const hiprz_scene view = m_adapter.scene().view();
                check(hiprz_upload_scene(m_ctx, &view));
                m_moved_frames = 0;
                break;
            }
            case WorldAdapter<Api>::Change::Shading: {
                const FlatScene& s = m_adapter.scene();
                check(hiprz_update_shading(m_ctx, s.materials.data(), uint32_t(s.materials.size()), s.spot_lights.data(), uint32_t(s.spot_lights.size()),
                                           s.direct_lights.data(), uint32_t(s.direct_lights.size())));
                break;
            }
            case WorldAdapter<Api>::Change::None: break;
        }
        hiprz_config c{};
        c.max_depth = config.tracing().maxDepth(), c.rpp = config.tracing().rpp();
        c.spot_samples = std::max<uint32_t>(config.lightSampling().spotLight(), 1u);      // cuda_kernel_data.cu:23-31
        c.direct_samples = std::max<uint32_t>(config.lightSampling().directLight(), 1u);
        c.seed = seed;
        check(hiprz_set_config(m_ctx, &c));
        std::vector<const void*> slots;
        for (uint32_t i : enabled) slots.push_back(static_cast<const void*>(&*cameras[i]));
        const bool slots_changed = slots != m_slots;
        if (slots_changed) {
            check(hiprz_set_camera_count(m_ctx, uint32_t(std::max<size_t>(slots.size(), 1))));
            m_slots = slots;
            m_records.assign(slots.size(), hiprz_camera{});
        }
        for (size_t k = 0; k < enabled.size(); ++k) {
            auto& cam = *cameras[enabled[k]];
            check(hiprz_select_camera(m_ctx, uint32_t(k)));
            if (slots_changed || cam.stateRegister().IsModified()) {
                const hiprz_camera rec = WorldAdapter<Api>::cameraRecord(cam);
                // an upload restarts accumulation; Camera::rayCastPixel only marks the camera modified (camera.cpp:159-165) and neither
                // reference engine restarts for that, so an unchanged record is not uploaded again
                if (slots_changed || std::memcmp(&rec, &m_records[k], sizeof rec) != 0) {
                    check(hiprz_upload_camera(m_ctx, &rec));
                    m_records[k] = rec;
                }
                check(hiprz_set_temporal_blend(m_ctx, cam.temporalBlend()));  // camera.hpp:111
                cam.stateRegister().MakeUnmodified();
            }
            check(hiprz_render(m_ctx, std::max<uint32_t>(config.tracing().rpp(), 1u)));
            check(hiprz_tonemap(m_ctx));
            if (sync) readback(world, cam);
        }
        if (!sync) m_pending_readback = true;
    }

private:
    void check(int rc) {
        if (rc != HIPRZ_OK) throw Exception(rc, hiprz_last_error(m_ctx));
    }
    template <class RZWorld, class RZCamera>
    void readback(RZWorld& world, RZCamera& cam) {
        const uint32_t w = cam.width(), h = cam.height();
        m_rgba.resize(size_t(w) * h * 4), m_depth.resize(size_t(w) * h);
        check(hiprz_read_rgba8(m_ctx, m_rgba.data(), m_rgba.size()));
        check(hiprz_read_depth(m_ctx, m_depth.data(), m_depth.size() * sizeof(float)));
        auto& image = cam.imageBuffer();
        auto& depth = cam.depthBuffer();
        for (uint32_t y = 0; y < h; ++y)
            for (uint32_t x = 0; x < w; ++x) {
                const uint8_t* px = &m_rgba[(size_t(y) * w + x) * 4];
                auto& out = image.Value(x, y);  // by member name: the byte order of Graphics::Color is the host library's business
                out.red = px[0], out.green = px[1], out.blue = px[2], out.alpha = px[3];
                depth.Value(x, y) = m_depth[size_t(y) * w + x];
            }
        uint64_t rays = 0;
        check(hiprz_ray_count(m_ctx, &rays));
        cam.rayCount(rays);
        // Kernel::rayCast after every frame (cpu_engine_renderer.cpp:176), stored as Cuda::EngineCore does (cuda_engine_core.cu:164-181):
        // snapshot instance i IS instance i of the world's container (WorldAdapter::refresh flattens it in container order), and the
        // slot is the one Instance::material(slot) is asked for
        const auto pixel = cam.getRayCastPixel();
        hiprz_raycast hit{};
        check(hiprz_ray_cast(m_ctx, pixel.x, pixel.y, &hit));
        auto& instances = world.template container<OT::Instance>();
        if (hit.instance >= 0 && uint32_t(hit.instance) < instances.count()) {
            const auto& instance = instances[uint32_t(hit.instance)];
            cam.m_raycasted_instance = instance;
            if (hit.material_slot >= 0 && uint32_t(hit.material_slot) < instance->materialCapacity()) cam.m_raycasted_material = instance->material(uint32_t(hit.material_slot));
            else cam.m_raycasted_material.release();
        } else {
            cam.m_raycasted_instance.release();
            cam.m_raycasted_material.release();
        }
    }

    hiprz_ctx* m_ctx;
    WorldAdapter<Api> m_adapter;
    std::vector<const void*> m_slots;  // camera k of the context mirrors this camera object
    std::vector<hiprz_camera> m_records;  // ... as this record
    std::vector<uint8_t> m_rgba;
    std::vector<float> m_depth;
    bool m_pending_readback = false;
    uint32_t m_moved_frames = 0;  // consecutive frames that went through the refit (Change::Moved)
};

#ifdef HIPRZ_RAYZATH_BUILD
}  // namespace RayZath::Hip
#include "world.hpp"
namespace RayZath::Hip {
struct RayZathApi {
    using ObjectType = RayZath::Engine::ObjectType;
};
using RayZathWorldAdapter = WorldAdapter<RayZathApi>;
using RayZathWorldRenderer = WorldRenderer<RayZathApi>;
#endif

}  // namespace RayZath::Hip
