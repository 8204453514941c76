// hip_engine.cpp — see hip_engine.hpp.  Host-only C++ (no device code): everything below the
// C-ABI lives in hiprz_api.hip / hiprz_host.cpp.
#include "hip_engine.hpp"

#include <cmath>
#include <cstring>
#include <map>

namespace RayZath::Hip {

// ---- Mesh ----
uint32_t Mesh::createVertex(float x, float y, float z) {
    vertices.insert(vertices.end(), {x, y, z});
    return uint32_t(vertices.size() / 3 - 1);
}
uint32_t Mesh::createTexcrd(float u, float v) {
    texcrds.insert(texcrds.end(), {u, v});
    return uint32_t(texcrds.size() / 2 - 1);
}
uint32_t Mesh::createTriangle(std::array<uint32_t, 3> vs, std::array<uint32_t, 3> ts, std::array<uint32_t, 3> ns,
                              uint32_t material_id) {
    tri_vertices.insert(tri_vertices.end(), vs.begin(), vs.end());
    tri_texcrds.insert(tri_texcrds.end(), ts.begin(), ts.end());
    tri_normals.insert(tri_normals.end(), ns.begin(), ns.end());
    tri_materials.push_back(material_id);
    return uint32_t(tri_materials.size() - 1);
}
std::shared_ptr<Mesh> Mesh::generateCube() {  // world.cpp:129-166
    auto m = std::make_shared<Mesh>();
    const float v[8][3] = {{-.5f, .5f, -.5f}, {-.5f, .5f, .5f}, {.5f, .5f, .5f}, {.5f, .5f, -.5f},
                           {-.5f, -.5f, -.5f}, {-.5f, -.5f, .5f}, {.5f, -.5f, .5f}, {.5f, -.5f, -.5f}};
    for (auto& p : v) m->createVertex(p[0], p[1], p[2]);
    m->createTexcrd(0, 0), m->createTexcrd(0, 1), m->createTexcrd(1, 1), m->createTexcrd(1, 0);
    const uint32_t t[12][3] = {{1, 2, 0}, {3, 0, 2}, {4, 7, 5}, {6, 5, 7}, {0, 3, 4}, {7, 4, 3},
                               {2, 1, 6}, {5, 6, 1}, {3, 2, 7}, {6, 7, 2}, {1, 0, 5}, {4, 5, 0}};
    for (int i = 0; i < 12; ++i)
        m->createTriangle({t[i][0], t[i][1], t[i][2]}, i % 2 == 0 ? std::array<uint32_t, 3>{1, 2, 0} : std::array<uint32_t, 3>{3, 0, 2});
    return m;
}

World::World() {
    material.color = Color{0xFF, 0xFF, 0xFF, 0x00};  // world.cpp:33-38
    material.ior(1.0f);
    default_material.color = Color{0xC0, 0xC0, 0xC0, 0xFF};  // Palette::LightGrey (value assumed)
}

// ---- flattening ----
hiprz_scene FlatScene::view() const {
    hiprz_scene s{};
    s.n_nodes = uint32_t(nodes.size()), s.nodes = nodes.data();
    s.tlas_root = 0;
    s.n_tlas_order = uint32_t(tlas_order.size()), s.tlas_order = tlas_order.data();
    s.n_tris = uint32_t(tris.size()), s.tris = tris.data(), s.tri_attrs = tri_attrs.data();
    s.n_instances = uint32_t(instances.size()), s.instances = instances.data();
    s.n_inst_materials = uint32_t(inst_materials.size()), s.inst_materials = inst_materials.data();
    s.n_materials = uint32_t(materials.size()), s.materials = materials.data();
    s.n_textures = uint32_t(textures.size()), s.textures = textures.data();
    s.texel_bytes = texels.size(), s.texels = texels.data();
    s.n_spot_lights = uint32_t(spot_lights.size()), s.spot_lights = spot_lights.data();
    s.n_direct_lights = uint32_t(direct_lights.size()), s.direct_lights = direct_lights.data();
    return s;
}

namespace {
void put3(float* dst, const vec3f& v) { dst[0] = v.x, dst[1] = v.y, dst[2] = v.z; }
void normalize3(float* v) {
    const float s = 1.0f / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] *= s, v[1] *= s, v[2] *= s;
}
}  // namespace

namespace {
// Transformation (render_parts.hpp:39-69) of an instance, composed through its groups as Transformation::operator*= does
// (render_parts.cpp:75-82): position rotated by the group's axes and moved by its position, axes rotated, scales multiplied.
struct Xform {
    float p[3], s[3], x[3], y[3], z[3];
};
void forward(const Xform& g, float* v) {  // CoordSystem::transformForward: x_axis * v.x + y_axis * v.y + z_axis * v.z
    const float a = v[0], b = v[1], c = v[2];
    for (int k = 0; k < 3; ++k) v[k] = (g.x[k] * a + g.y[k] * b) + g.z[k] * c;
}
Xform own_transform(const vec3f& position, const vec3f& rotation, const vec3f& scale) {
    Xform t;
    put3(t.p, position), put3(t.s, scale);
    float rot[3];
    put3(rot, rotation);
    hiprz_axes_from_rotation(rot, t.x, t.y, t.z);
    return t;
}
Xform in_group(const Instance& inst) {  // Instance::calculateBoundingBox, instance.cpp:125-133
    Xform t = own_transform(inst.position, inst.rotation, inst.scale);
    for (const Group* g = inst.group.get(); g; g = g->group.get()) {
        const Xform gt = own_transform(g->position, g->rotation, g->scale);
        forward(gt, t.p);
        for (int k = 0; k < 3; ++k) t.p[k] += gt.p[k];
        forward(gt, t.x), forward(gt, t.y), forward(gt, t.z);
        for (int k = 0; k < 3; ++k) t.s[k] *= gt.s[k];
    }
    return t;
}
void store(hiprz_instance& r, const Xform& t) {
    std::memcpy(r.position, t.p, 12), std::memcpy(r.scale, t.s, 12);
    std::memcpy(r.x_axis, t.x, 12), std::memcpy(r.y_axis, t.y, 12), std::memcpy(r.z_axis, t.z, 12);
}
void flatten_lights(const World& world, FlatScene& f) {
    for (const auto& l : world.spot_lights) {
        hiprz_spot_light r{};
        put3(r.position, l->position), put3(r.direction, l->direction);
        normalize3(r.direction);
        r.size = std::max(l->size, std::numeric_limits<float>::min()), r.emission = std::max(l->emission, 0.0f);
        r.color[0] = l->color.red, r.color[1] = l->color.green, r.color[2] = l->color.blue, r.color[3] = l->color.alpha;
        r.angle = std::min(std::max(l->beam_angle, 0.0f), 3.14159f), r.cos_angle = std::cos(r.angle);
        f.spot_lights.push_back(r);
    }
    for (const auto& l : world.direct_lights) {
        hiprz_direct_light r{};
        put3(r.direction, l->direction);
        normalize3(r.direction);
        r.emission = std::max(l->emission, 0.0f);
        r.color[0] = l->color.red, r.color[1] = l->color.green, r.color[2] = l->color.blue, r.color[3] = l->color.alpha;
        r.angular_size = std::min(std::max(l->angular_size, 0.0f), 3.14159265358979f), r.cos_angular_size = std::cos(r.angular_size);
        f.direct_lights.push_back(r);
    }
}
}  // namespace

FlatScene flatten(const World& world) {
    FlatScene f;
    std::map<const TextureBuffer*, int32_t> tex_index;
    auto tex_id = [&](const std::shared_ptr<TextureBuffer>& t) -> int32_t {
        if (!t) return -1;
        auto it = tex_index.find(t.get());
        if (it != tex_index.end()) return it->second;
        while (f.texels.size() % 4) f.texels.push_back(0);
        hiprz_texture rec{};
        rec.kind = t->kind, rec.width = t->width, rec.height = t->height, rec.offset = uint32_t(f.texels.size());
        rec.scale[0] = t->scale[0], rec.scale[1] = t->scale[1];
        rec.translation[0] = t->translation[0], rec.translation[1] = t->translation[1];
        rec.rotation = t->rotation, rec.cos_rotation = std::cos(t->rotation), rec.sin_rotation = std::sin(t->rotation);
        rec.sampling = t->sampling;
        f.texels.insert(f.texels.end(), t->bitmap.begin(), t->bitmap.end());
        f.textures.push_back(rec);
        f.maps.push_back(t.get());
        return tex_index[t.get()] = int32_t(f.textures.size() - 1);
    };
    std::map<const Material*, int32_t> mat_index;
    auto add_material = [&](const Material& m) {
        hiprz_material r{};
        r.color[0] = m.color.red, r.color[1] = m.color.green, r.color[2] = m.color.blue, r.color[3] = m.color.alpha;
        r.metalness = m.metalness(), r.roughness = m.roughness(), r.emission = m.emission(), r.ior = m.ior(), r.scattering = m.scattering();
        r.texture = tex_id(m.texture), r.normal_map = tex_id(m.normal_map), r.metalness_map = tex_id(m.metalness_map);
        r.roughness_map = tex_id(m.roughness_map), r.emission_map = tex_id(m.emission_map);
        mat_index[&m] = int32_t(f.materials.size());
        f.materials.push_back(r);
    };
    add_material(world.material);
    add_material(world.default_material);
    for (const auto& m : world.materials) add_material(*m);

    // one tree per distinct mesh
    struct MeshTree {
        std::vector<hiprz_node> nodes;
        std::vector<hiprz_tri> tris;
        std::vector<hiprz_tri_attr> attrs;
    };
    std::map<const Mesh*, size_t> mesh_slot;
    std::vector<MeshTree> trees;
    for (const auto& inst : world.instances) {
        if (!inst->mesh || mesh_slot.count(inst->mesh.get())) continue;
        const Mesh& m = *inst->mesh;
        hiprz_mesh_desc d{};
        d.n_vertices = uint32_t(m.vertices.size() / 3), d.vertices = m.vertices.data();
        d.n_texcrds = uint32_t(m.texcrds.size() / 2), d.texcrds = m.texcrds.data();
        d.n_normals = uint32_t(m.normals.size() / 3), d.normals = m.normals.data();
        d.n_triangles = uint32_t(m.tri_materials.size());
        d.tri_vertices = m.tri_vertices.data(), d.tri_texcrds = m.tri_texcrds.data();
        d.tri_normals = m.tri_normals.data(), d.tri_materials = m.tri_materials.data();
        MeshTree t;
        t.nodes.resize(2 * size_t(d.n_triangles) + 1);
        t.tris.resize(d.n_triangles ? d.n_triangles : 1);
        t.attrs.resize(d.n_triangles ? d.n_triangles : 1);
        uint32_t n = 0;
        if (hiprz_build_mesh_tree(&d, t.nodes.data(), uint32_t(t.nodes.size()), &n, t.tris.data(), t.attrs.data()) != HIPRZ_OK)
            throw Exception(HIPRZ_ERR_INVALID, "mesh with out-of-range indices");
        t.nodes.resize(n), t.tris.resize(d.n_triangles), t.attrs.resize(d.n_triangles);
        mesh_slot[&m] = trees.size();
        trees.push_back(std::move(t));
    }

    std::vector<uint8_t> has_mesh;
    for (const auto& inst : world.instances) {
        hiprz_instance r{};
        const Xform own = own_transform(inst->position, inst->rotation, inst->scale), grouped = in_group(*inst);
        store(r, grouped);  // the bounding box always comes from the composed transformation (instance.cpp:125-155)
        r.material_base = uint32_t(f.inst_materials.size());
        uint32_t count = 0;
        for (uint32_t k = 0; k < Instance::materialCapacity(); ++k)
            if (inst->materials[k]) count = k + 1;
        r.material_count = count;
        for (uint32_t k = 0; k < count; ++k) {
            const auto& m = inst->materials[k];
            if (m && !mat_index.count(m.get())) throw Exception(HIPRZ_ERR_INVALID, "instance uses a material that is not in the world");
            f.inst_materials.push_back(m ? mat_index[m.get()] : -1);
        }
        if (inst->mesh) hiprz_instance_bounds(inst->mesh->vertices.data(), uint32_t(inst->mesh->vertices.size() / 3), &r);
        if (world.group_transforms == World::GroupTransforms::Cpu) store(r, own);  // ... but the CPU kernel takes rays into the instance's OWN one (cpu_engine_kernel.cpp:308)
        has_mesh.push_back(inst->mesh ? 1 : 0);
        f.instances.push_back(r);
    }

    const uint32_t n_inst = uint32_t(f.instances.size());
    if (n_inst) {
        f.nodes.resize(2 * size_t(n_inst) + 1);
        f.tlas_order.resize(n_inst);
        uint32_t n = 0, n_order = 0;
        hiprz_build_world_tree(f.instances.data(), has_mesh.data(), n_inst, f.nodes.data(), uint32_t(f.nodes.size()), &n,
                               f.tlas_order.data(), &n_order);
        f.nodes.resize(n), f.tlas_order.resize(n_order);
    }
    std::vector<uint32_t> roots;
    for (auto& t : trees) {
        const uint32_t node_base = uint32_t(f.nodes.size()), tri_base = uint32_t(f.tris.size());
        roots.push_back(node_base);
        for (auto n : t.nodes) {
            n.begin += (n.meta & HIPRZ_NODE_LEAF) ? tri_base : node_base;
            f.nodes.push_back(n);
        }
        f.tris.insert(f.tris.end(), t.tris.begin(), t.tris.end());
        f.tri_attrs.insert(f.tri_attrs.end(), t.attrs.begin(), t.attrs.end());
    }
    for (size_t i = 0; i < world.instances.size(); ++i)
        if (world.instances[i]->mesh) f.instances[i].blas_root = roots[mesh_slot[world.instances[i]->mesh.get()]];

    flatten_lights(world, f);
    return f;
}

namespace {
hiprz_mesh_desc mesh_desc(const Mesh& m) {
    hiprz_mesh_desc d{};
    d.n_vertices = uint32_t(m.vertices.size() / 3), d.vertices = m.vertices.data();
    d.n_texcrds = uint32_t(m.texcrds.size() / 2), d.texcrds = m.texcrds.data();
    d.n_normals = uint32_t(m.normals.size() / 3), d.normals = m.normals.data();
    d.n_triangles = uint32_t(m.tri_materials.size());
    d.tri_vertices = m.tri_vertices.data(), d.tri_texcrds = m.tri_texcrds.data();
    d.tri_normals = m.tri_normals.data(), d.tri_materials = m.tri_materials.data();
    return d;
}
}  // namespace

FlatScene flattenMotion(const World& world, const std::vector<uint32_t>& uploaded_sources) {
    FlatScene f;
    // the meshes in flatten()'s order; every mesh's triangles in the order they were uploaded in
    std::map<const Mesh*, size_t> seen;
    size_t cursor = 0;
    for (const auto& inst : world.instances) {
        if (!inst->mesh || seen.count(inst->mesh.get())) continue;
        seen[inst->mesh.get()] = seen.size();
        const hiprz_mesh_desc d = mesh_desc(*inst->mesh);
        if (cursor + d.n_triangles > uploaded_sources.size()) return FlatScene{};
        const size_t base = f.tris.size();
        f.tris.resize(base + d.n_triangles), f.tri_attrs.resize(base + d.n_triangles);
        if (d.n_triangles &&
            hiprz_fill_triangles(&d, uploaded_sources.data() + cursor, d.n_triangles, f.tris.data() + base, f.tri_attrs.data() + base) != HIPRZ_OK)
            return FlatScene{};
        cursor += d.n_triangles;
    }
    if (cursor != uploaded_sources.size()) return FlatScene{};
    for (const auto& inst : world.instances) {
        hiprz_instance r{};
        const Xform own = own_transform(inst->position, inst->rotation, inst->scale), grouped = in_group(*inst);
        store(r, grouped);
        if (inst->mesh) hiprz_instance_bounds(inst->mesh->vertices.data(), uint32_t(inst->mesh->vertices.size() / 3), &r);
        if (world.group_transforms == World::GroupTransforms::Cpu) store(r, own);
        f.instances.push_back(r);
    }
    return f;
}

FlatScene flattenShading(const World& world) {
    // the map indices must be those of the uploaded scene: textures are numbered in first-use order over world material, default
    // material and the world's materials — exactly what flatten() does, so the numbering is replayed without copying any texels
    FlatScene f;
    std::map<const TextureBuffer*, int32_t> tex_index;
    auto tex_id = [&](const std::shared_ptr<TextureBuffer>& t) -> int32_t {
        if (!t) return -1;
        auto it = tex_index.find(t.get());
        if (it != tex_index.end()) return it->second;
        const int32_t id = int32_t(tex_index.size());
        f.maps.push_back(t.get());
        return tex_index[t.get()] = id;
    };
    auto add_material = [&](const Material& m) {
        hiprz_material r{};
        r.color[0] = m.color.red, r.color[1] = m.color.green, r.color[2] = m.color.blue, r.color[3] = m.color.alpha;
        r.metalness = m.metalness(), r.roughness = m.roughness(), r.emission = m.emission(), r.ior = m.ior(), r.scattering = m.scattering();
        r.texture = tex_id(m.texture), r.normal_map = tex_id(m.normal_map), r.metalness_map = tex_id(m.metalness_map);
        r.roughness_map = tex_id(m.roughness_map), r.emission_map = tex_id(m.emission_map);
        f.materials.push_back(r);
    };
    add_material(world.material);
    add_material(world.default_material);
    for (const auto& m : world.materials) add_material(*m);
    flatten_lights(world, f);
    return f;
}

hiprz_camera cameraRecord(const Camera& cam) {
    const float eps = std::numeric_limits<float>::epsilon();
    hiprz_camera c{};
    put3(c.position, cam.position);
    float rot[3];
    put3(rot, cam.rotation);
    hiprz_axes_look_at(rot, c.x_axis, c.y_axis, c.z_axis);  // camera.cpp:95-100
    c.width = std::max(cam.width, 1u), c.height = std::max(cam.height, 1u);
    c.fov = std::min(std::max(cam.fov, eps), 3.14159265358979f - eps);  // camera.cpp:100-110
    c.tan_half_fov = std::tan(c.fov * 0.5f);
    c.aspect_ratio = float(c.width) / float(c.height);
    c.near_far[0] = std::max(cam.near_plane, eps);
    c.near_far[1] = std::max(cam.far_plane, c.near_far[0] + eps);
    c.focal_distance = std::max(cam.focal_distance, eps);
    c.aperture = std::max(cam.aperture, eps);
    c.exposure_time = std::max(cam.exposure_time, eps);
    return c;
}

// ---- Engine ----
Engine::Engine(int device, int streams) : m_device(device) {
    int rc;
    if (streams > 1) {
        const std::vector<int> ids(size_t(streams), device);
        rc = hiprz_create_multi(&m_ctx, ids.data(), streams);
    } else {
        rc = hiprz_create(&m_ctx, device);
        m_streams_pending = streams == 0;
    }
    if (rc != HIPRZ_OK) throw Exception(rc, std::string("HIPGPU backend unavailable: ") + hiprz_last_error(nullptr));
    check(hiprz_set_tree(m_ctx, m_tree));
}
Engine::Engine(const std::vector<int>& devices) {
    const int rc = hiprz_create_multi(&m_ctx, devices.data(), int(devices.size()));
    if (rc != HIPRZ_OK) throw Exception(rc, std::string("HIPGPU backend unavailable: ") + hiprz_last_error(nullptr));
    check(hiprz_set_tree(m_ctx, m_tree));
}
Engine::~Engine() { hiprz_destroy(m_ctx); }

void Engine::check(int rc) {
    if (rc != HIPRZ_OK) throw Exception(rc, hiprz_last_error(m_ctx));
}
void Engine::mode(uint32_t compat_flags) {
    std::lock_guard<std::mutex> lock(m_mutex);
    check(hiprz_set_mode(m_ctx, compat_flags));  // reprojection works over several streams / devices: the context assembles the whole previous frame
    m_mode = compat_flags;
}
void Engine::shardMode(ShardMode mode) {
    std::lock_guard<std::mutex> lock(m_mutex);
    m_streams_pending = false;  // the context stays as it is: the mode is about ITS parts
    check(hiprz_set_shard_mode(m_ctx, uint32_t(mode)));
}
void Engine::tree(uint32_t tree) {
    std::lock_guard<std::mutex> lock(m_mutex);
    check(hiprz_set_tree(m_ctx, tree));
    m_tree = tree;
    m_last_world = nullptr;  // takes effect at the next scene upload: force one
}

void Engine::readback(Camera& camera, const World& world) {
    const size_t n = size_t(camera.width) * camera.height;
    camera.image_buffer.resize(n * 4);
    camera.depth_buffer.resize(n);
    check(hiprz_read_rgba8(m_ctx, camera.image_buffer.data(), n * 4));
    check(hiprz_read_depth(m_ctx, camera.depth_buffer.data(), n * sizeof(float)));
    check(hiprz_ray_count(m_ctx, &camera.ray_count));
    // Kernel::rayCast after every frame (cpu_engine_renderer.cpp:176; cuda_engine_core.cu:164-181): the instance and the instance's
    // material slot the ray through the camera's ray-cast pixel meets at the first-hit depth
    hiprz_raycast hit{};
    check(hiprz_ray_cast(m_ctx, camera.ray_cast_pixel[0], camera.ray_cast_pixel[1], &hit));
    camera.raycasted_instance.reset(), camera.raycasted_material.reset();
    if (hit.instance >= 0 && size_t(hit.instance) < world.instances.size()) {
        camera.raycasted_instance = world.instances[size_t(hit.instance)];
        if (hit.material_slot >= 0 && uint32_t(hit.material_slot) < Instance::materialCapacity())
            camera.raycasted_material = camera.raycasted_instance->materials[size_t(hit.material_slot)];
    }
}

std::vector<Camera*> Engine::enabledCameras(World& world) const {  // cpu_engine_renderer.cpp:97-100: every enabled camera
    std::vector<Camera*> out;
    if (world.camera.enabled) out.push_back(&world.camera);
    for (auto& c : world.cameras)
        if (c && c->enabled) out.push_back(c.get());
    return out;
}

void Engine::renderWorld(World& world, const RenderConfig& cfg, bool /*block*/, bool sync) {
    std::lock_guard<std::mutex> lock(m_mutex);
    if (m_deferred) {  // error of the previous, already returned, asynchronous frame
        Exception e = *m_deferred;
        m_deferred.reset();
        throw e;
    }
    if (m_streams_pending) {  // the first world: as many streams on the GPU as suit it
        m_streams_pending = false;
        const int streams = defaultStreams(world);
        if (streams > 1) {
            hiprz_ctx* several = nullptr;
            const std::vector<int> ids(size_t(streams), m_device);
            if (hiprz_create_multi(&several, ids.data(), streams) == HIPRZ_OK) {
                hiprz_destroy(m_ctx);
                m_ctx = several;
                check(hiprz_set_mode(m_ctx, m_mode));
                check(hiprz_set_tree(m_ctx, m_tree));
            }
        }
    }
    const std::vector<Camera*> cameras = enabledCameras(world);
    if (m_pending_readback) {  // pipelined frames of the previous non-sync call
        m_pending_readback = false;
        for (size_t k = 0; k < m_camera_slots.size(); ++k)
            for (Camera* cam : cameras)
                if (cam == m_camera_slots[k]) {
                    check(hiprz_select_camera(m_ctx, uint32_t(k)));
                    readback(*cam, world);
                }
    }
    // re-mirror what changed; any change restarts accumulation (cpu_engine_renderer.cpp:108-112)
    bool moved_in_place = false;
    if (world.isMoved() && !world.isModified() && m_last_world == &world && world.instances.size() == m_uploaded_instances) {
        // an animation frame: where the device built the trees it refits them and rebuilds the world tree; no tree is built on the host
        uint32_t tree = HIPRZ_TREE_REFERENCE;
        if (hiprz_tree(m_ctx, &tree) == HIPRZ_OK && (tree == HIPRZ_TREE_DEVICE || tree == HIPRZ_TREE_DEVICE_SAH)) {
            const FlatScene motion = flattenMotion(world, m_uploaded_sources);
            if (motion.instances.size() == m_uploaded_instances && motion.tris.size() == m_uploaded_sources.size()) {
                if (!motion.tris.empty()) check(hiprz_update_triangles(m_ctx, 0u, uint32_t(motion.tris.size()), motion.tris.data(), motion.tri_attrs.data()));
                if (!motion.instances.empty()) check(hiprz_update_instances(m_ctx, motion.instances.data(), uint32_t(motion.instances.size())));
                // a refitted tree keeps the topology it was built with: every kRebuildEvery-th moved frame the device builds the trees again
                // over the vertices it holds (config D twisted by three radians per unit: 28 ms per step refitted, 12 ms rebuilt)
                if (++m_moved_frames % kRebuildEvery == 0u) check(hiprz_rebuild_trees(m_ctx, tree));
                moved_in_place = true;
            }
        }
    }
    if (world.isMoved() && !moved_in_place) world.makeModified();  // no device trees (or another topology): an ordinary modification
    world.makeUnmoved();
    if (world.isModified() || m_last_world != &world) {
        const FlatScene flat = flatten(world);
        const hiprz_scene view = flat.view();
        check(hiprz_upload_scene(m_ctx, &view));
        world.makeUnmodified(), world.makeShadingUnmodified();
        m_last_world = &world;
        m_uploaded_maps = flat.maps;
        m_uploaded_sources.resize(flat.tris.size());
        for (size_t k = 0; k < flat.tris.size(); ++k) m_uploaded_sources[k] = flat.tris[k].source_index;
        m_uploaded_instances = flat.instances.size();
        m_moved_frames = 0;
    } else if (world.isShadingModified()) {  // materials / lights only: replaced in place, no tree is touched
        const FlatScene shading = flattenShading(world);
        // flattenShading numbers the maps in first-use order; those indices mean something only if they name the SAME map objects in
        // the same order as the uploaded scene's (a material re-pointed at another uploaded map changes the first-use order, a new
        // map has no texels on the device at all): otherwise the whole scene goes up again, as the adapter does (WorldAdapter::refresh)
        if (shading.maps == m_uploaded_maps) {
            check(hiprz_update_shading(m_ctx, shading.materials.data(), uint32_t(shading.materials.size()), shading.spot_lights.data(),
                                       uint32_t(shading.spot_lights.size()), shading.direct_lights.data(), uint32_t(shading.direct_lights.size())));
        } else {
            const FlatScene flat = flatten(world);
            const hiprz_scene view = flat.view();
            check(hiprz_upload_scene(m_ctx, &view));
            m_uploaded_maps = flat.maps;
            m_uploaded_sources.resize(flat.tris.size());
            for (size_t k = 0; k < flat.tris.size(); ++k) m_uploaded_sources[k] = flat.tris[k].source_index;
            m_uploaded_instances = flat.instances.size();
        }
        world.makeShadingUnmodified();
    }
    hiprz_config c{};
    c.max_depth = cfg.tracing.max_depth, c.rpp = cfg.tracing.rpp;
    c.spot_samples = std::max<uint32_t>(cfg.light_sampling.spot_light, 1u);      // cuda_kernel_data.cu:23-31
    c.direct_samples = std::max<uint32_t>(cfg.light_sampling.direct_light, 1u);
    c.seed = cfg.seed;
    check(hiprz_set_config(m_ctx, &c));
    // one frame state per enabled camera, in the order the reference iterates them
    bool slots_changed = m_camera_slots.size() != cameras.size();
    for (size_t k = 0; !slots_changed && k < cameras.size(); ++k) slots_changed = m_camera_slots[k] != cameras[k];
    if (slots_changed) {
        check(hiprz_set_camera_count(m_ctx, uint32_t(std::max<size_t>(cameras.size(), 1))));
        m_camera_slots.assign(cameras.begin(), cameras.end());
        m_camera_records.assign(cameras.size(), hiprz_camera{});
        for (Camera* cam : cameras) cam->makeModified();
    }
    for (size_t k = 0; k < cameras.size(); ++k) {
        Camera& cam = *cameras[k];
        check(hiprz_select_camera(m_ctx, uint32_t(k)));
        if (cam.isModified()) {
            const hiprz_camera rec = cameraRecord(cam);
            // an upload restarts accumulation (cpu_engine_renderer.cpp:108-112); a camera that only moved its ray-cast pixel
            // (Camera::rayCastPixel: MakeModified, not RequestUpdate) keeps accumulating in both reference engines
            if (slots_changed || std::memcmp(&rec, &m_camera_records[k], sizeof rec) != 0) {
                check(hiprz_upload_camera(m_ctx, &rec));
                m_camera_records[k] = rec;
            }
            check(hiprz_set_temporal_blend(m_ctx, cam.temporal_blend));
            cam.makeUnmodified();
        }
        check(hiprz_render(m_ctx, std::max(cfg.tracing.rpp, 1u)));
        check(hiprz_tonemap(m_ctx));
        if (sync) readback(cam, world);
    }
    // not sync: nothing has been waited for — the buffers are filled by the next call, and a device fault would surface at that
    // call's first hip* return
    if (!sync) m_pending_readback = true;
}

std::string Engine::timingsString() {
    std::lock_guard<std::mutex> lock(m_mutex);
    char buf[4096];
    if (hiprz_timings(m_ctx, buf, sizeof buf) != HIPRZ_OK) return {};
    return buf;
}

}  // namespace RayZath::Hip
