// adapter_check.cpp — runs rayzath_adapter.hpp's WorldAdapter over the test double (adapter_double.hpp) of a scene file and compares
// with Hip::flatten() of the same scene; then walks through the dirty-flag cases.  Built and run by tests/test_adapter.py.
//   usage: adapter_check scene.json
#include <cstdio>
#include <cstring>
#include <string>

#include "adapter_double.hpp"
#include "rayzath_adapter.hpp"
#include "scene_io.hpp"

using namespace RayZath::Hip;

template <class T>
static bool same(const char* what, const std::vector<T>& a, const std::vector<T>& b) {
    const bool ok = a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(T)) == 0);
    std::printf("%-16s %s (%zu records)\n", what, ok ? "equal" : "DIFFERENT", a.size());
    return ok;
}
static bool same_scene(const FlatScene& a, const FlatScene& b) {
    bool ok = same("nodes", a.nodes, b.nodes);
    ok &= same("tlas_order", a.tlas_order, b.tlas_order), ok &= same("tris", a.tris, b.tris), ok &= same("tri_attrs", a.tri_attrs, b.tri_attrs);
    ok &= same("instances", a.instances, b.instances), ok &= same("inst_materials", a.inst_materials, b.inst_materials);
    ok &= same("materials", a.materials, b.materials), ok &= same("textures", a.textures, b.textures), ok &= same("texels", a.texels, b.texels);
    ok &= same("spot_lights", a.spot_lights, b.spot_lights), ok &= same("direct_lights", a.direct_lights, b.direct_lights);
    return ok;
}


using OT = Double::ObjectType;
static void load_scene(const char* path, World& w) {
    IO::LoadLog l;
    IO::loadScene(path, w, l);
}
// vertices of the first real mesh and the first instance's transformation move; nothing else changes
static void move_world(World& w) {
    for (auto& inst : w.instances)
        if (inst->mesh && inst->mesh->tri_materials.size() >= 2) {
            auto& v = inst->mesh->vertices;
            for (size_t k = 0; k + 2 < v.size(); k += 3) v[k] = v[k] * 1.25f + 0.1f * v[k + 2], v[k + 1] = v[k + 1] * 0.9f - 0.05f * v[k];
            break;
        }
    w.instances[0]->position.x += 0.3f, w.instances[0]->rotation.y += 0.4f;
}
// the double follows the moved twin IN PLACE (the adapter knows maps and materials by identity): new vertices, new transformations and
// boxes; its trees stay as they were, as a refit leaves them
static void follow_twin(Double::World& w, const World& t, const FlatScene& f) {
    std::map<const Mesh*, uint32_t> slot;
    auto& meshes = w.container<OT::Mesh>();
    auto& instances = w.container<OT::Instance>();
    for (size_t i = 0; i < t.instances.size(); ++i) {
        const hiprz_instance& r = f.instances[i];
        Double::Transformation x;
        x.m_position = Double::detail::v3(r.position), x.m_scale = Double::detail::v3(r.scale);
        x.m_coord_system = Double::CoordSystem{Double::detail::v3(r.x_axis), Double::detail::v3(r.y_axis), Double::detail::v3(r.z_axis)};
        instances[uint32_t(i)]->m_transformation = x, instances[uint32_t(i)]->m_transformation_in_group = x;
        instances[uint32_t(i)]->m_bb = Double::BoundingBox{Double::detail::v3(r.bb_min), Double::detail::v3(r.bb_max)};
        instances[uint32_t(i)]->stateRegister().MakeModified();
        const auto& mesh = t.instances[i]->mesh;
        if (!mesh || slot.count(mesh.get())) continue;
        const uint32_t k = uint32_t(slot.size());
        slot[mesh.get()] = k;
        auto& items = meshes[k]->m_vertices.items;
        for (size_t q = 0; q < items.size(); ++q) items[q] = Double::vec3f{mesh->vertices[3 * q], mesh->vertices[3 * q + 1], mesh->vertices[3 * q + 2]};
        meshes[k]->stateRegister().MakeModified();
    }
}

// GPU: WorldRenderer over the double == Hip::Engine over the twin, frame for frame (two cameras, sync and pipelined calls, a
// shading-only change in between)
static bool render_check(const char* scene_path) {
    World twin;
    IO::LoadLog log;
    IO::loadScene(scene_path, twin, log);
    auto second = std::make_shared<Camera>(twin.camera);
    second->position.x += 0.8f, second->width = 80, second->height = 56;
    twin.cameras.push_back(second);
    const FlatScene flat = flatten(twin);
    auto world = Double::from_twin(twin, flat, flat);
    std::vector<Double::Handle<Double::Camera>> cams = {Double::add_camera(*world, twin.camera), Double::add_camera(*world, *second)};
    Engine engine(0);
    hiprz_ctx* ctx = nullptr;
    if (hiprz_create(&ctx, 0) != HIPRZ_OK) return std::printf("no device\n"), false;
    WorldRenderer<Double::Api> renderer(ctx);
    RenderConfig cfg;
    cfg.tracing.max_depth = 5, cfg.tracing.rpp = 3, cfg.light_sampling.spot_light = 2;
    Double::RenderConfig dcfg;
    dcfg.m_tracing.m_max_depth = 5, dcfg.m_tracing.m_rpp = 3, dcfg.m_light_sampling.m_spot = 2;
    bool ok = true;
    // picking: both sides look through the same pixels (camera.hpp:90); what they meet must be the same object of their world
    twin.camera.rayCastPixel(twin.camera.width / 2, twin.camera.height / 2), second->rayCastPixel(20, 30);
    cams[0]->rayCastPixel({twin.camera.width / 2, twin.camera.height / 2}), cams[1]->rayCastPixel({20, 30});
    auto& dinstances = world->container<Double::ObjectType::Instance>();
    auto& dmaterials = world->container<Double::ObjectType::Material>();
    auto ray_casts = [&](const char* what) {
        Camera* tw[2] = {&twin.camera, second.get()};
        for (int k = 0; k < 2; ++k) {
            int ti = -1, di = -1, tm = -1, dm = -1;
            for (size_t i = 0; i < twin.instances.size(); ++i)
                if (twin.instances[i] == tw[k]->raycasted_instance) ti = int(i);
            for (uint32_t i = 0; i < dinstances.count(); ++i)
                if (cams[k]->m_raycasted_instance && dinstances[i].p == cams[k]->m_raycasted_instance.p) di = int(i);
            for (size_t i = 0; i < twin.materials.size(); ++i)
                if (tw[k]->raycasted_material && twin.materials[i] == tw[k]->raycasted_material) tm = int(i);
            for (uint32_t i = 0; i < dmaterials.count(); ++i)
                if (cams[k]->m_raycasted_material && dmaterials[i].p == cams[k]->m_raycasted_material.p) dm = int(i);
            const bool eq = ti == di && tm == dm;
            std::printf("%-28s camera %d ray cast %s (instance %d, material %d)\n", what, k, eq ? "equal" : "DIFFERENT", di, dm);
            ok &= eq;
        }
    };
    auto compare = [&](const char* what) {
        Camera* tw[2] = {&twin.camera, second.get()};
        for (int k = 0; k < 2; ++k) {
            const size_t n = size_t(tw[k]->width) * tw[k]->height;
            bool eq = tw[k]->image_buffer.size() == 4 * n && tw[k]->ray_count == cams[k]->rayCount();
            for (size_t i = 0; eq && i < n; ++i) {
                const Double::Color& c = cams[k]->m_image.data[i];
                const uint8_t* p = &tw[k]->image_buffer[4 * i];
                eq = c.red == p[0] && c.green == p[1] && c.blue == p[2] && c.alpha == p[3] &&
                     std::memcmp(&cams[k]->m_depth.data[i], &tw[k]->depth_buffer[i], 4) == 0;
            }
            std::printf("%-28s camera %d %s (%llu rays)\n", what, k, eq ? "equal" : "DIFFERENT", (unsigned long long)cams[k]->rayCount());
            ok &= eq;
        }
    };
    engine.renderWorld(twin, cfg), renderer.renderWorld(*world, dcfg);
    compare("first frame");
    ray_casts("first frame");
    ok &= bool(cams[0]->m_raycasted_instance);  // the centre of the frame looks at something
    {   // the ray-cast pixel moves: another object, the same accumulation (ray counts go on)
        const uint64_t before = cams[0]->rayCount();
        twin.camera.rayCastPixel(3, twin.camera.height - 3), cams[0]->rayCastPixel({3, twin.camera.height - 3});
        engine.renderWorld(twin, cfg), renderer.renderWorld(*world, dcfg);
        compare("after moving the ray cast");
        ray_casts("after moving the ray cast");
        const bool went_on = cams[0]->rayCount() == before + 3ull * twin.camera.width * twin.camera.height;
        std::printf("accumulation went on         %s\n", went_on ? "yes" : "DIFFERENT");
        ok &= went_on;
    }
    engine.renderWorld(twin, cfg, true, false), renderer.renderWorld(*world, dcfg, true, false);  // pipelined: buffers filled by the next call
    engine.renderWorld(twin, cfg), renderer.renderWorld(*world, dcfg);
    compare("after a pipelined frame");
    twin.materials[0]->color.green = 40, twin.makeShadingModified();
    auto& materials = world->container<Double::ObjectType::Material>();
    materials[0]->m_color.green = 40, materials[0]->stateRegister().MakeModified();
    engine.renderWorld(twin, cfg), renderer.renderWorld(*world, dcfg);
    compare("after a material change");
    uint32_t captures = 0;
    hiprz_graph_captures(ctx, &captures);
    std::printf("graph captures %u\n", captures);
    hiprz_destroy(ctx);
    {   // a moved world over device-built trees: WorldRenderer refits on the device; the frame is a fresh engine's frame of the moved twin
        World t0, t1;
        load_scene(scene_path, t0), load_scene(scene_path, t1);
        move_world(t1);
        const FlatScene f0 = flatten(t0), f1 = flatten(t1);
        auto w = Double::from_twin(t0, f0, f0);
        auto cam = Double::add_camera(*w, t0.camera);
        hiprz_ctx* c2 = nullptr;
        if (hiprz_create(&c2, 0) != HIPRZ_OK || hiprz_set_tree(c2, HIPRZ_TREE_DEVICE_SAH) != HIPRZ_OK) return std::printf("no device\n"), false;
        WorldRenderer<Double::Api> r2(c2);
        r2.renderWorld(*w, dcfg);
        follow_twin(*w, t1, f1);
        r2.renderWorld(*w, dcfg);
        char timings[4096] = {0};
        hiprz_timings(c2, timings, sizeof timings);
        const bool refitted = std::strstr(timings, "refit mesh trees (device)") != nullptr;
        Engine fresh(0, 1);
        fresh.renderWorld(t1, cfg);
        const size_t n = size_t(t1.camera.width) * t1.camera.height;
        bool eq = t1.camera.ray_count == cam->rayCount();
        for (size_t i = 0; eq && i < n; ++i) {
            const Double::Color& c = cam->m_image.data[i];
            const uint8_t* p = &t1.camera.image_buffer[4 * i];
            eq = c.red == p[0] && c.green == p[1] && c.blue == p[2] && c.alpha == p[3] && std::memcmp(&cam->m_depth.data[i], &t1.camera.depth_buffer[i], 4) == 0;
        }
        std::printf("%-28s %s, %s\n", "moved world, device trees", eq ? "frame equal" : "frame DIFFERENT", refitted ? "refitted on the device" : "NOT REFITTED");
        if (!eq) {  // where: colours, depths, ray counts; and the twin's own makeMoved() path on the same scene
            size_t colours = 0, depths = 0;
            for (size_t i = 0; i < n; ++i) {
                const Double::Color& c = cam->m_image.data[i];
                const uint8_t* p = &t1.camera.image_buffer[4 * i];
                colours += !(c.red == p[0] && c.green == p[1] && c.blue == p[2] && c.alpha == p[3]);
                depths += std::memcmp(&cam->m_depth.data[i], &t1.camera.depth_buffer[i], 4) != 0;
            }
            std::printf("  differing colours %zu, depths %zu of %zu; rays %llu vs %llu\n", colours, depths, n, (unsigned long long)cam->rayCount(), (unsigned long long)t1.camera.ray_count);
            World t2;
            load_scene(scene_path, t2);
            Engine twin_engine(0, 1);
            twin_engine.tree(HIPRZ_TREE_DEVICE_SAH);
            twin_engine.renderWorld(t2, cfg);
            move_world(t2), t2.makeMoved();
            twin_engine.renderWorld(t2, cfg);
            size_t twin_colours = 0;
            for (size_t i = 0; i < 4 * n; ++i) twin_colours += t2.camera.image_buffer[i] != t1.camera.image_buffer[i];
            std::printf("  twin makeMoved vs fresh: %zu differing bytes\n", twin_colours);
        }
        ok &= eq && refitted;
        hiprz_destroy(c2);
    }
    return ok;
}

int main(int argc, char** argv) {
    if (argc < 2) return std::fprintf(stderr, "usage: %s scene.json\n", argv[0]), 2;
    World twin;
    IO::LoadLog log;
    IO::loadScene(argv[1], twin, log);
    twin.group_transforms = World::GroupTransforms::Cuda;
    const FlatScene flat_cuda = flatten(twin);
    twin.group_transforms = World::GroupTransforms::Cpu;
    const FlatScene flat = flatten(twin);
    auto world = Double::from_twin(twin, flat, flat_cuda);
    using Adapter = WorldAdapter<Double::Api>;
    bool ok = true;

    Adapter adapter;
    std::printf("# first refresh\n");
    ok &= adapter.refresh(*world) == Adapter::Change::Scene;
    ok &= same_scene(adapter.scene(), flat);
    ok &= !world->stateRegister().IsModified() && !world->container<OT::Instance>().stateRegister().IsModified();

    std::printf("# nothing modified\n");
    ok &= adapter.refresh(*world) == Adapter::Change::None;

    std::printf("# a material's colour and a light's emission\n");
    auto& materials = world->container<OT::Material>();
    materials[0]->m_color.red = uint8_t(materials[0]->m_color.red ^ 0x55);
    materials[0]->stateRegister().MakeModified();
    twin.materials[0]->color.red = uint8_t(twin.materials[0]->color.red ^ 0x55);
    if (world->container<OT::SpotLight>().count()) {
        world->container<OT::SpotLight>()[0]->m_emission = 7.5f, world->container<OT::SpotLight>()[0]->stateRegister().MakeModified();
        twin.spot_lights[0]->emission = 7.5f;
    }
    ok &= world->stateRegister().IsModified();  // the flag travelled up (updatable.cpp:23-27)
    ok &= adapter.refresh(*world) == Adapter::Change::Shading;
    const FlatScene shading = flattenShading(twin);
    ok &= same("materials", adapter.scene().materials, shading.materials) & same("spot_lights", adapter.scene().spot_lights, shading.spot_lights) &
          same("direct_lights", adapter.scene().direct_lights, shading.direct_lights);
    ok &= same_scene(adapter.scene(), flatten(twin));

    std::printf("# a material is given a map the uploaded scene does not hold: full refresh\n");
    {
        auto map = world->container<OT::RoughnessMap>().create();
        map->m_bitmap.w = 2, map->m_bitmap.h = 2, map->m_bitmap.data = {10, 20, 30, 40};
        map->m_scale = Double::vec2f{1, 1};
        std::get<3>(materials[0]->m_maps) = map;
        materials[0]->stateRegister().MakeModified();
        world->container<OT::RoughnessMap>().stateRegister().MakeUnmodified();  // even if only the material reports it
        auto t = std::make_shared<TextureBuffer>();
        t->kind = HIPRZ_TEX_R8, t->width = 2, t->height = 2, t->bitmap = {10, 20, 30, 40};
        twin.materials[0]->roughness_map = t;
    }
    ok &= adapter.refresh(*world) == Adapter::Change::Scene;
    // the twin numbers its maps in first-use order over the materials as well, so the snapshots still agree
    ok &= same_scene(adapter.scene(), flatten(twin));

    std::printf("# an instance reports a change: full refresh; the CUDA engine's group transformations\n");
    world->container<OT::Instance>()[0]->stateRegister().MakeModified();
    adapter.group_transforms = World::GroupTransforms::Cuda;
    twin.group_transforms = World::GroupTransforms::Cuda;
    ok &= adapter.refresh(*world) == Adapter::Change::Scene;
    ok &= same_scene(adapter.scene(), flatten(twin));

    std::printf("# vertices and transformations moved while the context holds device-built trees: records in the UPLOADED order\n");
    {
        World before, after;
        load_scene(argv[1], before), load_scene(argv[1], after);
        move_world(after);
        const FlatScene f0 = flatten(before), f1 = flatten(after);
        auto w0 = Double::from_twin(before, f0, f0);
        Adapter moving, rebuilding;
        ok &= moving.refresh(*w0, /*device_trees=*/true) == Adapter::Change::Scene;
        follow_twin(*w0, after, f1);
        const bool moved = moving.refresh(*w0, /*device_trees=*/true) == Adapter::Change::Moved;
        std::printf("%-16s %s\n", "change", moved ? "Moved" : "NOT Moved");
        ok &= moved;
        std::vector<uint32_t> sources(f0.tris.size());
        for (size_t k = 0; k < sources.size(); ++k) sources[k] = f0.tris[k].source_index;
        const FlatScene motion = flattenMotion(after, sources);
        ok &= same("moved tris", moving.scene().tris, motion.tris) & same("moved tri_attrs", moving.scene().tri_attrs, motion.tri_attrs);
        bool inst_eq = moving.scene().instances.size() == motion.instances.size();
        for (size_t i = 0; inst_eq && i < motion.instances.size(); ++i) {  // transformation and box (flattenMotion leaves the tables empty)
            const hiprz_instance &a = moving.scene().instances[i], &b = motion.instances[i];
            inst_eq = !std::memcmp(a.position, b.position, 12) && !std::memcmp(a.scale, b.scale, 12) && !std::memcmp(a.x_axis, b.x_axis, 12) &&
                      !std::memcmp(a.y_axis, b.y_axis, 12) && !std::memcmp(a.z_axis, b.z_axis, 12) && !std::memcmp(a.bb_min, b.bb_min, 12) &&
                      !std::memcmp(a.bb_max, b.bb_max, 12) && a.blas_root == f0.instances[i].blas_root && a.material_base == f0.instances[i].material_base;
        }
        std::printf("%-16s %s\n", "moved instances", inst_eq ? "equal" : "DIFFERENT");
        ok &= inst_eq;
        ok &= std::memcmp(moving.scene().tris.data(), f0.tris.data(), f0.tris.size() * sizeof(hiprz_tri)) != 0;  // (something did move)
        // the same change while the context holds host-built trees: a full refresh
        auto w0b = Double::from_twin(before, f0, f0);
        ok &= rebuilding.refresh(*w0b, false) == Adapter::Change::Scene;
        follow_twin(*w0b, after, f1);
        ok &= rebuilding.refresh(*w0b, false) == Adapter::Change::Scene;
        ok &= same("tris after a full refresh", rebuilding.scene().tris, motion.tris);
    }

    std::printf("# camera record\n");
    {
        const hiprz_camera ref = cameraRecord(twin.camera);
        const hiprz_camera got = Adapter::cameraRecord(*Double::add_camera(*world, twin.camera));
        const bool eq = std::memcmp(&got, &ref, sizeof ref) == 0;
        std::printf("%-16s %s\n", "camera", eq ? "equal" : "DIFFERENT");
        ok &= eq;
    }
    if (argc > 2 && std::string(argv[2]) == "render") ok &= render_check(argv[1]);
    std::printf(ok ? "ADAPTER OK\n" : "ADAPTER FAILED\n");
    return ok ? 0 : 1;
}
