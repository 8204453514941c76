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
    using OT = Double::ObjectType;
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
