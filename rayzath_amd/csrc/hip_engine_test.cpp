// hip_engine_test.cpp — drives the C++ host side (hip_engine.hpp) the way RayZath's facade drives a
// backend; tests/test_cpp_host.py compares what it writes with the Python host side on the same scene.
//   hip_engine_test flatten <out.bin>          pure host: dump the flattened snapshot
//   hip_engine_test render  <out.bin> <calls>  GPU: renderWorld <calls> times, dump the camera outputs
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "hip_engine.hpp"

using namespace RayZath::Hip;

static std::shared_ptr<Mesh> quad(const float v[4][3]) {
    auto m = std::make_shared<Mesh>();
    for (int i = 0; i < 4; ++i) m->createVertex(v[i][0], v[i][1], v[i][2]);
    m->createTexcrd(0, 0), m->createTexcrd(0, 1), m->createTexcrd(1, 1), m->createTexcrd(1, 0);
    m->createTriangle({0, 2, 1}, {0, 2, 1});
    m->createTriangle({0, 3, 2}, {0, 3, 2});
    return m;
}
static std::shared_ptr<Material> material(World& w, Color c, float metal, float rough, float emission, float ior) {
    auto m = std::make_shared<Material>();
    m->color = c, m->metalness(metal), m->roughness(rough), m->emission(emission), m->ior(ior);
    w.materials.push_back(m);
    return m;
}
static void instance(World& w, std::shared_ptr<Mesh> mesh, std::shared_ptr<Material> mat, vec3f pos, vec3f rot = {}, vec3f scale = {1, 1, 1}) {
    auto i = std::make_shared<Instance>();
    i->mesh = mesh, i->materials[0] = mat, i->position = pos, i->rotation = rot, i->scale = scale;
    w.instances.push_back(i);
}

// The same scene tests/test_cpp_host.py builds in Python: an open box room of explicit quads + two boxes.
static void build(World& w, uint32_t width, uint32_t height) {
    auto white = material(w, {230, 230, 230, 255}, 0, 1, 0, 1.5f), red = material(w, {200, 40, 40, 255}, 0, 1, 0, 1.5f);
    auto green = material(w, {40, 200, 40, 255}, 0, 1, 0, 1.5f), light = material(w, {255, 255, 255, 255}, 0, 1, 50, 1.5f);
    auto mirror = material(w, {0xF0, 0xF0, 0xF0, 0xFF}, 0.9f, 0, 0, 1.0f);
    const float fl[4][3] = {{-2, 0, -2}, {-2, 0, 2}, {2, 0, 2}, {2, 0, -2}};
    const float bk[4][3] = {{-2, -1, 2}, {-2, 3, 2}, {2, 3, 2}, {2, -1, 2}};
    const float lf[4][3] = {{-2, -1, -2}, {-2, 3, -2}, {-2, 3, 2}, {-2, -1, 2}};
    const float rt[4][3] = {{2, -1, -2}, {2, 3, -2}, {2, 3, 2}, {2, -1, 2}};
    const float lp[4][3] = {{-0.5f, 0, -0.5f}, {-0.5f, 0, 0.5f}, {0.5f, 0, 0.5f}, {0.5f, 0, -0.5f}};
    auto floor_mesh = quad(fl), cube = Mesh::generateCube();
    w.meshes = {floor_mesh, cube};
    instance(w, floor_mesh, white, {0, -1, 0});
    instance(w, floor_mesh, white, {0, 3, 0});
    instance(w, quad(bk), white, {0, 0, 0});
    instance(w, quad(lf), red, {0, 0, 0});
    instance(w, quad(rt), green, {0, 0, 0});
    instance(w, quad(lp), light, {0, 2.99f, 0});
    instance(w, cube, mirror, {-0.7f, 0.2f, 0.6f}, {0, 0.3f, 0}, {1.2f, 2.4f, 1.2f});
    instance(w, cube, white, {0.7f, -0.4f, -0.5f}, {0, -0.3f, 0}, {1.2f, 1.2f, 1.2f});
    w.camera.position = {0, 1, -3.5f};
    w.camera.width = width, w.camera.height = height;
    w.camera.focal_distance = 4.0f;
}

template <typename T>
static void dump(FILE* f, const char* name, const std::vector<T>& v) {
    char tag[16] = {0};
    std::strncpy(tag, name, 15);
    const uint64_t bytes = v.size() * sizeof(T);
    std::fwrite(tag, 1, 16, f);
    std::fwrite(&bytes, 8, 1, f);
    if (bytes) std::fwrite(v.data(), 1, bytes, f);
}

int main(int argc, char** argv) {
    if (argc < 3) return std::fprintf(stderr, "usage: %s flatten|render <out.bin> [calls]\n", argv[0]), 2;
    const std::string mode = argv[1];
    if (mode == "swapmaps") {  // a material re-pointed at another uploaded map: the shading-only path must not be taken blindly
        try {
            auto checker = [](uint8_t r, uint8_t g, uint8_t b) {
                auto t = std::make_shared<TextureBuffer>();
                t->kind = HIPRZ_TEX_RGBA8, t->width = 2, t->height = 2;
                t->bitmap = {r, g, b, 255, 20, 20, 20, 255, 20, 20, 20, 255, r, g, b, 255};
                t->scale[0] = t->scale[1] = 4.0f;
                return t;
            };
            auto textured = [&](World& w, bool swapped) {
                build(w, 96, 64);
                auto a = checker(250, 60, 60), b = checker(60, 60, 250);
                w.materials[0]->texture = swapped ? b : a;  // the white walls and the red wall carry a checker each
                w.materials[1]->texture = swapped ? a : b;
            };
            World world, fresh;
            textured(world, false), textured(fresh, true);
            RenderConfig cfg;
            cfg.tracing.max_depth = 4, cfg.tracing.rpp = 3;
            Engine engine(0, 1), reference(0, 1);
            engine.renderWorld(world, cfg);
            std::swap(world.materials[0]->texture, world.materials[1]->texture);  // both maps are uploaded already; their first-use order flips
            world.makeShadingModified();
            engine.renderWorld(world, cfg);
            reference.renderWorld(fresh, cfg);
            const bool eq = world.camera.image_buffer == fresh.camera.image_buffer && world.camera.ray_count == fresh.camera.ray_count;
            // a plain colour change afterwards still goes through the in-place path and agrees with a fresh upload as well
            world.materials[2]->color.blue = 200, world.makeShadingModified();
            fresh.materials[2]->color.blue = 200, fresh.makeModified();
            engine.renderWorld(world, cfg), reference.renderWorld(fresh, cfg);
            const bool eq2 = world.camera.image_buffer == fresh.camera.image_buffer;
            std::printf("swapped maps %s, colour change %s\n", eq ? "equal" : "DIFFERENT", eq2 ? "equal" : "DIFFERENT");
            return eq && eq2 ? 0 : 1;
        } catch (const Exception& e) {
            std::fprintf(stderr, "Hip::Exception %d: %s\n", e.code, e.what());
            return 1;
        }
    }
    if (mode == "moved") {  // World::makeMoved(): an animation frame goes through the device refit, not through a host-side tree build
        try {
            auto grid = [](int n) {  // a wavy sheet of 2 n^2 triangles
                auto m = std::make_shared<Mesh>();
                for (int j = 0; j <= n; ++j)
                    for (int i = 0; i <= n; ++i) m->createVertex(float(i) / n - 0.5f, 0.05f * std::sin(9.0f * i / n) * std::cos(7.0f * j / n), float(j) / n - 0.5f);
                for (int j = 0; j < n; ++j)
                    for (int i = 0; i < n; ++i) {
                        const uint32_t a = uint32_t(j * (n + 1) + i), b = a + 1, c = a + uint32_t(n + 1), d = c + 1;
                        m->createTriangle({a, c, b}), m->createTriangle({b, c, d});
                    }
                return m;
            };
            auto scene = [&](World& w, bool moved) {
                build(w, 96, 64);
                auto sheet = grid(40);
                w.meshes.push_back(sheet);
                instance(w, sheet, w.materials[4], {0.1f, 0.9f, 0.3f}, {0.5f, 0.2f, 0.1f}, {1.6f, 1.6f, 1.6f});
                if (moved) {
                    for (size_t k = 0; k < sheet->vertices.size(); k += 3) sheet->vertices[k + 1] = 0.08f * std::cos(11.0f * sheet->vertices[k]) + 0.3f * sheet->vertices[k + 2];
                    w.instances.back()->position = {-0.2f, 1.1f, 0.1f};
                    w.instances[6]->rotation = {0.1f, 0.9f, 0.0f};
                }
            };
            World world, fresh;
            scene(world, false), scene(fresh, true);
            RenderConfig cfg;
            cfg.tracing.max_depth = 4, cfg.tracing.rpp = 3;
            Engine engine(0, 1), reference(0, 1);
            engine.tree(HIPRZ_TREE_DEVICE_SAH);  // (the hosts' default would keep the snapshot's trees for a scene this small)
            engine.renderWorld(world, cfg);
            const std::vector<uint8_t> before = world.camera.image_buffer;
            auto& sheet = *world.meshes.back();
            for (size_t k = 0; k < sheet.vertices.size(); k += 3) sheet.vertices[k + 1] = 0.08f * std::cos(11.0f * sheet.vertices[k]) + 0.3f * sheet.vertices[k + 2];
            world.instances.back()->position = {-0.2f, 1.1f, 0.1f};
            world.instances[6]->rotation = {0.1f, 0.9f, 0.0f};
            world.makeMoved();
            engine.renderWorld(world, cfg);
            const bool refitted = engine.timingsString().find("refit mesh trees (device)") != std::string::npos;
            reference.renderWorld(fresh, cfg);
            const bool eq = world.camera.image_buffer == fresh.camera.image_buffer && world.camera.depth_buffer == fresh.camera.depth_buffer &&
                            world.camera.ray_count == fresh.camera.ray_count;
            // without device trees the same call is an ordinary modification
            World plain;
            scene(plain, false);
            Engine host_trees(0, 1);
            host_trees.tree(HIPRZ_TREE_REFERENCE);
            host_trees.renderWorld(plain, cfg);
            auto& sheet2 = *plain.meshes.back();
            for (size_t k = 0; k < sheet2.vertices.size(); k += 3) sheet2.vertices[k + 1] = 0.08f * std::cos(11.0f * sheet2.vertices[k]) + 0.3f * sheet2.vertices[k + 2];
            plain.instances.back()->position = {-0.2f, 1.1f, 0.1f};
            plain.instances[6]->rotation = {0.1f, 0.9f, 0.0f};
            plain.makeMoved();
            host_trees.renderWorld(plain, cfg);
            const bool eq2 = plain.camera.image_buffer == fresh.camera.image_buffer;
            std::printf("moved frame %s, %s, changed %s, on host trees %s\n", eq ? "equal" : "DIFFERENT", refitted ? "refitted on the device" : "NOT REFITTED",
                        before != world.camera.image_buffer ? "yes" : "NO", eq2 ? "equal" : "DIFFERENT");
            return eq && refitted && eq2 && before != world.camera.image_buffer ? 0 : 1;
        } catch (const Exception& e) {
            std::fprintf(stderr, "Hip::Exception %d: %s\n", e.code, e.what());
            return 1;
        }
    }
    if (mode == "sequence") {  // two engines over twin worlds through one random sequence of changes: the hosts' defaults against snapshot trees + full uploads
        try {
            const unsigned seed = argc > 3 ? unsigned(std::atoi(argv[3])) : 1u;
            uint32_t state = 0x9E3779B9u * (seed + 1u);
            auto rnd = [&]() { return state = state * 1664525u + 1013904223u, state >> 8; };
            auto grid = [](int n) {
                auto m = std::make_shared<Mesh>();
                for (int j = 0; j <= n; ++j)
                    for (int i = 0; i <= n; ++i) m->createVertex(float(i) / n - 0.5f, 0.05f * std::sin(9.0f * i / n) * std::cos(7.0f * j / n), float(j) / n - 0.5f);
                for (int j = 0; j < n; ++j)
                    for (int i = 0; i < n; ++i) {
                        const uint32_t a = uint32_t(j * (n + 1) + i), b = a + 1, c = a + uint32_t(n + 1), d = c + 1;
                        m->createTriangle({a, c, b}), m->createTriangle({b, c, d});
                    }
                return m;
            };
            auto scene = [&](World& w) {
                build(w, 96, 64);
                auto sheet = grid(40);
                w.meshes.push_back(sheet);
                instance(w, sheet, w.materials[4], {0.1f, 0.9f, 0.3f}, {0.5f, 0.2f, 0.1f}, {1.6f, 1.6f, 1.6f});
                auto lamp = std::make_shared<SpotLight>();
                lamp->position = {0.5f, 2.5f, -0.5f}, lamp->direction = {-0.2f, -1.0f, 0.3f}, lamp->emission = 40.0f;
                w.spot_lights.push_back(lamp);
            };
            World a_world, b_world;
            scene(a_world), scene(b_world);
            RenderConfig cfg;
            cfg.tracing.max_depth = 4, cfg.tracing.rpp = 2;
            Engine a(0, 1), b(0, 1);
            a.tree(HIPRZ_TREE_DEVICE_SAH);  // (the hosts' default would keep the snapshot's trees for a scene this small)
            b.tree(HIPRZ_TREE_REFERENCE);
            bool ok = true;
            int moved = 0, shaded = 0;
            for (int step = 0; step < 24 && ok; ++step) {
                const uint32_t op = step == 0 ? 0u : rnd() % 6u;
                const float amount = 0.02f + 0.01f * float(rnd() % 16u);
                const size_t which = rnd() % a_world.instances.size();
                for (World* w : {&a_world, &b_world}) {
                    const bool first = w == &a_world;
                    if (op == 1u) {  // vertices and a transformation move
                        auto& v = w->meshes.back()->vertices;
                        for (size_t k = 0; k < v.size(); k += 3) v[k + 1] = v[k + 1] * (1.0f + amount) + 0.02f * std::sin(13.0f * v[k]);
                        w->instances[which]->position.x += amount;
                        first ? w->makeMoved() : w->makeModified();
                    } else if (op == 2u) {  // only a transformation
                        w->instances[which]->rotation.y += amount;
                        first ? w->makeMoved() : w->makeModified();
                    } else if (op == 3u) {  // materials only: replaced in place on the first engine
                        w->materials[which % w->materials.size()]->color.green = uint8_t(w->materials[which % w->materials.size()]->color.green ^ 0x30);
                        first ? w->makeShadingModified() : w->makeModified();
                    } else if (op == 4u) {  // a moved frame AND a material change in one call
                        w->instances[which]->position.z -= amount;
                        w->spot_lights[0]->emission += 5.0f;
                        first ? (w->makeMoved(), w->makeShadingModified()) : w->makeModified();
                    } else if (op == 5u) {  // the camera
                        w->camera.position.x += amount;
                        w->camera.makeModified();
                    }
                }
                moved += op == 1u || op == 2u || op == 4u, shaded += op == 3u || op == 4u;
                a.renderWorld(a_world, cfg), b.renderWorld(b_world, cfg);
                ok = a_world.camera.image_buffer == b_world.camera.image_buffer && a_world.camera.depth_buffer == b_world.camera.depth_buffer &&
                     a_world.camera.ray_count == b_world.camera.ray_count;
                if (!ok) std::printf("step %d op %u: DIFFERENT\n", step, op);
            }
            const bool refitted = a.timingsString().find("refit mesh trees (device)") != std::string::npos;
            std::printf("sequence %s (%d moved frames, %d shading changes, %s)\n", ok ? "equal" : "DIFFERENT", moved, shaded, refitted ? "refitted on the device" : "never refitted");
            return ok && (moved == 0 || refitted) ? 0 : 1;
        } catch (const Exception& e) {
            std::fprintf(stderr, "Hip::Exception %d: %s\n", e.code, e.what());
            return 1;
        }
    }
    FILE* out = std::fopen(argv[2], "wb");
    if (!out) return std::perror("open"), 2;
    World world;
    build(world, 96, 64);
    try {
        if (mode == "flatten") {
            const FlatScene f = flatten(world);
            dump(out, "nodes", f.nodes), dump(out, "tlas_order", f.tlas_order), dump(out, "tris", f.tris);
            dump(out, "tri_attrs", f.tri_attrs), dump(out, "instances", f.instances), dump(out, "inst_materials", f.inst_materials);
            dump(out, "materials", f.materials);
            const hiprz_camera cam = cameraRecord(world.camera);
            dump(out, "camera", std::vector<hiprz_camera>{cam});
        } else {
            const int calls = argc > 3 ? std::atoi(argv[3]) : 2;
            Engine engine(0);
            RenderConfig cfg;
            cfg.tracing.max_depth = 4, cfg.tracing.rpp = 3;
            for (int i = 0; i < calls; ++i) engine.renderWorld(world, cfg, true, i % 2 == 0);  // alternate sync / pipelined
            engine.renderWorld(world, cfg, true, true);
            std::vector<float> accum(size_t(96) * 64 * 4);
            if (hiprz_read_accum(engine.context(), accum.data(), accum.size() * 4) != HIPRZ_OK) throw Exception(2, "read_accum");
            dump(out, "image", world.camera.image_buffer), dump(out, "depth", world.camera.depth_buffer), dump(out, "accum", accum);
            dump(out, "ray_count", std::vector<uint64_t>{world.camera.ray_count});
            // a broken world must surface as Hip::Exception, not as a device fault
            world.instances[0]->materials[0] = std::make_shared<Material>();  // not registered in the world
            world.makeModified();
            bool thrown = false;
            try {
                engine.renderWorld(world, cfg);
            } catch (const Exception& e) {
                thrown = e.code == HIPRZ_ERR_INVALID;
            }
            dump(out, "threw", std::vector<uint8_t>{uint8_t(thrown)});
            std::printf("%s", engine.timingsString().c_str());
        }
    } catch (const Exception& e) {
        std::fprintf(stderr, "Hip::Exception %d: %s\n", e.code, e.what());
        return 1;
    }
    std::fclose(out);
    return 0;
}
