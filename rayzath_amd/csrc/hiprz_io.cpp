// hiprz_io.cpp — the C-ABI of include/hiprz_io.h over scene_io.hpp.
#include "hiprz_io.h"

#include <string>

#include <cstring>

#include "image_io.hpp"
#include "scene_io.hpp"

using namespace RayZath::Hip;

struct hiprz_scene_file {
    World world;
    FlatScene flat;
    hiprz_scene scene{};
    hiprz_camera camera{};
    IO::LoadLog log;
    std::string log_text;
};

namespace {
thread_local std::string g_error;
}

extern "C" {

int hiprz_scene_file_load(const char* path, hiprz_scene_file** out) {
    if (out) *out = nullptr;
    if (!path || !out) return g_error = "null argument", HIPRZ_ERR_INVALID;
    auto* f = new hiprz_scene_file();
    try {
        const std::string p = path;
        if (p.size() > 4 && p.compare(p.size() - 4, 4, ".obj") == 0) IO::loadObjInstances(p, f->world, f->log);
        else IO::loadScene(p, f->world, f->log);
        f->flat = flatten(f->world);
        f->scene = f->flat.view();
        f->camera = cameraRecord(f->world.camera);
        f->log_text = f->log.str();
    } catch (const std::exception& e) {
        g_error = e.what();
        delete f;
        return HIPRZ_ERR_INVALID;
    }
    *out = f;
    return HIPRZ_OK;
}
void hiprz_scene_file_free(hiprz_scene_file* f) { delete f; }
const hiprz_scene* hiprz_scene_file_scene(const hiprz_scene_file* f) { return f ? &f->scene : nullptr; }
const hiprz_camera* hiprz_scene_file_camera(const hiprz_scene_file* f) { return f ? &f->camera : nullptr; }
uint32_t hiprz_scene_file_camera_count(const hiprz_scene_file* f) {
    if (!f) return 0u;
    uint32_t n = f->world.camera.enabled ? 1u : 0u;
    for (const auto& c : f->world.cameras) n += c && c->enabled ? 1u : 0u;
    return n;
}
int hiprz_scene_file_camera_at(const hiprz_scene_file* f, uint32_t index, hiprz_camera* out) {
    if (!f || !out) return -1;
    uint32_t k = 0;
    if (f->world.camera.enabled && k++ == index) return *out = cameraRecord(f->world.camera), 0;
    for (const auto& c : f->world.cameras)
        if (c && c->enabled && k++ == index) return *out = cameraRecord(*c), 0;
    return -1;
}
const char* hiprz_scene_file_log(const hiprz_scene_file* f) { return f ? f->log_text.c_str() : ""; }
uint32_t hiprz_scene_file_error_count(const hiprz_scene_file* f) { return f ? uint32_t(f->log.errors.size()) : 0u; }
uint32_t hiprz_scene_file_warning_count(const hiprz_scene_file* f) { return f ? uint32_t(f->log.warnings.size()) : 0u; }
int hiprz_scene_file_save(const hiprz_scene_file* f, const char* path, int kind) {
    if (!f || !path) return g_error = "null argument", HIPRZ_ERR_INVALID;
    try {
        if (kind == 0) IO::saveScene(path, f->world);
        else if (kind == 1) IO::saveOBJ(path, f->world);
        else return g_error = "kind: 0 = .json, 1 = .obj + .mtl", HIPRZ_ERR_INVALID;
    } catch (const std::exception& e) {
        g_error = e.what();
        return HIPRZ_ERR_INVALID;
    }
    return HIPRZ_OK;
}
int hiprz_image_read(const char* path, uint32_t channels, uint32_t* width_out, uint32_t* height_out, uint32_t* channels_out, uint8_t* pixels,
                     size_t capacity) {
    if (!path || channels > 4) return g_error = "hiprz_image_read: bad arguments", HIPRZ_ERR_INVALID;
    IO::Image img;
    std::string why;
    if (!IO::readImage(path, img, why)) return g_error = why, HIPRZ_ERR_INVALID;
    const uint32_t c = channels ? channels : img.channels;
    if (width_out) *width_out = img.width;
    if (height_out) *height_out = img.height;
    if (channels_out) *channels_out = c;
    if (!pixels) return HIPRZ_OK;
    const size_t bytes = size_t(img.width) * img.height * c;
    if (capacity < bytes) return g_error = "hiprz_image_read: destination too small", HIPRZ_ERR_INVALID;
    if (c == img.channels) std::memcpy(pixels, img.data.data(), bytes);
    else {
        const std::vector<uint8_t> converted = IO::convertChannels(img, c);
        std::memcpy(pixels, converted.data(), bytes);
    }
    return HIPRZ_OK;
}
int hiprz_image_write_png(const char* path, const uint8_t* pixels, uint32_t width, uint32_t height, uint32_t channels) {
    std::string why;
    if (!path || !IO::writePNG(path, pixels, width, height, channels, why)) return g_error = path ? why : "null path", HIPRZ_ERR_INVALID;
    return HIPRZ_OK;
}
int hiprz_image_read_f32(const char* path, uint32_t* width_out, uint32_t* height_out, float* pixels, size_t capacity) {
    if (!path) return g_error = "hiprz_image_read_f32: null path", HIPRZ_ERR_INVALID;
    uint32_t w = 0, h = 0;
    std::vector<float> values;
    std::string why;
    if (!IO::readImageF32(path, w, h, values, why)) return g_error = why, HIPRZ_ERR_INVALID;
    if (width_out) *width_out = w;
    if (height_out) *height_out = h;
    if (!pixels) return HIPRZ_OK;
    if (capacity < values.size()) return g_error = "hiprz_image_read_f32: destination too small", HIPRZ_ERR_INVALID;
    std::memcpy(pixels, values.data(), values.size() * sizeof(float));
    return HIPRZ_OK;
}
int hiprz_image_write_hdr(const char* path, const float* pixels, uint32_t width, uint32_t height) {
    std::string why;
    if (!path || !IO::writeHDR(path, pixels, width, height, why)) return g_error = path ? why : "null path", HIPRZ_ERR_INVALID;
    return HIPRZ_OK;
}
const char* hiprz_io_last_error(void) { return g_error.c_str(); }
}
