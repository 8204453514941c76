/* hiprz_io.h — C-ABI of the host library (libhiprz_host.so): scene files for the HIPGPU backend.
 *
 * Replaces, for this backend's stand-alone host side, what RayZath's core library does in
 *   Loader::loadScene / JsonLoader::load     (RayZath/loader.cpp:1039-1055, json_loader.cpp:1062-1117)  .json scenes
 *   OBJLoader::parseOBJ / loadInstances      (RayZath/loader.cpp:667-737, 738-1035)                      .obj geometry
 *   MTLLoader::parseMTL                      (RayZath/loader.cpp:334-638)                                .mtl materials
 *   JsonSaver / OBJSaver / MTLSaver          (RayZath/json_saver.cpp, saver.cpp)                         writers
 * and hands the result over as the POD snapshot hiprz_upload_scene / hiprz_upload_camera take (include/hiprz.h).
 * The C++ interface underneath is rayzath_amd/csrc/scene_io.hpp.  Plain pointers and sizes only. */
#ifndef HIPRZ_IO_H
#define HIPRZ_IO_H

#include <stddef.h>

#include "hiprz.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hiprz_scene_file hiprz_scene_file;

/* Load `path`: a .json scene, or an .obj (one instance per `o`/`g`, materials from its mtllibs, default camera).
 * On success *out owns the flattened scene; on failure (HIPRZ_ERR_INVALID) *out is NULL and hiprz_io_last_error()
 * says why.  Problems inside the file that the reference only logs (unknown statements, bad indices, missing
 * maps ...) do not fail the call: read them with hiprz_scene_file_log. */
int hiprz_scene_file_load(const char* path, hiprz_scene_file** out);
void hiprz_scene_file_free(hiprz_scene_file* file);
const hiprz_scene* hiprz_scene_file_scene(const hiprz_scene_file* file);   /* valid until hiprz_scene_file_free */
const hiprz_camera* hiprz_scene_file_camera(const hiprz_scene_file* file);  /* the first enabled camera (default camera if none) */
/* every ENABLED camera of the file, in file order: the reference renders all of them per call (cpu_engine_renderer.cpp:97-117) */
uint32_t hiprz_scene_file_camera_count(const hiprz_scene_file* file);
int hiprz_scene_file_camera_at(const hiprz_scene_file* file, uint32_t index, hiprz_camera* out);
/* "[message] ...\n[warning] ...\n[error] ...\n" in the reference's LoadResult order */
const char* hiprz_scene_file_log(const hiprz_scene_file* file);
uint32_t hiprz_scene_file_error_count(const hiprz_scene_file* file);
uint32_t hiprz_scene_file_warning_count(const hiprz_scene_file* file);
/* Write the loaded world back: `kind` 0 = .json with inline meshes, 1 = .obj + .mtl next to it. */
int hiprz_scene_file_save(const hiprz_scene_file* file, const char* path, int kind);

/* Image files of maps and saved frames (rayzath_amd/csrc/image_io.hpp; the reference: stbi_load / stbi_write_png, loader.cpp:36-98,
 * saver.cpp:16-60).  Decodes PNG (every colour type / bit depth, palette, tRNS, Adam7), baseline JPEG, BMP, TGA and binary PPM / PGM to 8 bits
 * per channel, rows top to bottom.  `channels` = 0 keeps the file's channel count (reported in *channels_out: 1 grey, 2 grey +
 * alpha, 3 RGB, 4 RGBA), 1..4 converts like stb_image does (grey = (77 r + 150 g + 29 b) >> 8, missing alpha = 255).
 * Call with pixels = NULL to learn the size; otherwise capacity must be >= width * height * channels. */
int hiprz_image_read(const char* path, uint32_t channels, uint32_t* width_out, uint32_t* height_out, uint32_t* channels_out,
                     uint8_t* pixels, size_t capacity);
/* 8-bit PNG from `channels` (1..4) interleaved channels, rows top to bottom. */
int hiprz_image_write_png(const char* path, const uint8_t* pixels, uint32_t width, uint32_t height, uint32_t channels);
/* One float per pixel, what stbi_loadf(path, .., 1) gives an emission map: Radiance .hdr (RGBE) as (r + g + b) / 3, any 8-bit
 * format as pow(grey / 255, 2.2).  pixels = NULL reports the size only; capacity counts floats. */
int hiprz_image_read_f32(const char* path, uint32_t* width_out, uint32_t* height_out, float* pixels, size_t capacity);
/* Radiance .hdr (RGBE) from one float per pixel (stbi_write_hdr(path, w, h, 1, data), saver.cpp:76-95). */
int hiprz_image_write_hdr(const char* path, const float* pixels, uint32_t width, uint32_t height);
const char* hiprz_io_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
