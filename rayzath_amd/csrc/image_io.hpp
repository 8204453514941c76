// image_io.hpp — image files for the host side's maps and saved frames.
//
// The reference decodes map files with stb_image (loader.cpp:36-98: stbi_load(path, .., 4) for textures and normal maps,
// stbi_load(path, .., 1) for metalness / roughness maps, stbi_loadf(path, .., 1) for emission maps) and writes them with
// stbi_write_png (saver.cpp:30,56).  stb_image is not part of this repository; this is an own decoder for the formats scene
// authors actually ship next to an .obj / .mtl:
//   PNG  all colour types and bit depths, palette + tRNS, Adam7 interlacing; chunk CRCs are verified; inflate by zlib
//   BMP  uncompressed 8-bit paletted, 24 and 32 bits per pixel, bottom-up or top-down
//   TGA  types 2 / 3 / 10 / 11 (true colour or grey, raw or run-length encoded), 8 / 24 / 32 bits, either row order
//   JPEG baseline, extended sequential and progressive (Huffman, 8 bits, grey or YCbCr, any 1-2-4 sampling, restart intervals; T.81 Annex G:
//        spectral selection + successive approximation, a file cut short decodes from the scans it has); arithmetic coding and 12 bits refused
//   PNM  binary P5 / P6
// with stb_image's conventions where a file leaves a choice: 16-bit samples keep their high byte, 1/2/4-bit grey is scaled to
// 0..255, a tRNS colour key becomes alpha 0, JPEG chroma is upsampled with its 3:1 triangle filter and converted with its fixed-point
// matrix, channels come out in R G B A order, rows top to bottom.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace RayZath::Hip::IO {

struct Image {
    uint32_t width = 0, height = 0, channels = 0;  // 1 grey, 2 grey + alpha, 3 RGB, 4 RGBA; 8 bits per channel
    std::vector<uint8_t> data;
};

// false + `why` when the file cannot be opened, is damaged or is of a kind that is not decoded here
bool readImage(const std::string& path, Image& out, std::string& why);
bool decodeImage(const uint8_t* bytes, size_t size, const std::string& name, Image& out, std::string& why);

// stb_image's channel conversion (stbi__convert_format): grey from RGB = (77 r + 150 g + 29 b) >> 8, missing alpha = 255
std::vector<uint8_t> convertChannels(const Image& img, uint32_t channels);

// stbi_loadf(path, .., 1): one float per pixel.  Radiance .hdr files (RGBE, flat or run-length encoded scanlines) give
// (r + g + b) / 3 of the decoded floats; every 8-bit format gives pow(grey / 255, 2.2), as stb_image's LDR-to-HDR conversion does.
bool readImageF32(const std::string& path, uint32_t& width, uint32_t& height, std::vector<float>& out, std::string& why);
// stbi_write_hdr(path, w, h, 1, data): Radiance RGBE with r = g = b = the value (mantissa of the largest component, shared exponent)
bool writeHDR(const std::string& path, const float* pixels, uint32_t width, uint32_t height, std::string& why);

// 8-bit PNG, colour type by `channels` (1, 2, 3 or 4), no interlacing, filter chosen per row (minimum sum of absolute differences)
bool writePNG(const std::string& path, const uint8_t* pixels, uint32_t width, uint32_t height, uint32_t channels, std::string& why);

}  // namespace RayZath::Hip::IO
