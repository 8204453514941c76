// image_io.cpp — see image_io.hpp.  Host-only C++17; inflate / deflate / crc32 come from zlib.
#include "image_io.hpp"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

namespace RayZath::Hip::IO {
namespace {

constexpr uint32_t kMaxSide = 32768u;

uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
uint32_t le32(const uint8_t* p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24); }
uint32_t le16(const uint8_t* p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8); }
void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(uint8_t(x >> 24)), v.push_back(uint8_t(x >> 16)), v.push_back(uint8_t(x >> 8)), v.push_back(uint8_t(x));
}

// ---------------------------------------------------------------------------------------------------------------------
// PNG (ISO/IEC 15948)
// ---------------------------------------------------------------------------------------------------------------------
const uint8_t kPngSignature[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};

int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// reverses the row filters of one (sub-)image in place; rows are `stride` bytes after their filter-type byte
bool unfilter(uint8_t* data, uint32_t rows, size_t stride, uint32_t bytes_per_pixel) {
    std::vector<uint8_t> zero(stride, 0);
    const uint8_t* prev = zero.data();
    for (uint32_t y = 0; y < rows; ++y) {
        uint8_t* line = data + size_t(y) * (stride + 1);
        const uint8_t type = line[0];
        uint8_t* cur = line + 1;
        if (type > 4) return false;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bytes_per_pixel ? cur[i - bytes_per_pixel] : 0, b = prev[i], c = i >= bytes_per_pixel ? prev[i - bytes_per_pixel] : 0;
            int add = 0;
            if (type == 1) add = a;
            else if (type == 2) add = b;
            else if (type == 3) add = (a + b) >> 1;
            else if (type == 4) add = paeth(a, b, c);
            cur[i] = uint8_t(cur[i] + add);
        }
        prev = cur;
    }
    return true;
}

struct PngHeader {
    uint32_t width = 0, height = 0;
    uint8_t depth = 0, color_type = 0, interlace = 0;
    uint32_t samples() const { return color_type == 0 ? 1u : color_type == 2 ? 3u : color_type == 3 ? 1u : color_type == 4 ? 2u : 4u; }
    uint32_t bits_per_pixel() const { return samples() * depth; }
};

bool decode_png(const uint8_t* bytes, size_t size, const std::string& name, Image& out, std::string& why) {
    auto bad = [&](const std::string& m) { return why = name + ": " + m, false; };
    size_t pos = 8;
    PngHeader h;
    bool have_header = false, have_end = false;
    std::vector<uint8_t> idat, palette, trns;
    while (!have_end) {
        if (pos + 12 > size) return bad("truncated PNG (chunk header)");
        const uint32_t len = be32(bytes + pos);
        const uint8_t* type = bytes + pos + 4;
        if (len > 0x7FFFFFFFu || pos + 12 + size_t(len) > size) return bad("truncated PNG (chunk data)");
        const uint8_t* data = bytes + pos + 8;
        if (uint32_t(crc32(crc32(0L, Z_NULL, 0), type, uInt(len + 4))) != be32(data + len)) return bad("PNG chunk CRC mismatch");
        const std::string t(reinterpret_cast<const char*>(type), 4);
        if (!have_header && t != "IHDR") return bad("PNG does not start with IHDR");
        if (t == "IHDR") {
            if (len != 13 || have_header) return bad("bad IHDR");
            h.width = be32(data), h.height = be32(data + 4), h.depth = data[8], h.color_type = data[9], h.interlace = data[12];
            if (data[10] != 0 || data[11] != 0 || h.interlace > 1) return bad("unknown PNG compression / filter / interlace method");
            if (h.width == 0 || h.height == 0 || h.width > kMaxSide || h.height > kMaxSide) return bad("PNG dimensions out of range");
            const uint8_t d = h.depth;
            const bool ok = (h.color_type == 0 && (d == 1 || d == 2 || d == 4 || d == 8 || d == 16)) || (h.color_type == 3 && (d == 1 || d == 2 || d == 4 || d == 8)) ||
                            ((h.color_type == 2 || h.color_type == 4 || h.color_type == 6) && (d == 8 || d == 16));
            if (!ok) return bad("invalid PNG colour type / bit depth");
            have_header = true;
        } else if (t == "PLTE") {
            if (len == 0 || len % 3 != 0 || len > 768) return bad("bad PLTE");
            palette.assign(data, data + len);
        } else if (t == "tRNS") {
            trns.assign(data, data + len);
        } else if (t == "IDAT") {
            idat.insert(idat.end(), data, data + len);
        } else if (t == "IEND") {
            have_end = true;
        } else if (!(type[0] & 0x20)) {
            return bad("unknown critical PNG chunk " + t);
        }
        pos += 12 + size_t(len);
    }
    if (idat.empty()) return bad("PNG without image data");
    if (h.color_type == 3 && palette.empty()) return bad("paletted PNG without PLTE");

    // sub-images: the whole picture, or the seven Adam7 passes
    struct Pass { uint32_t x0, y0, dx, dy, w, h; size_t offset; };
    static const uint32_t ax0[7] = {0, 4, 0, 2, 0, 1, 0}, ay0[7] = {0, 0, 4, 0, 2, 0, 1}, adx[7] = {8, 8, 4, 4, 2, 2, 1}, ady[7] = {8, 8, 8, 4, 4, 2, 2};
    std::vector<Pass> passes;
    const uint32_t bpp = h.bits_per_pixel();
    size_t raw_size = 0;
    auto add_pass = [&](uint32_t x0, uint32_t y0, uint32_t dx, uint32_t dy) {
        if (x0 >= h.width || y0 >= h.height) return;
        Pass p{x0, y0, dx, dy, (h.width - x0 + dx - 1) / dx, (h.height - y0 + dy - 1) / dy, raw_size};
        raw_size += size_t(p.h) * (1 + (size_t(p.w) * bpp + 7) / 8);
        passes.push_back(p);
    };
    if (h.interlace) for (int i = 0; i < 7; ++i) add_pass(ax0[i], ay0[i], adx[i], ady[i]);
    else add_pass(0, 0, 1, 1);

    std::vector<uint8_t> raw(raw_size);
    uLongf got = uLongf(raw_size);
    const int zr = uncompress(raw.data(), &got, idat.data(), uLong(idat.size()));
    if (zr != Z_OK || size_t(got) != raw_size) return bad("PNG image data does not inflate to the size its header announces");

    const bool key = !trns.empty() && (h.color_type == 0 || h.color_type == 2);
    if (key && trns.size() < (h.color_type == 0 ? 2u : 6u)) return bad("bad tRNS");
    const bool palette_alpha = h.color_type == 3 && !trns.empty();
    out.width = h.width, out.height = h.height;
    out.channels = h.color_type == 0 ? (key ? 2u : 1u) : h.color_type == 2 ? (key ? 4u : 3u) : h.color_type == 3 ? (palette_alpha ? 4u : 3u) : h.color_type == 4 ? 2u : 4u;
    out.data.assign(size_t(h.width) * h.height * out.channels, 0);
    const uint32_t n_samples = h.samples(), filter_bpp = std::max(1u, bpp / 8u);
    static const uint32_t grey_scale[5] = {0, 0xFF, 0x55, 0, 0x11};
    const uint32_t key16[3] = {key ? (uint32_t(trns[0]) << 8) | trns[1] : 0u, key && h.color_type == 2 ? (uint32_t(trns[2]) << 8) | trns[3] : 0u,
                               key && h.color_type == 2 ? (uint32_t(trns[4]) << 8) | trns[5] : 0u};
    for (const Pass& p : passes) {
        const size_t stride = (size_t(p.w) * bpp + 7) / 8;
        if (!unfilter(raw.data() + p.offset, p.h, stride, filter_bpp)) return bad("unknown PNG row filter");
        for (uint32_t y = 0; y < p.h; ++y) {
            const uint8_t* row = raw.data() + p.offset + size_t(y) * (stride + 1) + 1;
            for (uint32_t x = 0; x < p.w; ++x) {
                uint32_t sample[4] = {0, 0, 0, 0};  // at the file's bit depth
                for (uint32_t s = 0; s < n_samples; ++s) {
                    const size_t idx = size_t(x) * n_samples + s;
                    if (h.depth == 16) sample[s] = (uint32_t(row[2 * idx]) << 8) | row[2 * idx + 1];
                    else if (h.depth == 8) sample[s] = row[idx];
                    else {
                        const size_t bit = idx * h.depth;
                        sample[s] = (row[bit >> 3] >> (8 - h.depth - (bit & 7))) & ((1u << h.depth) - 1u);
                    }
                }
                uint8_t* o = &out.data[(size_t(p.y0 + y * p.dy) * h.width + (p.x0 + x * p.dx)) * out.channels];
                auto to8 = [&](uint32_t v) { return uint8_t(h.depth == 16 ? v >> 8 : h.depth == 8 ? v : v * grey_scale[h.depth]); };
                if (h.color_type == 3) {
                    if (3 * sample[0] + 2 >= palette.size()) return bad("PNG palette index out of range");
                    o[0] = palette[3 * sample[0]], o[1] = palette[3 * sample[0] + 1], o[2] = palette[3 * sample[0] + 2];
                    if (palette_alpha) o[3] = sample[0] < trns.size() ? trns[sample[0]] : 255;
                } else if (h.color_type == 0) {
                    o[0] = to8(sample[0]);
                    if (key) o[1] = sample[0] == key16[0] ? 0 : 255;
                } else if (h.color_type == 2) {
                    o[0] = to8(sample[0]), o[1] = to8(sample[1]), o[2] = to8(sample[2]);
                    if (key) o[3] = (sample[0] == key16[0] && sample[1] == key16[1] && sample[2] == key16[2]) ? 0 : 255;
                } else {
                    for (uint32_t s = 0; s < n_samples; ++s) o[s] = to8(sample[s]);
                }
            }
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// BMP
// ---------------------------------------------------------------------------------------------------------------------
bool decode_bmp(const uint8_t* b, size_t size, const std::string& name, Image& out, std::string& why) {
    auto bad = [&](const std::string& m) { return why = name + ": " + m, false; };
    if (size < 54) return bad("truncated BMP header");
    const uint32_t data_offset = le32(b + 10), dib = le32(b + 14);
    if (dib < 40) return bad("BMP with an OS/2 header is not decoded");
    const int32_t w = int32_t(le32(b + 18)), hs = int32_t(le32(b + 22));
    const uint32_t bpp = le16(b + 28), compression = le32(b + 30);
    uint32_t colors = le32(b + 46);
    const bool top_down = hs < 0;
    const uint32_t h = uint32_t(top_down ? -int64_t(hs) : hs);
    if (w <= 0 || h == 0 || uint32_t(w) > kMaxSide || h > kMaxSide) return bad("BMP dimensions out of range");
    if (!(bpp == 8 || bpp == 24 || bpp == 32)) return bad("only 8, 24 and 32 bits per pixel BMP files are decoded");
    uint32_t shift[4] = {16, 8, 0, 24};  // B G R A byte order of BI_RGB
    bool has_alpha = bpp == 32;
    if (compression == 3 && bpp == 32) {
        const size_t m = dib >= 52 ? 54 : 14 + size_t(dib);  // masks: inside a V2+ header or right behind BITMAPINFOHEADER
        if (m + 12 > size) return bad("truncated BMP bit masks");
        const uint32_t mask[4] = {le32(b + m), le32(b + m + 4), le32(b + m + 8), dib >= 56 && m + 16 <= size ? le32(b + m + 12) : 0u};
        for (int c = 0; c < 4; ++c) {
            if (mask[c] == 0xFFu) shift[c] = 0;
            else if (mask[c] == 0xFF00u) shift[c] = 8;
            else if (mask[c] == 0xFF0000u) shift[c] = 16;
            else if (mask[c] == 0xFF000000u) shift[c] = 24;
            else if (c == 3 && mask[c] == 0) has_alpha = false;
            else return bad("BMP bit masks that are not whole bytes are not decoded");
        }
    } else if (compression != 0) {
        return bad("compressed BMP files are not decoded");
    }
    const size_t stride = ((size_t(w) * bpp + 31) / 32) * 4;
    if (size_t(data_offset) + stride * h > size) return bad("truncated BMP pixel data");
    const uint8_t* pal = b + 14 + dib;
    if (bpp == 8) {
        if (colors == 0 || colors > 256) colors = 256;
        if (14 + size_t(dib) + 4 * size_t(colors) > size) return bad("truncated BMP palette");
    }
    out.width = uint32_t(w), out.height = h, out.channels = has_alpha ? 4u : 3u;
    out.data.assign(size_t(w) * h * out.channels, 255);
    bool any_alpha = false;
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* row = b + data_offset + stride * (top_down ? y : h - 1 - y);
        for (uint32_t x = 0; x < uint32_t(w); ++x) {
            uint8_t* o = &out.data[(size_t(y) * w + x) * out.channels];
            if (bpp == 8) {
                const uint32_t i = row[x];
                if (i >= colors) return bad("BMP palette index out of range");
                o[0] = pal[4 * i + 2], o[1] = pal[4 * i + 1], o[2] = pal[4 * i];
            } else if (bpp == 24) {
                o[0] = row[3 * x + 2], o[1] = row[3 * x + 1], o[2] = row[3 * x];
            } else {
                const uint32_t v = le32(row + 4 * x);
                o[0] = uint8_t(v >> shift[0]), o[1] = uint8_t(v >> shift[1]), o[2] = uint8_t(v >> shift[2]);
                if (has_alpha) o[3] = uint8_t(v >> shift[3]), any_alpha = any_alpha || o[3] != 0;
            }
        }
    }
    if (has_alpha && !any_alpha)  // an all-zero alpha channel is an unused one (stb_image does the same)
        for (size_t i = 3; i < out.data.size(); i += 4) out.data[i] = 255;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// TGA
// ---------------------------------------------------------------------------------------------------------------------
bool decode_tga(const uint8_t* b, size_t size, const std::string& name, Image& out, std::string& why) {
    auto bad = [&](const std::string& m) { return why = name + ": " + m, false; };
    if (size < 18) return bad("truncated TGA header");
    const uint32_t id_len = b[0], cmap_type = b[1], type = b[2], w = le16(b + 12), h = le16(b + 14), bpp = b[16], desc = b[17];
    if (cmap_type != 0 || !(type == 2 || type == 3 || type == 10 || type == 11)) return bad("only true-colour and grey TGA files (types 2, 3, 10, 11) are decoded");
    const bool grey = type == 3 || type == 11, rle = type >= 10;
    if ((grey && bpp != 8) || (!grey && bpp != 24 && bpp != 32)) return bad("only 8-bit grey and 24 / 32-bit colour TGA files are decoded");
    if (w == 0 || h == 0) return bad("TGA dimensions out of range");
    const uint32_t bytes = bpp / 8;
    out.width = w, out.height = h, out.channels = grey ? 1u : bytes;
    out.data.assign(size_t(w) * h * out.channels, 0);
    size_t pos = 18 + size_t(id_len);
    const size_t n = size_t(w) * h;
    uint8_t px[4] = {0, 0, 0, 255};
    size_t i = 0;
    auto store = [&](size_t idx, const uint8_t* p) {
        size_t y = idx / w, x = idx % w;
        if (!(desc & 0x20)) y = h - 1 - y;  // bottom-up unless bit 5 is set
        if (desc & 0x10) x = w - 1 - x;
        uint8_t* o = &out.data[(y * w + x) * out.channels];
        if (grey) o[0] = p[0];
        else {
            o[0] = p[2], o[1] = p[1], o[2] = p[0];
            if (bytes == 4) o[3] = p[3];
        }
    };
    while (i < n) {
        if (!rle) {
            if (pos + bytes > size) return bad("truncated TGA pixel data");
            store(i++, b + pos), pos += bytes;
            continue;
        }
        if (pos + 1 > size) return bad("truncated TGA packet");
        const uint32_t head = b[pos++], count = (head & 127u) + 1u;
        if (i + count > n) return bad("TGA packet runs past the image");
        if (head & 128u) {
            if (pos + bytes > size) return bad("truncated TGA packet");
            std::memcpy(px, b + pos, bytes), pos += bytes;
            for (uint32_t k = 0; k < count; ++k) store(i++, px);
        } else {
            if (pos + size_t(bytes) * count > size) return bad("truncated TGA packet");
            for (uint32_t k = 0; k < count; ++k) store(i++, b + pos), pos += bytes;
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// binary PNM (P5 / P6)
// ---------------------------------------------------------------------------------------------------------------------
bool decode_pnm(const uint8_t* b, size_t size, const std::string& name, Image& out, std::string& why) {
    auto bad = [&](const std::string& m) { return why = name + ": " + m, false; };
    size_t pos = 0;
    auto token = [&]() {
        std::string t;
        while (pos < size) {
            if (b[pos] == '#') while (pos < size && b[pos] != '\n') ++pos;
            else if (std::isspace(b[pos])) ++pos;
            else break;
        }
        while (pos < size && !std::isspace(b[pos])) t.push_back(char(b[pos++]));
        return t;
    };
    const std::string magic = token();
    const long w = std::atol(token().c_str()), h = std::atol(token().c_str()), maxv = std::atol(token().c_str());
    if (w <= 0 || h <= 0 || w > long(kMaxSide) || h > long(kMaxSide) || maxv != 255) return bad("unsupported PNM header");
    ++pos;  // the single whitespace after maxval
    out.width = uint32_t(w), out.height = uint32_t(h), out.channels = magic == "P6" ? 3u : 1u;
    const size_t n = size_t(w) * size_t(h) * out.channels;
    if (pos + n > size) return bad("truncated image");
    out.data.assign(b + pos, b + pos + n);
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// JPEG (ITU-T T.81): baseline / extended sequential and PROGRESSIVE DCT (spectral selection + successive approximation, Annex G), Huffman
// coded, 8-bit samples, 1 or 3 components, interleaved and per-component scans, restart intervals — what the reference reads through
// stb_image (loader.cpp:36-144).
// Chroma is upsampled the way stb_image does it (3:1 triangle filter for factor 2, replication otherwise) and converted with its
// 20-bit fixed-point YCbCr matrix; the inverse DCT is a separable double-precision one rounded to nearest, so a sample can differ
// from stb_image's integer IDCT by one code value.  Arithmetic-coded, lossless and hierarchical files are refused.
// ---------------------------------------------------------------------------------------------------------------------
struct JpegHuffman {
    uint8_t bits[17] = {0};
    uint8_t values[256] = {0};
    int32_t mincode[17], maxcode[18], valptr[17];
    bool defined = false;
    void build() {
        int32_t code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            valptr[len] = k;
            mincode[len] = code;
            code += bits[len];
            k += bits[len];
            maxcode[len] = bits[len] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        defined = true;
    }
};
struct JpegBits {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int count = 0;
    bool hit_marker = false;
    int bit() {
        if (count == 0) {
            uint8_t b = 0;
            if (p < end && !hit_marker) {
                b = *p++;
                if (b == 0xFF) {
                    if (p < end && *p == 0x00) ++p;          // stuffed zero
                    else hit_marker = true, b = 0, --p;      // a marker: feed zeros, leave it for the caller
                }
            }
            acc = b, count = 8;
        }
        --count;
        return int((acc >> count) & 1u);
    }
    int receive(int n) {
        int v = 0;
        for (int i = 0; i < n; ++i) v = (v << 1) | bit();
        return v;
    }
    void reset() { acc = 0, count = 0, hit_marker = false; }
};
int jpeg_decode_symbol(JpegBits& br, const JpegHuffman& h) {
    int32_t code = 0;
    for (int len = 1; len <= 16; ++len) {
        code = (code << 1) | br.bit();
        if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.values[h.valptr[len] + code - h.mincode[len]];
    }
    return -1;
}
int jpeg_extend(int v, int t) { return t && v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }
const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
void jpeg_idct(const int* coef, uint8_t* out, size_t stride) {
    static double basis[8][8];
    static bool ready = false;
    if (!ready) {
        for (int x = 0; x < 8; ++x)
            for (int u = 0; u < 8; ++u) basis[x][u] = (u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
        ready = true;
    }
    double tmp[64];
    for (int y = 0; y < 8; ++y)       // rows: over u
        for (int x = 0; x < 8; ++x) {
            double a = 0.0;
            for (int u = 0; u < 8; ++u) a += basis[x][u] * coef[y * 8 + u];
            tmp[y * 8 + x] = a;
        }
    for (int x = 0; x < 8; ++x)       // columns: over v
        for (int y = 0; y < 8; ++y) {
            double a = 0.0;
            for (int v = 0; v < 8; ++v) a += basis[y][v] * tmp[v * 8 + x];
            const long r = std::lround(a + 128.0);
            out[size_t(y) * stride + x] = uint8_t(r < 0 ? 0 : r > 255 ? 255 : r);
        }
}
struct JpegComponent {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0;
    uint32_t w = 0, hgt = 0;  // padded plane size
    std::vector<uint8_t> plane;
};
// stb_image's chroma upsampling of one row pair: factor 2 horizontally and / or vertically with a 3:1 triangle filter
void jpeg_upsample_row(const uint8_t* near_row, const uint8_t* far_row, uint32_t w, int hs, int vs, uint8_t* out) {
    if (hs == 1 && vs == 1) {
        std::memcpy(out, near_row, w);
    } else if (hs == 1 && vs == 2) {
        for (uint32_t i = 0; i < w; ++i) out[i] = uint8_t((3 * near_row[i] + far_row[i] + 2) >> 2);
    } else if (hs == 2 && vs == 1) {
        if (w == 1) { out[0] = out[1] = near_row[0]; return; }
        out[0] = near_row[0];
        out[1] = uint8_t((near_row[0] * 3 + near_row[1] + 2) >> 2);
        uint32_t i = 1;
        for (; i + 1 < w; ++i) {
            const int n = 3 * near_row[i] + 2;
            out[2 * i] = uint8_t((n + near_row[i - 1]) >> 2);
            out[2 * i + 1] = uint8_t((n + near_row[i + 1]) >> 2);
        }
        out[2 * i] = uint8_t((near_row[w - 2] * 3 + near_row[w - 1] + 2) >> 2);
        out[2 * i + 1] = near_row[w - 1];
    } else if (hs == 2 && vs == 2) {
        if (w == 1) { out[0] = out[1] = uint8_t((3 * near_row[0] + far_row[0] + 2) >> 2); return; }
        int t1 = 3 * near_row[0] + far_row[0];
        out[0] = uint8_t((t1 + 2) >> 2);
        for (uint32_t i = 1; i < w; ++i) {
            const int t0 = t1;
            t1 = 3 * near_row[i] + far_row[i];
            out[2 * i - 1] = uint8_t((3 * t0 + t1 + 8) >> 4);
            out[2 * i] = uint8_t((3 * t1 + t0 + 8) >> 4);
        }
        out[2 * w - 1] = uint8_t((t1 + 2) >> 2);
    } else {
        for (uint32_t i = 0; i < w; ++i)
            for (int j = 0; j < hs; ++j) out[i * hs + j] = near_row[i];
    }
}
// One scan's entropy-coded segment decoded into the frame's coefficient arrays.  Sequential scans (Ss = 0, Se = 63, Ah = Al = 0) and
// the four progressive procedures of T.81 Annex G: DC first / DC refinement (G.1.2.1), AC first with end-of-band runs (G.1.2.2),
// AC refinement with its correction bits (G.1.2.3, Figure G.7).  Coefficients stay UNquantised (point-transformed by Al) until all
// scans are in.
struct JpegScan {
    int n = 0;              // components of this scan
    int comp[3] = {0, 0, 0};
    int ss = 0, se = 63, ah = 0, al = 0;
};
bool decode_jpeg(const uint8_t* b, size_t size, const std::string& name, Image& out, std::string& why) {
    auto bad = [&](const std::string& m) { return why = name + ": " + m, false; };
    uint16_t quant[4][64] = {};
    JpegHuffman dc[4], ac[4];
    JpegComponent comp[3];
    std::vector<int16_t> coefs[3];       // per component: blocks in raster order of its padded block grid, 64 coefficients each (natural order)
    uint32_t blocks_w[3] = {0, 0, 0}, blocks_h[3] = {0, 0, 0};   // padded grid (whole MCUs)
    uint32_t own_w[3] = {0, 0, 0}, own_h[3] = {0, 0, 0};         // the component's own extent in blocks: what a scan of this component alone covers (A.2.3)
    int n_comp = 0, hmax = 1, vmax = 1;
    uint32_t width = 0, height = 0, restart_interval = 0, mcus_x = 0, mcus_y = 0;
    bool have_frame = false, progressive = false, any_scan = false;
    size_t pos = 2;
    while (true) {
        if (pos + 4 > size) {
            if (any_scan) break;  // no EOI: stb_image shows what it has, too
            return bad("truncated JPEG");
        }
        if (b[pos] != 0xFF) {
            if (any_scan) {  // stray bytes behind a scan: look for the next marker
                ++pos;
                continue;
            }
            return bad("JPEG marker expected");
        }
        while (pos < size && b[pos] == 0xFF) ++pos;
        if (pos >= size) break;
        const uint8_t marker = b[pos++];
        if (marker == 0xD9) {
            if (any_scan) break;
            return bad("JPEG ends before its scan");
        }
        if (marker == 0x00 || marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7)) continue;
        if (pos + 2 > size) return bad("truncated JPEG");
        const size_t len = (size_t(b[pos]) << 8) | b[pos + 1];
        if (len < 2 || pos + len > size) return bad("truncated JPEG segment");
        const uint8_t* d = b + pos + 2;
        const size_t n = len - 2;
        if (marker == 0xDB) {  // DQT
            size_t i = 0;
            while (i < n) {
                const int pq = d[i] >> 4, tq = d[i] & 15;
                ++i;
                if (tq > 3 || i + (pq ? 128 : 64) > n) return bad("bad DQT");
                for (int k = 0; k < 64; ++k) quant[tq][kZigzag[k]] = pq ? uint16_t((d[i + 2 * k] << 8) | d[i + 2 * k + 1]) : d[i + k];
                i += pq ? 128 : 64;
            }
        } else if (marker == 0xC4) {  // DHT
            size_t i = 0;
            while (i < n) {
                if (i + 17 > n) return bad("bad DHT");
                const int tc = d[i] >> 4, th = d[i] & 15;
                if (tc > 1 || th > 3) return bad("bad DHT");
                JpegHuffman& h = tc ? ac[th] : dc[th];
                int total = 0;
                for (int k = 1; k <= 16; ++k) h.bits[k] = d[i + k], total += d[i + k];
                i += 17;
                if (total > 256 || i + size_t(total) > n) return bad("bad DHT");
                std::memcpy(h.values, d + i, size_t(total));
                i += size_t(total);
                h.build();
            }
        } else if (marker == 0xC0 || marker == 0xC1 || marker == 0xC2) {  // SOF0 / SOF1 / SOF2 (progressive)
            if (have_frame) return bad("a second JPEG frame header");
            if (n < 6 || d[0] != 8) return bad("only 8-bit JPEG files are decoded");
            height = (uint32_t(d[1]) << 8) | d[2], width = (uint32_t(d[3]) << 8) | d[4];
            n_comp = d[5];
            if (width == 0 || height == 0) return bad("JPEG dimensions out of range");
            if (!(n_comp == 1 || n_comp == 3) || n < size_t(6 + 3 * n_comp)) return bad("only grey and three-component JPEG files are decoded");
            for (int c = 0; c < n_comp; ++c) {
                comp[c].id = d[6 + 3 * c], comp[c].h = d[7 + 3 * c] >> 4, comp[c].v = d[7 + 3 * c] & 15, comp[c].tq = d[8 + 3 * c];
                if (comp[c].h < 1 || comp[c].h > 4 || comp[c].v < 1 || comp[c].v > 4 || comp[c].tq > 3) return bad("bad JPEG frame header");
                hmax = std::max(hmax, comp[c].h), vmax = std::max(vmax, comp[c].v);
            }
            if (n_comp == 1) comp[0].h = comp[0].v = hmax = vmax = 1;  // a single component is never interleaved: one block per MCU
            mcus_x = (width + 8u * uint32_t(hmax) - 1) / (8u * uint32_t(hmax)), mcus_y = (height + 8u * uint32_t(vmax) - 1) / (8u * uint32_t(vmax));
            for (int c = 0; c < n_comp; ++c) {
                blocks_w[c] = mcus_x * uint32_t(comp[c].h), blocks_h[c] = mcus_y * uint32_t(comp[c].v);
                const uint32_t cw = (width * uint32_t(comp[c].h) + uint32_t(hmax) - 1) / uint32_t(hmax), ch = (height * uint32_t(comp[c].v) + uint32_t(vmax) - 1) / uint32_t(vmax);
                own_w[c] = (cw + 7) / 8, own_h[c] = (ch + 7) / 8;
                if (size_t(blocks_w[c]) * blocks_h[c] > (size_t(1) << 26)) return bad("JPEG dimensions out of range");
                coefs[c].assign(size_t(blocks_w[c]) * blocks_h[c] * 64u, 0);
            }
            progressive = marker == 0xC2;
            have_frame = true;
        } else if (marker >= 0xC3 && marker <= 0xCF && marker != 0xC8 && marker != 0xCC) {
            return bad("this JPEG coding process is not decoded (Huffman-coded baseline, extended sequential and progressive DCT are)");
        } else if (marker == 0xDD) {
            if (n < 2) return bad("bad DRI");
            restart_interval = (uint32_t(d[0]) << 8) | d[1];
        } else if (marker == 0xDA) {  // SOS: one scan; its entropy-coded segment follows the header
            if (!have_frame) return bad("JPEG scan before its frame header");
            JpegScan scan;
            scan.n = n ? d[0] : 0;
            if (scan.n < 1 || scan.n > n_comp || n < size_t(1 + 2 * scan.n + 3)) return bad("bad JPEG scan header");
            for (int k = 0; k < scan.n; ++k) {
                int c = 0;
                while (c < n_comp && comp[c].id != d[1 + 2 * k]) ++c;
                if (c == n_comp) return bad("bad JPEG scan header");
                for (int e = 0; e < k; ++e)
                    if (scan.comp[e] == c) return bad("bad JPEG scan header");
                scan.comp[k] = c;
                comp[c].td = d[2 + 2 * k] >> 4, comp[c].ta = d[2 + 2 * k] & 15;
                if (comp[c].td > 3 || comp[c].ta > 3) return bad("bad JPEG scan header");
            }
            scan.ss = d[1 + 2 * scan.n], scan.se = d[2 + 2 * scan.n], scan.ah = d[3 + 2 * scan.n] >> 4, scan.al = d[3 + 2 * scan.n] & 15;
            if (!progressive) {
                if (scan.ss != 0 || scan.se != 63 || scan.ah != 0 || scan.al != 0) return bad("bad JPEG scan header (a sequential scan covers all 64 coefficients)");
            } else if (scan.ss > scan.se || scan.se > 63 || scan.ah > 13 || scan.al > 13 || (scan.ss == 0 && scan.se != 0) || (scan.ss != 0 && scan.n != 1)) {
                return bad("bad progressive JPEG scan header");  // G.1.1.1.1: DC scans hold only DC, AC scans one component
            }
            const bool dc_scan = scan.ss == 0, ac_scan = scan.se > 0;
            for (int k = 0; k < scan.n; ++k) {
                const JpegComponent& cc = comp[scan.comp[k]];
                if ((dc_scan && scan.ah == 0 && !dc[cc.td].defined) || (ac_scan && !ac[cc.ta].defined)) return bad("JPEG scan uses an undefined Huffman table");
            }
            pos += len;
            JpegBits br{b + pos, b + size};
            // the scan's units: MCUs of the interleaved components, or the single component's own blocks in raster order (A.2.2, A.2.3)
            const bool interleaved = scan.n > 1;
            const uint32_t units_x = interleaved ? mcus_x : own_w[scan.comp[0]], units_y = interleaved ? mcus_y : own_h[scan.comp[0]];
            uint32_t until_restart = restart_interval, eob_run = 0;
            for (int c = 0; c < n_comp; ++c) comp[c].pred = 0;
            const int p1 = 1 << scan.al, m1 = -(1 << scan.al);
            auto block = [&](int c, uint32_t bx, uint32_t by) -> bool {   // one 8x8 block of this scan
                int16_t* q = &coefs[c][(size_t(by) * blocks_w[c] + bx) * 64u];
                if (dc_scan) {
                    if (scan.ah == 0) {
                        const int t = jpeg_decode_symbol(br, dc[comp[c].td]);
                        if (t < 0 || t > 15) return false;
                        comp[c].pred += jpeg_extend(br.receive(t), t);
                        q[0] = int16_t(comp[c].pred * p1);
                    } else if (br.bit()) {
                        q[0] = int16_t(q[0] | p1);
                    }
                }
                if (!ac_scan) return true;
                const JpegHuffman& h = ac[comp[c].ta];
                int k = std::max(scan.ss, 1);
                if (scan.ah == 0) {  // first pass over this band
                    if (eob_run) {
                        --eob_run;
                        return true;
                    }
                    while (k <= scan.se) {
                        const int rs = jpeg_decode_symbol(br, h);
                        if (rs < 0) return false;
                        const int r = rs >> 4, sz = rs & 15;
                        if (sz == 0) {
                            if (r == 15) {
                                k += 16;
                                continue;
                            }
                            eob_run = (1u << r) - 1u;   // EOBn: this band ends here, in this block and in the next eob_run ones
                            if (r) eob_run += uint32_t(br.receive(r));
                            break;
                        }
                        k += r;
                        if (k > scan.se) return false;
                        q[kZigzag[k]] = int16_t(jpeg_extend(br.receive(sz), sz) * p1);
                        ++k;
                    }
                    return true;
                }
                // refinement of the band: a correction bit for every coefficient that is already non-zero, new +-1 coefficients placed
                // behind runs of coefficients that are still zero (Figure G.7)
                auto correct = [&](int16_t& v) {
                    if (br.bit() && (v & p1) == 0) v = int16_t(v >= 0 ? v + p1 : v + m1);
                };
                if (eob_run == 0) {
                    while (k <= scan.se) {
                        const int rs = jpeg_decode_symbol(br, h);
                        if (rs < 0) return false;
                        int r = rs >> 4;
                        const int sz = rs & 15;
                        int fresh = 0;
                        if (sz == 0) {
                            if (r < 15) {
                                eob_run = (1u << r);   // this block included (the tail below consumes one)
                                if (r) eob_run += uint32_t(br.receive(r));
                                break;
                            }
                        } else {
                            if (sz != 1) return false;
                            fresh = br.bit() ? p1 : m1;
                        }
                        for (; k <= scan.se; ++k) {   // pass r zero-history coefficients, correcting the others on the way
                            int16_t& v = q[kZigzag[k]];
                            if (v != 0) {
                                correct(v);
                            } else if (r-- == 0) {
                                if (fresh) v = int16_t(fresh);
                                ++k;
                                break;
                            }
                        }
                    }
                }
                if (eob_run) {   // the rest of the band holds no new coefficients: corrections only
                    for (; k <= scan.se; ++k) {
                        int16_t& v = q[kZigzag[k]];
                        if (v != 0) correct(v);
                    }
                    --eob_run;
                }
                return true;
            };
            for (uint32_t uy = 0; uy < units_y; ++uy)
                for (uint32_t ux = 0; ux < units_x; ++ux) {
                    if (restart_interval && until_restart == 0) {  // RSTn: byte-align, skip the marker, reset predictors and the EOB run
                        br.reset();
                        while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) ++br.p;
                        if (br.p + 1 >= br.end) return bad("JPEG restart marker missing");
                        br.p += 2;
                        for (int c = 0; c < n_comp; ++c) comp[c].pred = 0;
                        eob_run = 0;
                        until_restart = restart_interval;
                    }
                    if (interleaved) {
                        for (int k = 0; k < scan.n; ++k) {
                            const int c = scan.comp[k];
                            for (int by = 0; by < comp[c].v; ++by)
                                for (int bx = 0; bx < comp[c].h; ++bx)
                                    if (!block(c, ux * uint32_t(comp[c].h) + uint32_t(bx), uy * uint32_t(comp[c].v) + uint32_t(by))) return bad("bad JPEG Huffman code");
                        }
                    } else if (!block(scan.comp[0], ux, uy)) {
                        return bad("bad JPEG Huffman code");
                    }
                    if (restart_interval) --until_restart;
                }
            any_scan = true;
            pos = size_t(br.p - b);   // the next marker is at or behind the reader's position
            if (!progressive && scan.n == n_comp) break;   // the one scan of an interleaved sequential file
            continue;
        }
        pos += len;
    }
    if (!any_scan) return bad("JPEG without a scan");
    // all scans are in: dequantise, inverse DCT
    {
        int coef[64];
        for (int c = 0; c < n_comp; ++c) {
            comp[c].w = blocks_w[c] * 8u, comp[c].hgt = blocks_h[c] * 8u;
            comp[c].plane.assign(size_t(comp[c].w) * comp[c].hgt, 0);
            for (uint32_t by = 0; by < blocks_h[c]; ++by)
                for (uint32_t bx = 0; bx < blocks_w[c]; ++bx) {
                    const int16_t* q = &coefs[c][(size_t(by) * blocks_w[c] + bx) * 64u];
                    for (int k = 0; k < 64; ++k) coef[k] = int(q[k]) * int(quant[comp[c].tq][k]);
                    jpeg_idct(coef, &comp[c].plane[size_t(by) * 8u * comp[c].w + size_t(bx) * 8u], comp[c].w);
                }
            coefs[c] = std::vector<int16_t>();
        }
    }
    out.width = width, out.height = height, out.channels = uint32_t(n_comp);
    out.data.assign(size_t(width) * height * n_comp, 0);
    if (n_comp == 1) {
        for (uint32_t y = 0; y < height; ++y) std::memcpy(&out.data[size_t(y) * width], &comp[0].plane[size_t(y) * comp[0].w], width);
        return true;
    }
    std::vector<uint8_t> rows[3];
    for (int c = 0; c < 3; ++c) rows[c].resize(size_t(comp[c].w) * size_t(hmax / comp[c].h) + 8);
    for (uint32_t y = 0; y < height; ++y) {
        for (int c = 0; c < 3; ++c) {
            const int hs = hmax / comp[c].h, vs = vmax / comp[c].v;
            if (hmax % comp[c].h || vmax % comp[c].v) return bad("fractional JPEG sampling ratios are not decoded");
            const uint32_t ch = (height * uint32_t(comp[c].v) + uint32_t(vmax) - 1) / uint32_t(vmax);  // the component's own height (not the MCU padding)
            uint32_t near_y = y / uint32_t(vs), far_y = near_y;
            if (vs == 2) far_y = (y & 1u) ? std::min(near_y + 1, ch - 1) : (near_y ? near_y - 1 : 0);  // the neighbour on the side this row leans to
            else if (vs > 2) far_y = near_y;
            const uint32_t src_w = (width + uint32_t(hs) - 1) / uint32_t(hs);
            jpeg_upsample_row(&comp[c].plane[size_t(near_y) * comp[c].w], &comp[c].plane[size_t(far_y) * comp[c].w], std::min(src_w, comp[c].w), hs, vs == 2 ? 2 : 1, rows[c].data());
        }
        uint8_t* o = &out.data[size_t(y) * width * 3];
        for (uint32_t x = 0; x < width; ++x) {  // stb_image's fixed-point YCbCr -> RGB
            const int yf = (int(rows[0][x]) << 20) + (1 << 19), cb = int(rows[1][x]) - 128, cr = int(rows[2][x]) - 128;
            int r = yf + cr * 1470208, g = yf + cr * -748800 + int(uint32_t(cb * -360960) & 0xFFFF0000u), bl = yf + cb * 1858048;  // int(x * 4096 + 0.5) << 8 of 1.402, 0.71414, 0.34414, 1.772
            r >>= 20, g >>= 20, bl >>= 20;
            o[3 * x] = uint8_t(r < 0 ? 0 : r > 255 ? 255 : r), o[3 * x + 1] = uint8_t(g < 0 ? 0 : g > 255 ? 255 : g), o[3 * x + 2] = uint8_t(bl < 0 ? 0 : bl > 255 ? 255 : bl);
        }
    }
    return true;
}

std::string lower_extension(const std::string& path) {
    const size_t p = path.find_last_of('.');
    std::string e = p == std::string::npos ? std::string() : path.substr(p);
    for (auto& c : e) c = char(std::tolower(static_cast<unsigned char>(c)));
    return e;
}

}  // namespace

bool decodeImage(const uint8_t* bytes, size_t size, const std::string& name, Image& out, std::string& why) {
    out = Image{};
    if (size >= 8 && std::memcmp(bytes, kPngSignature, 8) == 0) return decode_png(bytes, size, name, out, why);
    if (size >= 2 && bytes[0] == 'B' && bytes[1] == 'M') return decode_bmp(bytes, size, name, out, why);
    if (size >= 2 && bytes[0] == 'P' && (bytes[1] == '5' || bytes[1] == '6')) return decode_pnm(bytes, size, name, out, why);
    if (size >= 2 && bytes[0] == 0xFF && bytes[1] == 0xD8) return decode_jpeg(bytes, size, name, out, why);
    if (lower_extension(name) == ".tga") return decode_tga(bytes, size, name, out, why);
    return why = name + ": not an image format decoded here (PNG, JPEG, BMP, TGA, binary PPM / PGM; the reference uses stb_image)", false;
}

bool readImage(const std::string& path, Image& out, std::string& why) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return why = "failed to open " + path, false;
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return decodeImage(bytes.data(), bytes.size(), path, out, why);
}

// ---------------------------------------------------------------------------------------------------------------------
// Radiance RGBE (.hdr)
// ---------------------------------------------------------------------------------------------------------------------
namespace {
bool decode_hdr(const std::vector<uint8_t>& bytes, const std::string& name, uint32_t& width, uint32_t& height, std::vector<float>& out, std::string& why) {
    auto bad = [&](const std::string& m) { return why = name + ": " + m, false; };
    size_t pos = 0;
    auto line = [&]() {
        std::string l;
        while (pos < bytes.size() && bytes[pos] != '\n') l.push_back(char(bytes[pos++]));
        ++pos;
        return l;
    };
    const std::string magic = line();
    if (magic != "#?RADIANCE" && magic != "#?RGBE") return bad("not a Radiance .hdr file");
    bool format_ok = false;
    while (pos < bytes.size()) {
        const std::string l = line();
        if (l.empty()) break;
        if (l == "FORMAT=32-bit_rle_rgbe") format_ok = true;
    }
    if (!format_ok) return bad("unsupported .hdr format (32-bit_rle_rgbe is decoded)");
    const std::string res = line();
    long h = 0, w = 0;
    if (std::sscanf(res.c_str(), "-Y %ld +X %ld", &h, &w) != 2 || h <= 0 || w <= 0 || h > long(kMaxSide) || w > long(kMaxSide)) return bad("unsupported .hdr orientation / size");
    width = uint32_t(w), height = uint32_t(h);
    std::vector<uint8_t> rgbe(size_t(w) * size_t(h) * 4);
    const bool maybe_rle = w >= 8 && w < 32768;
    bool flat = !maybe_rle;
    for (long y = 0; y < h && !flat; ++y) {
        if (pos + 4 > bytes.size()) return bad("truncated .hdr");
        if (bytes[pos] != 2 || bytes[pos + 1] != 2 || (bytes[pos + 2] & 0x80)) {
            if (y != 0) return bad("mixed .hdr scanline encodings");
            flat = true;  // an old-style file: every pixel stored as it is
            break;
        }
        if (((long(bytes[pos + 2]) << 8) | bytes[pos + 3]) != w) return bad("bad .hdr scanline header");
        pos += 4;
        for (int c = 0; c < 4; ++c) {
            long x = 0;
            while (x < w) {
                if (pos >= bytes.size()) return bad("truncated .hdr");
                uint32_t count = bytes[pos++];
                if (count > 128) {  // a run
                    count -= 128;
                    if (x + long(count) > w || pos >= bytes.size()) return bad("bad .hdr run");
                    const uint8_t v = bytes[pos++];
                    for (uint32_t k = 0; k < count; ++k) rgbe[(size_t(y) * w + x++) * 4 + c] = v;
                } else {
                    if (count == 0 || x + long(count) > w || pos + count > bytes.size()) return bad("bad .hdr packet");
                    for (uint32_t k = 0; k < count; ++k) rgbe[(size_t(y) * w + x++) * 4 + c] = bytes[pos++];
                }
            }
        }
    }
    if (flat) {
        if (pos + rgbe.size() > bytes.size()) return bad("truncated .hdr");
        std::memcpy(rgbe.data(), bytes.data() + pos, rgbe.size());
    }
    out.resize(size_t(w) * size_t(h));
    for (size_t i = 0; i < out.size(); ++i) {
        const uint8_t* p = &rgbe[i * 4];
        out[i] = p[3] ? float(int(p[0]) + int(p[1]) + int(p[2])) * std::ldexp(1.0f, int(p[3]) - (128 + 8)) / 3.0f : 0.0f;
    }
    return true;
}
}  // namespace

bool readImageF32(const std::string& path, uint32_t& width, uint32_t& height, std::vector<float>& out, std::string& why) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return why = "failed to open " + path, false;
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (bytes.size() >= 2 && bytes[0] == '#' && bytes[1] == '?') return decode_hdr(bytes, path, width, height, out, why);
    Image img;
    if (!decodeImage(bytes.data(), bytes.size(), path, img, why)) return false;
    const std::vector<uint8_t> grey = convertChannels(img, 1);
    width = img.width, height = img.height;
    out.resize(grey.size());
    for (size_t i = 0; i < grey.size(); ++i) out[i] = float(std::pow(grey[i] / 255.0f, 2.2f));
    return true;
}

bool writeHDR(const std::string& path, const float* pixels, uint32_t width, uint32_t height, std::string& why) {
    if (!pixels || width == 0 || height == 0 || width > kMaxSide || height > kMaxSide) return why = "writeHDR: bad arguments", false;
    std::vector<uint8_t> file;
    const std::string header = "#?RADIANCE\n# Written by hiprz image_io\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n-Y " + std::to_string(height) + " +X " + std::to_string(width) + "\n";
    file.insert(file.end(), header.begin(), header.end());
    std::vector<uint8_t> row(size_t(width) * 4);
    for (uint32_t y = 0; y < height; ++y) {
        for (uint32_t x = 0; x < width; ++x) {
            const float v = pixels[size_t(y) * width + x];
            uint8_t* p = &row[size_t(x) * 4];
            if (!(v >= 1e-32f)) {
                p[0] = p[1] = p[2] = p[3] = 0;
            } else {
                int e;
                const float normalize = float(std::frexp(v, &e)) * 256.0f / v;
                p[0] = p[1] = p[2] = uint8_t(v * normalize), p[3] = uint8_t(e + 128);
            }
        }
        if (width < 8 || width >= 32768) {
            file.insert(file.end(), row.begin(), row.end());
            continue;
        }
        file.push_back(2), file.push_back(2), file.push_back(uint8_t(width >> 8)), file.push_back(uint8_t(width & 255));
        for (int c = 0; c < 4; ++c)  // new-style scanline: each component on its own, here as plain (non-run) packets of <= 128 bytes
            for (uint32_t x = 0; x < width; x += 128) {
                const uint32_t n = std::min(128u, width - x);
                file.push_back(uint8_t(n));
                for (uint32_t k = 0; k < n; ++k) file.push_back(row[size_t(x + k) * 4 + c]);
            }
    }
    std::ofstream f(path, std::ios::binary);
    if (!f.is_open()) return why = "failed to open " + path + " for writing", false;
    f.write(reinterpret_cast<const char*>(file.data()), std::streamsize(file.size()));
    return f.good() ? true : (why = "failed to write " + path, false);
}

std::vector<uint8_t> convertChannels(const Image& img, uint32_t channels) {
    const size_t n = size_t(img.width) * img.height;
    std::vector<uint8_t> out(n * channels);
    const uint32_t src = img.channels;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t* p = &img.data[i * src];
        const uint8_t r = p[0], g = src >= 3 ? p[1] : p[0], b = src >= 3 ? p[2] : p[0], a = src == 2 ? p[1] : src == 4 ? p[3] : 255;
        const uint8_t y = src >= 3 ? uint8_t((r * 77 + g * 150 + b * 29) >> 8) : p[0];
        uint8_t* o = &out[i * channels];
        if (channels == 1) o[0] = y;
        else if (channels == 2) o[0] = y, o[1] = a;
        else if (channels == 3) o[0] = r, o[1] = g, o[2] = b;
        else o[0] = r, o[1] = g, o[2] = b, o[3] = a;
    }
    return out;
}

bool writePNG(const std::string& path, const uint8_t* pixels, uint32_t width, uint32_t height, uint32_t channels, std::string& why) {
    if (!pixels || width == 0 || height == 0 || width > kMaxSide || height > kMaxSide || channels < 1 || channels > 4) return why = "writePNG: bad arguments", false;
    const size_t stride = size_t(width) * channels;
    std::vector<uint8_t> raw(size_t(height) * (stride + 1)), best(stride), trial(stride);
    const std::vector<uint8_t> zero(stride, 0);
    for (uint32_t y = 0; y < height; ++y) {
        const uint8_t* cur = pixels + size_t(y) * stride;
        const uint8_t* prev = y ? pixels + size_t(y - 1) * stride : zero.data();
        uint64_t best_cost = ~0ull;
        uint8_t best_type = 0;
        for (uint8_t type = 0; type < 5; ++type) {
            uint64_t cost = 0;
            for (size_t i = 0; i < stride; ++i) {
                const int a = i >= channels ? cur[i - channels] : 0, b = prev[i], c = i >= channels ? prev[i - channels] : 0;
                const int pred = type == 0 ? 0 : type == 1 ? a : type == 2 ? b : type == 3 ? (a + b) >> 1 : paeth(a, b, c);
                trial[i] = uint8_t(cur[i] - pred);
                cost += uint64_t(std::abs(int(int8_t(trial[i]))));
            }
            if (cost < best_cost) best_cost = cost, best_type = type, best.swap(trial);
        }
        raw[size_t(y) * (stride + 1)] = best_type;
        std::memcpy(&raw[size_t(y) * (stride + 1) + 1], best.data(), stride);
    }
    uLongf packed_size = compressBound(uLong(raw.size()));
    std::vector<uint8_t> packed(packed_size);
    if (compress2(packed.data(), &packed_size, raw.data(), uLong(raw.size()), 6) != Z_OK) return why = "writePNG: deflate failed", false;

    std::vector<uint8_t> file(kPngSignature, kPngSignature + 8);
    auto chunk = [&file](const char* type, const uint8_t* data, size_t len) {
        put_be32(file, uint32_t(len));
        const size_t start = file.size();
        file.insert(file.end(), type, type + 4);
        if (len) file.insert(file.end(), data, data + len);
        put_be32(file, uint32_t(crc32(crc32(0L, Z_NULL, 0), file.data() + start, uInt(len + 4))));
    };
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, width), put_be32(ihdr, height);
    static const uint8_t color_type[5] = {0, 0, 4, 2, 6};
    ihdr.push_back(8), ihdr.push_back(color_type[channels]), ihdr.push_back(0), ihdr.push_back(0), ihdr.push_back(0);
    chunk("IHDR", ihdr.data(), ihdr.size());
    chunk("IDAT", packed.data(), size_t(packed_size));
    chunk("IEND", nullptr, 0);
    std::ofstream f(path, std::ios::binary);
    if (!f.is_open()) return why = "failed to open " + path + " for writing", false;
    f.write(reinterpret_cast<const char*>(file.data()), std::streamsize(file.size()));
    return f.good() ? true : (why = "failed to write " + path, false);
}

}  // namespace RayZath::Hip::IO
