// scene_io.cpp — see scene_io.hpp.  Host-only C++17.
#include "scene_io.hpp"

#include "image_io.hpp"

#include "mini_json.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <limits>
#include <set>
#include <sstream>

namespace RayZath::Hip::IO {

namespace {

[[noreturn]] void fail(const std::string& m) { throw Exception(HIPRZ_ERR_INVALID, m); }

std::string trim(const std::string& s) {  // LoaderBase::trimSpaces
    size_t b = 0, e = s.size();
    while (b < e && std::isspace(static_cast<unsigned char>(s[b]))) ++b;
    while (e > b && std::isspace(static_cast<unsigned char>(s[e - 1]))) --e;
    return s.substr(b, e - b);
}
std::string rest_of_line(std::istringstream& in) {
    std::string r;
    std::getline(in, r);
    return trim(r);
}
std::string parent_dir(const std::string& path) {
    const size_t p = path.find_last_of("/\\");
    return p == std::string::npos ? std::string() : path.substr(0, p + 1);
}
std::string file_name(const std::string& path) {
    const size_t p = path.find_last_of("/\\");
    return p == std::string::npos ? path : path.substr(p + 1);
}
std::string extension(const std::string& path) {
    const std::string f = file_name(path);
    const size_t p = f.find_last_of('.');
    return p == std::string::npos ? std::string() : f.substr(p);
}
bool is_absolute(const std::string& p) { return !p.empty() && (p[0] == '/' || (p.size() > 1 && p[1] == ':')); }
std::string make_load_path(const std::string& p, const std::string& base_dir) { return is_absolute(p) ? p : base_dir + p; }
float clampf(float v, float lo, float hi) { return v < lo ? lo : v > hi ? hi : v; }

// ---- images (image_io.hpp: PNG, BMP, TGA, binary PPM / PGM) ----
// BitmapLoader::loadMap<...> (loader.cpp:36-98): RGBA8 (stbi_load(.., 4)) / RGBA8 with green negated / R8 (stbi_load(.., 1)) /
// R8 / R32F (stbi_loadf(.., 1): stb_image turns 8-bit data into floats with gamma 2.2 and scale 1)
std::shared_ptr<TextureBuffer> load_map(const std::string& path, uint32_t kind, bool normal_map, LoadLog& log) {
    auto t = std::make_shared<TextureBuffer>();
    t->kind = kind;
    std::string why;
    if (kind == HIPRZ_TEX_R32F) {  // stbi_loadf(.., 1): Radiance .hdr as it is, 8-bit formats through gamma 2.2
        std::vector<float> values;
        if (!readImageF32(path, t->width, t->height, values, why)) return log.error(why), nullptr;
        t->bitmap.resize(values.size() * 4);
        std::memcpy(t->bitmap.data(), values.data(), t->bitmap.size());
        return t;
    }
    Image img;
    if (!readImage(path, img, why)) {
        log.error(why);
        return nullptr;
    }
    t->width = img.width, t->height = img.height;
    const size_t n = size_t(img.width) * img.height;
    if (kind == HIPRZ_TEX_RGBA8) {
        t->bitmap = convertChannels(img, 4);
        if (normal_map)
            for (size_t i = 0; i < n; ++i) t->bitmap[i * 4 + 1] = uint8_t(-t->bitmap[i * 4 + 1]);  // loader.cpp:54-66
    } else {
        t->bitmap = convertChannels(img, 1);
    }
    return t;
}

float as_float(const Json& j) { return j.kind == Json::Float ? std::strtof(j.str.c_str(), nullptr) : float(j.num); }

vec3f to_vec3(const Json& j) {  // JsonTo<Math::vec3f>
    if (!j.is_array()) fail("Value is not an array.");
    if (j.items.size() != 3) fail("Array has to have three coordinates.");
    for (const auto& c : j.items)
        if (!c.is_number()) fail("Coordinates should be numbers.");
    return vec3f{as_float(j.items[0]), as_float(j.items[1]), as_float(j.items[2])};
}
void to_vec2(const Json& j, float out[2]) {
    if (!j.is_array()) fail("Value is not an array.");
    if (j.items.size() != 2) fail("Array has to have two coordinates.");
    for (const auto& c : j.items)
        if (!c.is_number()) fail("Coordinates should be numbers.");
    out[0] = as_float(j.items[0]), out[1] = as_float(j.items[1]);
}
Color to_color(const Json& j) {  // JsonTo<Graphics::Color>: floats are 0..1, integers 0..255, missing alpha = 255
    if (!j.is_array()) fail("Value is not an array.");
    if (j.items.size() < 3) fail("Color has at least three channels.");
    uint8_t v[4] = {0xF0, 0xF0, 0xF0, 0xFF};
    for (size_t i = 0; i < j.items.size() && i < 4; ++i) {
        const Json& c = j.items[i];
        if (!c.is_number()) fail("Color values should be numbers.");
        if (c.kind == Json::Float) v[i] = uint8_t(clampf(as_float(c), 0.0f, 1.0f) * 255.0f);
        else v[i] = uint8_t(std::min<uint32_t>(uint32_t(c.num < 0 ? 0 : c.num), 255u));
    }
    return Color{v[0], v[1], v[2], v[3]};
}

// Material::generateMaterial<Common::...> (material.cpp:93-198)
struct CommonMaterial {
    const char* statement;
    Color color;
    float metalness, roughness, emission, ior, scattering;
};
const CommonMaterial kCommonMaterials[] = {
    {"generate gold", {0xFF, 0xD7, 0x00, 0xFF}, 1.0f, 0.001f, 0.0f, 1.0f, 0.0f},
    {"generate silver", {0xC0, 0xC0, 0xC0, 0xFF}, 1.0f, 0.001f, 0.0f, 1.0f, 0.0f},
    {"generate copper", {0xB8, 0x73, 0x33, 0xFF}, 1.0f, 0.001f, 0.0f, 1.0f, 0.0f},
    {"generate glass", {0xFF, 0xFF, 0xFF, 0x00}, 0.0f, 0.0f, 0.0f, 1.45f, 0.0f},
    {"generate water", {0xFF, 0xFF, 0xFF, 0x00}, 0.0f, 0.0f, 0.0f, 1.33f, 0.0f},
    {"generate mirror", {0xF0, 0xF0, 0xF0, 0xFF}, 0.9f, 0.0f, 0.0f, 1.0f, 0.0f},
    {"generate rough wood", {0x96, 0x6F, 0x33, 0xFF}, 0.0f, 0.1f, 0.0f, 1.5f, 0.0f},
    {"generate polished wood", {0x96, 0x6F, 0x33, 0xFF}, 0.0f, 0.002f, 0.0f, 1.5f, 0.0f},
    {"generate paper", {0xFF, 0xFF, 0xFF, 0xFF}, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f},
    {"generate rubber", {0x00, 0x00, 0x00, 0xFF}, 0.0f, 0.018f, 0.0f, 1.3f, 0.0f},
    {"generate rough plastic", {0xFF, 0xFF, 0xFF, 0xFF}, 0.0f, 0.45f, 0.0f, 1.5f, 0.0f},
    {"generate polished plastic", {0xFF, 0xFF, 0xFF, 0xFF}, 0.0f, 0.0015f, 0.0f, 1.5f, 0.0f},
    {"generate porcelain", {0xFF, 0xFF, 0xFF, 0xFF}, 0.0f, 0.0f, 0.0f, 1.5f, 0.0f},
};

uint32_t create_normal(Mesh& m, float x, float y, float z) {
    m.normals.insert(m.normals.end(), {x, y, z});
    return uint32_t(m.normals.size() / 3 - 1);
}

}  // namespace

std::string LoadLog::str() const {
    std::string out;
    for (const auto& m : messages) out += "[message] " + m + "\n";
    for (const auto& m : warnings) out += "[warning] " + m + "\n";
    for (const auto& m : errors) out += "[error] " + m + "\n";
    return out;
}

// ---------------------------------------------------------------------------------------------------------
// procedural meshes
// ---------------------------------------------------------------------------------------------------------
std::shared_ptr<Mesh> generatePlane(uint32_t sides, float width, float height) {
    if (sides < 3) fail("plane needs at least three sides");
    auto m = std::make_shared<Mesh>();
    const float pi = 3.14159265358979323846f;
    const float delta = pi * 2.0f / float(sides), offset = delta * 0.5f;
    for (uint32_t i = 0; i < sides; ++i) {
        const float a = delta * float(i) + offset;
        const float s = std::sin(a), c = std::cos(a);
        const float px = 1.0f * c - 0.0f * s, py = 1.0f * s + 0.0f * c;  // vec2(1, 0).Rotate(a)
        m->createVertex(px * width, 0.0f, py * height);
        m->createTexcrd(px * 0.5f + 0.5f, py * 0.5f + 0.5f);
    }
    for (uint32_t i = 0; i + 2 < sides; ++i) m->createTriangle({0, i + 2, i + 1}, {0, i + 2, i + 1});
    return m;
}

std::shared_ptr<Mesh> generateSphere(uint32_t r, bool normals, bool texcrds) {
    if (r < 4) fail("sphere resolution must be at least 4");
    auto m = std::make_shared<Mesh>();
    const float pi = 3.14159265358979323846f, r_pi = float(1.0 / 3.14159265358979323846);
    const uint32_t half = r / 2;
    const float d_theta = pi / float(half), d_phi = 2.0f * pi / float(r);
    for (uint32_t t = 0; t + 1 < half; ++t)
        for (uint32_t p = 0; p < r; ++p) {  // (0,1,0).RotateX(theta).RotateY(phi)
            const float th = d_theta * (float(t) + 1.0f), ph = d_phi * float(p);
            const float y1 = std::cos(th), z1 = -std::sin(th);
            const float x2 = 0.0f * std::cos(ph) - z1 * std::sin(ph), z2 = 0.0f * std::sin(ph) + z1 * std::cos(ph);
            m->createVertex(x2, y1, z2);
        }
    const uint32_t ring = (half - 1) * r, top_v = ring, bottom_v = ring + 1;
    m->createVertex(0, 1, 0), m->createVertex(0, -1, 0);
    if (normals) m->normals = m->vertices;
    uint32_t top_t = 0, bottom_t = 0;
    if (texcrds) {
        for (uint32_t t = 0; t + 1 < half; ++t) {
            const float a_theta = d_theta * float(t + 1);
            for (uint32_t p = 0; p < r; ++p) m->createTexcrd((d_phi * float(p)) * 0.5f * r_pi, 1.0f - a_theta * r_pi);
            m->createTexcrd(1.0f, 1.0f - a_theta * r_pi);
        }
        top_t = uint32_t(m->texcrds.size() / 2);
        for (uint32_t p = 0; p < r; ++p) m->createTexcrd(float(p) / float(r) + 0.5f / float(r), 1.0f);
        bottom_t = top_t + r;
        for (uint32_t p = 0; p < r; ++p) m->createTexcrd(float(p) / float(r) + 0.5f / float(r), 0.0f);
    }
    const uint32_t n_uv = uint32_t(m->texcrds.size() / 2);
    const std::array<uint32_t, 3> unused{Mesh::ids_unused, Mesh::ids_unused, Mesh::ids_unused};
    auto tri = [&](std::array<uint32_t, 3> v, std::array<uint32_t, 3> t) {
        if (texcrds)
            for (auto& x : t) x = std::min(x, n_uv - 1u);  // the reference's bottom-fan ids reach below the last row for i = 0
        m->createTriangle(v, texcrds ? t : unused, normals ? v : unused);
    };
    for (uint32_t i = 0; i < r; ++i) {
        tri({top_v, (i + 1) % r, i}, {top_t + i, i + 1, i});
        tri({bottom_v, top_v - r + i, top_v - r + (i + 1) % r}, {bottom_t + i, top_t - r + i - 1u, top_t - r + i});
    }
    for (uint32_t t = 0; t + 2 < half; ++t)
        for (uint32_t p = 0; p < r; ++p) {
            tri({t * r + p, t * r + (p + 1) % r, (t + 1) * r + (p + 1) % r}, {t * (r + 1) + p, t * (r + 1) + p + 1, (t + 1) * (r + 1) + p + 1});
            tri({t * r + p, (t + 1) * r + (p + 1) % r, (t + 1) * r + p}, {t * (r + 1) + p, (t + 1) * (r + 1) + p + 1, (t + 1) * (r + 1) + p});
        }
    return m;
}

// Math::vec3 rotations as the reference's CUDA restatement spells them (cuda_render_parts.cuh:116-139; hiprz_host.cpp uses the same)
struct V3f {
    float x, y, z;
};
inline V3f rot_x(V3f v, float a) {
    const float s = std::sin(a), c = std::cos(a);
    return {v.x, v.y * c + v.z * s, v.y * -s + v.z * c};
}
inline V3f rot_y(V3f v, float a) {
    const float s = std::sin(a), c = std::cos(a);
    return {v.x * c - v.z * s, v.y, v.x * s + v.z * c};
}
inline V3f rot_z(V3f v, float a) {
    const float s = std::sin(a), c = std::cos(a);
    return {v.x * c + v.y * s, v.x * -s + v.y * c, v.z};
}
inline void push_normal(Mesh& m, V3f n) { m.normals.push_back(n.x), m.normals.push_back(n.y), m.normals.push_back(n.z); }

// World::generateMesh<CommonMesh::Cone> (world.cpp:342-397): unit base circle in y = 0, apex (0, 1, 0), two normals per side face
std::shared_ptr<Mesh> generateCone(uint32_t side_faces, bool normals) {
    if (side_faces < 3) fail("cone should have at least 3 side faces");
    auto m = std::make_shared<Mesh>();
    const float pi = 3.14159265358979323846f;
    const float delta_phi = pi * 2.0f / float(side_faces), offset_phi = delta_phi * 0.5f;
    for (uint32_t i = 0; i < side_faces; ++i) {
        const float angle = delta_phi * float(i) + offset_phi;
        m->createVertex(std::cos(angle), 0.0f, std::sin(angle));
    }
    const uint32_t apex = m->createVertex(0.0f, 1.0f, 0.0f);
    for (uint32_t i = 0; i < side_faces; ++i) {
        const float angle = delta_phi * float(i) + offset_phi;
        push_normal(*m, rot_y(rot_x(V3f{0.0f, 1.0f, 0.0f}, 0.25f * pi), angle + 0.5f * pi));
        push_normal(*m, rot_y(rot_x(V3f{0.0f, 1.0f, 0.0f}, 0.25f * pi), angle + 0.5f * pi + 0.5f * delta_phi));
    }
    const std::array<uint32_t, 3> unused{Mesh::ids_unused, Mesh::ids_unused, Mesh::ids_unused};
    for (uint32_t i = 0; i < side_faces; ++i)
        m->createTriangle({apex, (i + 1) % side_faces, i}, unused,
                          normals ? std::array<uint32_t, 3>{(i * 2 + 1) % (side_faces * 2), ((i + 1) * 2) % (side_faces * 2), i * 2} : unused);
    for (uint32_t i = 0; i + 2 < side_faces; ++i) m->createTriangle({0, i + 1, (i + 2) % side_faces});
    return m;
}

// World::generateMesh<CommonMesh::Cylinder> (world.cpp:399-479): radius 1, y in [-1, 1], one texcrd, one normal per side edge
std::shared_ptr<Mesh> generateCylinder(uint32_t faces, bool normals) {
    if (faces < 3) fail("cylinder should have at least 3 faces");
    auto m = std::make_shared<Mesh>();
    const float pi = 3.14159265358979323846f;
    const uint32_t n_vertices = faces * 2;
    m->createTexcrd(0.5f, 0.5f);
    const float delta_theta = pi * 2.0f / float(faces), offset_theta = delta_theta * 0.5f;
    for (uint32_t i = 0; i < faces; ++i) {
        const float angle = delta_theta * float(i) + offset_theta;
        m->createVertex(std::cos(angle), -1.0f, std::sin(angle));
        m->createVertex(std::cos(angle), +1.0f, std::sin(angle));
        if (normals) push_normal(*m, rot_y(V3f{1.0f, 0.0f, 0.0f}, angle));
    }
    auto v = [n_vertices](uint32_t idx) { return idx % n_vertices; };
    for (uint32_t i = 0; i + 2 < faces; ++i) {
        m->createTriangle({0, v((i + 1) * 2), v((i + 2) * 2)});              // bottom
        m->createTriangle({1, v((i + 2) * 2 + 1), v((i + 1) * 2 + 1)});      // top
    }
    for (uint32_t i = 0; i < faces; ++i) {                                    // side
        if (normals) {
            m->createTriangle({v(i * 2), v(i * 2 + 1), v((i + 1) * 2 + 1)}, {0, 0, 0}, {i, i, (i + 1) % faces});
            m->createTriangle({v(i * 2), v((i + 1) * 2 + 1), v((i + 1) * 2)}, {0, 0, 0}, {i, (i + 1) % faces, (i + 1) % faces});
        } else {
            m->createTriangle({v(i * 2), v(i * 2 + 1), v((i + 1) * 2 + 1)});
            m->createTriangle({v(i * 2), v((i + 1) * 2 + 1), v((i + 1) * 2)});
        }
    }
    return m;
}

// World::generateMesh<CommonMesh::Torus> (world.cpp:481-560): ring of radius `major_radius` in the xz plane, tube of `minor_radius`
std::shared_ptr<Mesh> generateTorus(uint32_t minor_resolution, uint32_t major_resolution, float minor_radius, float major_radius, bool normals, bool texcrds) {
    if (minor_resolution < 3 || major_resolution < 3) fail("resolution should be at least 3");
    auto m = std::make_shared<Mesh>();
    const float pi = 3.14159265358979323846f;
    const float d_phi = pi * 2.0f / float(major_resolution), offset_phi = d_phi * 0.5f, d_theta = pi * 2.0f / float(minor_resolution);
    for (uint32_t M = 0; M < major_resolution; ++M) {
        const float a_phi = d_phi * float(M) + offset_phi;
        for (uint32_t k = 0; k < minor_resolution; ++k) {
            const float a_theta = d_theta * float(k);
            const V3f center = rot_y(V3f{1.0f, 0.0f, 0.0f}, a_phi), normal = rot_y(rot_z(V3f{1.0f, 0.0f, 0.0f}, -a_theta), a_phi);
            m->createVertex(center.x * major_radius + normal.x * minor_radius, center.y * major_radius + normal.y * minor_radius,
                            center.z * major_radius + normal.z * minor_radius);
            if (normals) push_normal(*m, normal);
        }
    }
    if (texcrds)
        for (uint32_t M = 0; M <= major_resolution; ++M)
            for (uint32_t k = 0; k <= minor_resolution; ++k) m->createTexcrd(float(M) / float(major_resolution), float(k) / float(minor_resolution));
    const std::array<uint32_t, 3> unused{Mesh::ids_unused, Mesh::ids_unused, Mesh::ids_unused};
    const uint32_t r = minor_resolution, R = major_resolution;
    for (uint32_t M = 0; M < R; ++M)
        for (uint32_t k = 0; k < r; ++k) {
            const std::array<uint32_t, 3> v1{M * r + k, M * r + (k + 1) % r, ((M + 1) % R) * r + (k + 1) % r};
            const std::array<uint32_t, 3> t1{M * (r + 1) + k, M * (r + 1) + k + 1, (M + 1) * (r + 1) + k + 1};
            m->createTriangle(v1, texcrds ? t1 : unused, normals ? v1 : unused);
            const std::array<uint32_t, 3> v2{M * r + k, ((M + 1) % R) * r + (k + 1) % r, ((M + 1) % R) * r + k};
            const std::array<uint32_t, 3> t2{M * (r + 1) + k, (M + 1) * (r + 1) + k + 1, (M + 1) * (r + 1) + k};
            m->createTriangle(v2, texcrds ? t2 : unused, normals ? v2 : unused);
        }
    return m;
}

// ---------------------------------------------------------------------------------------------------------
// .mtl
// ---------------------------------------------------------------------------------------------------------
namespace {
struct MapDesc {
    std::string path;
    bool has_origin = false, has_scale = false;
    float origin[2] = {0, 0}, scale[2] = {1, 1};
};
// `map_Kd [-o u v] [-s u v] file`: a quoted string is the whole file name, otherwise the last token (loader.cpp:345-425)
MapDesc parse_map_statement(const std::string& statement, const std::string& where, LoadLog& log) {
    MapDesc map;
    if (statement.empty()) {
        log.error(where + "Map statement was empty (At least file name required).");
        return map;
    }
    std::istringstream params(statement);
    std::string p;
    while (params >> p) {
        if (p == "-o" || p == "-s") {
            float a, b;
            if (!(params >> a >> b)) {
                log.error(where + "option \"" + p + "\" needs two numbers");
                params.clear();
                continue;
            }
            if (p == "-o") map.has_origin = true, map.origin[0] = a, map.origin[1] = b;
            else map.has_scale = true, map.scale[0] = a, map.scale[1] = b;
        }
    }
    auto unescaped_quote = [&](size_t from) {
        for (size_t k = from; k < statement.size(); ++k)
            if (statement[k] == '"' && !(k > 0 && statement[k - 1] == '\\')) return k;
        return std::string::npos;
    };
    const size_t q0 = unescaped_quote(0);
    if (q0 != std::string::npos) {
        const size_t q1 = unescaped_quote(q0 + 1);
        if (q1 != std::string::npos) {
            map.path = statement.substr(q0 + 1, q1 - q0 - 1);
            return map;
        }
    }
    std::istringstream tokens(statement);
    while (tokens >> p) map.path = p;
    return map;
}
}  // namespace

std::vector<NamedMaterial> loadMTL(const std::string& path, LoadLog& log) {
    std::ifstream file(path);
    if (!file.is_open()) fail("cannot open " + path);
    const std::string dir = parent_dir(path);
    std::vector<NamedMaterial> out;
    std::set<std::string> unrecognized;
    uint32_t line_number = 0;
    auto number = [&](std::istringstream& in, float& v) { return bool(in >> v); };
    for (std::string raw; std::getline(file, raw); ++line_number) {
        const std::string line = trim(raw);
        if (line.empty()) continue;
        std::istringstream in(line);
        std::string statement;
        in >> statement;
        const std::string where = path + ':' + std::to_string(line_number) + ": ";
        if (statement == "#" || statement[0] == '#') continue;
        if (statement == "newmtl") {
            NamedMaterial m;
            m.name = rest_of_line(in);
            m.material = std::make_shared<Material>();  // ConStruct<Material>{}: LightGrey, 0, 0, 0, ior 1.5, 0 (material.hpp:137-150)
            out.push_back(std::move(m));
            continue;
        }
        if (out.empty()) {
            log.warning("statement before the first newmtl skipped");
            continue;
        }
        Material& mat = *out.back().material;
        auto attach = [&](std::shared_ptr<TextureBuffer>& slot, uint32_t kind, bool normal_map) {
            const MapDesc d = parse_map_statement(rest_of_line(in), where, log);
            if (d.path.empty()) return;
            auto t = load_map(make_load_path(d.path, dir), kind, normal_map, log);
            if (!t) return;
            if (d.has_origin) t->translation[0] = d.origin[0], t->translation[1] = d.origin[1];
            if (d.has_scale) t->scale[0] = d.scale[0], t->scale[1] = d.scale[1];
            slot = std::move(t);
        };
        float v = 0;
        if (statement == "Kd") {
            float c[3] = {0, 0, 0};
            if (!number(in, c[0])) {
                log.error(where + "Kd needs one or three numbers in [0, 1]");
                continue;
            }
            if (!number(in, c[1])) {
                c[1] = c[2] = c[0];
            } else if (!number(in, c[2])) {
                log.error(where + "Kd: the third (blue) component is not a number in [0, 1]");
                continue;
            }
            for (float& x : c) x = clampf(x, 0.0f, 1.0f);
            mat.color.red = uint8_t(c[0] * 255.0f), mat.color.green = uint8_t(c[1] * 255.0f), mat.color.blue = uint8_t(c[2] * 255.0f);
        } else if (statement == "Ns") {
            if (!number(in, v)) {
                log.error(where + "Ns needs a number in [1, 1000]");
                continue;
            }
            const float clamped = clampf(v, 1.0f, 1000.0f);
            if (clamped != v) log.warning(where + "Value " + std::to_string(v) + " is outside of [1.0, 1000.0] range. Clamped.");
            mat.roughness(1.0f - (std::log10(clamped) / std::log10(1000.0f)));
        } else if (statement == "d" || statement == "Tr") {
            if (!number(in, v)) {
                log.error(where + statement + " needs a number in [0, 1]");
                continue;
            }
            const float clamped = clampf(v, 0.0f, 1.0f);
            if (clamped != v) log.warning(where + std::to_string(v) + " clamped to [0, 1]");
            mat.color.alpha = uint8_t((statement == "d" ? clamped : 1.0f - clamped) * 255.0f);
        } else if (statement == "Ni") {
            if (!number(in, v)) {
                log.error(where + "Ni needs a number >= 1");
                continue;
            }
            if (v < 1.0f) log.warning(where + "Ni below 1 raised to 1");
            mat.ior(v);
        } else if (statement == "Pm" || statement == "Pr") {
            if (!number(in, v)) {
                log.error(where + statement + " needs a number in [0, 1]");
                continue;
            }
            const float clamped = clampf(v, 0.0f, 1.0f);
            if (clamped != v) log.warning(where + statement + " clamped to [0, 1]");
            if (statement == "Pm") mat.metalness(clamped);
            else mat.roughness(clamped);
        } else if (statement == "Ke") {
            if (!number(in, v)) {
                log.error(where + "Ke needs a number >= 0");
                continue;
            }
            if (v < 0.0f) log.warning(where + "Value for \"Ke\" is less than 0.0. Clamped.");
            mat.emission(v);
        } else if (statement == "map_Kd") {
            attach(mat.texture, HIPRZ_TEX_RGBA8, false);
        } else if (statement == "norm") {
            attach(mat.normal_map, HIPRZ_TEX_RGBA8, true);
        } else if (statement == "map_Pm") {
            attach(mat.metalness_map, HIPRZ_TEX_R8, false);
        } else if (statement == "map_Pr") {
            attach(mat.roughness_map, HIPRZ_TEX_R8, false);
        } else if (statement == "map_Ke") {
            attach(mat.emission_map, HIPRZ_TEX_R32F, false);
        } else if (unrecognized.insert(statement).second) {
            log.warning("unknown statement '" + statement + "' skipped");
        }
    }
    return out;
}

// ---------------------------------------------------------------------------------------------------------
// .obj
// ---------------------------------------------------------------------------------------------------------
ObjFile parseOBJ(const std::string& path, LoadLog& log) {
    std::ifstream file(path);
    if (!file.is_open()) fail("cannot open " + path);
    constexpr uint32_t npos = Mesh::ids_unused;
    ObjFile result;
    std::vector<std::array<float, 3>> vertices, normals;
    std::vector<std::array<float, 2>> texcrds;
    uint32_t material_count = 0, material_idx = 0;
    uint32_t vr[2] = {npos, 0}, tr[2] = {npos, 0}, nr[2] = {npos, 0};  // [min, max) of the indices the current mesh uses

    // a mesh owns the contiguous sub-range of the file's components its faces touch (loader.cpp:755-778)
    auto finish_mesh = [&](Mesh& mesh) {
        if (vr[0] == npos) vr[0] = 0;
        if (tr[0] == npos) tr[0] = 0;
        if (nr[0] == npos) nr[0] = 0;
        for (uint32_t i = vr[0]; i < vr[1]; ++i) mesh.createVertex(vertices[i][0], vertices[i][1], vertices[i][2]);
        for (uint32_t i = tr[0]; i < tr[1]; ++i) mesh.createTexcrd(texcrds[i][0], texcrds[i][1]);
        for (uint32_t i = nr[0]; i < nr[1]; ++i) create_normal(mesh, normals[i][0], normals[i][1], normals[i][2]);
        for (auto& x : mesh.tri_vertices)
            if (x != npos) x -= vr[0];
        for (auto& x : mesh.tri_texcrds)
            if (x != npos) x -= tr[0];  // (the reference also shifts unused ids, which then stop being "unused")
        for (auto& x : mesh.tri_normals)
            if (x != npos) x -= nr[0];
    };

    std::set<std::string> unrecognized;
    uint32_t line_number = 0;
    for (std::string raw; std::getline(file, raw); ++line_number) {
        const std::string line = trim(raw);
        if (line.empty()) continue;
        std::istringstream in(line);
        std::string statement;
        in >> statement;
        const std::string ln = std::to_string(line_number);
        if (statement[0] == '#') continue;
        if (statement == "mtllib") {
            const std::string lib = rest_of_line(in);
            if (std::find(result.mtllibs.begin(), result.mtllibs.end(), lib) == result.mtllibs.end()) result.mtllibs.push_back(lib);
            continue;
        }
        if (statement == "v") {
            std::array<float, 3> v{};
            if (!(in >> v[0] >> v[1] >> v[2])) log.error("line " + ln + ": v needs three numbers");
            v[2] = -v[2];  // right-handed -> left-handed
            vertices.push_back(v);
            continue;
        }
        if (statement == "vt") {
            std::array<float, 2> t{};
            if (!(in >> t[0] >> t[1])) log.error("line " + ln + ": vt needs two numbers");
            texcrds.push_back(t);
            continue;
        }
        if (statement == "vn") {
            std::array<float, 3> n{};
            if (!(in >> n[0] >> n[1] >> n[2])) log.error("line " + ln + ": vn needs three numbers");
            n[2] = -n[2];
            if (std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) < std::numeric_limits<float>::epsilon()) {
                log.warning("Line " + ln + ": zero-length normal");
                n = {0.0f, 1.0f, 0.0f};
            }
            normals.push_back(n);
            continue;
        }
        if (statement == "o" || statement == "g") {
            if (!result.meshes.empty()) finish_mesh(*result.meshes.back().mesh);
            ObjMesh m;
            m.name = rest_of_line(in);
            m.mesh = std::make_shared<Mesh>();
            result.meshes.push_back(std::move(m));
            material_count = material_idx = 0;
            vr[0] = tr[0] = nr[0] = npos, vr[1] = tr[1] = nr[1] = 0;
            continue;
        }
        if (result.meshes.empty()) {
            log.warning("line " + ln + ": no o / g statement yet, skipped");
            continue;
        }
        ObjMesh& cur = result.meshes.back();
        if (statement == "usemtl") {
            const std::string name = rest_of_line(in);
            const auto it = cur.material_ids.find(name);
            if (it != cur.material_ids.end()) {
                material_idx = it->second;
            } else if (material_count == Instance::materialCapacity()) {
                log.warning("The declaration of usage of material \"" + name + "\" on line " + ln + " reached the limit of 64 materials per object. Ignored.");
            } else {
                material_idx = material_count;
                cur.material_ids[name] = material_count++;
            }
        } else if (statement == "f") {
            constexpr size_t max_n_gon = 8;
            int32_t idx[max_n_gon][3] = {};
            uint32_t corners = 0;
            std::string buff;
            while (corners < max_n_gon && in >> buff) {
                size_t begin = 0;
                for (int k = 0; k < 3; ++k) {
                    const size_t end = buff.find('/', begin);
                    const std::string part = buff.substr(begin, end == std::string::npos ? std::string::npos : end - begin);
                    if (!part.empty()) {
                        char* stop = nullptr;
                        const long v = std::strtol(part.c_str(), &stop, 10);
                        if (stop == part.c_str() || *stop != '\0') log.error("line " + ln + ": corner " + std::to_string(corners) + " of the face has a malformed index");
                        else idx[corners][k] = int32_t(v);
                    }
                    if (end == std::string::npos) break;
                    begin = end + 1;
                }
                ++corners;
            }
            if (corners < 3) {
                log.error("line " + ln + ": a face needs at least three corners");
                continue;
            }
            uint32_t triplet[max_n_gon][3];
            const size_t sizes[3] = {vertices.size(), texcrds.size(), normals.size()};
            const char* what[3] = {"vertex index", "texture coordinate index", "normal index"};
            for (uint32_t c = 0; c < corners; ++c)
                for (int k = 0; k < 3; ++k) {
                    const int32_t v = idx[c][k];
                    if (v > 0 && size_t(v) <= sizes[k]) triplet[c][k] = uint32_t(v - 1);
                    else if (v < 0 && size_t(-int64_t(v)) <= sizes[k]) triplet[c][k] = uint32_t(sizes[k] - size_t(-int64_t(v)));
                    else {
                        triplet[c][k] = npos;
                        if (v != 0) log.error("On line " + ln + ": " + what[k] + " outside of range.");
                    }
                }
            uint32_t* ranges[3] = {vr, tr, nr};
            for (uint32_t c = 0; c < corners; ++c)
                for (int k = 0; k < 3; ++k)
                    if (triplet[c][k] != npos) {
                        ranges[k][0] = std::min(ranges[k][0], triplet[c][k]);
                        ranges[k][1] = std::max(ranges[k][1], triplet[c][k] + 1u);
                    }
            for (uint32_t i = 0; i + 2 < corners; ++i)  // fan with swapped winding (loader.cpp:1008-1016)
                cur.mesh->createTriangle({triplet[0][0], triplet[i + 2][0], triplet[i + 1][0]}, {triplet[0][1], triplet[i + 2][1], triplet[i + 1][1]},
                                         {triplet[0][2], triplet[i + 2][2], triplet[i + 1][2]}, material_idx);
        } else if (unrecognized.insert(statement).second) {
            log.warning("unknown statement '" + statement + "' skipped");
        }
    }
    if (!result.meshes.empty()) finish_mesh(*result.meshes.back().mesh);
    return result;
}

std::vector<std::shared_ptr<Instance>> loadObjInstances(const std::string& path, World& world, LoadLog& log) {
    if (extension(path) != ".obj") fail(path + ": not an .obj file");
    ObjFile obj = parseOBJ(path, log);
    std::map<std::string, std::shared_ptr<Material>> materials;
    for (const auto& lib : obj.mtllibs) {
        try {
            for (auto& m : loadMTL(make_load_path(lib, parent_dir(path)), log)) {
                if (!materials.emplace(m.name, m.material).second)
                    log.error(path + ": its material libraries define '" + m.name + "' more than once");
                else world.materials.push_back(m.material);
            }
        } catch (const Exception& e) {
            log.error(e.what());
        }
    }
    std::vector<std::shared_ptr<Instance>> instances;
    for (const auto& om : obj.meshes) {
        auto inst = std::make_shared<Instance>();
        inst->mesh = om.mesh;
        world.meshes.push_back(om.mesh);
        for (const auto& [name, slot] : om.material_ids) {
            const auto it = materials.find(name);
            if (it == materials.end()) log.error("Failed to obtain \"" + name + "\" material.");
            else inst->materials[slot] = it->second;
        }
        world.instances.push_back(inst);
        instances.push_back(inst);
    }
    world.makeModified();
    return instances;
}

// ---------------------------------------------------------------------------------------------------------
// .json scene
// ---------------------------------------------------------------------------------------------------------
namespace {
struct SceneLoader {
    World& world;
    LoadLog& log;
    std::string dir;
    std::map<std::string, std::shared_ptr<Material>> materials;
    std::map<std::string, std::shared_ptr<Mesh>> meshes;
    std::map<std::string, std::shared_ptr<TextureBuffer>> maps[5];  // Texture, NormalMap, MetalnessMap, RoughnessMap, EmissionMap
    bool have_camera = false;
    std::map<std::string, std::shared_ptr<Instance>> instances;  // by name, for the groups

    // json_loader.cpp:75-163: a map object (or the name of one loaded before)
    std::shared_ptr<TextureBuffer> load_texture(const Json& j, int which) {
        static const uint32_t kinds[5] = {HIPRZ_TEX_RGBA8, HIPRZ_TEX_RGBA8, HIPRZ_TEX_R8, HIPRZ_TEX_R8, HIPRZ_TEX_R32F};
        if (j.is_string()) {
            const auto it = maps[which].find(j.str);
            if (it == maps[which].end()) return log.error("no map named '" + j.str + "' has been loaded so far"), nullptr;
            return it->second;
        }
        if (!j.is_object()) return log.error("a map is given by name or as an object"), nullptr;
        std::shared_ptr<TextureBuffer> t;
        if (const Json* f = j.find("file"); f && f->is_string()) t = load_map(make_load_path(f->str, dir), kinds[which], which == 1, log);
        if (!t) return nullptr;
        for (const auto& [key, value] : j.members) {
            if (key == "scale") to_vec2(value, t->scale);
            else if (key == "rotation" && value.is_number()) t->rotation = as_float(value);
            else if (key == "translation") to_vec2(value, t->translation);
            // "filter mode" / "address mode" (json_loader.cpp:120-150): kept in the record; only the CUDA-compat mode reads them — the CPU
            // kernel point-samples with wrap-around whatever they say (render_parts.hpp:209-221)
            else if (key == "filter mode" && value.is_string()) t->sampling = (t->sampling & ~0xFFu) | (value.str == "linear" ? HIPRZ_TEX_FILTER_LINEAR : HIPRZ_TEX_FILTER_POINT);
            else if (key == "address mode" && value.is_string())
                t->sampling = (t->sampling & 0xFFu) | (value.str == "clamp" ? HIPRZ_TEX_ADDRESS_CLAMP : value.str == "mirror" ? HIPRZ_TEX_ADDRESS_MIRROR : value.str == "border" ? HIPRZ_TEX_ADDRESS_BORDER : HIPRZ_TEX_ADDRESS_WRAP);
        }
        if (const Json* n = j.find("name"); n && n->is_string()) maps[which][n->str] = t;
        return t;
    }

    void generate_material(const Json& j, Material& m) {  // json_loader.cpp:327-392
        for (const auto& c : kCommonMaterials)
            if (j.contains(c.statement)) {
                m.color = c.color;
                m.metalness(c.metalness), m.roughness(c.roughness), m.emission(c.emission), m.ior(c.ior), m.scattering(c.scattering);
                return;
            }
    }
    void do_load_material(const Json& j, Material& m) {  // json_loader.cpp:282-326
        const float inf = std::numeric_limits<float>::infinity();
        for (const auto& [key, value] : j.members) {
            if (key == "color") m.color = to_color(value);
            else if (key == "metalness" && value.is_number()) m.metalness(clampf(as_float(value), 0.0f, 1.0f));
            else if (key == "roughness" && value.is_number()) m.roughness(clampf(as_float(value), 0.0f, 1.0f));
            else if (key == "emission" && value.is_number()) m.emission(clampf(as_float(value), 0.0f, inf));
            else if (key == "ior" && value.is_number()) m.ior(clampf(as_float(value), 1.0f, inf));
            else if (key == "scattering" && value.is_number()) m.scattering(clampf(as_float(value), 0.0f, inf));
            else if (key == "texture") m.texture = load_texture(value, 0);
            else if (key == "normal map") m.normal_map = load_texture(value, 1);
            else if (key == "metalness map") m.metalness_map = load_texture(value, 2);
            else if (key == "roughness map") m.roughness_map = load_texture(value, 3);
            else if (key == "emission map") m.emission_map = load_texture(value, 4);
        }
    }
    // the world / default material: generate statement, then a whole .mtl, then the json properties (json_loader.cpp:252-281)
    void load_into(const Json& j, Material& m) {
        if (!j.is_object()) return log.error("a material is given by name or as an object");
        generate_material(j, m);
        if (const Json* f = j.find("file")) {
            if (!f->is_string()) log.error("\"file\" must be a string");
            else if (auto loaded = loadMTL(make_load_path(f->str, dir), log); !loaded.empty()) m = *loaded.front().material;
        }
        do_load_material(j, m);
    }
    std::shared_ptr<Material> load_material(const Json& j) {  // json_loader.cpp:190-251
        if (j.is_string()) {
            const auto it = materials.find(j.str);
            if (it == materials.end()) return log.error("no material named '" + j.str + "' has been loaded so far"), nullptr;
            return it->second;
        }
        if (!j.is_object()) return log.error("a material is given by name or as an object"), nullptr;
        std::shared_ptr<Material> m;
        std::string name = "material name";
        if (const Json* f = j.find("file")) {
            if (!f->is_string()) {
                log.error("\"file\" must be a string");
            } else {
                auto loaded = loadMTL(make_load_path(f->str, dir), log);
                if (loaded.size() != 1) log.warning(f->str + " holds " + std::to_string(loaded.size()) + " materials where one is wanted");
                else m = loaded.front().material, name = loaded.front().name;
            }
        }
        if (!m) m = std::make_shared<Material>();  // ConStruct<Material>{} (material.hpp:137-150)
        if (const Json* n = j.find("name"); n && n->is_string()) name = n->str;
        do_load_material(j, *m);
        world.materials.push_back(m);
        if (!materials.emplace(name, m).second) log.warning("Loading material with ambigous name \"" + name + "\".");
        log.message("Loaded material \"" + name + "\".");
        return m;
    }

    std::shared_ptr<Mesh> generate_mesh(const Json& j) {  // json_loader.cpp:394-537
        static const char* statements[] = {"generate cube", "generate plane", "generate sphere", "generate cone", "generate cylinder", "generate torus"};
        const char* which = nullptr;
        for (const char* s : statements)
            if (const Json* g = j.find(s)) {
                if (!g->is_object()) return log.error(std::string("\"") + s + "\" takes an object of parameters"), nullptr;
                which = s;
            }
        if (!which) return nullptr;
        const Json& g = *j.find(which);
        auto resolution = [](const Json& v) { return uint32_t(std::max(as_float(v), 3.0f)); };
        if (!std::strcmp(which, "generate cube")) return Mesh::generateCube();
        if (!std::strcmp(which, "generate plane")) {
            uint32_t sides = 4;
            float width = 1.0f, height = 1.0f;  // CommonMeshParameters<Plane> (world.hpp)
            for (const auto& [key, value] : g.members) {
                if (key == "resolution" && value.is_number()) sides = resolution(value);
                if (key == "width" && value.is_number()) width = as_float(value);
                if (key == "height" && value.is_number()) height = as_float(value);
            }
            return generatePlane(sides, width, height);
        }
        if (!std::strcmp(which, "generate sphere")) {
            uint32_t res = 16;
            bool normals = true, texcrds = true;
            for (const auto& [key, value] : g.members) {
                if (key == "resolution" && value.is_number()) res = resolution(value);
                if (key == "normals" && value.is_bool()) normals = value.b;
                if (key == "texcrds" && value.is_bool()) texcrds = value.b;
                if (key == "type" && value.is_string() && value.str != "uvsphere") fail(value.str == "icosphere" ? "icosphere is declared but not implemented by the reference (world.cpp:336)" : "unknown sphere type '" + value.str + "'");
            }
            return generateSphere(std::max(res, 4u), normals, texcrds);
        }
        if (!std::strcmp(which, "generate cone")) {
            uint32_t faces = 16;
            bool normals = true;  // CommonMeshParameters<Cone> (world.hpp); "texcrds" is parsed by the reference and has no effect
            for (const auto& [key, value] : g.members) {
                if (key == "resolution" && value.is_number()) faces = resolution(value);
                if (key == "normals" && value.is_bool()) normals = value.b;
            }
            return generateCone(faces, normals);
        }
        if (!std::strcmp(which, "generate cylinder")) {
            uint32_t faces = 16;
            bool normals = true;
            for (const auto& [key, value] : g.members) {
                if (key == "resolution" && value.is_number()) faces = resolution(value);
                if (key == "normals" && value.is_bool()) normals = value.b;
            }
            return generateCylinder(faces, normals);
        }
        uint32_t minor_res = 16, major_res = 32;  // "generate torus": CommonMeshParameters<Torus>
        float minor_radius = 0.25f, major_radius = 1.0f;
        bool normals = true, texcrds = true;
        for (const auto& [key, value] : g.members) {
            if (key == "minor resolution" && value.is_number()) minor_res = resolution(value);
            if (key == "major resolution" && value.is_number()) major_res = resolution(value);
            if (key == "minor radious" && value.is_number()) minor_radius = std::max(as_float(value), 0.0f);
            if (key == "major radious" && value.is_number()) major_radius = std::max(as_float(value), 0.0f);
            if (key == "normals" && value.is_bool()) normals = value.b;
            if (key == "texcrds" && value.is_bool()) texcrds = value.b;
        }
        return generateTorus(minor_res, major_res, minor_radius, major_radius, normals, texcrds);
    }
    std::shared_ptr<Mesh> load_mesh(const Json& j) {  // json_loader.cpp:538-662
        if (j.is_string()) {
            const auto it = meshes.find(j.str);
            if (it == meshes.end()) return log.error("no mesh named '" + j.str + "' has been loaded so far"), nullptr;
            return it->second;
        }
        if (!j.is_object()) return log.error("a mesh is given by name or as an object"), nullptr;
        if (!j.contains("name") && !j.contains("file")) return log.error("an inline mesh needs a \"name\""), nullptr;
        std::string name = "default";
        if (const Json* n = j.find("name"); n && n->is_string()) name = n->str;
        auto publish = [&](std::shared_ptr<Mesh> m, const std::string& as) {
            world.meshes.push_back(m);
            if (!meshes.emplace(as, m).second) log.warning("Loading mesh with ambigous name \"" + as + "\".");
            log.message("Loaded mesh \"" + as + "\".");
            return m;
        };
        if (auto m = generate_mesh(j)) return publish(m, name);
        if (const Json* f = j.find("file")) {
            if (!f->is_string()) {
                log.error("File name has to be a string.");
            } else {
                ObjFile obj = parseOBJ(make_load_path(f->str, dir), log);
                if (obj.meshes.size() != 1) log.warning(f->str + " holds " + std::to_string(obj.meshes.size()) + " meshes where one is wanted");
                if (obj.meshes.empty()) fail("no mesh loaded from " + f->str);
                return publish(obj.meshes.front().mesh, obj.meshes.front().name);
            }
        }
        auto m = std::make_shared<Mesh>();
        for (const auto& [key, value] : j.members) {
            if (!value.is_array()) continue;
            if (key == "vertices")
                for (const auto& v : value.items) {
                    const vec3f p = to_vec3(v);
                    m->createVertex(p.x, p.y, p.z);
                }
            else if (key == "texcrds")
                for (const auto& v : value.items) {
                    float t[2];
                    to_vec2(v, t);
                    m->createTexcrd(t[0], t[1]);
                }
            else if (key == "normals")
                for (const auto& v : value.items) {
                    const vec3f p = to_vec3(v);
                    create_normal(*m, p.x, p.y, p.z);
                }
        }
        for (const auto& [key, value] : j.members) {
            if (key != "triangles" || !value.is_array()) continue;
            for (const auto& t : value.items) {
                if (!t.is_object()) continue;
                std::array<uint32_t, 3> ids[3];
                for (auto& a : ids) a = {Mesh::ids_unused, Mesh::ids_unused, Mesh::ids_unused};
                uint32_t material_idx = 0;
                for (const auto& [ck, cv] : t.members) {
                    const int k = ck == "v" ? 0 : ck == "t" ? 1 : ck == "n" ? 2 : -1;
                    if (k >= 0) {
                        if (!cv.is_array() || cv.items.size() != 3) fail("triangle component \"" + ck + "\" needs three indices");
                        for (int c = 0; c < 3; ++c) ids[k][c] = uint32_t(cv.items[c].num);
                    } else if (ck == "m" && cv.kind == Json::Int) {
                        material_idx = uint32_t(cv.num);
                    }
                }
                m->createTriangle(ids[0], ids[1], ids[2], material_idx);
            }
        }
        return publish(m, name);
    }

    void load_camera(const Json& j) {  // json_loader.cpp:664-711
        if (!j.is_object()) return log.error("Value of camera definition has to be an object.");
        Camera c;
        bool enabled = true;
        std::string name = "name";
        for (const auto& [key, value] : j.members) {
            if (key == "name" && value.is_string()) name = value.str;
            else if (key == "position") c.position = to_vec3(value);
            else if (key == "rotation") c.rotation = to_vec3(value);
            else if (key == "resolution") {
                float r[2];
                to_vec2(value, r);
                c.width = uint32_t(r[0]), c.height = uint32_t(r[1]);
            } else if (key == "fov" && value.is_number()) c.fov = as_float(value);
            else if (key == "near plane" && value.is_number()) c.near_plane = as_float(value);
            else if (key == "far plane" && value.is_number()) c.far_plane = as_float(value);
            else if (key == "near far") {
                float nf[2];
                to_vec2(value, nf);
                c.near_plane = nf[0], c.far_plane = nf[1];
            } else if (key == "focal distance" && value.is_number()) c.focal_distance = as_float(value);
            else if (key == "aperture" && value.is_number()) c.aperture = as_float(value);
            else if (key == "exposure time" && value.is_number()) c.exposure_time = as_float(value);
            else if (key == "enabled" && value.is_bool()) enabled = value.b;
            else if (key == "temporal blend" && value.is_number()) c.temporal_blend = std::min(std::max(as_float(value), 0.0f), 1.0f);  // json_loader.cpp:701, camera.cpp:154-156
        }
        log.message("Loaded camera \"" + name + "\".");
        c.enabled = enabled;
        if (enabled && !have_camera) world.camera = c, have_camera = true;  // World::camera: the first enabled one ...
        else world.cameras.push_back(std::make_shared<Camera>(c));          // ... the others follow; the renderer skips disabled ones
    }
    // json_loader.cpp:886-1033: groups carry a transformation over the instances (and sub-groups) they list.  How it reaches the
    // flattened scene is the World's group_transforms setting (hip_engine.hpp): like the CPU engine by default.
    void load_groups(const Json& objects) {
        const Json* groups = objects.find("Group");
        if (!groups) return;
        std::vector<std::pair<std::shared_ptr<Group>, const Json*>> loaded;
        std::map<std::string, std::shared_ptr<Group>> by_name;
        auto load_group = [&](const Json& j) {
            if (!j.is_object()) return log.error("a group is an object");
            auto g = std::make_shared<Group>();
            std::string name = "name";
            for (const auto& [key, value] : j.members) {
                if (key == "name" && value.is_string()) name = value.str;
                else if (key == "position") g->position = to_vec3(value);
                else if (key == "rotation") g->rotation = to_vec3(value);
                else if (key == "scale") g->scale = to_vec3(value);
            }
            if (by_name.count(name)) return log.error("group '" + name + "' is defined twice");
            by_name[name] = g;
            world.groups.push_back(g);
            loaded.emplace_back(g, &j);
            if (const Json* members = j.find("objects")) {
                if (!members->is_array()) return log.error("group '" + name + "': \"objects\" is a list of instance names");
                for (const Json& m : members->items) {
                    if (!m.is_string()) {
                        log.error("group '" + name + "': an object is named by a string");
                        continue;
                    }
                    const auto it = instances.find(m.str);
                    if (it == instances.end()) log.error("group '" + name + "': no instance named '" + m.str + "'");
                    else it->second->group = g;
                }
            }
            log.message("Loaded group \"" + name + "\".");
        };
        if (groups->is_object()) return load_group(*groups);
        if (!groups->is_array()) return;
        for (const Json& j : groups->items) load_group(j);
        for (const auto& [g, j] : loaded) {  // sub-groups, once every group exists
            const Json* subs = j->find("groups");
            if (!subs) continue;
            if (!subs->is_array()) {
                log.error("\"groups\" is a list of group names");
                continue;
            }
            for (const Json& sname : subs->items) {
                const auto it = sname.is_string() ? by_name.find(sname.str) : by_name.end();
                if (it == by_name.end()) {
                    log.error("a sub-group names a group that does not exist");
                    continue;
                }
                bool circular = it->second == g;
                for (const Group* up = g->group.get(); up && !circular; up = up->group.get()) circular = up == it->second.get();
                if (circular) log.error("groups must not contain themselves: '" + sname.str + "' is skipped");
                else it->second->group = g;
            }
        }
    }
    void load_spot_light(const Json& j) {  // json_loader.cpp:713-748
        if (!j.is_object()) return log.error("Value of spot light definition has to be an object.");
        auto l = std::make_shared<SpotLight>();
        for (const auto& [key, value] : j.members) {
            if (key == "position") l->position = to_vec3(value);
            else if (key == "direction") l->direction = to_vec3(value);
            else if (key == "color") l->color = to_color(value);
            else if (key == "size" && value.is_number()) l->size = as_float(value);
            else if (key == "emission" && value.is_number()) l->emission = as_float(value);
            else if (key == "angle" && value.is_number()) l->beam_angle = as_float(value);
        }
        world.spot_lights.push_back(l);
    }
    void load_direct_light(const Json& j) {  // json_loader.cpp:749-780
        if (!j.is_object()) return log.error("Value of direct light definition has to be an object.");
        auto l = std::make_shared<DirectLight>();
        for (const auto& [key, value] : j.members) {
            if (key == "direction") l->direction = to_vec3(value);
            else if (key == "color") l->color = to_color(value);
            else if (key == "emission" && value.is_number()) l->emission = as_float(value);
            else if (key == "size" && value.is_number()) l->angular_size = as_float(value);
        }
        world.direct_lights.push_back(l);
    }
    void load_instance(const Json& j) {  // json_loader.cpp:782-885
        if (!j.is_object()) return log.error("Value of instance definition has to be an object.");
        std::shared_ptr<Instance> inst;
        if (const Json* f = j.find("file")) {
            if (!f->is_string()) return log.error("instance: \"file\" must be a string");
            auto loaded = loadObjInstances(make_load_path(f->str, dir), world, log);
            if (loaded.size() != 1) log.warning(f->str + " holds " + std::to_string(loaded.size()) + " instances where one is wanted");
            if (!loaded.empty()) inst = loaded.front();
        }
        if (!inst) inst = std::make_shared<Instance>(), world.instances.push_back(inst);
        uint32_t material_count = 0;
        std::string name = "name";
        auto set_material = [&](std::shared_ptr<Material> m) {
            if (material_count < Instance::materialCapacity()) inst->materials[material_count++] = std::move(m);
        };
        for (const auto& [key, value] : j.members) {
            if (key == "name" && value.is_string()) name = value.str;
            else if (key == "position") inst->position = to_vec3(value);
            else if (key == "rotation") inst->rotation = to_vec3(value);
            else if (key == "scale") inst->scale = to_vec3(value);
            else if (key == "Material") {
                if (value.is_object()) {
                    if (material_count < Instance::materialCapacity()) set_material(load_material(value));
                } else if (value.is_array()) {
                    for (const auto& m : value.items)
                        if (material_count < Instance::materialCapacity()) set_material(load_material(m));
                } else if (value.is_string() && material_count < Instance::materialCapacity()) {
                    const auto it = materials.find(value.str);
                    if (it == materials.end()) log.error("instance " + name + ": no material named '" + value.str + "'");
                    else set_material(it->second);
                }
            } else if (key == "Mesh") {
                if (inst->mesh) log.warning("instance " + name + ": second mesh reference skipped");
                else inst->mesh = load_mesh(value);
            }
        }
        if (material_count >= Instance::materialCapacity()) log.error("Reached the limit of 64 materials per instance in definition of \"" + name + "\".");
        instances[name] = inst;
        log.message("Loaded instance \"" + name + "\".");
    }

    template <typename F>
    void each(const Json& objects, const char* key, F&& load_one) {  // objectLoad (json_loader.cpp:1034-1061)
        const Json* j = objects.find(key);
        if (!j) return;
        auto guarded = [&](const Json& item) {
            try {
                load_one(item);
            } catch (const Exception& e) {
                log.error(std::string("Failed to load ") + key + ". " + e.what());
            }
        };
        if (j->is_array())
            for (const auto& item : j->items) guarded(item);
        else guarded(*j);
    }

    void load_world(const Json& root) {  // json_loader.cpp:1062-1097
        world.materials.clear(), world.meshes.clear(), world.instances.clear(), world.spot_lights.clear(), world.direct_lights.clear();
        world.camera = Camera{};
        const World fresh;
        world.material = fresh.material, world.default_material = fresh.default_material;
        if (const Json* objects = root.find("Objects")) {
            static const char* map_keys[5] = {"Texture", "NormalMap", "MetalnessMap", "RoughnessMap", "EmissionMap"};
            for (int k = 0; k < 5; ++k) each(*objects, map_keys[k], [&](const Json& j) { load_texture(j, k); });
            each(*objects, "Material", [&](const Json& j) { load_material(j); });
            each(*objects, "Mesh", [&](const Json& j) { load_mesh(j); });
            each(*objects, "Camera", [&](const Json& j) { load_camera(j); });
            each(*objects, "SpotLight", [&](const Json& j) { load_spot_light(j); });
            each(*objects, "DirectLight", [&](const Json& j) { load_direct_light(j); });
            each(*objects, "Instance", [&](const Json& j) { load_instance(j); });
            load_groups(*objects);
        }
        if (const Json* m = root.find("Material")) load_into(*m, world.material);
        if (const Json* m = root.find("DefaultMaterial")) load_into(*m, world.default_material);
        world.makeModified();
        world.camera.makeModified();
    }
};
}  // namespace

void loadScene(const std::string& path, World& world, LoadLog& log) {
    if (extension(path) != ".json") fail("scene files end in .json, not '" + extension(path) + "'");
    std::ifstream file(path, std::ios::binary);
    if (!file.is_open()) fail("cannot open " + path);
    std::stringstream text;
    text << file.rdbuf();
    const std::string s = text.str();
    Json root;
    try {
        root = parseJson(s);
    } catch (const std::runtime_error& e) {
        fail("Failed to parse file " + file_name(path) + ". Reason: " + e.what());
    }
    SceneLoader loader{world, log, parent_dir(path), {}, {}, {}, false};
    loader.load_world(root);
}

// ---------------------------------------------------------------------------------------------------------
// writers
// ---------------------------------------------------------------------------------------------------------
namespace {
std::string num(float v) {
    char b[40];
    std::snprintf(b, sizeof b, "%.9g", double(v));
    std::string s = b;
    if (s.find_first_of(".eEn") == std::string::npos) s += ".0";  // keep it a float for the colour / number rules
    return s;
}
std::string vec3(const vec3f& v) { return "[" + num(v.x) + ", " + num(v.y) + ", " + num(v.z) + "]"; }
std::string color(const Color& c) {
    return "[" + std::to_string(c.red) + ", " + std::to_string(c.green) + ", " + std::to_string(c.blue) + ", " + std::to_string(c.alpha) + "]";
}
// Maps of a saved scene (JsonSaver::saveMap, json_saver.cpp:117-157; BitmapSaver, saver.cpp:16-95): one file per distinct map under
// <scene dir>/maps/<kind>/ — RGBA PNG for textures and normal maps, grey PNG for metalness / roughness, Radiance .hdr for emission.
// Deliberate difference: a normal map is written with its green channel negated back, so that the file holds what was loaded
// (the reference writes the in-memory bitmap, whose green the loader negated: every save / load cycle flips it there).
static const char* address_mode_name(uint32_t sampling) {
    switch (sampling & 0xFF00u) {
        case HIPRZ_TEX_ADDRESS_CLAMP: return "clamp";
        case HIPRZ_TEX_ADDRESS_MIRROR: return "mirror";
        case HIPRZ_TEX_ADDRESS_BORDER: return "border";
        default: return "wrap";
    }
}
struct SavedMaps {
    static constexpr const char* kJsonKeys[5] = {"Texture", "NormalMap", "MetalnessMap", "RoughnessMap", "EmissionMap"};
    static constexpr const char* kDirs[5] = {"texture", "normal", "metalness", "roughness", "emission"};
    static constexpr const char* kMaterialKeys[5] = {"texture", "normal map", "metalness map", "roughness map", "emission map"};
    std::vector<const TextureBuffer*> list[5];
    std::map<const TextureBuffer*, std::string> name[5], file[5];  // file: relative to the scene directory, '/' separators

    static const std::shared_ptr<TextureBuffer>& slot(const Material& m, int k) {
        return k == 0 ? m.texture : k == 1 ? m.normal_map : k == 2 ? m.metalness_map : k == 3 ? m.roughness_map : m.emission_map;
    }
    void collect(const Material& m) {
        for (int k = 0; k < 5; ++k)
            if (const TextureBuffer* t = slot(m, k).get(); t && !name[k].count(t)) {
                name[k][t] = std::string(kDirs[k]) + " " + std::to_string(list[k].size());
                list[k].push_back(t);
            }
    }
    void write(const std::string& scene_dir) {
        for (int k = 0; k < 5; ++k)
            for (size_t i = 0; i < list[k].size(); ++i) {
                const TextureBuffer& t = *list[k][i];
                const std::string rel_dir = std::string("maps/") + kDirs[k], rel = rel_dir + "/" + kDirs[k] + "_" + std::to_string(i) + (k == 4 ? ".hdr" : ".png");
                std::error_code ec;
                std::filesystem::create_directories(scene_dir + rel_dir, ec);
                std::string why;
                bool ok;
                if (k == 4) {
                    ok = writeHDR(scene_dir + rel, reinterpret_cast<const float*>(t.bitmap.data()), t.width, t.height, why);
                } else if (k == 1) {
                    std::vector<uint8_t> px = t.bitmap;
                    for (size_t p = 1; p < px.size(); p += 4) px[p] = uint8_t(-px[p]);
                    ok = writePNG(scene_dir + rel, px.data(), t.width, t.height, 4, why);
                } else {
                    ok = writePNG(scene_dir + rel, t.bitmap.data(), t.width, t.height, k == 0 ? 4 : 1, why);
                }
                if (!ok) fail(why);
                file[k][&t] = rel;
            }
    }
};

std::string material_body(const Material& m, const std::string& name, const SavedMaps* maps = nullptr) {
    std::string s = "{";
    if (!name.empty()) s += "\"name\": \"" + name + "\", ";
    s += "\"color\": " + color(m.color) + ", \"metalness\": " + num(m.metalness()) + ", \"roughness\": " + num(m.roughness()) +
         ", \"emission\": " + num(m.emission()) + ", \"ior\": " + num(m.ior()) + ", \"scattering\": " + num(m.scattering());
    if (maps)
        for (int k = 0; k < 5; ++k)
            if (const TextureBuffer* t = SavedMaps::slot(m, k).get()) s += std::string(", \"") + SavedMaps::kMaterialKeys[k] + "\": \"" + maps->name[k].at(t) + "\"";
    return s + "}";
}
}  // namespace

void saveScene(const std::string& path, const World& world) {
    std::ofstream out(path);
    if (!out.is_open()) fail("Failed to open file " + path + " for writing");
    std::map<const Material*, std::string> mat_name;
    std::map<const Mesh*, std::string> mesh_name;
    SavedMaps maps;
    for (const auto& m : world.materials) maps.collect(*m);
    maps.collect(world.material), maps.collect(world.default_material);
    const size_t slash = path.find_last_of("/\\");
    maps.write(slash == std::string::npos ? std::string() : path.substr(0, slash + 1));
    out << "{\n \"Objects\": {";
    for (int k = 0; k < 5; ++k) {
        if (maps.list[k].empty()) continue;
        out << "\n  \"" << SavedMaps::kJsonKeys[k] << "\": [";
        for (size_t i = 0; i < maps.list[k].size(); ++i) {
            const TextureBuffer& t = *maps.list[k][i];
            out << (i ? ",\n   " : "\n   ") << "{\"name\": \"" << maps.name[k][&t] << "\", \"filter mode\": \"" << ((t.sampling & 0xFFu) == HIPRZ_TEX_FILTER_LINEAR ? "linear" : "point") << "\", \"address mode\": \"" << address_mode_name(t.sampling) << "\", \"scale\": [" << num(t.scale[0]) << ", "
                << num(t.scale[1]) << "], \"rotation\": " << num(t.rotation) << ", \"translation\": [" << num(t.translation[0]) << ", " << num(t.translation[1]) << "], \"file\": \""
                << maps.file[k][&t] << "\"}";
        }
        out << "\n  ],";
    }
    out << "\n  \"Material\": [";
    for (size_t i = 0; i < world.materials.size(); ++i) {
        mat_name[world.materials[i].get()] = "material " + std::to_string(i);
        out << (i ? ",\n   " : "\n   ") << material_body(*world.materials[i], mat_name[world.materials[i].get()], &maps);
    }
    out << "\n  ],\n  \"Mesh\": [";
    size_t n_mesh = 0;
    for (const auto& inst : world.instances) {
        if (!inst->mesh || mesh_name.count(inst->mesh.get())) continue;
        const Mesh& m = *inst->mesh;
        const std::string name = "mesh " + std::to_string(n_mesh);
        mesh_name[&m] = name;
        out << (n_mesh++ ? ",\n   " : "\n   ") << "{\"name\": \"" << name << "\",\n    \"vertices\": [";
        for (size_t i = 0; i < m.vertices.size() / 3; ++i) out << (i ? ", " : "") << "[" << num(m.vertices[3 * i]) << ", " << num(m.vertices[3 * i + 1]) << ", " << num(m.vertices[3 * i + 2]) << "]";
        out << "],\n    \"texcrds\": [";
        for (size_t i = 0; i < m.texcrds.size() / 2; ++i) out << (i ? ", " : "") << "[" << num(m.texcrds[2 * i]) << ", " << num(m.texcrds[2 * i + 1]) << "]";
        out << "],\n    \"normals\": [";
        for (size_t i = 0; i < m.normals.size() / 3; ++i) out << (i ? ", " : "") << "[" << num(m.normals[3 * i]) << ", " << num(m.normals[3 * i + 1]) << ", " << num(m.normals[3 * i + 2]) << "]";
        out << "],\n    \"triangles\": [";
        for (size_t t = 0; t < m.tri_materials.size(); ++t) {
            auto ids = [&](const std::vector<uint32_t>& a) { return "[" + std::to_string(a[3 * t]) + ", " + std::to_string(a[3 * t + 1]) + ", " + std::to_string(a[3 * t + 2]) + "]"; };
            out << (t ? ", " : "") << "{\"v\": " << ids(m.tri_vertices);
            if (m.tri_texcrds[3 * t] != Mesh::ids_unused) out << ", \"t\": " << ids(m.tri_texcrds);
            if (m.tri_normals[3 * t] != Mesh::ids_unused) out << ", \"n\": " << ids(m.tri_normals);
            out << ", \"m\": " << m.tri_materials[t] << "}";
        }
        out << "]}";
    }
    const Camera& c = world.camera;
    out << "\n  ],\n  \"Camera\": [\n   {\"name\": \"camera\", \"position\": " << vec3(c.position) << ", \"rotation\": " << vec3(c.rotation) << ", \"resolution\": [" << c.width << ", "
        << c.height << "], \"fov\": " << num(c.fov) << ", \"near plane\": " << num(c.near_plane) << ", \"far plane\": " << num(c.far_plane) << ", \"focal distance\": "
        << num(c.focal_distance) << ", \"aperture\": " << num(c.aperture) << ", \"exposure time\": " << num(c.exposure_time) << ", \"temporal blend\": " << num(c.temporal_blend) << ", \"enabled\": true}\n  ],\n  \"SpotLight\": [";
    for (size_t i = 0; i < world.spot_lights.size(); ++i) {
        const SpotLight& l = *world.spot_lights[i];
        out << (i ? ",\n   " : "\n   ") << "{\"name\": \"spot " << i << "\", \"position\": " << vec3(l.position) << ", \"direction\": " << vec3(l.direction) << ", \"color\": " << color(l.color)
            << ", \"size\": " << num(l.size) << ", \"emission\": " << num(l.emission) << ", \"angle\": " << num(l.beam_angle) << "}";
    }
    out << "\n  ],\n  \"DirectLight\": [";
    for (size_t i = 0; i < world.direct_lights.size(); ++i) {
        const DirectLight& l = *world.direct_lights[i];
        out << (i ? ",\n   " : "\n   ") << "{\"name\": \"direct " << i << "\", \"direction\": " << vec3(l.direction) << ", \"color\": " << color(l.color) << ", \"emission\": " << num(l.emission)
            << ", \"size\": " << num(l.angular_size) << "}";
    }
    out << "\n  ],\n  \"Instance\": [";
    for (size_t i = 0; i < world.instances.size(); ++i) {
        const Instance& inst = *world.instances[i];
        out << (i ? ",\n   " : "\n   ") << "{\"name\": \"instance " << i << "\", \"position\": " << vec3(inst.position) << ", \"rotation\": " << vec3(inst.rotation) << ", \"scale\": "
            << vec3(inst.scale) << ", \"Material\": [";
        uint32_t count = 0;
        for (uint32_t k = 0; k < Instance::materialCapacity(); ++k)
            if (inst.materials[k]) count = k + 1;
        for (uint32_t k = 0; k < count; ++k) {
            if (!inst.materials[k] || !mat_name.count(inst.materials[k].get())) fail("saveScene: instance material slots must be filled without gaps with materials of the world");
            out << (k ? ", " : "") << "\"" << mat_name[inst.materials[k].get()] << "\"";
        }
        out << "]";
        if (inst.mesh) out << ", \"Mesh\": \"" << mesh_name[inst.mesh.get()] << "\"";
        out << "}";
    }
    out << "\n  ]\n },\n \"Material\": " << material_body(world.material, "", &maps) << ",\n \"DefaultMaterial\": " << material_body(world.default_material, "", &maps) << "\n}\n";
}

void saveOBJ(const std::string& path, const World& world) {
    if (extension(path) != ".obj") fail("saveOBJ: path must end in .obj");
    const std::string stem = path.substr(0, path.size() - 4), mtl_path = stem + ".mtl";
    std::ofstream obj(path), mtl(mtl_path);
    if (!obj.is_open() || !mtl.is_open()) fail("Failed to open " + path + " / " + mtl_path + " for writing");
    std::map<const Material*, std::string> mat_name;
    SavedMaps maps;
    for (const auto& m : world.materials) maps.collect(*m);
    const size_t slash = path.find_last_of("/\\");
    maps.write(slash == std::string::npos ? std::string() : path.substr(0, slash + 1));
    for (size_t i = 0; i < world.materials.size(); ++i) {
        const Material& m = *world.materials[i];
        const std::string name = "material_" + std::to_string(i);
        mat_name[&m] = name;
        mtl << "newmtl " << name << "\nKd " << num(m.color.red / 255.0f) << ' ' << num(m.color.green / 255.0f) << ' ' << num(m.color.blue / 255.0f) << "\nd " << num(m.color.alpha / 255.0f)
            << "\nNi " << num(m.ior()) << "\nPm " << num(m.metalness()) << "\nPr " << num(m.roughness()) << "\nKe " << num(m.emission()) << "\n";
        static const char* statements[5] = {"map_Kd", "norm", "map_Pm", "map_Pr", "map_Ke"};  // MTLSaver::saveMTL (saver.cpp:97-170)
        for (int k = 0; k < 5; ++k)
            if (const TextureBuffer* t = SavedMaps::slot(m, k).get())
                mtl << statements[k] << " -o " << num(t->translation[0]) << ' ' << num(t->translation[1]) << " -s " << num(t->scale[0]) << ' ' << num(t->scale[1]) << " \"" << maps.file[k][t]
                    << "\"\n";
        mtl << "\n";
    }
    obj << "mtllib " << file_name(mtl_path) << "\n";
    size_t v_base = 0, t_base = 0, n_base = 0;
    for (size_t i = 0; i < world.instances.size(); ++i) {
        const Instance& inst = *world.instances[i];
        if (!inst.mesh) continue;
        const Mesh& m = *inst.mesh;
        obj << "o instance_" << i << "\n";
        for (size_t k = 0; k < m.vertices.size() / 3; ++k) obj << "v " << num(m.vertices[3 * k]) << ' ' << num(m.vertices[3 * k + 1]) << ' ' << num(-m.vertices[3 * k + 2]) << "\n";
        for (size_t k = 0; k < m.texcrds.size() / 2; ++k) obj << "vt " << num(m.texcrds[2 * k]) << ' ' << num(m.texcrds[2 * k + 1]) << "\n";
        for (size_t k = 0; k < m.normals.size() / 3; ++k) obj << "vn " << num(m.normals[3 * k]) << ' ' << num(m.normals[3 * k + 1]) << ' ' << num(-m.normals[3 * k + 2]) << "\n";
        uint32_t current = 0xFFFFFFFFu;
        for (size_t t = 0; t < m.tri_materials.size(); ++t) {
            if (m.tri_materials[t] != current) {
                current = m.tri_materials[t];
                const auto& mat = current < Instance::materialCapacity() ? inst.materials[current] : nullptr;
                if (mat && mat_name.count(mat.get())) obj << "usemtl " << mat_name[mat.get()] << "\n";
            }
            obj << "f";
            for (int c : {0, 2, 1}) {  // the loader fans (0, i+2, i+1): write the corners so that it reads them back in this order
                obj << ' ' << (v_base + m.tri_vertices[3 * t + c] + 1);
                const bool has_t = m.tri_texcrds[3 * t + c] != Mesh::ids_unused, has_n = m.tri_normals[3 * t + c] != Mesh::ids_unused;
                if (has_t || has_n) obj << '/';
                if (has_t) obj << (t_base + m.tri_texcrds[3 * t + c] + 1);
                if (has_n) obj << '/' << (n_base + m.tri_normals[3 * t + c] + 1);
            }
            obj << "\n";
        }
        v_base += m.vertices.size() / 3, t_base += m.texcrds.size() / 2, n_base += m.normals.size() / 3;
    }
}

}  // namespace RayZath::Hip::IO
