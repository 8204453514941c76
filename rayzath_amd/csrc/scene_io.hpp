// scene_io.hpp — scene files for the HIPGPU backend's host side: RayZath's `.json` scene schema
// (json_loader.cpp:75-1117), Wavefront `.obj` geometry (loader.cpp:738-1035) and `.mtl` material libraries
// (loader.cpp:334-638), read into the stand-alone World twin of hip_engine.hpp, and a `.json` / `.obj` / `.mtl` writer.
//
// Same statements, defaults, clamping and conventions as the reference's loaders: OBJ z is negated (right- to
// left-handed), polygons of up to 8 corners are fanned with swapped winding (0, i+2, i+1), indices may be negative,
// a mesh gets the contiguous sub-range of the file's vertices / texcrds / normals its faces use, `usemtl` maps names to at
// most 64 per-mesh material slots; MTL `Ns` -> roughness = 1 - log10(Ns)/3, `d` / `Tr` -> colour alpha, `Pm` `Pr` `Ke`
// `Ni`, `map_Kd` `norm` `map_Pm` `map_Pr` `map_Ke` with `-o` / `-s`; JSON colours are ints 0-255 or floats 0-1.
// Image files: the reference decodes them with stb_image, which is not part of this repository — image_io.hpp decodes PNG,
// baseline JPEG, BMP, TGA and binary PPM / PGM with stb_image's conventions; anything else is reported in the log and the map is left unset.  Groups are parsed and
// ignored: the CPU kernel this backend follows uses the instance's own transformation (cpu_engine_kernel.cpp:308).
#pragma once

#include <map>
#include <string>
#include <vector>

#include "hip_engine.hpp"

namespace RayZath::Hip::IO {

struct LoadLog {  // LoadResult of the reference (loader.hpp): messages, warnings, errors in order of appearance
    std::vector<std::string> messages, warnings, errors;
    void message(std::string s) { messages.push_back(std::move(s)); }
    void warning(std::string s) { warnings.push_back(std::move(s)); }
    void error(std::string s) { errors.push_back(std::move(s)); }
    std::string str() const;
};

struct ObjMesh {
    std::string name;
    std::shared_ptr<Mesh> mesh;
    std::map<std::string, uint32_t> material_ids;  // usemtl name -> slot
};
struct ObjFile {
    std::vector<ObjMesh> meshes;
    std::vector<std::string> mtllibs;  // as written in the file, in order of first appearance
};
struct NamedMaterial {
    std::string name;
    std::shared_ptr<Material> material;
};

// throw Hip::Exception(HIPRZ_ERR_INVALID, ...) when the file cannot be opened / parsed at all
ObjFile parseOBJ(const std::string& path, LoadLog& log);
std::vector<NamedMaterial> loadMTL(const std::string& path, LoadLog& log);
// OBJLoader::loadInstances: one instance per `o` / `g`, materials from the file's mtllibs; everything is added to `world`
std::vector<std::shared_ptr<Instance>> loadObjInstances(const std::string& path, World& world, LoadLog& log);
// Loader::loadScene / JsonLoader::load: replaces the content of `world` (the first enabled camera becomes world.camera)
void loadScene(const std::string& json_path, World& world, LoadLog& log);

// Writers: `saveScene` writes <path> (.json) with every mesh inline; `saveOBJ` writes one `o` per instance mesh of the
// world plus <stem>.mtl next to it.
void saveScene(const std::string& json_path, const World& world);
void saveOBJ(const std::string& obj_path, const World& world);

// procedural meshes of world.cpp reachable from scene files ("generate cube|plane|sphere|cone|cylinder|torus")
std::shared_ptr<Mesh> generatePlane(uint32_t sides, float width, float height);             // world.cpp:168-200
std::shared_ptr<Mesh> generateSphere(uint32_t resolution, bool normals, bool texcrds);     // world.cpp:202-341 (UV sphere)
std::shared_ptr<Mesh> generateCone(uint32_t side_faces, bool normals);                     // world.cpp:342-397
std::shared_ptr<Mesh> generateCylinder(uint32_t faces, bool normals);                      // world.cpp:399-479
std::shared_ptr<Mesh> generateTorus(uint32_t minor_resolution, uint32_t major_resolution, float minor_radius, float major_radius, bool normals,
                                    bool texcrds);                                         // world.cpp:481-560

}  // namespace RayZath::Hip::IO
