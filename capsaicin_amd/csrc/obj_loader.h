// obj_loader.h — the slice of the tinyobjloader surface the reference consumes (asset_load_system.cpp:47-55,
// 76-89, 100-150): attrib_t / index_t / shape_t / material_t and LoadObj().  The reference's submodule directory
// is empty (third_party/tinyobjloader), so this is a from-scratch parser with the same names and argument meaning.
#pragma once

#include <string>
#include <vector>

namespace tinyobj
{
struct attrib_t
{
    std::vector<float> vertices;   // 3 per position
    std::vector<float> normals;    // 3 per normal
    std::vector<float> texcoords;  // 2 per texcoord
};

struct index_t
{
    int vertex_index;
    int normal_index;    // -1 = absent
    int texcoord_index;  // -1 = absent
};

struct mesh_t
{
    std::vector<index_t> indices;       // 3 per triangle (polygons are fan-triangulated)
    std::vector<int>     material_ids;  // 1 per triangle, -1 = none
};

struct shape_t
{
    std::string name;
    mesh_t      mesh;
};

struct material_t
{
    std::string name;
    float       diffuse[3]  = {0.f, 0.f, 0.f};   // Kd
    float       specular[3] = {0.f, 0.f, 0.f};   // Ks
    float       emission[3] = {0.f, 0.f, 0.f};   // Ke
    float       shininess   = 1.f;               // Ns
    std::string diffuse_texname;                 // map_Kd
};

// Returns false (and fills *err) when the OBJ cannot be opened or a record is malformed.  A missing or unreadable
// `mtllib` file only appends to *warn; materials then stay empty and every face material id is -1.
bool LoadObj(attrib_t* attrib, std::vector<shape_t>* shapes, std::vector<material_t>* materials, std::string* warn,
             std::string* err, const char* filename, const char* mtl_basedir = nullptr);
}  // namespace tinyobj
