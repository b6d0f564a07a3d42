// OBJ/MTL mesh loader with the reference's class surface ("OpenGL Raytracer/OBJ.h":7-79).
#pragma once
#include <string>
#include <vector>

#include "vecmath.h"

// OBJ.h:7-24.  32 bytes, the layout GetVertexData() hands to UploadToSSBO (main.cpp:148-163).
struct VertexData {
    rth::vec3 point;
    rth::vec2 texCoord;
    rth::vec3 normal;

    VertexData() = default;
    VertexData(rth::vec3 p, rth::vec2 t, rth::vec3 n) : point(p), texCoord(t), normal(n) {}
};
static_assert(sizeof(VertexData) == 32, "VertexData must stay 8 packed floats");

// OBJ.h:26-47.
struct MeshMaterialData {
    std::string Name = "none";
    unsigned int Illum = 0;
    float Kd[3] = {0, 0, 0};
    float Ka[3] = {0, 0, 0};
    float Tf[3] = {0, 0, 0};
    std::string map_Kd = "none";
    std::string bump = "none";
    std::string disp = "none";
    std::string occlusion = "none";
    float Ni = 0.0f;
};

// OBJ.h:49-54.
struct VertexGroup {
    std::string name;
    std::vector<VertexData> vertices;
    MeshMaterialData materialData;
};

// OBJ.h:56-79.  The GL buffer members of the reference are commented out there and absent here.
class Mesh {
    std::vector<VertexGroup> groups;
    std::string error;

public:
    // OBJ.cpp:150-206.  Appends the file's materials to `data`; false when the file cannot be read.
    bool LoadMaterialData(std::string filepath, std::vector<MeshMaterialData> &data);
    // OBJ.cpp:9-148.  Replaces the mesh's groups; false (and GetError()) on failure, where the
    // reference prints and returns (a missing file) or reads out of bounds (a bad index).
    bool LoadFromObjFile(std::string dir, std::string filename);

    int GetGroupCount() const { return static_cast<int>(groups.size()); }
    const std::string &GetGroupName(int groupIndex) const { return groups.at(groupIndex).name; }
    const std::string &GetError() const { return error; }

    // OBJ.h:75-78; out-of-range groups throw std::out_of_range instead of reading wild memory.
    MeshMaterialData GetGroupMaterialData(int groupIndex) const { return groups.at(groupIndex).materialData; }
    int GetVertexCount(int groupIndex) const { return static_cast<int>(groups.at(groupIndex).vertices.size()); }
    int GetVertexDataSize(int groupIndex) const {
        return static_cast<int>(groups.at(groupIndex).vertices.size() * sizeof(VertexData));
    }
    const VertexData *GetVertexData(int groupIndex) const { return groups.at(groupIndex).vertices.data(); }
};
