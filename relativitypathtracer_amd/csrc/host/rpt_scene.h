// rpt_scene.h — host scene model: the containers and loaders either side of the render path.
//
// Mirrors the reference's host-side scene state (Render.h:10-31, Mesh.h:5-16, Object.h:23-24):
// same container names, same element layouts, same loader names and argument meaning.  The
// reference keeps these as process globals; here they live in one Scene value so that several
// scenes (and several GPUs) can coexist in a process.
#pragma once
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "rpt_vector.h"

namespace rpt {

// Mesh.h:5-16 — one shared SoA pool for every imported mesh
struct Mesh {
    std::vector<rpt_float3> vertices;
    std::vector<uint32_t> triangles;   // 9 words per triangle: [v,uv,n] x 3
    std::vector<rpt_float2> uvs;
    std::vector<rpt_float3> normals;
    std::vector<rpt_octree> octree;
    std::vector<int32_t> octreeTris;
    std::vector<int> meshIndices;      // octree root index of mesh k

    void GenerateOctree(int firstTriIndex);     // Mesh.cpp:5-28
};

bool AABBTriangleIntersection(Mesh const &mesh, int octreeIndex, int triIndex);   // Octree.cpp:6-169
void Subdivide(Mesh &mesh, int octreeIndex, int minTris, int depth);              // Octree.cpp:171-248

// decoded texture: interleaved RGB8, row-major, top row first
struct TextureImage {
    int width = 0, height = 0;
    std::vector<uint8_t> rgb;
};
using TextureDecoder = std::function<bool(const std::string &path, TextureImage &out, std::string &err)>;

struct Scene {
    // Object.h:23-24
    std::vector<rpt_object> cpu_objects;
    std::vector<rpt_float3> velocities;
    // Render.cpp:8-22
    rpt_float3 cameraVelocity = make_float3(0, 0, 0);
    rpt_float4 cameraPos = make_float4(0, 0, 0, 0);   // (t, x, y, z)
    bool stopTime = true;
    int interval = -1;
    rpt_float3 white_point = make_float3(1, 1, 1);
    float ambient = 1.0f;
    Mesh theMesh;
    std::vector<uint8_t> textures;
    std::vector<int> textureValues;    // {byte offset, width, height} per texture

    // asset resolution (the reference opens paths relative to the working directory on a
    // case-insensitive file system)
    std::string assetRoot = ".";
    std::map<std::string, std::string> aliases;
    TextureDecoder decoder;            // defaults to the built-in binary PPM reader
    std::string lastError;

    bool inputScene(std::istream &in);                 // Render.cpp:211-416
    bool ReadTexture(const std::string &path);         // Render.cpp:418-434
    bool AddTexture(const uint8_t *rgb, int width, int height);
    bool ReadOBJ(const std::string &path);             // Render.cpp:436-538
    bool ReadOBJGeometry(const std::string &path, int &firstTriIndex);   // the same without Mesh::GenerateOctree
    bool AppendOctree(const rpt_octree *nodes, size_t node_count, const int32_t *tris, size_t tri_count);
    bool finalizeIndices();                            // Render.cpp:393-413
    void updateObjects();                              // Render.cpp:179-200 (per-frame Lorentz refresh)
    void accelerate(rpt_float3 direction, int frame_ms);   // Render.cpp:159-176 (WASDQE)
    void advanceTime(int frame_ms);                    // Render.cpp:177
    void toggleInterval();                             // Render.cpp:140
    std::string resolve(const std::string &path) const;
    rpt_scene_desc desc() const;
    bool finalized = false;
};

bool ReadPPM(const std::string &path, TextureImage &out, std::string &err);

}  // namespace rpt
