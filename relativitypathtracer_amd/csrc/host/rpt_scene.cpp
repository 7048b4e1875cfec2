// rpt_scene.cpp — scene front-end: the stdin DSL, OBJ import with smooth-normal synthesis, the
// texture pool, and the per-frame relativistic refresh of Object[].
//
// Restates, with the same names and argument meaning, the reference's Render.cpp:
//   inputScene   211-416   DSL parser (grammar: README.md:17-75)
//   ReadTexture  418-434   appends one decoded RGB8 image to the byte pool
//   ReadOBJ      436-538   v/vt/vn/f import, 9 words per triangle, area-weighted vertex normals
//   render       149-200   the camera-velocity and Lorentz-matrix part of the frame callback
// "Next" rows f1/f2 of SURVEY.md §8(f).  Differences from the reference, all on paths where the
// reference has undefined or fatal behaviour: errors are returned (lastError) instead of exit();
// input ends at EOF as well as at the R command; paths are resolved against assetRoot with a
// case-insensitive fallback; image decoding is delegated to a TextureDecoder.
#include "rpt_scene.h"

#include <dirent.h>
#include <strings.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

namespace rpt {

namespace {

// "Each # is one floating-point value": strtod, then skip exactly one separator (Render.cpp:266-269)
void parseArgs(const char *curr, float *args, int count) {
    for (int arg = 0; arg < count; arg++) {
        char *endptr = nullptr;
        args[arg] = (float)std::strtod(curr, &endptr);
        curr = *endptr ? endptr + 1 : endptr;
    }
}

bool fileExists(const std::string &p) {
    std::ifstream f(p);
    return (bool)f;
}

// look `name` up in `dir` ignoring case (the reference ran on a case-insensitive file system:
// Scenes/shadows.txt names Models/Pear.obj, the file is Models/pear.obj)
std::string findIgnoringCase(const std::string &dir, const std::string &name) {
    DIR *d = opendir(dir.empty() ? "." : dir.c_str());
    if (!d) return std::string();
    std::string found;
    while (dirent *e = readdir(d)) {
        if (strcasecmp(e->d_name, name.c_str()) == 0) {
            found = e->d_name;
            break;
        }
    }
    closedir(d);
    return found;
}

}  // namespace

std::string Scene::resolve(const std::string &path) const {
    std::string p = path;
    auto it = aliases.find(p);
    if (it != aliases.end()) p = it->second;
    std::string full = (p.size() && p[0] == '/') ? p : assetRoot + "/" + p;
    for (char &c : full) if (c == '\\') c = '/';
    if (fileExists(full)) return full;
    // walk the components case-insensitively
    std::string built = full[0] == '/' ? "/" : "";
    std::stringstream ss(full);
    std::string comp;
    bool first = true;
    while (std::getline(ss, comp, '/')) {
        if (comp.empty()) continue;
        std::string dir = built.empty() ? "." : built;
        std::string hit = findIgnoringCase(dir, comp);
        if (hit.empty()) hit = comp;
        if (!built.empty() && built.back() != '/') built += "/";
        built += hit;
        first = false;
    }
    (void)first;
    return built;
}

bool ReadPPM(const std::string &path, TextureImage &out, std::string &err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "cannot open " + path; return false; }
    std::string magic;
    f >> magic;
    if (magic != "P6") { err = "no decoder for " + path + " (built-in reader handles binary PPM only)"; return false; }
    int vals[3], n = 0;
    while (n < 3 && f) {
        f >> std::ws;
        if (f.peek() == '#') { std::string skip; std::getline(f, skip); continue; }
        f >> vals[n++];
    }
    f.get();
    if (n < 3 || vals[0] <= 0 || vals[1] <= 0 || vals[2] != 255) { err = "bad PPM header in " + path; return false; }
    out.width = vals[0];
    out.height = vals[1];
    out.rgb.resize((size_t)3 * out.width * out.height);
    f.read((char *)out.rgb.data(), (std::streamsize)out.rgb.size());
    if ((size_t)f.gcount() != out.rgb.size()) { err = "truncated PPM " + path; return false; }
    return true;
}

// Render.cpp:418-434 — pool entry is {byte offset, width, height}; pixels interleaved RGB, top row first
bool Scene::AddTexture(const uint8_t *rgb, int width, int height) {
    if (!rgb || width <= 0 || height <= 0) { lastError = "bad texture dimensions"; return false; }
    textureValues.push_back((int)textures.size());
    textureValues.push_back(width);
    textureValues.push_back(height);
    textures.insert(textures.end(), rgb, rgb + (size_t)3 * width * height);
    return true;
}

bool Scene::ReadTexture(const std::string &path) {
    TextureImage img;
    std::string err;
    const std::string full = resolve(path);
    const bool ok = decoder ? decoder(full, img, err) : ReadPPM(full, img, err);
    if (!ok) { lastError = "ReadTexture(" + path + "): " + err; return false; }
    if (img.rgb.size() != (size_t)3 * img.width * img.height) { lastError = "ReadTexture(" + path + "): decoder size mismatch"; return false; }
    return AddTexture(img.rgb.data(), img.width, img.height);
}

// Render.cpp:436-538
bool Scene::ReadOBJ(const std::string &path) {
    int firstTriIndex = 0;
    if (!ReadOBJGeometry(path, firstTriIndex)) return false;
    theMesh.meshIndices.push_back((int)theMesh.octree.size());
    theMesh.GenerateOctree(firstTriIndex);
    return true;
}

// An octree built elsewhere (rpt_build_octree on the GPU) for the geometry ReadOBJGeometry just appended:
// node and triangle indices are already absolute (based at the current array sizes).
bool Scene::AppendOctree(const rpt_octree *nodes, size_t node_count, const int32_t *tris, size_t tri_count) {
    if (!nodes || !node_count || (!tris && tri_count)) { lastError = "AppendOctree: empty octree"; return false; }
    theMesh.meshIndices.push_back((int)theMesh.octree.size());
    theMesh.octree.insert(theMesh.octree.end(), nodes, nodes + node_count);
    theMesh.octreeTris.insert(theMesh.octreeTris.end(), tris, tris + tri_count);
    return true;
}

// Render.cpp:436-533: everything ReadOBJ does before GenerateOctree
bool Scene::ReadOBJGeometry(const std::string &path, int &firstTriIndexOut) {
    Mesh &mesh = theMesh;
    if (path.size() < 4 || path.substr(path.size() - 4, 4) != ".obj") { lastError = "ReadOBJ: not an .obj file: " + path; return false; }
    std::ifstream file(resolve(path));
    if (!file) { lastError = "ReadOBJ: cannot open " + path; return false; }

    std::map<int, std::vector<int>> vertToTrisMap;   // position index -> triangles that need a synthesised normal there
    const int firstTriIndex = (int)mesh.triangles.size();
    const int firstVertIndex = (int)mesh.vertices.size();
    const int firstNormIndex = (int)mesh.normals.size();
    const int firstUVIndex = (int)mesh.uvs.size();
    std::string line;
    int lineno = 0;
    auto syntax = [&](void) {
        lastError = "ReadOBJ(" + path + "): invalid syntax on line " + std::to_string(lineno);
        return false;
    };
    while (std::getline(file, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream stream(line);
        std::string prefix;
        stream >> prefix;
        if (prefix == "v") {
            rpt_float3 vert = make_float3(0, 0, 0);
            stream >> vert.x >> vert.y >> vert.z;
            if (stream.fail()) return syntax();
            mesh.vertices.push_back(vert);
        } else if (prefix == "vt") {
            rpt_float2 uv = {0, 0};
            stream >> uv.x >> uv.y;
            if (stream.fail()) return syntax();
            mesh.uvs.push_back(uv);
        } else if (prefix == "vn") {
            rpt_float3 norm = make_float3(0, 0, 0);
            stream >> norm.x >> norm.y >> norm.z;
            if (stream.fail()) return syntax();
            mesh.normals.push_back(normalize(norm));
        } else if (prefix == "f") {
            // only the first three corners are read: polygons are not triangulated (Render.cpp:486)
            const int triIndex = (int)mesh.triangles.size() / 9;
            for (int corner = 0; corner < 3; corner++) {
                std::string tok;
                if (!(stream >> tok)) return syntax();
                std::istringstream fields(tok);
                std::string vert, uv, norm;
                std::getline(fields, vert, '/');
                if (vert.empty()) return syntax();
                const int vertIndex = (int)std::strtoul(vert.c_str(), nullptr, 10) - 1 + firstVertIndex;
                if (!std::getline(fields, uv, '/')) uv = "1";           // no vt: first uv of this file
                else if (uv.empty()) return syntax();                    // "v//n": the reference throws here
                if (!std::getline(fields, norm, '/')) {
                    norm = "1";                                          // placeholder, replaced below
                    vertToTrisMap[vertIndex].push_back(triIndex);
                }
                mesh.triangles.push_back((uint32_t)vertIndex);
                mesh.triangles.push_back((uint32_t)(std::strtoul(uv.c_str(), nullptr, 10) - 1 + firstUVIndex));
                mesh.triangles.push_back((uint32_t)(std::strtol(norm.c_str(), nullptr, 10) - 1 + firstNormIndex));
            }
        }
        lineno++;
    }
    if ((int)mesh.triangles.size() == firstTriIndex) { lastError = "ReadOBJ(" + path + "): no faces"; return false; }
    for (size_t w = firstTriIndex; w < mesh.triangles.size(); w += 3)
        if (mesh.triangles[w] >= mesh.vertices.size()) { lastError = "ReadOBJ(" + path + "): vertex index out of range"; return false; }

    // area-weighted smooth normals for corners that had no vn, one new normal per vertex in
    // ascending vertex order (Render.cpp:508-533)
    for (const auto &kv : vertToTrisMap) {
        const int vertIndex = kv.first;
        rpt_float3 N = make_float3(0, 0, 0);
        for (int triIndex : kv.second) {
            const int AIndex = (int)mesh.triangles[9 * triIndex + 3 * 0];
            const int BIndex = (int)mesh.triangles[9 * triIndex + 3 * 1];
            const int CIndex = (int)mesh.triangles[9 * triIndex + 3 * 2];
            const rpt_float3 A = mesh.vertices[AIndex];
            const rpt_float3 B = mesh.vertices[BIndex];
            const rpt_float3 C = mesh.vertices[CIndex];
            N += cross(B - A, C - A);   // not normalised: weight = twice the triangle's area
            const int corner = AIndex == vertIndex ? 0 : (BIndex == vertIndex ? 1 : (CIndex == vertIndex ? 2 : -1));
            if (corner >= 0) mesh.triangles[2 + 9 * triIndex + 3 * corner] = (uint32_t)mesh.normals.size();
        }
        mesh.normals.push_back(normalize(N));
    }
    firstTriIndexOut = firstTriIndex;
    return true;
}

// Render.cpp:211-392
bool Scene::inputScene(std::istream &in) {
    white_point = make_float3(1, 1, 1);
    ambient = 1.0f;
    std::string line;
    bool done = false;
    std::string warnings;
    auto warn = [&](const std::string &m) { warnings += m + "\n"; };
    while (!done && std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::vector<char> buf(line.begin(), line.end());
        buf.push_back('\0');
        char *save = nullptr;
        for (char *tok = strtok_r(buf.data(), " ", &save); !done && tok; tok = strtok_r(nullptr, " ", &save)) {
            float args[10];
            const size_t len = std::strlen(tok);
            const bool needObject = std::strchr("pctlvf", tok[0]) != nullptr;
            if (needObject && cpu_objects.empty()) { warn(std::string("Object must be defined before \"") + tok + "\""); continue; }
            if (std::strchr("OpctlvfTMAW", tok[0]) && len < 2) { warn(std::string("command missing argument: \"") + tok + "\""); continue; }
            switch (tok[0]) {
            case 'O': {
                int type = -1;
                if (tok[1] == 's') type = RPT_SPHERE;
                else if (tok[1] == 'c') type = RPT_CUBE;
                else if (tok[1] == 'm') {
                    if (len != 3) { warn("Object mesh command missing argument"); break; }
                    type = RPT_MESH;
                } else { warn(std::string("Object command unrecognized argument: \"") + (tok + 1) + "\""); break; }
                cpu_objects.push_back(defaultObject());
                velocities.push_back(make_float3(0, 0, 0));
                cpu_objects.back().type = type;
                if (type == RPT_MESH) cpu_objects.back().meshIndex = std::atoi(tok + 2);
                break;
            }
            case 'p':
                parseArgs(tok + 1, args, 10);
                TRS(cpu_objects.back(), make_float3(args[0], args[1], args[2]), args[3],
                    make_float3(args[4], args[5], args[6]), make_float3(args[7], args[8], args[9]));
                break;
            case 'c':
                parseArgs(tok + 1, args, 3);
                cpu_objects.back().color = make_float3(args[0], args[1], args[2]);
                break;
            case 't': cpu_objects.back().textureIndex = std::atoi(tok + 1); break;
            case 'l': cpu_objects.back().light = std::atoi(tok + 1) != 0; break;
            case 'v':
                parseArgs(tok + 1, args, 3);
                velocities.back() = make_float3(args[0], args[1], args[2]);
                break;
            case 'f':
                parseArgs(tok + 1, args, 2);
                cpu_objects.back().flashPeriod = args[0];
                cpu_objects.back().flashDuration = args[1];
                break;
            case 'T': if (!ReadTexture(tok + 1)) return false; break;
            case 'M': if (!ReadOBJ(tok + 1)) return false; break;
            case 'A': ambient = (float)std::atof(tok + 1); break;
            case 'W':
                parseArgs(tok + 1, args, 3);
                white_point = make_float3(args[0], args[1], args[2]);
                break;
            case 'I': interval = 0; break;
            case 'R': done = true; break;
            default: warn(std::string("Unrecognized command: \"") + tok + "\"");
            }
        }
    }
    if (!finalizeIndices()) return false;
    lastError = warnings;   // non-fatal diagnostics (the reference prints these to stderr and carries on)
    return true;
}

// Render.cpp:393-413 — t<k> -> {byte offset, width, height}; m<k> -> octree root index
bool Scene::finalizeIndices() {
    if (finalized) return true;
    for (rpt_object &object : cpu_objects) {
        int index = object.textureIndex;
        if (index != -1) {
            if (index < 0 || 3 * (index + 1) > (int)textureValues.size()) {
                lastError = "Error: Texture index " + std::to_string(index) + " out of range";
                return false;
            }
            object.textureIndex = textureValues[3 * index + 0];
            object.textureWidth = textureValues[3 * index + 1];
            object.textureHeight = textureValues[3 * index + 2];
        }
        if (object.type == RPT_MESH) {
            index = object.meshIndex;
            if (index < 0 || index >= (int)theMesh.meshIndices.size()) {
                lastError = "Error: Mesh index " + std::to_string(index) + " out of range";
                return false;
            }
            object.meshIndex = theMesh.meshIndices[index];
        }
    }
    // vt-less meshes point every corner at uvs[firstUVIndex], which the reference then reads past
    // the end of an empty buffer; give those reads a defined (0,0) entry without moving any index
    uint32_t maxUV = 0;
    bool any = false;
    for (size_t w = 1; w < theMesh.triangles.size(); w += 3) { any = true; if (theMesh.triangles[w] > maxUV) maxUV = theMesh.triangles[w]; }
    if (any && maxUV < (1u << 28))
        while (theMesh.uvs.size() <= maxUV) theMesh.uvs.push_back(rpt_float2{0, 0});
    finalized = true;
    return true;
}

// Render.cpp:179-200 — every frame: L_obj = boost(v_obj) * boost(-v_cam), its inverse, and the
// camera event expressed in the object's rest frame
void Scene::updateObjects() {
    rpt_float4 cameraLorentz[4];
    rpt_float4 cameraInvLorentz[4];
    Lorentz(cameraLorentz, cameraVelocity);
    Lorentz(cameraInvLorentz, -cameraVelocity);
    for (size_t i = 0; i < cpu_objects.size(); i++) {
        rpt_object &o = cpu_objects[i];
        setLorentzBoost(o, velocities[i]);
        MatrixMultiplyLeft(o.Lorentz, cameraInvLorentz);
        MatrixMultiplyRight(cameraLorentz, o.InvLorentz);
        o.stationaryCam = make_float4(dot(o.Lorentz[0], cameraPos), dot(o.Lorentz[1], cameraPos),
                                      dot(o.Lorentz[2], cameraPos), dot(o.Lorentz[3], cameraPos));
    }
}

// Render.cpp:159-176 — one frame of held WASDQE keys: rapidity step tanh(ms/5000) along `direction`
void Scene::accelerate(rpt_float3 direction, int frame_ms) {
    if (magnitude(direction) != 0) {
        const rpt_float3 dV = std::tanh(frame_ms / 5000.0f) * normalize(direction);
        cameraVelocity = AddVelocity(cameraVelocity, dV);
    }
}

// Render.cpp:177 — the 3-component += leaves the spatial part (0,0) and clears .w
void Scene::advanceTime(int frame_ms) {
    if (!stopTime) cameraPos += make_float4(frame_ms / 1000.0f, 0, 0, 0);
}

void Scene::toggleInterval() { interval = -!interval; }   // Render.cpp:140

rpt_scene_desc Scene::desc() const {
    rpt_scene_desc d;
    std::memset(&d, 0, sizeof d);
    auto ptr = [](const auto &v) { return v.empty() ? nullptr : v.data(); };
    d.objects = ptr(cpu_objects);          d.object_count = cpu_objects.size();
    d.vertices = ptr(theMesh.vertices);    d.vertex_count = theMesh.vertices.size();
    d.normals = ptr(theMesh.normals);      d.normal_count = theMesh.normals.size();
    d.uvs = ptr(theMesh.uvs);              d.uv_count = theMesh.uvs.size();
    d.triangles = ptr(theMesh.triangles);  d.triangle_words = theMesh.triangles.size();
    d.octrees = ptr(theMesh.octree);       d.octree_count = theMesh.octree.size();
    d.octreeTris = ptr(theMesh.octreeTris); d.octree_tri_count = theMesh.octreeTris.size();
    d.textures = ptr(textures);            d.texture_bytes = textures.size();
    return d;
}

}  // namespace rpt
