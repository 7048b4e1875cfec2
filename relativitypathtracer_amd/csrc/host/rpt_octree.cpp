// rpt_octree.cpp — per-mesh octree with neighbour links, the acceleration structure the render
// path walks (format: rpt_octree in rpt_layout.h).
//
// Restates the reference's builder — Mesh.cpp:5-28 (root), Octree.cpp:171-248 (8-way split,
// neighbour wiring, valence-based stop rule) and Octree.cpp:6-169 (triangle/box overlap by
// separating axes) — with the same fp32 operation order, so the node and octreeTris buffers come
// out byte-identical to what the reference host would upload.  "Next" row f2 of SURVEY.md §8(f).
#include <cmath>
#include <unordered_map>

#include "rpt_scene.h"

namespace rpt {

namespace {

struct AxisTest {
    int edge;      // 0 = B-A, 1 = C-B, 2 = A-C
    int form;      // 0: e.z*p.y - e.y*p.z | 1: -e.z*p.x + e.x*p.z | 2: e.y*p.x - e.x*p.y
    int p, q;      // which two (centre-relative) vertices are projected: 0 = A, 1 = B, 2 = C
};

// the nine edge-cross-axis tests in the reference's order, with the vertex pair each one projects
// (Octree.cpp:25-135)
const AxisTest kAxisTests[9] = {
    {0, 0, 0, 2}, {0, 1, 0, 2}, {0, 2, 1, 2},
    {1, 0, 0, 2}, {1, 1, 0, 2}, {1, 2, 0, 1},
    {2, 0, 0, 1}, {2, 1, 0, 1}, {2, 2, 1, 2},
};

inline float project(int form, const rpt_float3 &e, const rpt_float3 &p) {
    switch (form) {
    case 0: return e.z * p.y - e.y * p.z;
    case 1: return -e.z * p.x + e.x * p.z;
    default: return e.y * p.x - e.x * p.y;
    }
}

inline float radius(int form, const rpt_float3 &abs_e, const rpt_float3 &ext) {
    switch (form) {
    case 0: return abs_e.z * ext.y + abs_e.y * ext.z;
    case 1: return abs_e.z * ext.x + abs_e.x * ext.z;
    default: return abs_e.y * ext.x + abs_e.x * ext.y;
    }
}

}  // namespace

// Octree.cpp:6-169
bool AABBTriangleIntersection(Mesh const &mesh, int octreeIndex, int triIndex) {
    const rpt_float3 A = mesh.vertices[mesh.triangles[9 * triIndex + 3 * 0]];
    const rpt_float3 B = mesh.vertices[mesh.triangles[9 * triIndex + 3 * 1]];
    const rpt_float3 C = mesh.vertices[mesh.triangles[9 * triIndex + 3 * 2]];
    const rpt_float3 bmin = mesh.octree[octreeIndex].min;
    const rpt_float3 bmax = mesh.octree[octreeIndex].max;
    const rpt_float3 center = (bmin + bmax) / 2;
    const rpt_float3 extents = (bmax - bmin) / 2;

    const rpt_float3 off[3] = {A - center, B - center, C - center};
    const rpt_float3 ba = off[1] - off[0];
    const rpt_float3 cb = off[2] - off[1];
    const rpt_float3 ac = off[0] - off[2];
    const rpt_float3 edges[3] = {ba, cb, ac};

    for (const AxisTest &t : kAxisTests) {
        const rpt_float3 &e = edges[t.edge];
        const rpt_float3 abs_e = make_float3(std::fabs(e.x), std::fabs(e.y), std::fabs(e.z));
        float lo = project(t.form, e, off[t.p]);
        float hi = project(t.form, e, off[t.q]);
        if (lo > hi) {
            const float tmp = lo;
            lo = hi;
            hi = tmp;
        }
        const float rad = radius(t.form, abs_e, extents);
        if (lo > rad || hi < -rad) return false;
    }
    {   // triangle plane against the box's most/least aligned corners (Octree.cpp:136-160)
        const rpt_float3 normal = cross(ba, cb);
        rpt_float3 vmin = make_float3(0, 0, 0), vmax = make_float3(0, 0, 0);
        const float n[3] = {normal.x, normal.y, normal.z};
        const float ex[3] = {extents.x, extents.y, extents.z};
        const float a[3] = {off[0].x, off[0].y, off[0].z};
        float *lo = &vmin.x, *hi = &vmax.x;
        for (int k = 0; k < 3; k++) {
            if (n[k] > 0) {
                lo[k] = -ex[k] - a[k];
                hi[k] = ex[k] - a[k];
            } else {
                lo[k] = ex[k] - a[k];
                hi[k] = -ex[k] - a[k];
            }
        }
        if (dot(normal, vmin) > 0) return false;
        if (dot(normal, vmax) < 0) return false;
    }
    {   // triangle bounds against the box (Octree.cpp:161-167)
        const rpt_float3 lo = elementwise_min(elementwise_min(off[0], off[1]), off[2]);
        const rpt_float3 hi = elementwise_max(elementwise_max(off[0], off[1]), off[2]);
        if (lo.x > extents.x || hi.x < -extents.x) return false;
        if (lo.y > extents.y || hi.y < -extents.y) return false;
        if (lo.z > extents.z || hi.z < -extents.z) return false;
    }
    return true;
}

// Octree.cpp:171-248
void Subdivide(Mesh &mesh, int octreeIndex, int minTris, int depth) {
    if (depth <= 0 || mesh.octree[octreeIndex].trisCount <= minTris) return;
    const rpt_float3 extents = mesh.octree[octreeIndex].max - mesh.octree[octreeIndex].min;
    const rpt_float3 half_extents = extents / 2;
    const rpt_float3 ex = make_float3(half_extents.x, 0, 0);
    const rpt_float3 ey = make_float3(0, half_extents.y, 0);
    const rpt_float3 ez = make_float3(0, 0, half_extents.z);
    const int trisStart = mesh.octree[octreeIndex].trisIndex;
    const int trisCount = mesh.octree[octreeIndex].trisCount;

    // stop rule handed to the children: the largest number of this node's triangles that share
    // one vertex (Octree.cpp:180-190)
    std::unordered_map<uint32_t, int> trisPerVertex;
    int maxTrisPerVertex = 0;
    for (int t = trisStart; t < trisStart + trisCount; t++) {
        const int triIndex = mesh.octreeTris[t];
        for (int k = 0; k < 3; k++) {
            const int n = ++trisPerVertex[mesh.triangles[9 * triIndex + 3 * k]];
            if (n > maxTrisPerVertex) maxTrisPerVertex = n;
        }
    }

    // eight children, pushed even when empty, x-major so that children[z + 2y + 4x] are consecutive
    for (int x = 0; x < 2; x++)
        for (int y = 0; y < 2; y++)
            for (int z = 0; z < 2; z++) {
                rpt_octree child;
                for (int &c : child.children) c = -1;
                for (int &n : child.neighbors) n = -1;
                child.min = mesh.octree[octreeIndex].min + ex * (float)x + ey * (float)y + ez * (float)z;
                child.max = child.min + half_extents;
                child.trisIndex = (int)mesh.octreeTris.size();
                child.trisCount = 0;
                const int childOctreeIndex = (int)mesh.octree.size();
                mesh.octree[octreeIndex].children[z + 2 * y + 4 * x] = childOctreeIndex;
                mesh.octree.push_back(child);
                for (int t = trisStart; t < trisStart + trisCount; t++) {
                    const int triIndex = mesh.octreeTris[t];
                    if (AABBTriangleIntersection(mesh, childOctreeIndex, triIndex)) {
                        mesh.octreeTris.push_back(triIndex);
                        mesh.octree[childOctreeIndex].trisCount++;
                    }
                }
            }

    // neighbour links: sibling across the shared face, else the parent's neighbour on that side
    // (side ids 0/1 = -z/+z, 2/3 = -x/+x, 4/5 = -y/+y; Octree.cpp:213-244)
    const rpt_octree parent = mesh.octree[octreeIndex];
    for (int childIndex = 0; childIndex < 8; childIndex++) {
        const int bit[3] = {childIndex & 1, (childIndex >> 2) & 1, (childIndex >> 1) & 1};   // z, x, y
        const int step[3] = {1, 4, 2};
        rpt_octree &child = mesh.octree[parent.children[childIndex]];
        for (int axis = 0; axis < 3; axis++) {
            const int lo = 2 * axis, hi = 2 * axis + 1;
            if (bit[axis] == 0) {
                child.neighbors[lo] = parent.neighbors[lo];
                child.neighbors[hi] = parent.children[childIndex + step[axis]];
            } else {
                child.neighbors[lo] = parent.children[childIndex - step[axis]];
                child.neighbors[hi] = parent.neighbors[hi];
            }
        }
    }
    for (int i = 0; i < 8; i++) Subdivide(mesh, mesh.octree[octreeIndex].children[i], maxTrisPerVertex, depth - 1);
}

// Mesh.cpp:5-28
void Mesh::GenerateOctree(int firstTriIndex) {
    rpt_octree root;
    for (int &c : root.children) c = -1;
    for (int &n : root.neighbors) n = -1;
    root.trisCount = 0;
    root.trisIndex = (int)octreeTris.size();
    root.min = vertices[triangles[firstTriIndex]];
    root.max = vertices[triangles[firstTriIndex]];
    // bounds over the face-corner vertices of this mesh only (every third word is a position index)
    for (size_t i = firstTriIndex / 3; i < triangles.size() / 3; i++) {
        const rpt_float3 vert = vertices[triangles[3 * i]];
        root.min = elementwise_min(root.min, vert);
        root.max = elementwise_max(root.max, vert);
    }
    // the root lists every triangle imported so far, earlier meshes included (Mesh.cpp:16-19)
    for (size_t i = 0; i < triangles.size() / 9; i++) {
        octreeTris.push_back((int32_t)i);
        root.trisCount++;
    }
    const int octreeIndex = (int)octree.size();
    octree.push_back(root);
    Subdivide(*this, octreeIndex, 0, 6);
}

}  // namespace rpt
