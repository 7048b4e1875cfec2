// rpt_octree_build.hip.h — the octree builder's triangle/box classification on the GPU.
//
// The reference builds each mesh's octree on one CPU core (Mesh.cpp:5-28, Octree.cpp:171-248): for every
// node that is split, each of its triangles is tested against each of the eight child boxes with a
// separating-axis test (Octree.cpp:6-169) and appended, in order, to the child's list.  That product —
// (children of a level) x (triangles of their parent) — is the whole cost and is data parallel.  Here it
// runs level by level on the device: one thread per (child, parent-list entry) evaluates the SAT with the
// reference's fp32 operation order (no contraction) into a flag byte and a per-chunk count; the host turns
// the counts into offsets; a second kernel compacts the surviving triangle ids IN ORDER into the next
// level's lists.  Everything that is O(nodes) — child boxes, the valence stop rule, neighbour links and the
// reference's depth-first numbering — stays on the host, so the node and octreeTris arrays come out
// byte-identical to the host builder's (tests/test_gpu_octree.py).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/rpt_layout.h"

#pragma clang fp contract(off)

namespace rptb {

struct ChildDesc {            // one child box of a node being split, and where its parent's triangle list lives
    float minx, miny, minz, maxx, maxy, maxz;
    unsigned int list_begin;  // parent's list: first entry in the level's list buffer
    unsigned int list_count;
    unsigned int flag_base;   // this child's flags: flags[flag_base + i], i < list_count
    unsigned int chunk_base;  // this child's first 256-entry chunk in the chunk arrays
};

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 half(V3 a) { return v3(a.x / 2, a.y / 2, a.z / 2); }

__device__ __forceinline__ float project(int form, V3 e, V3 p) {
    return form == 0 ? (e.z * p.y - e.y * p.z) : (form == 1 ? (-e.z * p.x + e.x * p.z) : (e.y * p.x - e.x * p.y));
}
__device__ __forceinline__ float radius(int form, V3 ae, V3 ext) {
    return form == 0 ? (ae.z * ext.y + ae.y * ext.z) : (form == 1 ? (ae.z * ext.x + ae.x * ext.z) : (ae.y * ext.x + ae.x * ext.y));
}

// Octree.cpp:6-169 — same nine edge tests (with the vertex pairs the reference projects), plane test and
// bounds test, same operation order as csrc/host/rpt_octree.cpp::AABBTriangleIntersection
__device__ inline bool triangle_overlaps_box(V3 A, V3 B, V3 C, V3 bmin, V3 bmax) {
    const V3 center = half(bmin + bmax);
    const V3 extents = half(bmax - bmin);
    const V3 off[3] = {A - center, B - center, C - center};
    const V3 edges[3] = {off[1] - off[0], off[2] - off[1], off[0] - off[2]};
    const int tests[9][4] = {{0, 0, 0, 2}, {0, 1, 0, 2}, {0, 2, 1, 2}, {1, 0, 0, 2}, {1, 1, 0, 2}, {1, 2, 0, 1}, {2, 0, 0, 1}, {2, 1, 0, 1}, {2, 2, 1, 2}};
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const V3 e = edges[tests[t][0]];
        const V3 ae = v3(__builtin_fabsf(e.x), __builtin_fabsf(e.y), __builtin_fabsf(e.z));
        float lo = project(tests[t][1], e, off[tests[t][2]]);
        float hi = project(tests[t][1], e, off[tests[t][3]]);
        if (lo > hi) { const float tmp = lo; lo = hi; hi = tmp; }
        const float rad = radius(tests[t][1], ae, extents);
        if (lo > rad || hi < -rad) return false;
    }
    {
        const V3 ba = edges[0], cb = edges[1];
        const V3 n = v3(ba.y * cb.z - ba.z * cb.y, ba.z * cb.x - ba.x * cb.z, ba.x * cb.y - ba.y * cb.x);
        V3 vmin, vmax;
        if (n.x > 0) { vmin.x = -extents.x - off[0].x; vmax.x = extents.x - off[0].x; } else { vmin.x = extents.x - off[0].x; vmax.x = -extents.x - off[0].x; }
        if (n.y > 0) { vmin.y = -extents.y - off[0].y; vmax.y = extents.y - off[0].y; } else { vmin.y = extents.y - off[0].y; vmax.y = -extents.y - off[0].y; }
        if (n.z > 0) { vmin.z = -extents.z - off[0].z; vmax.z = extents.z - off[0].z; } else { vmin.z = extents.z - off[0].z; vmax.z = -extents.z - off[0].z; }
        // 4-component dot of the reference with w = 0 on both sides: (x + y + z) + 0*0
        if (n.x * vmin.x + n.y * vmin.y + n.z * vmin.z + 0.0f * 0.0f > 0) return false;
        if (n.x * vmax.x + n.y * vmax.y + n.z * vmax.z + 0.0f * 0.0f < 0) return false;
    }
    {
        const float lox = (off[0].x < off[1].x ? off[0].x : off[1].x), hix = (off[0].x > off[1].x ? off[0].x : off[1].x);
        const float loy = (off[0].y < off[1].y ? off[0].y : off[1].y), hiy = (off[0].y > off[1].y ? off[0].y : off[1].y);
        const float loz = (off[0].z < off[1].z ? off[0].z : off[1].z), hiz = (off[0].z > off[1].z ? off[0].z : off[1].z);
        const float lx = lox < off[2].x ? lox : off[2].x, hx = hix > off[2].x ? hix : off[2].x;
        const float ly = loy < off[2].y ? loy : off[2].y, hy = hiy > off[2].y ? hiy : off[2].y;
        const float lz = loz < off[2].z ? loz : off[2].z, hz = hiz > off[2].z ? hiz : off[2].z;
        if (lx > extents.x || hx < -extents.x) return false;
        if (ly > extents.y || hy < -extents.y) return false;
        if (lz > extents.z || hz < -extents.z) return false;
    }
    return true;
}

// One thread per (chunk, lane): chunk -> (child, 256 consecutive entries of its parent's list).
__global__ __launch_bounds__(256) void sat_flag_kernel(const rpt_float3 *__restrict__ vertices, const uint32_t *__restrict__ triangles,
                                                       const int32_t *__restrict__ level_lists, const ChildDesc *__restrict__ children,
                                                       const unsigned int *__restrict__ chunk_child, unsigned char *__restrict__ flags,
                                                       unsigned int *__restrict__ chunk_counts, unsigned int n_chunks) {
    const unsigned int chunk = blockIdx.x;
    if (chunk >= n_chunks) return;
    const ChildDesc c = children[chunk_child[chunk]];
    const unsigned int i = (chunk - c.chunk_base) * 256u + threadIdx.x;
    bool keep = false;
    if (i < c.list_count) {
        const int tri = level_lists[c.list_begin + i];
        const rpt_float3 a = vertices[triangles[9 * tri + 0]], b = vertices[triangles[9 * tri + 3]], cc = vertices[triangles[9 * tri + 6]];
        keep = triangle_overlaps_box(v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(cc.x, cc.y, cc.z), v3(c.minx, c.miny, c.minz),
                                     v3(c.maxx, c.maxy, c.maxz));
        flags[c.flag_base + i] = keep ? 1 : 0;
    }
    __shared__ unsigned int wave_counts[4];
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0) wave_counts[threadIdx.x >> 6] = (unsigned int)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) chunk_counts[chunk] = wave_counts[0] + wave_counts[1] + wave_counts[2] + wave_counts[3];
}

// Ordered compaction: entry i of the parent's list that survived goes to out[chunk_out[chunk] + (survivors before i in the chunk)].
__global__ __launch_bounds__(256) void sat_compact_kernel(const int32_t *__restrict__ level_lists, const ChildDesc *__restrict__ children,
                                                          const unsigned int *__restrict__ chunk_child, const unsigned char *__restrict__ flags,
                                                          const unsigned int *__restrict__ chunk_out, int32_t *__restrict__ next_lists,
                                                          unsigned int n_chunks) {
    const unsigned int chunk = blockIdx.x;
    if (chunk >= n_chunks) return;
    const ChildDesc c = children[chunk_child[chunk]];
    const unsigned int i = (chunk - c.chunk_base) * 256u + threadIdx.x;
    const bool keep = i < c.list_count && flags[c.flag_base + i] != 0;
    __shared__ unsigned int wave_counts[4];
    const unsigned long long m = __ballot(keep);
    const unsigned int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_counts[wave] = (unsigned int)__popcll(m);
    __syncthreads();
    unsigned int before = (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
    for (unsigned int w = 0; w < wave; w++) before += wave_counts[w];
    if (keep) next_lists[chunk_out[chunk] + before] = level_lists[c.list_begin + i];
}

}  // namespace rptb
