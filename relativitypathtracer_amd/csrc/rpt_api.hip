// rpt_api.hip — C-ABI of librpt_hip.so (include/rpt.h): context, scene upload with validation,
// per-frame Object[] refresh, launch and read-back.  Host glue only; the device code is in
// rpt_kernels.hip.h.  Replaces the reference's CLSetup.cpp / main.cpp:33-59 enqueue sequence.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <functional>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rpt.h"
#include "rpt_kernels.hip.h"       /* (+ rpt_diag_walks / rpt_diag_kernels / rpt_persistent under RPT_DIAGNOSTICS) */
#include "rpt_octree_build.hip.h"
#include "rpt_screen_bounds.hpp"
#include "rpt_bounds_certify.hpp"
#include "rpt_workers.hpp"

#pragma clang fp contract(off)

// rpt_relaxed.hip: the opt-in build of the same kernels with OpenCL's default arithmetic (variants 50 / 51)
extern "C" int rpt_launch_relaxed_kernel(int waves_per_simd, const void *args, size_t args_bytes, unsigned grid_x, unsigned grid_y, void *stream);

#define RPT_STAGING_SLOTS 4
#ifdef RPT_DIAGNOSTICS      /* RPT_HOST_PROFILE=1: where the host time of rpt_set_objects goes, summed per context and printed by rpt_destroy */
#include <chrono>
#define RPT_HOST_MARK(K) do { const auto now_ = std::chrono::steady_clock::now(); if (K) ctx->host_us[K] += std::chrono::duration<double, std::micro>(now_ - ctx->host_t).count(); ctx->host_t = now_; if ((K) == 5) ctx->host_calls++; } while (0)
#else
#define RPT_HOST_MARK(K) do { } while (0)
#endif
#define RPT_GRID_ROOTS_MAX 64     /* octree roots that get a descend_from_root table (16 KB each) */
#define RPT_RECT_BATCH_MIN 3      /* screen bounds to recompute in one rpt_set_objects before the helper threads are asked (rpt_workers.hpp); 8 until the bounds were proven as well (round 4: twice the work per object) */

namespace {

struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;      // bytes in use
    size_t capacity = 0;
};

void release(DeviceBuffer &b);

// The resident scene geometry: immutable between two rpt_upload_scene calls, so contexts that keep several frames
// in flight share ONE copy (rpt_share_scene); the last context holding it frees it.
struct Geometry {
    int device = 0;
    DeviceBuffer vertices, normals, uvs, triangles, octrees, octreeTris, textures;
    DeviceBuffer dnodes, dtris, dlinks;       // derived layouts (rpt_kernels.hip.h)
    DeviceBuffer dfirst;                      // first triangle record of every node's list, by node index (the latency walk)
    DeviceBuffer dseen;                       // (diagnostics library) per (node, face): {leaf across the face, mask of this leaf's list entries also in that leaf's list}
    DeviceBuffer dgrids;                      // 16^3 cells per octree root: where four child steps from the root end (descend_from_root)
    int grid_roots = 0;
    bool compact_ok = false;                  // derived octree layout usable (children consecutive)
    unsigned long long generation = 0;        // unique per upload (the rectangle cache of a context names its geometry by this, not by address)
    std::vector<int> node_new_index;          // reference node index -> index in the derived, breadth-first numbering
    int top_count = 0;                        // derived nodes [0, top_count) are the forest's top levels (<= RPT_TOP_MAX)
    std::vector<float> host_node_bounds;      // min.xyz,max.xyz per octree node (culling spheres of mesh roots)
    std::vector<float> node_tri_K, node_tri_L;       // per node: the largest |e1| |e2| and the longest edge of the triangles its list names (mesh_segment_apart's margin)
    std::vector<uint8_t> node_holds_its_triangles;   // per node: every triangle of its list lies inside its box (true of a mesh's root
                                                     // unless its list also holds an earlier mesh's triangles, Mesh.cpp:16-19)
    size_t vertex_count = 0, normal_count = 0, uv_count = 0, triangle_words = 0, octree_count = 0, octree_tri_count = 0;
    ~Geometry() {
        (void)hipSetDevice(device);
        for (DeviceBuffer *b : {&vertices, &normals, &uvs, &triangles, &octrees, &octreeTris, &textures, &dnodes, &dtris, &dlinks, &dfirst, &dgrids, &dseen}) release(*b);
    }
};

}  // namespace

struct rpt_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    hipEvent_t last_event = nullptr;                  // the event recorded last on `stream` (a launch's end or a staging copy's)
    std::string error;

    DeviceBuffer objects;
    std::shared_ptr<Geometry> geo;                    // never null
    DeviceBuffer counters, wave_times;
    DeviceBuffer tile_masks;                          // per-tile object masks of the prepass
    DeviceBuffer verify_planes;                       // rpt_verify_frame: two scratch colour planes + the difference count
    DeviceBuffer claim_counters;                      // persistent kernels: two sets of per-queue claim counters (rpt_persistent.hip.h)
    int cu_count = 0;
    unsigned int claim_epoch = 0;
    std::vector<uint8_t> host_objects;                // last Object[] (DObj depends on `interval`: rebuilt when it changes)
    DeviceBuffer dobjs;
    std::vector<rptb::Rect> rects;                    // last frame's per-object rectangles (reused for unchanged objects)
    int rect_interval = 0x7fffffff;
    unsigned long long rect_geo_generation = ~0ull;   // Geometry::generation the cached rectangles were computed against
    DeviceBuffer owned_out, owned_plane, owned_rgb;
    void *pinned_objects = nullptr;                   // RPT_STAGING_SLOTS pinned slots of Object[] + DObj[]
    size_t pinned_capacity = 0;
    hipEvent_t staging_done[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned int staging_used = 0, staging_next = 0;
    int object_count = 0;
    bool scene_uploaded = false;

    float white_point[3] = {1, 1, 1};
    float ambient = 1.0f;
    int width = 0, height = 0, interval = -1;
    bool params_set = false;

    void *external_out = nullptr;
    void *external_plane = nullptr;
    void *external_rgb = nullptr;
    std::vector<hipEvent_t> timing_events;   // pairs: begin,end per frame
    int timing_frames = -1;                  // -1 = timing region not active
    bool want_owned_rgb = false;
    int first_tile = 0, tile_step = 1, run_log2 = 0;
    bool colour_plane = false;
    int variant = 0;
    int last_variant = 0;                             // the kernel the last launch was made with (rpt_last_variant)
    int msaa = 1;                                     // MSAASAMPLES (rpt_set_msaa)
    int serial = 0;                                   // creation index of this context in the process (diagnostics output)
#ifdef RPT_DIAGNOSTICS
    std::chrono::steady_clock::time_point host_t;     // RPT_HOST_MARK
    double host_us[6] = {0, 0, 0, 0, 0, 0};
    long host_calls = 0;
#endif
    float last_ms = 0.0f;
    bool frame_rendered = false;
    bool latency_call = false;                        // the launch in progress comes from the blocking rpt_render()
    bool has_mesh = true;                             // the current Object[] holds a mesh object (rpt_set_objects); picks the kernel
};

namespace {

int fail(rpt_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->error = msg;
    return code;
}

#define RPT_HIP(ctx, call)                                                                           \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(ctx, RPT_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_));     \
    } while (0)

int reserve(rpt_ctx *ctx, DeviceBuffer &b, size_t bytes) {
    if (bytes > b.capacity || !b.ptr) {      // (!ptr: an empty array still gets a valid, never dereferenced, pointer)
        if (b.ptr) RPT_HIP(ctx, hipFree(b.ptr));
        b.ptr = nullptr;
        b.capacity = 0;
        // at least 16 B so that an empty array still has a valid (never dereferenced) pointer
        const size_t cap = bytes < 16 ? 16 : bytes;
        RPT_HIP(ctx, hipMalloc(&b.ptr, cap));
        b.capacity = cap;
    }
    b.bytes = bytes;
    return RPT_OK;
}

int upload(rpt_ctx *ctx, DeviceBuffer &b, const void *src, size_t bytes) {
    if (int rc = reserve(ctx, b, bytes)) return rc;
    if (bytes) RPT_HIP(ctx, hipMemcpy(b.ptr, src, bytes, hipMemcpyHostToDevice));
    return RPT_OK;
}

void release(DeviceBuffer &b) {
    if (b.ptr) (void)hipFree(b.ptr);
    b = DeviceBuffer();
}

// opencl_kernel.cl:607-616 on the host (same fp32 expression; this file is built without contraction)
float hable_host(float x) {
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}

uint32_t to_u8_host(float c) {   // same definition as the kernel's to_u8
    const float t = c * 255;
    if (!(t == t)) return 0u;
    if (t <= 0.0f) return 0u;
    if (t >= 255.0f) return 255u;
    return (uint32_t)(int)t;
}

// Every index the kernel will follow must stay inside its array: a GPU fault can reset the node.
int validate_geometry(rpt_ctx *ctx, const rpt_scene_desc &s) {
    if (s.triangle_words % RPT_TRI_STRIDE) return fail(ctx, RPT_ERR_SCENE, "triangles: word count is not a multiple of 9");
    const size_t n_tris = s.triangle_words / RPT_TRI_STRIDE;
    for (size_t t = 0; t < n_tris; t++)
        for (int k = 0; k < 3; k++) {
            const uint32_t *w = s.triangles + 9 * t + 3 * k;
            if (w[0] >= s.vertex_count) return fail(ctx, RPT_ERR_SCENE, "triangles: vertex index out of range");
            if (w[1] >= s.uv_count) return fail(ctx, RPT_ERR_SCENE, "triangles: uv index out of range (pad uvs to >= 1 entry for vt-less meshes)");
            if (w[2] >= s.normal_count) return fail(ctx, RPT_ERR_SCENE, "triangles: normal index out of range");
        }
    for (size_t i = 0; i < s.octree_tri_count; i++)
        if (s.octreeTris[i] < 0 || (size_t)s.octreeTris[i] >= n_tris) return fail(ctx, RPT_ERR_SCENE, "octreeTris: triangle index out of range");
    const long long n_nodes = (long long)s.octree_count;
    for (size_t i = 0; i < s.octree_count; i++) {
        const rpt_octree &n = s.octrees[i];
        if (n.trisCount < 0 || n.trisIndex < 0 || (size_t)n.trisIndex + (size_t)n.trisCount > s.octree_tri_count)
            return fail(ctx, RPT_ERR_SCENE, "octree: triangle range out of bounds");
        for (int c = 0; c < 8; c++) {
            if (n.children[c] < -1 || n.children[c] >= n_nodes) return fail(ctx, RPT_ERR_SCENE, "octree: child index out of range");
            if (n.children[0] != -1 && n.children[c] == -1) return fail(ctx, RPT_ERR_SCENE, "octree: interior node with a missing child");
        }
        for (int c = 0; c < 6; c++)
            if (n.neighbors[c] < -1 || n.neighbors[c] >= n_nodes) return fail(ctx, RPT_ERR_SCENE, "octree: neighbour index out of range");
    }
    // the child links must form a forest (no cycles), or the descent loops would not terminate
    std::vector<uint8_t> state(s.octree_count, 0);   // 0 unvisited, 1 on the DFS stack, 2 done
    std::vector<std::pair<int, int>> stack;
    for (size_t r = 0; r < s.octree_count; r++) {
        if (state[r]) continue;
        stack.push_back({(int)r, 0});
        state[r] = 1;
        while (!stack.empty()) {
            auto &top = stack.back();
            const rpt_octree &n = s.octrees[top.first];
            if (n.children[0] == -1 || top.second == 8) {
                state[top.first] = 2;
                stack.pop_back();
                continue;
            }
            const int c = n.children[top.second++];
            if (state[c] == 1) return fail(ctx, RPT_ERR_SCENE, "octree: child links form a cycle");
            if (state[c] == 0) {
                state[c] = 1;
                stack.push_back({c, 0});
            }
        }
    }
    return RPT_OK;
}

// Derived octree layout: one 64-B DNode per node and one 48-B DTri per leaf triangle reference.
// Only the numbers the reference stores (and B-A, C-A, which intersect_triangle would form from
// them with the same IEEE subtraction) go in, so traversal results are unchanged.
// Nodes are renumbered breadth first over the whole forest — every root, then every node of level 1, and so on — so that
// (a) the eight children of a node stay consecutive (they are one block in the reference's numbering too, Octree.cpp:191-269)
// and (b) the top levels of every octree are the first `top_count` records: the persistent kernels keep those nodes' links
// in LDS.  Geometry::node_new_index maps the reference's node index (Object::meshIndex) to the derived one (DObj::root).
int build_derived_geometry(rpt_ctx *ctx, const rpt_scene_desc &s) {
    Geometry &g = *ctx->geo;
    g.compact_ok = false;
    g.top_count = 0;
    g.grid_roots = 0;
    g.node_new_index.clear();
    if (s.octree_count == 0) {
        g.compact_ok = true;
        if (int rc = reserve(ctx, g.dnodes, 0)) return rc;
        if (int rc = reserve(ctx, g.dlinks, 0)) return rc;
        return reserve(ctx, g.dtris, 0);
    }
    const size_t n = s.octree_count;
    if (n > (size_t)RPT_LINK_CHILD_MASK) return RPT_OK;                // a link holds 24 bits of child index
    std::vector<uint8_t> is_child(n, 0);
    for (size_t i = 0; i < n; i++) {
        const rpt_octree &o = s.octrees[i];
        if (o.children[0] == -1) continue;
        for (int c = 0; c < 8; c++) {
            if (o.children[c] != o.children[0] + c) return RPT_OK;     // not consecutive: keep the general kernel
            if (is_child[(size_t)o.children[c]]) return RPT_OK;        // a node with two parents: not a forest of trees
            is_child[(size_t)o.children[c]] = 1;
        }
    }
    std::vector<int> order;                                            // derived index -> reference index
    order.reserve(n);
    std::vector<int> &new_index = g.node_new_index;
    new_index.assign(n, -1);
    std::vector<int> level, next;
    for (size_t i = 0; i < n; i++) if (!is_child[i]) level.push_back((int)i);
    while (!level.empty()) {
        next.clear();
        for (int i : level) {
            new_index[(size_t)i] = (int)order.size();
            order.push_back(i);
            if (s.octrees[i].children[0] != -1)
                for (int c = 0; c < 8; c++) next.push_back(s.octrees[i].children[c]);
        }
        if (order.size() <= (size_t)RPT_TOP_MAX) g.top_count = (int)order.size();    // whole levels only
        level.swap(next);
    }
    if (order.size() != n) return RPT_OK;                              // (cannot happen after validate_geometry: every node is a root or a child)
    std::vector<rptd::DNode> nodes(n);
    std::vector<int32_t> links(n);
    std::vector<rptd::DTri> tris;
    for (size_t k = 0; k < n; k++) {
        const rpt_octree &o = s.octrees[order[k]];
        rptd::DNode &d = nodes[k];
        std::memset(&d, 0, sizeof d);
        d.minx = o.min.x; d.miny = o.min.y; d.minz = o.min.z;
        d.maxx = o.max.x; d.maxy = o.max.y; d.maxz = o.max.z;
        d.link = -1;
        if (o.children[0] != -1) {
            unsigned int leaf_mask = 0;
            for (int c = 0; c < 8; c++) leaf_mask |= (s.octrees[o.children[c]].children[0] == -1 ? 1u : 0u) << c;
            d.link = (int)((unsigned int)new_index[(size_t)o.children[0]] | (leaf_mask << 24));
            if (d.link == -1) return RPT_OK;                            // (first child 0xffffff with eight leaf children: reserved for "leaf")
        }
        links[k] = d.link;
        for (int c = 0; c < 6; c++) d.nb[c] = o.neighbors[c] == -1 ? -1 : new_index[(size_t)o.neighbors[c]];
        d.leafBegin = (int)tris.size();
        d.leafCount = 0;
        if (o.children[0] == -1) {
            d.leafCount = o.trisCount;
            for (int t0 = o.trisIndex; t0 < o.trisIndex + o.trisCount; t0++) {
                const int t = s.octreeTris[t0];
                const rpt_float3 &A = s.vertices[s.triangles[9 * t + 0]];
                const rpt_float3 &B = s.vertices[s.triangles[9 * t + 3]];
                const rpt_float3 &C = s.vertices[s.triangles[9 * t + 6]];
                rptd::DTri r;
                std::memset(&r, 0, sizeof r);
                r.ax = A.x; r.ay = A.y; r.az = A.z;
                r.e1x = B.x - A.x; r.e1y = B.y - A.y; r.e1z = B.z - A.z;
                r.e2x = C.x - A.x; r.e2y = C.y - A.y; r.e2z = C.z - A.z;
                r.tri = t;
                tris.push_back(r);
            }
        }
    }
    if (tris.size() > (size_t)RPT_NODE_BEGIN_MASK) return RPT_OK;      // (16.7 M triangle references: the general kernel takes such a scene)
    {   // the first record of every list again, by node index (load_first_tri) — before leafBegin gets the count packed into it
        std::vector<rptd::DTri> first(n);
        std::memset(first.data(), 0, first.size() * sizeof(rptd::DTri));
        for (size_t k = 0; k < n; k++) if (nodes[k].leafCount > 0) first[k] = tris[(size_t)nodes[k].leafBegin];
        if (int rc = upload(ctx, g.dfirst, first.data(), first.size() * sizeof(rptd::DTri))) return rc;
    }
#ifdef RPT_DIAGNOSTICS
    {   // descend_from_root's tables (measurement arms 593 / 605): for every root (the first records of the breadth-first numbering; at most RPT_GRID_ROOTS_MAX of
        // them) and every cell of the 16^3 grid over its box, the node that four child steps reach — or the leaf met before
        size_t roots = 0;
        for (size_t i = 0; i < n; i++) roots += is_child[i] ? 0 : 1;
        if (roots > (size_t)RPT_GRID_ROOTS_MAX) roots = RPT_GRID_ROOTS_MAX;
        std::vector<int32_t> grids(roots * RPT_GRID_CELLS);
        for (size_t r = 0; r < roots; r++)
            for (int cx = 0; cx < 16; cx++) for (int cy = 0; cy < 16; cy++) for (int cz = 0; cz < 16; cz++) {
                int node = (int)r, lvl = 0;
                while (nodes[(size_t)node].link != -1 && lvl < RPT_GRID_LEVELS) {
                    const int sh = RPT_GRID_LEVELS - 1 - lvl;
                    const int child = (((cx >> sh) & 1) << 2) | (((cy >> sh) & 1) << 1) | ((cz >> sh) & 1);      // opencl_kernel.cl:257: z + 2 y + 4 x
                    node = (nodes[(size_t)node].link & RPT_LINK_CHILD_MASK) + child;
                    lvl++;
                }
                grids[r * RPT_GRID_CELLS + (size_t)((cx * 16 + cy) * 16 + cz)] = node | (lvl << 24) | ((nodes[(size_t)node].link == -1 ? 1 : 0) << 28);
            }
        if (int rc = upload(ctx, g.dgrids, grids.data(), grids.size() * sizeof(int32_t))) return rc;
        g.grid_roots = (int)roots;
    }
    {   // octree_walk<..., DEDUP>'s table (measurement arms 705 / 717): per leaf and face, the leaf across that face (if the link points
        // at a leaf) and which of this leaf's first 32 list entries name a triangle that is also in THAT leaf's list
        std::vector<uint32_t> seen(n * 12, 0u);
        std::vector<int> sorted_ids;
        for (size_t k = 0; k < n; k++) {
            for (int f = 0; f < 6; f++) { seen[(k * 6 + f) * 2] = 0xffffffffu; seen[(k * 6 + f) * 2 + 1] = 0u; }
            if (nodes[k].link != -1 || nodes[k].leafCount <= 0) continue;
            for (int f = 0; f < 6; f++) {
                const int q = nodes[k].nb[f];
                if (q < 0 || nodes[(size_t)q].link != -1 || nodes[(size_t)q].leafCount <= 0) continue;
                sorted_ids.clear();
                for (int t = 0; t < nodes[(size_t)q].leafCount; t++) sorted_ids.push_back(tris[(size_t)nodes[(size_t)q].leafBegin + (size_t)t].tri);
                std::sort(sorted_ids.begin(), sorted_ids.end());
                uint32_t mask = 0u;
                for (int t = 0; t < nodes[k].leafCount && t < 32; t++)
                    if (std::binary_search(sorted_ids.begin(), sorted_ids.end(), tris[(size_t)nodes[k].leafBegin + (size_t)t].tri)) mask |= 1u << t;
                seen[(k * 6 + f) * 2] = (uint32_t)q;
                seen[(k * 6 + f) * 2 + 1] = mask;
            }
        }
        if (int rc = upload(ctx, g.dseen, seen.data(), seen.size() * sizeof(uint32_t))) return rc;
    }
#endif
    for (size_t k = 0; k < n; k++)          // leafBegin | min(leafCount, 255) << 24: see load_node_rec
        nodes[k].leafBegin = (int)((unsigned int)nodes[k].leafBegin | ((unsigned int)(nodes[k].leafCount < 255 ? nodes[k].leafCount : 255) << 24));
    // (16 B of slack behind the triangle records: the cooperative 16-B staging loads of the persistent kernels never start past the
    // last record, but keep the buffer's end away from them all the same)
    if (int rc = upload(ctx, g.dnodes, nodes.data(), nodes.size() * sizeof(rptd::DNode))) return rc;
    if (int rc = upload(ctx, g.dlinks, links.data(), links.size() * sizeof(int32_t))) return rc;
    if (int rc = upload(ctx, g.dtris, tris.data(), tris.size() * sizeof(rptd::DTri))) return rc;
    g.compact_ok = true;
    return RPT_OK;
}

// The constants of mesh_segment_apart (rpt_kernels.hip.h, where the derivation is): may a shadow ray's segment to the light be
// dropped for mesh object `o` when it stays beyond a plane of the root box, and with which margins?  Everything the argument
// assumes about THIS object in THIS frame is checked here, in double, on the float matrices the kernel reads; if anything fails
// the object is simply never culled (mslope < 0).
void mesh_segment_cull_record(const rpt_ctx *ctx, const rpt_object &o, rptd::DObj &d) {
    d.mslope = -1.0f;
    d.mcw = d.mconst = d.ms0 = 0.0f;
    d.mh[0] = d.mh[1] = d.mh[2] = -1.0f;
    d.pad2 = 0.0f;
    if (o.type != RPT_MESH || !ctx->geo->compact_ok) return;
    const size_t mi = (size_t)o.meshIndex;
    if (o.meshIndex < 0 || mi * 6 + 5 >= ctx->geo->host_node_bounds.size() || mi >= ctx->geo->node_tri_K.size()) return;
    const float *nb = &ctx->geo->host_node_bounds[mi * 6];
    const double U = 5.9604644775390625e-8;
    // mesh_ray_misses_root: half extents about the centre the kernel reads, grown by 8u max(|lo|, |hi|)
    const float cen[3] = {d.cbx, d.cby, d.cbz};
    double h1 = 0.0;
    float mh[3];
    for (int k = 0; k < 3; k++) {
        if (!std::isfinite(nb[k]) || !std::isfinite(nb[k + 3]) || !std::isfinite(cen[k]) || !(nb[k] <= nb[k + 3])) return;
        const double h = std::max((double)nb[k + 3] - cen[k], (double)cen[k] - nb[k]);
        h1 += h;
        mh[k] = std::nextafter((float)(h + 8.0 * U * std::max(std::fabs((double)nb[k]), std::fabs((double)nb[k + 3])) + 1.0e-30), INFINITY);
        if (!std::isfinite(mh[k])) return;
    }
    for (int k = 0; k < 3; k++) d.mh[k] = mh[k];
    // mesh_segment_apart: only for a mesh whose lists stay inside its root box, with light propagation on (interval 0: no light is
    // ever sampled, opencl_kernel.cl:572), and only if everything the derivation assumes holds for these matrices
    if (d.mesh_in_box == 0.0f || ctx->interval != -1) return;
    const double K = ctx->geo->node_tri_K[mi], L = ctx->geo->node_tri_L[mi];
    double M3[3][3], I3[3][3], Mt[3], It[3], Ls[3][3], L0[3];
    for (int r = 0; r < 3; r++) {
        M3[r][0] = o.M[r].x; M3[r][1] = o.M[r].y; M3[r][2] = o.M[r].z; Mt[r] = o.M[r].w;
        I3[r][0] = o.InvM[r].x; I3[r][1] = o.InvM[r].y; I3[r][2] = o.InvM[r].z; It[r] = o.InvM[r].w;
        L0[r] = o.Lorentz[r + 1].x; Ls[r][0] = o.Lorentz[r + 1].y; Ls[r][1] = o.Lorentz[r + 1].z; Ls[r][2] = o.Lorentz[r + 1].w;
    }
    // c1 = ||M3 InvM3 - I||_F + 16u || |M3| |InvM3| ||_F,  c0 = |M3 InvM.t + M.t| + 16u (|| |M3| |InvM.t| || + |M.t|)
    double r3 = 0.0, km = 0.0, rt = 0.0, mit = 0.0, mt = 0.0;
    for (int i = 0; i < 3; i++) {
        double t = Mt[i], ta = 0.0;
        for (int j = 0; j < 3; j++) {
            double v = 0.0, va = 0.0;
            for (int k = 0; k < 3; k++) { v += M3[i][k] * I3[k][j]; va += std::fabs(M3[i][k]) * std::fabs(I3[k][j]); }
            v -= i == j ? 1.0 : 0.0;
            r3 += v * v;
            km += va * va;
            t += M3[i][j] * It[j];
            ta += std::fabs(M3[i][j]) * std::fabs(It[j]);
        }
        rt += t * t;
        mit += ta * ta;
        mt += Mt[i] * Mt[i];
    }
    const double c1 = std::sqrt(r3) + 16.0 * U * std::sqrt(km), c0 = std::sqrt(rt) + 16.0 * U * (std::sqrt(mit) + std::sqrt(mt));
    // dmin: |dw| >= (1 - |Ls^-1 L0|) / ||Ls^-1||_F for every unit light direction (dw = Ls nd - L0); halved for dw's own rounding
    double Li[3][3];
    if (!rptb::detail::invert3(Ls, Li)) return;
    double cvec[3], fro = 0.0, labs = 0.0;
    for (int i = 0; i < 3; i++) {
        cvec[i] = Li[i][0] * L0[0] + Li[i][1] * L0[1] + Li[i][2] * L0[2];
        for (int j = 0; j < 3; j++) { fro += Li[i][j] * Li[i][j]; labs += Ls[i][j] * Ls[i][j]; }
        labs += L0[i] * L0[i];
    }
    const double cn = std::sqrt(cvec[0] * cvec[0] + cvec[1] * cvec[1] + cvec[2] * cvec[2]);
    if (!(cn < 1.0 - 1.0e-6) || !(fro > 0.0)) return;
    const double dmin = 0.5 * (1.0 - cn) / std::sqrt(fro);
    if (!(16.0 * U * std::sqrt(labs) <= dmin)) return;                       // the float noise of dw itself
    const double slope = 16.2 * K + 3.2 * U;
    if (!(c1 <= 4.0e-4) || !(c0 <= 0.1 * dmin) || !(slope <= 0.25) || !std::isfinite(c1) || !std::isfinite(c0) || !std::isfinite(L) || !std::isfinite(h1)) return;
    d.mconst = std::nextafter((float)(slope * h1 * 1.001 + 4.0 * U * L + 1.0e-30), INFINITY);
    d.mslope = std::nextafter((float)(slope * 1.001), INFINITY);
    d.mcw = std::nextafter((float)(1.01 * c1 / dmin), INFINITY);
    d.ms0 = std::nextafter((float)(1.0e-4 + 1.01 * c0 / dmin), INFINITY);
}

// Per-frame DObj records: the primary-ray origin in each object's space and what follows from it
// (opencl_kernel.cl:314,318 / 336,341), with the kernel's operation order.
void build_dobjs(const rpt_ctx *ctx, const rpt_object *objs, int count, rptd::DObj *out) {
    for (int i = 0; i < count; i++) {
        const rpt_object &o = objs[i];
        const float vx = o.stationaryCam.y, vy = o.stationaryCam.z, vz = o.stationaryCam.w;
        float p[3];
        for (int r = 0; r < 3; r++) p[r] = o.InvM[r].x * vx + o.InvM[r].y * vy + o.InvM[r].z * vz + o.InvM[r].w * 1.0f;
        rptd::DObj d;
        std::memset(&d, 0, sizeof d);
        d.ox = p[0]; d.oy = p[1]; d.oz = p[2];
        const float rx = -p[0], ry = -p[1], rz = -p[2];
        d.sphere_c = (rx * rx + ry * ry + rz * rz) - 1.0f;
        const float ax = std::fabs(p[0]), ay = std::fabs(p[1]), az = std::fabs(p[2]);
        const float m1 = ax < ay ? ay : ax;
        const float m2 = m1 < az ? az : m1;
        d.winding = m2 < 1.0f ? -1.0f : 1.0f;
        // ---- culling data (approximate on purpose; see rpt_tile_bin_kernel) ----
        // object-space direction of a primary ray with camera direction nd: InvM3 * (interval * L[1..3][0] + L[1..3][1..3] * nd)
        const float interval = (float)ctx->interval;
        bool finite = true;
        for (int r = 0; r < 3; r++) {
            const float m[3] = {o.InvM[r].x, o.InvM[r].y, o.InvM[r].z};
            for (int c = 0; c < 3; c++) {
                const float lc[3] = {(&o.Lorentz[1].x)[c + 1], (&o.Lorentz[2].x)[c + 1], (&o.Lorentz[3].x)[c + 1]};
                d.B[3 * r + c] = m[0] * lc[0] + m[1] * lc[1] + m[2] * lc[2];
                finite = finite && std::isfinite(d.B[3 * r + c]);
            }
            d.b[r] = interval * (m[0] * o.Lorentz[1].x + m[1] * o.Lorentz[2].x + m[2] * o.Lorentz[3].x);
            finite = finite && std::isfinite(d.b[r]) && std::isfinite(p[r]);
        }
        d.cbx = d.cby = d.cbz = 0.0f;
        float radius = -1.0f;
        // (sphere: the float discriminant b^2 - c is off by up to ~1.5e-6 |oc|^2, see sphere_rect in rpt_screen_bounds.hpp)
        if (o.type == RPT_SPHERE) radius = std::sqrt(1.0f + 1.5e-6f * (d.sphere_c + 1.0f));
        else if (o.type == RPT_CUBE) radius = 1.7320508f;
        else if (o.type == RPT_MESH && (size_t)o.meshIndex * 6 + 5 < ctx->geo->host_node_bounds.size()) {
            const float *nb = &ctx->geo->host_node_bounds[(size_t)o.meshIndex * 6];
            d.cbx = 0.5f * (nb[0] + nb[3]); d.cby = 0.5f * (nb[1] + nb[4]); d.cbz = 0.5f * (nb[2] + nb[5]);
            const float ex = nb[3] - nb[0], ey = nb[4] - nb[1], ez = nb[5] - nb[2];
            radius = 0.5f * std::sqrt(ex * ex + ey * ey + ez * ez);
        }
        // a relative gamma beyond 20: the float evaluation of the boosted direction is too noisy for this approximate test
        finite = finite && std::isfinite(o.Lorentz[0].x) && std::fabs(o.Lorentz[0].x) <= 20.0f;
        d.rb = (finite && radius >= 0.0f && std::isfinite(radius)) ? radius * 1.02f + 1.0e-5f : -1.0f;
        d.root = (o.type == RPT_MESH && ctx->geo->compact_ok && o.meshIndex >= 0 && (size_t)o.meshIndex < ctx->geo->node_new_index.size())
                     ? ctx->geo->node_new_index[(size_t)o.meshIndex] : 0;
        d.mesh_in_box = (o.type == RPT_MESH && o.meshIndex >= 0 && (size_t)o.meshIndex < ctx->geo->node_holds_its_triangles.size() &&
                         ctx->geo->node_holds_its_triangles[(size_t)o.meshIndex] && ctx->geo->compact_ok) ? 1.0f : 0.0f;
        mesh_segment_cull_record(ctx, o, d);
        out[i] = d;
    }
}

// Per-frame image-plane rectangle of every object (rpt_screen_bounds.hpp) for the in-kernel lane-parallel cull.  A
// rectangle is a pure function of the object's 320 bytes, the interval and (meshes) the root bounds: an object whose
// record is byte-identical to the previous frame's (a camera at rest, a paused scene) keeps its rectangle.
struct RectBatch {
    rpt_ctx *ctx;
    const rpt_object *objs;
    rptb::Rect *out;
    int todo[64];
};
void rect_batch_item(void *arg, int k) {
    RectBatch &b = *(RectBatch *)arg;
    const int i = b.todo[k];
    const rpt_object &o = b.objs[i];
    const float *root = nullptr;
    if (o.type == RPT_MESH && o.meshIndex >= 0 && (size_t)o.meshIndex * 6 + 5 < b.ctx->geo->host_node_bounds.size())
        root = &b.ctx->geo->host_node_bounds[(size_t)o.meshIndex * 6];
    b.out[i] = b.ctx->rects[i] = rptb::certified_object_rect(o, b.ctx->interval, root);      // proposed by sampling, PROVEN or dropped (rpt_bounds_certify.hpp)
}
void build_rects(rpt_ctx *ctx, const rpt_object *objs, int count, rptb::Rect *out) {
    const bool comparable = ctx->rect_interval == ctx->interval && ctx->rect_geo_generation == ctx->geo->generation &&
                            ctx->rects.size() == (size_t)count && ctx->host_objects.size() == (size_t)count * sizeof(rpt_object);
    const rpt_object *prev = comparable ? (const rpt_object *)ctx->host_objects.data() : nullptr;
    ctx->rects.resize((size_t)count);
    RectBatch batch;
    batch.ctx = ctx;
    batch.objs = objs;
    batch.out = out;
    int n_todo = 0;
    for (int i = 0; i < count; i++) {
        if (i >= 64) { out[i] = ctx->rects[i] = rptb::full_rect(); continue; }      // the wave's mask has 64 bits: later objects are always tested
        if (prev && std::memcmp(&prev[i], &objs[i], sizeof objs[i]) == 0) { out[i] = ctx->rects[i]; continue; }
        batch.todo[n_todo++] = i;
    }
    // the bounds that have to be recomputed: a few on the submitting thread, a batch shared with the helper threads (rpt_workers.hpp;
    // cubes.txt, 34 moving objects: profiles/r03_host_threads.txt)
    if (n_todo >= RPT_RECT_BATCH_MIN) rpth::Workers::instance().parallel_for(n_todo, rect_batch_item, &batch);
    else for (int k = 0; k < n_todo; k++) rect_batch_item(&batch, k);
    ctx->rect_interval = ctx->interval;
    ctx->rect_geo_generation = ctx->geo->generation;
}

int validate_objects(rpt_ctx *ctx, const rpt_object *objs, int count) {
    for (int i = 0; i < count; i++) {
        const rpt_object &o = objs[i];
        if (o.type < RPT_SPHERE || o.type > RPT_MESH) return fail(ctx, RPT_ERR_SCENE, "object: unknown type");
        if (o.type == RPT_MESH && (o.meshIndex < 0 || (size_t)o.meshIndex >= ctx->geo->octree_count))
            return fail(ctx, RPT_ERR_SCENE, "object: meshIndex is not an octree node");
        if (o.textureIndex != -1) {
            if (o.textureIndex < 0 || o.textureWidth <= 0 || o.textureHeight <= 0 ||
                (unsigned long long)o.textureIndex + 3ull * (unsigned long long)o.textureWidth * (unsigned long long)o.textureHeight > ctx->geo->textures.bytes)
                return fail(ctx, RPT_ERR_SCENE, "object: texture lies outside the texture pool");
        }
    }
    return RPT_OK;
}

int local_tile_count(const rpt_ctx *ctx);

int ensure_outputs(rpt_ctx *ctx) {
    const size_t px = (size_t)ctx->width * ctx->height;
    if (ctx->colour_plane) {
        const int local_tiles = local_tile_count(ctx);
        if (!ctx->external_plane)
            if (int rc = reserve(ctx, ctx->owned_plane, (size_t)local_tiles * RPT_TILE_ROWS * ctx->width * 4)) return rc;
    } else if (!ctx->external_out) {
        if (px * 16 > ctx->owned_out.capacity || ctx->owned_out.bytes != px * 16) {
            if (int rc = reserve(ctx, ctx->owned_out, px * 16)) return rc;
            RPT_HIP(ctx, hipMemsetAsync(ctx->owned_out.ptr, 0, px * 16, ctx->stream));
        }
    }
    if (ctx->want_owned_rgb) {
        if (int rc = reserve(ctx, ctx->owned_rgb, px * 12)) return rc;
    }
    return RPT_OK;
}

// local tiles of this context: local tile t is global tile (t >> run_log2) * tile_step + first_tile + (t & (run - 1))
int local_tile_count(const rpt_ctx *ctx) {
    const int tiles = (ctx->height + RPT_TILE_ROWS - 1) / RPT_TILE_ROWS;
    const int run = 1 << ctx->run_log2;
    if (ctx->first_tile >= tiles) return 0;
    const int full_periods = (tiles - ctx->first_tile) / ctx->tile_step;          // periods whose whole run may lie inside
    const int rest = tiles - ctx->first_tile - full_periods * ctx->tile_step;    // tiles of the last, partial period
    return full_periods * run + (rest < run ? rest : run);
}

// The band of tile rows that holds the meshes' screen rectangles (whole-frame contexts only), grown by one row each way:
// where the frame's longest waves live.  false: no such band.
bool mesh_band(const rpt_ctx *ctx, float aspect, int tile_rows, int &ty0, int &ty1) {
    if (!(ctx->first_tile == 0 && ctx->tile_step == 1 && ctx->run_log2 == 0 && ctx->rects.size() == (size_t)ctx->object_count)) return false;
    float v0 = 3e38f, v1 = -3e38f;
    const rpt_object *objs = (const rpt_object *)ctx->host_objects.data();
    for (int i = 0; i < ctx->object_count; i++)
        if (objs[i].type == RPT_MESH && ctx->rects[i].u0 <= ctx->rects[i].u1) {
            v0 = std::min(v0, ctx->rects[i].v0);
            v1 = std::max(v1, ctx->rects[i].v1);
        }
    if (!(v0 <= v1)) return false;
    const float H = (float)ctx->height;
    auto clampi = [](float x, int lo, int hi) { return x < (float)lo ? lo : (x > (float)hi ? hi : (int)x); };
    ty0 = clampi(std::floor((v0 + 0.5f) * H / 8.0f) - 1, 0, tile_rows - 1);
    ty1 = clampi(std::floor((v1 + 0.5f) * H / 8.0f) + 1, 0, tile_rows - 1);
    (void)aspect;
    return ty1 >= ty0;
}

#ifdef RPT_DIAGNOSTICS
// Persistent kernels (rpt_persistent.hip.h): claim geometry, counters, grid = what the chip holds at once.
int launch_persistent(rpt_ctx *ctx, rptd::KernelArgs &a, int tiles, int v) {
    a.tiles_x = (ctx->width + 7) / 8;
    a.runs_x = (a.tiles_x + RPT_SKY_RUN - 1) / RPT_SKY_RUN;
    a.first_ty = 0;
    a.first_h = 0;
    int ty0 = 0, ty1 = -1;
    static const bool no_band = std::getenv("RPT_NO_BAND") != nullptr;      // (experiment knob)
    if (!no_band && mesh_band(ctx, a.aspect, tiles, ty0, ty1)) { a.first_ty = ty0; a.first_h = ty1 - ty0 + 1; }
    const long long band = (long long)a.first_h * a.tiles_x, sky = (long long)(tiles - a.first_h) * a.runs_x;
    // index / tiles_x by multiplication with ceil(2^32 / tiles_x): exact while index * tiles_x < 2^32
    if (a.runs_x < 2 || (band + sky + 64) * a.tiles_x >= (1ll << 32)) return -1;     // (tiny or enormous frames: the caller takes the per-tile-launch kernel)
    a.tiles_x_magic = (unsigned int)((1ull << 32) / (unsigned long long)a.tiles_x + 1ull);
    a.runs_x_magic = (unsigned int)((1ull << 32) / (unsigned long long)a.runs_x + 1ull);
    a.band_tiles = (int)band;
    a.sky_runs = (int)sky;
    const size_t counter_bytes = (size_t)2 * RPT_CLAIM_QUEUES * RPT_CLAIM_STRIDE * sizeof(unsigned int);
    if (!ctx->claim_counters.ptr) {
        if (int rc = reserve(ctx, ctx->claim_counters, counter_bytes)) return rc;
        RPT_HIP(ctx, hipMemsetAsync(ctx->claim_counters.ptr, 0, counter_bytes, ctx->stream));
        ctx->claim_epoch = 0;
    }
    a.claim_set = (int)(ctx->claim_epoch++ & 1u);
    unsigned int *claims = (unsigned int *)ctx->claim_counters.ptr;
    const long long want = (band + sky + 3) / 4;
    static const int wgs_per_cu = std::getenv("RPT_WGS_PER_CU") ? std::atoi(std::getenv("RPT_WGS_PER_CU")) : 5;     // (experiment knob)
    const unsigned int wgs = (unsigned int)std::max(1ll, std::min((long long)ctx->cu_count * wgs_per_cu, want));
    switch (v) {
    case 60: hipLaunchKernelGGL(rptd::rpt_render_kernel_persistent_w5, dim3(wgs), dim3(256), 0, ctx->stream, a, a.out16, a.plane, a.debug_rgb, claims); break;
    case 62: hipLaunchKernelGGL(rptd::rpt_render_kernel_persistent_direct_w5, dim3(wgs), dim3(256), 0, ctx->stream, a, a.out16, a.plane, a.debug_rgb, claims); break;
    case 63: hipLaunchKernelGGL(rptd::rpt_render_kernel_persistent_classic_w5, dim3(wgs), dim3(256), 0, ctx->stream, a, a.out16, a.plane, a.debug_rgb, claims); break;
    default: return fail(ctx, RPT_ERR_ARG, "unknown persistent kernel variant");
    }
    return RPT_OK;
}
#endif

#ifdef RPT_DIAGNOSTICS
// librpt_hip_diag.so only: the measurement arms (rpt_diag_kernels.hip.h)
#define RPT_LAUNCH_X(N) case N: hipLaunchKernelGGL(rptd::rpt_render_kernel_x##N, grid, dim3(256), 0, ctx->stream, a); break;
int launch_diagnostic(rpt_ctx *ctx, rptd::KernelArgs &a, dim3 grid, int tiles, int v) {
    switch (v) {
    case 26: {   // round 1's default: per-tile object masks from a prepass kernel
        const int tiles_x = ((ctx->width + 31) / 32) * 4;          // tiles per row as the 32-pixel-wide blocks see them
        const int n_tiles = tiles_x * tiles;
        if (int rc = reserve(ctx, ctx->tile_masks, (size_t)n_tiles * 8)) return rc;
        a.mask_tiles_x = tiles_x;
        a.n_tiles = n_tiles;
        a.tile_masks = (unsigned long long *)ctx->tile_masks.ptr;
        hipLaunchKernelGGL(rptd::rpt_tile_bin_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, ctx->stream, a);
        hipLaunchKernelGGL(rptd::rpt_render_kernel_v1_masked_w5, grid, dim3(256), 0, ctx->stream, a);
        break;
    }
    case 40: hipLaunchKernelGGL(rptd::rpt_render_kernel_ballot_w4, dim3(grid.x * 4, grid.y), dim3(64), 0, ctx->stream, a); break;      // (kernel 41's body: one wave per workgroup)
    case 42: hipLaunchKernelGGL(rptd::rpt_render_kernel_ballot_w6, dim3(grid.x * 4, grid.y), dim3(64), 0, ctx->stream, a); break;
    case 141: hipLaunchKernelGGL(rptd::rpt_render_kernel_r02walk_w5, grid, dim3(256), 0, ctx->stream, a); break;
    case 143: hipLaunchKernelGGL(rptd::rpt_render_kernel_r02walk_first_w5, grid, dim3(256), 0, ctx->stream, a); break;
    case 60: case 62: case 63: {
        const int rc = launch_persistent(ctx, a, tiles, v);
        if (rc > 0) return rc;
        if (rc < 0) hipLaunchKernelGGL(rptd::rpt_render_kernel_ballot_w5, dim3(grid.x * 4, grid.y), dim3(64), 0, ctx->stream, a);     // (tiny or enormous frames: the product kernel)
        break;
    }
    case 61: hipLaunchKernelGGL(rptd::rpt_render_kernel_queue_w5, grid, dim3(256), 0, ctx->stream, a); break;
    RPT_LAUNCH_X(256) RPT_LAUNCH_X(257) RPT_LAUNCH_X(259) RPT_LAUNCH_X(261) RPT_LAUNCH_X(263) RPT_LAUNCH_X(265) RPT_LAUNCH_X(269)
    RPT_LAUNCH_X(273) RPT_LAUNCH_X(277) RPT_LAUNCH_X(285) RPT_LAUNCH_X(305) RPT_LAUNCH_X(317) RPT_LAUNCH_X(337) RPT_LAUNCH_X(349) RPT_LAUNCH_X(401) RPT_LAUNCH_X(785) RPT_LAUNCH_X(529) RPT_LAUNCH_X(541) RPT_LAUNCH_X(561) RPT_LAUNCH_X(573) RPT_LAUNCH_X(589) RPT_LAUNCH_X(605) RPT_LAUNCH_X(621) RPT_LAUNCH_X(625) RPT_LAUNCH_X(637) RPT_LAUNCH_X(641) RPT_LAUNCH_X(653) RPT_LAUNCH_X(593)
    case 1257: hipLaunchKernelGGL(rptd::rpt_render_kernel_x257_w6, grid, dim3(256), 0, ctx->stream, a); break;
    case 2257: hipLaunchKernelGGL(rptd::rpt_render_kernel_x257_w4, grid, dim3(256), 0, ctx->stream, a); break;
    case 2259: hipLaunchKernelGGL(rptd::rpt_render_kernel_x259_w4, grid, dim3(256), 0, ctx->stream, a); break;
    case 2263: hipLaunchKernelGGL(rptd::rpt_render_kernel_x263_w4, grid, dim3(256), 0, ctx->stream, a); break;
    case 657: hipLaunchKernelGGL(rptd::rpt_render_kernel_x657, dim3(grid.x * 4, grid.y), dim3(64), 0, ctx->stream, a); break;
    case 669: hipLaunchKernelGGL(rptd::rpt_render_kernel_x669, dim3(grid.x * 4, grid.y), dim3(64), 0, ctx->stream, a); break;
    case 673: hipLaunchKernelGGL(rptd::rpt_render_kernel_x673, dim3(grid.x * 4, grid.y), dim3(64), 0, ctx->stream, a); break;
    case 705: hipLaunchKernelGGL(rptd::rpt_render_kernel_x705, dim3(grid.x * 4, grid.y), dim3(64), 0, ctx->stream, a); break;      // kernel 41's walk without the repeated triangle tests
    case 717: hipLaunchKernelGGL(rptd::rpt_render_kernel_x717, dim3(grid.x * 4, grid.y), dim3(64), 0, ctx->stream, a); break;      // kernel 43's walk likewise
    RPT_LAUNCH_X(689) RPT_LAUNCH_X(701)          // kernel 41's / 43's walk launched four waves per workgroup (what the product did before)
    case 2573: hipLaunchKernelGGL(rptd::rpt_render_kernel_x573_w4, grid, dim3(256), 0, ctx->stream, a); break;      // the latency kernel at 4 waves per SIMD (128 VGPRs, no scratch)
    case 7:
        if (int rc = reserve(ctx, ctx->counters, 16 * sizeof(unsigned long long))) return rc;
        RPT_HIP(ctx, hipMemsetAsync(ctx->counters.ptr, 0, 16 * sizeof(unsigned long long), ctx->stream));
        a.counters = (unsigned long long *)ctx->counters.ptr;
        hipLaunchKernelGGL(rptd::rpt_render_kernel_v1_diag, grid, dim3(256), 0, ctx->stream, a);
        break;
    case 8: hipLaunchKernelGGL(rptd::rpt_render_kernel_primary_only, grid, dim3(256), 0, ctx->stream, a); break;
    case 11:
        if (int rc = reserve(ctx, ctx->wave_times, (size_t)grid.x * grid.y * 4 * 10 * sizeof(unsigned long long))) return rc;
        RPT_HIP(ctx, hipMemsetAsync(ctx->wave_times.ptr, 0, ctx->wave_times.bytes, ctx->stream));
        a.wave_times = (unsigned long long *)ctx->wave_times.ptr;
        hipLaunchKernelGGL(rptd::rpt_render_kernel_v1_timeline, grid, dim3(256), 0, ctx->stream, a);
        break;
    default: return fail(ctx, RPT_ERR_ARG, "unknown kernel variant");
    }
    return RPT_OK;
}
#endif

int launch(rpt_ctx *ctx) {
    if (!ctx->scene_uploaded) return fail(ctx, RPT_ERR_STATE, "rpt_render before rpt_upload_scene");
    if (!ctx->params_set) return fail(ctx, RPT_ERR_STATE, "rpt_render before rpt_set_params");
    if (int rc = ensure_outputs(ctx)) return rc;

    rptd::KernelArgs a;
    std::memset(&a, 0, sizeof a);
    a.dnodes = (const rptd::DNode *)ctx->geo->dnodes.ptr;
    a.dtris = (const rptd::DTri *)ctx->geo->dtris.ptr;
    a.links = (const int *)ctx->geo->dlinks.ptr;
    a.first_tris = (const rptd::DTri *)ctx->geo->dfirst.ptr;
    a.root_grids = (const int *)ctx->geo->dgrids.ptr;       // (null and 0 in the product library)
    a.seen_before = (const uint2 *)ctx->geo->dseen.ptr;     // (likewise)
    a.grid_roots = ctx->geo->grid_roots;
    a.top_count = ctx->geo->top_count;
    a.dobjs = (const rptd::DObj *)((const char *)ctx->objects.ptr + (size_t)ctx->object_count * sizeof(rpt_object));
    a.objects = (const rpt_object *)ctx->objects.ptr;
    a.rects = (const float4 *)((const char *)ctx->objects.ptr + (size_t)ctx->object_count * (sizeof(rpt_object) + sizeof(rptd::DObj)));
    a.vertices = (const rpt_float3 *)ctx->geo->vertices.ptr;
    a.normals = (const rpt_float3 *)ctx->geo->normals.ptr;
    a.uvs = (const rpt_float2 *)ctx->geo->uvs.ptr;
    a.triangles = (const uint32_t *)ctx->geo->triangles.ptr;
    a.octrees = (const rpt_octree *)ctx->geo->octrees.ptr;
    a.octreeTris = (const int32_t *)ctx->geo->octreeTris.ptr;
    a.textures = (const uint8_t *)ctx->geo->textures.ptr;
    a.texture_bytes = (long long)ctx->geo->textures.bytes;
    a.out16 = ctx->colour_plane ? nullptr : (rpt_pixel *)(ctx->external_out ? ctx->external_out : ctx->owned_out.ptr);
    a.plane = ctx->colour_plane ? (uint32_t *)(ctx->external_plane ? ctx->external_plane : ctx->owned_plane.ptr) : nullptr;
    a.debug_rgb = (float *)(ctx->external_rgb ? ctx->external_rgb : (ctx->want_owned_rgb ? ctx->owned_rgb.ptr : nullptr));
    for (int c = 0; c < 3; c++) a.hable_wp[c] = hable_host(ctx->white_point[c]);
    {   // background colour of opencl_kernel.cl:565 through the tonemap and pack of :649-657, once per frame
        const float bg[3] = {0.15f, 0.15f, 0.25f};
        uint32_t packed = 1u << 24;
        for (int c = 0; c < 3; c++) {
            const float m = hable_host(bg[c]) / a.hable_wp[c];
            a.bg_mapped[c] = 1.0f < m ? 1.0f : m;
            packed |= to_u8_host(a.bg_mapped[c]) << (8 * c);
        }
        a.bg_packed = packed;
    }
    a.ambient = ctx->ambient;
    a.aspect = (float)ctx->width / (float)ctx->height;
    a.inv_width = 1.0f / (float)ctx->width;
    a.inv_height = 1.0f / (float)ctx->height;
    // the diagonal slabs hold inside their window only (rpt_screen_bounds.hpp): frames up to 4 : 1, |v| <= 1/2 always
    a.diagonals = 0;
    if (0.5f * a.aspect + 2.0f * a.inv_height * a.aspect <= (float)rptb::DIAG_WINDOW_U && ctx->rects.size() == (size_t)ctx->object_count)
        for (const rptb::Rect &r : ctx->rects) a.diagonals |= rptb::has_diagonals(r) ? 1 : 0;
    a.object_count = ctx->object_count;
    a.width = ctx->width;
    a.height = ctx->height;
    a.interval = ctx->interval;
    a.first_tile = ctx->first_tile;
    a.tile_step = ctx->tile_step;
    a.run_log2 = ctx->run_log2;

    const int tiles = local_tile_count(ctx);
    if (tiles == 0) return RPT_OK;
    const dim3 grid((ctx->width + 31) / 32, tiles);         // the measurement arms: four waves (a 32 x 8 strip) per workgroup
    const dim3 grid1(((ctx->width + 31) / 32) * 4, tiles);  // the product kernels: one wave (an 8 x 8 tile) per workgroup
    // variant 0 = default.  rpt_render_async() is a throughput call (frames in flight fill each other's gaps): the in-wave
    // cull kernel in natural order (41).  The blocking rpt_render() is a latency call — its caller waits for this frame, and the
    // frame is as long as its longest wave — so the band of tile rows that holds the meshes is dispatched first and the walk asks
    // for its triangle records an iteration ahead (43).  A frame whose Object[] holds no mesh gets the kernel without the octree
    // walk (44): 8 waves per SIMD instead of 5.  An octree the derived layout cannot hold gets the general kernel (1).
    // Frames in flight that are too small to fill the chip with walks wait for latency as well (profiles/r03_latency_walk_ab.txt:
    // 43 ahead of 41 up to 1920x1080 — bunny -6 %, shadows -19 % — level at 2560x1440, behind from 3200x1800): 43 for those too.
    const bool small_frame = (size_t)tiles * RPT_TILE_ROWS * (size_t)ctx->width <= (size_t)RPT_LATENCY_KERNEL_MAX_PIXELS;
    int v = ctx->variant == 0 ? (!ctx->has_mesh ? 44 : (ctx->latency_call || small_frame ? 43 : 41)) : ctx->variant;
    if (v == 44 && ctx->has_mesh) v = 41;          // (asked for explicitly on a scene with meshes: the full kernel)
    if (!ctx->geo->compact_ok && v != 44) v = 1;
    // The per-object regions are PROVEN for the window |u| <= 2, |v| <= 1/2 (rpt_bounds_certify.hpp): a frame wider than 4 : 1 has
    // pixels outside it, and beyond 2^20 pixels a side a tile's 1.5-pixel skirt is no longer large against float rounding — such
    // frames are rendered by the un-culled kernel (same pixels, every object tested everywhere).
    const bool window_holds_frame = 0.5f * a.aspect <= (float)rptb::cert::WINDOW_U && ctx->width <= (1 << 20) && ctx->height <= (1 << 20);
    if (!window_holds_frame && (v == 41 || v == 43 || v == 44)) v = 3;
    const bool band_first = v == 43
#ifdef RPT_DIAGNOSTICS
                            || v == 143 || v == 2573 || (v >= 256 && v < 1000 && (v & 8))
#endif
        ;
    a.first_h = 0;
    if (band_first) {      // whole-frame contexts only: a band that does not already start the frame, at most half of it
        int ty0 = 0, ty1 = -1;
        if (mesh_band(ctx, a.aspect, tiles, ty0, ty1) && ty0 > 0 && (ty1 - ty0 + 1) * 2 < tiles) { a.first_ty = ty0; a.first_h = ty1 - ty0 + 1; }
    }
    a.msaa = ctx->msaa;
    if (ctx->msaa > 1) {       // MSAASAMPLES > 1: the multi-sample form of the default kernel (46), or of the un-culled one (47)
        if (v != 3 && v != 41 && v != 43 && v != 44) return fail(ctx, RPT_ERR_ARG, "rpt_set_msaa > 1 is implemented for the default kernels and variant 3 (derived octree layouts)");
        v = v == 3 ? 47 : 46;
    }
    switch (v) {
    case 46: hipLaunchKernelGGL(rptd::rpt_render_kernel_msaa_w5, grid1, dim3(64), 0, ctx->stream, a); break;
    case 47: hipLaunchKernelGGL(rptd::rpt_render_kernel_msaa_unculled_w5, grid1, dim3(64), 0, ctx->stream, a); break;
    case 1: hipLaunchKernelGGL(rptd::rpt_render_kernel_v0, grid1, dim3(64), 0, ctx->stream, a); break;
    case 3: hipLaunchKernelGGL(rptd::rpt_render_kernel_unculled_w5, grid1, dim3(64), 0, ctx->stream, a); break;
    case 41: hipLaunchKernelGGL(rptd::rpt_render_kernel_ballot_w5, grid1, dim3(64), 0, ctx->stream, a); break;
    case 43: hipLaunchKernelGGL(rptd::rpt_render_kernel_ballot_first_w5, grid1, dim3(64), 0, ctx->stream, a); break;
    case 44: hipLaunchKernelGGL(rptd::rpt_render_kernel_analytic_w8, grid1, dim3(64), 0, ctx->stream, a); break;
    case 50:
    case 51:
        if (rpt_launch_relaxed_kernel(v == 51 ? 6 : 5, &a, sizeof a, grid1.x, grid1.y, (void *)ctx->stream)) return fail(ctx, RPT_ERR_DEVICE, "relaxed-arithmetic kernel launch failed");
        break;
    default:
#ifdef RPT_DIAGNOSTICS
        if (int rc = launch_diagnostic(ctx, a, grid, tiles, v)) return rc;
        break;
#else
        return fail(ctx, RPT_ERR_ARG, "unknown kernel variant");
#endif
    }
    RPT_HIP(ctx, hipGetLastError());
    ctx->last_variant = v;
    return RPT_OK;
}

}  // namespace

extern "C" {

#ifdef RPT_DIAGNOSTICS
const char *rpt_version(void) { return "rpt-hip 0.2 (gfx950, diagnostics build)"; }
#else
const char *rpt_version(void) { return "rpt-hip 0.2 (gfx950)"; }
#endif

int rpt_create(rpt_ctx **out, int device_ordinal) {
    if (!out) return RPT_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RPT_ERR_DEVICE;
    if (device_ordinal < 0 || device_ordinal >= n) return RPT_ERR_ARG;
    rpt_ctx *ctx = new (std::nothrow) rpt_ctx();
    if (!ctx) return RPT_ERR_NOMEM;
    ctx->device = device_ordinal;
    (void)rpth::Workers::instance();       // the helper pool exists from the first context on, not from the first animated frame (rpt_workers.hpp: fork())
    {
        static std::atomic<int> created{0};
        ctx->serial = created.fetch_add(1);
    }
    ctx->geo = std::make_shared<Geometry>();
    ctx->geo->device = device_ordinal;
    if (hipSetDevice(device_ordinal) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||

        hipEventCreate(&ctx->ev_begin) != hipSuccess || hipEventCreate(&ctx->ev_end) != hipSuccess) {
        rpt_destroy(ctx);
        return RPT_ERR_DEVICE;
    }
    ctx->stream = ctx->own_stream;
    if (hipDeviceGetAttribute(&ctx->cu_count, hipDeviceAttributeMultiprocessorCount, device_ordinal) != hipSuccess || ctx->cu_count <= 0) ctx->cu_count = 256;
    *out = ctx;
    return RPT_OK;
}

int rpt_create_multi(rpt_ctx **out, const int *device_ordinals, int n) {
    if (!out || !device_ordinals || n <= 0) return RPT_ERR_ARG;
    for (int i = 0; i < n; i++) out[i] = nullptr;
    for (int i = 0; i < n; i++) {
        const int rc = rpt_create(&out[i], device_ordinals[i]);
        if (rc != RPT_OK) {
            for (int k = 0; k < i; k++) { rpt_destroy(out[k]); out[k] = nullptr; }
            return rc;
        }
    }
    return RPT_OK;
}

void rpt_destroy(rpt_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    // work of this context may still be running on an external stream (rpt_set_stream) whose handle the caller may already
    // have destroyed: wait for the device rather than for a handle that may be dead, then free
    (void)hipDeviceSynchronize();
#ifdef RPT_DIAGNOSTICS
    if (ctx->host_calls > 0 && getenv("RPT_HOST_PROFILE"))
        fprintf(stderr, "[rpt] context %d: rpt_set_objects x %ld, us per call: wait for the staging slot %.1f, copy + DObj %.1f, screen bounds %.1f, hipMemcpyAsync %.1f, hipEventRecord %.1f\n",
                ctx->serial, ctx->host_calls, ctx->host_us[1] / ctx->host_calls, ctx->host_us[2] / ctx->host_calls, ctx->host_us[3] / ctx->host_calls,
                ctx->host_us[4] / ctx->host_calls, ctx->host_us[5] / ctx->host_calls);
#endif
    ctx->geo.reset();
    for (DeviceBuffer *b : {&ctx->objects, &ctx->dobjs, &ctx->counters, &ctx->wave_times, &ctx->tile_masks, &ctx->claim_counters, &ctx->verify_planes, &ctx->owned_out, &ctx->owned_plane, &ctx->owned_rgb})
        release(*b);
    if (ctx->pinned_objects) (void)hipHostFree(ctx->pinned_objects);
    for (hipEvent_t e : ctx->staging_done) if (e) (void)hipEventDestroy(e);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    for (hipEvent_t e : ctx->timing_events) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char *rpt_last_error(const rpt_ctx *ctx) { return ctx ? ctx->error.c_str() : "null context"; }

int rpt_upload_scene(rpt_ctx *ctx, const rpt_scene_desc *s) {
    if (!ctx || !s) return RPT_ERR_ARG;
    if ((s->object_count && !s->objects) || (s->vertex_count && !s->vertices) || (s->normal_count && !s->normals) ||
        (s->uv_count && !s->uvs) || (s->triangle_words && !s->triangles) || (s->octree_count && !s->octrees) ||
        (s->octree_tri_count && !s->octreeTris) || (s->texture_bytes && !s->textures))
        return fail(ctx, RPT_ERR_ARG, "rpt_upload_scene: null pointer with nonzero count");
    if (s->object_count > (1u << 20)) return fail(ctx, RPT_ERR_ARG, "rpt_upload_scene: too many objects");
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc = validate_geometry(ctx, *s)) return rc;
    RPT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->scene_uploaded = false;
    // a fresh Geometry: contexts sharing the previous one (rpt_share_scene) keep it until they let go
    ctx->geo = std::make_shared<Geometry>();
    ctx->geo->device = ctx->device;
    {   // (an address can be reused by a later Geometry; a generation cannot)
        static std::atomic<unsigned long long> next_generation{1};
        ctx->geo->generation = next_generation.fetch_add(1);
    }
    ctx->rects.clear();
    if (int rc = upload(ctx, ctx->geo->vertices, s->vertices, s->vertex_count * sizeof(rpt_float3))) return rc;
    if (int rc = upload(ctx, ctx->geo->normals, s->normals, s->normal_count * sizeof(rpt_float3))) return rc;
    if (int rc = upload(ctx, ctx->geo->uvs, s->uvs, s->uv_count * sizeof(rpt_float2))) return rc;
    if (int rc = upload(ctx, ctx->geo->triangles, s->triangles, s->triangle_words * sizeof(uint32_t))) return rc;
    if (int rc = upload(ctx, ctx->geo->octrees, s->octrees, s->octree_count * sizeof(rpt_octree))) return rc;
    if (int rc = upload(ctx, ctx->geo->octreeTris, s->octreeTris, s->octree_tri_count * sizeof(int32_t))) return rc;
    if (int rc = upload(ctx, ctx->geo->textures, s->textures, s->texture_bytes)) return rc;
    ctx->geo->vertex_count = s->vertex_count;
    ctx->geo->normal_count = s->normal_count;
    ctx->geo->uv_count = s->uv_count;
    ctx->geo->triangle_words = s->triangle_words;
    ctx->geo->octree_count = s->octree_count;
    ctx->geo->octree_tri_count = s->octree_tri_count;
    if (int rc = build_derived_geometry(ctx, *s)) return rc;
    ctx->geo->host_node_bounds.resize(s->octree_count * 6);
    for (size_t i = 0; i < s->octree_count; i++) {
        float *b = &ctx->geo->host_node_bounds[6 * i];
        b[0] = s->octrees[i].min.x; b[1] = s->octrees[i].min.y; b[2] = s->octrees[i].min.z;
        b[3] = s->octrees[i].max.x; b[4] = s->octrees[i].max.y; b[5] = s->octrees[i].max.z;
    }
    // Which nodes hold all of their triangles inside their own box?  (Asked of mesh roots by the shadow-segment cull: a hit on a
    // triangle is a point of that triangle.)  Indices were validated above.
    // ... and how large are those triangles?  K = max |e1| |e2|, L = the longest edge, with e1 = fl(B - A), e2 = fl(C - A) as the walk
    // uses them: the float error of an accepted triangle hit is bounded through them (mesh_segment_apart, rpt_kernels.hip.h).
    ctx->geo->node_holds_its_triangles.assign(s->octree_count, 0);
    ctx->geo->node_tri_K.assign(s->octree_count, 0.0f);
    ctx->geo->node_tri_L.assign(s->octree_count, 0.0f);
    for (size_t i = 0; i < s->octree_count; i++) {
        const rpt_octree &o = s->octrees[i];
        bool inside = o.trisCount >= 0;
        double K = 0.0, L = 0.0;
        for (int k = o.trisIndex; inside && k < o.trisIndex + o.trisCount; k++) {
            const int t = s->octreeTris[k];
            const rpt_float3 *vv[3];
            for (int c = 0; c < 3 && inside; c++) {
                const rpt_float3 &v = s->vertices[s->triangles[9 * t + 3 * c]];
                vv[c] = &v;
                inside = v.x >= o.min.x && v.x <= o.max.x && v.y >= o.min.y && v.y <= o.max.y && v.z >= o.min.z && v.z <= o.max.z;
            }
            if (!inside) break;
            const float e1[3] = {vv[1]->x - vv[0]->x, vv[1]->y - vv[0]->y, vv[1]->z - vv[0]->z}, e2[3] = {vv[2]->x - vv[0]->x, vv[2]->y - vv[0]->y, vv[2]->z - vv[0]->z};
            const double l1 = std::sqrt((double)e1[0] * e1[0] + (double)e1[1] * e1[1] + (double)e1[2] * e1[2]);
            const double l2 = std::sqrt((double)e2[0] * e2[0] + (double)e2[1] * e2[1] + (double)e2[2] * e2[2]);
            K = std::max(K, l1 * l2);
            L = std::max(L, std::max(l1, l2));
        }
        ctx->geo->node_holds_its_triangles[i] = inside ? 1 : 0;
        ctx->geo->node_tri_K[i] = std::nextafter((float)K, INFINITY);
        ctx->geo->node_tri_L[i] = std::nextafter((float)L, INFINITY);
    }
    ctx->scene_uploaded = true;
    const int rc = rpt_set_objects(ctx, s->objects, (int)s->object_count);
    if (rc) ctx->scene_uploaded = false;
    return rc;
}

int rpt_share_scene(rpt_ctx *ctx, rpt_ctx *owner) {
    if (!ctx || !owner || ctx == owner) return RPT_ERR_ARG;
    if (!owner->scene_uploaded) return fail(ctx, RPT_ERR_STATE, "rpt_share_scene: the owner has no scene");
    if (ctx->device != owner->device) return fail(ctx, RPT_ERR_ARG, "rpt_share_scene: contexts are on different devices");
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    RPT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->geo = owner->geo;
    ctx->scene_uploaded = true;
    const int count = (int)(owner->host_objects.size() / sizeof(rpt_object));
    const int rc = rpt_set_objects(ctx, count ? owner->host_objects.data() : nullptr, count);
    if (rc) ctx->scene_uploaded = false;
    return rc;
}

int rpt_set_objects(rpt_ctx *ctx, const void *objects, int count) {
    if (!ctx || count < 0 || (count && !objects)) return RPT_ERR_ARG;
    if (!ctx->scene_uploaded) return fail(ctx, RPT_ERR_STATE, "rpt_set_objects before rpt_upload_scene");
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc = validate_objects(ctx, (const rpt_object *)objects, count)) return rc;
    const size_t bytes = (size_t)count * sizeof(rpt_object);
    const size_t dbytes = (size_t)count * (sizeof(rptd::DObj) + sizeof(rptb::Rect));     // DObj[count] then Rect[count]
    // staging ring of pinned slots: the copy of frame k may still be in flight when frame k+1 is staged, so
    // each slot has an event and is reused only once its own transfer has completed (no per-frame stream sync)
    const size_t slot_bytes = ((bytes + dbytes + 255) / 256) * 256;
    if (slot_bytes * RPT_STAGING_SLOTS > ctx->pinned_capacity) {
        RPT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->pinned_objects) RPT_HIP(ctx, hipHostFree(ctx->pinned_objects));
        ctx->pinned_objects = nullptr;
        ctx->pinned_capacity = 0;
        const size_t cap = (slot_bytes < 4096 ? 4096 : slot_bytes * 2) * RPT_STAGING_SLOTS;
        RPT_HIP(ctx, hipHostMalloc(&ctx->pinned_objects, cap, hipHostMallocDefault));
        ctx->pinned_capacity = cap;
        for (int k = 0; k < RPT_STAGING_SLOTS; k++)
            if (!ctx->staging_done[k]) RPT_HIP(ctx, hipEventCreateWithFlags(&ctx->staging_done[k], hipEventDisableTiming));
        ctx->staging_used = 0;
    }
    // Object[] and DObj[] live back to back in one device buffer: one transfer per frame
    if (bytes + dbytes > ctx->objects.capacity) RPT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (int rc = reserve(ctx, ctx->objects, bytes + dbytes)) return rc;
    if (bytes) {
        const int k = (int)(ctx->staging_next % RPT_STAGING_SLOTS);
        RPT_HOST_MARK(0);
        if (ctx->staging_used & (1u << k)) RPT_HIP(ctx, hipEventSynchronize(ctx->staging_done[k]));
        RPT_HOST_MARK(1);
        char *slot = (char *)ctx->pinned_objects + (ctx->pinned_capacity / RPT_STAGING_SLOTS) * k;
        std::memcpy(slot, objects, bytes);
        build_dobjs(ctx, (const rpt_object *)objects, count, (rptd::DObj *)(slot + bytes));
        RPT_HOST_MARK(2);
        build_rects(ctx, (const rpt_object *)objects, count, (rptb::Rect *)(slot + bytes + (size_t)count * sizeof(rptd::DObj)));
        RPT_HOST_MARK(3);
        RPT_HIP(ctx, hipMemcpyAsync(ctx->objects.ptr, slot, bytes + dbytes, hipMemcpyHostToDevice, ctx->stream));
        RPT_HOST_MARK(4);
        RPT_HIP(ctx, hipEventRecord(ctx->staging_done[k], ctx->stream));
        RPT_HOST_MARK(5);
        ctx->last_event = ctx->staging_done[k];
        ctx->staging_used |= 1u << k;
        ctx->staging_next++;
    }
    ctx->object_count = count;
    ctx->has_mesh = false;
    for (int i = 0; i < count; i++) ctx->has_mesh = ctx->has_mesh || ((const rpt_object *)objects)[i].type == RPT_MESH;
    ctx->host_objects.assign((const uint8_t *)objects, (const uint8_t *)objects + bytes);
    return RPT_OK;
}

int rpt_set_params(rpt_ctx *ctx, const float white_point[3], float ambient, int width, int height, int interval) {
    if (!ctx || !white_point) return RPT_ERR_ARG;
    if (width <= 0 || height <= 0 || (long long)width * height > (1ll << 31) - 1) return fail(ctx, RPT_ERR_ARG, "rpt_set_params: bad resolution");
    for (int c = 0; c < 3; c++) ctx->white_point[c] = white_point[c];
    ctx->ambient = ambient;
    ctx->width = width;
    ctx->height = height;
    const bool interval_changed = ctx->interval != interval;
    ctx->interval = interval;
    ctx->params_set = true;
    if (interval_changed && ctx->scene_uploaded && ctx->object_count > 0) {
        const std::vector<uint8_t> copy = ctx->host_objects;      // the culling record of DObj uses `interval`
        return rpt_set_objects(ctx, copy.data(), ctx->object_count);
    }
    return RPT_OK;
}

int rpt_set_output(rpt_ctx *ctx, void *device_ptr_or_null) {
    if (!ctx) return RPT_ERR_ARG;
    ctx->external_out = device_ptr_or_null;
    return RPT_OK;
}

int rpt_set_rows(rpt_ctx *ctx, int first_tile, int tile_step, int colour_plane) {
    return rpt_set_tile_pattern(ctx, first_tile, tile_step, 1, colour_plane);
}

int rpt_set_tile_pattern(rpt_ctx *ctx, int first_tile, int tile_step, int run, int colour_plane) {
    if (!ctx || first_tile < 0 || tile_step < 1 || run < 1 || run > tile_step || (run & (run - 1))) return RPT_ERR_ARG;
    int lg = 0;
    while ((1 << lg) < run) lg++;
    ctx->first_tile = first_tile;
    ctx->tile_step = tile_step;
    ctx->run_log2 = lg;
    ctx->colour_plane = colour_plane != 0;
    return RPT_OK;
}

int rpt_set_stream(rpt_ctx *ctx, void *hip_stream) {
    if (!ctx) return RPT_ERR_ARG;
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    // drain what this context has enqueued on the stream it is leaving.  If that is an external stream the caller has
    // destroyed in the meantime, the handle is dead: the context's last launch is then waited for through its event.
    if (ctx->stream == ctx->own_stream) RPT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    else if (ctx->last_event) RPT_HIP(ctx, hipEventSynchronize(ctx->last_event));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return RPT_OK;
}

int rpt_set_debug_rgb(rpt_ctx *ctx, void *p) {
    if (!ctx) return RPT_ERR_ARG;
    ctx->want_owned_rgb = (p == (void *)1);
    ctx->external_rgb = ctx->want_owned_rgb ? nullptr : p;
    return RPT_OK;
}

int rpt_object_screen_rect(const void *object, int interval, const float *root_bounds_or_null, float rect_out[4]) {
    if (!object || !rect_out) return RPT_ERR_ARG;
    const rptb::Rect r = rptb::certified_object_rect(*(const rpt_object *)object, interval, root_bounds_or_null);
    rect_out[0] = r.u0; rect_out[1] = r.v0; rect_out[2] = r.u1; rect_out[3] = r.v1;
    return RPT_OK;
}

int rpt_object_screen_bounds(const void *object, int interval, const float *root_bounds_or_null, float bounds_out[8]) {
    if (!object || !bounds_out) return RPT_ERR_ARG;
    const rptb::Rect r = rptb::certified_object_rect(*(const rpt_object *)object, interval, root_bounds_or_null);
    const float v[8] = {r.u0, r.v0, r.u1, r.v1, r.p_lo, r.p_hi, r.m_lo, r.m_hi};
    for (int k = 0; k < 8; k++) bounds_out[k] = v[k];
    return RPT_OK;
}

int rpt_mesh_segment_cull_record(rpt_ctx *ctx, int object_index, float out[10]) {
    if (!ctx || !out || object_index < 0 || (size_t)(object_index + 1) * sizeof(rpt_object) > ctx->host_objects.size()) return RPT_ERR_ARG;
    const rpt_object *objs = (const rpt_object *)ctx->host_objects.data();
    std::vector<rptd::DObj> d((size_t)object_index + 1);
    build_dobjs(ctx, objs, object_index + 1, d.data());
    const rptd::DObj &r = d[(size_t)object_index];
    const int mi = objs[object_index].meshIndex;
    const bool known = objs[object_index].type == RPT_MESH && mi >= 0 && (size_t)mi < ctx->geo->node_tri_K.size();
    for (int k = 0; k < 3; k++) out[k] = r.mh[k];
    out[3] = r.mconst; out[4] = r.mslope; out[5] = r.mcw; out[6] = r.ms0;
    out[7] = known ? ctx->geo->node_tri_K[(size_t)mi] : 0.0f;
    out[8] = known ? ctx->geo->node_tri_L[(size_t)mi] : 0.0f;
    out[9] = r.mesh_in_box;
    return RPT_OK;
}

int rpt_object_screen_bounds_proposed(const void *object, int interval, const float *root_bounds_or_null, float bounds_out[8]) {
    if (!object || !bounds_out) return RPT_ERR_ARG;
    const rptb::Rect r = rptb::proposed_object_rect(*(const rpt_object *)object, interval, root_bounds_or_null);
    const float v[8] = {r.u0, r.v0, r.u1, r.v1, r.p_lo, r.p_hi, r.m_lo, r.m_hi};
    for (int k = 0; k < 8; k++) bounds_out[k] = v[k];
    return RPT_OK;
}

int rpt_certify_screen_bounds(const void *object, int interval, const float *root_bounds_or_null, const float bounds[8], int stats_out[4]) {
    if (!object || !bounds) return RPT_ERR_ARG;
    const rptb::Rect r{bounds[0], bounds[1], bounds[2], bounds[3], bounds[4], bounds[5], bounds[6], bounds[7]};
    rptb::cert::Stats st{0, 0, 0, 0};
    const bool ok = rptb::cert::certify(*(const rpt_object *)object, interval, root_bounds_or_null, r, &st);
    if (stats_out) { stats_out[0] = st.reason; stats_out[1] = st.tests; stats_out[2] = st.max_depth; stats_out[3] = st.segments; }
    return ok ? 1 : 0;
}

int rpt_set_variant(rpt_ctx *ctx, int variant) {
    if (!ctx) return RPT_ERR_ARG;
    switch (variant) {
    case 0: case 1: case 3: case 41: case 43: case 44: case 50: case 51: break;
    default:
#ifdef RPT_DIAGNOSTICS
        break;       // the diagnostics library knows many more (rpt_diag_kernels.hip.h); an unknown number fails at the launch
#else
        return fail(ctx, RPT_ERR_ARG, "rpt_set_variant: unknown variant (measurement arms and instrumented kernels exist in librpt_hip_diag.so only)");
#endif
    }
    ctx->variant = variant;
    return RPT_OK;
}

int rpt_last_variant(const rpt_ctx *ctx) { return ctx ? ctx->last_variant : 0; }

int rpt_set_msaa(rpt_ctx *ctx, int samples_per_axis) {
    if (!ctx) return RPT_ERR_ARG;
    if (samples_per_axis < 1 || samples_per_axis > 8) return fail(ctx, RPT_ERR_ARG, "rpt_set_msaa: 1..8 samples per axis");
    ctx->msaa = samples_per_axis;
    return RPT_OK;
}

int rpt_verify_frame(rpt_ctx *ctx, unsigned long long *differing_pixels) {
    if (!ctx || !differing_pixels) return RPT_ERR_ARG;
    if (!ctx->scene_uploaded) return fail(ctx, RPT_ERR_STATE, "rpt_verify_frame before rpt_upload_scene");
    if (!ctx->params_set) return fail(ctx, RPT_ERR_STATE, "rpt_verify_frame before rpt_set_params");
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t words = (size_t)local_tile_count(ctx) * RPT_TILE_ROWS * ctx->width;
    if (int rc = reserve(ctx, ctx->verify_planes, 2 * words * 4 + 8)) return rc;
    uint32_t *plane_a = (uint32_t *)ctx->verify_planes.ptr, *plane_b = plane_a + words;
    unsigned long long *count = (unsigned long long *)(plane_b + words);
    RPT_HIP(ctx, hipMemsetAsync(plane_a, 0, 2 * words * 4 + 8, ctx->stream));     // (rows of the last tile beyond the frame's height are never written)
    // render twice into the scratch planes: the kernel a frame would get, then the un-culled one; the context's own outputs,
    // its variant and its debug hook are put back whatever happens
    struct Saved { void *plane, *rgb; bool colour_plane, want_rgb; int variant; } saved = {ctx->external_plane, ctx->external_rgb, ctx->colour_plane, ctx->want_owned_rgb, ctx->variant};
    ctx->colour_plane = true;
    ctx->external_rgb = nullptr;
    ctx->want_owned_rgb = false;
    int rc = RPT_OK;
    ctx->external_plane = plane_a;
    rc = launch(ctx);
    const int verified_variant = ctx->last_variant;      // what rpt_last_variant reports afterwards: the kernel that was CHECKED, not the un-culled one
    if (rc == RPT_OK) {
        ctx->external_plane = plane_b;
        ctx->variant = 3;
        rc = launch(ctx);
    }
    ctx->last_variant = verified_variant;
    ctx->external_plane = saved.plane; ctx->external_rgb = saved.rgb; ctx->colour_plane = saved.colour_plane; ctx->want_owned_rgb = saved.want_rgb; ctx->variant = saved.variant;
    if (rc != RPT_OK) return rc;
    if (words) {
        const unsigned int blocks = (unsigned int)std::min<size_t>((words + 255) / 256, 4096);
        hipLaunchKernelGGL(rptd::rpt_count_differences_kernel, dim3(blocks), dim3(256), 0, ctx->stream, plane_a, plane_b, words, count);
        RPT_HIP(ctx, hipGetLastError());
    }
    RPT_HIP(ctx, hipMemcpyAsync(differing_pixels, count, 8, hipMemcpyDeviceToHost, ctx->stream));
    RPT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RPT_OK;
}

int rpt_render_async(rpt_ctx *ctx) {
    if (!ctx) return RPT_ERR_ARG;
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    hipEvent_t eb = ctx->ev_begin, ee = ctx->ev_end;
    const bool timed = ctx->timing_frames >= 0 && (size_t)(2 * ctx->timing_frames + 1) < ctx->timing_events.size();
    if (timed) {
        eb = ctx->timing_events[2 * ctx->timing_frames];
        ee = ctx->timing_events[2 * ctx->timing_frames + 1];
    }
    RPT_HIP(ctx, hipEventRecord(eb, ctx->stream));
    if (int rc = launch(ctx)) return rc;
    RPT_HIP(ctx, hipEventRecord(ee, ctx->stream));
    ctx->last_event = ee;
    if (timed) ctx->timing_frames++;
    else if (ctx->timing_frames >= 0) return fail(ctx, RPT_ERR_STATE, "timing region is full");
    ctx->frame_rendered = true;
    return RPT_OK;
}

int rpt_timing_begin(rpt_ctx *ctx, int max_frames) {
    if (!ctx || max_frames <= 0 || max_frames > (1 << 20)) return RPT_ERR_ARG;
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    while (ctx->timing_events.size() < (size_t)(2 * max_frames)) {
        hipEvent_t e = nullptr;
        RPT_HIP(ctx, hipEventCreate(&e));
        ctx->timing_events.push_back(e);
    }
    ctx->timing_frames = 0;
    return RPT_OK;
}

int rpt_timing_end(rpt_ctx *ctx, float *total_ms, int *frames) {
    if (!ctx || !total_ms || !frames) return RPT_ERR_ARG;
    if (ctx->timing_frames < 0) return fail(ctx, RPT_ERR_STATE, "rpt_timing_end without rpt_timing_begin");
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    float sum = 0.0f;
    for (int i = 0; i < ctx->timing_frames; i++) {
        float ms = 0.0f;
        RPT_HIP(ctx, hipEventSynchronize(ctx->timing_events[2 * i + 1]));
        RPT_HIP(ctx, hipEventElapsedTime(&ms, ctx->timing_events[2 * i], ctx->timing_events[2 * i + 1]));
        sum += ms;
    }
    *total_ms = sum;
    *frames = ctx->timing_frames;
    ctx->timing_frames = -1;
    return RPT_OK;
}

int rpt_timing_end_frames(rpt_ctx *ctx, float *per_frame_ms, int capacity, int *frames) {
    if (!ctx || !per_frame_ms || capacity < 0 || !frames) return RPT_ERR_ARG;
    if (ctx->timing_frames < 0) return fail(ctx, RPT_ERR_STATE, "rpt_timing_end_frames without rpt_timing_begin");
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    const int n = ctx->timing_frames < capacity ? ctx->timing_frames : capacity;
    for (int i = 0; i < n; i++) {
        RPT_HIP(ctx, hipEventSynchronize(ctx->timing_events[2 * i + 1]));
        RPT_HIP(ctx, hipEventElapsedTime(&per_frame_ms[i], ctx->timing_events[2 * i], ctx->timing_events[2 * i + 1]));
    }
    *frames = n;
    ctx->timing_frames = -1;
    return RPT_OK;
}

int rpt_timing_end_spans(rpt_ctx *ctx, const rpt_ctx *base, float *begin_ms, float *end_ms, int capacity, int *frames) {
    if (!ctx || !base || !begin_ms || !end_ms || capacity < 0 || !frames) return RPT_ERR_ARG;
    if (ctx->timing_frames < 0) return fail(ctx, RPT_ERR_STATE, "rpt_timing_end_spans without rpt_timing_begin");
    if (base->timing_events.empty() || base->device != ctx->device) return fail(ctx, RPT_ERR_ARG, "rpt_timing_end_spans: the base context has no timing events on this device");
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    const int n = ctx->timing_frames < capacity ? ctx->timing_frames : capacity;
    RPT_HIP(ctx, hipEventSynchronize(base->timing_events[0]));
    for (int i = 0; i < n; i++) {
        RPT_HIP(ctx, hipEventSynchronize(ctx->timing_events[2 * i + 1]));
        RPT_HIP(ctx, hipEventElapsedTime(&begin_ms[i], base->timing_events[0], ctx->timing_events[2 * i]));
        RPT_HIP(ctx, hipEventElapsedTime(&end_ms[i], base->timing_events[0], ctx->timing_events[2 * i + 1]));
    }
    *frames = n;
    ctx->timing_frames = -1;
    return RPT_OK;
}

int rpt_set_plane_output(rpt_ctx *ctx, void *device_ptr_or_null) {
    if (!ctx) return RPT_ERR_ARG;
    ctx->external_plane = device_ptr_or_null;
    return RPT_OK;
}

int rpt_sync(rpt_ctx *ctx) {
    if (!ctx) return RPT_ERR_ARG;
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    RPT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RPT_OK;
}

int rpt_render(rpt_ctx *ctx) {
    if (!ctx) return RPT_ERR_ARG;
    ctx->latency_call = true;
    const int rc = rpt_render_async(ctx);
    ctx->latency_call = false;
    if (rc) return rc;
    return rpt_sync(ctx);
}

void *rpt_output_ptr(rpt_ctx *ctx) {
    if (!ctx) return nullptr;
    return ctx->external_out ? ctx->external_out : ctx->owned_out.ptr;
}

size_t rpt_output_bytes(rpt_ctx *ctx) { return ctx ? (size_t)ctx->width * ctx->height * 16 : 0; }

void *rpt_colour_plane_ptr(rpt_ctx *ctx) { return ctx ? (ctx->external_plane ? ctx->external_plane : ctx->owned_plane.ptr) : nullptr; }

int rpt_read_framebuffer(rpt_ctx *ctx, void *host_dst, size_t bytes) {
    if (!ctx || !host_dst) return RPT_ERR_ARG;
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    const void *src = ctx->colour_plane ? rpt_colour_plane_ptr(ctx) : rpt_output_ptr(ctx);
    const size_t have = ctx->colour_plane ? (size_t)local_tile_count(ctx) * RPT_TILE_ROWS * ctx->width * 4 : rpt_output_bytes(ctx);
    if (!src || !ctx->frame_rendered) return fail(ctx, RPT_ERR_STATE, "rpt_read_framebuffer: nothing rendered yet");
    if (bytes > have) return fail(ctx, RPT_ERR_ARG, "rpt_read_framebuffer: more bytes requested than the framebuffer holds");
    RPT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RPT_HIP(ctx, hipMemcpy(host_dst, src, bytes, hipMemcpyDeviceToHost));
    return RPT_OK;
}

int rpt_read_debug_rgb(rpt_ctx *ctx, void *host_dst, size_t bytes) {
    if (!ctx || !host_dst) return RPT_ERR_ARG;
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    const void *src = ctx->external_rgb ? ctx->external_rgb : (ctx->want_owned_rgb ? ctx->owned_rgb.ptr : nullptr);
    if (!src || !ctx->frame_rendered) return fail(ctx, RPT_ERR_STATE, "rpt_read_debug_rgb: debug RGB is not enabled or nothing rendered");
    if (bytes > (size_t)ctx->width * ctx->height * 12) return fail(ctx, RPT_ERR_ARG, "rpt_read_debug_rgb: too many bytes");
    RPT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RPT_HIP(ctx, hipMemcpy(host_dst, src, bytes, hipMemcpyDeviceToHost));
    return RPT_OK;
}

int rpt_last_frame_ms(rpt_ctx *ctx, float *ms) {
    if (!ctx || !ms) return RPT_ERR_ARG;
    if (!ctx->frame_rendered) return fail(ctx, RPT_ERR_STATE, "rpt_last_frame_ms: nothing rendered yet");
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    RPT_HIP(ctx, hipEventSynchronize(ctx->ev_end));
    RPT_HIP(ctx, hipEventElapsedTime(&ctx->last_ms, ctx->ev_begin, ctx->ev_end));
    *ms = ctx->last_ms;
    return RPT_OK;
}

int rpt_timed_frames(rpt_ctx *ctx, int frames, float *avg_ms) {
    if (!ctx || frames <= 0 || !avg_ms) return RPT_ERR_ARG;
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    for (int w = 0; w < 2; w++) {          // untimed: allocate outputs, warm caches, measure the row costs
        if (int rc = launch(ctx)) return rc;
        RPT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    RPT_HIP(ctx, hipEventRecord(ctx->ev_begin, ctx->stream));
    for (int i = 0; i < frames; i++)
        if (int rc = launch(ctx)) return rc;
    RPT_HIP(ctx, hipEventRecord(ctx->ev_end, ctx->stream));
    RPT_HIP(ctx, hipEventSynchronize(ctx->ev_end));
    float ms = 0;
    RPT_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
    ctx->frame_rendered = true;
    ctx->last_ms = ms / frames;
    *avg_ms = ctx->last_ms;
    return RPT_OK;
}

int rpt_scatter_colour_plane_on(rpt_ctx *ctx, void *hip_stream, const void *planes, void *out16, int width, int height,
                                int n_ranks, int plane_stride_words);

int rpt_scatter_colour_plane(rpt_ctx *ctx, const void *planes, void *out16, int width, int height, int n_ranks,
                             int plane_stride_words, int reserved) {
    return rpt_scatter_colour_plane_on(ctx, ctx ? (void *)ctx->stream : nullptr, planes, out16, width, height, n_ranks, plane_stride_words);
}

int rpt_scatter_colour_plane_on(rpt_ctx *ctx, void *hip_stream, const void *planes, void *out16, int width, int height,
                                int n_ranks, int plane_stride_words) {
    if (!ctx || !planes || !out16 || width <= 0 || height <= 0 || n_ranks <= 0 || plane_stride_words < 0) return RPT_ERR_ARG;
    const int tiles = (height + RPT_TILE_ROWS - 1) / RPT_TILE_ROWS;
    const long long need = (long long)((tiles + n_ranks - 1) / n_ranks) * RPT_TILE_ROWS * width;
    if ((long long)plane_stride_words < need) return fail(ctx, RPT_ERR_ARG, "rpt_scatter_colour_plane: plane stride smaller than one rank's plane");
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    const dim3 grid((width + 255) / 256, height);
    hipLaunchKernelGGL(rptd::rpt_scatter_plane_kernel, grid, dim3(256), 0, hip_stream ? (hipStream_t)hip_stream : ctx->stream, (const uint32_t *)planes,
                       (rpt_pixel *)out16, width, height, n_ranks, (size_t)plane_stride_words);
    RPT_HIP(ctx, hipGetLastError());
    return RPT_OK;
}

int rpt_pack_colour_plane3_on(rpt_ctx *ctx, void *hip_stream, const void *plane4, void *plane3, size_t pixels) {
    if (!ctx || !plane4 || !plane3) return RPT_ERR_ARG;
    if (pixels % 4) return fail(ctx, RPT_ERR_ARG, "rpt_pack_colour_plane3: pixel count is not a multiple of 4");
    if (pixels == 0) return RPT_OK;
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t quads = pixels / 4;
    hipLaunchKernelGGL(rptd::rpt_pack_plane3_kernel, dim3((unsigned int)((quads + 255) / 256)), dim3(256), 0,
                       hip_stream ? (hipStream_t)hip_stream : ctx->stream, (const uint4 *)plane4, (uint32_t *)plane3, quads);
    RPT_HIP(ctx, hipGetLastError());
    return RPT_OK;
}

int rpt_scatter_colour_plane3_on(rpt_ctx *ctx, void *hip_stream, const void *planes3, void *out16, int width, int height,
                                 int n_ranks, size_t plane_stride_bytes) {
    if (!ctx || !planes3 || !out16 || width <= 0 || height <= 0 || n_ranks <= 0) return RPT_ERR_ARG;
    const int tiles = (height + RPT_TILE_ROWS - 1) / RPT_TILE_ROWS;
    const unsigned long long need = 3ull * (unsigned long long)((tiles + n_ranks - 1) / n_ranks) * RPT_TILE_ROWS * (unsigned long long)width;
    if ((unsigned long long)plane_stride_bytes < need) return fail(ctx, RPT_ERR_ARG, "rpt_scatter_colour_plane3: plane stride smaller than one rank's plane");
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    const dim3 grid((width + 255) / 256, height);
    hipLaunchKernelGGL(rptd::rpt_scatter_plane3_kernel, grid, dim3(256), 0, hip_stream ? (hipStream_t)hip_stream : ctx->stream,
                       (const uint8_t *)planes3, (rpt_pixel *)out16, width, height, n_ranks, plane_stride_bytes);
    RPT_HIP(ctx, hipGetLastError());
    return RPT_OK;
}

int rpt_scatter_helper_planes3_on(rpt_ctx *ctx, void *hip_stream, const void *planes3, void *out16, int width, int height,
                                  int n_ranks, int root_run, size_t plane_stride_bytes) {
    if (!ctx || !planes3 || !out16 || width <= 0 || height <= 0 || n_ranks < 1 || root_run < 1) return RPT_ERR_ARG;
    const int period = root_run + n_ranks - 1;
    const int tiles = (height + RPT_TILE_ROWS - 1) / RPT_TILE_ROWS;
    const unsigned long long need = 3ull * (unsigned long long)((tiles + period - 1) / period) * RPT_TILE_ROWS * (unsigned long long)width;
    if ((unsigned long long)plane_stride_bytes < need) return fail(ctx, RPT_ERR_ARG, "rpt_scatter_helper_planes3: plane stride smaller than one helper's plane");
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    const dim3 grid((width + 255) / 256, height);
    hipLaunchKernelGGL(rptd::rpt_scatter_helper_planes3_kernel, grid, dim3(256), 0, hip_stream ? (hipStream_t)hip_stream : ctx->stream,
                       (const uint8_t *)planes3, (rpt_pixel *)out16, width, height, period, root_run, plane_stride_bytes);
    RPT_HIP(ctx, hipGetLastError());
    return RPT_OK;
}

int rpt_read_counters(rpt_ctx *ctx, unsigned long long out[16]) {
    if (!ctx || !out) return RPT_ERR_ARG;
    if (!ctx->counters.ptr) return fail(ctx, RPT_ERR_STATE, "rpt_read_counters: render with the diagnostic variant (7) first");
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    RPT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RPT_HIP(ctx, hipMemcpy(out, ctx->counters.ptr, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RPT_OK;
}

int rpt_read_wave_times(rpt_ctx *ctx, unsigned long long *out, size_t max_words, size_t *words) {
    if (!ctx || !out || !words) return RPT_ERR_ARG;
    if (!ctx->wave_times.ptr) return fail(ctx, RPT_ERR_STATE, "rpt_read_wave_times: render with the diagnostic variant (7) first");
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    RPT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    size_t n = ctx->wave_times.bytes / sizeof(unsigned long long);
    if (n > max_words) n = max_words;
    RPT_HIP(ctx, hipMemcpy(out, ctx->wave_times.ptr, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    *words = n;
    return RPT_OK;
}

int rpt_probe(rpt_ctx *ctx, int which, const void *host_in, void *host_out, int n) {
    static const int in_w[6] = {15, 12, 4, 3, 3, 6}, out_w[6] = {4, 5, 3, 3, 2, 12};
    if (!ctx || which < 0 || which > 5 || !host_in || !host_out || n <= 0) return RPT_ERR_ARG;
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    float *d_in = nullptr, *d_out = nullptr;
    RPT_HIP(ctx, hipMalloc((void **)&d_in, sizeof(float) * in_w[which] * n));
    if (hipMalloc((void **)&d_out, sizeof(float) * out_w[which] * n) != hipSuccess) {
        (void)hipFree(d_in);
        return fail(ctx, RPT_ERR_NOMEM, "rpt_probe: hipMalloc");
    }
    int rc = RPT_OK;
    if (hipMemcpy(d_in, host_in, sizeof(float) * in_w[which] * n, hipMemcpyHostToDevice) != hipSuccess) rc = RPT_ERR_DEVICE;
    if (!rc) {
        hipLaunchKernelGGL(rptd::rpt_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, which, d_in, d_out, n);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
            hipMemcpy(host_out, d_out, sizeof(float) * out_w[which] * n, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(ctx, RPT_ERR_DEVICE, "rpt_probe: device error");
    }
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return rc;
}

// the arguments the ray-level probes need: the resident scene and the current Object[] / DObj[] (no frame, no outputs)
static void probe_kernel_args(rpt_ctx *ctx, rptd::KernelArgs &a) {
    std::memset(&a, 0, sizeof a);
    a.dnodes = (const rptd::DNode *)ctx->geo->dnodes.ptr;
    a.dtris = (const rptd::DTri *)ctx->geo->dtris.ptr;
    a.links = (const int *)ctx->geo->dlinks.ptr;
    a.first_tris = (const rptd::DTri *)ctx->geo->dfirst.ptr;
    a.root_grids = (const int *)ctx->geo->dgrids.ptr;       // (null and 0 in the product library)
    a.seen_before = (const uint2 *)ctx->geo->dseen.ptr;
    a.grid_roots = ctx->geo->grid_roots;
    a.top_count = ctx->geo->top_count;
    a.dobjs = (const rptd::DObj *)((const char *)ctx->objects.ptr + (size_t)ctx->object_count * sizeof(rpt_object));
    a.objects = (const rpt_object *)ctx->objects.ptr;
    a.vertices = (const rpt_float3 *)ctx->geo->vertices.ptr;
    a.normals = (const rpt_float3 *)ctx->geo->normals.ptr;
    a.uvs = (const rpt_float2 *)ctx->geo->uvs.ptr;
    a.triangles = (const uint32_t *)ctx->geo->triangles.ptr;
    a.octrees = (const rpt_octree *)ctx->geo->octrees.ptr;
    a.octreeTris = (const int32_t *)ctx->geo->octreeTris.ptr;
    a.textures = (const uint8_t *)ctx->geo->textures.ptr;
    a.texture_bytes = (long long)ctx->geo->textures.bytes;
    a.object_count = ctx->object_count;
    a.interval = ctx->interval;
}

int rpt_probe_walk(rpt_ctx *ctx, int object_index, const float *host_rays, float *host_out, int n) {
    if (!ctx || !host_rays || !host_out || n <= 0) return RPT_ERR_ARG;
    if (!ctx->scene_uploaded || ctx->object_count <= 0) return fail(ctx, RPT_ERR_STATE, "rpt_probe_walk before rpt_upload_scene / rpt_set_objects");
    if (object_index < 0 || object_index >= ctx->object_count || ctx->host_objects.size() < (size_t)(object_index + 1) * sizeof(rpt_object) ||
        ((const rpt_object *)ctx->host_objects.data())[object_index].type != RPT_MESH)
        return fail(ctx, RPT_ERR_ARG, "rpt_probe_walk: not a mesh object");
    if (!ctx->geo->compact_ok) return fail(ctx, RPT_ERR_STATE, "rpt_probe_walk: this octree has no derived layout (children not consecutive)");
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    rptd::KernelArgs a;
    probe_kernel_args(ctx, a);
    float *d_in = nullptr, *d_out = nullptr;
    RPT_HIP(ctx, hipMalloc((void **)&d_in, sizeof(float) * 6 * (size_t)n));
    if (hipMalloc((void **)&d_out, sizeof(float) * 24 * (size_t)n) != hipSuccess) {
        (void)hipFree(d_in);
        return fail(ctx, RPT_ERR_NOMEM, "rpt_probe_walk: hipMalloc");
    }
    int rc = RPT_OK;
    if (hipMemcpy(d_in, host_rays, sizeof(float) * 6 * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) rc = RPT_ERR_DEVICE;
    if (!rc) {
        hipLaunchKernelGGL(rptd::rpt_probe_walk_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, a, object_index, d_in, d_out, n);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
            hipMemcpy(host_out, d_out, sizeof(float) * 24 * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(ctx, RPT_ERR_DEVICE, "rpt_probe_walk: device error");
    }
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return rc;
}

int rpt_probe_division(rpt_ctx *ctx, int mode, unsigned int seed, int blocks, int per_thread, unsigned long long counts_out[5], float *samples_out, int max_samples) {
    if (!ctx || !counts_out || mode < 0 || mode > 3 || blocks <= 0 || blocks > 65536 || per_thread <= 0 || max_samples < 0) return RPT_ERR_ARG;
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long *d_counts = nullptr;
    float *d_samples = nullptr;
    RPT_HIP(ctx, hipMalloc((void **)&d_counts, 5 * sizeof(unsigned long long)));
    if (max_samples > 0 && samples_out && hipMalloc((void **)&d_samples, sizeof(float) * 4 * (size_t)max_samples) != hipSuccess) {
        (void)hipFree(d_counts);
        return fail(ctx, RPT_ERR_NOMEM, "rpt_probe_division: hipMalloc");
    }
    int rc = RPT_OK;
    if (hipMemsetAsync(d_counts, 0, 5 * sizeof(unsigned long long), ctx->stream) != hipSuccess) rc = RPT_ERR_DEVICE;
    if (!rc && d_samples && hipMemsetAsync(d_samples, 0, sizeof(float) * 4 * (size_t)max_samples, ctx->stream) != hipSuccess) rc = RPT_ERR_DEVICE;
    if (!rc) {
        hipLaunchKernelGGL(rptd::rpt_probe_division_kernel, dim3((unsigned int)blocks), dim3(256), 0, ctx->stream, mode, (uint32_t)seed, per_thread, d_counts, d_samples, max_samples);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
            hipMemcpy(counts_out, d_counts, 5 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess ||
            (d_samples && hipMemcpy(samples_out, d_samples, sizeof(float) * 4 * (size_t)max_samples, hipMemcpyDeviceToHost) != hipSuccess))
            rc = fail(ctx, RPT_ERR_DEVICE, "rpt_probe_division: device error");
    }
    (void)hipFree(d_counts);
    if (d_samples) (void)hipFree(d_samples);
    return rc;
}

int rpt_probe_object(rpt_ctx *ctx, int which, int object_index, const float *host_in, float *host_out, int n) {
    if (!ctx || !host_in || !host_out || n <= 0 || which < 0 || which > 3) return RPT_ERR_ARG;
    if (!ctx->scene_uploaded || ctx->object_count <= 0) return fail(ctx, RPT_ERR_STATE, "rpt_probe_object before rpt_upload_scene / rpt_set_objects");
    if (object_index < 0 || object_index >= ctx->object_count) return fail(ctx, RPT_ERR_ARG, "rpt_probe_object: no such object");
    if (!ctx->geo->compact_ok) return fail(ctx, RPT_ERR_STATE, "rpt_probe_object: this octree has no derived layout (children not consecutive)");
    RPT_HIP(ctx, hipSetDevice(ctx->device));
    rptd::KernelArgs a;
    probe_kernel_args(ctx, a);
    const size_t in_floats = which == 0 ? 8 : which == 1 ? 9 : which == 2 ? 4 : 3, out_floats = which == 1 ? 2 : which == 2 ? 16 : 8;
    float *d_in = nullptr, *d_out = nullptr;
    RPT_HIP(ctx, hipMalloc((void **)&d_in, sizeof(float) * in_floats * (size_t)n));
    if (hipMalloc((void **)&d_out, sizeof(float) * out_floats * (size_t)n) != hipSuccess) {
        (void)hipFree(d_in);
        return fail(ctx, RPT_ERR_NOMEM, "rpt_probe_object: hipMalloc");
    }
    int rc = RPT_OK;
    if (hipMemcpy(d_in, host_in, sizeof(float) * in_floats * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) rc = RPT_ERR_DEVICE;
    if (!rc) {
        hipLaunchKernelGGL(rptd::rpt_probe_object_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, a, which, object_index, d_in, d_out, n);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
            hipMemcpy(host_out, d_out, sizeof(float) * out_floats * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(ctx, RPT_ERR_DEVICE, "rpt_probe_object: device error");
    }
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return rc;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// GPU octree build: replaces Mesh::GenerateOctree (Mesh.cpp:5-28) + Subdivide
// (Octree.cpp:171-248).  The triangle/box classification of each level runs on the device
// (rpt_octree_build.hip.h); the O(nodes) bookkeeping — child boxes, the valence stop rule, the reference's
// depth-first numbering of nodes and lists, neighbour links — is done here with the host builder's
// arithmetic, so the output is byte-identical to csrc/host/rpt_octree.cpp.
namespace {

struct BuildNode {
    float mn[3], mx[3];
    unsigned int list_begin = 0, list_count = 0;   // in its level's list buffer
    int min_tris = 0, depth = 0;
    int first_child = -1;                           // index in the next level's node array, -1 = not split
    int valence = 0;
};

struct DevTemp {
    void *p = nullptr;
    ~DevTemp() { if (p) (void)hipFree(p); }
};

}  // namespace

extern "C" void rpt_free_host(void *p) { std::free(p); }

extern "C" int rpt_build_octree(rpt_ctx *ctx, const rpt_float3 *vertices, size_t vertex_count, const uint32_t *triangles,
                                size_t triangle_words, size_t first_triangle_word, int node_index_base, int tri_index_base,
                                rpt_octree **nodes_out, size_t *node_count, int32_t **tris_out, size_t *tri_count) {
    if (!ctx || !vertices || !triangles || !nodes_out || !node_count || !tris_out || !tri_count) return RPT_ERR_ARG;
    if (triangle_words % 9 || first_triangle_word % 9 || first_triangle_word >= triangle_words)
        return fail(ctx, RPT_ERR_ARG, "rpt_build_octree: triangle words must be a multiple of 9 and the mesh must have triangles");
    const size_t n_tris_total = triangle_words / 9;
    if (n_tris_total > (size_t)0x3fffffff) return fail(ctx, RPT_ERR_ARG, "rpt_build_octree: too many triangles");
    for (size_t w = 0; w < triangle_words; w += 3)
        if (triangles[w] >= vertex_count) return fail(ctx, RPT_ERR_SCENE, "rpt_build_octree: vertex index out of range");
    RPT_HIP(ctx, hipSetDevice(ctx->device));

    // root (Mesh.cpp:6-21): bounds over this mesh's face-corner vertices, list = every triangle imported so far
    rptb::BNode root;
    std::memset(&root, 0, sizeof root);
    {
        const rpt_float3 v0 = vertices[triangles[first_triangle_word]];
        root.mn[0] = root.mx[0] = v0.x; root.mn[1] = root.mx[1] = v0.y; root.mn[2] = root.mx[2] = v0.z;
        for (size_t i = first_triangle_word / 3; i < triangle_words / 3; i++) {
            const rpt_float3 v = vertices[triangles[3 * i]];
            const float c[3] = {v.x, v.y, v.z};
            for (int k = 0; k < 3; k++) {
                root.mn[k] = root.mn[k] < c[k] ? root.mn[k] : c[k];
                root.mx[k] = root.mx[k] > c[k] ? root.mx[k] : c[k];
            }
        }
        root.list_begin = 0;
        root.list_count = (unsigned int)n_tris_total;
        root.min_tris = 0;
        root.depth = 6;
        root.first_child = -1;
        root.valence = 0;
    }
    const int max_depth = root.depth;                                  // levels 0 .. max_depth
    size_t level_nodes[8], node_offset[8], nodes_total = 0;           // at most 8^level nodes on a level
    for (int l = 0; l <= max_depth; l++) {
        level_nodes[l] = (size_t)1 << (3 * l);
        node_offset[l] = nodes_total;
        nodes_total += level_nodes[l];
    }

    DevTemp d_vertices, d_triangles;
    RPT_HIP(ctx, hipMalloc(&d_vertices.p, vertex_count * sizeof(rpt_float3)));
    RPT_HIP(ctx, hipMalloc(&d_triangles.p, triangle_words * sizeof(uint32_t)));
    RPT_HIP(ctx, hipMemcpy(d_vertices.p, vertices, vertex_count * sizeof(rpt_float3), hipMemcpyHostToDevice));
    RPT_HIP(ctx, hipMemcpy(d_triangles.p, triangles, triangle_words * sizeof(uint32_t), hipMemcpyHostToDevice));

    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (std::getenv("RPT_OCTREE_TIMING")) { auto t = std::chrono::steady_clock::now(); std::fprintf(stderr, "  octree %-10s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - T0).count()); T0 = t; } };
    std::vector<std::vector<BuildNode>> levels;
    std::vector<std::vector<int32_t>> lists;          // host copies of every level's concatenated lists
    // One submission per attempt; an attempt whose lists outgrow `cap` entries per level says so in the header and is repeated.
    size_t cap = std::max<size_t>(16 * n_tris_total, 4096);
    if (const char *e = std::getenv("RPT_OCTREE_LIST_CAP"))            // test hook: start small, so that the repeat path runs
        cap = std::max<size_t>(n_tris_total, (size_t)std::strtoull(e, nullptr, 10));
    for (int attempt = 0;; attempt++) {
        if (8 * cap > 0xffffffffull) return fail(ctx, RPT_ERR_NOMEM, "rpt_build_octree: triangle lists exceed what the builder indexes with 32 bits");
        size_t hash_slots = 1;
        while (hash_slots < 6 * cap) hash_slots <<= 1;                 // three vertex occurrences per entry, at most half full
        const size_t chunk_cap = cap / 32 + level_nodes[max_depth] + 8; // 8 * (entries / 256 + splitting nodes), rounded up
        // one arena for everything the levels need
        size_t arena_bytes = 0;
        auto carve = [&](size_t bytes) { const size_t at = arena_bytes; arena_bytes += (bytes + 255) & ~(size_t)255; return at; };
        const size_t o_hdr = carve(sizeof(rptb::BuildHeader)), o_nodes = carve(nodes_total * sizeof(rptb::BNode)),
                     o_lists = carve((size_t)(max_depth + 1) * cap * sizeof(int32_t)), o_children = carve(level_nodes[max_depth] * sizeof(rptb::ChildDesc)),
                     o_split = carve(level_nodes[max_depth] / 8 * sizeof(unsigned int) + 16), o_flags = carve(8 * cap + 16),
                     o_counts = carve(chunk_cap * sizeof(unsigned int)), o_chunk_out = carve(chunk_cap * sizeof(unsigned int)),
                     o_keys = carve(hash_slots * sizeof(unsigned long long)), o_hits = carve(hash_slots * sizeof(unsigned int));
        DevTemp arena;
        RPT_HIP(ctx, hipMalloc(&arena.p, arena_bytes));
        struct Part { void *p; } d_hdr{(char *)arena.p + o_hdr}, d_nodes{(char *)arena.p + o_nodes}, d_lists{(char *)arena.p + o_lists},
            d_children{(char *)arena.p + o_children}, d_split{(char *)arena.p + o_split}, d_flags{(char *)arena.p + o_flags},
            d_counts{(char *)arena.p + o_counts}, d_chunk_out{(char *)arena.p + o_chunk_out}, d_keys{(char *)arena.p + o_keys},
            d_hits{(char *)arena.p + o_hits};
        lap("alloc");
        rptb::BNode *nodes_dev = (rptb::BNode *)d_nodes.p;
        int32_t *lists_dev = (int32_t *)d_lists.p;
        rptb::BuildHeader *hdr_dev = (rptb::BuildHeader *)d_hdr.p;

        rptb::BuildHeader hdr;
        std::memset(&hdr, 0, sizeof hdr);
        hdr.n_nodes[0] = 1;
        hdr.list_len[0] = (unsigned int)n_tris_total;
        RPT_HIP(ctx, hipMemcpyAsync(hdr_dev, &hdr, sizeof hdr, hipMemcpyHostToDevice, ctx->stream));
        RPT_HIP(ctx, hipMemcpyAsync(nodes_dev, &root, sizeof root, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(rptb::iota_kernel, dim3((unsigned int)((n_tris_total + 255) / 256)), dim3(256), 0, ctx->stream, lists_dev,
                           (unsigned int)n_tris_total);
        const unsigned int entry_blocks = (unsigned int)((cap + 255) / 256);
        for (int l = 0; l < max_depth; l++) {
            rptb::BNode *nodes_l = nodes_dev + node_offset[l], *nodes_next = nodes_dev + node_offset[l + 1];
            int32_t *list_l = lists_dev + (size_t)l * cap, *list_next = lists_dev + (size_t)(l + 1) * cap;
            const size_t chunks_l = std::min(chunk_cap, cap / 32 + level_nodes[l + 1] + 8);
            RPT_HIP(ctx, hipMemsetAsync(d_keys.p, 0xff, hash_slots * sizeof(unsigned long long), ctx->stream));
            RPT_HIP(ctx, hipMemsetAsync(d_hits.p, 0, hash_slots * sizeof(unsigned int), ctx->stream));
            hipLaunchKernelGGL(rptb::valence_kernel, dim3(entry_blocks), dim3(256), 0, ctx->stream, (const uint32_t *)d_triangles.p,
                               (const int32_t *)list_l, nodes_l, (const rptb::BuildHeader *)hdr_dev, l, (unsigned long long *)d_keys.p,
                               (unsigned int *)d_hits.p, (unsigned int)(hash_slots - 1));
            hipLaunchKernelGGL(rptb::split_kernel, dim3(1), dim3(1024), 0, ctx->stream, nodes_l, hdr_dev, l, (rptb::ChildDesc *)d_children.p,
                               (unsigned int *)d_split.p, (unsigned int)level_nodes[l + 1]);
            hipLaunchKernelGGL(rptb::sat_flag_kernel, dim3((unsigned int)chunks_l), dim3(256), 0, ctx->stream, (const rpt_float3 *)d_vertices.p,
                               (const uint32_t *)d_triangles.p, (const int32_t *)list_l, (const rptb::ChildDesc *)d_children.p,
                               (const rptb::BuildHeader *)hdr_dev, (unsigned char *)d_flags.p, (unsigned int *)d_counts.p);
            hipLaunchKernelGGL(rptb::chunk_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const unsigned int *)d_counts.p,
                               (unsigned int *)d_chunk_out.p, hdr_dev, l, (unsigned int)cap);
            hipLaunchKernelGGL(rptb::sat_compact_kernel, dim3((unsigned int)chunks_l), dim3(256), 0, ctx->stream, (const int32_t *)list_l,
                               (const rptb::ChildDesc *)d_children.p, (const rptb::BuildHeader *)hdr_dev, (const unsigned char *)d_flags.p,
                               (const unsigned int *)d_chunk_out.p, list_next, (unsigned int)cap);
            hipLaunchKernelGGL(rptb::next_nodes_kernel, dim3((unsigned int)((level_nodes[l + 1] + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const rptb::BNode *)nodes_l, (const rptb::ChildDesc *)d_children.p, (const unsigned int *)d_split.p,
                               (const unsigned int *)d_chunk_out.p, (const rptb::BuildHeader *)hdr_dev, l, nodes_next);
        }
        RPT_HIP(ctx, hipGetLastError());
        lap("submit");
        RPT_HIP(ctx, hipMemcpyAsync(&hdr, hdr_dev, sizeof hdr, hipMemcpyDeviceToHost, ctx->stream));
        RPT_HIP(ctx, hipStreamSynchronize(ctx->stream));             // the only wait of the build
        lap("wait");
        if (hdr.overflow) {
            // the level that overflowed said how many entries it wanted; deeper levels can want more still (at most 8x per level in
            // theory, ~1.3x in practice), so that figure gets a quarter on top, and the build may be repeated up to six times
            if (attempt >= 6) return fail(ctx, RPT_ERR_NOMEM, "rpt_build_octree: triangle lists keep outgrowing their buffers");
            cap = std::max<size_t>(cap * 2, (size_t)hdr.needed + (size_t)hdr.needed / 4);
            continue;
        }
        // read the levels back: nodes and lists of every level that has nodes
        levels.clear();
        lists.clear();
        for (int l = 0; l <= max_depth && hdr.n_nodes[l] > 0; l++) {
            if (hdr.n_nodes[l] > level_nodes[l] || hdr.list_len[l] > cap) return fail(ctx, RPT_ERR_DEVICE, "rpt_build_octree: inconsistent level sizes");
            std::vector<rptb::BNode> dev_nodes(hdr.n_nodes[l]);
            RPT_HIP(ctx, hipMemcpy(dev_nodes.data(), nodes_dev + node_offset[l], dev_nodes.size() * sizeof(rptb::BNode), hipMemcpyDeviceToHost));
            std::vector<BuildNode> nodes(dev_nodes.size());
            for (size_t i = 0; i < nodes.size(); i++) {
                const rptb::BNode &d = dev_nodes[i];
                BuildNode &b = nodes[i];
                for (int k = 0; k < 3; k++) { b.mn[k] = d.mn[k]; b.mx[k] = d.mx[k]; }
                b.list_begin = d.list_begin; b.list_count = d.list_count;
                b.min_tris = d.min_tris; b.depth = d.depth; b.first_child = d.first_child; b.valence = d.valence;
                if ((size_t)d.list_begin + d.list_count > hdr.list_len[l] || (d.first_child >= 0 && (l == max_depth || (size_t)d.first_child + 8 > hdr.n_nodes[l + 1])))
                    return fail(ctx, RPT_ERR_DEVICE, "rpt_build_octree: inconsistent node record");
            }
            std::vector<int32_t> list(hdr.list_len[l]);
            if (!list.empty())
                RPT_HIP(ctx, hipMemcpy(list.data(), lists_dev + (size_t)l * cap, list.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
            levels.push_back(std::move(nodes));
            lists.push_back(std::move(list));
        }
        lap("readback");
        break;
    }
    lap("free");

    // the reference's numbering: a node's eight children and their lists are appended when the node is split,
    // and the children are then visited depth first (Octree.cpp:191-246)
    std::vector<rpt_octree> out_nodes;
    std::vector<int32_t> out_tris;
    {
        size_t n_nodes_all = 0, n_entries_all = 0;
        for (size_t l = 0; l < levels.size(); l++) { n_nodes_all += levels[l].size(); n_entries_all += lists[l].size(); }
        out_nodes.reserve(n_nodes_all);
        out_tris.reserve(n_entries_all);
    }
    auto make_node = [&](const BuildNode &b, const std::vector<int32_t> &list) {
        rpt_octree o;
        std::memset(&o, 0, sizeof o);
        o.min.x = b.mn[0]; o.min.y = b.mn[1]; o.min.z = b.mn[2];
        o.max.x = b.mx[0]; o.max.y = b.mx[1]; o.max.z = b.mx[2];
        o.trisIndex = tri_index_base + (int)out_tris.size();
        o.trisCount = (int)b.list_count;
        for (int &c : o.children) c = -1;
        for (int &n : o.neighbors) n = -1;
        out_tris.insert(out_tris.end(), list.begin() + b.list_begin, list.begin() + b.list_begin + b.list_count);
        out_nodes.push_back(o);
    };
    make_node(levels[0][0], lists[0]);
    struct Frame { size_t level; int index; int out; };
    // explicit recursion: emit(level, index in level, output index)
    std::function<void(size_t, int, int)> emit = [&](size_t level, int index, int out) {
        const BuildNode &b = levels[level][index];
        if (b.first_child < 0) return;
        int child_out[8];
        for (int k = 0; k < 8; k++) {
            child_out[k] = (int)out_nodes.size();
            out_nodes[out].children[k] = node_index_base + child_out[k];
            make_node(levels[level + 1][b.first_child + k], lists[level + 1]);
        }
        const rpt_octree parent = out_nodes[out];
        for (int k = 0; k < 8; k++) {        // neighbour links (Octree.cpp:213-244): -z,+z,-x,+x,-y,+y
            const int bit[3] = {k & 1, (k >> 2) & 1, (k >> 1) & 1}, step[3] = {1, 4, 2};
            rpt_octree &c = out_nodes[child_out[k]];
            for (int axis = 0; axis < 3; axis++) {
                const int lo = 2 * axis, hi = 2 * axis + 1;
                if (bit[axis] == 0) { c.neighbors[lo] = parent.neighbors[lo]; c.neighbors[hi] = parent.children[k + step[axis]]; }
                else { c.neighbors[lo] = parent.children[k - step[axis]]; c.neighbors[hi] = parent.neighbors[hi]; }
            }
        }
        for (int k = 0; k < 8; k++) emit(level + 1, b.first_child + k, child_out[k]);
    };
    emit(0, 0, 0);

    lap("emit");
    rpt_octree *nodes_host = (rpt_octree *)std::malloc(out_nodes.size() * sizeof(rpt_octree));
    int32_t *tris_host = (int32_t *)std::malloc((out_tris.size() ? out_tris.size() : 1) * sizeof(int32_t));
    if (!nodes_host || !tris_host) {
        std::free(nodes_host);
        std::free(tris_host);
        return fail(ctx, RPT_ERR_NOMEM, "rpt_build_octree: out of host memory");
    }
    std::memcpy(nodes_host, out_nodes.data(), out_nodes.size() * sizeof(rpt_octree));
    if (!out_tris.empty()) std::memcpy(tris_host, out_tris.data(), out_tris.size() * sizeof(int32_t));
    *nodes_out = nodes_host;
    *node_count = out_nodes.size();
    *tris_out = tris_host;
    *tri_count = out_tris.size();
    return RPT_OK;
}
