// rpt_kernels.hip.h — the per-pixel render path for gfx950 (CDNA4, wave64).
//
// What is computed is the reference's render_kernel (opencl_kernel.cl:620-660) and everything it
// calls; how it is computed is organised for MI355X:
//   * one wavefront owns an 8x8 pixel tile (coherent rays, 8 full 128-B lines per store wave) and is a workgroup
//     of its own (its slot is free again when IT ends, not when the longest of four neighbours does); the grid is
//     (4 ceil(W/32), row tiles): 129 600 workgroups at 4K.  The measurement arms keep four waves per workgroup;
//   * Object[] is indexed with a wave-uniform loop counter, so the matrices arrive through the
//     scalar cache into SGPRs (s_load) and are broadcast for free; per-frame constants that do
//     not depend on the pixel (aspect ratio, hable(white_point)) are computed once on the host
//     with the same IEEE operations;
//   * 16 B/pixel framebuffer stores are one global_store_dwordx4 per lane; the multi-GPU variant
//     writes only the 4-B packed colour into a compact plane;
//   * what a wavefront can skip as a whole it skips with a __ballot: its object mask for the primary rays comes from
//     per-object screen bounds (host, rpt_screen_bounds.hpp) tested lane-parallel (wave_object_mask), and a shadow ray's
//     sphere and cube tests are dropped when no lane's segment to the light can reach the object's box (intersect_object).
// fp32 arithmetic order is that of the reference expression by expression (see the oracle for the
// built-in semantics); UB of the reference is neutralised exactly as in oracle/rpt_oracle.c.
#pragma once
#include "../../include/rpt_layout.h"
#include "rpt_device_math.hip.h"

#ifdef RPT_RELAXED_FP
#pragma clang fp contract(fast)      /* rpt_relaxed.hip only: the opt-in "OpenCL-conformant arithmetic" build of the same source */
#else
#pragma clang fp contract(off)
#endif

namespace rptd {

#define RPT_EPSILON 0.0000001f       /* opencl_kernel.cl:6 */
#define RPT_MAX_LEAF_STEPS 4096
#define RPT_LINK_CHILD_MASK 0x00ffffff   /* DNode::link: low 24 bits = first child, high 8 = which children are leaves */
#define RPT_TOP_MAX 3072                 /* node links of the octrees' top levels kept in LDS by the persistent kernels (12 KB) */
#define RPT_PI_D 3.14159265358979323846264338327950288   /* OpenCL C M_PI (double) */

// ---- derived, device-only layouts (built by the library at upload / per frame; values are the
// reference's own numbers or IEEE results of the reference's own operations, so nothing rounds
// differently) ----------------------------------------------------------------------------------
// One octree node in one 64-B line.  Children of a node are consecutive in the reference's builder
// (Octree.cpp:191-269 pushes the eight children back to back), so children[k] = firstChild + k.
// The derived array is numbered breadth first over the whole forest (all roots, then all nodes of level 1, ...): the top levels
// of every octree are the first KernelArgs::top_count records, which is what the persistent kernels keep in LDS.
struct alignas(64) DNode {
    float minx, miny, minz; int link;            // -1 = leaf; else firstChild | (leaf mask of the eight children) << 24
    float maxx, maxy, maxz; int leafBegin;       // first DTri of a leaf
    int leafCount; int nb[6]; int pad;           // neighbours -z,+z,-x,+x,-y,+y
};
static_assert(sizeof(DNode) == 64, "DNode is one 64-B line");
// One leaf triangle reference, gathered: A, B-A, C-A (the ray-independent part of
// intersect_triangle, opencl_kernel.cl:108-109) and the triangle id, instead of the
// octreeTris -> triangles -> vertices chain of three dependent loads.
struct alignas(16) DTri { float ax, ay, az, e1x; float e1y, e1z, e2x, e2y; float e2z; int tri; int pad0, pad1; };
static_assert(sizeof(DTri) == 48, "DTri is three 16-B loads");
// Per object, per frame: the primary-ray origin in object space (every primary ray of a frame
// starts at the camera event, opencl_kernel.cl:386-389) and what follows from it alone.
struct alignas(16) DObj {
    float ox, oy, oz; float sphere_c;        // exact (kernel operation order): used by the intersectors
    float winding;
    // conservative culling data (approximate arithmetic is fine here, see rpt_tile_bin_kernel):
    float cbx, cby, cbz;                     // bounding-sphere centre in object space
    float rb;                                // bounding-sphere radius, inflated; < 0 = never cull this object
    float B[9];                              // object-space direction = B * nd + b for a camera direction nd
    float b[3];
    float mesh_in_box;                       // mesh objects: 1 = every triangle the octree can report lies inside the root's box
    int root;                                // mesh objects: the root's index in the derived (breadth-first) node numbering
    float pad;
    // the shadow-ray culls of a mesh object (mesh_ray_misses_root, mesh_segment_apart below; build_dobjs derives them per frame):
    // half extents of the root box about (cbx, cby, cbz) grown by 8u max(|lo|, |hi|) (mh[0] < 0: no cull of this object at all);
    // the segment cull's margin = mconst + mslope * (L1 distance of the ray origin from the centre) (mslope < 0: no segment cull);
    // the allowance of the segment's end per unit of the rest-frame origin's L1 norm.
    float mh[3];
    float mconst, mslope, mcw;
    float ms0;                               // the constant part of that allowance (1e-4 + what the object's own translation costs)
    float pad2;
};
static_assert(sizeof(DObj) == 128, "DObj");

struct KernelArgs {
    // ---- the first 64 bytes are everything a wave that hits nothing needs (its cull and its store): one scalar load at the top
    // of the kernel instead of one per place of first use (each is a dependent round trip in a wave that lives a microsecond)
    // per-object image-plane rectangles (rpt_screen_bounds.hpp), tested lane-parallel by each wavefront (V >= 20)
    const float4 *rects;                     // [2 * object_count] per object: u0, v0, u1, v1 on the plane z = 0.5, then the
                                             // diagonal slabs p_lo, p_hi (u + v) and m_lo, m_hi (u - v)
    rpt_pixel *out16;        // 16 B/pixel framebuffer (full frame addressing) or null
    uint32_t *plane;         // compact 4 B/pixel colour plane (local tile addressing) or null
    float *debug_rgb;        // 3 floats/pixel, full frame addressing, or null
    int object_count;
    int width, height;
    int diagonals;                           // some object has diagonal slabs and the frame lies inside their window
    float inv_width, inv_height;             // 1/width, 1/height (for the cull only: approximate is fine there)
    float aspect;            // (float)width / (float)height
    uint32_t bg_packed;      // the packed R,G,B,1 word of a miss pixel
    // ---- second line: tile addressing, dispatch order, the rest of the scalars
    int first_tile, tile_step;      // local tile t holds global tile (t >> run_log2) * tile_step + first_tile + (t & (run - 1))
    int run_log2;                   // run = 1 << run_log2 consecutive tiles per period of tile_step tiles (rpt_set_tile_pattern)
    int interval;
    // dispatch order (V == 23): the strips [first_sx, first_sx + first_w) x [first_ty, first_ty + first_h) — where the meshes
    // are, i.e. where the frame's longest waves live — are handed out FIRST, the rest in natural order; first_w = 0: off
    int first_sx, first_ty, first_w, first_h;
    int msaa;                       // MSAASAMPLES of opencl_kernel.cl:7 when it is not 1 (rpt_set_msaa; the kernels of render_pixel_body_msaa only)
    float bg_mapped[3];      // min(hable(background)/hable(white_point), 1): what every miss pixel maps to
    float ambient;
    float hable_wp[3];       // hable(white_point), host-computed
    // per-tile object masks: 8x8-pixel tiles of this context's rows, classified once per frame by rpt_tile_bin_kernel
    int mask_tiles_x, n_tiles;               // tiles per row, tiles in this context's rows
    unsigned long long *tile_masks;          // [n_tiles] bit i = primary rays of the tile may hit object i (i < 64)
    // ---- the scene
    const DNode *dnodes;
    const DTri *dtris;
    const DObj *dobjs;
    const int *links;               // DNode::link of every node again, 4 B apart: what a descent reads per level
    const DTri *first_tris;         // per node: the first triangle record of its leaf list again, addressable by the NODE's index
    const uint2 *seen_before;       // (diagnostics library: arms 705 / 717) per (node, entry face): {the leaf across that face, bit mask of this leaf's first 32 list entries that are also in THAT leaf's list}; product: null
    const int *root_grids;          // (diagnostics library: arms 593 / 605) per octree root, 16^3 cells -> node | level << 24 | leaf << 28; product: null
    int grid_roots;
    int top_count;                  // nodes [0, top_count) are the forest's top levels (whole levels, <= RPT_TOP_MAX)
    // persistent kernels (rpt_persistent.hip.h): the band of tile rows that holds the meshes (first_ty, first_h above) is
    // claimed tile by tile from per-queue counters, the other rows are dealt statically in runs of RPT_SKY_RUN tiles
    int tiles_x;                    // 8x8 tiles per row of tiles
    unsigned int tiles_x_magic;     // ceil(2^32 / tiles_x): t / tiles_x = mulhi(t, magic) for every t the host allows
    int runs_x;                     // runs per row of tiles outside the band
    unsigned int runs_x_magic;
    int band_tiles, sky_runs;
    int claim_set;                  // which of the two counter sets this launch counts in
    const rpt_object *objects;
    const rpt_float3 *vertices;
    const rpt_float3 *normals;
    const rpt_float2 *uvs;
    const uint32_t *triangles;
    const rpt_octree *octrees;
    const int32_t *octreeTris;
    const uint8_t *textures;
    long long texture_bytes;
    unsigned long long *wave_times; // diagnostic build only (variant 11): ten words per wave, {start, end} of s_memrealtime (100 MHz) + loop accounting
    unsigned long long *counters;   // diagnostic builds only (variant 7): [0..2] lane-level leaf/tri/descent
                                    // iterations, [3..5] the same counted once per executing wave
};

struct Hit {                 // opencl_kernel.cl:38-44
    float dist;
    f3 normal;
    f2 uv;
    int object;
};

struct Ray { f3 origin, dir; };

// ---------------------------------------------------------------------------------------------
// opencl_kernel.cl:55-73
RPT_DEV f3 createCamRayDir(float x_coord, float y_coord, int width, int height, float aspect_ratio) {
    const float fx = x_coord / (float)width;
    const float fy = y_coord / (float)height;
    const float fx2 = (fx - 0.5f) * aspect_ratio;
    const float fy2 = fy - 0.5f;
    return normalize(mk3(fx2, fy2, 0.5f));
}

// opencl_kernel.cl:106-126
RPT_DEV bool intersect_triangle(f3 A, f3 B, f3 C, const Ray &ray, float &dist, f2 &uv) {
    const f3 v0v1 = B - A;
    const f3 v0v2 = C - A;
    const f3 pvec = cross(ray.dir, v0v2);
    const float det = dot(v0v1, pvec);
    if (det < RPT_EPSILON && -RPT_EPSILON < det) return false;
    const float invDet = 1 / det;
    const f3 tvec = ray.origin - A;
    uv.x = dot(tvec, pvec) * invDet;
    if (uv.x < 0 || uv.x > 1) return false;
    const f3 qvec = cross(tvec, v0v1);
    uv.y = dot(ray.dir, qvec) * invDet;
    if (uv.y < 0 || uv.x + uv.y > 1) return false;
    dist = dot(v0v2, qvec) * invDet;
    return true;
}

// opencl_kernel.cl:128-170.  bounds[sign] is written as a select so nothing is indexed dynamically.
RPT_DEV bool intersect_AABB(f3 bmin, f3 bmax, const Ray &ray, f2 &d, int &closeSide, int &farSide) {
    const f3 origin = ray.origin;
    const f3 inv_dir = mk3(1.0f / ray.dir.x, 1.0f / ray.dir.y, 1.0f / ray.dir.z);
    const int sx = inv_dir.x < 0 ? 1 : 0, sy = inv_dir.y < 0 ? 1 : 0, sz = inv_dir.z < 0 ? 1 : 0;
    d.x = ((sx ? bmax.x : bmin.x) - origin.x) * inv_dir.x;
    d.y = ((sx ? bmin.x : bmax.x) - origin.x) * inv_dir.x;
    closeSide = 2 + sx;
    farSide = 3 - sx;
    const float tymin = ((sy ? bmax.y : bmin.y) - origin.y) * inv_dir.y;
    const float tymax = ((sy ? bmin.y : bmax.y) - origin.y) * inv_dir.y;
    if ((d.x > tymax) || (tymin > d.y)) return false;
    if (tymin > d.x) { d.x = tymin; closeSide = 4 + sy; }
    if (tymax < d.y) { d.y = tymax; farSide = 5 - sy; }
    const float tzmin = ((sz ? bmax.z : bmin.z) - origin.z) * inv_dir.z;
    const float tzmax = ((sz ? bmin.z : bmax.z) - origin.z) * inv_dir.z;
    if ((d.x > tzmax) || (tzmin > d.y)) return false;
    if (tzmin > d.x) { d.x = tzmin; closeSide = sz; }
    if (tzmax < d.y) { d.y = tzmax; farSide = 1 - sz; }
    return d.y > 0;
}

// opencl_kernel.cl:172-198 with the direction reciprocals and signs hoisted out of the leaf walk
// (scaledDir is constant along one traversal, so 1/scaledDir is computed once: same values).
struct ExitPlan { f3 scaledDir, inv_dir; int sx, sy, sz; };

RPT_DEV ExitPlan makeExitPlan(f3 scaledDir) {
    ExitPlan p;
    p.scaledDir = scaledDir;
    p.inv_dir = mk3(1.0f / scaledDir.x, 1.0f / scaledDir.y, 1.0f / scaledDir.z);
    p.sx = p.inv_dir.x < 0;
    p.sy = p.inv_dir.y < 0;
    p.sz = p.inv_dir.z < 0;
    return p;
}

RPT_DEV int getOppositeBoxSide(const ExitPlan &p, f3 &uv) {
    const float dx = ((float)(1 - p.sx) - uv.x) * p.inv_dir.x;
    const float dy = ((float)(1 - p.sy) - uv.y) * p.inv_dir.y;
    const float dz = ((float)(1 - p.sz) - uv.z) * p.inv_dir.z;
    float t;
    int side;
    if (dx < dy) {
        if (dx < dz) { t = dx; side = 3 - p.sx; } else { t = dz; side = 1 - p.sz; }
    } else {
        if (dy < dz) { t = dy; side = 5 - p.sy; } else { t = dz; side = 1 - p.sz; }
    }
    uv = uv + p.scaledDir * t;
    return side;
}

// EXPERIMENT, not in the product build (RPT_PACKED_EXIT_FACES=1 turns it on in the throughput walk): the same plan with the three
// candidate exit faces in ONE register, picked by a shift that depends on the step.  Written as above, the compiler hoists 3 - sx,
// 5 - sy, 1 - sz out of the leaf loop as three registers; in kernel 41 (96 registers for five waves) two of them are spilled and
// reloaded inside the loop.  The packed form removes those reloads (scratch 12 -> 8 B, the rest outside the loops) for one
// instruction more per step.  Measured twice, A/B/A/B/A/B against the library of record: -1.5 / -1.7 / -0.7 % in flight in one
// build, +0.2 / -0.7 / -1.6 % in the next: inside the run-to-run spread of the bench line, so the record stays as it is
// (profiles/r04_exitplan_ab.txt).
#ifndef RPT_PACKED_EXIT_FACES
#define RPT_PACKED_EXIT_FACES 0
#endif
struct PackedExitPlan { f3 scaledDir, inv_dir, far; int sides; };

RPT_DEV PackedExitPlan makePackedExitPlan(f3 scaledDir) {
    PackedExitPlan p;
    p.scaledDir = scaledDir;
    p.inv_dir = mk3(1.0f / scaledDir.x, 1.0f / scaledDir.y, 1.0f / scaledDir.z);
    const int sx = p.inv_dir.x < 0, sy = p.inv_dir.y < 0, sz = p.inv_dir.z < 0;
    p.far = mk3((float)(1 - sx), (float)(1 - sy), (float)(1 - sz));
    p.sides = (3 - sx) | ((5 - sy) << 8) | ((1 - sz) << 16);
    return p;
}

RPT_DEV int getOppositeBoxSide(const PackedExitPlan &p, f3 &uv) {
    const float dx = (p.far.x - uv.x) * p.inv_dir.x;
    const float dy = (p.far.y - uv.y) * p.inv_dir.y;
    const float dz = (p.far.z - uv.z) * p.inv_dir.z;
    float t;
    int shift;
    if (dx < dy) {
        if (dx < dz) { t = dx; shift = 0; } else { t = dz; shift = 16; }
    } else {
        if (dy < dz) { t = dy; shift = 8; } else { t = dz; shift = 16; }
    }
    uv = uv + p.scaledDir * t;
    return (p.sides >> shift) & 0xff;
}

template <bool PACKED> struct ExitPlanOf { typedef ExitPlan type; static RPT_DEV ExitPlan make(f3 d) { return makeExitPlan(d); } };
template <> struct ExitPlanOf<true> { typedef PackedExitPlan type; static RPT_DEV PackedExitPlan make(f3 d) { return makePackedExitPlan(d); } };

// The same step for the common case 0 <= uv < 1.5 on every axis (+0 included, -0/NaN/negative excluded by
// the unsigned compare on the bit patterns): there round(c) is (c >= 0.5), min(c, 1-eps) keeps that bit, and
// 2*fmod(m, 0.5) is 2*(m - 0.5*bit) — every operation exact, so the results are those of the general form.
RPT_DEV int octree_child_step(f3 &uv);
RPT_DEV int octree_child_step_fast(f3 &uv) {
    const unsigned int lim = 0x3FC00000u;   // 1.5f
    const bool in_range = (__float_as_uint(uv.x) < lim) && (__float_as_uint(uv.y) < lim) && (__float_as_uint(uv.z) < lim);
    if (!in_range) return octree_child_step(uv);
    const float top = 1.0f - RPT_EPSILON;
    const bool bx = uv.x >= 0.5f, by = uv.y >= 0.5f, bz = uv.z >= 0.5f;
    const float mx = top < uv.x ? top : uv.x, my = top < uv.y ? top : uv.y, mz = top < uv.z ? top : uv.z;
    uv.x = 2.0f * (mx - (bx ? 0.5f : 0.0f));
    uv.y = 2.0f * (my - (by ? 0.5f : 0.0f));
    uv.z = 2.0f * (mz - (bz ? 0.5f : 0.0f));
    return (bz ? 1 : 0) + (by ? 2 : 0) + (bx ? 4 : 0);
}

// child selection and re-normalisation of opencl_kernel.cl:237-238 / 257-258
RPT_DEV int octree_child_step(f3 &uv) {
    const float fidx = __builtin_roundf(uv.z) + 2 * __builtin_roundf(uv.y) + 4 * __builtin_roundf(uv.x);
    const int childIndex = !(fidx >= 0.0f) ? 0 : (fidx > 7.0f ? 7 : (int)fidx);
    uv.x = 2.0f * fmod_half(cl_min(uv.x, 1.0f - RPT_EPSILON));
    uv.y = 2.0f * fmod_half(cl_min(uv.y, 1.0f - RPT_EPSILON));
    uv.z = 2.0f * fmod_half(cl_min(uv.z, 1.0f - RPT_EPSILON));
    return childIndex;
}

// ---- octree storage policies -------------------------------------------------------------------
// Node<0>: the reference's 96-B nodes read field by field (a traversal step needs min/max,
//          (trisIndex,trisCount), children[0], one children[k] and one neighbors[k], not the whole
//          struct the reference copies).  Works for any valid octree.
// (The derived 64-B DNode + gathered DTri records are read by octree_walk below, record by record.)
typedef float v4f __attribute__((ext_vector_type(4)));   // native vectors: one 16-B load, SROA-friendly
typedef int v4i __attribute__((ext_vector_type(4)));
template <int V> struct NodeRef;

template <> struct NodeRef<0> {
    int idx;
    RPT_DEV void load(const KernelArgs &a, int i) { idx = i; }
    RPT_DEV f3 bmin(const KernelArgs &a) const { return ld3(a.octrees[idx].min); }
    RPT_DEV f3 bmax(const KernelArgs &a) const { return ld3(a.octrees[idx].max); }
    RPT_DEV bool is_leaf(const KernelArgs &a) const { return a.octrees[idx].children[0] == -1; }
    RPT_DEV int child(const KernelArgs &a, int k) const { return a.octrees[idx].children[k]; }
    RPT_DEV int neighbor(const KernelArgs &a, int side) const { return a.octrees[idx].neighbors[side]; }
    RPT_DEV int tri_begin(const KernelArgs &a) const { return a.octrees[idx].trisIndex; }
    RPT_DEV int tri_count(const KernelArgs &a) const { return a.octrees[idx].trisCount; }
    // triangle k of the leaf: A, B-A, C-A and its id
    RPT_DEV void tri(const KernelArgs &a, int k, f3 &A, f3 &v0v1, f3 &v0v2, int &id) const {
        id = a.octreeTris[k];
        A = ld3(a.vertices[a.triangles[9 * id + 3 * 0]]);
        const f3 B = ld3(a.vertices[a.triangles[9 * id + 3 * 1]]);
        const f3 C = ld3(a.vertices[a.triangles[9 * id + 3 * 2]]);
        v0v1 = B - A;
        v0v2 = C - A;
    }
};
// opencl_kernel.cl:106-126 with the two edge vectors supplied
RPT_DEV bool intersect_triangle_edges(f3 A, f3 v0v1, f3 v0v2, const Ray &ray, float &dist, f2 &uv) {
    const f3 pvec = cross(ray.dir, v0v2);
    const float det = dot(v0v1, pvec);
    if (det < RPT_EPSILON && -RPT_EPSILON < det) return false;
    const float invDet = 1 / det;
    const f3 tvec = ray.origin - A;
    uv.x = dot(tvec, pvec) * invDet;
    if (uv.x < 0 || uv.x > 1) return false;
    const f3 qvec = cross(tvec, v0v1);
    uv.y = dot(ray.dir, qvec) * invDet;
    if (uv.y < 0 || uv.x + uv.y > 1) return false;
    dist = dot(v0v2, qvec) * invDet;
    return true;
}
// The walk's stop test (opencl_kernel.cl:283): length(exit point - origin) > hit.dist.  Until a triangle has been hit,
// hit.dist is the caller's 1e20f, and sqrt(s) > 1e20f holds for no finite float s (sqrt(FLT_MAX) < 1.9e19), for s = +inf
// only, and not for NaN: so while no lane of the wave has a hit the square root is not needed to decide it — same answer.
RPT_DEV bool exit_is_past_hit(f3 v, float hit_dist, bool didHit) {
    const float s = dot(v, v);
    if (__ballot(didHit) == 0ull && hit_dist == 1e20f) return s == __builtin_inff();
    return __builtin_sqrtf(s) > hit_dist;
}
// opencl_kernel.cl:287-306: normal, texture coordinates and the distance re-measured in the caller's frame, from the walk's
// closest triangle (hit.dist parametric, hit.uv barycentric on entry)
RPT_DEV void mesh_hit_finish(const KernelArgs &a, const rpt_object &obj, f3 origin, f3 dir, int hitTri, f3 world_origin,
                             float world_dirlen, Hit &hit) {
    const float u = hit.uv.x, v = hit.uv.y;
    const float w = 1.0f - u - v;
    const f3 normA = ld3(a.normals[a.triangles[2 + 9 * hitTri + 3 * 0]]);
    const f3 normB = ld3(a.normals[a.triangles[2 + 9 * hitTri + 3 * 1]]);
    const f3 normC = ld3(a.normals[a.triangles[2 + 9 * hitTri + 3 * 2]]);
    hit.normal = normalize(applyTranspose(obj.InvM, normA * w + normB * u + normC * v));
    const rpt_float2 uvA = a.uvs[a.triangles[1 + 9 * hitTri + 3 * 0]];
    const rpt_float2 uvB = a.uvs[a.triangles[1 + 9 * hitTri + 3 * 1]];
    const rpt_float2 uvC = a.uvs[a.triangles[1 + 9 * hitTri + 3 * 2]];
    hit.uv.x = w * uvA.x + u * uvB.x + v * uvC.x;
    hit.uv.y = w * uvA.y + u * uvB.y + v * uvC.y;
    const f3 objPoint = origin + dir * hit.dist;
    const f3 worldPoint = transformPoint(obj.M, objPoint);
    hit.dist = length(worldPoint - world_origin) / world_dirlen;
}

// opencl_kernel.cl:200-308 on the reference's own layouts (any valid octree; the fallback kernel, variant 1): from the point
// where the ray is in object space.  newRay = object-space ray (direction normalised); world_origin/world_dirlen are
// ray->origin.yzw and |ray->dir.yzw|.
RPT_DEV bool octree_core_ref(const KernelArgs &a, const rpt_object &obj, const Ray &newRay, f3 world_origin, float world_dirlen, Hit &hit) {
    NodeRef<0> node;
    int currOctreeIndex = obj.meshIndex;
    node.load(a, currOctreeIndex);
    f2 d;
    int closeSide, farSide;
    f3 nmin = node.bmin(a), nmax = node.bmax(a);
    if (!intersect_AABB(nmin, nmax, newRay, d, closeSide, farSide)) return false;
    f3 uv = newRay.origin + newRay.dir * d.x;
    if (d.x < 0) {   // ray starts inside the root: descend to the leaf holding the origin
        uv = (newRay.origin - nmin) / (nmax - nmin);
        while (!node.is_leaf(a)) {
            currOctreeIndex = node.child(a, octree_child_step(uv));
            node.load(a, currOctreeIndex);
        }
        nmin = node.bmin(a);
        nmax = node.bmax(a);
        if (!intersect_AABB(nmin, nmax, newRay, d, closeSide, farSide)) return false;
        uv = newRay.origin + newRay.dir * d.x;
    }
    const ExitPlan plan = makeExitPlan(normalize(newRay.dir / (nmax - nmin)));
    bool didHit = false;
    int hitTri = 0;
    int steps = 0;
    while (currOctreeIndex != -1) {
        if (++steps > RPT_MAX_LEAF_STEPS) break;
        node.load(a, currOctreeIndex);
        nmin = node.bmin(a);
        nmax = node.bmax(a);
        uv = (uv - nmin) / (nmax - nmin);
        while (!node.is_leaf(a)) {
            currOctreeIndex = node.child(a, octree_child_step(uv));
            node.load(a, currOctreeIndex);
            nmin = node.bmin(a);
            nmax = node.bmax(a);
        }
        const int trisIndex = node.tri_begin(a);
        const int trisEnd = trisIndex + node.tri_count(a);
        for (int i = trisIndex; i < trisEnd; i++) {
            f3 A, v0v1, v0v2;
            int tri;
            node.tri(a, i, A, v0v1, v0v2, tri);
            float dist;
            f2 triUV;
            if (intersect_triangle_edges(A, v0v1, v0v2, newRay, dist, triUV)) {
                if (0 <= dist && dist < hit.dist) {
                    hitTri = tri;
                    hit.dist = dist;
                    hit.uv = triUV;
                    didHit = true;
                }
            }
        }
        const f3 extents = nmax - nmin;
        farSide = getOppositeBoxSide(plan, uv);
        uv = nmin + uv * extents;
        currOctreeIndex = node.neighbor(a, farSide);
        if (exit_is_past_hit(uv - newRay.origin, hit.dist, didHit)) break;
    }
    if (!didHit) return false;
    mesh_hit_finish(a, obj, newRay.origin, newRay.dir, hitTri, world_origin, world_dirlen, hit);
    return true;
}

// ---- the walk on the derived layouts (what ships) -------------------------------------------------------------------------
// opencl_kernel.cl:200-308 once more, arithmetic unchanged; what is organised for the machine is WHEN memory is asked for — a
// leaf step of the reference's loop is a chain of dependent round trips (node, one per descent level, one per triangle, the
// neighbour), and with five waves on a SIMD the waves wait on exactly that chain (profiles/r02_*_pmc_summary.json: half of their
// life), so the chain is made shorter, not the traffic smaller:
//   * a node is read as ONE 64-B record (box, link, first triangle, count), a triangle as one 40-B record (A, B-A, C-A, id);
//   * a descent reads 4 bytes per level from the compact link array, and the parent's link says which children are leaves, so no
//     level is spent on finding that out;
//   * the exit face of a leaf does not depend on its triangles (getOppositeBoxSide works on the ray and the entry point alone):
//     it is found BEFORE the triangle loop, and the neighbour's index travels while the triangles are tested;
//   * PIPELINE + FIRST (kernel 43: the blocking call, whose frame is as long as its longest wave, and small frames in flight):
//     triangle records are asked for one iteration ahead, and a leaf's first record together with its node record
//     (load_first_tri).  Ten more live registers: 44 B of scratch at five waves per SIMD, worth it where frames wait for
//     latency (bunny 4K one frame at a time 0.196 -> 0.186 ms, 1080p 0.166 -> 0.151), not where the chip is full of walks
//     (4K in flight 0.088 -> 0.095 ms per frame): profiles/r03_walk_ab.txt, r03_latency_walk_ab.txt.
// Measured against round 2's walk (profiles/r03_walk_ab.txt): bunny 4K 0.0949 -> 0.0903 ms per frame in flight, 0.201 -> 0.191
// one at a time; zero scratch instead of 12 B.  What was tried on top and lost is in the diagnostics build (rpt_diag_walks.hip.h).
// hi.w of a node record = leafBegin | min(leafCount, 255) << 24 (build_derived_geometry): the count of a leaf's list travels with the
// box, so a node visit of the throughput walk is two 16-B loads, not three instructions; a list of 255 or more reads the full count
// from its own field, and so does the latency walk always (PACKED_COUNT = false: see mesh_walk).
#define RPT_NODE_BEGIN_MASK 0x00ffffff
struct NodeRec { v4f lo, hi; int count; };
template <bool PACKED_COUNT = true>
RPT_DEV NodeRec load_node_rec(const KernelArgs &a, int i) {
    const v4f *p = reinterpret_cast<const v4f *>(a.dnodes + i);
    NodeRec r;
    r.lo = p[0];
    r.hi = p[1];
    if (PACKED_COUNT) {
        r.count = (int)(__float_as_uint(r.hi.w) >> 24);
        if (r.count == 255) r.count = a.dnodes[i].leafCount;
    } else {
        r.count = a.dnodes[i].leafCount;
    }
    return r;
}
struct TriRec { v4f t0, t1; float e2z; int tri; };
// LATE_ID: the triangle's id is not read with every record tested (9 dwords instead of 10 through the L1's return path, which is what
// frames in flight wait for: profiles/r03_td_bound.txt) — the walk remembers the RECORD it hit and reads that one id at the end.
template <bool LATE_ID = false>
RPT_DEV TriRec load_tri_rec(const KernelArgs &a, int k) {
    const v4f *p = reinterpret_cast<const v4f *>(a.dtris + k);
    TriRec r;
    r.t0 = p[0];
    r.t1 = p[1];
    if (LATE_ID) {
        r.e2z = *reinterpret_cast<const float *>(p + 2);
        r.tri = k;
    } else {
        const float2 t2 = *reinterpret_cast<const float2 *>(p + 2);
        r.e2z = t2.x;
        r.tri = __float_as_int(t2.y);
    }
    return r;
}
// The first record of a node's list, by NODE index (48 B per node, zeros where the list is empty): its address is known as soon as
// the node's is, so the latency walk asks for it together with the node record — one exposed round trip less per non-empty leaf,
// 48 B more asked of the L1 per node visited.
template <bool LATE_ID = false>
RPT_DEV TriRec load_first_tri(const KernelArgs &a, int node) {
    const v4f *p = reinterpret_cast<const v4f *>(a.first_tris + node);
    TriRec r;
    r.t0 = p[0];
    r.t1 = p[1];
    if (LATE_ID) {
        r.e2z = *reinterpret_cast<const float *>(p + 2);
        r.tri = 0;          // (the walk puts the record's index in when it tests the record)
    } else {
        const float2 t2 = *reinterpret_cast<const float2 *>(p + 2);
        r.e2z = t2.x;
        r.tri = __float_as_int(t2.y);
    }
    return r;
}
RPT_DEV void test_tri_rec(const TriRec &r, const Ray &ray, Hit &hit, int &hitTri, bool &didHit) {
    float dist;
    f2 triUV;
    if (intersect_triangle_edges(mk3(r.t0.x, r.t0.y, r.t0.z), mk3(r.t0.w, r.t1.x, r.t1.y), mk3(r.t1.z, r.t1.w, r.e2z), ray, dist, triUV)) {
        if (0 <= dist && dist < hit.dist) {
            hitTri = r.tri;
            hit.dist = dist;
            hit.uv = triUV;
            didHit = true;
        }
    }
}

// from an inner node (link != -1) down to the leaf that holds uv (opencl_kernel.cl:256-261: the same child steps)
RPT_DEV int descend_to_leaf(const KernelArgs &a, int link, f3 &uv) {
    int idx;
    for (;;) {
        const int k = octree_child_step_fast(uv);
        idx = (link & RPT_LINK_CHILD_MASK) + k;
        if ((link >> (24 + k)) & 1) break;          // the link says this child is a leaf: no lookup
        link = a.links[idx];
    }
    return idx;
}

// PIPELINE: triangle records one iteration ahead.  FIRST (with PIPELINE): the first record of a leaf comes with its node record.
// MEASUREMENT ARM (ROOT_GRID; diagnostics library, arms 593 / 605; exact, and not adopted: profiles/r03_root_grid_ab.txt — the link words of
// an octree's top levels are a handful of hot addresses, a 16-KB table read per lane is not).
// The descent from a ROOT (opencl_kernel.cl:256-261 at the start of a walk) in one lookup.  For 0 <= c < 1.5 the child step of a
// component is a bit extraction (octree_child_step_fast): with m = min(c, 1 - eps), the child bit at level k is bit k of m's binary
// fraction and the re-normalised coordinate after L levels is frac(2^L m) — every product and difference exact.  A component in
// (-2^-10, 0) — an entry point a rounding below its face — selects child 0 at every level (round(c) = -0) and is doubled per level
// (2 fmod(c, 0.5) = 2 c, exact): truncation gives both.  So the cell (trunc(16 m.x), trunc(16 m.y), trunc(16 m.z)) of a 16^3 table
// names the node four levels down, or the leaf above that level with the level it lives on, and uv leaves as L single child steps
// would leave it; deeper trees continue from there.  Any other component (NaN, >= 1.5, more negative) takes the serial descent.
#define RPT_GRID_LEVELS 4
#define RPT_GRID_CELLS 4096
#ifndef RPT_DIAGNOSTICS
RPT_DEV int descend_from_root(const KernelArgs &a, int, int link, f3 &uv) { return descend_to_leaf(a, link, uv); }     // (the product library has no tables)
#else
RPT_DEV int descend_from_root(const KernelArgs &a, int root, int link, f3 &uv) {
    const float lo = -0x1p-10f;
    const bool ok = (uv.x > lo) & (uv.x < 1.5f) & (uv.y > lo) & (uv.y < 1.5f) & (uv.z > lo) & (uv.z < 1.5f) & (root < a.grid_roots);
    if (!ok) return descend_to_leaf(a, link, uv);
    const float top = 1.0f - RPT_EPSILON;
    const float mx = top < uv.x ? top : uv.x, my = top < uv.y ? top : uv.y, mz = top < uv.z ? top : uv.z;
    const int cx = (int)(mx * 16.0f), cy = (int)(my * 16.0f), cz = (int)(mz * 16.0f);
    const int e = a.root_grids[root * RPT_GRID_CELLS + ((cx * 16 + cy) * 16 + cz)];
    const float s = (float)(1 << ((e >> 24) & 7));
    const float x = mx * s, y = my * s, z = mz * s;
    uv.x = x - (float)(int)x;
    uv.y = y - (float)(int)y;
    uv.z = z - (float)(int)z;
    int node = e & RPT_LINK_CHILD_MASK;
    if (!((e >> 28) & 1)) node = descend_to_leaf(a, a.links[node], uv);       // an inner node four levels down: the tree goes deeper here
    return node;
}

#endif

// MEASUREMENT ARMS (UNIFORM = 1 / 2; diagnostics library, arms 657 / 669 / 673; bit-identical, 3-16 % slower: profiles/r03_td_bound.txt).
#ifndef RPT_DIAGNOSTICS
template <bool PACKED_COUNT, int UNIFORM>
RPT_DEV NodeRec load_node_rec_u(const KernelArgs &a, int curr, bool &uni) { uni = false; return load_node_rec<PACKED_COUNT>(a, curr); }
template <bool LATE_ID>
RPT_DEV TriRec load_tri_rec_leader(const KernelArgs &a, int k) { return load_tri_rec<LATE_ID>(a, k); }
#else
// UNIFORM = 1: where every active lane of the wave stands in the SAME node (bunny 4K: 42 % of the wave's leaf steps, shadows
// 74 %: profiles/r02_divergence.txt) the node record and the leaf's triangle records are read ONCE for the wave — the address is made
// wave-uniform with readfirstlane, so the loads go through the scalar cache into SGPRs instead of 64 times through the vector L1's
// return path — and the arithmetic takes them as scalar operands.  Same operations on the same values.
// UNIFORM = 1: through the scalar cache (above).  UNIFORM = 2: ONE lane of the wave makes the vector loads (an L1 hit as before, one
// lane's worth of work for the return path instead of the wave's) and readfirstlane hands the dwords to everybody as scalars.
RPT_DEV float bcast_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
RPT_DEV v4f bcast_v4(v4f v) { v4f r; r.x = bcast_f(v.x); r.y = bcast_f(v.y); r.z = bcast_f(v.z); r.w = bcast_f(v.w); return r; }
template <bool PACKED_COUNT, int UNIFORM>
RPT_DEV NodeRec load_node_rec_u(const KernelArgs &a, int curr, bool &uni) {
    uni = false;
    if (UNIFORM) {
        const int u = __builtin_amdgcn_readfirstlane(curr);
        uni = __ballot(curr != u) == 0ull;
        if (uni && UNIFORM == 1) return load_node_rec<PACKED_COUNT>(a, u);
        if (uni) {
            const bool leader = (int)(threadIdx.x & 63) == __ffsll((long long)__ballot(true)) - 1;
            NodeRec r;
            r.lo = r.hi = v4f{0.0f, 0.0f, 0.0f, 0.0f};
            r.count = 0;
            if (leader) r = load_node_rec<PACKED_COUNT>(a, curr);
            r.lo = bcast_v4(r.lo);
            r.hi = bcast_v4(r.hi);
            r.count = __builtin_amdgcn_readfirstlane(r.count);
            return r;
        }
    }
    return load_node_rec<PACKED_COUNT>(a, curr);
}
template <bool LATE_ID>
RPT_DEV TriRec load_tri_rec_leader(const KernelArgs &a, int k) {
    const bool leader = (int)(threadIdx.x & 63) == __ffsll((long long)__ballot(true)) - 1;
    TriRec r;
    r.t0 = r.t1 = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    r.e2z = 0.0f;
    r.tri = 0;
    if (leader) r = load_tri_rec<LATE_ID>(a, k);
    r.t0 = bcast_v4(r.t0);
    r.t1 = bcast_v4(r.t1);
    r.e2z = bcast_f(r.e2z);
    r.tri = __builtin_amdgcn_readfirstlane(r.tri);
    return r;
}

#endif

// DEDUP (measurement arms 705 / 717): a triangle that overlaps k leaves is in all k lists, and one walk tests it up
// to k times (23.8 % of the bunny's triangle tests, 28.4 % of the pear's: profiles/r04_repeated_triangle_tests.txt).  A repeat can
// never change the walk's state — the test's outcome depends on the ray and the triangle only, and the update rule
// 0 <= dist < hit.dist (opencl_kernel.cl:270) with a non-increasing hit.dist makes a second application a no-op (accepted before:
// now dist == hit.dist or larger, not "<"; rejected before: rejected again; NaN: false both times).  Skipped are list entries whose
// triangle is ALSO IN THE LIST OF THE LEAF THIS WALK VISITED IMMEDIATELY BEFORE: the host marks, per leaf and face, which of the
// leaf's first 32 entries are in the list of the leaf across that face (seen_before[node * 6 + face] = {that leaf, mask}); the lane
// uses the mask of the face it entered through (the previous step's exit face, flipped) only if the leaf recorded there IS the
// leaf it came from — an integer compare, nothing geometric is assumed — and every entry of that leaf's list was either tested
// in the previous step or skipped there for the same reason (induction over the steps).  The record's address is known with
// the node's: its load travels with the node record and does not lengthen the chain.
template <bool PIPELINE, bool FIRST, bool PACKED_COUNT = true, bool ROOT_GRID = false, bool LATE_ID = false, int UNIFORM = 0, bool DEDUP = false>
RPT_DEV bool octree_walk(const KernelArgs &a, const rpt_object &obj, int root, const Ray &newRay, f3 world_origin,
                         float world_dirlen, Hit &hit) {
    int curr = root;
    int prev_leaf = -1, entry_face = 0;
    bool uni = false;
    NodeRec rec = load_node_rec<PACKED_COUNT>(a, curr);
    f2 d;
    int closeSide, farSide;
    f3 nmin = mk3(rec.lo.x, rec.lo.y, rec.lo.z), nmax = mk3(rec.hi.x, rec.hi.y, rec.hi.z);
    if (!intersect_AABB(nmin, nmax, newRay, d, closeSide, farSide)) return false;
    f3 uv = newRay.origin + newRay.dir * d.x;
    if (d.x < 0) {   // ray starts inside the root: descend to the leaf holding the origin
        uv = (newRay.origin - nmin) / (nmax - nmin);
        if (__float_as_int(rec.lo.w) != -1) {
            curr = ROOT_GRID ? descend_from_root(a, root, __float_as_int(rec.lo.w), uv) : descend_to_leaf(a, __float_as_int(rec.lo.w), uv);
            rec = load_node_rec_u<PACKED_COUNT, UNIFORM>(a, curr, uni);
        }
        nmin = mk3(rec.lo.x, rec.lo.y, rec.lo.z);
        nmax = mk3(rec.hi.x, rec.hi.y, rec.hi.z);
        if (!intersect_AABB(nmin, nmax, newRay, d, closeSide, farSide)) return false;
        uv = newRay.origin + newRay.dir * d.x;
    }
    const typename ExitPlanOf<PACKED_COUNT && RPT_PACKED_EXIT_FACES>::type plan = ExitPlanOf<PACKED_COUNT && RPT_PACKED_EXIT_FACES>::make(normalize(newRay.dir / (nmax - nmin)));
    bool didHit = false;
    int hitTri = 0;
    TriRec first;
    if (FIRST) first = load_first_tri<LATE_ID>(a, curr);
    for (int steps = 1; steps <= RPT_MAX_LEAF_STEPS; steps++) {
        nmin = mk3(rec.lo.x, rec.lo.y, rec.lo.z);
        nmax = mk3(rec.hi.x, rec.hi.y, rec.hi.z);
        uv = (uv - nmin) / (nmax - nmin);
        if (__float_as_int(rec.lo.w) != -1) {
            // (only a walk's first step can stand on a root: nobody's neighbour link points at one)
            curr = (ROOT_GRID && steps == 1) ? descend_from_root(a, root, __float_as_int(rec.lo.w), uv) : descend_to_leaf(a, __float_as_int(rec.lo.w), uv);
            rec = load_node_rec_u<PACKED_COUNT, UNIFORM>(a, curr, uni);
            if (FIRST) first = load_first_tri<LATE_ID>(a, curr);
            nmin = mk3(rec.lo.x, rec.lo.y, rec.lo.z);
            nmax = mk3(rec.hi.x, rec.hi.y, rec.hi.z);
        }
        int i = __float_as_int(rec.hi.w) & RPT_NODE_BEGIN_MASK;
        const int trisEnd = i + rec.count;
        unsigned int seen = 0u;
        if (DEDUP && prev_leaf >= 0) {
            const uint2 e = a.seen_before[(size_t)curr * 6 + entry_face];
            seen = (int)e.x == prev_leaf ? e.y : 0u;
        }
        const int listBegin = i;
        farSide = getOppositeBoxSide(plan, uv);             // the way out, before the triangles
        const int next = a.dnodes[curr].nb[farSide];
        if (UNIFORM && uni) {          // one list for the whole wave: records through the scalar cache
            const int ue = __builtin_amdgcn_readfirstlane(trisEnd);
            for (int k = __builtin_amdgcn_readfirstlane(i); k < ue; k++)
                test_tri_rec(UNIFORM == 2 ? load_tri_rec_leader<LATE_ID>(a, k) : load_tri_rec<LATE_ID>(a, k), newRay, hit, hitTri, didHit);
        } else if (PIPELINE) {
            if (i < trisEnd) {
                TriRec cur = FIRST ? first : load_tri_rec<LATE_ID>(a, i);
                for (; i < trisEnd; i++) {
                    TriRec nxt = cur;
                    if (i + 1 < trisEnd) nxt = load_tri_rec<LATE_ID>(a, i + 1);
                    if (LATE_ID) cur.tri = i;
                    if (!(DEDUP && i - listBegin < 32 && ((seen >> (i - listBegin)) & 1u))) test_tri_rec(cur, newRay, hit, hitTri, didHit);     // (the record was asked for an iteration ago: only the arithmetic is saved here)
                    cur = nxt;
                }
            }
        } else if (DEDUP) {
            for (; i < trisEnd; i++) {
                if (i - listBegin < 32 && ((seen >> (i - listBegin)) & 1u)) continue;                    // tested in the previous leaf: neither loaded nor tested
                test_tri_rec(load_tri_rec<LATE_ID>(a, i), newRay, hit, hitTri, didHit);
            }
        } else {
            for (; i < trisEnd; i++) test_tri_rec(load_tri_rec<LATE_ID>(a, i), newRay, hit, hitTri, didHit);
        }
        uv = nmin + uv * (nmax - nmin);
        if (exit_is_past_hit(uv - newRay.origin, hit.dist, didHit) || next == -1) break;
        if (DEDUP) { prev_leaf = curr; entry_face = farSide ^ 1; }      // sides 0/1 = -z/+z, 2/3 = -x/+x, 4/5 = -y/+y: the face entered is the face left, flipped
        curr = next;
        rec = load_node_rec_u<PACKED_COUNT, UNIFORM>(a, curr, uni);
        if (FIRST) first = load_first_tri<LATE_ID>(a, curr);
    }
    if (!didHit) return false;
    if (LATE_ID) hitTri = a.dtris[hitTri].tri;
    mesh_hit_finish(a, obj, newRay.origin, newRay.dir, hitTri, world_origin, world_dirlen, hit);
    return true;
}

#ifdef RPT_DIAGNOSTICS
}  // namespace rptd
#include "rpt_diag_walks.hip.h"      /* librpt_hip_diag.so only: round 2's walk with its instrumentation, the experiment arms */
namespace rptd {
#endif

// Which walk a kernel variant uses.  V = 0: the reference's layouts; V = 23 (kernel 43: the blocking call, and frames in flight
// too small to fill the chip with walks): the latency form — records an iteration ahead, a leaf's first record with its node.
template <int V>
RPT_DEV bool mesh_walk(const KernelArgs &a, const rpt_object &obj, int i, const Ray &newRay, f3 world_origin, float world_dirlen, Hit &hit) {
#ifdef RPT_DIAGNOSTICS
    if (diag_walk_selected<V>()) return diag_walk<V>(a, obj, a.dobjs[i].root, newRay, world_origin, world_dirlen, hit);
#endif
    if (V == 0) return octree_core_ref(a, obj, newRay, world_origin, world_dirlen, hit);
    // (the packed leaf count pays in the throughput walk — one instruction less per node visit, -0.5...-1 % — and costs the latency
    // walk 2-4.5 %, whose count then sits behind a shift and a compare instead of arriving beside the box: profiles/r03_packed_count_ab.txt)
    return octree_walk<V == 23, V == 23, V != 23, false, true>(a, obj, a.dobjs[i].root, newRay, world_origin, world_dirlen, hit);
}

RPT_DEV float max3(f3 v) { return cl_max(cl_max(v.x, v.y), v.z); }   // opencl_kernel.cl:310
RPT_DEV float cube_winding(f3 origin) {
    return max3(mk3(__builtin_fabsf(origin.x), __builtin_fabsf(origin.y), __builtin_fabsf(origin.z))) < 1.0f ? -1.0f : 1.0f;
}

// opencl_kernel.cl:312-333 from the object-space ray (dir normalised, scale = its former length)
RPT_DEV bool cube_core(const rpt_object &obj, f3 origin, float winding, f3 dir, float scale, Hit &hit) {
    f3 sgn = mk3(-cl_sign(dir.x), -cl_sign(dir.y), -cl_sign(dir.z));
    const f3 d = (sgn * winding - origin) / dir;
#define RPT_TEST(U, V, W) ((d.U >= 0.0f) && (__builtin_fabsf(origin.V + dir.V * d.U) < 1.0f) && (__builtin_fabsf(origin.W + dir.W * d.U) < 1.0f))
    if (RPT_TEST(x, y, z)) sgn = mk3(sgn.x, 0, 0);
    else if (RPT_TEST(y, z, x)) sgn = mk3(0, sgn.y, 0);
    else sgn = mk3(0, 0, RPT_TEST(z, x, y) ? sgn.z : 0);
#undef RPT_TEST
    const bool any = (sgn.x != 0) || (sgn.y != 0) || (sgn.z != 0);
    if (!any) return false;          // the reference fills hit with NaNs here and discards it
    const float dist = (sgn.x != 0) ? d.x : ((sgn.y != 0) ? d.y : d.z);
    const f3 objPt = origin + dir * dist;
    hit.dist = dist / scale;
    hit.normal = normalize(applyTranspose(obj.InvM, sgn));
    if (sgn.x != 0) { hit.uv.x = (objPt.y + 1) / 2; hit.uv.y = (objPt.z + 1) / 2; }
    else if (sgn.y != 0) { hit.uv.x = (objPt.x + 1) / 2; hit.uv.y = (objPt.z + 1) / 2; }
    else { hit.uv.x = (objPt.x + 1) / 2; hit.uv.y = (objPt.y + 1) / 2; }
    return true;
}

// opencl_kernel.cl:335-359 from the object-space ray; c = dot(rayToSphere,rayToSphere) - 1.
// The (u,v) of a sphere hit is only ever consumed by the texture fetch, so it is evaluated only
// for textured spheres (want_uv); the double-precision divide by M_PI is the reference's (M_PI is a
// double constant in OpenCL C).
RPT_DEV bool sphere_core(const rpt_object &obj, f3 rayToSphere, float c, f3 dir, float scale, Hit &hit, bool want_uv) {
    const float b = dot(rayToSphere, dir);
    float disc = b * b - c;
    if (disc < 0.0f) return false;
    disc = __builtin_sqrtf(disc);
    float dist;
    if ((b - disc) > RPT_EPSILON) dist = b - disc;
    else if ((b + disc) > RPT_EPSILON) dist = b + disc;
    else return false;
    const f3 objPt = -rayToSphere + dir * dist;
    hit.dist = dist / scale;
    hit.normal = normalize(applyTranspose(obj.InvM, objPt));
    if (want_uv) {
        hit.uv.x = (float)(0.5f + rpt_atan2f(objPt.z, objPt.x) / (2 * RPT_PI_D));
        hit.uv.y = (float)(rpt_asinf(objPt.y) / RPT_PI_D + 0.5f);
    } else {
        hit.uv.x = 0.0f;
        hit.uv.y = 0.0f;
    }
    return true;
}

// ---- The shadow-segment culls.  sample_light (opencl_kernel.cl:488-545) asks of every object but the light: is it hit at a
// distance below lightDist?  For a wave whose lanes ALL satisfy the predicates below the answer is "no" without the normalisation
// (a square root, three IEEE divisions) and the intersector.  The predicates are sufficient conditions in the kernel's own float
// arithmetic; u = 2^-24, first-order bounds with the constants rounded up (numerical cross-check: tests/test_float_error_bounds.py).
//
// Sphere and cube (unit shapes in object space).  Notation: o = origin, D = dir (both floats, as computed above), scale =
// fl(|D|), g = fl(D / scale) the direction the intersector works with, `dist` its float result, hit.dist = fl(dist / scale).
//  (i) WHERE the reported hit lies.  Let P = o + g dist, exactly.  cube_core: dist = fl(fl(+-1 - o_U) / g_U) >= 0 and
//      |fl(o_V + fl(g_V dist))| < 1 give |P_U -+ 1| <= 2.1u (1 + |o_U|), |P_V| < 1 + 2.1u + 1.1u |o_V|: P in [-m1, m1]^3 with
//      m1 = 1 + 4u (1 + |o|max).  sphere_core: with b = fl(o' . g), c = fl(fl(|o|^2) - 1), disc = fl(fl(b b) - c), sq = fl(sqrt disc),
//      dist = fl(b -+ sq) one finds |P|^2 = 1 + E,  E = 2 b beta + (c - c~) + u-terms of b^2, disc, sq^2 + 2 sq (beta + rho) +
//      theta dist^2  with |beta| <= 3.1u |o| (the dot product), |c - c~| <= 4.1u |o|^2 + u, |rho| <= u |dist|, |theta| <= 8.1u
//      (|g|^2 - 1), sq <= max(|o|, 1), |dist| <= 2 max(|o|, 1):  E <= 58u (1 + |o|^2), every |P_k| <= |P| <= 1 + 29u (1 + |o|^2).
//  (ii) THAT it lies on the tested segment.  g_k = D_k / scale (1 + d1), hit.dist = dist / scale (1 + d2), |d1|, |d2| <= u:
//      P_k = o_k + D_k tau_k with tau_k = hit.dist (1 + 2.1u) in [0, lightDist (1 + 2.1u)) — inside [0, s], s = fl(lightDist * 1.001
//      + 1e-4).  So P_k lies between o_k and E_k = o_k + D_k s, and the float end point e_k = fl(o_k + fl(D_k s)) obeys
//      |e_k - E_k| <= u (|o_k| + 2.1 |e_k|).
//  (iii) Hence, if o_k > m and e_k > m (or both < -m) for one k, every point of the segment has |x_k| > m (1 - 2.1u) - 1.1u |o_k|,
//      which excludes P as soon as   m >= 1 + 29u (1 + |o|^2) + 2.1u m + 1.1u |o|max   (the cube's 4u (1 + |o|max) is smaller):
//      m = 1 + 2e-6 (1 + |o|^2)  [2e-6 = 33.6u]  does it with room for its own three float roundings.  A NaN compares false: kept.
RPT_DEV bool unit_segment_apart(f3 origin, f3 dir, float seg_max) {
    const float s = seg_max * 1.001f + 1.0e-4f, m = 1.0f + 2.0e-6f * (1.0f + dot(origin, origin));
    const f3 e = origin + dir * s;
    return ((origin.x > m) & (e.x > m)) | ((origin.x < -m) & (e.x < -m)) |
           ((origin.y > m) & (e.y > m)) | ((origin.y < -m) & (e.y < -m)) |
           ((origin.z > m) & (e.z > m)) | ((origin.z < -m) & (e.z < -m));
}
// A mesh (only one whose octree lists nothing but triangles inside the root's box: the host says which — a second mesh's lists also
// carry the first one's triangles, Mesh.cpp:16-19).  The walk accepts a triangle where Moeller-Trumbore in float says so
// (opencl_kernel.cl:106-126: |det| >= 1e-7, 0 <= u <= 1, v >= 0, u + v <= 1, 0 <= dist < best) and re-measures the winner's distance in
// the rest frame: hit.dist = |M (o + g dist) - wo| / |dw| (:301-303; wo, dw = the rest-frame event and direction).
//  (i) WHERE.  With T = A + u e1 + v e2 (a point of the triangle, up to u max(|e1|, |e2|)) and Q = o + g dist, Cramer's rule carried
//      through the float operations gives   |Q - T| <= 27.2u tau |e1| |e2| / |det| + u (tau + 2.1 dist + 2.1 max(|e1|, |e2|))
//      (tau = |o - A|; every numerator and the determinant are 3-term dots of one cross product: 6.8u of their operands' norms each),
//      and |det| >= 1e-7 makes the first term <= 16.2 tau K, K = the largest |e1| |e2| of the mesh's triangles (host, at upload).
//      tau <= t1 + h1, t1 = the L1 distance of o from the box centre, h1 = the box's half extents summed: Q lies within
//      (16.2 K + 3.2u) (t1 + h1) + 4u L  of the root box (L = longest edge).
//  (ii) ON THE SEGMENT.  As above Q_k = o_k + D_k sigma_k, sigma_k = (dist / scale)(1 + u).  dist / scale is NOT hit.dist here:
//      with R3 = M3 InvM3 - I, rt = M3 InvM.t + M.t and kM = || |M3| |InvM3| ||_F the re-measured distance obeys
//      dist / scale <= (hit.dist (1 + 7u) + (c1 |wo| + c0) / |dw|) / (1 - c1),  c1 = ||R3||_F + 16u kM,  c0 = |rt| + 16u (|| |M3| |InvM.t| || + |M.t|).
//      The host checks c1 <= 4e-4 and hands over mcw = 1.01 c1 / dmin and ms0 = 1e-4 + 1.01 c0 / dmin (dmin: a lower bound of |dw|
//      over all unit light directions): then sigma_k <= s = fl(lightDist * 1.001 + (ms0 + mcw |wo|_1)).
//  (iii) Both end points beyond the same plane of the box grown by  (16.2 K + 3.2u)(t1 + h1) + 4u L  [= mconst + mslope t1; mh
//      already holds 8u max(|lo_k|, |hi_k|)]  + 2e-6 (|o_k| + |e_k| + |c_k|)  [the float end point, the subtraction of the centre and
//      the compares] exclude Q.  The argument needs 16.2 K small to be of any use: the host enables this cull only for meshes
//      with 16.2 K + 3.2u <= 0.25 (bunny.obj: 0.0046; pear.obj, whose triangles are up to half a unit long in a model five units
//      tall: 3.1 — for such a mesh a ray within 1e-3 rad of a triangle's plane can be given ANY distance by the float test, the
//      reference's included, and no margin short of the mesh's own size excludes that; it keeps mesh_ray_misses_root only).
RPT_DEV bool mesh_segment_apart(const DObj &pre, f3 wo, f3 origin, f3 dir, float seg_max) {
    const float s = seg_max * 1.001f + (pre.ms0 + pre.mcw * (__builtin_fabsf(wo.x) + __builtin_fabsf(wo.y) + __builtin_fabsf(wo.z)));
    const f3 e = origin + dir * s;
    const float px = origin.x - pre.cbx, py = origin.y - pre.cby, pz = origin.z - pre.cbz;     // (relative to the centre: the box is |x - c| <= mh)
    const float ex = e.x - pre.cbx, ey = e.y - pre.cby, ez = e.z - pre.cbz;
    const float gd = pre.mconst + pre.mslope * (__builtin_fabsf(px) + __builtin_fabsf(py) + __builtin_fabsf(pz));
    const float gx = pre.mh[0] + gd + 2.0e-6f * (__builtin_fabsf(origin.x) + __builtin_fabsf(e.x) + __builtin_fabsf(pre.cbx));
    const float gy = pre.mh[1] + gd + 2.0e-6f * (__builtin_fabsf(origin.y) + __builtin_fabsf(e.y) + __builtin_fabsf(pre.cby));
    const float gz = pre.mh[2] + gd + 2.0e-6f * (__builtin_fabsf(origin.z) + __builtin_fabsf(e.z) + __builtin_fabsf(pre.cbz));
    return ((px > gx) & (ex > gx)) | ((px < -gx) & (ex < -gx)) |
           ((py > gy) & (ey > gy)) | ((py < -gy) & (ey < -gy)) |
           ((pz > gz) & (ez > gz)) | ((pz < -gz) & (ez < -gz));
}
// ANY mesh, any ray: the walk reports nothing unless the float slab test of the root box passes (opencl_kernel.cl:128-170, 228-230),
// and a passing test means that the exact forward ray o + t g, t > 0, meets the box grown per axis by 3.1u |b - o_k| (the six plane
// distances are t = fl(fl(b - o_k) fl(1 / g_k)) = t_exact (1 + 3.1u); the slab logic leaves a float T > 0 between all near and far
// distances; the exact point o + g T lies that close to every slab — rpt_bounds_certify.hpp, section 2).  g_k = D_k / scale (1 + u):
// as a direction, D with every component perturbed by one rounding.  So the walk can be skipped where NO ray o + t D', t > 0,
// D'_k = D_k (1 +- u), meets the box |x_k - c_k| <= H_k, H_k = half extent + 4u (max(|lo_k|, |hi_k|) + |o_k|) — decided here without
// the normalisation and the six divisions by the separating axes of a ray and a box, in float with the roundings accounted for:
//   h_k = mh_k + 6e-7 |o_k| >= H_k (1 + 4u)     (mh_k holds 8u max(|lo_k|, |hi_k|); 6e-7 = 10u)
//   p_k = fl(c_k - o_k) = (c_k - o_k)(1 + u)
//   box axes:   p_k < -h_k and D_k >= 0, or p_k > h_k and D_k <= 0   (the origin beyond a side, moving away or along it)
//   cross axes: the LINE misses if |p_j D'_k - p_k D'_j| > H_j |D'_k| + H_k |D'_j|.  A = fl(fl(p_j D_k) - fl(p_k D_j)) is within
//               3.2u W + u |A| of the left side, W = |p_j D_k| + |p_k D_j|; S = fl(fl(h_j |D_k|) + fl(h_k |D_j|)) (1 + u) bounds
//               the right side: asked is  |A| > S * 1.000001 + 5e-7 W   (5e-7 = 8.4u > 3.2u + u + what S and W themselves round by).
// A NaN or an infinity compares false: the walk runs.
RPT_DEV bool mesh_ray_misses_root(const DObj &pre, f3 origin, f3 dir) {
    const float px = pre.cbx - origin.x, py = pre.cby - origin.y, pz = pre.cbz - origin.z;
    const float hx = pre.mh[0] + 6.0e-7f * __builtin_fabsf(origin.x), hy = pre.mh[1] + 6.0e-7f * __builtin_fabsf(origin.y), hz = pre.mh[2] + 6.0e-7f * __builtin_fabsf(origin.z);
    bool miss = ((px < -hx) & (dir.x >= 0.0f)) | ((px > hx) & (dir.x <= 0.0f)) |
                ((py < -hy) & (dir.y >= 0.0f)) | ((py > hy) & (dir.y <= 0.0f)) |
                ((pz < -hz) & (dir.z >= 0.0f)) | ((pz > hz) & (dir.z <= 0.0f));
    const float ax = __builtin_fabsf(dir.x), ay = __builtin_fabsf(dir.y), az = __builtin_fabsf(dir.z);
    {   // axis x cross D: components (y, z)
        const float m1 = py * dir.z, m2 = pz * dir.y;
        miss = miss | (__builtin_fabsf(m1 - m2) > (hy * az + hz * ay) * 1.000001f + 5.0e-7f * (__builtin_fabsf(m1) + __builtin_fabsf(m2)));
    }
    {   // axis y: (z, x)
        const float m1 = pz * dir.x, m2 = px * dir.z;
        miss = miss | (__builtin_fabsf(m1 - m2) > (hz * ax + hx * az) * 1.000001f + 5.0e-7f * (__builtin_fabsf(m1) + __builtin_fabsf(m2)));
    }
    {   // axis z: (x, y)
        const float m1 = px * dir.y, m2 = py * dir.x;
        miss = miss | (__builtin_fabsf(m1 - m2) > (hx * ay + hy * ax) * 1.000001f + 5.0e-7f * (__builtin_fabsf(m1) + __builtin_fabsf(m2)));
    }
    return miss;
}

// One object against one ray given as a 4-D event + 4-D direction in the object's rest frame
// (the general form: shadow rays, and primary rays of the V = 0 kernel).
// seg_max > 0 (shadow rays): the caller only asks whether the object is hit at a distance below seg_max (sample_light:
// dist < lightDist); if no lane of the wave can get "yes" (unit_segment_apart / mesh_segment_apart above, __ballot), the
// normalisation, its three IEEE divisions and the intersector are skipped for the whole wave.
// (Measured also: the slab test with v_rcp_f32 as a second stage, and the same for mesh roots as a ray test: no
// further gain on any scene — three quarter-rate reciprocals cost what they save)
template <int V>
RPT_DEV bool intersect_object(const KernelArgs &a, int i, f4 origin4, f4 dir4, Hit &hit, float seg_max = -1.0f) {
    const rpt_object &obj = a.objects[i];
    const f3 origin = transformPoint(obj.InvM, yzw(origin4));
    f3 dir = transformDirection(obj.InvM, yzw(dir4));
    if (V >= 20 && seg_max > 0.0f && obj.type != RPT_MESH) {
        if (__ballot(!unit_segment_apart(origin, dir, seg_max)) == 0ull) return false;
    }
    if (V >= 20 && seg_max > 0.0f && obj.type == RPT_MESH && a.dobjs[i].mh[0] >= 0.0f) {
        const DObj &pre = a.dobjs[i];
        bool idle = mesh_ray_misses_root(pre, origin, dir);
        if (pre.mslope >= 0.0f) idle = idle | mesh_segment_apart(pre, yzw(origin4), origin, dir, seg_max);      // (wave-uniform branch)
        if (__ballot(!idle) == 0ull) return false;
    }
    const float scale = length(dir);
    dir = dir / scale;
    switch (obj.type) {
    case RPT_SPHERE: {
        const f3 rayToSphere = -origin;
        return sphere_core(obj, rayToSphere, dot(rayToSphere, rayToSphere) - 1.0f, dir, scale, hit, obj.textureIndex != -1);
    }
    case RPT_CUBE:
        return cube_core(obj, origin, cube_winding(origin), dir, scale, hit);
    case RPT_MESH: {
        if (V == 24) return false;      // the analytic-only kernel is launched for scenes without mesh objects only
        Ray newRay;
        newRay.origin = origin;
        newRay.dir = dir;
        return mesh_walk<V>(a, obj, i, newRay, yzw(origin4), length(yzw(dir4)), hit);
    }
    default:
        return false;
    }
}

// Primary rays of the V >= 1 kernels: the object-space origin and what depends on it alone come
// from the per-frame DObj record; only rows 1..3 of Lorentz * (interval, d) are formed (row 0, the
// time component, is needed for the flash test of the final hit only).
template <int V>
RPT_DEV bool intersect_object_primary(const KernelArgs &a, int i, f4 rayDir, Hit &hit) {
    const rpt_object &obj = a.objects[i];
    const DObj &pre = a.dobjs[i];
    const f3 d3 = mk3(dot(ld4(obj.Lorentz[1]), rayDir), dot(ld4(obj.Lorentz[2]), rayDir), dot(ld4(obj.Lorentz[3]), rayDir));
    f3 dir = transformDirection(obj.InvM, d3);
    const float scale = length(dir);
    dir = dir / scale;
    const f3 origin = mk3(pre.ox, pre.oy, pre.oz);
    switch (obj.type) {
    case RPT_SPHERE:
        return sphere_core(obj, -origin, pre.sphere_c, dir, scale, hit, obj.textureIndex != -1);
    case RPT_CUBE:
        return cube_core(obj, origin, pre.winding, dir, scale, hit);
    case RPT_MESH: {
        if (V == 24) return false;
        Ray newRay;
        newRay.origin = origin;
        newRay.dir = dir;
        const f3 cam3 = mk3(obj.stationaryCam.y, obj.stationaryCam.z, obj.stationaryCam.w);
        return mesh_walk<V>(a, obj, i, newRay, cam3, length(d3), hit);
    }
    default:
        return false;
    }
}

RPT_DEV float texel(const KernelArgs &a, long long addr) {
    addr = addr < 0 ? 0 : addr;
    addr = addr >= a.texture_bytes ? a.texture_bytes - 1 : addr;
    return a.textures[addr] / 255.0f;
}
RPT_DEV f3 texel3(const KernelArgs &a, int offset, int width, int x, int y) {
    const long long base = (long long)offset + 3 * ((long long)width * y + x);
    return mk3(texel(a, base + 0), texel(a, base + 1), texel(a, base + 2));
}

// bilinear RGB8 fetch of opencl_kernel.cl:427-471 (upper clamps only; the odd 4th tap is the reference's)
RPT_DEV f3 sample_texture(const KernelArgs &a, const rpt_object &ho, f2 huv) {
    const int width = ho.textureWidth;
    const int height = ho.textureHeight;
    const float u = width * huv.x;
    const float v = height * (1.0f - huv.y);
    int x = imin(f2i_sat(__builtin_floorf(u)), width - 1);
    int y = imin(f2i_sat(__builtin_floorf(v)), height - 1);
    const float u_ratio = u - x;
    const float v_ratio = v - y;
    const float u_opp = 1 - u_ratio;
    const float v_opp = 1 - v_ratio;
    const int offset = ho.textureIndex;
    f3 result = texel3(a, offset, width, x, y) * u_opp;
    x = iclamp(x + 1, 0, width - 1);
    result = result + texel3(a, offset, width, x, y) * u_ratio;
    result = result * v_opp;
    y = iclamp(y + 1, 0, height - 1);
    f3 result2 = texel3(a, offset, width, x, y) * u_ratio;
    x = iclamp(x - 1, 0, width - 1);
    result2 = result2 + texel3(a, offset, width, x, y) * u_opp;
    result2 = result2 * v_ratio;
    return result + result2;
}

// opencl_kernel.cl:488-545: true when something other than the light blocks the segment
template <int V>
RPT_DEV bool sample_light_occluded(const KernelArgs &a, f4 origin4, f4 dir4, float lightDist, int lightIndex) {
    const f3 nd = normalize(yzw(dir4));
    const f4 lightDir0 = mk4((float)a.interval, nd.x, nd.y, nd.z);
    for (int i = 0; i < a.object_count; i++) {
        if (i != lightIndex) {
            Hit newHit;
            newHit.dist = 1e20f;
            const f4 newEvent0 = transformPoint4D(a.objects[i].Lorentz, origin4);
            const f4 lightDir = transformPoint4D(a.objects[i].Lorentz, lightDir0);
            if (intersect_object<V>(a, i, newEvent0, lightDir, newHit, lightDist)) {
                if (newHit.dist < lightDist) return true;
            }
        }
    }
    return false;
}

// opencl_kernel.cl:361-486 + 548-604: closest hit over the object list, surface colour, lights
// Returns false (and leaves `color` untouched) when the ray hits nothing: the caller then uses the
// per-frame background constants instead of tonemapping (0.15,0.15,0.25) again for every pixel.
template <int V>
RPT_DEV bool trace(const KernelArgs &a, f3 camdir, unsigned long long object_mask, f3 &color_out) {
    const float inf = 1e20f;
    Hit hit;
    hit.dist = inf;
    hit.object = -1;
    const f3 nd = normalize(camdir);
    const f4 rayDir = mk4((float)a.interval, nd.x, nd.y, nd.z);

    for (int i = 0; i < a.object_count; i++) {
        // wave-uniform skip of objects whose bounding volume no ray of this tile can reach (a miss for every
        // lane in the reference too, so skipping it changes nothing)
        if (i < 64 && !((object_mask >> i) & 1ull)) continue;
        Hit newHit;
        newHit.dist = inf;
        bool got;
        if (V == 0) got = intersect_object<0>(a, i, ld4(a.objects[i].stationaryCam), transformPoint4D(a.objects[i].Lorentz, rayDir), newHit);
        else got = intersect_object_primary<V>(a, i, rayDir, newHit);
        if (got) {
            if (newHit.dist < hit.dist) {
                hit = newHit;
                hit.object = i;
            }
        }
    }
    if (hit.object < 0) return false;
#ifdef RPT_DIAGNOSTICS
    if (V == 5) {   // stop after the closest hit (timing of the primary walk alone)
        color_out = mk3(hit.dist, hit.normal.x + hit.uv.x, hit.normal.y + hit.normal.z + hit.uv.y);
        return true;
    }
#endif

    const rpt_object &ho = a.objects[hit.object];
    f3 hcolor = ho.textureIndex != -1 ? sample_texture(a, ho, hit.uv) : ld3(ho.color);
    if (ho.flashPeriod > 0) {   // proper-time flash, opencl_kernel.cl:476-482: event.x of the winning hit
        const float event_x = ho.stationaryCam.x + dot(ld4(ho.Lorentz[0]), rayDir) * hit.dist;
        const float period = ho.flashPeriod;
        const float duration = ho.flashDuration;
        if (event_x - period * __builtin_floorf(event_x / period) < duration) hcolor = hcolor * 2;
    }

    f3 color = hcolor * (a.interval != 0 ? a.ambient : 1.0f);
    if (ho.light) color = color + hcolor;
    if (a.interval != 0) {
        for (int i = 0; i < a.object_count; i++) {
            if (i != hit.object && a.objects[i].light) {
                const rpt_object &lo = a.objects[i];
                const f4 cameraPos_ObjFrame = ld4(ho.stationaryCam);
                const f4 rayDir_ObjFrame = transformPoint4D(ho.Lorentz, rayDir);
                f4 hitPos_ObjFrame = cameraPos_ObjFrame + rayDir_ObjFrame * hit.dist;
                hitPos_ObjFrame = hitPos_ObjFrame + mk4(0, hit.normal.x * 0.001f, hit.normal.y * 0.001f, hit.normal.z * 0.001f);
                const f4 hitPos = transformPoint4D(ho.InvLorentz, hitPos_ObjFrame);
                const f4 hitPos_LightFrame = transformPoint4D(lo.Lorentz, hitPos);
                const f3 lightPos3_LightFrame = mk3(lo.M[0].w, lo.M[1].w, lo.M[2].w);
                const f3 lightDir3_LightFrame = lightPos3_LightFrame - yzw(hitPos_LightFrame);
                const f4 lightDir_LightFrame = mk4(a.interval * length(lightDir3_LightFrame), lightDir3_LightFrame.x,
                                                   lightDir3_LightFrame.y, lightDir3_LightFrame.z);
                const f4 lightDir = transformPoint4D(lo.InvLorentz, lightDir_LightFrame);
                const f4 lightDir_ObjFrame = transformPoint4D(ho.Lorentz, lightDir);
                const f3 lightDir3_ObjFrame = yzw(lightDir_ObjFrame);
                const f3 unitLightDir3 = normalize(lightDir3_ObjFrame);
                const float ndotl = dot(hit.normal, unitLightDir3);
                if (ndotl > 0) {
                    const f3 ld = normalize(yzw(lightDir));
                    const f4 shadowDir = mk4((float)a.interval, ld.x, ld.y, ld.z);
                    if (!sample_light_occluded<V>(a, hitPos, shadowDir, length(yzw(lightDir)), i)) {
                        const float k = ndotl / (1.0f + 0.1f * length(lightDir3_ObjFrame) +
                                                 0.01f * dot(lightDir3_ObjFrame, lightDir3_ObjFrame));
                        color = color + hcolor * k * ld3(lo.color);
                    }
                }
            }
        }
    }
    color_out = color;
    return true;
}

// opencl_kernel.cl:607-616
RPT_DEV float hable1(float x) {
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}

RPT_DEV uint32_t to_u8(float c) {   // (unsigned char)(c * 255): saturating, NaN -> 0
    const float t = c * 255;
    if (!(t == t)) return 0u;
    if (t <= 0.0f) return 0u;
    if (t >= 255.0f) return 255u;
    return (uint32_t)(int)t;
}

// tonemap + pack of opencl_kernel.cl:649-657; returns the little-endian R,G,B,1 word
RPT_DEV uint32_t tonemap_pack(const KernelArgs &a, f3 color, f3 &mapped) {
    mapped.x = cl_min(hable1(color.x) / a.hable_wp[0], 1.0f);
    mapped.y = cl_min(hable1(color.y) / a.hable_wp[1], 1.0f);
    mapped.z = cl_min(hable1(color.z) / a.hable_wp[2], 1.0f);
    return to_u8(mapped.x) | (to_u8(mapped.y) << 8) | (to_u8(mapped.z) << 16) | (1u << 24);
}

// The wavefront's object mask, computed by the wavefront itself: lane i compares the image-plane rectangle of object i
// (rpt_screen_bounds.hpp: outside it no primary ray reaches the object; where it pays, an octagon: the rectangle with
// corners cut by two diagonal slabs) with the wave's 8x8-pixel tile, grown by a pixel
// and a half on every side, and one __ballot makes the 64 answers the mask — in SGPRs, wave-uniform, with no prepass
// kernel, no mask buffer and no dependent load behind it.  Pixel (x, y) looks through the plane point
// ((x/W - 0.5) * aspect, y/H - 0.5) (opencl_kernel.cl:57-63).  NaNs compare false, so a broken rectangle keeps its object.
// One 16-byte framebuffer pixel, written with a NON-TEMPORAL store (global_store_dwordx4 ... nt): the framebuffer is written once
// and never read by these kernels, and a 4K frame is 133 MB against 4 MB of L2 per XCD — written with the default policy the
// stream of pixels competes with the octree and the triangle records the walks live on.  Measured A/B/A/B
// (profiles/r02_nontemporal_store_ab.txt): bunny 4K 0.109 -> 0.098 ms per frame in flight, 0.218 -> 0.208 one at a time.
RPT_DEV void store_pixel(void *out16, size_t id, uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    v4u pv;
    pv.x = x; pv.y = y; pv.z = z; pv.w = w;
    __builtin_nontemporal_store(pv, reinterpret_cast<v4u *>(out16) + id);
}

RPT_DEV unsigned long long wave_object_mask(const KernelArgs &a, int tile_x0, int tile_y0) {
    const int lane = threadIdx.x & 63;
    // one 16-B load per lane, issued unconditionally (lanes beyond the object count re-read rectangle 0), and four compares
    // without branches: one memory round trip, no divergence.  No early return for a scene without objects either: the buffer
    // behind `rects` always holds at least one record's worth of bytes (rpt_api.hip: reserve), and a branch here would put the
    // loads of `rects` and of the reciprocals behind it — one more dependent round trip in every wave.
    const int n = a.object_count;
    const int slot = (lane < n) ? lane : 0;
    const float4 r = a.rects[2 * slot];
    const float iw = a.inv_width, ih = a.inv_height;
    const float tu0 = (((float)tile_x0 - 1.5f) * iw - 0.5f) * a.aspect, tu1 = (((float)tile_x0 + 8.5f) * iw - 0.5f) * a.aspect;
    const float tv0 = ((float)tile_y0 - 1.5f) * ih - 0.5f, tv1 = ((float)tile_y0 + 8.5f) * ih - 0.5f;
    bool outside = (r.z < tu0) | (r.x > tu1) | (r.w < tv0) | (r.y > tv1);
    if (a.diagonals) {      // wave-uniform: the octagon's four diagonal sides (u + v and u - v over the tile's corners)
        const float4 g = a.rects[2 * slot + 1];
        outside = outside | (g.y < tu0 + tv0) | (g.x > tu1 + tv1) | (g.w < tu0 - tv1) | (g.z > tu1 - tv0);
    }
    const bool keep = (lane < n) & !outside;
    return __ballot(keep);
}

template <int V> RPT_DEV constexpr bool culled_variant() { return V >= 20; }
template <int V> RPT_DEV constexpr bool zorder_lanes() { return V == 641 || V == 653; }
template <int V> RPT_DEV constexpr bool one_wave_workgroups() { return V == 0 || V == 1 || V == 20 || V == 23 || V == 24 || V == 657 || V == 669 || V == 673 || V == 705 || V == 717; }     // the product kernels (+ three arms re-measured that way)
template <int V> RPT_DEV constexpr bool band_first_variant() { return V == 23 || V == 123 || (V >= 256 && V < 1000 && (V & 8)); }

// ---------------------------------------------------------------------------------------------
// One thread per pixel, wave = 8x8 tile, workgroup = 32x8 strip.
//   V = 0: reads the reference layouts only (general fallback, any valid octree; no culling)
//   V = 1: derived layouts, every object tested for every pixel (the un-culled form rpt_verify_frame compares with)
//   V = 20: derived layouts + the wave's object mask from per-object image-plane rectangles + __ballot (rpt_render_async)
//   V = 23: 20 with the band of tile rows that holds the meshes dispatched first and the pipelined walk (the blocking rpt_render)
//   V = 24: 20 without the octree walk compiled in (frames whose Object[] holds no mesh)
//   other values: diagnostics build only (rpt_diag_kernels.hip.h)
template <int V>
RPT_DEV void render_pixel_body(const KernelArgs &a) {
    const int lane = threadIdx.x & 63;
    // The product kernels are launched ONE WAVE per workgroup (blockDim 64, grid.x = tiles per row): a wave slot is handed back when
    // its wave ends, not when the longest of four neighbours does — next to a tile that walks for 70 us sit tiles that only store
    // (profiles/r03_one_wave_workgroups_ab.txt: bunny 4K in flight -8 %, one at a time -3 %).  The measurement arms keep 4 x 64.
    const int wave = one_wave_workgroups<V>() ? ((int)blockIdx.x & 3) : (int)(threadIdx.x >> 6);
    const int strip = one_wave_workgroups<V>() ? ((int)blockIdx.x >> 2) : (int)blockIdx.x;      // 32-pixel-wide strip of that row
#ifdef RPT_DIAGNOSTICS
    const DiagWaveClock diag_clock0 = diag_wave_begin<V>();
#endif
    int tile_row = (int)blockIdx.y;              // 8-row tiles of this context, natural order
    if (band_first_variant<V>() && a.first_h > 0) {
        // Workgroups are handed out in the order of their linear index, i.e. row of strips by row of strips.  One frame at a
        // time, what ends the frame is the last of its long waves, so the band of tile rows that holds the meshes goes first
        // (whole rows, in their natural order: neighbours stay neighbours) and the other rows follow in order.
        const int y = (int)blockIdx.y, rh = a.first_h;
        tile_row = y < rh ? a.first_ty + y : (y - rh < a.first_ty ? y - rh : y);
    }
    // lane -> pixel of the wave's 8x8 tile: row by row.  Diagnostics library, arms 641 / 653: along the Z curve, so that the four lanes the
    // memory pipeline handles together are a 2x2 block of pixels, not a 4x1 run (level, +1 % in flight at 4K and 8K: r03_td_bound.txt)
#ifdef RPT_DIAGNOSTICS
    const int col_in_tile = zorder_lanes<V>() ? ((lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4)) : (lane & 7);
    const int row_in_tile = zorder_lanes<V>() ? (((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4)) : (lane >> 3);
    const int x_coord = strip * 32 + wave * 8 + col_in_tile;
#else
    const int row_in_tile = lane >> 3;
    const int x_coord = strip * 32 + wave * 8 + (lane & 7);
#endif
    const int local_row = tile_row * RPT_TILE_ROWS + row_in_tile;
    const int global_tile = (tile_row >> a.run_log2) * a.tile_step + a.first_tile + (tile_row & ((1 << a.run_log2) - 1));
    const int y_coord = global_tile * RPT_TILE_ROWS + row_in_tile;
    // the wave's object mask comes from a __ballot over ALL 64 lanes (lane i answers for object i), so it is formed
    // before the lanes of a partial tile leave
    unsigned long long object_mask = ~0ull;
    if (culled_variant<V>()) object_mask = wave_object_mask(a, strip * 32 + wave * 8, global_tile * RPT_TILE_ROWS);
#ifdef RPT_DIAGNOSTICS
    if (V == 10) object_mask = a.tile_masks[__builtin_amdgcn_readfirstlane(tile_row * a.mask_tiles_x + (int)blockIdx.x * 4 + wave)];   // the prepass's per-tile mask
#endif
    if (x_coord >= a.width || y_coord >= a.height) return;   // the reference has no guard (UB)

    f3 color;
    f3 mapped = mk3(0.0f, 0.0f, 0.0f);
    bool traced = false;
    uint32_t packed = a.bg_packed;
    const bool masked = culled_variant<V>() || V == 10;
    if (!masked || object_mask != 0 || a.object_count > 64) {
        const f3 camdir = createCamRayDir((float)x_coord, (float)y_coord, a.width, a.height, a.aspect);
        if (trace<V>(a, camdir, object_mask, color)) {
            packed = tonemap_pack(a, color, mapped);
            traced = true;
        }
    }

    const size_t id = (size_t)y_coord * a.width + x_coord;
#ifdef RPT_DIAGNOSTICS
    if (V == 785 && object_mask == 0ull) return;      // EXPERIMENT (wrong image): what do the sky tiles' stores cost the walks?
#endif
    if (a.out16) store_pixel(a.out16, id, __float_as_uint((float)x_coord), __float_as_uint((float)y_coord), packed, 0u);
    // (the 4-byte plane likewise: measured against the default policy on a rank's share of the frame, bunny 4K 0.0581 -> 0.0566 ms, shadows the same)
    if (a.plane) __builtin_nontemporal_store(packed, a.plane + (size_t)local_row * a.width + x_coord);
    if (a.debug_rgb) {
        if (!traced) mapped = mk3(a.bg_mapped[0], a.bg_mapped[1], a.bg_mapped[2]);      // (read here only: a miss pixel's store needs nothing beyond the first line of the arguments)
        a.debug_rgb[3 * id + 0] = mapped.x;
        a.debug_rgb[3 * id + 1] = mapped.y;
        a.debug_rgb[3 * id + 2] = mapped.z;
    }
#ifdef RPT_DIAGNOSTICS
    diag_wave_end<V>(a, diag_clock0);
#endif
}

// opencl_kernel.cl:641-648 with MSAASAMPLES = a.msaa > 1 (a compile-time constant of the reference, 1 as shipped; rpt_set_msaa): a.msaa^2
// camera rays per pixel at (x + sx/n, y + sy/n), colours summed in the reference's order (a miss contributes the background of :565)
// and divided by n^2 before the tonemap.  A function of its own so that the one-sample kernels stay what they are.  V = 20: the
// wave's object mask as in render_pixel_body — the tile it is tested against is grown by a pixel and a half, the samples stay
// within one pixel; V = 1: no cull (what rpt_verify_frame compares with).
template <int V>
RPT_DEV void render_pixel_body_msaa(const KernelArgs &a) {
    const int lane = threadIdx.x & 63;
    const int wave = (int)blockIdx.x & 3;              // one wave per workgroup, as the one-sample product kernels
    const int tile_row = (int)blockIdx.y;
    const int strip = (int)blockIdx.x >> 2;
    const int row_in_tile = lane >> 3;
    const int x_coord = strip * 32 + wave * 8 + (lane & 7);
    const int local_row = tile_row * RPT_TILE_ROWS + row_in_tile;
    const int global_tile = (tile_row >> a.run_log2) * a.tile_step + a.first_tile + (tile_row & ((1 << a.run_log2) - 1));
    const int y_coord = global_tile * RPT_TILE_ROWS + row_in_tile;
    unsigned long long object_mask = ~0ull;
    if (culled_variant<V>()) object_mask = wave_object_mask(a, strip * 32 + wave * 8, global_tile * RPT_TILE_ROWS);
    if (x_coord >= a.width || y_coord >= a.height) return;
    const bool any = !culled_variant<V>() || object_mask != 0 || a.object_count > 64;
    const int n = a.msaa;
    f3 sum = mk3(0.0f, 0.0f, 0.0f);
    for (int sy = 0; sy < n; sy++) {
        for (int sx = 0; sx < n; sx++) {
            f3 c = mk3(0.15f, 0.15f, 0.25f);
            if (any) {
                const f3 camdir = createCamRayDir((float)x_coord + (float)sx / (float)n, (float)y_coord + (float)sy / (float)n, a.width, a.height, a.aspect);
                f3 traced;
                if (trace<V>(a, camdir, object_mask, traced)) c = traced;
            }
            sum = sum + c;
        }
    }
    const float n2 = (float)(n * n);
    sum = mk3(sum.x / n2, sum.y / n2, sum.z / n2);
    f3 mapped;
    const uint32_t packed = tonemap_pack(a, sum, mapped);
    const size_t id = (size_t)y_coord * a.width + x_coord;
    if (a.out16) store_pixel(a.out16, id, __float_as_uint((float)x_coord), __float_as_uint((float)y_coord), packed, 0u);
    if (a.plane) __builtin_nontemporal_store(packed, a.plane + (size_t)local_row * a.width + x_coord);
    if (a.debug_rgb) {
        a.debug_rgb[3 * id + 0] = mapped.x;
        a.debug_rgb[3 * id + 1] = mapped.y;
        a.debug_rgb[3 * id + 2] = mapped.z;
    }
}

#ifndef RPT_RELAXED_FP    /* rpt_relaxed.hip instantiates its own two kernels and nothing else from here on */
// Product kernels (rpt_set_variant; the number in the comment is the variant).
__global__ __launch_bounds__(64) void rpt_render_kernel_v0(const KernelArgs a) { render_pixel_body<0>(a); }                                                              // 1: any valid octree
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5))) void rpt_render_kernel_unculled_w5(const KernelArgs a) { render_pixel_body<1>(a); }         // 3: no cull (rpt_verify_frame; the escape hatch)
// the wave's object mask from the per-object screen rectangles by lane-parallel test + __ballot, 5 waves per SIMD (96 VGPRs, 8 B of scratch outside the loops)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5))) void rpt_render_kernel_ballot_w5(const KernelArgs a) { render_pixel_body<20>(a); }          // 41 = rpt_render_async
// the same with the tile rows that hold the meshes dispatched first and the latency walk (44 B of scratch): latency, not throughput
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5))) void rpt_render_kernel_ballot_first_w5(const KernelArgs a) { render_pixel_body<23>(a); }    // 43 = the blocking rpt_render; rpt_render_async below RPT_LATENCY_KERNEL_MAX_PIXELS
// without the octree walk compiled in, for frames whose Object[] holds no mesh: 61 VGPRs, no scratch, EIGHT waves per SIMD
// (arch 1080p 0.0370 -> 0.0301 ms per frame in flight, cubes.txt 4K 0.0898 -> 0.0725)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void rpt_render_kernel_analytic_w8(const KernelArgs a) { render_pixel_body<24>(a); }        // 44

// MSAASAMPLES > 1 (rpt_set_msaa): culled and un-culled; rpt_last_variant reports them as 46 / 47
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5))) void rpt_render_kernel_msaa_w5(const KernelArgs a) { render_pixel_body_msaa<20>(a); }            // 46
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5))) void rpt_render_kernel_msaa_unculled_w5(const KernelArgs a) { render_pixel_body_msaa<1>(a); }    // 47

#ifdef RPT_DIAGNOSTICS
}  // namespace rptd
#include "rpt_diag_kernels.hip.h"    /* librpt_hip_diag.so only: instrumented kernels, round 1's prepass, the A/B arms of rounds 2 and 3 */
namespace rptd {
#endif

// Root-side reassembly after the gather: plane of rank r, local tile k -> global tile r + k*n_ranks.
__global__ __launch_bounds__(256) void rpt_scatter_plane_kernel(const uint32_t *planes, rpt_pixel *out16, int width,
                                                               int height, int n_ranks, size_t plane_stride_words) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= width || y >= height) return;
    const int tile = y / RPT_TILE_ROWS;
    const int rank = tile % n_ranks;
    const int local_row = (tile / n_ranks) * RPT_TILE_ROWS + (y % RPT_TILE_ROWS);
    const uint32_t packed = planes[(size_t)rank * plane_stride_words + (size_t)local_row * width + x];
    uint4 px;
    px.x = __float_as_uint((float)x);
    px.y = __float_as_uint((float)y);
    px.z = packed;
    px.w = 0u;
    store_pixel(out16, (size_t)y * width + x, px.x, px.y, px.z, px.w);
}

// The exchange carries 3 bytes per pixel: the fourth byte of every packed colour is the constant 1
// (opencl_kernel.cl:657).  Four pixels (four words) of a colour plane become three words, R0 G0 B0 R1 | G1 B1 R2 G2 |
// B2 R3 G3 B3; a plane's pixel count is a multiple of 8 (whole 8-row tiles).
__global__ __launch_bounds__(256) void rpt_pack_plane3_kernel(const uint4 *plane4, uint32_t *plane3, size_t quads) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= quads) return;
    const uint4 p = plane4[q];
    plane3[3 * q + 0] = (p.x & 0xffffffu) | (p.y << 24);
    plane3[3 * q + 1] = ((p.y >> 8) & 0xffffu) | (p.z << 16);
    plane3[3 * q + 2] = ((p.z >> 16) & 0xffu) | (p.w << 8);
}

// Root-side reassembly of gathered 3-byte planes (rpt_scatter_plane_kernel for the packed form).
__global__ __launch_bounds__(256) void rpt_scatter_plane3_kernel(const uint8_t *planes, rpt_pixel *out16, int width, int height,
                                                                int n_ranks, size_t plane_stride_bytes) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= width || y >= height) return;
    const int tile = y / RPT_TILE_ROWS;
    const int rank = tile % n_ranks;
    const int local_row = (tile / n_ranks) * RPT_TILE_ROWS + (y % RPT_TILE_ROWS);
    const uint8_t *src = planes + (size_t)rank * plane_stride_bytes + 3 * ((size_t)local_row * width + x);
    uint4 px;
    px.x = __float_as_uint((float)x);
    px.y = __float_as_uint((float)y);
    px.z = (uint32_t)src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16) | (1u << 24);
    px.w = 0u;
    store_pixel(out16, (size_t)y * width + x, px.x, px.y, px.z, px.w);
}

// Reassembly for the weighted split (rpt_set_tile_pattern): per period of `period` tiles the root renders the first
// `root_run` straight into the framebuffer, helper j (1..n_ranks-1) the tile root_run + j - 1 into its 3-byte plane
// (local tile = period index).  Only the helpers' tiles are written here; the root's are already in place.
__global__ __launch_bounds__(256) void rpt_scatter_helper_planes3_kernel(const uint8_t *planes, rpt_pixel *out16, int width, int height,
                                                                        int period, int root_run, size_t plane_stride_bytes) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= width || y >= height) return;
    const int tile = y / RPT_TILE_ROWS;
    const int slot = tile % period;
    if (slot < root_run) return;
    const int rank = slot - root_run + 1;
    const int local_row = (tile / period) * RPT_TILE_ROWS + (y % RPT_TILE_ROWS);
    const uint8_t *src = planes + (size_t)rank * plane_stride_bytes + 3 * ((size_t)local_row * width + x);
    uint4 px;
    px.x = __float_as_uint((float)x);
    px.y = __float_as_uint((float)y);
    px.z = (uint32_t)src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16) | (1u << 24);
    px.w = 0u;
    store_pixel(out16, (size_t)y * width + x, px.x, px.y, px.z, px.w);
}

// rpt_verify_frame: how many packed colours differ between two colour planes (one ballot + popcount per wave-iteration)
__global__ __launch_bounds__(256) void rpt_count_differences_kernel(const uint32_t *p, const uint32_t *q, size_t words, unsigned long long *out) {
    unsigned long long mine = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) mine += __popcll(__ballot(p[i] != q[i]));
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(out, mine);     // (every lane of a wave holds the wave's count: lane 0 reports it)
}

// Known-answer probes of single device functions.
__global__ void rpt_probe_kernel(int which, const float *in, float *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (which == 0) {
        const float *p = in + 15 * i;
        Ray r;
        r.origin = mk3(p[9], p[10], p[11]);
        r.dir = mk3(p[12], p[13], p[14]);
        float dist = 0;
        f2 uv = {0, 0};
        const bool h = intersect_triangle(mk3(p[0], p[1], p[2]), mk3(p[3], p[4], p[5]), mk3(p[6], p[7], p[8]), r, dist, uv);
        out[4 * i + 0] = h ? 1.0f : 0.0f;
        out[4 * i + 1] = h ? dist : 0;
        out[4 * i + 2] = h ? uv.x : 0;
        out[4 * i + 3] = h ? uv.y : 0;
    } else if (which == 1) {
        const float *p = in + 12 * i;
        Ray r;
        r.origin = mk3(p[6], p[7], p[8]);
        r.dir = mk3(p[9], p[10], p[11]);
        f2 d = {0, 0};
        int cs = 0, fs = 0;
        const bool h = intersect_AABB(mk3(p[0], p[1], p[2]), mk3(p[3], p[4], p[5]), r, d, cs, fs);
        out[5 * i + 0] = h ? 1.0f : 0.0f;
        out[5 * i + 1] = h ? d.x : 0;
        out[5 * i + 2] = h ? d.y : 0;
        out[5 * i + 3] = h ? (float)cs : 0;
        out[5 * i + 4] = h ? (float)fs : 0;
    } else if (which == 2) {
        const float *p = in + 4 * i;
        const int w = (int)p[2], hgt = (int)p[3];
        const f3 dir = createCamRayDir(p[0], p[1], w, hgt, (float)w / (float)hgt);
        out[3 * i + 0] = dir.x;
        out[3 * i + 1] = dir.y;
        out[3 * i + 2] = dir.z;
    } else if (which == 3) {
        out[3 * i + 0] = hable1(in[3 * i + 0]);
        out[3 * i + 1] = hable1(in[3 * i + 1]);
        out[3 * i + 2] = hable1(in[3 * i + 2]);
    } else if (which == 4) {    // asin(a), atan2(b, c) of the textured-sphere (u,v)
        out[2 * i + 0] = rpt_asinf(in[3 * i + 0]);
        out[2 * i + 1] = rpt_atan2f(in[3 * i + 1], in[3 * i + 2]);
    } else {    // the walk's two pure steps: exit face of a leaf, child selection (general and fast form)
        const float *p = in + 6 * i;
        f3 uv = mk3(p[3], p[4], p[5]);
        const int side = getOppositeBoxSide(makeExitPlan(mk3(p[0], p[1], p[2])), uv);
        out[12 * i + 0] = (float)side; out[12 * i + 1] = uv.x; out[12 * i + 2] = uv.y; out[12 * i + 3] = uv.z;
        f3 a = mk3(p[3], p[4], p[5]), b = a;
        const int ca = octree_child_step(a), cb = octree_child_step_fast(b);
        out[12 * i + 4] = (float)ca; out[12 * i + 5] = a.x; out[12 * i + 6] = a.y; out[12 * i + 7] = a.z;
        out[12 * i + 8] = (float)cb; out[12 * i + 9] = b.x; out[12 * i + 10] = b.y; out[12 * i + 11] = b.z;
    }
}

// rpt_probe_division: are the shared-reciprocal quotients of rpt_device_math.hip.h equal to IEEE division bit for bit?  Every thread
// draws `per_thread` (numerators, denominator) sets from a counter-based generator and compares div3_shared_unguarded<1> and <2>
// with x / s on those INSIDE the fast path's domain, and the guarded div3_shared<2> on ALL of them.  mode 0: random significands,
// exponents over the whole domain (and beyond it for the guarded form); 1: the same with the denominator's significand all ones;
// 2: normalize() itself — a random vector, s = sqrt(dot(v, v)) as length() forms it; 3: arbitrary bit patterns (NaN, infinities,
// zeros, denormals: only the guarded form is compared).  counts[0..3] = sets compared unguarded, mismatching quotients with one
// round, with two rounds, mismatches of the guarded form; the first few mismatching sets go to `samples` (4 floats each).
RPT_DEV uint32_t probe_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ __launch_bounds__(256) void rpt_probe_division_kernel(int mode, uint32_t seed, int per_thread, unsigned long long *counts, float *samples, int max_samples) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n_cmp = 0, bad1 = 0, bad2 = 0, badg = 0;
    for (int it = 0; it < per_thread; it++) {
        const uint32_t base = probe_hash(seed ^ probe_hash(tid * 0x9e3779b9u + (uint32_t)it));
        uint32_t w[4];
        for (int k = 0; k < 4; k++) w[k] = probe_hash(base + 0x632be5abu * (uint32_t)(k + 1));
        float v[4];
        if (mode == 3) {
            for (int k = 0; k < 4; k++) v[k] = __uint_as_float(w[k]);
        } else {
            for (int k = 0; k < 4; k++) {
                const uint32_t sign = w[k] & 0x80000000u, mant = (mode == 1 && k == 3) ? 0x7fffffu : (w[k] & 0x7fffffu);
                const int span = k == 3 ? 100 : 140;                      // exponents: the domain and a little beyond
                const int e = 127 - span / 2 + (int)((w[k] >> 23) % (uint32_t)span);
                v[k] = __uint_as_float(sign | ((uint32_t)e << 23) | mant);
            }
            if (mode == 2) {
                const f3 t = mk3(v[0], v[1], v[2]);
                v[3] = length(t);
            }
        }
        const f3 a = mk3(v[0], v[1], v[2]);
        const float s = v[3];
        const f3 ref = mk3(a.x / s, a.y / s, a.z / s);
        auto same = [](float p, float q) { return __float_as_uint(p) == __float_as_uint(q) || (p != p && q != q); };
        bool report = false;
        if (div3_shared_domain(a, s)) {
            n_cmp++;
            const f3 q1 = div3_shared_unguarded<1>(a, s), q2 = div3_shared_unguarded<2>(a, s);
            const int b1 = !same(q1.x, ref.x) + !same(q1.y, ref.y) + !same(q1.z, ref.z);
            const int b2 = !same(q2.x, ref.x) + !same(q2.y, ref.y) + !same(q2.z, ref.z);
            bad1 += b1; bad2 += b2;
            report = b2 != 0 || (b1 != 0 && mode != 1);
        }
        const f3 g = div3_shared<2>(a, s);
        const int bg = !same(g.x, ref.x) + !same(g.y, ref.y) + !same(g.z, ref.z);
        badg += bg;
        if ((report || bg) && samples) {
            const unsigned long long slot = atomicAdd(&counts[4], 1ull);
            if (slot < (unsigned long long)max_samples) { float *o = samples + 4 * slot; o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = s; }
        }
    }
    atomicAdd(&counts[0], n_cmp); atomicAdd(&counts[1], bad1); atomicAdd(&counts[2], bad2); atomicAdd(&counts[3], badg);
}

// Known-answer probes at OBJECT level (rpt_probe_object; the oracle's counterpart is rpt_oracle_object_rays): which =
//   0: one 4-D ray {origin4, dir4} in the rest frame of object `object` through intersect_object, the general form every shadow ray
//      and the V = 0 kernel's primary rays take: out 8 = {hit, dist, normal.xyz, uv.xy, 0}   (opencl_kernel.cl:312-359, 200-308)
//   1: sample_light on a shadow ray {origin4, dir4, lightDist} of the camera frame with light `object`: out 2 = {occluded as the
//      un-culled kernel decides it, occluded as the culled kernels decide it (segment culls, __ballot over the wave)}   (:488-545)
//   2: the transforms on {x, y, z, w}: out 16 = transformPoint(InvM), transformPoint4D(Lorentz), transformDirection(InvM),
//      applyTranspose(InvM) of object `object`   (:75-104)
//   3: a primary ray with camera direction {x, y, z} (normalised here as trace() does) through intersect_object_primary, the form
//      the default kernels use (origin and constants from the per-frame DObj record): out 8 as in 0
__global__ __launch_bounds__(64) void rpt_probe_object_kernel(const KernelArgs a, int which, int object, const float *in, float *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = i < n ? i : n - 1;          // (idle lanes repeat the last ray: the culled forms ballot over whole waves)
    if (which == 0 || which == 3) {
        Hit hit;
        hit.dist = 1e20f;
        hit.normal = mk3(0.0f, 0.0f, 0.0f);
        hit.uv.x = hit.uv.y = 0.0f;
        hit.object = -1;
        bool h;
        if (which == 0) {
            const float *p = in + 8 * (size_t)j;
            h = intersect_object<1>(a, object, mk4(p[0], p[1], p[2], p[3]), mk4(p[4], p[5], p[6], p[7]), hit);
        } else {
            const float *p = in + 3 * (size_t)j;
            const f3 nd = normalize(mk3(p[0], p[1], p[2]));
            h = intersect_object_primary<20>(a, object, mk4((float)a.interval, nd.x, nd.y, nd.z), hit);
        }
        if (i < n) {
            float *o = out + 8 * (size_t)i;
            o[0] = h ? 1.0f : 0.0f;
            o[1] = h ? hit.dist : 0.0f;
            o[2] = h ? hit.normal.x : 0.0f; o[3] = h ? hit.normal.y : 0.0f; o[4] = h ? hit.normal.z : 0.0f;
            o[5] = h ? hit.uv.x : 0.0f; o[6] = h ? hit.uv.y : 0.0f;
            o[7] = 0.0f;
        }
    } else if (which == 1) {
        const float *p = in + 9 * (size_t)j;
        const f4 o4 = mk4(p[0], p[1], p[2], p[3]), d4 = mk4(p[4], p[5], p[6], p[7]);
        const bool plain = sample_light_occluded<1>(a, o4, d4, p[8], object);
        const bool culled = sample_light_occluded<20>(a, o4, d4, p[8], object);
        if (i < n) { out[2 * (size_t)i] = plain ? 1.0f : 0.0f; out[2 * (size_t)i + 1] = culled ? 1.0f : 0.0f; }
    } else if (i < n) {
        const float *p = in + 4 * (size_t)i;
        float *o = out + 16 * (size_t)i;
        const rpt_object &obj = a.objects[object];
        const f3 v = mk3(p[0], p[1], p[2]);
        const f3 t0 = transformPoint(obj.InvM, v);
        const f4 t1 = transformPoint4D(obj.Lorentz, mk4(p[0], p[1], p[2], p[3]));
        const f3 t2 = transformDirection(obj.InvM, v);
        const f3 t3 = applyTranspose(obj.InvM, v);
        o[0] = t0.x; o[1] = t0.y; o[2] = t0.z; o[3] = 0.0f;
        o[4] = t1.x; o[5] = t1.y; o[6] = t1.z; o[7] = t1.w;
        o[8] = t2.x; o[9] = t2.y; o[10] = t2.z; o[11] = 0.0f;
        o[12] = t3.x; o[13] = t3.y; o[14] = t3.z; o[15] = 0.0f;
    }
}

// Known-answer probe of the octree walk at RAY level: every ray (object-space origin and direction) through the three walks the
// product library holds — the reference's layouts (octree_core_ref), the throughput walk (octree_walk<false, false>) and the latency
// walk (octree_walk<true, true>) — with the hit re-measured from the origin (0, 0, 0) at unit direction length.  8 floats per walk
// and ray: hit flag, dist, normal.xyz, uv.xy, 0.
__global__ __launch_bounds__(256) void rpt_probe_walk_kernel(const KernelArgs a, int object, const float *rays, float *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r;
    r.origin = mk3(rays[6 * i + 0], rays[6 * i + 1], rays[6 * i + 2]);
    r.dir = mk3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    const rpt_object &obj = a.objects[object];
    const int root = a.dobjs[object].root;
    for (int w = 0; w < 3; w++) {
        Hit hit;
        hit.dist = 1e20f;
        hit.normal = mk3(0.0f, 0.0f, 0.0f);
        hit.uv.x = hit.uv.y = 0.0f;
        hit.object = -1;
        const bool h = w == 0 ? octree_core_ref(a, obj, r, mk3(0.0f, 0.0f, 0.0f), 1.0f, hit)
                     : w == 1 ? octree_walk<false, false, true, false, true>(a, obj, root, r, mk3(0.0f, 0.0f, 0.0f), 1.0f, hit)
                              : octree_walk<true, true, false, false, true>(a, obj, root, r, mk3(0.0f, 0.0f, 0.0f), 1.0f, hit);
        float *o = out + ((size_t)i * 3 + w) * 8;
        o[0] = h ? 1.0f : 0.0f;
        o[1] = h ? hit.dist : 0.0f;
        o[2] = h ? hit.normal.x : 0.0f;
        o[3] = h ? hit.normal.y : 0.0f;
        o[4] = h ? hit.normal.z : 0.0f;
        o[5] = h ? hit.uv.x : 0.0f;
        o[6] = h ? hit.uv.y : 0.0f;
        o[7] = 0.0f;
    }
}

#endif  /* !RPT_RELAXED_FP */

}  // namespace rptd
