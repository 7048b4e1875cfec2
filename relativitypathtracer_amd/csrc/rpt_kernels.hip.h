// rpt_kernels.hip.h — the per-pixel render path for gfx950 (CDNA4, wave64).
//
// What is computed is the reference's render_kernel (opencl_kernel.cl:620-660) and everything it
// calls; how it is computed is organised for MI355X:
//   * one wavefront owns an 8x8 pixel tile (coherent rays, 8 full 128-B lines per store wave),
//     a 256-thread workgroup owns a 32x8 strip; the grid is (ceil(W/32), row tiles) so it is
//     >> 256 workgroups at every benchmark resolution;
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
};
static_assert(sizeof(DObj) == 96, "DObj");

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
    const int *links;               // DNode::link of every node again, 4 B apart: what a descent reads below the levels held in LDS
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
// Node<1>: the derived 64-B DNode + gathered DTri records (one dependent load per node, one per
//          triangle).  Needs consecutive children, which the library checks at upload.
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

template <> struct NodeRef<1> {
    // the 64-B record as four 16-B loads held in scalars (no struct copy: keeps it in registers)
    v4f lo, hi;         // min.xyz | firstChild , max.xyz | leafBegin   (ints carried as float bits)
    v4i q2, q3;         // leafCount, nb[0..2] , nb[3..5], pad
    RPT_DEV void load(const KernelArgs &a, int i) {
        const v4f *p = reinterpret_cast<const v4f *>(a.dnodes + i);
        lo = p[0];
        hi = p[1];
        q2 = reinterpret_cast<const v4i *>(p)[2];
        q3 = reinterpret_cast<const v4i *>(p)[3];
    }
    RPT_DEV f3 bmin(const KernelArgs &) const { return mk3(lo.x, lo.y, lo.z); }
    RPT_DEV f3 bmax(const KernelArgs &) const { return mk3(hi.x, hi.y, hi.z); }
    RPT_DEV int link() const { return __float_as_int(lo.w); }
    RPT_DEV bool is_leaf(const KernelArgs &) const { return link() == -1; }
    RPT_DEV int child(const KernelArgs &, int k) const { return (link() & RPT_LINK_CHILD_MASK) + k; }
    RPT_DEV int neighbor(const KernelArgs &, int side) const {   // select chain: no dynamic register indexing
        int r = q2.y;
        r = side == 1 ? q2.z : r;
        r = side == 2 ? q2.w : r;
        r = side == 3 ? q3.x : r;
        r = side == 4 ? q3.y : r;
        r = side == 5 ? q3.z : r;
        return r;
    }
    RPT_DEV int tri_begin(const KernelArgs &) const { return __float_as_int(hi.w); }
    RPT_DEV int tri_count(const KernelArgs &) const { return q2.x; }
    RPT_DEV void tri(const KernelArgs &a, int k, f3 &A, f3 &v0v1, f3 &v0v2, int &id) const {
        const v4f *p = reinterpret_cast<const v4f *>(a.dtris + k);
        const v4f t0 = p[0], t1 = p[1], t2 = p[2];
        A = mk3(t0.x, t0.y, t0.z);
        v0v1 = mk3(t0.w, t1.x, t1.y);
        v0v2 = mk3(t1.z, t1.w, t2.x);
        id = __float_as_int(t2.y);
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

// opencl_kernel.cl:256-308 from the point where the ray is in object space.  newRay = object-space
// ray (direction normalised); world_origin/world_dirlen are ray->origin.yzw and |ray->dir.yzw|.
// Diagnostic cycle accounting (V == 4 only): per wave, shader-clock cycles and wave-level iteration counts of
// the three loops of the walk, accumulated in LDS by the first active lane.
__shared__ unsigned long long rpt_diag_lds[4][8];
template <int V>
RPT_DEV void diag_add(int slot, unsigned long long v) {
    if (V == 4) {
        const unsigned long long m = __ballot(1);
        if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) rpt_diag_lds[threadIdx.x >> 6][slot] += v;
    }
}
template <int V>
RPT_DEV unsigned long long diag_clock() { return V == 4 ? (unsigned long long)clock64() : 0ull; }

// Diagnostic counting (V == 2 only): how many loop iterations lanes need vs. how many the wave executes.
template <int V>
RPT_DEV void count_iter(const KernelArgs &a, int which) {
    if (V == 2) {
        const unsigned long long m = __ballot(1);
        const int lane = threadIdx.x & 63;
        if (lane == __ffsll((long long)m) - 1) {
            atomicAdd(&a.counters[which], (unsigned long long)__popcll(m));
            atomicAdd(&a.counters[3 + which], 1ull);
        }
    }
}

// The walk's stop test (opencl_kernel.cl:283): length(exit point - origin) > hit.dist.  Until a triangle has been hit,
// hit.dist is the caller's 1e20f, and sqrt(s) > 1e20f holds for no finite float s (sqrt(FLT_MAX) < 1.9e19), for s = +inf
// only, and not for NaN: so while no lane of the wave has a hit the square root is not needed to decide it — same answer.
RPT_DEV bool exit_is_past_hit(f3 v, float hit_dist, bool didHit) {
    const float s = dot(v, v);
    if (__ballot(didHit) == 0ull && hit_dist == 1e20f) return s == __builtin_inff();
    return __builtin_sqrtf(s) > hit_dist;
}

template <int V>
RPT_DEV bool octree_core(const KernelArgs &a, const rpt_object &obj, int root, const Ray &newRay, f3 world_origin,
                         float world_dirlen, Hit &hit) {
    NodeRef<(V == 0 ? 0 : 1)> node;   // V >= 1: derived layouts (root = the mesh's root in THEIR numbering, DObj::root)
    int currOctreeIndex = root;
    node.load(a, currOctreeIndex);
    f2 d;
    int closeSide, farSide;
    f3 nmin = node.bmin(a), nmax = node.bmax(a);
    if (!intersect_AABB(nmin, nmax, newRay, d, closeSide, farSide)) return false;
    f3 uv = newRay.origin + newRay.dir * d.x;

    if (d.x < 0) {   // ray starts inside the root: descend to the leaf holding the origin
        uv = (newRay.origin - nmin) / (nmax - nmin);
        while (!node.is_leaf(a)) {
            const int childIndex = V == 0 ? octree_child_step(uv) : octree_child_step_fast(uv);
            currOctreeIndex = node.child(a, childIndex);
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
        count_iter<V>(a, 0);
        if (V == 2) {   // diagnostic: how many DIFFERENT nodes do the active lanes of this wave stand in right now?
            unsigned long long todo = __ballot(1);
            const unsigned long long all = todo;
            int distinct = 0;
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int n0 = __shfl(currOctreeIndex, leader);
                todo &= ~__ballot(currOctreeIndex == n0);
                distinct++;
            }
            if ((int)(threadIdx.x & 63) == __ffsll((long long)all) - 1) {
                atomicAdd(&a.counters[8], (unsigned long long)distinct);
                atomicAdd(&a.counters[9], (unsigned long long)__popcll(all));
                atomicAdd(&a.counters[10 + (distinct <= 1 ? 0 : distinct <= 2 ? 1 : distinct <= 4 ? 2 : distinct <= 8 ? 3 : distinct <= 16 ? 4 : 5)], 1ull);
            }
        }
        const unsigned long long t_leaf0 = diag_clock<V>();
        node.load(a, currOctreeIndex);
        nmin = node.bmin(a);
        nmax = node.bmax(a);
        uv = (uv - nmin) / (nmax - nmin);
        bool descended = false;
        const unsigned long long t_desc0 = diag_clock<V>();
        while (!node.is_leaf(a)) {
            const int childIndex = V == 0 ? octree_child_step(uv) : octree_child_step_fast(uv);
            currOctreeIndex = node.child(a, childIndex);
            node.load(a, currOctreeIndex);
            descended = true;
            count_iter<V>(a, 2);
            diag_add<V>(3, 1);
        }
        const unsigned long long t_tri0 = diag_clock<V>();
        diag_add<V>(2, t_tri0 - t_desc0);
        if (descended) {
            nmin = node.bmin(a);
            nmax = node.bmax(a);
        }
        const int trisIndex = node.tri_begin(a);
        const int trisEnd = trisIndex + node.tri_count(a);
        for (int i = trisIndex; i < trisEnd; i++) {
            f3 A, v0v1, v0v2;
            int tri;
            node.tri(a, i, A, v0v1, v0v2, tri);
            count_iter<V>(a, 1);
            diag_add<V>(1, 1);
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
        const unsigned long long t_tri1 = diag_clock<V>();
        diag_add<V>(0, t_tri1 - t_tri0);
        const f3 extents = nmax - nmin;
        farSide = getOppositeBoxSide(plan, uv);
        uv = nmin + uv * extents;
        // (derived layout: the neighbour index is READ when the leaf is left — one more L1 hit per leaf step — instead of all six
        // being held in registers through the triangle loop: 28 -> 12 B of scratch at 5 waves per SIMD, 100 -> 80 B at 6;
        // bunny 4K 0.0958 -> 0.0935 ms per frame in flight)
        currOctreeIndex = V == 0 ? node.neighbor(a, farSide) : a.dnodes[currOctreeIndex].nb[farSide];
        const bool stop = exit_is_past_hit(uv - newRay.origin, hit.dist, didHit);
        diag_add<V>(4, (diag_clock<V>() - t_tri1) + (t_desc0 - t_leaf0));
        diag_add<V>(5, 1);
        if (stop) break;
    }
    if (V == 2) {   // diagnostic: longest single walk (leaf steps) and a coarse histogram of walk lengths
        atomicMax(&a.counters[6], (unsigned long long)steps);
        if (steps > 32) atomicAdd(&a.counters[7], 1ull);
    }
    if (!didHit) return false;

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
    const f3 objPoint = newRay.origin + newRay.dir * hit.dist;
    const f3 worldPoint = transformPoint(obj.M, objPoint);
    hit.dist = length(worldPoint - world_origin) / world_dirlen;
    return true;
}

// ---- the same walk with its memory round trips re-ordered (V >= 256: flags in the low bits; results cannot differ: only WHEN
// a record is asked for changes, never what is computed from it) -------------------------------------------------------------
//   1: the exit face of a leaf does not depend on its triangles (getOppositeBoxSide works on the ray and the entry point alone),
//      so it is found BEFORE the triangle loop and the neighbour's index is on its way while the triangles are tested
//   2: ... and once that index is there (after the first triangle), so is the neighbour's record: the next leaf step starts with
//      its node already in registers
//   4: triangle records are asked for one iteration ahead
struct NodeRec { v4f lo, hi; int count; };
RPT_DEV NodeRec load_node_rec(const KernelArgs &a, int i) {
    const v4f *p = reinterpret_cast<const v4f *>(a.dnodes + i);
    NodeRec r;
    r.lo = p[0];
    r.hi = p[1];
    r.count = a.dnodes[i].leafCount;
    return r;
}
struct TriRec { v4f t0, t1; float e2z; int tri; };
RPT_DEV TriRec load_tri_rec(const KernelArgs &a, int k) {
    const v4f *p = reinterpret_cast<const v4f *>(a.dtris + k);
    TriRec r;
    r.t0 = p[0];
    r.t1 = p[1];
    const float2 t2 = *reinterpret_cast<const float2 *>(p + 2);
    r.e2z = t2.x;
    r.tri = __float_as_int(t2.y);
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

template <int F>
RPT_DEV bool octree_core_v2(const KernelArgs &a, const rpt_object &obj, int root, const Ray &newRay, f3 world_origin,
                            float world_dirlen, Hit &hit) {
    int curr = root;
    NodeRec rec = load_node_rec(a, curr);
    f2 d;
    int closeSide, farSide;
    f3 nmin = mk3(rec.lo.x, rec.lo.y, rec.lo.z), nmax = mk3(rec.hi.x, rec.hi.y, rec.hi.z);
    if (!intersect_AABB(nmin, nmax, newRay, d, closeSide, farSide)) return false;
    f3 uv = newRay.origin + newRay.dir * d.x;
    if (d.x < 0) {
        uv = (newRay.origin - nmin) / (nmax - nmin);
        if (__float_as_int(rec.lo.w) != -1) {
            int link = __float_as_int(rec.lo.w);
            while (link != -1) {
                curr = (link & RPT_LINK_CHILD_MASK) + octree_child_step_fast(uv);
                link = a.dnodes[curr].link;
            }
            rec = load_node_rec(a, curr);
        }
        nmin = mk3(rec.lo.x, rec.lo.y, rec.lo.z);
        nmax = mk3(rec.hi.x, rec.hi.y, rec.hi.z);
        if (!intersect_AABB(nmin, nmax, newRay, d, closeSide, farSide)) return false;
        uv = newRay.origin + newRay.dir * d.x;
    }
    const ExitPlan plan = makeExitPlan(normalize(newRay.dir / (nmax - nmin)));
    bool didHit = false;
    int hitTri = 0;
    for (int steps = 1; steps <= RPT_MAX_LEAF_STEPS; steps++) {
        nmin = mk3(rec.lo.x, rec.lo.y, rec.lo.z);
        nmax = mk3(rec.hi.x, rec.hi.y, rec.hi.z);
        uv = (uv - nmin) / (nmax - nmin);
        int link = __float_as_int(rec.lo.w);
        if (link != -1) {
            while (link != -1) {
                const int k = octree_child_step_fast(uv);
                curr = (link & RPT_LINK_CHILD_MASK) + k;
                if ((F & 16) && ((link >> (24 + k)) & 1)) break;          // the link says this child is a leaf: no lookup
                link = (F & 16) ? a.links[curr] : a.dnodes[curr].link;
            }
            rec = load_node_rec(a, curr);
            nmin = mk3(rec.lo.x, rec.lo.y, rec.lo.z);
            nmax = mk3(rec.hi.x, rec.hi.y, rec.hi.z);
        }
        int i = __float_as_int(rec.hi.w);
        const int trisEnd = i + rec.count;
        // the way out, before the triangles
        farSide = getOppositeBoxSide(plan, uv);
        const int next = a.dnodes[curr].nb[farSide];
        NodeRec nrec;
        nrec.lo = nrec.hi = rec.lo;
        nrec.count = 0;
        if (F & 4) {
            if (i < trisEnd) {
                TriRec cur = load_tri_rec(a, i);
                bool fetched = false;
                for (; i < trisEnd; i++) {
                    TriRec nxt = cur;
                    if (i + 1 < trisEnd) nxt = load_tri_rec(a, i + 1);
                    test_tri_rec(cur, newRay, hit, hitTri, didHit);
                    if ((F & 2) && !fetched) { if (next != -1) nrec = load_node_rec(a, next); fetched = true; }
                    cur = nxt;
                }
            } else if (F & 2) {
                if (next != -1) nrec = load_node_rec(a, next);
            }
        } else {
            if (i < trisEnd) {
                test_tri_rec(load_tri_rec(a, i), newRay, hit, hitTri, didHit);
                i++;
            }
            if (F & 2) { if (next != -1) nrec = load_node_rec(a, next); }
            for (; i < trisEnd; i++) test_tri_rec(load_tri_rec(a, i), newRay, hit, hitTri, didHit);
        }
        uv = nmin + uv * (nmax - nmin);
        const bool stop = exit_is_past_hit(uv - newRay.origin, hit.dist, didHit);
        if (stop || next == -1) break;
        curr = next;
        rec = (F & 2) ? nrec : load_node_rec(a, curr);
    }
    if (!didHit) return false;
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
    const f3 objPoint = newRay.origin + newRay.dir * hit.dist;
    const f3 worldPoint = transformPoint(obj.M, objPoint);
    hit.dist = length(worldPoint - world_origin) / world_dirlen;
    return true;
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

// One object against one ray given as a 4-D event + 4-D direction in the object's rest frame
// (the general form: shadow rays, and primary rays of the V = 0 kernel).
// seg_max > 0 (shadow rays): the caller only asks whether the object is hit at a distance below seg_max (sample_light:
// dist < lightDist).  A hit at distance s lies at origin + dir * s (dir not yet normalised: hit.dist is measured in its
// units, opencl_kernel.cl:328,354) ON the object, and a sphere's or a cube's surface lies inside [-1,1]^3: if, for every
// active lane of the wave, the segment origin + dir * [0, seg_max] stays beyond one of that box's six planes (the segment's
// bounding box against the unit box grown by a margin: three multiply-adds and twelve compares; a NaN compares false and
// keeps the object), no lane can get an answer other than "not hit below seg_max", and the normalisation, its three IEEE
// divisions and the intersector are skipped for the whole wave (__ballot).  A mesh gets the same treatment against its root
// box only where the host has found that every triangle its octree lists lies inside that box (DObj.mesh_in_box, below):
// the reference's walk accepts a triangle where the RAY meets its plane, and a second mesh's lists also hold the first
// mesh's triangles (Mesh.cpp:16-19), which can be anywhere — such a mesh is left alone.
// (Measured also: the slab test with v_rcp_f32 as a second stage, and the same for mesh roots as a ray test: no
// further gain on any scene — three quarter-rate reciprocals cost what they save)
template <int V>
RPT_DEV bool intersect_object(const KernelArgs &a, int i, f4 origin4, f4 dir4, Hit &hit, float seg_max = -1.0f) {
    const rpt_object &obj = a.objects[i];
    const f3 origin = transformPoint(obj.InvM, yzw(origin4));
    f3 dir = transformDirection(obj.InvM, yzw(dir4));
    if (V >= 20 && seg_max > 0.0f && obj.type != RPT_MESH) {
        // m: the intersectors' own float error grows with the origin's distance D (in object units) — the sphere's b^2 - c by up
        // to ~1.5e-6 D^2 (so a "hit" can lie that much outside the unit sphere), the cube's slab products by ~4e-7 D
        const float s = seg_max * 1.001f + 1.0e-4f, m = 1.002f + 0.75e-6f * dot(origin, origin);
        const f3 e = origin + dir * s;
        const bool apart = ((origin.x > m) & (e.x > m)) | ((origin.x < -m) & (e.x < -m)) |
                           ((origin.y > m) & (e.y > m)) | ((origin.y < -m) & (e.y < -m)) |
                           ((origin.z > m) & (e.z > m)) | ((origin.z < -m) & (e.z < -m));
        if (__ballot(!apart) == 0ull) return false;
    }
    // The same for a mesh whose root box holds all the triangles its octree lists (not so for a second mesh of a scene, whose
    // lists also carry the first one's triangles, Mesh.cpp:16-19: the host says which): a hit below seg_max is a point of one of
    // those triangles on the segment, so a segment that stays beyond one of the box's planes cannot produce one, and the
    // normalisation, the slab test and the walk are skipped for the whole wave.  The margin covers the slab test's and the
    // triangle test's float error (relative to the box and to the coordinates' size).
    if (V >= 20 && seg_max > 0.0f && obj.type == RPT_MESH && a.dobjs[i].mesh_in_box != 0.0f) {
        const DNode &root = a.dnodes[a.dobjs[i].root];
        const float s = seg_max * 1.001f + 1.0e-4f;
        const f3 e = origin + dir * s;
        const float mx = 0.002f * (root.maxx - root.minx) + 2.0e-6f * (__builtin_fabsf(origin.x) + __builtin_fabsf(e.x)) + 1.0e-6f;
        const float my = 0.002f * (root.maxy - root.miny) + 2.0e-6f * (__builtin_fabsf(origin.y) + __builtin_fabsf(e.y)) + 1.0e-6f;
        const float mz = 0.002f * (root.maxz - root.minz) + 2.0e-6f * (__builtin_fabsf(origin.z) + __builtin_fabsf(e.z)) + 1.0e-6f;
        const bool apart = ((origin.x > root.maxx + mx) & (e.x > root.maxx + mx)) | ((origin.x < root.minx - mx) & (e.x < root.minx - mx)) |
                           ((origin.y > root.maxy + my) & (e.y > root.maxy + my)) | ((origin.y < root.miny - my) & (e.y < root.miny - my)) |
                           ((origin.z > root.maxz + mz) & (e.z > root.maxz + mz)) | ((origin.z < root.minz - mz) & (e.z < root.minz - mz));
        if (__ballot(!apart) == 0ull) return false;
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
        if (V >= 256) return octree_core_v2<(V & 23)>(a, obj, a.dobjs[i].root, newRay, yzw(origin4), length(yzw(dir4)), hit);
        return octree_core<V>(a, obj, V == 0 ? obj.meshIndex : a.dobjs[i].root, newRay, yzw(origin4), length(yzw(dir4)), hit);
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
        if (V >= 256) return octree_core_v2<(V & 23)>(a, obj, pre.root, newRay, cam3, length(d3), hit);
        return octree_core<V>(a, obj, pre.root, newRay, cam3, length(d3), hit);
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
    if (V == 3) {   // diagnostic: stop after the closest hit (timing of the primary walk alone; not a product path)
        color_out = mk3(hit.dist, hit.normal.x + hit.uv.x, hit.normal.y + hit.normal.z + hit.uv.y);
        return true;
    }

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

// ---------------------------------------------------------------------------------------------
// One thread per pixel, wave = 8x8 tile, workgroup = 32x8 strip.
//   V = 0: reads the reference layouts only (general fallback, any valid octree)
//   V = 1: derived DNode/DTri/DObj layouts
//   V = 2, 3, 4: diagnostic builds (loop counters, primary rays only, per-wave timeline; -DRPT_DIAGNOSTICS only)
//   V = 10: per-tile object masks from the prepass kernel (round 1's default, kept for A/B)
//   V = 20: the wave's object mask from per-object image-plane rectangles + __ballot (the default)
template <int V>
RPT_DEV void render_pixel_body(const KernelArgs &a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    unsigned long long t_start = 0;
    if (V == 4) {
        t_start = wall_clock64();
        if ((threadIdx.x & 63) < 8) rpt_diag_lds[threadIdx.x >> 6][threadIdx.x & 63] = 0;
        rpt_diag_lds[threadIdx.x >> 6][6] = clock64();
    }
    int tile_row = (int)blockIdx.y;              // 8-row tiles of this context, natural order
    int strip = (int)blockIdx.x;                 // 32-pixel-wide strip of that row
    if ((V == 23 || (V >= 256 && (V & 8))) && a.first_h > 0) {
        // Workgroups are handed out in the order of their linear index, i.e. row of strips by row of strips.  One frame at a
        // time, what ends the frame is the last of its long waves, so the band of tile rows that holds the meshes goes first
        // (whole rows, in their natural order: neighbours stay neighbours) and the other rows follow in order.
        const int y = (int)blockIdx.y, rh = a.first_h;
        tile_row = y < rh ? a.first_ty + y : (y - rh < a.first_ty ? y - rh : y);
    }
    const int row_in_tile = lane >> 3;
    const int x_coord = strip * 32 + wave * 8 + (lane & 7);
    const int local_row = tile_row * RPT_TILE_ROWS + row_in_tile;
    const int global_tile = (tile_row >> a.run_log2) * a.tile_step + a.first_tile + (tile_row & ((1 << a.run_log2) - 1));
    const int y_coord = global_tile * RPT_TILE_ROWS + row_in_tile;
    // the wave's object mask comes from a __ballot over ALL 64 lanes (lane i answers for object i), so it is formed
    // before the lanes of a partial tile leave
    unsigned long long object_mask = ~0ull;
    if (V >= 20) object_mask = wave_object_mask(a, strip * 32 + wave * 8, global_tile * RPT_TILE_ROWS);
    if (x_coord >= a.width || y_coord >= a.height) return;   // the reference has no guard (UB)

    f3 color;
    f3 mapped = mk3(0.0f, 0.0f, 0.0f);
    bool traced = false;
    uint32_t packed = a.bg_packed;
    if (V == 10) {   // per-tile object mask of the prepass
        const int tile = __builtin_amdgcn_readfirstlane(tile_row * a.mask_tiles_x + (int)blockIdx.x * 4 + wave);
        object_mask = a.tile_masks[tile];
    }
    const bool masked = V == 10 || V >= 20;
    if (!masked || object_mask != 0 || a.object_count > 64) {
        const f3 camdir = createCamRayDir((float)x_coord, (float)y_coord, a.width, a.height, a.aspect);
        if (trace<V>(a, camdir, object_mask, color)) {
            packed = tonemap_pack(a, color, mapped);
            traced = true;
        }
    }

    const size_t id = (size_t)y_coord * a.width + x_coord;
    if (a.out16) {
        uint4 px;
        px.x = __float_as_uint((float)x_coord);
        px.y = __float_as_uint((float)y_coord);
        px.z = packed;
        px.w = 0u;
        store_pixel(a.out16, id, px.x, px.y, px.z, px.w);
    }
    // (the 4-byte plane likewise: measured against the default policy on a rank's share of the frame, bunny 4K 0.0581 -> 0.0566 ms, shadows the same)
    if (a.plane) __builtin_nontemporal_store(packed, a.plane + (size_t)local_row * a.width + x_coord);
    if (a.debug_rgb) {
        if (!traced) mapped = mk3(a.bg_mapped[0], a.bg_mapped[1], a.bg_mapped[2]);      // (read here only: a miss pixel's store needs nothing beyond the first line of the arguments)
        a.debug_rgb[3 * id + 0] = mapped.x;
        a.debug_rgb[3 * id + 1] = mapped.y;
        a.debug_rgb[3 * id + 2] = mapped.z;
    }
    if (V == 4 && a.wave_times) {
        const unsigned long long t_end = wall_clock64();
        const unsigned long long m = __ballot(1);
        if (lane == __ffsll((long long)m) - 1) {
            const size_t w = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave;
            a.wave_times[10 * w] = t_start;
            a.wave_times[10 * w + 1] = t_end;
            rpt_diag_lds[wave][7] = clock64();
            for (int q = 0; q < 8; q++) a.wave_times[10 * w + 2 + q] = rpt_diag_lds[wave][q];
        }
    }
}

#ifndef RPT_RELAXED_FP    /* rpt_relaxed.hip instantiates its own two kernels and nothing else from here on */
// Product kernels (rpt_set_variant): the default, the general fallback and the A/B forms kept for measurement.
__global__ __launch_bounds__(256) void rpt_render_kernel_v0(const KernelArgs a) { render_pixel_body<0>(a); }                                                            // 1
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void rpt_render_kernel_v1_w4(const KernelArgs a) { render_pixel_body<1>(a); }           // 3
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void rpt_render_kernel_v1_masked_w5(const KernelArgs a) { render_pixel_body<10>(a); }   // 26
// V = 20: the wave's object mask from the per-object screen rectangles by lane-parallel test + __ballot (no prepass)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void rpt_render_kernel_ballot_w4(const KernelArgs a) { render_pixel_body<20>(a); }       // 40
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void rpt_render_kernel_ballot_w5(const KernelArgs a) { render_pixel_body<20>(a); }       // 41 = default
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void rpt_render_kernel_ballot_w6(const KernelArgs a) { render_pixel_body<20>(a); }       // 42
// V = 24: the same without the octree walk compiled in, for frames whose Object[] holds no mesh: 61 VGPRs, no scratch, EIGHT waves
// per SIMD (the walk is what needs 96 registers).  arch 1080p 0.0370 -> 0.0301 ms per frame in flight, cubes.txt 4K 0.0898 -> 0.0725.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void rpt_render_kernel_analytic_w8(const KernelArgs a) { render_pixel_body<24>(a); }     // 44
// V = 23: 20 + the strips that hold the meshes handed out first (dispatch order only)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void rpt_render_kernel_ballot_first_w5(const KernelArgs a) { render_pixel_body<23>(a); }   // 43
// experiment arms of round 3 (walk with re-ordered round trips): 257, 259, 261, 263 (+8: mesh band first)
#define RPT_X_KERNEL(N) __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void rpt_render_kernel_x##N(const KernelArgs a) { render_pixel_body<N>(a); }
#define RPT_XW_KERNEL(N, W) __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W))) void rpt_render_kernel_x##N##_w##W(const KernelArgs a) { render_pixel_body<N>(a); }
RPT_XW_KERNEL(257, 6) RPT_XW_KERNEL(257, 4) RPT_XW_KERNEL(263, 4) RPT_XW_KERNEL(259, 4) RPT_X_KERNEL(273) RPT_X_KERNEL(277)
RPT_X_KERNEL(256) RPT_X_KERNEL(257) RPT_X_KERNEL(259) RPT_X_KERNEL(261) RPT_X_KERNEL(263) RPT_X_KERNEL(265) RPT_X_KERNEL(269)
#ifdef RPT_DIAGNOSTICS   /* librpt_hip_diag.so only (make diag): loop counters, primary rays only, per-wave timeline */
__global__ __launch_bounds__(256) void rpt_render_kernel_v1_diag(const KernelArgs a) { render_pixel_body<2>(a); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void rpt_render_kernel_v1_timeline(const KernelArgs a) { render_pixel_body<4>(a); }
__global__ __launch_bounds__(256) void rpt_render_kernel_primary_only(const KernelArgs a) { render_pixel_body<3>(a); }
#endif

// ---------------------------------------------------------------------------------------------
// Tile-mask prepass (one thread per 8x8 tile).  For every object it asks whether ANY primary ray
// of the tile can reach the object's bounding sphere: the tile's rays (object space) lie in a cone
// around the centre ray whose half-angle is taken from the four corner rays (of the tile grown by
// half a pixel) with a 1.5x safety factor; the sphere subtends asin(r/d) around the direction to its
// centre; the object is dropped for the tile only if the two cones are clearly disjoint.  Radii are
// inflated on the host and every doubtful case (origin inside or near the sphere, degenerate or
// non-finite directions, wide tiles) keeps the object.  The arithmetic here is approximate on
// purpose: it only decides which exact tests are skipped, and a skipped test is one the reference
// would have failed for every pixel of the tile.
RPT_DEV f3 cull_dir(const DObj &o, f3 nd) {
    return mk3(o.B[0] * nd.x + o.B[1] * nd.y + o.B[2] * nd.z + o.b[0],
               o.B[3] * nd.x + o.B[4] * nd.y + o.B[5] * nd.z + o.b[1],
               o.B[6] * nd.x + o.B[7] * nd.y + o.B[8] * nd.z + o.b[2]);
}

__global__ __launch_bounds__(256) void rpt_tile_bin_kernel(const KernelArgs a) {
    const int tile = blockIdx.x * 256 + threadIdx.x;
    const bool valid = tile < a.n_tiles;
    unsigned long long mask = 0;
    if (valid) {
        const int tx = tile % a.mask_tiles_x, trow = tile / a.mask_tiles_x;
        const float x0 = (float)(tx * 8);
        const float y0 = (float)(((trow >> a.run_log2) * a.tile_step + a.first_tile + (trow & ((1 << a.run_log2) - 1))) * RPT_TILE_ROWS);
        const float xs[5] = {x0 + 3.5f, x0 - 0.5f, x0 + 7.5f, x0 - 0.5f, x0 + 7.5f};
        const float ys[5] = {y0 + 3.5f, y0 - 0.5f, y0 - 0.5f, y0 + 7.5f, y0 + 7.5f};
        f3 nd[5];
        for (int k = 0; k < 5; k++) {
            const f3 p = mk3((xs[k] / (float)a.width - 0.5f) * a.aspect, ys[k] / (float)a.height - 0.5f, 0.5f);
            nd[k] = p * (1.0f / __builtin_sqrtf(dot(p, p)));
        }
        const int n = a.object_count < 64 ? a.object_count : 64;
        for (int i = 0; i < n; i++) {
            const DObj &o = a.dobjs[i];
            bool keep = true;
            if (o.rb >= 0.0f) {
                f3 u[5];
                float lmin = 3.0e38f, lmax = 0.0f;
                for (int k = 0; k < 5; k++) {
                    const f3 d = cull_dir(o, nd[k]);
                    const float l = __builtin_sqrtf(dot(d, d));
                    lmin = l < lmin ? l : lmin;
                    lmax = l > lmax ? l : lmax;
                    u[k] = d * (1.0f / l);
                }
                // Angles through their SINES, |u x v| (accurate for the tiny angles that strongly anisotropic object
                // scales produce; acos of a cosine near 1 loses them in fp32), and bounded instead of evaluated:
                // for an angle below 0.5 rad, sin <= angle <= 1.05 sin.  The tile's half-angle and the sphere's
                // angular radius are over-estimated, the angle to the sphere's centre is under-estimated.
                float sTile = 0.0f;
                bool tile_ok = true;
                for (int k = 1; k < 5; k++) {
                    const f3 cr = cross(u[0], u[k]);
                    const float sn = __builtin_sqrtf(dot(cr, cr));
                    sTile = sn > sTile ? sn : sTile;
                    tile_ok = tile_ok && dot(u[0], u[k]) > 0.0f;
                }
                const float thTile = 1.05f * sTile;                 // >= the true half-angle while sTile < 0.47
                const f3 to = mk3(o.cbx - o.ox, o.cby - o.oy, o.cbz - o.oz);
                const float dist = __builtin_sqrtf(dot(to, to));
                const bool sane = tile_ok && (sTile < 0.2f) && (lmin > 0.05f * lmax) && (lmax < 1.0e30f) && (dist > 1.05f * o.rb) && (dist < 1.0e30f);
                if (sane) {
                    const f3 ca = cross(u[0], to);
                    const float sinAng = __builtin_sqrtf(dot(ca, ca)) / dist;
                    const float angLow = dot(u[0], to) > 0.0f ? sinAng : 1.0f;   // angle >= its sine; behind: >= pi/2 > 1
                    const float xs_ = o.rb / dist;
                    const float thObj = xs_ < 0.45f ? 1.05f * xs_ : asinf(fminf(xs_, 1.0f));   // asin(x) <= 1.05 x below 0.45; near objects pay for the asin
                    keep = !(angLow > thObj + 1.5f * thTile + 1.0e-4f);     // NaN anywhere -> keep
                }
            }
            if (keep) mask |= 1ull << i;
        }
        if (a.object_count > 64) mask |= 0ull;   // objects >= 64 are never culled (trace() tests them always)
        a.tile_masks[tile] = mask;
    }
}

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

#endif  /* !RPT_RELAXED_FP */

}  // namespace rptd
