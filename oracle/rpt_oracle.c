/*
 * rpt_oracle.c — CPU restatement of the reference's per-pixel render path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP render
 * path and the "cpu_baseline" leg of bench.py.  Nothing in the product
 * (relativitypathtracer_amd/, include/) imports, links or calls it.
 *
 * PARITY UNPINNED (bit level): the reference ships no tests, golden vectors or
 * fixtures for this path, and it cannot be built in this image without writing
 * stand-ins (its kernel is OpenCL C that the ROCm clang rejects for amdgcn —
 * opencl_kernel.cl:213 passes a __global pointer to a __private parameter — and
 * an x86 build would need an OpenCL built-in library the image lacks; its host
 * sources need <windows.h>, GLUT and GLEW).  Which float rounding the author's
 * GPU used is therefore unknown.  What the restatement below IS pinned by
 * (tests/test_oracle.py) is the reference's own output:
 * (a) its window grabs, reproduced at <= 1 LSB: the static scenes (arch1.png every
 *     pixel, cube1.png) and the MOVING-camera grabs arch2.png (0.95c towards the
 *     arch, light delay and shadows: all but 4 of 3.5 M pixels), cube2.png /
 *     cube3.png (0.9c, without / with light propagation) at the camera states
 *     recovered by tests/golden/fit_reference_camera.py;
 * (b) the per-ray work counts the survey recorded from the reference
 *     (SURVEY.md §8a).
 *
 * Every function follows one function of /root/reference/opencl_kernel.cl and
 * cites it.  Arithmetic is scalar IEEE-754 binary32, evaluated in exactly the
 * source order of the reference expression, no FMA contraction (build with
 * -ffp-contract=off, baseline x86-64), correctly rounded / and sqrt.
 *
 * OpenCL built-in semantics used (OpenCL C 1.2 §6.12, in their plainest form):
 *   dot        a.x*b.x + a.y*b.y + a.z*b.z (+ a.w*b.w), summed left to right
 *   cross      (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x)
 *   length     sqrt(dot(v,v));   normalize  v / length(v) (three divides)
 *   min(x,y)   y < x ? y : x;    max(x,y)   x < y ? y : x     (§6.12.4)
 *   sign       +1, -1, +-0 preserved, 0 for NaN
 *   fmod, floor, round (half away from zero), fabs: exact C99 equivalents
 *   asin, atan2: implementation-defined in OpenCL (<= 4 ulp); fixed here as the fdlibm single-precision algorithms
 *                (oracle_asinf / oracle_atan2f below; only reachable through textured spheres)
 *
 * Reference undefined behaviour that is neutralised here — identically in the
 * HIP kernel — without touching any defined result:
 *   - no work-item bounds guard        -> rows/pixels outside the image are not rendered
 *   - NaN/Inf child index in the octree descent -> clamped to 0..7 (NaN -> 0)
 *   - unbounded leaf walk              -> capped at RPT_MAX_LEAF_STEPS
 *   - float->int / float->uchar of NaN or out-of-range values -> saturating, NaN -> 0
 *   - texture taps whose byte address falls outside the texture pool -> address clamped
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rpt_layout.h"
#include "rpt_oracle.h"

#define EPSILON 0.0000001f          /* opencl_kernel.cl:6 */
#define RPT_MAX_LEAF_STEPS 4096

typedef struct { float x, y; } f2;
typedef struct { float x, y, z; } f3;
typedef struct { float x, y, z, w; } f4;

typedef struct { f3 origin, dir; } Ray;         /* opencl_kernel.cl:9-12  */
typedef struct { f4 origin, dir; } Ray4D;       /* opencl_kernel.cl:14-17 */
typedef struct {                                /* opencl_kernel.cl:38-44 */
    float dist;
    f3 normal;
    f2 uv;
    f3 color;
    int object;
} Hit;

typedef struct {
    const rpt_object *objects;
    int object_count;
    const rpt_float3 *vertices;
    const rpt_float3 *normals;
    const rpt_float2 *uvs;
    const uint32_t *triangles;
    const rpt_octree *octrees;
    const int32_t *octreeTris;
    const uint8_t *textures;
    int64_t texture_bytes;
    int interval;
} Scene;

static __thread rpt_oracle_stats *tls_stats;
#define STAT(field) do { if (tls_stats) tls_stats->field++; } while (0)
/* statistics only: when was triangle t last tested — in which walk (serial number) and in which leaf visit of that walk */
static __thread uint32_t *tls_seen_walk, *tls_seen_leaf;
static __thread size_t tls_seen_cap;
static __thread uint32_t tls_walk_serial;
static void stat_tri_test(int tri, uint32_t leaf_serial) {
    if (!tls_stats || tri < 0) return;
    if ((size_t)tri >= tls_seen_cap) {
        size_t cap = tls_seen_cap ? tls_seen_cap : 4096;
        while (cap <= (size_t)tri) cap *= 2;
        uint32_t *a = (uint32_t *)realloc(tls_seen_walk, cap * sizeof *a), *b = a ? (uint32_t *)realloc(tls_seen_leaf, cap * sizeof *b) : NULL;
        if (!a || !b) { if (a) tls_seen_walk = a; return; }
        memset(a + tls_seen_cap, 0, (cap - tls_seen_cap) * sizeof *a);
        memset(b + tls_seen_cap, 0, (cap - tls_seen_cap) * sizeof *b);
        tls_seen_walk = a; tls_seen_leaf = b; tls_seen_cap = cap;
    }
    if (tls_seen_walk[tri] != tls_walk_serial) tls_stats->distinct_tri_tests++;
    else if (tls_seen_leaf[tri] + 1 == leaf_serial) tls_stats->repeats_of_previous_leaf++;
    tls_seen_walk[tri] = tls_walk_serial;
    tls_seen_leaf[tri] = leaf_serial;
}

/* ---- vector helpers (built-in semantics listed in the header) ---- */
static inline f3 F3(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline f4 F4(float x, float y, float z, float w) { f4 r = { x, y, z, w }; return r; }
static inline f3 xyz(rpt_float4 v) { return F3(v.x, v.y, v.z); }
static inline f4 ld4(rpt_float4 v) { return F4(v.x, v.y, v.z, v.w); }
static inline f3 yzw(f4 v) { return F3(v.y, v.z, v.w); }
static inline f3 add3(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul3(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 div3(f3 a, f3 b) { return F3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline f3 muls3(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
static inline f3 divs3(f3 a, float s) { return F3(a.x / s, a.y / s, a.z / s); }
static inline f3 neg3(f3 a) { return F3(-a.x, -a.y, -a.z); }
static inline f4 add4(f4 a, f4 b) { return F4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline f4 muls4(f4 a, float s) { return F4(a.x * s, a.y * s, a.z * s, a.w * s); }
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float dot4(f4 a, f4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
static inline f3 cross3(f3 a, f3 b) {
    return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float length3(f3 v) { return sqrtf(dot3(v, v)); }
static inline f3 normalize3(f3 v) { float l = length3(v); return F3(v.x / l, v.y / l, v.z / l); }
static inline float cl_min(float x, float y) { return y < x ? y : x; }
static inline float cl_max(float x, float y) { return x < y ? y : x; }
static inline int imin(int x, int y) { return y < x ? y : x; }
static inline int iclamp(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline float cl_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : (x == 0.0f ? x : 0.0f)); }
/* saturating float -> int, NaN -> 0 (neutralises C UB; equals a C cast for in-range values) */
static inline int f2i_sat(float f) {
    if (!(f == f)) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

/* opencl_kernel.cl:55-73 */
static Ray createCamRay(const float x_coord, const float y_coord, const int width, const int height) {
    float fx = (float)x_coord / (float)width;
    float fy = (float)y_coord / (float)height;
    float aspect_ratio = (float)(width) / (float)(height);
    float fx2 = (fx - 0.5f) * aspect_ratio;
    float fy2 = fy - 0.5f;
    f3 pixel_pos = F3(fx2, fy2, 0.5f);
    Ray ray;
    ray.origin = F3(0, 0, 0);
    ray.dir = normalize3(pixel_pos);
    return ray;
}

/* opencl_kernel.cl:75-82 */
static f3 transformPoint(const rpt_float4 M[4], const f3 v) {
    f4 V = F4(v.x, v.y, v.z, 1.0f);
    return F3(dot4(ld4(M[0]), V), dot4(ld4(M[1]), V), dot4(ld4(M[2]), V));
}

/* opencl_kernel.cl:84-91 */
static f4 transformPoint4D(const rpt_float4 M[4], const f4 v) {
    return F4(dot4(ld4(M[0]), v), dot4(ld4(M[1]), v), dot4(ld4(M[2]), v), dot4(ld4(M[3]), v));
}

/* opencl_kernel.cl:93-99 */
static f3 transformDirection(const rpt_float4 M[4], const f3 v) {
    return F3(dot3(xyz(M[0]), v), dot3(xyz(M[1]), v), dot3(xyz(M[2]), v));
}

/* opencl_kernel.cl:102-104 */
static f3 applyTranspose(const rpt_float4 M[4], const f3 v) {
    return add3(add3(muls3(xyz(M[0]), v.x), muls3(xyz(M[1]), v.y)), muls3(xyz(M[2]), v.z));
}

/* opencl_kernel.cl:106-126 */
static int intersect_triangle(const f3 A, const f3 B, const f3 C, const Ray *ray, float *dist, f2 *uv) {
    f3 v0v1 = sub3(B, A);
    f3 v0v2 = sub3(C, A);
    f3 pvec = cross3(ray->dir, v0v2);
    float det = dot3(v0v1, pvec);
    if (det < EPSILON && -EPSILON < det) return 0;

    float invDet = 1 / det;

    f3 tvec = sub3(ray->origin, A);
    uv->x = dot3(tvec, pvec) * invDet;
    if (uv->x < 0 || uv->x > 1) return 0;

    f3 qvec = cross3(tvec, v0v1);
    uv->y = dot3(ray->dir, qvec) * invDet;
    if (uv->y < 0 || uv->x + uv->y > 1) return 0;

    *dist = dot3(v0v2, qvec) * invDet;
    return 1;
}

/* opencl_kernel.cl:128-170 */
static int intersect_AABB(const f3 bounds[2], const Ray *ray, f2 *d, int *closeSide, int *farSide) {
    f3 origin = ray->origin;
    f3 dir = ray->dir;
    f3 inv_dir = F3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    int sign[3] = { inv_dir.x < 0 ? 1 : 0, inv_dir.y < 0 ? 1 : 0, inv_dir.z < 0 ? 1 : 0 };
    d->x = (bounds[sign[0]].x - origin.x) * inv_dir.x;
    d->y = (bounds[1 - sign[0]].x - origin.x) * inv_dir.x;
    *closeSide = 2 + sign[0];
    *farSide = 3 - sign[0];
    float tymin = (bounds[sign[1]].y - origin.y) * inv_dir.y;
    float tymax = (bounds[1 - sign[1]].y - origin.y) * inv_dir.y;
    if ((d->x > tymax) || (tymin > d->y)) {
        return 0;
    }
    if (tymin > d->x) {
        d->x = tymin;
        *closeSide = 4 + sign[1];
    }
    if (tymax < d->y) {
        d->y = tymax;
        *farSide = 5 - sign[1];
    }
    float tzmin = (bounds[sign[2]].z - origin.z) * inv_dir.z;
    float tzmax = (bounds[1 - sign[2]].z - origin.z) * inv_dir.z;
    if ((d->x > tzmax) || (tzmin > d->y)) {
        return 0;
    }
    if (tzmin > d->x) {
        d->x = tzmin;
        *closeSide = sign[2];
    }
    if (tzmax < d->y) {
        d->y = tzmax;
        *farSide = 1 - sign[2];
    }
    return d->y > 0;
}

/* opencl_kernel.cl:172-198 (closeSide is dead in the reference too) */
static int getOppositeBoxSide(const f3 scaledDir, f3 *uv) {
    f3 inv_dir = F3(1.0f / scaledDir.x, 1.0f / scaledDir.y, 1.0f / scaledDir.z);
    int sign[3] = { inv_dir.x < 0, inv_dir.y < 0, inv_dir.z < 0 };
    float dx = ((float)(1 - sign[0]) - uv->x) * inv_dir.x;
    float dy = ((float)(1 - sign[1]) - uv->y) * inv_dir.y;
    float dz = ((float)(1 - sign[2]) - uv->z) * inv_dir.z;
    if (dx < dy) {
        if (dx < dz) {
            *uv = add3(*uv, muls3(scaledDir, dx));
            return 3 - sign[0];
        } else {
            *uv = add3(*uv, muls3(scaledDir, dz));
            return 1 - sign[2];
        }
    } else {
        if (dy < dz) {
            *uv = add3(*uv, muls3(scaledDir, dy));
            return 5 - sign[1];
        } else {
            *uv = add3(*uv, muls3(scaledDir, dz));
            return 1 - sign[2];
        }
    }
}

/* child selection of opencl_kernel.cl:237-238 / 257-258 */
static inline int octree_child_step(f3 *uv) {
    float fidx = roundf(uv->z) + 2 * roundf(uv->y) + 4 * roundf(uv->x);
    int childIndex = !(fidx >= 0.0f) ? 0 : (fidx > 7.0f ? 7 : (int)fidx);   /* UB guard: NaN/out of range */
    uv->x = 2.0f * fmodf(cl_min(uv->x, 1.0f - EPSILON), 0.5f);
    uv->y = 2.0f * fmodf(cl_min(uv->y, 1.0f - EPSILON), 0.5f);
    uv->z = 2.0f * fmodf(cl_min(uv->z, 1.0f - EPSILON), 0.5f);
    return childIndex;
}

/* opencl_kernel.cl:206-306: the walk from the object-space ray on; the hit is re-measured from world_origin in units of
 * world_dirlen (:303-305).  Split off :200-205 so that rpt_oracle_octree_rays can feed it object-space rays directly. */
static int intersect_octree_core(const Scene *s, const int index, Ray newRay, f3 world_origin, float world_dirlen, Hit *hit) {
    const rpt_object *obj = &s->objects[index];
    const rpt_octree *octrees = s->octrees;
    int currOctreeIndex = obj->meshIndex;
    f2 d;
    int closeSide, farSide;
    f3 bounds[2] = { xyz(octrees[currOctreeIndex].min), xyz(octrees[currOctreeIndex].max) };
    int didHit = 0;
    int hitTri = 0;
    if (!intersect_AABB(bounds, &newRay, &d, &closeSide, &farSide)) {
        return 0;
    }
    STAT(root_aabb_hits);
    if (tls_stats && ++tls_walk_serial == 0) {      /* (serial 0 = "never": on wrap-around forget everything) */
        memset(tls_seen_walk, 0, tls_seen_cap * sizeof *tls_seen_walk);
        tls_walk_serial = 1;
    }
    f3 uv = add3(newRay.origin, muls3(newRay.dir, d.x));

    if (d.x < 0) {
        STAT(inside_starts);
        rpt_octree curr = octrees[currOctreeIndex];
        uv = div3(sub3(newRay.origin, xyz(curr.min)), sub3(xyz(curr.max), xyz(curr.min)));
        while (curr.children[0] != -1) {
            int childIndex = octree_child_step(&uv);
            currOctreeIndex = curr.children[childIndex];
            curr = octrees[currOctreeIndex];
            STAT(inside_descent_steps);
        }
        bounds[0] = xyz(curr.min);
        bounds[1] = xyz(curr.max);
        if (!intersect_AABB(bounds, &newRay, &d, &closeSide, &farSide)) {
            return 0;
        }
        uv = add3(newRay.origin, muls3(newRay.dir, d.x));
    }

    f3 scaledDir = div3(newRay.dir, sub3(xyz(octrees[currOctreeIndex].max), xyz(octrees[currOctreeIndex].min)));
    scaledDir = normalize3(scaledDir);
    int steps = 0;
    while (currOctreeIndex != -1) {
        if (++steps > RPT_MAX_LEAF_STEPS) break;                             /* UB guard */
        rpt_octree curr = octrees[currOctreeIndex];
        f3 extents = sub3(xyz(curr.max), xyz(curr.min));
        uv = div3(sub3(uv, xyz(curr.min)), extents);
        while (curr.children[0] != -1) {
            int childIndex = octree_child_step(&uv);
            currOctreeIndex = curr.children[childIndex];
            curr = octrees[currOctreeIndex];
            STAT(descent_steps);
        }
        STAT(leaf_visits);
        for (int i = curr.trisIndex; i < curr.trisIndex + curr.trisCount; i++) {
            int tri = s->octreeTris[i];
            f3 A = xyz(s->vertices[s->triangles[9 * tri + 3 * 0]]);
            f3 B = xyz(s->vertices[s->triangles[9 * tri + 3 * 1]]);
            f3 C = xyz(s->vertices[s->triangles[9 * tri + 3 * 2]]);
            float dist;
            f2 triUV;
            STAT(tri_tests);
            stat_tri_test(tri, (uint32_t)steps);
            if (intersect_triangle(A, B, C, &newRay, &dist, &triUV)) {
                if (0 <= dist && dist < hit->dist) {
                    hitTri = tri;
                    hit->dist = dist;
                    hit->uv = triUV;
                    didHit = 1;
                }
            }
        }
        extents = sub3(xyz(curr.max), xyz(curr.min));
        farSide = getOppositeBoxSide(scaledDir, &uv);
        closeSide = farSide - 2 * (farSide % 2) + 1;
        (void)closeSide;
        uv = add3(xyz(curr.min), mul3(uv, extents));
        currOctreeIndex = curr.neighbors[farSide];
        if (length3(sub3(uv, newRay.origin)) > hit->dist) {
            break;
        }
    }
    if (didHit) {
        float u = hit->uv.x;
        float v = hit->uv.y;

        f3 normA = xyz(s->normals[s->triangles[2 + 9 * hitTri + 3 * 0]]);
        f3 normB = xyz(s->normals[s->triangles[2 + 9 * hitTri + 3 * 1]]);
        f3 normC = xyz(s->normals[s->triangles[2 + 9 * hitTri + 3 * 2]]);
        float w = 1.0f - u - v;
        f3 nrm = add3(add3(muls3(normA, w), muls3(normB, u)), muls3(normC, v));
        hit->normal = normalize3(applyTranspose(obj->InvM, nrm));

        rpt_float2 uvA = s->uvs[s->triangles[1 + 9 * hitTri + 3 * 0]];
        rpt_float2 uvB = s->uvs[s->triangles[1 + 9 * hitTri + 3 * 1]];
        rpt_float2 uvC = s->uvs[s->triangles[1 + 9 * hitTri + 3 * 2]];
        hit->uv.x = w * uvA.x + u * uvB.x + v * uvC.x;
        hit->uv.y = w * uvA.y + u * uvB.y + v * uvC.y;

        f3 objPoint = add3(newRay.origin, muls3(newRay.dir, hit->dist));
        f3 worldPoint = transformPoint(obj->M, objPoint);
        hit->dist = length3(sub3(worldPoint, world_origin)) / world_dirlen;
        return 1;
    }
    return 0;
}

/* opencl_kernel.cl:200-308 */
static int intersect_octree(const Scene *s, const int index, const Ray4D *ray, Hit *hit) {
    const rpt_object *obj = &s->objects[index];
    STAT(octree_calls);
    Ray newRay;
    newRay.origin = transformPoint(obj->InvM, yzw(ray->origin));
    newRay.dir = transformDirection(obj->InvM, yzw(ray->dir));
    float scale = length3(newRay.dir);
    newRay.dir = divs3(newRay.dir, scale);
    return intersect_octree_core(s, index, newRay, yzw(ray->origin), length3(yzw(ray->dir)), hit);
}

/* opencl_kernel.cl:310 */
static float max3(f3 v) { return cl_max(cl_max(v.x, v.y), v.z); }

/* opencl_kernel.cl:312-333 */
static int intersect_cube(const Scene *s, int index, const Ray4D *ray, Hit *hit) {
    const rpt_object *obj = &s->objects[index];
    STAT(cube_tests);
    f3 origin = transformPoint(obj->InvM, yzw(ray->origin));
    f3 dir = transformDirection(obj->InvM, yzw(ray->dir));
    float scale = length3(dir);
    dir = divs3(dir, scale);
    float winding = max3(F3(fabsf(origin.x), fabsf(origin.y), fabsf(origin.z))) < 1.0f ? -1.0f : 1.0f;
    f3 sgn = F3(-cl_sign(dir.x), -cl_sign(dir.y), -cl_sign(dir.z));
    f3 d = div3(sub3(muls3(sgn, winding), origin), dir);
#define TEST(U, V, W) ((d.U >= 0.0f) && (fabsf(origin.V + dir.V * d.U) < 1.0f) && (fabsf(origin.W + dir.W * d.U) < 1.0f))
    if (TEST(x, y, z)) sgn = F3(sgn.x, 0, 0);
    else if (TEST(y, z, x)) sgn = F3(0, sgn.y, 0);
    else sgn = F3(0, 0, TEST(z, x, y) ? sgn.z : 0);
#undef TEST
    float dist = (sgn.x != 0) ? d.x : ((sgn.y != 0) ? d.y : d.z);
    f3 objPt = add3(origin, muls3(dir, dist));
    hit->dist = dist / scale;
    hit->normal = normalize3(applyTranspose(obj->InvM, sgn));
    if (sgn.x != 0) { hit->uv.x = (objPt.y + 1) / 2; hit->uv.y = (objPt.z + 1) / 2; }
    else if (sgn.y != 0) { hit->uv.x = (objPt.x + 1) / 2; hit->uv.y = (objPt.z + 1) / 2; }
    else { hit->uv.x = (objPt.x + 1) / 2; hit->uv.y = (objPt.y + 1) / 2; }
    return (sgn.x != 0) || (sgn.y != 0) || (sgn.z != 0);
}

/* opencl_kernel.cl:335-359 */
/* asin / atan2 of opencl_kernel.cl:356-357.  OpenCL leaves their last bits to the implementation (<= 4 ulp): the
 * oracle fixes them as the fdlibm single-precision algorithms below (argument reduction + minimax polynomial, every
 * step an IEEE +,-,*,/ or sqrt in the stated precision, no contraction), and the HIP kernel restates the same
 * steps, so the two agree bit for bit where the platforms' libm builds would not. */
static inline int32_t f2bits(float x) { int32_t i; memcpy(&i, &x, 4); return i; }

static float oracle_asinf(float x) {
    const float pS0 = 1.6666586697e-01f, pS1 = -4.2743422091e-02f, pS2 = -8.6563630030e-03f, qS1 = -7.0662963390e-01f;
    const double pio2 = 1.570796326794896558e+00;
    const int32_t hx = f2bits(x), ix = hx & 0x7fffffff;
    if (ix >= 0x3f800000) {
        if (ix == 0x3f800000) return (float)(x * pio2);
        return (x - x) / (x - x);
    }
    if (ix < 0x3f000000) {
        if (ix < 0x39800000) return x;
        float t = x * x;
        float p = t * (pS0 + t * (pS1 + t * pS2));
        float q = 1.0f + t * qS1;
        float w = p / q;
        return x + x * w;
    }
    float w0 = 1.0f - fabsf(x);
    float t = w0 * 0.5f;
    float p = t * (pS0 + t * (pS1 + t * pS2));
    float q = 1.0f + t * qS1;
    double s = sqrt((double)t);
    float w = p / q;
    float r = (float)(pio2 - 2.0 * (s + s * (double)w));
    return hx > 0 ? r : -r;
}

static float oracle_atanf(float x) {
    static const float atanhi[4] = { 4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f };
    static const float atanlo[4] = { 5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f };
    static const float aT[5] = { 3.3333328366e-01f, -1.9999158382e-01f, 1.4253635705e-01f, -1.0648017377e-01f, 6.1687607318e-02f };
    const int32_t hx = f2bits(x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c800000) {
        if (ix > 0x7f800000) return x + x;
        float r = atanhi[3] + atanlo[3];
        return hx > 0 ? r : -r;
    }
    if (ix < 0x3ee00000) {
        if (ix < 0x39800000) return x;
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    float z = x * x;
    float w = z * z;
    float s1 = z * (aT[0] + w * (aT[2] + w * aT[4]));
    float s2 = w * (aT[1] + w * aT[3]);
    if (id < 0) return x - x * (s1 + s2);
    float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return hx < 0 ? -r : r;
}

static float oracle_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = f2bits(x), hy = f2bits(y), ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return oracle_atanf(y);
    int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        switch (m) { case 0: case 1: return y; case 2: return pi + tiny; default: return -pi - tiny; }
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) { case 0: return pi_o_4 + tiny; case 1: return -pi_o_4 - tiny; case 2: return 3.0f * pi_o_4 + tiny; default: return -3.0f * pi_o_4 - tiny; }
        }
        switch (m) { case 0: return 0.0f; case 1: return -0.0f; case 2: return pi + tiny; default: return -pi - tiny; }
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 26) { z = pi_o_2 + 0.5f * pi_lo; m &= 1; }
    else if (k < -26 && hx < 0) z = 0.0f;
    else z = oracle_atanf(fabsf(y / x));
    switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

static int intersect_sphere(const Scene *s, const int index, const Ray4D *ray, Hit *hit) {
    const rpt_object *obj = &s->objects[index];
    STAT(sphere_tests);
    f3 rayToSphere = neg3(transformPoint(obj->InvM, yzw(ray->origin)));
    f3 dir = transformDirection(obj->InvM, yzw(ray->dir));
    float scale = length3(dir);
    dir = divs3(dir, scale);
    float b = dot3(rayToSphere, dir);
    float c = dot3(rayToSphere, rayToSphere) - 1.0f;
    float disc = b * b - c;
    if (disc < 0.0f) return 0;
    else disc = sqrtf(disc);
    float dist;
    if ((b - disc) > EPSILON) {
        dist = b - disc;
    } else if ((b + disc) > EPSILON) {
        dist = b + disc;
    } else {
        return 0;
    }
    f3 objPt = add3(neg3(rayToSphere), muls3(dir, dist));
    hit->dist = dist / scale;
    hit->normal = normalize3(applyTranspose(obj->InvM, objPt));
    /* M_PI is a double constant in OpenCL C: the divide and add are done in double, rounded once */
    hit->uv.x = (float)(0.5f + oracle_atan2f(objPt.z, objPt.x) / (2 * M_PI));
    hit->uv.y = (float)(oracle_asinf(objPt.y) / M_PI + 0.5f);
    return 1;
}

static inline float texel(const Scene *s, int64_t addr) {
    /* UB guard: a tap outside the texture pool reads the nearest in-pool byte */
    if (addr < 0) addr = 0;
    if (addr >= s->texture_bytes) addr = s->texture_bytes - 1;
    return s->textures[addr] / 255.0f;
}
static inline f3 texel3(const Scene *s, int offset, int width, int x, int y) {
    int64_t base = (int64_t)offset + 3 * ((int64_t)width * y + x);
    return F3(texel(s, base + 0), texel(s, base + 1), texel(s, base + 2));
}

/* opencl_kernel.cl:361-486 */
static int intersect_scene(const Scene *s, const Ray *ray, Hit *hit) {
    float inf = 1e20f;
    hit->dist = inf;
    int didHit = 0;
    f4 event = F4(0, 0, 0, 0);
    const int interval = s->interval;

    for (int i = 0; i < s->object_count; i++) {
        Hit newHit;
        newHit.dist = inf;
        Ray4D newRay;
        f4 newEvent0 = ld4(s->objects[i].stationaryCam);
        f3 nd = normalize3(ray->dir);
        f4 lightDir = F4((float)interval, nd.x, nd.y, nd.z);
        lightDir = transformPoint4D(s->objects[i].Lorentz, lightDir);
        newRay.origin = newEvent0;
        newRay.dir = lightDir;

        int got = 0;
        switch (s->objects[i].type) {
        case RPT_SPHERE: got = intersect_sphere(s, i, &newRay, &newHit); break;
        case RPT_CUBE:   got = intersect_cube(s, i, &newRay, &newHit); break;
        case RPT_MESH:   got = intersect_octree(s, i, &newRay, &newHit); break;
        }
        if (got) {
            if (newHit.dist < hit->dist) {
                event = add4(newEvent0, muls4(lightDir, newHit.dist));
                *hit = newHit;
                hit->object = i;
                didHit = 1;
            }
        }
    }
    if (didHit) {
        const rpt_object *ho = &s->objects[hit->object];
        if (ho->textureIndex != -1) {
            int width = ho->textureWidth;
            int height = ho->textureHeight;
            float u = width * hit->uv.x;
            float v = height * (1.0f - hit->uv.y);
            int x = imin(f2i_sat(floorf(u)), width - 1);
            int y = imin(f2i_sat(floorf(v)), height - 1);
            float u_ratio = u - x;
            float v_ratio = v - y;
            float u_opp = 1 - u_ratio;
            float v_opp = 1 - v_ratio;

            int offset = ho->textureIndex;
            f3 result = muls3(texel3(s, offset, width, x, y), u_opp);
            x = iclamp(x + 1, 0, width - 1);
            result = add3(result, muls3(texel3(s, offset, width, x, y), u_ratio));
            result = muls3(result, v_opp);
            y = iclamp(y + 1, 0, height - 1);
            f3 result2 = muls3(texel3(s, offset, width, x, y), u_ratio);
            x = iclamp(x - 1, 0, width - 1);
            result2 = add3(result2, muls3(texel3(s, offset, width, x, y), u_opp));
            result2 = muls3(result2, v_ratio);

            hit->color = add3(result, result2);
        } else {
            hit->color = xyz(ho->color);
        }
        /* Periodic flash, opencl_kernel.cl:476-482 */
        if (ho->flashPeriod > 0) {
            float period = ho->flashPeriod;
            float duration = ho->flashDuration;
            if (event.x - period * floorf(event.x / period) < duration) {
                hit->color = muls3(hit->color, 2);
            }
        }
        return 1;
    }
    return 0;
}

/* opencl_kernel.cl:488-545 */
static int sample_light(const Scene *s, const Ray4D *ray, float lightDist, const int lightIndex) {
    float inf = 1e20f;
    const int interval = s->interval;
    STAT(shadow_rays);
    for (int i = 0; i < s->object_count; i++) {
        if (i != lightIndex) {
            Hit newHit;
            newHit.dist = inf;
            Ray4D newRay;
            f4 newEvent0 = transformPoint4D(s->objects[i].Lorentz, ray->origin);
            f3 nd = normalize3(yzw(ray->dir));
            f4 lightDir = F4((float)interval, nd.x, nd.y, nd.z);
            lightDir = transformPoint4D(s->objects[i].Lorentz, lightDir);
            newRay.origin = newEvent0;
            newRay.dir = lightDir;

            int got = 0;
            switch (s->objects[i].type) {
            case RPT_SPHERE: got = intersect_sphere(s, i, &newRay, &newHit); break;
            case RPT_CUBE:   got = intersect_cube(s, i, &newRay, &newHit); break;
            case RPT_MESH:   got = intersect_octree(s, i, &newRay, &newHit); break;
            }
            if (got) {
                if (newHit.dist < lightDist) {
                    return i;
                }
            }
        }
    }
    return -1;
}

/* opencl_kernel.cl:548-604 */
static f3 trace(const Scene *s, const float ambient, const Ray *camray) {
    Hit hit;
    const int interval = s->interval;
    if (!intersect_scene(s, camray, &hit))
        return F3(0.15f, 0.15f, 0.25f);
    STAT(pixels_hit);

    const rpt_object *ho = &s->objects[hit.object];
    f3 color = muls3(hit.color, (interval != 0 ? ambient : 1.0f));

    if (ho->light) {
        color = add3(color, hit.color);
    }
    if (interval != 0) {
        for (int i = 0; i < s->object_count; i++) {
            if (i != hit.object && s->objects[i].light) {
                const rpt_object *lo = &s->objects[i];
                f4 cameraPos_ObjFrame = ld4(ho->stationaryCam);
                f3 nd = normalize3(camray->dir);
                f4 rayDir = F4((float)interval, nd.x, nd.y, nd.z);
                f4 rayDir_ObjFrame = transformPoint4D(ho->Lorentz, rayDir);
                f4 hitPos_ObjFrame = add4(cameraPos_ObjFrame, muls4(rayDir_ObjFrame, hit.dist));
                hitPos_ObjFrame = add4(hitPos_ObjFrame,
                                       F4(0, hit.normal.x * 0.001f, hit.normal.y * 0.001f, hit.normal.z * 0.001f));
                f4 hitPos = transformPoint4D(ho->InvLorentz, hitPos_ObjFrame);
                f4 hitPos_LightFrame = transformPoint4D(lo->Lorentz, hitPos);
                f3 hitPos3_LightFrame = yzw(hitPos_LightFrame);
                f3 lightPos3_LightFrame = F3(lo->M[0].w, lo->M[1].w, lo->M[2].w);
                f3 lightDir3_LightFrame = sub3(lightPos3_LightFrame, hitPos3_LightFrame);
                f4 lightDir_LightFrame = F4(interval * length3(lightDir3_LightFrame),
                                            lightDir3_LightFrame.x, lightDir3_LightFrame.y, lightDir3_LightFrame.z);
                f4 lightDir = transformPoint4D(lo->InvLorentz, lightDir_LightFrame);
                f4 lightDir_ObjFrame = transformPoint4D(ho->Lorentz, lightDir);
                f3 lightDir3_ObjFrame = yzw(lightDir_ObjFrame);
                f3 unitLightDir3 = normalize3(lightDir3_ObjFrame);

                if (dot3(hit.normal, unitLightDir3) > 0) {
                    Ray4D newRay;
                    f3 ld = normalize3(yzw(lightDir));
                    newRay.dir = F4((float)interval, ld.x, ld.y, ld.z);
                    newRay.origin = hitPos;
                    int shadowIndex = sample_light(s, &newRay, length3(yzw(lightDir)), i);
                    if (shadowIndex == -1) {
                        float k = dot3(hit.normal, unitLightDir3) /
                                  (1.0f + 0.1f * length3(lightDir3_ObjFrame) +
                                   0.01f * dot3(lightDir3_ObjFrame, lightDir3_ObjFrame));
                        color = add3(color, mul3(muls3(hit.color, k), xyz(lo->color)));
                    }
                }
            }
        }
    }
    return color;
}

/* opencl_kernel.cl:607-616 */
static f3 hable(const f3 x) {
    float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    f3 r;
    r.x = ((x.x * (A * x.x + C * B) + D * E) / (x.x * (A * x.x + B) + D * F)) - E / F;
    r.y = ((x.y * (A * x.y + C * B) + D * E) / (x.y * (A * x.y + B) + D * F)) - E / F;
    r.z = ((x.z * (A * x.z + C * B) + D * E) / (x.z * (A * x.z + B) + D * F)) - E / F;
    return r;
}

static inline uint8_t to_u8(float c) {  /* (unsigned char)(c * 255), saturating, NaN -> 0 */
    float t = c * 255;
    if (!(t == t)) return 0;
    if (t <= 0.0f) return 0;
    if (t >= 255.0f) return 255;
    return (uint8_t)(int)t;
}

/* opencl_kernel.cl:620-660, one work item */
static void render_pixel(const Scene *s, const f3 white_point, const float ambient, const int width,
                         const int height, const int msaa, unsigned int work_item_id, rpt_pixel *output, float *rgb_out) {
    unsigned int x_coord = work_item_id % width;
    unsigned int y_coord = work_item_id / width;

    f3 finalcolor;
    if (msaa <= 1) {
        /* MSAASAMPLES = 1 (opencl_kernel.cl:7), the reference as shipped: x + 0/1 = x, 0 + c = c, c / 1 = c — the loop of
         * :642-648 leaves trace()'s colour as it is, so it is not spelled out (the golden screenshots pin THIS path) */
        Ray camray = createCamRay((float)x_coord, (float)y_coord, width, height);
        finalcolor = trace(s, ambient, &camray);
    } else {
        /* opencl_kernel.cl:641-648 with MSAASAMPLES = msaa (a compile-time constant a maintainer may edit; no reference output
         * exists for any value but 1: parity unpinned beyond the arithmetic below) */
        finalcolor = F3(0.0f, 0.0f, 0.0f);
        for (int y = 0; y < msaa; y++) {
            for (int x = 0; x < msaa; x++) {
                Ray camray = createCamRay((float)x_coord + (float)x / msaa, (float)y_coord + (float)y / msaa, width, height);
                finalcolor = add3(finalcolor, trace(s, ambient, &camray));
            }
        }
        finalcolor = divs3(finalcolor, (float)(msaa * msaa));
    }
    finalcolor = div3(hable(finalcolor), hable(white_point));
    finalcolor = F3(cl_min(finalcolor.x, 1.0f), cl_min(finalcolor.y, 1.0f), cl_min(finalcolor.z, 1.0f));

    if (rgb_out) {
        rgb_out[3 * (size_t)work_item_id + 0] = finalcolor.x;
        rgb_out[3 * (size_t)work_item_id + 1] = finalcolor.y;
        rgb_out[3 * (size_t)work_item_id + 2] = finalcolor.z;
    }
    if (output) {
        rpt_pixel *p = &output[work_item_id];
        p->x = (float)x_coord;
        p->y = (float)y_coord;
        p->rgba[0] = to_u8(finalcolor.x);
        p->rgba[1] = to_u8(finalcolor.y);
        p->rgba[2] = to_u8(finalcolor.z);
        p->rgba[3] = 1;
        p->unspecified = 0;
    }
}

/* ------------------------------------------------------------------------- */
/* driver: rows [row_begin,row_end) split into row tiles pulled by threads     */

typedef struct {
    const rpt_oracle_args *a;
    Scene scene;
    int row_begin, row_end, tile_rows;
    volatile int next_tile;
    rpt_oracle_stats *stats;   /* per-thread array or NULL */
    int want_stats;
} Job;

typedef struct { Job *job; int tid; } Worker;

static void *worker_main(void *p) {
    Worker *w = (Worker *)p;
    Job *job = w->job;
    const rpt_oracle_args *a = job->a;
    rpt_oracle_stats local;
    memset(&local, 0, sizeof local);
    tls_stats = job->want_stats ? &local : NULL;
    f3 wp = F3(a->white_point[0], a->white_point[1], a->white_point[2]);
    for (;;) {
        int t = __sync_fetch_and_add(&job->next_tile, 1);
        int r0 = job->row_begin + t * job->tile_rows;
        if (r0 >= job->row_end) break;
        int r1 = r0 + job->tile_rows;
        if (r1 > job->row_end) r1 = job->row_end;
        for (int y = r0; y < r1; y++) {
            for (int x = 0; x < a->width; x++) {
                unsigned int id = (unsigned int)y * (unsigned int)a->width + (unsigned int)x;
                render_pixel(&job->scene, wp, a->ambient, a->width, a->height, a->msaa, id,
                             (rpt_pixel *)a->out_pixels, a->out_rgb);
            }
        }
    }
    if (job->want_stats) job->stats[w->tid] = local;
    free(tls_seen_walk);
    free(tls_seen_leaf);
    tls_seen_walk = tls_seen_leaf = NULL;
    tls_seen_cap = 0;
    tls_stats = NULL;
    return NULL;
}

int rpt_oracle_render(const rpt_oracle_args *a, int row_begin, int row_end, int threads,
                      rpt_oracle_stats *stats) {
    if (!a || a->width <= 0 || a->height <= 0) return -1;
    if (row_begin < 0) row_begin = 0;
    if (row_end > a->height) row_end = a->height;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;

    Job job;
    memset(&job, 0, sizeof job);
    job.a = a;
    job.scene.objects = (const rpt_object *)a->objects;
    job.scene.object_count = a->object_count;
    job.scene.vertices = (const rpt_float3 *)a->vertices;
    job.scene.normals = (const rpt_float3 *)a->normals;
    job.scene.uvs = (const rpt_float2 *)a->uvs;
    job.scene.triangles = (const uint32_t *)a->triangles;
    job.scene.octrees = (const rpt_octree *)a->octrees;
    job.scene.octreeTris = (const int32_t *)a->octreeTris;
    job.scene.textures = (const uint8_t *)a->textures;
    job.scene.texture_bytes = (int64_t)a->texture_bytes;
    job.scene.interval = a->interval;
    job.row_begin = row_begin;
    job.row_end = row_end;
    job.tile_rows = 4;
    job.next_tile = 0;
    job.want_stats = stats != NULL;
    rpt_oracle_stats per_thread[256];
    job.stats = per_thread;
    memset(per_thread, 0, sizeof per_thread);

    pthread_t th[256];
    Worker wk[256];
    for (int i = 0; i < threads; i++) {
        wk[i].job = &job;
        wk[i].tid = i;
        if (i > 0 && pthread_create(&th[i], NULL, worker_main, &wk[i]) != 0) return -2;
    }
    worker_main(&wk[0]);
    for (int i = 1; i < threads; i++) pthread_join(th[i], NULL);

    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (int i = 0; i < threads; i++) {
            const uint64_t *src = (const uint64_t *)&per_thread[i];
            uint64_t *dst = (uint64_t *)stats;
            for (size_t k = 0; k < sizeof(rpt_oracle_stats) / sizeof(uint64_t); k++) dst[k] += src[k];
        }
    }
    return 0;
}

/* ---- per-function entry points for known-answer tests against the HIP probes ---- */

int rpt_oracle_tri(const float *A, const float *B, const float *C, const float *org, const float *dir, float *out3) {
    Ray r = { F3(org[0], org[1], org[2]), F3(dir[0], dir[1], dir[2]) };
    float dist = 0; f2 uv = { 0, 0 };
    int h = intersect_triangle(F3(A[0], A[1], A[2]), F3(B[0], B[1], B[2]), F3(C[0], C[1], C[2]), &r, &dist, &uv);
    out3[0] = h ? dist : 0; out3[1] = h ? uv.x : 0; out3[2] = h ? uv.y : 0;
    return h;
}

int rpt_oracle_aabb(const float *bmin, const float *bmax, const float *org, const float *dir, float *d2, int *sides2) {
    Ray r = { F3(org[0], org[1], org[2]), F3(dir[0], dir[1], dir[2]) };
    f3 b[2] = { F3(bmin[0], bmin[1], bmin[2]), F3(bmax[0], bmax[1], bmax[2]) };
    f2 d = { 0, 0 }; int cs = 0, fs = 0;
    int h = intersect_AABB(b, &r, &d, &cs, &fs);
    d2[0] = h ? d.x : 0; d2[1] = h ? d.y : 0; sides2[0] = h ? cs : 0; sides2[1] = h ? fs : 0;
    return h;
}

void rpt_oracle_camray(float x, float y, int w, int h, float *dir3) {
    Ray r = createCamRay(x, y, w, h);
    dir3[0] = r.dir.x; dir3[1] = r.dir.y; dir3[2] = r.dir.z;
}

/* The octree walk at ray level (tests/test_gpu_kat.py against rpt_probe_walk): n object-space rays {origin.xyz, dir.xyz} through
 * the mesh object `object_index`; out8 per ray = {hit, dist, normal.xyz, uv.xy, 0}, the distance re-measured from (0,0,0) at unit
 * direction length. */
int rpt_oracle_octree_rays(const rpt_oracle_args *a, int object_index, const float *rays, float *out8, int n) {
    if (!a || !rays || !out8 || object_index < 0 || object_index >= a->object_count) return -1;
    Scene sc;
    memset(&sc, 0, sizeof sc);
    sc.objects = (const rpt_object *)a->objects;
    sc.object_count = a->object_count;
    sc.vertices = (const rpt_float3 *)a->vertices;
    sc.normals = (const rpt_float3 *)a->normals;
    sc.uvs = (const rpt_float2 *)a->uvs;
    sc.triangles = (const uint32_t *)a->triangles;
    sc.octrees = (const rpt_octree *)a->octrees;
    sc.octreeTris = (const int32_t *)a->octreeTris;
    if (sc.objects[object_index].type != RPT_MESH) return -1;
    for (int i = 0; i < n; i++) {
        Ray r;
        r.origin = (f3){rays[6 * i + 0], rays[6 * i + 1], rays[6 * i + 2]};
        r.dir = (f3){rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]};
        Hit hit;
        memset(&hit, 0, sizeof hit);
        hit.dist = 1e20f;
        const f3 zero = {0.0f, 0.0f, 0.0f};
        const int h = intersect_octree_core(&sc, object_index, r, zero, 1.0f, &hit);
        float *o = out8 + 8 * (size_t)i;
        o[0] = h ? 1.0f : 0.0f;
        o[1] = h ? hit.dist : 0.0f;
        o[2] = h ? hit.normal.x : 0.0f;
        o[3] = h ? hit.normal.y : 0.0f;
        o[4] = h ? hit.normal.z : 0.0f;
        o[5] = h ? hit.uv.x : 0.0f;
        o[6] = h ? hit.uv.y : 0.0f;
        o[7] = 0.0f;
    }
    return 0;
}


/* ---- object-level known-answer entry points (intersect_sphere :335-359, intersect_cube :312-333,
 * transformPoint* :75-104, sample_light :488-545) against rpt_probe_object of the HIP library ---- */
static void scene_from_args(const rpt_oracle_args *a, Scene *sc) {
    memset(sc, 0, sizeof *sc);
    sc->objects = (const rpt_object *)a->objects;
    sc->object_count = a->object_count;
    sc->vertices = (const rpt_float3 *)a->vertices;
    sc->normals = (const rpt_float3 *)a->normals;
    sc->uvs = (const rpt_float2 *)a->uvs;
    sc->triangles = (const uint32_t *)a->triangles;
    sc->octrees = (const rpt_octree *)a->octrees;
    sc->octreeTris = (const int32_t *)a->octreeTris;
    sc->textures = (const uint8_t *)a->textures;
    sc->texture_bytes = (int64_t)a->texture_bytes;
    sc->interval = a->interval;
}

/* which = 0: n rays {origin4, dir4} (8 floats, the object's rest frame) through object `object_index` with the intersector of its
 *            type (intersect_sphere / intersect_cube / intersect_octree); out8 = {hit, dist, normal.xyz, uv.xy, 0}
 * which = 1: sample_light: n shadow rays {origin4, dir4, lightDist} (9 floats, camera frame) against the whole scene with light
 *            `object_index`; out1 = index of the first occluder, or -1
 * which = 2: the four transforms on n vectors {x, y, z, w} (4 floats): out16 = {transformPoint(InvM, xyz), 0, transformPoint4D(Lorentz, v),
 *            transformDirection(InvM, xyz), 0, applyTranspose(InvM, xyz), 0} of object `object_index` */
int rpt_oracle_object_rays(const rpt_oracle_args *a, int which, int object_index, const float *in, float *out, int n) {
    if (!a || !in || !out || object_index < 0 || object_index >= a->object_count) return -1;
    Scene sc;
    scene_from_args(a, &sc);
    const rpt_object *obj = &sc.objects[object_index];
    for (int i = 0; i < n; i++) {
        if (which == 0) {
            const float *p = in + 8 * (size_t)i;
            Ray4D r;
            r.origin = F4(p[0], p[1], p[2], p[3]);
            r.dir = F4(p[4], p[5], p[6], p[7]);
            Hit hit;
            memset(&hit, 0, sizeof hit);
            hit.dist = 1e20f;
            int h = 0;
            switch (obj->type) {
            case RPT_SPHERE: h = intersect_sphere(&sc, object_index, &r, &hit); break;
            case RPT_CUBE:   h = intersect_cube(&sc, object_index, &r, &hit); break;
            case RPT_MESH:   h = intersect_octree(&sc, object_index, &r, &hit); break;
            default: return -1;
            }
            float *o = out + 8 * (size_t)i;
            o[0] = h ? 1.0f : 0.0f;
            o[1] = h ? hit.dist : 0.0f;
            o[2] = h ? hit.normal.x : 0.0f; o[3] = h ? hit.normal.y : 0.0f; o[4] = h ? hit.normal.z : 0.0f;
            o[5] = h ? hit.uv.x : 0.0f; o[6] = h ? hit.uv.y : 0.0f;
            o[7] = 0.0f;
        } else if (which == 1) {
            const float *p = in + 9 * (size_t)i;
            Ray4D r;
            r.origin = F4(p[0], p[1], p[2], p[3]);
            r.dir = F4(p[4], p[5], p[6], p[7]);
            out[i] = (float)sample_light(&sc, &r, p[8], object_index);
        } else if (which == 2) {
            const float *p = in + 4 * (size_t)i;
            float *o = out + 16 * (size_t)i;
            const f3 v = F3(p[0], p[1], p[2]);
            const f3 t0 = transformPoint(obj->InvM, v);
            const f4 t1 = transformPoint4D(obj->Lorentz, F4(p[0], p[1], p[2], p[3]));
            const f3 t2 = transformDirection(obj->InvM, v);
            const f3 t3 = applyTranspose(obj->InvM, v);
            o[0] = t0.x; o[1] = t0.y; o[2] = t0.z; o[3] = 0.0f;
            o[4] = t1.x; o[5] = t1.y; o[6] = t1.z; o[7] = t1.w;
            o[8] = t2.x; o[9] = t2.y; o[10] = t2.z; o[11] = 0.0f;
            o[12] = t3.x; o[13] = t3.y; o[14] = t3.z; o[15] = 0.0f;
        } else {
            return -1;
        }
    }
    return 0;
}

/* out4 = {side, uv'.xyz} of getOppositeBoxSide; out4b = {childIndex, uv'.xyz} of the octree child step */
void rpt_oracle_walk_steps(const float *scaledDir3, const float *uv3, float *out4, float *out4b) {
    f3 uv = F3(uv3[0], uv3[1], uv3[2]);
    int side = getOppositeBoxSide(F3(scaledDir3[0], scaledDir3[1], scaledDir3[2]), &uv);
    out4[0] = (float)side; out4[1] = uv.x; out4[2] = uv.y; out4[3] = uv.z;
    f3 uc = F3(uv3[0], uv3[1], uv3[2]);
    int child = octree_child_step(&uc);
    out4b[0] = (float)child; out4b[1] = uc.x; out4b[2] = uc.y; out4b[3] = uc.z;
}

void rpt_oracle_asin_atan2(float a, float y, float x, float *out2) {
    out2[0] = oracle_asinf(a);
    out2[1] = oracle_atan2f(y, x);
}

void rpt_oracle_hable(const float *in3, float *out3) {
    f3 r = hable(F3(in3[0], in3[1], in3[2]));
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
