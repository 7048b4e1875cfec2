/*
 * rpt_oracle.h — C interface of the CPU oracle (test infrastructure; see rpt_oracle.c).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 */
#ifndef RPT_ORACLE_H
#define RPT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors the 15 kernel arguments of the reference (CLSetup.cpp:150-164). */
typedef struct rpt_oracle_args {
    const void *objects;      /* rpt_object[object_count]            arg 0 */
    int32_t object_count;     /*                                      arg 1 */
    const void *vertices;     /* float3 (16 B) []                     arg 2 */
    const void *normals;      /* float3 (16 B) []                     arg 3 */
    const void *uvs;          /* float2 []                            arg 4 */
    const void *triangles;    /* u32[9 * T]                           arg 5 */
    const void *octrees;      /* rpt_octree[]                         arg 6 */
    const void *octreeTris;   /* i32[]                                arg 7 */
    const void *textures;     /* u8[]                                 arg 8 */
    uint64_t texture_bytes;   /* size of the pool (address guard)           */
    float white_point[3];     /*                                      arg 9 */
    float ambient;            /*                                      arg 10 */
    int32_t width, height;    /*                                      arg 11, 12 */
    int32_t interval;         /*                                      arg 13 */
    void *out_pixels;         /* rpt_pixel[width*height] or NULL      arg 14 */
    float *out_rgb;           /* float[3*width*height] tonemapped RGB before packing, or NULL */
    int32_t msaa;             /* MSAASAMPLES of opencl_kernel.cl:7 (a compile-time constant there); 0 or 1 = the reference as shipped */
} rpt_oracle_args;

/* Work counters (whole call, summed over threads); names follow SURVEY.md §8a. */
typedef struct rpt_oracle_stats {
    uint64_t shadow_rays, sphere_tests, cube_tests, octree_calls, root_aabb_hits,
             inside_starts, inside_descent_steps, descent_steps, leaf_visits, tri_tests, pixels_hit;
    /* how often does one walk test the SAME triangle again (a triangle overlapping k leaves is listed in all of them)?
     * distinct_tri_tests: triangles tested at least once per walk, summed over walks (tri_tests - this = repeated tests);
     * repeats_of_previous_leaf: repeated tests whose triangle was also in the list of the leaf visited immediately before. */
    uint64_t distinct_tri_tests, repeats_of_previous_leaf;
} rpt_oracle_stats;

/* Render rows [row_begin,row_end) with `threads` host threads. stats may be NULL. 0 = ok. */
int rpt_oracle_render(const rpt_oracle_args *a, int row_begin, int row_end, int threads,
                      rpt_oracle_stats *stats);

/* Per-function known-answer entry points. */
int  rpt_oracle_tri(const float *A, const float *B, const float *C, const float *org, const float *dir, float *out3);
int  rpt_oracle_aabb(const float *bmin, const float *bmax, const float *org, const float *dir, float *d2, int *sides2);
void rpt_oracle_camray(float x, float y, int w, int h, float *dir3);
void rpt_oracle_hable(const float *in3, float *out3);
void rpt_oracle_walk_steps(const float *scaledDir3, const float *uv3, float *out4, float *out4b);
/* opencl_kernel.cl:206-306 on n object-space rays {origin.xyz, dir.xyz} through mesh object `object_index`; out8 per ray =
 * {hit, dist, normal.xyz, uv.xy, 0} (distance re-measured from the origin of the object's frame at unit direction length). */
int rpt_oracle_octree_rays(const rpt_oracle_args *a, int object_index, const float *rays, float *out8, int n);
/* Object-level entry points (see rpt_oracle.c): which = 0 one object's intersector on 4-D rays, 1 sample_light, 2 the transforms. */
int rpt_oracle_object_rays(const rpt_oracle_args *a, int which, int object_index, const float *in, float *out, int n);
void rpt_oracle_asin_atan2(float a, float y, float x, float *out2);   /* out2 = {asin(a), atan2(y, x)} as the oracle defines them */

#ifdef __cplusplus
}
#endif
#endif
