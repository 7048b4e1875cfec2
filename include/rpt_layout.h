/*
 * rpt_layout.h — byte-level buffer layouts of the render path.
 *
 * These are the exact layouts the reference host writes and its kernel reads:
 *   Object   reference Object.h:6-22   == opencl_kernel.cl:21-36   (320 B)
 *   Octree   reference Octree.h:4-12   == opencl_kernel.cl:46-53   ( 96 B)
 *   Mesh SoA reference Mesh.h:7-13     (vertices/normals 16 B stride, uvs 8 B,
 *            triangles 9 x u32 per triangle [v,uv,n]x3, octreeTris i32,
 *            textures interleaved RGB8 rows, top row first)
 *   pixel    reference opencl_kernel.cl:652-659, gl_interop.cpp:56-57 (16 B)
 *
 * cl_float3 is cl_float4 (16 B, align 16) on the host side (CL/cl_platform.h),
 * so every float3 slot is 16 bytes with an unused 4th lane.
 *
 * Shared by plain C (oracle), C++ (host library) and HIP device code.
 */
#ifndef RPT_LAYOUT_H
#define RPT_LAYOUT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rpt_float2 { float x, y; } rpt_float2;                 /* cl_float2 */
typedef struct rpt_float4 { float x, y, z, w; } rpt_float4;           /* cl_float4 == cl_float3 */
typedef rpt_float4 rpt_float3;                                        /* 16 B slot, .w unused */

enum rpt_object_type { RPT_SPHERE = 0, RPT_CUBE = 1, RPT_MESH = 2 }; /* Object.h:4 */

typedef struct rpt_object {
    rpt_float4 M[4];          /*   0  object -> world, row-major rows          */
    rpt_float4 InvM[4];       /*  64  world -> object                          */
    rpt_float4 Lorentz[4];    /* 128  camera frame -> object rest frame (t,x,y,z) */
    rpt_float4 InvLorentz[4]; /* 192  object rest frame -> camera frame        */
    rpt_float4 stationaryCam; /* 256  camera event in the object's rest frame  */
    rpt_float3 color;         /* 272                                           */
    int32_t    type;          /* 288  rpt_object_type                          */
    int32_t    meshIndex;     /* 292  index of the mesh's octree root          */
    int32_t    textureIndex;  /* 296  BYTE offset into the texture pool, -1 = none */
    int32_t    textureWidth;  /* 300                                           */
    int32_t    textureHeight; /* 304                                           */
    uint8_t    light;         /* 308  bool                                     */
    uint8_t    _pad[3];       /* 309                                           */
    float      flashPeriod;   /* 312                                           */
    float      flashDuration; /* 316                                           */
} rpt_object;

typedef struct rpt_octree {
    rpt_float3 min;           /*  0 */
    rpt_float3 max;           /* 16 */
    int32_t    trisIndex;     /* 32  first entry in octreeTris */
    int32_t    trisCount;     /* 36 */
    int32_t    children[8];   /* 40  index z + 2y + 4x, -1 = leaf */
    int32_t    neighbors[6];  /* 72  -z,+z,-x,+x,-y,+y ; -1 = outside */
} rpt_octree;

typedef struct rpt_pixel {
    float   x, y;             /* 0,4  pixel coordinates as floats (GL vertex)  */
    uint8_t rgba[4];          /* 8    R,G,B,1 (GL colour, reinterpreted float) */
    uint32_t unspecified;     /* 12   never read by the consumer               */
} rpt_pixel;

/*
 * The eight read-only scene arrays the reference uploads once (main.cpp:33-55), in the
 * reference's layouts, as {pointer, element count} pairs.  Zero-length arrays are legal and
 * their pointer may be NULL (main.cpp:34 passes NULL for empty vectors).
 */
typedef struct rpt_scene_desc {
    const rpt_object *objects;    size_t object_count;     /* Object[]            arg 0,1 */
    const rpt_float3 *vertices;   size_t vertex_count;     /* cl_float3[]         arg 2   */
    const rpt_float3 *normals;    size_t normal_count;     /* cl_float3[]         arg 3   */
    const rpt_float2 *uvs;        size_t uv_count;         /* cl_float2[]         arg 4   */
    const uint32_t   *triangles;  size_t triangle_words;   /* u32[9*T] (words)    arg 5   */
    const rpt_octree *octrees;    size_t octree_count;     /* Octree[]            arg 6   */
    const int32_t    *octreeTris; size_t octree_tri_count; /* i32[]               arg 7   */
    const uint8_t    *textures;   size_t texture_bytes;    /* u8[] RGB8 pool      arg 8   */
} rpt_scene_desc;

#define RPT_TRI_STRIDE 9      /* u32 per triangle: v0,uv0,n0, v1,uv1,n1, v2,uv2,n2 */

#ifdef __cplusplus
}
#define RPT_SA(c, m) static_assert(c, m)
#else
#define RPT_SA(c, m) _Static_assert(c, m)
#endif

RPT_SA(sizeof(rpt_float2) == 8, "cl_float2");
RPT_SA(sizeof(rpt_float4) == 16, "cl_float4");
RPT_SA(sizeof(rpt_object) == 320, "Object is 320 B");
RPT_SA(offsetof(rpt_object, InvM) == 64, "InvM");
RPT_SA(offsetof(rpt_object, Lorentz) == 128, "Lorentz");
RPT_SA(offsetof(rpt_object, InvLorentz) == 192, "InvLorentz");
RPT_SA(offsetof(rpt_object, stationaryCam) == 256, "stationaryCam");
RPT_SA(offsetof(rpt_object, color) == 272, "color");
RPT_SA(offsetof(rpt_object, type) == 288, "type");
RPT_SA(offsetof(rpt_object, meshIndex) == 292, "meshIndex");
RPT_SA(offsetof(rpt_object, textureIndex) == 296, "textureIndex");
RPT_SA(offsetof(rpt_object, textureWidth) == 300, "textureWidth");
RPT_SA(offsetof(rpt_object, textureHeight) == 304, "textureHeight");
RPT_SA(offsetof(rpt_object, light) == 308, "light");
RPT_SA(offsetof(rpt_object, flashPeriod) == 312, "flashPeriod");
RPT_SA(offsetof(rpt_object, flashDuration) == 316, "flashDuration");
RPT_SA(sizeof(rpt_octree) == 96, "Octree is 96 B");
RPT_SA(offsetof(rpt_octree, max) == 16, "max");
RPT_SA(offsetof(rpt_octree, trisIndex) == 32, "trisIndex");
RPT_SA(offsetof(rpt_octree, trisCount) == 36, "trisCount");
RPT_SA(offsetof(rpt_octree, children) == 40, "children");
RPT_SA(offsetof(rpt_octree, neighbors) == 72, "neighbors");
RPT_SA(sizeof(rpt_pixel) == 16, "pixel is 16 B");

#endif /* RPT_LAYOUT_H */
