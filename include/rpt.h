/*
 * rpt.h — C-ABI of librpt_hip.so: the per-pixel render path of the Relativity Path Tracer as
 * hand-written HIP for MI355X (gfx950).
 *
 * The reference has no plugin/FFI interface for this path; its boundary is the sequence of OpenCL
 * calls its C++ host makes.  Each entry point below replaces one of those call sites (file:line
 * under the reference root), keeps the reference's buffer layouts (rpt_layout.h) and its
 * conventions: the host owns every source array and the library copies (the reference uses
 * blocking writes everywhere); calls are externally blocking unless named *_async; one host thread
 * per context; every call returns 0 on success / nonzero on failure and never throws or aborts
 * across the boundary (the reference ignores every cl_int; here they are reported).
 *
 *   rpt_create / rpt_destroy   initOpenCL()                        CLSetup.cpp:64-135
 *   rpt_upload_scene           8x cl::Buffer + enqueueWriteBuffer   main.cpp:33-55
 *   rpt_share_scene            (none: the reference keeps one frame in flight)
 *   rpt_set_objects            per-frame write of Object[] + setArg(0)   Render.cpp:202-203
 *   rpt_set_params             initCLKernel() setArg 1,9..13; resize/interval re-binds
 *                                                                   CLSetup.cpp:150-163, Render.cpp:116-117,141
 *   rpt_set_output             cl::BufferGL(vbo) + setArg(14)       main.cpp:58, Render.cpp:114-118
 *   rpt_render                 runKernel()                          CLSetup.cpp:167-191
 *
 * Not in the reference (it never reads back, has one device and no timing): rpt_read_framebuffer,
 * rpt_last_frame_ms, rpt_timed_frames, rpt_timing_*, the *_async/stream calls, rpt_create_multi, rpt_set_rows /
 * rpt_set_tile_pattern (pixel-row tiles for multi-GPU sharding), rpt_pack_/rpt_scatter_* (the exchange's two kernels),
 * rpt_build_octree (GPU counterpart of Mesh::GenerateOctree) and the test hooks rpt_probe, rpt_probe_walk, rpt_probe_object,
 * rpt_probe_division, rpt_set_debug_rgb, rpt_verify_frame, rpt_object_screen_rect / _bounds / _bounds_proposed,
 * rpt_certify_screen_bounds and rpt_mesh_segment_cull_record (the last five are host code: no device needed).
 *
 * There is no CPU or OpenCL fallback: without a gfx950 device rpt_create fails.
 */
#ifndef RPT_H
#define RPT_H

#include "rpt_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rpt_ctx rpt_ctx;

enum rpt_status {
    RPT_OK = 0,
    RPT_ERR_ARG = 1,        /* null / out-of-range argument */
    RPT_ERR_STATE = 2,      /* call made before the state it needs was set */
    RPT_ERR_SCENE = 3,      /* scene buffers fail validation (an index points outside its array) */
    RPT_ERR_DEVICE = 4,     /* HIP runtime error; see rpt_last_error */
    RPT_ERR_NOMEM = 5
};

#define RPT_TILE_ROWS 8     /* height of one pixel-row tile (the sharding unit of rpt_set_rows) */
/* rpt_render_async on a context that renders at most this many pixels per frame launches the latency kernel (43) like the
 * blocking call: so few walks do not fill the chip even with three frames in flight (measured crossover: 2560x1440). */
#define RPT_LATENCY_KERNEL_MAX_PIXELS 3000000

const char *rpt_version(void);

/* Create a context on HIP device `device_ordinal` (replaces platform/device pick, context+queue
 * creation and the run-time program build: the gfx950 code object is prebuilt). */
int rpt_create(rpt_ctx **out, int device_ordinal);
/* One context per listed device (multi-GPU hosts: SURVEY.md §8b); all or nothing — on failure every context already
 * created is destroyed, out[] is nulled and the first error is returned.  The same device may be listed more than once
 * (frame slots). */
int rpt_create_multi(rpt_ctx **out, const int *device_ordinals, int n);
void rpt_destroy(rpt_ctx *ctx);
const char *rpt_last_error(const rpt_ctx *ctx);

/* Upload the eight read-only scene arrays (copied; validated so that no index can leave its
 * array).  The Object[] in `scene` is taken as the first rpt_set_objects. */
int rpt_upload_scene(rpt_ctx *ctx, const rpt_scene_desc *scene);

/* Frames in flight (not in the reference, whose runKernel() finishes every frame before the next starts,
 * CLSetup.cpp:167-191).  One frame's critical path is the serial octree walk of its dearest pixel, which leaves
 * most of the GPU idle; a host that wants frame RATE keeps 2-3 frames in flight, one context per frame slot, each
 * with its own stream and output, rendering frame f in slot f mod n (rpt_set_objects + rpt_render_async, and
 * rpt_sync before the slot's output is consumed).  rpt_share_scene gives `ctx` the scene already resident in
 * `owner` (same device) instead of a second copy: the geometry is reference-counted, so either context may be
 * destroyed or given another scene at any time.  Object[] and the per-context settings are not shared. */
int rpt_share_scene(rpt_ctx *ctx, rpt_ctx *owner);

/* Refresh Object[] (count * 320 B, copied).  Called every frame by the reference's render(). */
int rpt_set_objects(rpt_ctx *ctx, const void *objects, int count);

/* Scalars: white_point (3 floats), ambient, width, height, interval (-1 or 0). */
int rpt_set_params(rpt_ctx *ctx, const float white_point[3], float ambient, int width, int height, int interval);

/* Output framebuffer: a device pointer to at least width*height*16 B (e.g. an interop buffer), or
 * NULL for a library-owned buffer (headless). */
int rpt_set_output(rpt_ctx *ctx, void *device_ptr_or_null);

/* Pixel-row tiles rendered by this context: tiles first_tile, first_tile+tile_step, ... of
 * RPT_TILE_ROWS rows each (default 0,1 = the whole frame).  With colour_plane != 0 the context
 * renders into a compact library-owned plane of 4 B/pixel (the packed R,G,B,1 word only), its
 * k-th local tile holding global tile first_tile + k*tile_step: the unit that is gathered. */
int rpt_set_rows(rpt_ctx *ctx, int first_tile, int tile_step, int colour_plane);
/* The general form: per period of `tile_step` tiles this context renders the `run` (a power of two, <= tile_step)
 * consecutive tiles that start at first_tile; local tile t is global tile (t / run) * tile_step + first_tile + t % run.
 * rpt_set_rows is run = 1.  Used for the WEIGHTED multi-GPU split: the root of the gather renders `run` tiles of every
 * period of run + N - 1 straight into the framebuffer (colour_plane = 0) and helper j the single tile run + j - 1 into
 * its plane — the root takes the larger share because its pixels need no exchange. */
int rpt_set_tile_pattern(rpt_ctx *ctx, int first_tile, int tile_step, int run, int colour_plane);

/* Launch on this HIP stream (hipStream_t as void*; NULL = the context's own stream).  The stream stays the caller's: it
 * must outlive every launch made on it (rpt_sync, or the caller's own synchronisation, before it is destroyed).  Switching
 * away from an external stream waits for the context's last launch through an event, not through the stream handle, and
 * rpt_destroy waits for the device — so a handle destroyed after its work has finished is never touched again. */
int rpt_set_stream(rpt_ctx *ctx, void *hip_stream);

/* Test hook: also write the tonemapped float RGB before 8-bit packing (3 floats/pixel, row-major,
 * width*height*12 B) to this device pointer; NULL disables. 1 = library-owned buffer. */
int rpt_set_debug_rgb(rpt_ctx *ctx, void *device_ptr_or_null_or_1);

/* Kernel variant: 0 = default (fastest validated); the others select alternative implementations of the same path for
 * A/B measurement.  All produce identical results.
 *   0   default: 44 when the current Object[] holds no mesh; else 43 for the blocking rpt_render and for contexts of at
 *       most RPT_LATENCY_KERNEL_MAX_PIXELS, 41 for rpt_render_async above that; 1 when the octree's children are not stored
 *       consecutively
 *   1   reads the reference's Octree/triangle layouts only (any valid octree; no culling)
 *   3   derived layouts, every object tested for every pixel, 5 waves per SIMD: the NO-CULL escape hatch, and the frame
 *       rpt_verify_frame compares the culled kernels with
 *   41  every wavefront builds its own object mask from per-object image-plane rectangles (computed on the host in
 *       rpt_set_objects) with one lane-parallel test + __ballot, 5 waves per SIMD: what rpt_render_async launches
 *   43  41 with the band of tile rows that holds the meshes dispatched first (whole-frame contexts) and the latency form of
 *       the walk: triangle records asked for one iteration ahead, a leaf's first record together with its node record
 *       (a second copy of it, addressable by node index): what the blocking rpt_render launches
 *   44  41 without the octree walk compiled in (61 VGPRs, no scratch, 8 waves per SIMD): what frames without a mesh object
 *       get; asked for explicitly while Object[] holds a mesh, 41 is launched instead
 *   50, 51      NOT bit-exact, opt-in only: 41 compiled with the arithmetic OpenCL C allows by default (fma contraction,
 *               2.5-ulp division, 3-ulp sqrt; csrc/rpt_relaxed.hip), 5 / 6 waves per SIMD.  Never chosen by variant 0.
 * Everything else — instrumented kernels (7 loop counters, 8 primary rays only, 11 per-wave timeline) and the measurement arms
 * of rounds 1-3 (26 prepass masks, 40 / 42 other occupancies, 141 / 143 round 2's walk, 60-63 persistent workgroups with LDS
 * staging and the per-workgroup ray queue, 256+ walk experiments) — is NOT in the product library:
 * `make -C relativitypathtracer_amd/csrc diag` builds librpt_hip_diag.so with them (csrc/rpt_diag_kernels.hip.h). */
int rpt_set_variant(rpt_ctx *ctx, int variant);
/* MSAASAMPLES of opencl_kernel.cl:7 — a compile-time constant of the reference, 1 as shipped; a maintainer who edits it gets
 * n x n camera rays per pixel at (x + i/n, y + j/n), summed and divided by n^2 before the tonemap (:641-648).  1 (default) = the
 * kernels above; 2..8 = the multi-sample form of the default kernel (in-wave cull; reported by rpt_last_variant as 46) or, with
 * variant 3, of the un-culled kernel (47); other variants refuse.  No reference output exists for any value but 1: the
 * arithmetic is the oracle's (tests/test_gpu_parity.py), its pin is the one-sample path's. */
int rpt_set_msaa(rpt_ctx *ctx, int samples_per_axis);
/* The kernel (a number of the list above) this context's last launch was made with; 0 before the first launch.  What
 * variant 0 resolved to: tests and bench.py name the kernel they measured from this, not from a copy of the rule. */
int rpt_last_variant(const rpt_ctx *ctx);

/* A culled-vs-un-culled self-check on the device.  The default kernels drop objects per wavefront from conservatively
 * sampled screen bounds and shadow rays per wavefront from segment-vs-box tests; a wrong bound would make an object vanish from
 * a tile without any error.  rpt_verify_frame renders the context's CURRENT state (objects, parameters, rows) once with the
 * kernel a frame would get (rpt_render_async's choice, or the variant set) and once with the un-culled kernel (3) into scratch
 * buffers, compares the packed colours of every pixel on the device and returns the number of pixels that differ (0 = the cull
 * changed nothing).  The context's framebuffer is not touched.  Cost: two frames + one reduction; meant for tests, soak runs and
 * a host that wants to check a new kind of scene, not for every frame. */
int rpt_verify_frame(rpt_ctx *ctx, unsigned long long *differing_pixels);

/* The per-object cull record of the default kernel, exposed for tests (host code, needs no device): the rectangle
 * {u0, v0, u1, v1} on the camera's image plane z = 0.5 (pixel (x, y) of a W x H frame looks through
 * ((x/W - 0.5) * W/H, y/H - 0.5)) outside which no primary ray can reach `object` (one 320-B Object with its per-frame
 * Lorentz / stationaryCam fields set).  root_bounds: min.xyz, max.xyz of a mesh object's octree root, else NULL.
 * +-3e38 on every side = never culled; u0 > u1 = not visible at all. */
int rpt_object_screen_rect(const void *object, int interval, const float *root_bounds_or_null, float rect_out[4]);
/* The whole record: the rectangle, then the diagonal slabs {p_lo, p_hi} on u + v and {m_lo, m_hi} on u - v that cut its
 * corners where that pays (+-3e38 = no cut); the slabs hold for |u| <= 2, |v| <= 0.55 (frames up to 4 : 1). */
int rpt_object_screen_bounds(const void *object, int interval, const float *root_bounds_or_null, float bounds_out[8]);
/* Both calls above return what the kernel USES: the region proposed by the outline sampling of csrc/rpt_screen_bounds.hpp if
 * csrc/rpt_bounds_certify.hpp could PROVE it (no pixel of a frame of at most 4 : 1 outside it can make the kernel's float
 * arithmetic report a hit of the object; the argument is in that file's header), else the full plane — the object is then
 * tested for every pixel, as in the reference (opencl_kernel.cl:382-425).  The two calls below expose the halves, for tests
 * and tools: the raw proposal, and the proof attempt for ANY claimed region (1 = proven, 0 = not; stats_out, if not NULL,
 * receives {reason, segment tests used, deepest halving, boundary segments}: reason 0 proven, 1 non-finite input, 2 the
 * boosted directions do not cover the sphere once, 3 float noise too large, 4 ray origin inside or near the shape, 5 no
 * witness / witness outside the claim, 6 test budget exhausted, 7 a boundary point's exact ray meets the shape). */
/* Test hook (host code): the shadow-ray cull record of mesh object `object_index` of the context's current Object[] —
 * {half extents xyz of the root box as mesh_ray_misses_root uses them (< 0: no cull of this object), constant and slope of the
 * segment cull's margin (slope < 0: no segment cull: the mesh's triangles are too large for it to be provable, or a matrix is
 * too ill-conditioned), allowance of the segment's end per unit of the rest-frame origin's L1 norm and its constant part,
 * K = the mesh's largest |e1| |e2|, L = its longest edge, 1 if the mesh's lists stay inside its root box}
 * (csrc/rpt_kernels.hip.h: mesh_ray_misses_root, mesh_segment_apart; csrc/rpt_api.hip: mesh_segment_cull_record). */
int rpt_mesh_segment_cull_record(rpt_ctx *ctx, int object_index, float out[10]);
int rpt_object_screen_bounds_proposed(const void *object, int interval, const float *root_bounds_or_null, float bounds_out[8]);
int rpt_certify_screen_bounds(const void *object, int interval, const float *root_bounds_or_null, const float bounds[8], int stats_out[4]);

/* Render one frame and wait for it (the reference's runKernel + finish). */
int rpt_render(rpt_ctx *ctx);
/* Enqueue one frame on the context's stream without waiting; rpt_sync waits. */
int rpt_render_async(rpt_ctx *ctx);
int rpt_sync(rpt_ctx *ctx);

void *rpt_output_ptr(rpt_ctx *ctx);          /* device pointer of the current framebuffer */
size_t rpt_output_bytes(rpt_ctx *ctx);
void *rpt_colour_plane_ptr(rpt_ctx *ctx);    /* device pointer of the compact plane (rpt_set_rows) */
/* Render the compact colour plane into caller-owned device memory (at least
 * local_tiles*RPT_TILE_ROWS*width*4 B), e.g. the send buffer of the gather; NULL = library-owned. */
int rpt_set_plane_output(rpt_ctx *ctx, void *device_ptr_or_null);

int rpt_read_framebuffer(rpt_ctx *ctx, void *host_dst, size_t bytes);
int rpt_read_debug_rgb(rpt_ctx *ctx, void *host_dst, size_t bytes);
int rpt_last_frame_ms(rpt_ctx *ctx, float *ms);     /* device time of the last rendered frame */
/* Render `frames` frames back to back and report the average device time per frame, measured
 * with HIP events on the launch stream. */
int rpt_timed_frames(rpt_ctx *ctx, int frames, float *avg_ms);

/* Per-frame device timing over a region: after rpt_timing_begin every rpt_render[_async] brackets
 * its kernel with its own pair of HIP events on the launch stream (up to `max_frames`);
 * rpt_timing_end waits for them and returns the summed kernel time and the frame count. */
int rpt_timing_begin(rpt_ctx *ctx, int max_frames);
int rpt_timing_end(rpt_ctx *ctx, float *total_ms, int *frames);
/* The same, returning every frame's launch duration (up to `capacity`) instead of their sum. */
int rpt_timing_end_frames(rpt_ctx *ctx, float *per_frame_ms, int capacity, int *frames);
/* The same region as SPANS: when each timed launch began and ended on the device, in ms after the FIRST timed launch of `base`
 * began (base: any context of the same device whose timing region started with this one's — frames in flight run on several
 * contexts, their spans share one clock this way).  Ends the region like rpt_timing_end_frames. */
int rpt_timing_end_spans(rpt_ctx *ctx, const rpt_ctx *base, float *begin_ms, float *end_ms, int capacity, int *frames);

/* Root side of the multi-GPU exchange: expand `n_ranks` gathered colour planes (rank r's plane at
 * planes + r*plane_stride_bytes) into the 16 B/pixel framebuffer `out16` (x, y, packed colour). */
int rpt_scatter_colour_plane(rpt_ctx *ctx, const void *planes, void *out16, int width, int height,
                             int n_ranks, int plane_stride_words, int reserved);
/* The same on an explicit HIP stream (NULL = the context's launch stream): lets the root overlap the
 * reassembly of frame k with the rendering of frame k+1. */
int rpt_scatter_colour_plane_on(rpt_ctx *ctx, void *hip_stream, const void *planes, void *out16, int width, int height,
                                int n_ranks, int plane_stride_words);

/* The exchange at 3 bytes per pixel: the fourth byte of every packed colour is the constant 1 (opencl_kernel.cl:657),
 * so a rank may send 3/4 of its plane.  rpt_pack_colour_plane3_on rewrites `pixels` (a multiple of 4; a plane of
 * whole 8-row tiles always is) packed words at `plane4` as 3*pixels bytes at `plane3`, on `hip_stream` (NULL = the
 * context's launch stream); rpt_scatter_colour_plane3_on is rpt_scatter_colour_plane_on for gathered 3-byte planes
 * that lie `plane_stride_bytes` apart. */
int rpt_pack_colour_plane3_on(rpt_ctx *ctx, void *hip_stream, const void *plane4, void *plane3, size_t pixels);
int rpt_scatter_colour_plane3_on(rpt_ctx *ctx, void *hip_stream, const void *planes3, void *out16, int width, int height,
                                 int n_ranks, size_t plane_stride_bytes);
/* Reassembly for the weighted split: planes3 holds n_ranks gathered 3-byte planes plane_stride_bytes apart (slot 0, the
 * root's, is not read); the helpers' tiles are written into out16, the root's own tiles (already rendered there) are
 * left alone. */
int rpt_scatter_helper_planes3_on(rpt_ctx *ctx, void *hip_stream, const void *planes3, void *out16, int width, int height,
                                  int n_ranks, int root_run, size_t plane_stride_bytes);

/* Diagnostic variant 7 only (librpt_hip_diag.so): loop-iteration counters of the octree walk of the last frame —
 * [0..2] leaf steps / triangle tests / descent steps summed over lanes, [3..5] the same counted
 * once per executing wavefront (lane sum / (64 * wave count) = SIMD utilisation of that loop); [6] longest walk,
 * [7] walks > 32 steps, [8] sum over leaf steps of distinct nodes among the active lanes, [9] sum of active lanes,
 * [10..15] histogram of distinct nodes per step (1, 2, 3-4, 5-8, 9-16, >16). */
int rpt_read_counters(rpt_ctx *ctx, unsigned long long out[16]);

/* Diagnostic variant 11 only (librpt_hip_diag.so): ten words per wavefront of the last frame — {start, end} stamps (100 MHz
 * s_memrealtime) and the cycle / iteration accounting of the walk's loops (tools/timeline.py);
 * wave w = (blockIdx.y*gridDim.x + blockIdx.x)*4 + wave-in-block. */
int rpt_read_wave_times(rpt_ctx *ctx, unsigned long long *out, size_t max_words, size_t *words);

/* GPU octree build (replaces Mesh::GenerateOctree, Mesh.cpp:5-28, and Subdivide, Octree.cpp:171-248): builds the
 * octree of the mesh whose triangles start at word `first_triangle_word` of `triangles` (the root lists every
 * triangle imported so far, as the reference's does).  All levels run on the device in one submission on the context's
 * stream (stop rule, split decisions, triangle/box classification, ordered lists); the call waits once, reads the levels
 * back and numbers them as the reference does: node order, lists, neighbour links are byte-identical to the host builder's.
 * (RPT_OCTREE_TIMING=1 in the environment prints the call's phases to stderr.)  Node and list
 * indices in the output are absolute, based at node_index_base / tri_index_base (the current lengths of the
 * host's octree and octreeTris arrays).  The two arrays are malloc()ed; release them with rpt_free_host. */
int rpt_build_octree(rpt_ctx *ctx, const rpt_float3 *vertices, size_t vertex_count, const uint32_t *triangles,
                     size_t triangle_words, size_t first_triangle_word, int node_index_base, int tri_index_base,
                     rpt_octree **nodes_out, size_t *node_count, int32_t **tris_out, size_t *tri_count);
void rpt_free_host(void *p);

/* Known-answer probes of individual device functions (tests): which = 0 intersect_triangle
 * (in 15 floats -> out 4), 1 intersect_AABB (12 -> 5), 2 createCamRay (4 -> 3), 3 hable (3 -> 3), 4 asin / atan2 of the
 * textured-sphere (u,v) (3 -> 2), 5 the walk's pure steps: exit face of a leaf and child selection, general and fast (6 -> 12). */
int rpt_probe(rpt_ctx *ctx, int which, const void *host_in, void *host_out, int n);
/* Test hook, one object's functions at ray level (the oracle's counterpart: rpt_oracle_object_rays): which = 0 n 4-D rays
 * {origin4, dir4} of object `object_index`'s rest frame through its intersector in the general form of opencl_kernel.cl:312-359 /
 * 200-308 (8 floats in, {hit, dist, normal.xyz, uv.xy, 0} out); 1 n shadow rays {origin4, dir4, lightDist} of the camera frame
 * through sample_light (:488-545) with light `object_index` (9 in; out 2 = occluded as the un-culled kernel decides it, and as the
 * culled kernels do — the same wave-level segment culls as in a frame); 2 the transforms of :75-104 on n vectors (4 in, 16 out:
 * transformPoint(InvM), transformPoint4D(Lorentz), transformDirection(InvM), applyTranspose(InvM)); 3 n primary rays given by
 * their camera direction through the form the default kernels use (3 in, 8 out as in 0). */
int rpt_probe_object(rpt_ctx *ctx, int which, int object_index, const float *host_in, float *host_out, int n);
/* Test hook / experiment (csrc/rpt_device_math.hip.h: three quotients by one scalar through ONE refined reciprocal): compares the
 * shared-reciprocal quotients with IEEE division on blocks * 256 * per_thread generated (x, y, z, s) sets on the device.  mode 0
 * random, 1 denominators with an all-ones significand, 2 normalize() (s = sqrt(dot(v, v))), 3 arbitrary bit patterns.  counts_out =
 * {sets inside the fast path's domain, mismatching quotients with one residual correction, with two, mismatches of the guarded
 * form over ALL sets, mismatching sets seen}; the first max_samples mismatching sets (4 floats each) go to samples_out. */
int rpt_probe_division(rpt_ctx *ctx, int mode, unsigned int seed, int blocks, int per_thread, unsigned long long counts_out[5], float *samples_out, int max_samples);
/* Test hook, the octree walk at ray level: n rays {origin.xyz, dir.xyz} in the object space of mesh object `object_index` of the
 * current Object[] go through the three walks of the product library — the reference's layouts (opencl_kernel.cl:200-308 as
 * written), the throughput walk of kernel 41 and the latency walk of kernel 43 — and host_out receives 3 x 8 floats per ray:
 * {hit, dist, normal.xyz, uv.xy, 0} per walk, the distance re-measured from the object's origin as :303-305 does.  The three
 * must agree bit for bit (tests/test_gpu_kat.py). */
int rpt_probe_walk(rpt_ctx *ctx, int object_index, const float *host_rays, float *host_out, int n);

#ifdef __cplusplus
}
#endif
#endif /* RPT_H */
