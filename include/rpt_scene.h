/*
 * rpt_scene.h — C-ABI of librpt_scene.so, the host-side scene front-end and per-frame
 * relativistic update that sit either side of the render path.
 *
 * Pure host C++ behind plain C entry points (no GPU, no HIP): it produces, in the reference's
 * layouts, the buffers that rpt.h's rpt_upload_scene()/rpt_set_objects() consume.
 *
 * Reference interfaces replaced (file:line under the reference root):
 *   rpt_scene_input            inputScene()               Render.cpp:211-416 (stdin DSL, README.md:17-75)
 *   rpt_scene_read_obj         ReadOBJ()                  Render.cpp:436-538 (+ Mesh.cpp:5-28, Octree.cpp:6-248)
 *   rpt_scene_read_texture     ReadTexture()              Render.cpp:418-434
 *   rpt_scene_add_texture_rgb8 ReadTexture() after decode Render.cpp:424-427
 *   rpt_scene_update_objects   render(), Lorentz part     Render.cpp:179-200 (+ Vector.cpp:175-232)
 *   rpt_scene_accelerate       render(), WASDQE keys      Render.cpp:149-176 (+ Vector.cpp:189-193)
 *   rpt_scene_advance_time     render()                   Render.cpp:177
 *   rpt_scene_toggle_interval  render(), I key            Render.cpp:136-147
 *   rpt_scene_reset_velocity   render(), R key            Render.cpp:149-156
 *   rpt_write_ppm, rpt_write_png  drawGL() (headless stand-in for the GL_POINTS draw) gl_interop.cpp:51-67
 *
 * Conventions: every function returning int returns 0 on success, nonzero on failure, and never
 * throws across the boundary; rpt_scene_last_error() describes the last failure (or holds the
 * parser's non-fatal diagnostics after a successful rpt_scene_input).  A scene is not thread-safe.
 */
#ifndef RPT_SCENE_H
#define RPT_SCENE_H

#include "rpt_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rpt_scene rpt_scene;

/* Decoder callback for T<path>: fill *rgb with a malloc()ed interleaved RGB8 image (top row first);
 * the library free()s it.  Return 0 on success. */
typedef int (*rpt_texture_decoder)(const char *path, unsigned char **rgb, int *width, int *height, void *user);

rpt_scene *rpt_scene_create(void);
void rpt_scene_destroy(rpt_scene *s);
const char *rpt_scene_last_error(const rpt_scene *s);

/* asset lookup: paths in scene files are resolved against `dir`, falling back to a
 * case-insensitive match; an alias replaces one path by another before resolution */
int rpt_scene_set_asset_root(rpt_scene *s, const char *dir);
int rpt_scene_add_alias(rpt_scene *s, const char *from, const char *to);
int rpt_scene_set_texture_decoder(rpt_scene *s, rpt_texture_decoder fn, void *user);

/* scene construction */
int rpt_scene_input(rpt_scene *s, const char *text);                 /* the whole DSL text, as piped to stdin */
int rpt_scene_read_obj(rpt_scene *s, const char *path);
/* ReadOBJ() in two steps, for an octree built elsewhere (rpt_build_octree on the GPU, include/rpt.h): import the
 * geometry (vertices, 9-word triangles, synthesised normals; Render.cpp:436-533), then append the finished
 * octree (what Mesh::GenerateOctree would have produced; indices absolute). */
int rpt_scene_read_obj_geometry(rpt_scene *s, const char *path, size_t *first_triangle_word);
int rpt_scene_append_octree(rpt_scene *s, const rpt_octree *nodes, size_t node_count, const int32_t *tris, size_t tri_count);
int rpt_scene_read_texture(rpt_scene *s, const char *path);
int rpt_scene_add_texture_rgb8(rpt_scene *s, const unsigned char *rgb, int width, int height);

/* camera / time state (Render.cpp:8-13): velocity in units of c, position as (t,x,y,z) */
int rpt_scene_set_camera(rpt_scene *s, const float velocity[3], const float position_txyz[4]);
int rpt_scene_get_camera(const rpt_scene *s, float velocity[3], float position_txyz[4]);
int rpt_scene_accelerate(rpt_scene *s, const float direction[3], int frame_ms);
int rpt_scene_reset_velocity(rpt_scene *s);
int rpt_scene_set_paused(rpt_scene *s, int paused);
int rpt_scene_advance_time(rpt_scene *s, int frame_ms);
int rpt_scene_set_interval(rpt_scene *s, int interval);              /* -1 light propagation on, 0 off */
int rpt_scene_toggle_interval(rpt_scene *s);

/* per-frame refresh of Object.Lorentz / InvLorentz / stationaryCam from the camera and object velocities */
int rpt_scene_update_objects(rpt_scene *s);

/* views of the current buffers and scalars; pointers stay valid until the scene is next modified */
int rpt_scene_get_desc(const rpt_scene *s, rpt_scene_desc *out);
int rpt_scene_get_params(const rpt_scene *s, float white_point[3], float *ambient, int *interval);
int rpt_scene_get_velocities(const rpt_scene *s, const rpt_float3 **velocities, size_t *count);
int rpt_scene_get_mesh_roots(const rpt_scene *s, const int **roots, size_t *count);

/* framebuffer consumer: write a 16 B/pixel framebuffer (row 0 = bottom, as GL draws it) as a binary
 * PPM with the top row first */
int rpt_write_ppm(const char *path, const void *pixels16, int width, int height);
/* the same image as an 8-bit RGB PNG (stored, i.e. uncompressed, zlib stream: no compression library needed) */
int rpt_write_png(const char *path, const void *pixels16, int width, int height);

#ifdef __cplusplus
}
#endif
#endif /* RPT_SCENE_H */
