// rpt_scene_capi.cpp — extern "C" entry points of librpt_scene.so (include/rpt_scene.h).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <vector>

#include "../../../include/rpt_scene.h"
#include "rpt_scene.h"

struct rpt_scene {
    rpt::Scene scene;
};

#define RPT_GUARD(s)  if (!(s)) return -1; try {
#define RPT_END(s)    } catch (const std::exception &e) { (s)->scene.lastError = e.what(); return -2; } catch (...) { (s)->scene.lastError = "unknown error"; return -2; }

extern "C" {

rpt_scene *rpt_scene_create(void) {
    try { return new rpt_scene(); } catch (...) { return nullptr; }
}

void rpt_scene_destroy(rpt_scene *s) { delete s; }

const char *rpt_scene_last_error(const rpt_scene *s) { return s ? s->scene.lastError.c_str() : "null scene"; }

int rpt_scene_set_asset_root(rpt_scene *s, const char *dir) {
    RPT_GUARD(s)
    s->scene.assetRoot = dir ? dir : ".";
    return 0;
    RPT_END(s)
}

int rpt_scene_add_alias(rpt_scene *s, const char *from, const char *to) {
    RPT_GUARD(s)
    if (!from || !to) return -1;
    s->scene.aliases[from] = to;
    return 0;
    RPT_END(s)
}

int rpt_scene_set_texture_decoder(rpt_scene *s, rpt_texture_decoder fn, void *user) {
    RPT_GUARD(s)
    if (!fn) { s->scene.decoder = nullptr; return 0; }
    s->scene.decoder = [fn, user](const std::string &path, rpt::TextureImage &out, std::string &err) {
        unsigned char *rgb = nullptr;
        int w = 0, h = 0;
        if (fn(path.c_str(), &rgb, &w, &h, user) != 0 || !rgb || w <= 0 || h <= 0) {
            std::free(rgb);
            err = "decoder failed for " + path;
            return false;
        }
        out.width = w;
        out.height = h;
        out.rgb.assign(rgb, rgb + (size_t)3 * w * h);
        std::free(rgb);
        return true;
    };
    return 0;
    RPT_END(s)
}

int rpt_scene_input(rpt_scene *s, const char *text) {
    RPT_GUARD(s)
    if (!text) return -1;
    std::istringstream in(text);
    return s->scene.inputScene(in) ? 0 : 1;
    RPT_END(s)
}

int rpt_scene_read_obj(rpt_scene *s, const char *path) {
    RPT_GUARD(s)
    if (!path) return -1;
    return s->scene.ReadOBJ(path) ? 0 : 1;
    RPT_END(s)
}

int rpt_scene_read_obj_geometry(rpt_scene *s, const char *path, size_t *first_triangle_word) {
    RPT_GUARD(s)
    if (!path || !first_triangle_word) return -1;
    int first = 0;
    if (!s->scene.ReadOBJGeometry(path, first)) return 1;
    *first_triangle_word = (size_t)first;
    return 0;
    RPT_END(s)
}

int rpt_scene_append_octree(rpt_scene *s, const rpt_octree *nodes, size_t node_count, const int32_t *tris, size_t tri_count) {
    RPT_GUARD(s)
    return s->scene.AppendOctree(nodes, node_count, tris, tri_count) ? 0 : 1;
    RPT_END(s)
}

int rpt_scene_read_texture(rpt_scene *s, const char *path) {
    RPT_GUARD(s)
    if (!path) return -1;
    return s->scene.ReadTexture(path) ? 0 : 1;
    RPT_END(s)
}

int rpt_scene_add_texture_rgb8(rpt_scene *s, const unsigned char *rgb, int width, int height) {
    RPT_GUARD(s)
    return s->scene.AddTexture(rgb, width, height) ? 0 : 1;
    RPT_END(s)
}

int rpt_scene_set_camera(rpt_scene *s, const float v[3], const float p[4]) {
    RPT_GUARD(s)
    if (v) s->scene.cameraVelocity = rpt::make_float3(v[0], v[1], v[2]);
    if (p) s->scene.cameraPos = rpt::make_float4(p[0], p[1], p[2], p[3]);
    return 0;
    RPT_END(s)
}

int rpt_scene_get_camera(const rpt_scene *s, float v[3], float p[4]) {
    if (!s) return -1;
    if (v) { v[0] = s->scene.cameraVelocity.x; v[1] = s->scene.cameraVelocity.y; v[2] = s->scene.cameraVelocity.z; }
    if (p) { p[0] = s->scene.cameraPos.x; p[1] = s->scene.cameraPos.y; p[2] = s->scene.cameraPos.z; p[3] = s->scene.cameraPos.w; }
    return 0;
}

int rpt_scene_accelerate(rpt_scene *s, const float d[3], int frame_ms) {
    RPT_GUARD(s)
    if (!d) return -1;
    s->scene.accelerate(rpt::make_float3(d[0], d[1], d[2]), frame_ms);
    return 0;
    RPT_END(s)
}

int rpt_scene_reset_velocity(rpt_scene *s) {
    RPT_GUARD(s)
    s->scene.cameraVelocity = rpt::make_float3(0, 0, 0);
    return 0;
    RPT_END(s)
}

int rpt_scene_set_paused(rpt_scene *s, int paused) {
    RPT_GUARD(s)
    s->scene.stopTime = paused != 0;
    return 0;
    RPT_END(s)
}

int rpt_scene_advance_time(rpt_scene *s, int frame_ms) {
    RPT_GUARD(s)
    s->scene.advanceTime(frame_ms);
    return 0;
    RPT_END(s)
}

int rpt_scene_set_interval(rpt_scene *s, int interval) {
    RPT_GUARD(s)
    s->scene.interval = interval;
    return 0;
    RPT_END(s)
}

int rpt_scene_toggle_interval(rpt_scene *s) {
    RPT_GUARD(s)
    s->scene.toggleInterval();
    return 0;
    RPT_END(s)
}

int rpt_scene_update_objects(rpt_scene *s) {
    RPT_GUARD(s)
    s->scene.updateObjects();
    return 0;
    RPT_END(s)
}

int rpt_scene_get_desc(const rpt_scene *s, rpt_scene_desc *out) {
    if (!s || !out) return -1;
    *out = s->scene.desc();
    return 0;
}

int rpt_scene_get_params(const rpt_scene *s, float wp[3], float *ambient, int *interval) {
    if (!s) return -1;
    if (wp) { wp[0] = s->scene.white_point.x; wp[1] = s->scene.white_point.y; wp[2] = s->scene.white_point.z; }
    if (ambient) *ambient = s->scene.ambient;
    if (interval) *interval = s->scene.interval;
    return 0;
}

int rpt_scene_get_velocities(const rpt_scene *s, const rpt_float3 **v, size_t *count) {
    if (!s || !v || !count) return -1;
    *v = s->scene.velocities.empty() ? nullptr : s->scene.velocities.data();
    *count = s->scene.velocities.size();
    return 0;
}

int rpt_scene_get_mesh_roots(const rpt_scene *s, const int **roots, size_t *count) {
    if (!s || !roots || !count) return -1;
    *roots = s->scene.theMesh.meshIndices.empty() ? nullptr : s->scene.theMesh.meshIndices.data();
    *count = s->scene.theMesh.meshIndices.size();
    return 0;
}

int rpt_write_ppm(const char *path, const void *pixels16, int width, int height) {
    if (!path || !pixels16 || width <= 0 || height <= 0) return -1;
    FILE *f = std::fopen(path, "wb");
    if (!f) return 1;
    std::fprintf(f, "P6\n%d %d\n255\n", width, height);
    const rpt_pixel *px = (const rpt_pixel *)pixels16;
    for (int row = height - 1; row >= 0; row--)
        for (int x = 0; x < width; x++) std::fwrite(px[(size_t)row * width + x].rgba, 1, 3, f);
    return std::fclose(f) == 0 ? 0 : 1;
}

// PNG without a compression library: 8-bit RGB, filter 0, zlib "stored" blocks (RFC 1950/1951), CRC-32 and
// Adler-32 computed here.  Lossless like the PPM, and every viewer opens it.
int rpt_write_png(const char *path, const void *pixels16, int width, int height) {
    if (!path || !pixels16 || width <= 0 || height <= 0) return -1;
    static uint32_t crc_table[256];
    if (!crc_table[1])
        for (uint32_t n = 0; n < 256; n++) {
            uint32_t c = n;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            crc_table[n] = c;
        }
    FILE *f = std::fopen(path, "wb");
    if (!f) return 1;
    auto be32 = [](uint8_t *p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; };
    auto chunk = [&](const char *type, const std::vector<uint8_t> &data) {
        uint8_t head[8];
        be32(head, (uint32_t)data.size());
        std::memcpy(head + 4, type, 4);
        std::fwrite(head, 1, 8, f);
        if (!data.empty()) std::fwrite(data.data(), 1, data.size(), f);
        uint32_t c = 0xffffffffu;
        for (int i = 4; i < 8; i++) c = crc_table[(c ^ head[i]) & 0xff] ^ (c >> 8);
        for (uint8_t b : data) c = crc_table[(c ^ b) & 0xff] ^ (c >> 8);
        uint8_t tail[4];
        be32(tail, c ^ 0xffffffffu);
        std::fwrite(tail, 1, 4, f);
    };
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::fwrite(sig, 1, 8, f);
    std::vector<uint8_t> ihdr(13);
    be32(&ihdr[0], (uint32_t)width);
    be32(&ihdr[4], (uint32_t)height);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;      // 8 bit, truecolour, no interlace
    chunk("IHDR", ihdr);
    // raw scanlines, top row first (framebuffer row 0 is the bottom row, gl_interop.cpp:51-67), filter byte 0
    const rpt_pixel *px = (const rpt_pixel *)pixels16;
    const size_t stride = 1 + (size_t)width * 3;
    std::vector<uint8_t> raw(stride * height);
    for (int row = 0; row < height; row++) {
        uint8_t *dst = &raw[stride * row];
        *dst++ = 0;
        const rpt_pixel *src = px + (size_t)(height - 1 - row) * width;
        for (int x = 0; x < width; x++, dst += 3) std::memcpy(dst, src[x].rgba, 3);
    }
    std::vector<uint8_t> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78);
    z.push_back(0x01);
    uint32_t s1 = 1, s2 = 0;
    for (size_t off = 0; off < raw.size();) {
        const size_t n = raw.size() - off < 65535 ? raw.size() - off : 65535;
        z.push_back(off + n == raw.size() ? 1 : 0);                          // BFINAL, BTYPE = 00 (stored)
        z.push_back(n & 0xff); z.push_back(n >> 8);
        z.push_back(~n & 0xff); z.push_back((~n >> 8) & 0xff);
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for (size_t i = off; i < off + n; i++) { s1 = (s1 + raw[i]) % 65521u; s2 = (s2 + s1) % 65521u; }
        off += n;
    }
    uint8_t ad[4];
    be32(ad, (s2 << 16) | s1);
    z.insert(z.end(), ad, ad + 4);
    chunk("IDAT", z);
    chunk("IEND", {});
    return std::fclose(f) == 0 ? 0 : 1;
}

}  // extern "C"
