// rpt_diag_kernels.hip.h — librpt_hip_diag.so only (make diag; included by rpt_kernels.hip.h under RPT_DIAGNOSTICS).
// Kernels that are measurement arms, not product: the instrumented walk (loop counters 7, primary rays only 8, per-wave
// timeline 11), round 1's prepass + masked kernel (26), the default at 4 / 6 waves per SIMD (40 / 42), round 2's walk (141 / 143),
// round 3's walk experiments (256 + flags), the persistent-workgroup kernels with LDS staging (60, 62, 63) and the per-workgroup
// ray queue (61) of rpt_persistent.hip.h.  What each one measured: profiles/r03_*.txt, DESIGN.md 6.
#pragma once

namespace rptd {

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


__global__ __launch_bounds__(256) void rpt_render_kernel_v1_diag(const KernelArgs a) { render_pixel_body<2>(a); }                                                      // 7
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void rpt_render_kernel_v1_timeline(const KernelArgs a) { render_pixel_body<4>(a); }      // 11
__global__ __launch_bounds__(256) void rpt_render_kernel_primary_only(const KernelArgs a) { render_pixel_body<5>(a); }                                                 // 8
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void rpt_render_kernel_v1_masked_w5(const KernelArgs a) { render_pixel_body<10>(a); }    // 26
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void rpt_render_kernel_ballot_w4(const KernelArgs a) { render_pixel_body<20>(a); }        // 40
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 6))) void rpt_render_kernel_ballot_w6(const KernelArgs a) { render_pixel_body<20>(a); }        // 42
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void rpt_render_kernel_r02walk_w5(const KernelArgs a) { render_pixel_body<120>(a); }      // 141: round 2's walk, natural order
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void rpt_render_kernel_r02walk_first_w5(const KernelArgs a) { render_pixel_body<123>(a); } // 143: round 2's walk, mesh band first
#define RPT_X_KERNEL(N) __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void rpt_render_kernel_x##N(const KernelArgs a) { render_pixel_body<N>(a); }
#define RPT_X1_KERNEL(N) __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5))) void rpt_render_kernel_x##N(const KernelArgs a) { render_pixel_body<N>(a); }
RPT_X1_KERNEL(657) RPT_X1_KERNEL(669) RPT_X1_KERNEL(673) RPT_X1_KERNEL(705) RPT_X1_KERNEL(717)      /* one wave per workgroup, like the product kernels they are compared with */
#define RPT_XW_KERNEL(N, W) __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W))) void rpt_render_kernel_x##N##_w##W(const KernelArgs a) { render_pixel_body<N>(a); }
RPT_XW_KERNEL(257, 6) RPT_XW_KERNEL(257, 4) RPT_XW_KERNEL(263, 4) RPT_XW_KERNEL(259, 4) RPT_XW_KERNEL(573, 4)
RPT_X_KERNEL(256) RPT_X_KERNEL(257) RPT_X_KERNEL(259) RPT_X_KERNEL(261) RPT_X_KERNEL(263) RPT_X_KERNEL(265) RPT_X_KERNEL(269)
RPT_X_KERNEL(273) RPT_X_KERNEL(277) RPT_X_KERNEL(285) RPT_X_KERNEL(305) RPT_X_KERNEL(317) RPT_X_KERNEL(337) RPT_X_KERNEL(349) RPT_X_KERNEL(401) RPT_X_KERNEL(785) RPT_X_KERNEL(529) RPT_X_KERNEL(541) RPT_X_KERNEL(561) RPT_X_KERNEL(573) RPT_X_KERNEL(589) RPT_X_KERNEL(605) RPT_X_KERNEL(621) RPT_X_KERNEL(625) RPT_X_KERNEL(637) RPT_X_KERNEL(641) RPT_X_KERNEL(653) RPT_X_KERNEL(689) RPT_X_KERNEL(701) RPT_X_KERNEL(593)

}  // namespace rptd

#include "rpt_persistent.hip.h"
