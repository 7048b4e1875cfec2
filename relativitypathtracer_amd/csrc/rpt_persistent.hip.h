// rpt_persistent.hip.h — the persistent-workgroup form of the render path (gfx950, wave64).
//
// Same per-pixel arithmetic as rpt_kernels.hip.h (every intersector, the shading and the tonemap are the functions of that
// file, called with the same operands in the same order: results are bit-identical by construction); what differs is how the
// work is organised on the chip:
//   * the grid is as many workgroups as the chip holds at once (5 per CU x 4 waves) and every WAVE claims 8x8-pixel tiles from
//     one atomic counter until the frame is done — no wave launch, no argument load and no rectangle load per tile, and the band
//     of tile rows that holds the meshes (the frame's long walks) is claimed first, tile by tile, while the rest goes in runs;
//   * what every tile of a frame reads again is staged ONCE per workgroup in LDS: the objects' screen rectangles (the in-wave
//     cull) and the links of the octrees' top levels (breadth-first numbering: nodes [0, top_count)), so a descent from a root
//     runs on ds_read until it leaves those levels and on 4-byte loads below them (DNode::link says which children are leaves:
//     no lookup is spent on finding that out);
//   * the octree walk runs in WAVE-UNIFORM control flow (a lane that has no ray, or whose ray has left the octree, carries a false
//     predicate instead of leaving the loop), so that at every leaf step the whole wave can stage the triangle records of the
//     few distinct leaves its lanes stand in (2.4 on average for 52 lanes, profiles/r02_divergence.txt) into its own LDS arena
//     with ONE cooperative 16-byte load per leaf, and the triangle loop then reads LDS broadcasts: the vector memory pipeline
//     sees each record once per wave instead of once per lane and iteration, and a leaf step waits for global memory once for
//     all of its triangles instead of once per triangle.
// The per-workgroup ray queue is at the end of this file (render_strip_queued).
// MEASURED AND NOT ADOPTED (round 3; profiles/r03_persistent_lds_ab.txt, r03_rayqueue_ab.txt): every form here is bit-identical
// to the oracle and every form is slower than the kernel that is launched tile by tile — which is why this file is part of the
// diagnostics library only.  DESIGN.md 6.3 has the numbers and the mechanism (one memory counter per wave, in issue order:
// a persistent wave's next tile waits for its last tile's store and for its claim; a queue's barriers put four waves in lock step).
#pragma once
/* included by rpt_diag_kernels.hip.h (diagnostics build), after rpt_kernels.hip.h */

#ifndef RPT_RELAXED_FP
#pragma clang fp contract(off)

namespace rptd {

#define RPT_SKY_RUN 4            /* tiles per claim outside the band of mesh rows */
#define RPT_ARENA_TRIS 48        /* triangle records a wave can stage per leaf step (48 B each) */
#define RPT_QUEUE_MAX 256        /* rays a workgroup's queue holds: one per pixel of its 32x8 strip */

struct QueuedRay {               // 32 B: one walk request
    float dx, dy, dz;            // object-space direction, normalised
    float ox, oy, oz;            // object-space origin
    int pixel;                   // slot of the pixel in the strip (wave * 64 + lane)
    int pad;
};
struct QueuedResult { float dist, u, v; int tri; };      // tri < 0: no hit

struct PersistentLds {
    float4 rects[128];                                   // per object: rectangle, diagonal slabs (KernelArgs::rects)
    int top_links[RPT_TOP_MAX];                          // DNode::link of nodes [0, top_count)
};
struct ArenaLds { v4f tri[4][RPT_ARENA_TRIS * 3]; };      // per wave: staged triangle records
struct QueueLds {
    QueuedRay rays[RPT_QUEUE_MAX];
    QueuedResult results[RPT_QUEUE_MAX];
    int count;
    unsigned int strip_claim;
};

__shared__ PersistentLds rpt_plds;
__shared__ ArenaLds rpt_alds;
__shared__ QueueLds rpt_qlds;

// (Written so that the LDS read and the global load stay two instructions: a select between the two ADDRESSES would make the
// compiler issue one flat_load, which waits on both memory counters and takes the slow path through the aperture check.)
template <bool TOPLDS>
RPT_DEV int node_link(const KernelArgs &a, int idx) {
    if (!TOPLDS) return a.links[idx];
    int w = *reinterpret_cast<const volatile int *>(&rpt_plds.top_links[idx < RPT_TOP_MAX ? idx : 0]);   // (volatile: keeps it a ds_read of its own)
    if (idx >= a.top_count) w = a.links[idx];
    return w;
}

// From an inner node with link word w down to the leaf that holds uv (opencl_kernel.cl:256-261: the same child steps; only
// what is READ per level differs — one link word instead of a node record).
template <bool TOPLDS>
RPT_DEV int descend_links(const KernelArgs &a, int w, f3 &uv) {
    int idx;
    for (;;) {
        const int k = octree_child_step_fast(uv);
        idx = (w & RPT_LINK_CHILD_MASK) + k;
        if ((w >> (24 + k)) & 1) break;          // that child is a leaf
        w = node_link<TOPLDS>(a, idx);
    }
    return idx;
}

RPT_DEV void load_box(const KernelArgs &a, int idx, f3 &nmin, f3 &nmax, int &link, int &leafBegin) {
    const v4f *p = reinterpret_cast<const v4f *>(a.dnodes + idx);
    const v4f lo = p[0], hi = p[1];
    nmin = mk3(lo.x, lo.y, lo.z);
    nmax = mk3(hi.x, hi.y, hi.z);
    link = __float_as_int(lo.w);
    leafBegin = __float_as_int(hi.w) & RPT_NODE_BEGIN_MASK;
}

// opencl_kernel.cl:200-286 for the lanes whose `alive` is set, called by ALL lanes of the wave in uniform control flow.
// In: object-space ray (direction normalised), hit_dist = the caller's 1e20f.  Out: didHit, and for a hit the parametric
// distance, the barycentric (u,v) and the triangle id (what opencl_kernel.cl:287-306 then works from: mesh_hit_finish).
template <bool STAGE, bool TOPLDS = true>
RPT_DEV void walk_uniform(const KernelArgs &a, int wave, int root, f3 origin, f3 dir, bool alive, float &hit_dist, f2 &hit_uv,
                          int &hitTri, bool &didHit) {
    const int lane = threadIdx.x & 63;
    Ray ray;
    ray.origin = origin;
    ray.dir = dir;
    f3 nmin, nmax;
    int link, leafBegin;
    load_box(a, root, nmin, nmax, link, leafBegin);       // root is wave-uniform: scalar loads
    f2 d;
    int closeSide, farSide;
    alive = intersect_AABB(nmin, nmax, ray, d, closeSide, farSide) && alive;
    int curr = root;
    f3 uv = origin + dir * d.x;
    if (alive && d.x < 0) {     // the ray starts inside the root: go to the leaf that holds the origin
        uv = (origin - nmin) / (nmax - nmin);
        if (link != -1) {
            curr = descend_links<TOPLDS>(a, link, uv);
            load_box(a, curr, nmin, nmax, link, leafBegin);
        }
        alive = intersect_AABB(nmin, nmax, ray, d, closeSide, farSide);
        uv = origin + dir * d.x;
    }
    const ExitPlan plan = makeExitPlan(normalize(dir / (nmax - nmin)));
    didHit = false;
    hitTri = 0;
    v4f *arena = rpt_alds.tri[wave];
    for (int steps = 1; steps <= RPT_MAX_LEAF_STEPS; steps++) {
        alive = alive && curr != -1;
        if (__ballot(alive) == 0ull) break;
        int leafCount = 0;
        leafBegin = 0;
        if (alive) {
            load_box(a, curr, nmin, nmax, link, leafBegin);
            uv = (uv - nmin) / (nmax - nmin);
            if (link != -1) {
                curr = descend_links<TOPLDS>(a, link, uv);
                load_box(a, curr, nmin, nmax, link, leafBegin);
            }
            leafCount = a.dnodes[curr].leafCount;
        }
        // ---- the wave stages the records of the distinct leaves its lanes stand in (skipped while no lane has two triangles to
        // test: one triangle costs one round trip either way)
        int my_rec = -1;
        if (STAGE) {
            unsigned long long todo = __ballot(leafCount > 0);
            if (__ballot(leafCount > 1) != 0ull) {
                int used = 0;
                while (todo != 0ull) {
                    const int leader = __ffsll((long long)todo) - 1;
                    const int lb = __builtin_amdgcn_readlane(leafBegin, leader);
                    const int lc = __builtin_amdgcn_readlane(leafCount, leader);
                    const bool same = leafCount > 0 && leafBegin == lb;
                    todo &= ~__ballot(same);
                    if (used + lc > RPT_ARENA_TRIS) continue;           // no room left: those lanes read global memory
                    const v4f *src = reinterpret_cast<const v4f *>(a.dtris + lb);
                    for (int p0 = 0; p0 < 3 * lc; p0 += 64) {           // one pass for up to 21 triangles
                        const int piece = p0 + lane;
                        if (piece < 3 * lc) arena[3 * used + piece] = src[piece];
                    }
                    if (same) my_rec = used;
                    used += lc;
                }
            }
        }
        if (alive) {
            for (int k = 0; k < leafCount; k++) {
                v4f t0, t1;
                float e2z;
                int tri;
                if (STAGE && my_rec >= 0) {
                    const v4f *q = arena + 3 * (my_rec + k);
                    t0 = q[0];
                    t1 = q[1];
                    const float2 t2 = *reinterpret_cast<const float2 *>(q + 2);
                    e2z = t2.x;
                    tri = __float_as_int(t2.y);
                } else {
                    const v4f *q = reinterpret_cast<const v4f *>(a.dtris + leafBegin + k);
                    t0 = q[0];
                    t1 = q[1];
                    const float2 t2 = *reinterpret_cast<const float2 *>(q + 2);
                    e2z = t2.x;
                    tri = __float_as_int(t2.y);
                }
                float dist;
                f2 triUV;
                if (intersect_triangle_edges(mk3(t0.x, t0.y, t0.z), mk3(t0.w, t1.x, t1.y), mk3(t1.z, t1.w, e2z), ray, dist, triUV)) {
                    if (0 <= dist && dist < hit_dist) {
                        hitTri = tri;
                        hit_dist = dist;
                        hit_uv = triUV;
                        didHit = true;
                    }
                }
            }
            const f3 extents = nmax - nmin;
            farSide = getOppositeBoxSide(plan, uv);
            uv = nmin + uv * extents;
            curr = a.dnodes[curr].nb[farSide];
            if (exit_is_past_hit(uv - origin, hit_dist, didHit)) alive = false;
        }
    }
}

// sample_light (opencl_kernel.cl:488-545) for the lanes whose `need` is set, in uniform control flow.  The reference returns
// at the first occluder; the answer is the OR over the objects, so the order in which a lane finds its occluder is free.
template <bool STAGE>
RPT_DEV bool shadow_uniform(const KernelArgs &a, int wave, f4 origin4, f4 dir4, float lightDist, int lightIndex, bool need) {
    const f3 nd = normalize(yzw(dir4));
    const f4 lightDir0 = mk4((float)a.interval, nd.x, nd.y, nd.z);
    bool occluded = false;
    for (int j = 0; j < a.object_count; j++) {
        if (j == lightIndex) continue;
        const bool act = need && !occluded;
        if (__ballot(act) == 0ull) break;
        const rpt_object &obj = a.objects[j];
        const f4 ev = transformPoint4D(obj.Lorentz, origin4);
        const f4 ld = transformPoint4D(obj.Lorentz, lightDir0);
        const f3 origin = transformPoint(obj.InvM, yzw(ev));
        f3 dir = transformDirection(obj.InvM, yzw(ld));
        // the wave-level segment culls of intersect_object (rpt_kernels.hip.h), same margins
        if (obj.type != RPT_MESH) {
            if (__ballot(act && !(unit_segment_apart(origin, dir, lightDist) && lightDist > 0.0f)) == 0ull) continue;
        } else if (a.dobjs[j].mh[0] >= 0.0f) {
            bool idle = mesh_ray_misses_root(a.dobjs[j], origin, dir);
            if (a.dobjs[j].mslope >= 0.0f) idle = idle | (mesh_segment_apart(a.dobjs[j], yzw(ev), origin, dir, lightDist) && lightDist > 0.0f);
            if (__ballot(act && !idle) == 0ull) continue;
        }
        const float scale = length(dir);
        dir = dir / scale;
        Hit nh;
        nh.dist = 1e20f;
        bool got = false;
        switch (obj.type) {
        case RPT_SPHERE: {
            const f3 rayToSphere = -origin;
            if (act) got = sphere_core(obj, rayToSphere, dot(rayToSphere, rayToSphere) - 1.0f, dir, scale, nh, obj.textureIndex != -1);
            break;
        }
        case RPT_CUBE:
            if (act) got = cube_core(obj, origin, cube_winding(origin), dir, scale, nh);
            break;
        case RPT_MESH: {
            int tri;
            walk_uniform<STAGE>(a, wave, a.dobjs[j].root, origin, dir, act, nh.dist, nh.uv, tri, got);
            if (got) mesh_hit_finish(a, obj, origin, dir, tri, yzw(ev), length(yzw(ld)), nh);
            break;
        }
        default: break;
        }
        if (act && got && nh.dist < lightDist) occluded = true;
    }
    return occluded;
}

// The light loop of trace (opencl_kernel.cl:572-601) for the lanes with a hit, in uniform control flow.
template <bool STAGE>
RPT_DEV f3 shade_uniform(const KernelArgs &a, int wave, f4 rayDir, const Hit &hit, bool h) {
    const int ho_i = h ? hit.object : 0;
    const rpt_object &ho = a.objects[ho_i];
    f3 hcolor = mk3(0.0f, 0.0f, 0.0f), color = mk3(0.0f, 0.0f, 0.0f);
    if (h) {
        hcolor = ho.textureIndex != -1 ? sample_texture(a, ho, hit.uv) : ld3(ho.color);
        if (ho.flashPeriod > 0) {   // proper-time flash, opencl_kernel.cl:476-482
            const float event_x = ho.stationaryCam.x + dot(ld4(ho.Lorentz[0]), rayDir) * hit.dist;
            const float period = ho.flashPeriod;
            const float duration = ho.flashDuration;
            if (event_x - period * __builtin_floorf(event_x / period) < duration) hcolor = hcolor * 2;
        }
        color = hcolor * (a.interval != 0 ? a.ambient : 1.0f);
        if (ho.light) color = color + hcolor;
    }
    if (a.interval == 0) return color;
    for (int i = 0; i < a.object_count; i++) {
        const rpt_object &lo = a.objects[i];
        if (!lo.light) continue;
        bool need = h && i != hit.object;
        f4 hitPos = mk4(0, 0, 0, 0), lightDir = mk4(0, 0, 0, 0);
        f3 lightDir3_ObjFrame = mk3(0, 0, 0);
        float ndotl = 0.0f;
        if (need) {
            const f4 cameraPos_ObjFrame = ld4(ho.stationaryCam);
            const f4 rayDir_ObjFrame = transformPoint4D(ho.Lorentz, rayDir);
            f4 hitPos_ObjFrame = cameraPos_ObjFrame + rayDir_ObjFrame * hit.dist;
            hitPos_ObjFrame = hitPos_ObjFrame + mk4(0, hit.normal.x * 0.001f, hit.normal.y * 0.001f, hit.normal.z * 0.001f);
            hitPos = transformPoint4D(ho.InvLorentz, hitPos_ObjFrame);
            const f4 hitPos_LightFrame = transformPoint4D(lo.Lorentz, hitPos);
            const f3 lightPos3_LightFrame = mk3(lo.M[0].w, lo.M[1].w, lo.M[2].w);
            const f3 lightDir3_LightFrame = lightPos3_LightFrame - yzw(hitPos_LightFrame);
            const f4 lightDir_LightFrame = mk4(a.interval * length(lightDir3_LightFrame), lightDir3_LightFrame.x,
                                               lightDir3_LightFrame.y, lightDir3_LightFrame.z);
            lightDir = transformPoint4D(lo.InvLorentz, lightDir_LightFrame);
            const f4 lightDir_ObjFrame = transformPoint4D(ho.Lorentz, lightDir);
            lightDir3_ObjFrame = yzw(lightDir_ObjFrame);
            const f3 unitLightDir3 = normalize(lightDir3_ObjFrame);
            ndotl = dot(hit.normal, unitLightDir3);
            need = ndotl > 0;
        }
        if (__ballot(need) == 0ull) continue;
        const f3 ld = normalize(yzw(lightDir));
        const f4 shadowDir = mk4((float)a.interval, ld.x, ld.y, ld.z);
        const bool occluded = shadow_uniform<STAGE>(a, wave, hitPos, shadowDir, length(yzw(lightDir)), i, need);
        if (need && !occluded) {
            const float k = ndotl / (1.0f + 0.1f * length(lightDir3_ObjFrame) + 0.01f * dot(lightDir3_ObjFrame, lightDir3_ObjFrame));
            color = color + hcolor * k * ld3(lo.color);
        }
    }
    return color;
}

// intersect_scene (opencl_kernel.cl:361-425) for the wave's 64 pixels in uniform control flow: the closest hit over the
// objects of the wave's mask.
template <bool STAGE>
RPT_DEV void closest_hit_uniform(const KernelArgs &a, int wave, f4 rayDir, unsigned long long object_mask, bool valid, Hit &hit) {
    const float inf = 1e20f;
    hit.dist = inf;
    hit.object = -1;
    hit.normal = mk3(0, 0, 0);
    hit.uv.x = hit.uv.y = 0.0f;
    for (int i = 0; i < a.object_count; i++) {
        if (i < 64 && !((object_mask >> i) & 1ull)) continue;
        const rpt_object &obj = a.objects[i];
        const DObj &pre = a.dobjs[i];
        const f3 d3 = mk3(dot(ld4(obj.Lorentz[1]), rayDir), dot(ld4(obj.Lorentz[2]), rayDir), dot(ld4(obj.Lorentz[3]), rayDir));
        f3 dir = transformDirection(obj.InvM, d3);
        const float scale = length(dir);
        dir = dir / scale;
        const f3 origin = mk3(pre.ox, pre.oy, pre.oz);
        Hit newHit;
        newHit.dist = inf;
        bool got = false;
        switch (obj.type) {
        case RPT_SPHERE: got = sphere_core(obj, -origin, pre.sphere_c, dir, scale, newHit, obj.textureIndex != -1); break;
        case RPT_CUBE: got = cube_core(obj, origin, pre.winding, dir, scale, newHit); break;
        case RPT_MESH: {
            int tri;
            walk_uniform<STAGE>(a, wave, pre.root, origin, dir, valid, newHit.dist, newHit.uv, tri, got);
            if (got) mesh_hit_finish(a, obj, origin, dir, tri, mk3(obj.stationaryCam.y, obj.stationaryCam.z, obj.stationaryCam.w), length(d3), newHit);
            break;
        }
        default: break;
        }
        if (valid && got && newHit.dist < hit.dist) {
            hit = newHit;
            hit.object = i;
        }
    }
}

// the wave's object mask from the rectangles in LDS (wave_object_mask of rpt_kernels.hip.h, same comparisons)
RPT_DEV unsigned long long wave_object_mask_lds(const KernelArgs &a, int tile_x0, int tile_y0) {
    const int lane = threadIdx.x & 63;
    const int n = a.object_count;
    const int slot = (lane < n) ? lane : 0;
    const float4 r = rpt_plds.rects[2 * slot];
    const float iw = a.inv_width, ih = a.inv_height;
    const float tu0 = (((float)tile_x0 - 1.5f) * iw - 0.5f) * a.aspect, tu1 = (((float)tile_x0 + 8.5f) * iw - 0.5f) * a.aspect;
    const float tv0 = ((float)tile_y0 - 1.5f) * ih - 0.5f, tv1 = ((float)tile_y0 + 8.5f) * ih - 0.5f;
    bool outside = (r.z < tu0) | (r.x > tu1) | (r.w < tv0) | (r.y > tv1);
    if (a.diagonals) {
        const float4 g = rpt_plds.rects[2 * slot + 1];
        outside = outside | (g.y < tu0 + tv0) | (g.x > tu1 + tv1) | (g.w < tu0 - tv1) | (g.z > tu1 - tv0);
    }
    const bool keep = (lane < n) & !outside;
    return __ballot(keep);
}

// What a persistent kernel WRITES in global memory arrives as __restrict__ kernel parameters of its own (not through KernelArgs):
// the kernel stores pixels and then goes on reading the scene for its next tile, and only with the stores known not to alias the
// scene do Object[], DObj[] and the mesh roots keep arriving through the scalar cache into SGPRs (a load that a store may have
// clobbered is issued as a vector load: sixteen registers per matrix and a trip through the texture path per wave).
struct Outputs {
    rpt_pixel *out16;
    uint32_t *plane;
    float *debug_rgb;
    unsigned int *claims;        // two sets of RPT_CLAIM_QUEUES counters, RPT_CLAIM_STRIDE words apart
};

RPT_DEV void store_tile_pixel(const KernelArgs &a, const Outputs &o, int x_coord, int y_coord, int local_row, uint32_t packed, bool traced, f3 mapped) {
    const size_t id = (size_t)y_coord * a.width + x_coord;
    if (o.out16) store_pixel(o.out16, id, __float_as_uint((float)x_coord), __float_as_uint((float)y_coord), packed, 0u);
    if (o.plane) __builtin_nontemporal_store(packed, o.plane + (size_t)local_row * a.width + x_coord);
    if (o.debug_rgb) {
        if (!traced) mapped = mk3(a.bg_mapped[0], a.bg_mapped[1], a.bg_mapped[2]);
        o.debug_rgb[3 * id + 0] = mapped.x;
        o.debug_rgb[3 * id + 1] = mapped.y;
        o.debug_rgb[3 * id + 2] = mapped.z;
    }
}

// A/B arm (variant 63): the persistent skeleton around the per-pixel trace of rpt_kernels.hip.h (lanes leave the walk's loops
// instead of carrying predicates; no triangle staging, node records instead of links): what the skeleton alone costs or gains.
RPT_DEV void render_tile_classic(const KernelArgs &a, const Outputs &o, int tile_col, int tile_row) {
    const int lane = threadIdx.x & 63;
    const int global_tile = (tile_row >> a.run_log2) * a.tile_step + a.first_tile + (tile_row & ((1 << a.run_log2) - 1));
    const int x_coord = tile_col * 8 + (lane & 7);
    const int y_coord = global_tile * RPT_TILE_ROWS + (lane >> 3);
    const int local_row = tile_row * RPT_TILE_ROWS + (lane >> 3);
    const unsigned long long object_mask = wave_object_mask_lds(a, tile_col * 8, global_tile * RPT_TILE_ROWS);
    if (x_coord < a.width && y_coord < a.height) {
        uint32_t packed = a.bg_packed;
        f3 mapped = mk3(0.0f, 0.0f, 0.0f), color;
        bool traced = false;
        if (object_mask != 0ull || a.object_count > 64) {
            const f3 camdir = createCamRayDir((float)x_coord, (float)y_coord, a.width, a.height, a.aspect);
            if (trace<20>(a, camdir, object_mask, color)) {
                packed = tonemap_pack(a, color, mapped);
                traced = true;
            }
        }
        store_tile_pixel(a, o, x_coord, y_coord, local_row, packed, traced, mapped);
    }
}

// One 8x8 tile by one wave (render_kernel, opencl_kernel.cl:620-660, for 64 pixels).
template <bool STAGE>
RPT_DEV void render_tile(const KernelArgs &a, const Outputs &o, int wave, int tile_col, int tile_row) {
    const int lane = threadIdx.x & 63;
    const int global_tile = (tile_row >> a.run_log2) * a.tile_step + a.first_tile + (tile_row & ((1 << a.run_log2) - 1));
    const int x_coord = tile_col * 8 + (lane & 7);
    const int y_coord = global_tile * RPT_TILE_ROWS + (lane >> 3);
    const int local_row = tile_row * RPT_TILE_ROWS + (lane >> 3);
    const bool valid = x_coord < a.width && y_coord < a.height;
    const unsigned long long object_mask = wave_object_mask_lds(a, tile_col * 8, global_tile * RPT_TILE_ROWS);
    uint32_t packed = a.bg_packed;
    f3 mapped = mk3(0.0f, 0.0f, 0.0f);
    bool traced = false;
    if (object_mask != 0ull || a.object_count > 64) {
        const f3 camdir = createCamRayDir((float)x_coord, (float)y_coord, a.width, a.height, a.aspect);
        const f3 nd = normalize(camdir);
        const f4 rayDir = mk4((float)a.interval, nd.x, nd.y, nd.z);
        Hit hit;
        closest_hit_uniform<STAGE>(a, wave, rayDir, object_mask, valid, hit);
        const bool h = hit.object >= 0;
        if (__ballot(h) != 0ull) {
            const f3 color = shade_uniform<STAGE>(a, wave, rayDir, hit, h);
            if (h) {
                packed = tonemap_pack(a, color, mapped);
                traced = true;
            }
        }
    }
    if (valid) store_tile_pixel(a, o, x_coord, y_coord, local_row, packed, traced, mapped);
}

// ---- work distribution ---------------------------------------------------------------------------------------------------
// One counter for the whole frame does not scale: atomics on ONE address retire at about one per 15-20 ns on this chip
// (measured: the first form of this kernel, one claim per tile, spent 3.4 ms on the 227 000 claims of an 8K frame), and a wave
// count of 5 120 makes even one claim per wave and frame a 0.1 ms affair.  So:
//   * the band's tiles (the meshes' tile rows: where the long walks are) are dealt round-robin to RPT_CLAIM_QUEUES queues, each
//     with a counter on a cache line of its own, each served by the waves whose index is congruent to it: dynamic (a wave that
//     draws a long walk simply draws fewer tiles), contention spread over 64 addresses, no counter sees more than a few
//     hundred atomics per frame;
//   * everything else (sky: uniform cost) is dealt statically, run r of RPT_SKY_RUN tiles to wave r mod waves, after the band;
//   * the counters live twice: frame k counts in set k & 1 and its first workgroup zeroes the other set for frame k + 1 (which the
//     stream orders after this launch), so nobody has to find out who leaves last.
#define RPT_CLAIM_QUEUES 64
#define RPT_CLAIM_STRIDE 64      /* unsigned ints between two counters: 256 B */

RPT_DEV unsigned int claim_issue(const KernelArgs &a, const Outputs &o, int queue) {      // lane 0's VGPR holds the claim once the atomic is back
    unsigned int c = 0;
    if ((threadIdx.x & 63) == 0) c = atomicAdd(&o.claims[(a.claim_set * RPT_CLAIM_QUEUES + queue) * RPT_CLAIM_STRIDE], 1u);
    return c;
}

RPT_DEV void decode_band_tile(const KernelArgs &a, unsigned int t, int &tile_col, int &tile_row) {
    const unsigned int q = __umulhi(t, a.tiles_x_magic);
    tile_col = (int)(t - q * (unsigned int)a.tiles_x);
    tile_row = a.first_ty + (int)q;
}

RPT_DEV void decode_sky_run(const KernelArgs &a, unsigned int r, int &tile_col, int &tile_row, int &count) {
    const unsigned int q = __umulhi(r, a.runs_x_magic);
    tile_col = (int)(r - q * (unsigned int)a.runs_x) * RPT_SKY_RUN;
    tile_row = (int)q < a.first_ty ? (int)q : (int)q + a.first_h;
    count = a.tiles_x - tile_col < RPT_SKY_RUN ? a.tiles_x - tile_col : RPT_SKY_RUN;
}

RPT_DEV void stage_workgroup_lds(const KernelArgs &a, const Outputs &o) {
    const int n_rect = 2 * (a.object_count < 64 ? a.object_count : 64);
    for (int i = threadIdx.x; i < (n_rect > 2 ? n_rect : 2); i += 256) rpt_plds.rects[i] = a.rects[i < n_rect ? i : 0];   // (the buffer behind rects holds at least one record)
    for (int i = threadIdx.x; i < a.top_count; i += 256) rpt_plds.top_links[i] = a.links[i];
    if (blockIdx.x == 0 && threadIdx.x < RPT_CLAIM_QUEUES) o.claims[((a.claim_set ^ 1) * RPT_CLAIM_QUEUES + threadIdx.x) * RPT_CLAIM_STRIDE] = 0u;
    __syncthreads();
}

typedef const __attribute__((address_space(4))) KernelArgs *KernargPtr;

template <int MODE>
RPT_DEV void persistent_body(const KernelArgs &a0, const Outputs &o) {
    stage_workgroup_lds(a0, o);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned int g = blockIdx.x * 4u + (unsigned int)wave, n_waves = gridDim.x * 4u;
    const int queue = (int)(g % RPT_CLAIM_QUEUES);
    // ONE loop over the wave's tiles, band first, so that render_tile (and the two octree walks inside it) exists once in the
    // kernel's code.  Everything here is wave-uniform (SGPRs).
    bool in_band = a0.band_tiles > 0;
    unsigned int claim = 0, next_v = 0;
    if (in_band) claim = __builtin_amdgcn_readfirstlane(claim_issue(a0, o, queue));
    unsigned int run = g;
    int in_run = 0;
    for (;;) {
        // The arguments are read from the kernel-argument segment again for every tile, as a freshly launched wave would: hoisted out
        // of this loop (they are loop-invariant, and the compiler knows it) the sixty-odd scalars would have to stay live across
        // everything a tile does and end up spilled to vector-register lanes and read back per tile — measured: +116 instructions per
        // tile.  The empty asm makes the pointer opaque per iteration; the segment is constant address space, so the reads stay s_loads.
        KernargPtr kp = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        const KernelArgs &a = *(const KernelArgs *)kp;
        int tile_col, tile_row;
        if (in_band) {
            const unsigned int t = claim * RPT_CLAIM_QUEUES + (unsigned int)queue;
            if (t >= (unsigned int)a.band_tiles) { in_band = false; continue; }
            next_v = claim_issue(a, o, queue);            // on its way while this tile is rendered
            decode_band_tile(a, t, tile_col, tile_row);
        } else {
            if (run >= (unsigned int)a.sky_runs) break;
            int count;
            decode_sky_run(a, run, tile_col, tile_row, count);
            tile_col += in_run;
            if (++in_run >= count) { in_run = 0; run += n_waves; }
        }
        if (MODE == 2) render_tile_classic(a, o, tile_col, tile_row);
        else render_tile<MODE == 0>(a, o, wave, tile_col, tile_row);
        if (in_band) claim = __builtin_amdgcn_readfirstlane(next_v);
    }
}

#define RPT_PERSISTENT_PARAMS const KernelArgs a, rpt_pixel *__restrict__ out16, uint32_t *__restrict__ plane, float *__restrict__ debug_rgb, unsigned int *__restrict__ claims
#define RPT_PERSISTENT_OUTPUTS Outputs o; o.out16 = out16; o.plane = plane; o.debug_rgb = debug_rgb; o.claims = claims
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void rpt_render_kernel_persistent_w5(RPT_PERSISTENT_PARAMS) { RPT_PERSISTENT_OUTPUTS; persistent_body<0>(a, o); }          // 60
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void rpt_render_kernel_persistent_direct_w5(RPT_PERSISTENT_PARAMS) { RPT_PERSISTENT_OUTPUTS; persistent_body<1>(a, o); }  // 62: A/B arm, triangle records straight from global memory

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void rpt_render_kernel_persistent_classic_w5(RPT_PERSISTENT_PARAMS) { RPT_PERSISTENT_OUTPUTS; persistent_body<2>(a, o); }  // 63: A/B arm

// ================================================================================================================
// The per-workgroup ray queue (variant 61).  A workgroup owns a 32x8-pixel strip, one 8x8 tile per wave, launched
// like the default kernel.  Whatever needs an octree walk — a primary ray that enters a mesh's root box, a shadow ray that has to
// be tested against a mesh — is not walked by the pixel's own lane: the lane pushes {object-space ray, pixel slot} into the
// workgroup's LDS queue (__ballot + popcount prefix inside the wave, one ds_add per wave for the base), and after a barrier the
// workgroup's waves pull 64 entries at a time and walk them with full lanes (walk_uniform), write {distance, barycentrics,
// triangle} to the pixel's result slot, and after a second barrier every pixel continues with its own result.  Per-ray
// arithmetic is the same functions with the same operands: bit-identical by construction.  Barriers sit only in control flow
// that is uniform over the WORKGROUP (loops over objects and lights, the queue's count), never behind a per-wave condition.
RPT_DEV int queue_push_slot(bool push) {         // the entry index for this lane's ray, -1 if it does not push
    const unsigned long long m = __ballot(push);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (m != 0ull) {
        if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&rpt_qlds.count, __popcll(m));
        base = __builtin_amdgcn_readlane(base, __ffsll((long long)m) - 1);
    }
    return push ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}

// walk every queued ray of object `root` (all waves; between two barriers)
RPT_DEV void queue_drain(const KernelArgs &a, int wave, int root) {
    const int lane = threadIdx.x & 63;
    const int n = rpt_qlds.count;
    for (int first = wave * 64; first < n; first += 256) {
        const int e = first + lane;
        const bool alive = e < n;
        const QueuedRay r = rpt_qlds.rays[alive ? e : first];
        float dist = 1e20f;
        f2 uv;
        uv.x = uv.y = 0.0f;
        int tri;
        bool got;
        walk_uniform<false, false>(a, wave, root, mk3(r.ox, r.oy, r.oz), mk3(r.dx, r.dy, r.dz), alive, dist, uv, tri, got);
        if (alive) {
            QueuedResult q;
            q.dist = dist; q.u = uv.x; q.v = uv.y; q.tri = got ? tri : -1;
            rpt_qlds.results[r.pixel] = q;
        }
    }
}

// one mesh object for the whole strip: push, drain, read back.  `want`: this lane has a ray for the object.
RPT_DEV bool queue_round(const KernelArgs &a, int wave, int root, bool want, f3 origin, f3 dir, float &dist, f2 &uv, int &tri) {
    const int slot = threadIdx.x;
    if (threadIdx.x == 0) rpt_qlds.count = 0;
    rpt_qlds.results[slot].tri = -1;
    __syncthreads();
    // only rays that enter the root box are queued (the walk's own first test, opencl_kernel.cl:219: repeated by the walker)
    if (want) {
        const DNode &rn = a.dnodes[root];
        Ray ray;
        ray.origin = origin;
        ray.dir = dir;
        f2 d;
        int cs, fs;
        want = intersect_AABB(mk3(rn.minx, rn.miny, rn.minz), mk3(rn.maxx, rn.maxy, rn.maxz), ray, d, cs, fs);
    }
    const int e = queue_push_slot(want);
    if (e >= 0) {
        QueuedRay r;
        r.dx = dir.x; r.dy = dir.y; r.dz = dir.z;
        r.ox = origin.x; r.oy = origin.y; r.oz = origin.z;
        r.pixel = slot;
        r.pad = 0;
        rpt_qlds.rays[e] = r;
    }
    __syncthreads();
    if (rpt_qlds.count == 0) return false;                      // (workgroup-uniform: nobody asked; one barrier less)
    queue_drain(a, wave, root);
    __syncthreads();
    const QueuedResult q = rpt_qlds.results[slot];
    dist = q.dist;
    uv.x = q.u;
    uv.y = q.v;
    tri = q.tri;
    return q.tri >= 0;
}

RPT_DEV void render_strip_queued(const KernelArgs &a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tile_row = (int)blockIdx.y, strip = (int)blockIdx.x;
    const int global_tile = (tile_row >> a.run_log2) * a.tile_step + a.first_tile + (tile_row & ((1 << a.run_log2) - 1));
    const int x_coord = strip * 32 + wave * 8 + (lane & 7);
    const int y_coord = global_tile * RPT_TILE_ROWS + (lane >> 3);
    const int local_row = tile_row * RPT_TILE_ROWS + (lane >> 3);
    const bool valid = x_coord < a.width && y_coord < a.height;
    const unsigned long long object_mask = wave_object_mask(a, strip * 32 + wave * 8, global_tile * RPT_TILE_ROWS);
    // the strip's mask, formed the same way by every wave of the workgroup from the strip's own bounds (a rectangle that misses the
    // 32x8 strip grown by 1.5 pixels misses each of its tiles grown by 1.5 pixels): what decides whether a barrier round happens
    unsigned long long strip_mask;
    {
        const int n = a.object_count;
        const int slot = (lane < n) ? lane : 0;
        const float4 r = a.rects[2 * slot];
        const float tu0 = (((float)(strip * 32) - 1.5f) * a.inv_width - 0.5f) * a.aspect, tu1 = (((float)(strip * 32) + 32.5f) * a.inv_width - 0.5f) * a.aspect;
        const float tv0 = ((float)(global_tile * RPT_TILE_ROWS) - 1.5f) * a.inv_height - 0.5f, tv1 = ((float)(global_tile * RPT_TILE_ROWS) + 8.5f) * a.inv_height - 0.5f;
        const bool outside = (r.z < tu0) | (r.x > tu1) | (r.w < tv0) | (r.y > tv1);
        strip_mask = __ballot((lane < n) & !outside);
    }
    uint32_t packed = a.bg_packed;
    f3 mapped = mk3(0.0f, 0.0f, 0.0f);
    bool traced = false;
    if (strip_mask == 0ull && a.object_count <= 64) {           // workgroup-uniform: nothing can be hit anywhere in the strip
        if (valid) {
            Outputs o;
            o.out16 = a.out16; o.plane = a.plane; o.debug_rgb = a.debug_rgb; o.claims = nullptr;
            store_tile_pixel(a, o, x_coord, y_coord, local_row, packed, traced, mapped);
        }
        return;
    }
    const f3 camdir = createCamRayDir((float)x_coord, (float)y_coord, a.width, a.height, a.aspect);
    const f3 nd = normalize(camdir);
    const f4 rayDir = mk4((float)a.interval, nd.x, nd.y, nd.z);
    const float inf = 1e20f;
    Hit hit;
    hit.dist = inf;
    hit.object = -1;
    hit.normal = mk3(0, 0, 0);
    hit.uv.x = hit.uv.y = 0.0f;
    // ---- closest hit (opencl_kernel.cl:361-425), objects in their order
    for (int i = 0; i < a.object_count; i++) {
        const rpt_object &obj = a.objects[i];
        const bool in_strip = i >= 64 || ((strip_mask >> i) & 1ull);
        const bool in_tile = i >= 64 || ((object_mask >> i) & 1ull);
        if (obj.type != RPT_MESH ? !in_tile : !in_strip) continue;          // wave-uniform for analytic objects, workgroup-uniform for meshes
        const DObj &pre = a.dobjs[i];
        const f3 d3 = mk3(dot(ld4(obj.Lorentz[1]), rayDir), dot(ld4(obj.Lorentz[2]), rayDir), dot(ld4(obj.Lorentz[3]), rayDir));
        f3 dir = transformDirection(obj.InvM, d3);
        const float scale = length(dir);
        dir = dir / scale;
        const f3 origin = mk3(pre.ox, pre.oy, pre.oz);
        Hit newHit;
        newHit.dist = inf;
        bool got = false;
        switch (obj.type) {
        case RPT_SPHERE: got = sphere_core(obj, -origin, pre.sphere_c, dir, scale, newHit, obj.textureIndex != -1); break;
        case RPT_CUBE: got = cube_core(obj, origin, pre.winding, dir, scale, newHit); break;
        case RPT_MESH: {
            int tri;
            got = queue_round(a, wave, pre.root, valid && in_tile, origin, dir, newHit.dist, newHit.uv, tri);
            if (got) mesh_hit_finish(a, obj, origin, dir, tri, mk3(obj.stationaryCam.y, obj.stationaryCam.z, obj.stationaryCam.w), length(d3), newHit);
            break;
        }
        default: break;
        }
        if (valid && got && newHit.dist < hit.dist) {
            hit = newHit;
            hit.object = i;
        }
    }
    // ---- shading (opencl_kernel.cl:548-604)
    const bool h = hit.object >= 0;
    const int ho_i = h ? hit.object : 0;
    const rpt_object &ho = a.objects[ho_i];
    f3 hcolor = mk3(0.0f, 0.0f, 0.0f), color = mk3(0.0f, 0.0f, 0.0f);
    if (h) {
        hcolor = ho.textureIndex != -1 ? sample_texture(a, ho, hit.uv) : ld3(ho.color);
        if (ho.flashPeriod > 0) {
            const float event_x = ho.stationaryCam.x + dot(ld4(ho.Lorentz[0]), rayDir) * hit.dist;
            const float period = ho.flashPeriod;
            const float duration = ho.flashDuration;
            if (event_x - period * __builtin_floorf(event_x / period) < duration) hcolor = hcolor * 2;
        }
        color = hcolor * (a.interval != 0 ? a.ambient : 1.0f);
        if (ho.light) color = color + hcolor;
    }
    if (a.interval != 0) {
        for (int i = 0; i < a.object_count; i++) {
            const rpt_object &lo = a.objects[i];
            if (!lo.light) continue;                       // uniform
            bool need = h && i != hit.object;
            f4 hitPos = mk4(0, 0, 0, 0), lightDir = mk4(0, 0, 0, 1);
            f3 lightDir3_ObjFrame = mk3(0, 0, 0);
            float ndotl = 0.0f;
            if (need) {
                const f4 cameraPos_ObjFrame = ld4(ho.stationaryCam);
                const f4 rayDir_ObjFrame = transformPoint4D(ho.Lorentz, rayDir);
                f4 hitPos_ObjFrame = cameraPos_ObjFrame + rayDir_ObjFrame * hit.dist;
                hitPos_ObjFrame = hitPos_ObjFrame + mk4(0, hit.normal.x * 0.001f, hit.normal.y * 0.001f, hit.normal.z * 0.001f);
                hitPos = transformPoint4D(ho.InvLorentz, hitPos_ObjFrame);
                const f4 hitPos_LightFrame = transformPoint4D(lo.Lorentz, hitPos);
                const f3 lightPos3_LightFrame = mk3(lo.M[0].w, lo.M[1].w, lo.M[2].w);
                const f3 lightDir3_LightFrame = lightPos3_LightFrame - yzw(hitPos_LightFrame);
                const f4 lightDir_LightFrame = mk4(a.interval * length(lightDir3_LightFrame), lightDir3_LightFrame.x,
                                                   lightDir3_LightFrame.y, lightDir3_LightFrame.z);
                lightDir = transformPoint4D(lo.InvLorentz, lightDir_LightFrame);
                const f4 lightDir_ObjFrame = transformPoint4D(ho.Lorentz, lightDir);
                lightDir3_ObjFrame = yzw(lightDir_ObjFrame);
                const f3 unitLightDir3 = normalize(lightDir3_ObjFrame);
                ndotl = dot(hit.normal, unitLightDir3);
                need = ndotl > 0;
            }
            // sample_light (opencl_kernel.cl:488-545): occluders in their order; a mesh occluder is one barrier round
            const f3 ldn = normalize(yzw(lightDir));
            const f4 lightDir0 = mk4((float)a.interval, ldn.x, ldn.y, ldn.z);
            const float lightDist = length(yzw(lightDir));
            bool occluded = false;
            for (int j = 0; j < a.object_count; j++) {
                if (j == i) continue;
                const rpt_object &obj = a.objects[j];
                const bool act = need && !occluded;
                if (obj.type != RPT_MESH && __ballot(act) == 0ull) continue;       // (a mesh round has barriers: every wave goes through it)
                const f4 ev = transformPoint4D(obj.Lorentz, hitPos);
                const f4 ld = transformPoint4D(obj.Lorentz, lightDir0);
                const f3 origin = transformPoint(obj.InvM, yzw(ev));
                f3 dir = transformDirection(obj.InvM, yzw(ld));
                bool want = act;
                if (obj.type != RPT_MESH) {
                    if (__ballot(act && !(unit_segment_apart(origin, dir, lightDist) && lightDist > 0.0f)) == 0ull) continue;
                } else if (a.dobjs[j].mh[0] >= 0.0f) {
                    // per LANE here: a lane whose ray or segment cannot reach the box does not queue a ray (the wave-wide form of this
                    // cull in intersect_object skips only when no lane can)
                    bool idle = mesh_ray_misses_root(a.dobjs[j], origin, dir);
                    if (a.dobjs[j].mslope >= 0.0f) idle = idle | (mesh_segment_apart(a.dobjs[j], yzw(ev), origin, dir, lightDist) && lightDist > 0.0f);
                    want = act && !idle;
                }
                const float scale = length(dir);
                dir = dir / scale;
                Hit nh;
                nh.dist = 1e20f;
                bool got = false;
                switch (obj.type) {
                case RPT_SPHERE: {
                    const f3 rayToSphere = -origin;
                    if (act) got = sphere_core(obj, rayToSphere, dot(rayToSphere, rayToSphere) - 1.0f, dir, scale, nh, obj.textureIndex != -1);
                    break;
                }
                case RPT_CUBE:
                    if (act) got = cube_core(obj, origin, cube_winding(origin), dir, scale, nh);
                    break;
                case RPT_MESH: {
                    int tri;
                    got = queue_round(a, wave, a.dobjs[j].root, want, origin, dir, nh.dist, nh.uv, tri);
                    if (got) mesh_hit_finish(a, obj, origin, dir, tri, yzw(ev), length(yzw(ld)), nh);
                    break;
                }
                default: break;
                }
                if (act && got && nh.dist < lightDist) occluded = true;
            }
            if (need && !occluded) {
                const float k = ndotl / (1.0f + 0.1f * length(lightDir3_ObjFrame) + 0.01f * dot(lightDir3_ObjFrame, lightDir3_ObjFrame));
                color = color + hcolor * k * ld3(lo.color);
            }
        }
    }
    if (h) {
        packed = tonemap_pack(a, color, mapped);
        traced = true;
    }
    if (valid) {
        Outputs o;
        o.out16 = a.out16;
        o.plane = a.plane;
        o.debug_rgb = a.debug_rgb;
        o.claims = nullptr;
        store_tile_pixel(a, o, x_coord, y_coord, local_row, packed, traced, mapped);
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void rpt_render_kernel_queue_w5(const KernelArgs a) { render_strip_queued(a); }   // 61

}  // namespace rptd
#endif  /* !RPT_RELAXED_FP */
