// rpt_diag_walks.hip.h — librpt_hip_diag.so only (make diag; included by rpt_kernels.hip.h under RPT_DIAGNOSTICS).
// The walks that are NOT the product: round 2's walk (node-policy form) with its loop counters and per-wave cycle accounting
// (tools/divergence.py, tools/timeline.py), kept as the A/B baseline of round 3, and the experiment arms of round 3 around the
// product walk (flags below).  Every one of them computes what octree_walk computes; tests/test_gpu_diag_arms.py checks that.
#pragma once

namespace rptd {

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
    RPT_DEV int tri_begin(const KernelArgs &) const { return __float_as_int(hi.w) & RPT_NODE_BEGIN_MASK; }
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

template <int V>
RPT_DEV bool octree_core_diag(const KernelArgs &a, const rpt_object &obj, int root, const Ray &newRay, f3 world_origin,
                         float world_dirlen, Hit &hit) {
    NodeRef<1> node;
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
            const int childIndex = octree_child_step_fast(uv);
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
            const int childIndex = octree_child_step_fast(uv);
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
        currOctreeIndex = a.dnodes[currOctreeIndex].nb[farSide];
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

// opencl_kernel.cl:106-126 with two certain misses found WITHOUT the division (and the division, eleven dependent instructions,
// skipped for the wave when no lane is left): with a = dot(tvec, pvec), uv.x = a * (1 / det) in the reference.
//   * a and det of opposite sign, |a| >= 2^-60, |det| <= 2^60: 1/det is a normal number of det's sign and at least 2^-61 in
//     magnitude, the product is at least 2^-121 in magnitude — no underflow to a signed zero — and negative: uv.x < 0, a miss;
//   * same sign and |a| >= |det| * (1 + 2^-19) (the float product, itself within 2^-24 of exact): a / det >= 1 + 2^-20, and the
//     two roundings of a * fl(1/det) lose at most 2^-23 of that: uv.x > 1, a miss.
// NaNs fail every comparison and take the general path, as does everything in between.
RPT_DEV bool intersect_triangle_edges_prereject(f3 A, f3 v0v1, f3 v0v2, const Ray &ray, float &dist, f2 &uv) {
    const f3 pvec = cross(ray.dir, v0v2);
    const float det = dot(v0v1, pvec);
    if (det < RPT_EPSILON && -RPT_EPSILON < det) return false;
    const f3 tvec = ray.origin - A;
    const float a = dot(tvec, pvec);
    const bool opposite = ((__float_as_uint(a) ^ __float_as_uint(det)) >> 31) != 0u;
    const float fa = __builtin_fabsf(a), fd = __builtin_fabsf(det);
    const bool below = opposite && fa >= 8.67361737988403547e-19f && fd <= 1.152921504606846976e18f;
    const bool above = !opposite && fa >= fd * 1.0000019073486328125f;
    const bool certain_miss = below || above;
    if (__ballot(!certain_miss) == 0ull) return false;
    if (certain_miss) return false;
    const float invDet = 1 / det;
    uv.x = a * invDet;
    if (uv.x < 0 || uv.x > 1) return false;
    const f3 qvec = cross(tvec, v0v1);
    uv.y = dot(ray.dir, qvec) * invDet;
    if (uv.y < 0 || uv.x + uv.y > 1) return false;
    dist = dot(v0v2, qvec) * invDet;
    return true;
}

template <bool PREREJECT>
RPT_DEV void test_tri_rec_x(const TriRec &r, const Ray &ray, Hit &hit, int &hitTri, bool &didHit) {
    float dist;
    f2 triUV;
    if (PREREJECT ? intersect_triangle_edges_prereject(mk3(r.t0.x, r.t0.y, r.t0.z), mk3(r.t0.w, r.t1.x, r.t1.y), mk3(r.t1.z, r.t1.w, r.e2z), ray, dist, triUV)
                  : intersect_triangle_edges(mk3(r.t0.x, r.t0.y, r.t0.z), mk3(r.t0.w, r.t1.x, r.t1.y), mk3(r.t1.z, r.t1.w, r.e2z), ray, dist, triUV)) {
        if (0 <= dist && dist < hit.dist) {
            hitTri = r.tri;
            hit.dist = dist;
            hit.uv = triUV;
            didHit = true;
        }
    }
}

// The product walk with its round trips re-ordered further; flags (kernel variant = 256 + flags, + 8 for the mesh band first):
//   1: exit face and neighbour index before the triangle loop            (adopted: part of octree_walk)
//   2: the neighbour's RECORD asked for after the first triangle          (lost: nine more live registers, 40 B of scratch)
//   4: triangle records one iteration ahead                              (adopted for the blocking call's kernel)
//  16: descents read the compact link array and stop at leaf-flagged children (adopted)
//  32: s_setprio raised with the length of the walk                      (lost: -1..-12 %)
//  64: division-free pre-rejection of certain misses + __ballot          (lost: -3..-5 %; exactness: rpt_probe which = 6)
// 128: two triangle records per round trip                              (like 4 for latency, lost for throughput)
template <int F>
RPT_DEV bool octree_walk_x(const KernelArgs &a, const rpt_object &obj, int root, const Ray &newRay, f3 world_origin,
                            float world_dirlen, Hit &hit) {
    int curr = root;
    NodeRec rec = load_node_rec<false>(a, curr);
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
            rec = load_node_rec<false>(a, curr);
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
        if (F & 32) {       // a long walk is what a blocking frame waits for: let it win the SIMD's issue arbitration
            if (steps == 10) __builtin_amdgcn_s_setprio(1);
            if (steps == 24) __builtin_amdgcn_s_setprio(2);
        }
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
            rec = load_node_rec<false>(a, curr);
            nmin = mk3(rec.lo.x, rec.lo.y, rec.lo.z);
            nmax = mk3(rec.hi.x, rec.hi.y, rec.hi.z);
        }
        int i = __float_as_int(rec.hi.w) & RPT_NODE_BEGIN_MASK;
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
                    test_tri_rec_x<(F & 64) != 0>(cur, newRay, hit, hitTri, didHit);
                    if ((F & 2) && !fetched) { if (next != -1) nrec = load_node_rec<false>(a, next); fetched = true; }
                    cur = nxt;
                }
            } else if (F & 2) {
                if (next != -1) nrec = load_node_rec<false>(a, next);
            }
        } else if (F & 128) {     // two records per round trip
            for (; i + 1 < trisEnd; i += 2) {
                const TriRec r0 = load_tri_rec(a, i), r1 = load_tri_rec(a, i + 1);
                test_tri_rec_x<(F & 64) != 0>(r0, newRay, hit, hitTri, didHit);
                test_tri_rec_x<(F & 64) != 0>(r1, newRay, hit, hitTri, didHit);
            }
            if (i < trisEnd) test_tri_rec_x<(F & 64) != 0>(load_tri_rec(a, i), newRay, hit, hitTri, didHit);
        } else {
            if (i < trisEnd) {
                test_tri_rec_x<(F & 64) != 0>(load_tri_rec(a, i), newRay, hit, hitTri, didHit);
                i++;
            }
            if (F & 2) { if (next != -1) nrec = load_node_rec<false>(a, next); }
            for (; i < trisEnd; i++) test_tri_rec_x<(F & 64) != 0>(load_tri_rec(a, i), newRay, hit, hitTri, didHit);
        }
        uv = nmin + uv * (nmax - nmin);
        const bool stop = exit_is_past_hit(uv - newRay.origin, hit.dist, didHit);
        if (stop || next == -1) break;
        curr = next;
        rec = (F & 2) ? nrec : load_node_rec<false>(a, curr);
    }
    if (F & 32) __builtin_amdgcn_s_setprio(0);
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

// Experiment (variants 529 / 541): the six neighbour indices come with the node record (the whole 64-B line in one round trip)
// and the one the ray leaves through is picked BEFORE the triangle loop (the exit face is known by then), so an empty leaf costs
// one round trip instead of two and no neighbour index is ever loaded on its own.
struct NodeRecN { v4f lo, hi; v4i q2, q3; };
RPT_DEV NodeRecN load_node_rec_n(const KernelArgs &a, int i) {
    const v4f *p = reinterpret_cast<const v4f *>(a.dnodes + i);
    NodeRecN r;
    r.lo = p[0];
    r.hi = p[1];
    r.q2 = reinterpret_cast<const v4i *>(p)[2];
    r.q3 = reinterpret_cast<const v4i *>(p)[3];
    return r;
}
template <bool PIPELINE, bool FIRST = false>
RPT_DEV bool octree_walk_nbrec(const KernelArgs &a, const rpt_object &obj, int root, const Ray &newRay, f3 world_origin,
                               float world_dirlen, Hit &hit) {
    int curr = root;
    NodeRecN rec = load_node_rec_n(a, curr);
    f2 d;
    int closeSide, farSide;
    f3 nmin = mk3(rec.lo.x, rec.lo.y, rec.lo.z), nmax = mk3(rec.hi.x, rec.hi.y, rec.hi.z);
    if (!intersect_AABB(nmin, nmax, newRay, d, closeSide, farSide)) return false;
    f3 uv = newRay.origin + newRay.dir * d.x;
    if (d.x < 0) {
        uv = (newRay.origin - nmin) / (nmax - nmin);
        if (__float_as_int(rec.lo.w) != -1) {
            curr = descend_to_leaf(a, __float_as_int(rec.lo.w), uv);
            rec = load_node_rec_n(a, curr);
        }
        nmin = mk3(rec.lo.x, rec.lo.y, rec.lo.z);
        nmax = mk3(rec.hi.x, rec.hi.y, rec.hi.z);
        if (!intersect_AABB(nmin, nmax, newRay, d, closeSide, farSide)) return false;
        uv = newRay.origin + newRay.dir * d.x;
    }
    const ExitPlan plan = makeExitPlan(normalize(newRay.dir / (nmax - nmin)));
    bool didHit = false;
    int hitTri = 0;
    TriRec first;
    if (FIRST) first = load_first_tri(a, curr);
    for (int steps = 1; steps <= RPT_MAX_LEAF_STEPS; steps++) {
        nmin = mk3(rec.lo.x, rec.lo.y, rec.lo.z);
        nmax = mk3(rec.hi.x, rec.hi.y, rec.hi.z);
        uv = (uv - nmin) / (nmax - nmin);
        if (__float_as_int(rec.lo.w) != -1) {
            curr = descend_to_leaf(a, __float_as_int(rec.lo.w), uv);
            rec = load_node_rec_n(a, curr);
            if (FIRST) first = load_first_tri(a, curr);
            nmin = mk3(rec.lo.x, rec.lo.y, rec.lo.z);
            nmax = mk3(rec.hi.x, rec.hi.y, rec.hi.z);
        }
        int i = __float_as_int(rec.hi.w) & RPT_NODE_BEGIN_MASK;
        const int trisEnd = i + rec.q2.x;
        farSide = getOppositeBoxSide(plan, uv);
        int next = rec.q2.y;
        next = farSide == 1 ? rec.q2.z : next;
        next = farSide == 2 ? rec.q2.w : next;
        next = farSide == 3 ? rec.q3.x : next;
        next = farSide == 4 ? rec.q3.y : next;
        next = farSide == 5 ? rec.q3.z : next;
        if (PIPELINE) {
            if (i < trisEnd) {
                TriRec cur = FIRST ? first : load_tri_rec(a, i);
                for (; i < trisEnd; i++) {
                    TriRec nxt = cur;
                    if (i + 1 < trisEnd) nxt = load_tri_rec(a, i + 1);
                    test_tri_rec(cur, newRay, hit, hitTri, didHit);
                    cur = nxt;
                }
            }
        } else {
            for (; i < trisEnd; i++) test_tri_rec(load_tri_rec(a, i), newRay, hit, hitTri, didHit);
        }
        uv = nmin + uv * (nmax - nmin);
        if (exit_is_past_hit(uv - newRay.origin, hit.dist, didHit) || next == -1) break;
        curr = next;
        rec = load_node_rec_n(a, curr);
        if (FIRST) first = load_first_tri(a, curr);
    }
    if (!didHit) return false;
    mesh_hit_finish(a, obj, newRay.origin, newRay.dir, hitTri, world_origin, world_dirlen, hit);
    return true;
}

// Variants 561 / 573: the A/B arms of the product's latency walk (octree_walk<true, true>: a leaf's first triangle record is
// addressable by the node's index and asked for with the node record) in natural order / mesh band first, against 273 / 285.
// per-wave timeline (V == 4): start / end of the wave on the 100 MHz wall clock + the loop accounting of rpt_diag_lds
struct DiagWaveClock { unsigned long long t_start; };
template <int V>
RPT_DEV DiagWaveClock diag_wave_begin() {
    DiagWaveClock c;
    c.t_start = 0;
    if (V == 4) {
        c.t_start = wall_clock64();
        if ((threadIdx.x & 63) < 8) rpt_diag_lds[threadIdx.x >> 6][threadIdx.x & 63] = 0;
        rpt_diag_lds[threadIdx.x >> 6][6] = clock64();
    }
    return c;
}
template <int V>
RPT_DEV void diag_wave_end(const KernelArgs &a, DiagWaveClock c) {
    if (V == 4 && a.wave_times) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const unsigned long long t_end = wall_clock64();
        const unsigned long long m = __ballot(1);
        if (lane == __ffsll((long long)m) - 1) {
            const size_t w = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave;
            a.wave_times[10 * w] = c.t_start;
            a.wave_times[10 * w + 1] = t_end;
            rpt_diag_lds[wave][7] = clock64();
            for (int q = 0; q < 8; q++) a.wave_times[10 * w + 2 + q] = rpt_diag_lds[wave][q];
        }
    }
}

template <int V> RPT_DEV constexpr bool diag_walk_selected() { return V == 2 || V == 4 || V == 5 || V == 10 || V == 120 || V == 121 || V == 122 || V == 123 || V >= 256; }
template <int V>
RPT_DEV bool diag_walk(const KernelArgs &a, const rpt_object &obj, int root, const Ray &newRay, f3 world_origin, float world_dirlen, Hit &hit) {
    if (V == 561 || V == 573) return octree_walk<true, true, false, false, true>(a, obj, root, newRay, world_origin, world_dirlen, hit);    // = kernel 43's walk
    if (V == 605) return octree_walk<true, true, false, true, true>(a, obj, root, newRay, world_origin, world_dirlen, hit);           // 573 WITH the root table (descend_from_root; lost)
    if (V == 593) return octree_walk<false, false, true, true, true>(a, obj, root, newRay, world_origin, world_dirlen, hit);          // kernel 41's walk WITH the root table (level)
    if (V == 589) return octree_walk<true, true, true, false, true>(a, obj, root, newRay, world_origin, world_dirlen, hit);                // 573 WITH the packed leaf count (lost: r03_packed_count_ab.txt)
    if (V == 625) return octree_walk<false, false, true, false, false>(a, obj, root, newRay, world_origin, world_dirlen, hit);    // kernel 41's walk with the triangle id read with EVERY record (before LATE_ID), natural order
    if (V == 637) return octree_walk<true, true, false, false, false>(a, obj, root, newRay, world_origin, world_dirlen, hit);     // kernel 43's walk likewise, mesh band first
    if (V == 641) return octree_walk<false, false, true, false, true>(a, obj, root, newRay, world_origin, world_dirlen, hit);     // kernel 41's walk; the LANES of the wave follow the Z curve (render_pixel_body)
    if (V == 653) return octree_walk<true, true, false, false, true>(a, obj, root, newRay, world_origin, world_dirlen, hit);      // kernel 43's walk, likewise, mesh band first
    if (V == 673) return octree_walk<false, false, true, false, true, 2>(a, obj, root, newRay, world_origin, world_dirlen, hit);        // likewise with ONE lane's vector loads + readfirstlane (UNIFORM = 2)
    if (V == 657) return octree_walk<false, false, true, false, true, 1>(a, obj, root, newRay, world_origin, world_dirlen, hit);     // kernel 41's walk + UNIFORM (scalar loads where the wave stands in one node)
    if (V == 669) return octree_walk<true, true, false, false, true, 1>(a, obj, root, newRay, world_origin, world_dirlen, hit);      // kernel 43's walk + UNIFORM, mesh band first
    if (V == 705) return octree_walk<false, false, true, false, true, 0, true>(a, obj, root, newRay, world_origin, world_dirlen, hit);      // kernel 41's walk + DEDUP: list entries tested in the previous leaf are neither loaded nor tested
    if (V == 717) return octree_walk<true, true, false, false, true, 0, true>(a, obj, root, newRay, world_origin, world_dirlen, hit);       // kernel 43's walk + DEDUP (records are prefetched: the arithmetic only), mesh band first
    if (V == 689) return octree_walk<false, false, true, false, true>(a, obj, root, newRay, world_origin, world_dirlen, hit);     // kernel 41's walk launched FOUR waves per workgroup (a 32 x 8 strip), as every measurement arm is and the product was
    if (V == 701) return octree_walk<true, true, false, false, true>(a, obj, root, newRay, world_origin, world_dirlen, hit);      // kernel 43's walk, likewise, mesh band first
    if (V == 621) return octree_walk_nbrec<true, true>(a, obj, root, newRay, world_origin, world_dirlen, hit);      // 541 + a leaf's first triangle with its node: a whole step's data in one round trip
    if (V == 529 || V == 541) return octree_walk_nbrec<V == 541>(a, obj, root, newRay, world_origin, world_dirlen, hit);
    if (V >= 256) return octree_walk_x<((V == 785 ? 273 : V) & 247)>(a, obj, root, newRay, world_origin, world_dirlen, hit);
    return octree_core_diag<V>(a, obj, root, newRay, world_origin, world_dirlen, hit);
}

}  // namespace rptd
