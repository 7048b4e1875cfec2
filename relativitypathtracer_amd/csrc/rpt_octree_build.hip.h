// rpt_octree_build.hip.h — the octree builder on the GPU.
//
// The reference builds each mesh's octree on one CPU core (Mesh.cpp:5-28, Octree.cpp:171-248): for every
// node that is split, each of its triangles is tested against each of the eight child boxes with a
// separating-axis test (Octree.cpp:6-169) and appended, in order, to the child's list.  That product —
// (children of a level) x (triangles of their parent) — is the whole cost and is data parallel.
//
// Here the whole build is ONE submission: the six levels are enqueued back to back on the context's stream, six
// kernels per level, and nothing is read back before the last one has been enqueued.  What a level's kernels need to
// know about the level before (how many nodes, how long its lists are, how many chunks of work there are) lives in a
// header in device memory; grids are launched at their upper bounds (8^level nodes; the list capacity) and surplus
// workgroups leave at once.  Per level:
//   valence_kernel     the stop rule's "most triangles of this node around one vertex" (Octree.cpp:180-190): one thread
//                      per list entry, (node, vertex) occurrences counted in a hash table, atomicMax into the node;
//   split_kernel       one workgroup: which nodes split (depth > 0 and more triangles than the parent's valence,
//                      Octree.cpp:172), their eight child boxes with the reference's arithmetic, and — by prefix sums
//                      over the nodes — where each child's flags and 256-entry chunks of work begin;
//   sat_flag_kernel    one thread per (child, parent-list entry): the SAT with the reference's fp32 operation order
//                      (no contraction) into a flag byte, survivors counted per chunk;
//   chunk_scan_kernel  one workgroup: the counts become offsets into the next level's list buffer;
//   sat_compact_kernel the surviving triangle ids, IN ORDER, into the next level's lists;
//   next_nodes_kernel  the children become the next level's nodes.
// The host then reads the levels back once and does what is sequential by nature — the reference's depth-first
// numbering of nodes and lists and the neighbour links (Octree.cpp:191-246) — so the node and octreeTris arrays come
// out byte-identical to the host builder's (tests/test_gpu_octree.py).  A list that outgrows its buffer raises a flag
// in the header (nothing is written out of bounds) and the host repeats the build with four times the capacity.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/rpt_layout.h"

#pragma clang fp contract(off)

namespace rptb {

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 half(V3 a) { return v3(a.x / 2, a.y / 2, a.z / 2); }

__device__ __forceinline__ float project(int form, V3 e, V3 p) {
    return form == 0 ? (e.z * p.y - e.y * p.z) : (form == 1 ? (-e.z * p.x + e.x * p.z) : (e.y * p.x - e.x * p.y));
}
__device__ __forceinline__ float radius(int form, V3 ae, V3 ext) {
    return form == 0 ? (ae.z * ext.y + ae.y * ext.z) : (form == 1 ? (ae.z * ext.x + ae.x * ext.z) : (ae.y * ext.x + ae.x * ext.y));
}

// Octree.cpp:6-169 — same nine edge tests (with the vertex pairs the reference projects), plane test and
// bounds test, same operation order as csrc/host/rpt_octree.cpp::AABBTriangleIntersection
__device__ inline bool triangle_overlaps_box(V3 A, V3 B, V3 C, V3 bmin, V3 bmax) {
    const V3 center = half(bmin + bmax);
    const V3 extents = half(bmax - bmin);
    const V3 off[3] = {A - center, B - center, C - center};
    const V3 edges[3] = {off[1] - off[0], off[2] - off[1], off[0] - off[2]};
    const int tests[9][4] = {{0, 0, 0, 2}, {0, 1, 0, 2}, {0, 2, 1, 2}, {1, 0, 0, 2}, {1, 1, 0, 2}, {1, 2, 0, 1}, {2, 0, 0, 1}, {2, 1, 0, 1}, {2, 2, 1, 2}};
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const V3 e = edges[tests[t][0]];
        const V3 ae = v3(__builtin_fabsf(e.x), __builtin_fabsf(e.y), __builtin_fabsf(e.z));
        float lo = project(tests[t][1], e, off[tests[t][2]]);
        float hi = project(tests[t][1], e, off[tests[t][3]]);
        if (lo > hi) { const float tmp = lo; lo = hi; hi = tmp; }
        const float rad = radius(tests[t][1], ae, extents);
        if (lo > rad || hi < -rad) return false;
    }
    {
        const V3 ba = edges[0], cb = edges[1];
        const V3 n = v3(ba.y * cb.z - ba.z * cb.y, ba.z * cb.x - ba.x * cb.z, ba.x * cb.y - ba.y * cb.x);
        V3 vmin, vmax;
        if (n.x > 0) { vmin.x = -extents.x - off[0].x; vmax.x = extents.x - off[0].x; } else { vmin.x = extents.x - off[0].x; vmax.x = -extents.x - off[0].x; }
        if (n.y > 0) { vmin.y = -extents.y - off[0].y; vmax.y = extents.y - off[0].y; } else { vmin.y = extents.y - off[0].y; vmax.y = -extents.y - off[0].y; }
        if (n.z > 0) { vmin.z = -extents.z - off[0].z; vmax.z = extents.z - off[0].z; } else { vmin.z = extents.z - off[0].z; vmax.z = -extents.z - off[0].z; }
        // 4-component dot of the reference with w = 0 on both sides: (x + y + z) + 0*0
        if (n.x * vmin.x + n.y * vmin.y + n.z * vmin.z + 0.0f * 0.0f > 0) return false;
        if (n.x * vmax.x + n.y * vmax.y + n.z * vmax.z + 0.0f * 0.0f < 0) return false;
    }
    {
        const float lox = (off[0].x < off[1].x ? off[0].x : off[1].x), hix = (off[0].x > off[1].x ? off[0].x : off[1].x);
        const float loy = (off[0].y < off[1].y ? off[0].y : off[1].y), hiy = (off[0].y > off[1].y ? off[0].y : off[1].y);
        const float loz = (off[0].z < off[1].z ? off[0].z : off[1].z), hiz = (off[0].z > off[1].z ? off[0].z : off[1].z);
        const float lx = lox < off[2].x ? lox : off[2].x, hx = hix > off[2].x ? hix : off[2].x;
        const float ly = loy < off[2].y ? loy : off[2].y, hy = hiy > off[2].y ? hiy : off[2].y;
        const float lz = loz < off[2].z ? loz : off[2].z, hz = hiz > off[2].z ? hiz : off[2].z;
        if (lx > extents.x || hx < -extents.x) return false;
        if (ly > extents.y || hy < -extents.y) return false;
        if (lz > extents.z || hz < -extents.z) return false;
    }
    return true;
}

struct BNode {                // a node of one level while the tree is being built (48 B)
    float mn[3], mx[3];
    unsigned int list_begin, list_count;   // its triangle list in the level's list buffer
    int min_tris, depth;                    // stop rule inputs: the parent's valence; levels left
    int first_child;                        // index of its first child in the next level's node array, -1 = not split
    int valence;
};

struct BuildHeader {          // what the kernels of one level tell the next (device memory; read back once at the end)
    unsigned int n_nodes[8];  // nodes per level 0..6
    unsigned int list_len[8]; // entries in each level's list buffer
    unsigned int n_children, n_chunks, flag_total;   // of the level being split (scratch, rewritten per level)
    unsigned int overflow;    // a list outgrew its buffer: the build is repeated with a larger one
    unsigned int needed;      // ... that holds at least this many entries per level (what the level that overflowed asked for)
};

constexpr unsigned long long HASH_EMPTY = ~0ull;

__device__ __forceinline__ bool node_splits(const BNode &n) { return n.depth > 0 && (int)n.list_count > n.min_tris; }   // Octree.cpp:172

// greatest index i in [0, n) with key(i) <= x (keys ascending; n >= 1 and key(0) <= x assumed)
template <class Key>
__device__ __forceinline__ unsigned int last_not_above(unsigned int n, unsigned int x, Key key) {
    unsigned int lo = 0, hi = n;              // invariant: key(lo) <= x, key(hi) > x (hi == n: sentinel)
    while (hi - lo > 1) {
        const unsigned int mid = lo + (hi - lo) / 2;
        if (key(mid) <= x) lo = mid; else hi = mid;
    }
    return lo;
}

// exclusive prefix sum over the 1024 threads of the workgroup; *total = the sum.  Every thread must call it.
__device__ inline unsigned int block_exclusive_scan(unsigned int v, unsigned int *total, unsigned int *smem /* [17] */) {
    const unsigned int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int x = v;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int y = __shfl_up(x, d);
        if (lane >= (unsigned int)d) x += y;
    }
    if (lane == 63) smem[wave] = x;
    __syncthreads();
    if (wave == 0) {
        const unsigned int w = lane < 16 ? smem[lane] : 0u;
        unsigned int xi = w;
        for (int d = 1; d < 16; d <<= 1) {
            const unsigned int y = __shfl_up(xi, d);
            if (lane >= (unsigned int)d) xi += y;
        }
        if (lane < 16) smem[lane] = xi - w;
        if (lane == 15) smem[16] = xi;
    }
    __syncthreads();
    const unsigned int r = x - v + smem[wave];
    *total = smem[16];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void iota_kernel(int32_t *__restrict__ list, unsigned int n) {
    const unsigned int i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) list[i] = (int32_t)i;
}

// Octree.cpp:180-190.  One thread per entry of the level's list; entries of nodes that will not split are skipped.
__global__ __launch_bounds__(256) void valence_kernel(const uint32_t *__restrict__ triangles, const int32_t *__restrict__ list,
                                                      BNode *__restrict__ nodes, const BuildHeader *__restrict__ hdr, int level,
                                                      unsigned long long *__restrict__ keys, unsigned int *__restrict__ counts,
                                                      unsigned int hash_mask) {
    const unsigned int e = blockIdx.x * 256u + threadIdx.x;
    const unsigned int n = hdr->n_nodes[level];
    if (n == 0 || e >= hdr->list_len[level] || hdr->overflow) return;
    const unsigned int node = last_not_above(n, e, [&](unsigned int i) { return nodes[i].list_begin; });
    const BNode nd = nodes[node];
    if (!node_splits(nd) || e - nd.list_begin >= nd.list_count) return;
    const int tri = list[e];
    for (int k = 0; k < 3; k++) {
        const unsigned long long key = ((unsigned long long)node << 32) | (unsigned long long)triangles[9 * tri + 3 * k];
        unsigned long long h = key * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        unsigned int slot = (unsigned int)h & hash_mask;
        for (unsigned int probe = 0; probe <= hash_mask; probe++, slot = (slot + 1) & hash_mask) {      // bounded: the table has twice the entries
            const unsigned long long prev = atomicCAS(&keys[slot], HASH_EMPTY, key);
            if (prev == HASH_EMPTY || prev == key) {
                const unsigned int c = atomicAdd(&counts[slot], 1u) + 1u;
                // every entry of a node competes for ONE word; a (possibly stale, never too large) look first spares almost all of the atomics
                if ((int)c > __atomic_load_n(&nodes[node].valence, __ATOMIC_RELAXED)) atomicMax(&nodes[node].valence, (int)c);
                break;
            }
        }
    }
}

struct ChildDesc {            // one child box of a node being split, and where its parent's triangle list lives
    float minx, miny, minz, maxx, maxy, maxz;
    unsigned int list_begin;  // parent's list: first entry in the level's list buffer
    unsigned int list_count;
    unsigned int flag_base;   // this child's flags: flags[flag_base + i], i < list_count
    unsigned int chunk_base;  // this child's first 256-entry chunk in the chunk arrays
};

// One workgroup of 1024 threads walks the level's nodes in order.
__global__ __launch_bounds__(1024) void split_kernel(BNode *__restrict__ nodes, BuildHeader *__restrict__ hdr, int level,
                                                     ChildDesc *__restrict__ children, unsigned int *__restrict__ split_nodes,
                                                     unsigned int child_capacity) {
    __shared__ unsigned int smem[17];
    const unsigned int n = hdr->overflow ? 0u : hdr->n_nodes[level];
    unsigned int rank_base = 0, count_base = 0, chunk_base = 0;      // totals over the nodes before this tile
    for (unsigned int first = 0; first < n; first += 1024u) {
        const unsigned int i = first + threadIdx.x;
        BNode nd;
        bool split = false;
        if (i < n) {
            nd = nodes[i];
            split = node_splits(nd);
        }
        unsigned int t_rank, t_count, t_chunks;
        const unsigned int my_count = split ? nd.list_count : 0u, my_chunks = split ? (nd.list_count + 255u) / 256u : 0u;
        const unsigned int rank = rank_base + block_exclusive_scan(split ? 1u : 0u, &t_rank, smem);
        const unsigned int counts_before = count_base + block_exclusive_scan(my_count, &t_count, smem);
        const unsigned int chunks_before = chunk_base + block_exclusive_scan(my_chunks, &t_chunks, smem);
        if (i < n) {
            nodes[i].first_child = split ? (int)(8u * rank) : -1;
            if (split && 8u * rank + 7u < child_capacity) {
                split_nodes[rank] = i;
                const float hx = (nd.mx[0] - nd.mn[0]) / 2, hy = (nd.mx[1] - nd.mn[1]) / 2, hz = (nd.mx[2] - nd.mn[2]) / 2;
                for (int x = 0; x < 2; x++)
                    for (int y = 0; y < 2; y++)
                        for (int z = 0; z < 2; z++) {       // creation order == child index z + 2y + 4x (Octree.cpp:191-201)
                            const unsigned int k = (unsigned int)(z + 2 * y + 4 * x);
                            ChildDesc c;
                            // child.min = min + ex*x + ey*y + ez*z, component by component, zeros included
                            c.minx = nd.mn[0] + hx * (float)x + 0.0f * (float)y + 0.0f * (float)z;
                            c.miny = nd.mn[1] + 0.0f * (float)x + hy * (float)y + 0.0f * (float)z;
                            c.minz = nd.mn[2] + 0.0f * (float)x + 0.0f * (float)y + hz * (float)z;
                            c.maxx = c.minx + hx; c.maxy = c.miny + hy; c.maxz = c.minz + hz;
                            c.list_begin = nd.list_begin;
                            c.list_count = nd.list_count;
                            c.flag_base = 8u * counts_before + k * nd.list_count;
                            c.chunk_base = 8u * chunks_before + k * my_chunks;
                            children[8u * rank + k] = c;
                        }
            }
        }
        rank_base += t_rank; count_base += t_count; chunk_base += t_chunks;
    }
    if (threadIdx.x == 0) {
        hdr->n_children = 8u * rank_base;
        hdr->flag_total = 8u * count_base;
        hdr->n_chunks = 8u * chunk_base;
        if (level + 1 < 8) hdr->n_nodes[level + 1] = 8u * rank_base;
        if (8u * rank_base > child_capacity) hdr->overflow = 1u;      // cannot happen (8^level bound); kept as a guard
    }
}

// One thread per (chunk, lane): chunk -> (child, 256 consecutive entries of its parent's list).
__global__ __launch_bounds__(256) void sat_flag_kernel(const rpt_float3 *__restrict__ vertices, const uint32_t *__restrict__ triangles,
                                                       const int32_t *__restrict__ level_lists, const ChildDesc *__restrict__ children,
                                                       const BuildHeader *__restrict__ hdr, unsigned char *__restrict__ flags,
                                                       unsigned int *__restrict__ chunk_counts) {
    const unsigned int chunk = blockIdx.x;
    if (hdr->overflow || chunk >= hdr->n_chunks) return;
    const unsigned int child = last_not_above(hdr->n_children, chunk, [&](unsigned int i) { return children[i].chunk_base; });
    const ChildDesc c = children[child];
    const unsigned int i = (chunk - c.chunk_base) * 256u + threadIdx.x;
    bool keep = false;
    if (i < c.list_count) {
        const int tri = level_lists[c.list_begin + i];
        const rpt_float3 a = vertices[triangles[9 * tri + 0]], b = vertices[triangles[9 * tri + 3]], cc = vertices[triangles[9 * tri + 6]];
        keep = triangle_overlaps_box(v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(cc.x, cc.y, cc.z), v3(c.minx, c.miny, c.minz),
                                     v3(c.maxx, c.maxy, c.maxz));
        flags[c.flag_base + i] = keep ? 1 : 0;
    }
    __shared__ unsigned int wave_counts[4];
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0) wave_counts[threadIdx.x >> 6] = (unsigned int)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) chunk_counts[chunk] = wave_counts[0] + wave_counts[1] + wave_counts[2] + wave_counts[3];
}

// One workgroup: chunk counts -> offsets into the next level's list buffer; its length into the header.
__global__ __launch_bounds__(1024) void chunk_scan_kernel(const unsigned int *__restrict__ chunk_counts, unsigned int *__restrict__ chunk_out,
                                                          BuildHeader *__restrict__ hdr, int level, unsigned int list_capacity) {
    __shared__ unsigned int smem[17];
    const unsigned int n = hdr->overflow ? 0u : hdr->n_chunks;
    unsigned long long base = 0;
    for (unsigned int first = 0; first < n; first += 1024u) {
        const unsigned int i = first + threadIdx.x;
        const unsigned int v = i < n ? chunk_counts[i] : 0u;
        unsigned int total;
        const unsigned int before = block_exclusive_scan(v, &total, smem);
        if (i < n) chunk_out[i] = (unsigned int)(base + before);      // base <= capacity is checked below before anything is written there
        base += total;                                                   // (summed to the end even past the capacity: the host wants to know how much was asked for)
    }
    if (threadIdx.x == 0) {
        if (base > (unsigned long long)list_capacity) {
            hdr->overflow = 1u;
            hdr->needed = base > 0xffffffffull ? 0xffffffffu : (unsigned int)base;
            base = 0;
        }
        if (level + 1 < 8) hdr->list_len[level + 1] = (unsigned int)base;
    }
}

// Ordered compaction: entry i of the parent's list that survived goes to out[chunk_out[chunk] + (survivors before i in the chunk)].
__global__ __launch_bounds__(256) void sat_compact_kernel(const int32_t *__restrict__ level_lists, const ChildDesc *__restrict__ children,
                                                          const BuildHeader *__restrict__ hdr, const unsigned char *__restrict__ flags,
                                                          const unsigned int *__restrict__ chunk_out, int32_t *__restrict__ next_lists,
                                                          unsigned int list_capacity) {
    const unsigned int chunk = blockIdx.x;
    if (hdr->overflow || chunk >= hdr->n_chunks) return;
    const unsigned int child = last_not_above(hdr->n_children, chunk, [&](unsigned int i) { return children[i].chunk_base; });
    const ChildDesc c = children[child];
    const unsigned int i = (chunk - c.chunk_base) * 256u + threadIdx.x;
    const bool keep = i < c.list_count && flags[c.flag_base + i] != 0;
    __shared__ unsigned int wave_counts[4];
    const unsigned long long m = __ballot(keep);
    const unsigned int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_counts[wave] = (unsigned int)__popcll(m);
    __syncthreads();
    unsigned int before = (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
    for (unsigned int w = 0; w < wave; w++) before += wave_counts[w];
    const unsigned int dst = chunk_out[chunk] + before;
    if (keep && dst < list_capacity) next_lists[dst] = level_lists[c.list_begin + i];
}

// The children of the level become the nodes of the next one (Octree.cpp:191-211): lists consecutive in child order.
__global__ __launch_bounds__(256) void next_nodes_kernel(const BNode *__restrict__ nodes, const ChildDesc *__restrict__ children,
                                                         const unsigned int *__restrict__ split_nodes, const unsigned int *__restrict__ chunk_out,
                                                         const BuildHeader *__restrict__ hdr, int level, BNode *__restrict__ next_nodes) {
    const unsigned int c = blockIdx.x * 256u + threadIdx.x;
    if (hdr->overflow || c >= hdr->n_children) return;
    const ChildDesc d = children[c];
    const BNode parent = nodes[split_nodes[c / 8u]];
    BNode b;
    b.mn[0] = d.minx; b.mn[1] = d.miny; b.mn[2] = d.minz;
    b.mx[0] = d.maxx; b.mx[1] = d.maxy; b.mx[2] = d.maxz;
    const unsigned int begin = chunk_out[d.chunk_base];
    const unsigned int end = c + 1 < hdr->n_children ? chunk_out[children[c + 1].chunk_base] : hdr->list_len[level + 1];
    b.list_begin = begin;
    b.list_count = end - begin;
    b.min_tris = parent.valence;
    b.depth = parent.depth - 1;
    b.first_child = -1;
    b.valence = 0;
    next_nodes[c] = b;
}

}  // namespace rptb
