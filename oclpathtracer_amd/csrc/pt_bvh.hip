// pt_bvh.hip -- LBVH over the scene's triangles, built on the GPU at scene upload (gfx950 only).
//
// The reference finds the closest hit by brute force (intersectWorld, GenerateColors.cl:137-154);
// SURVEY S8f rank 3 replaces the O(N) loop for large scenes.  The closest hit is order-free: the
// winner of the reference's ascending loop with its strict `t < tmax` (:125) is argmin (t, index)
// over the triangles whose exact test passes, so ANY traversal that applies the same exact test
// (pt_tri_pass2's arithmetic) to a superset of those triangles and keeps the lexicographic minimum
// returns the same (t, u, v, index) bit for bit.  The hierarchy only has to be conservative: a
// node is skipped when the ray misses its box grown by PT_BVH_EPS x (largest |coordinate|).
// (A binary32 Moeller-Trumbore test can accept a hit that lies outside the triangle by
// ~1e-6 x distance / cos(incidence); for rays within ~0.05 degrees of a triangle's plane that
// displacement is unbounded, so no finite box margin is PROVABLY conservative.  The margin covers
// cos(incidence) >= 1e-2 with a factor 10 to spare; tests compare against brute force.)
//
// Build (Karras 2012): 30-bit Morton code of the box centre | triangle index -> 64-bit keys,
// hipcub radix sort, one thread per internal node finds its range and split, bottom-up box refit
// with one atomic flag per internal node (fp32 boxes of both children in the binary node).  The binary
// tree is then collapsed three levels at a time into the eight-child nodes the trace kernel walks
// (PtBvh8Node, pt_kernels.h: 80 bytes in a 128-byte slot, 8-bit boxes in the node's frame, children
// stored consecutively, slots assigned by octant): count / scan / assign / emit below.
#include "pt_kernels.h"

#include <hipcub/hipcub.hpp>

#define PT_BVH_EPS 1.2e-4f
// BIG triangles stay out of the hierarchy.  An LBVH places a triangle by its centre: one that spans a good part of
// the scene (the Cornell box's walls among 10^6 centimetre-sized ones) sits at a deep leaf and inflates the boxes
// of all ~20 of its ancestors to its own size, and every ray then enters hundreds of such nodes.  Triangles whose
// box is longer than 1/PT_BVH_BIG_DIV of the scene's longest side -- if there are at most PT_BVH_BIG_MAX of
// them -- get an empty box in the tree and are searched by the brute-force two-pass search instead
// (pt_trace_bvh_body), in ascending index order like the reference's loop; the closest hit is the lexicographic
// minimum of (t, index) over both searches, so the result is unchanged.
#define PT_BVH_BIG_DIV 16.0f

namespace {

__device__ __forceinline__ unsigned pt_ordered(float f)
{
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b ^ 0x80000000u);  // unsigned order == float order
}
__device__ __forceinline__ float pt_unordered(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

__device__ __forceinline__ bool pt_tri_box(const PtRawTriangle& t, float lo[3], float hi[3])
{
    bool finite = true;
    for (int k = 0; k < 3; ++k) {
        const float a = t.p1[k], b = t.p2[k], c = t.p3[k];
        finite = finite && __builtin_isfinite(a) && __builtin_isfinite(b) && __builtin_isfinite(c);
        lo[k] = fminf(a, fminf(b, c));
        hi[k] = fmaxf(a, fmaxf(b, c));
    }
    return finite;
}

// bounds[0..2] = min, [3..5] = max of the finite triangles' boxes, [6] = largest |coordinate| (ordered keys)
__global__ void pt_bvh_bounds_kernel(const PtRawTriangle* __restrict__ raw, int ntri, unsigned* __restrict__ bounds)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntri) return;
    float lo[3], hi[3];
    if (!pt_tri_box(raw[i], lo, hi)) return;  // a triangle with a non-finite vertex is never hit (t is NaN or Inf)
    float m = 0.0f;
    for (int k = 0; k < 3; ++k) {
        atomicMin(&bounds[k], pt_ordered(lo[k]));
        atomicMax(&bounds[3 + k], pt_ordered(hi[k]));
        m = fmaxf(m, fmaxf(fabsf(lo[k]), fabsf(hi[k])));
    }
    atomicMax(&bounds[6], pt_ordered(m));
}

// bounds[8] = bit pattern of the big-triangle threshold (0x7f800000 = +Inf: no split), bounds[9] = number found
__device__ __forceinline__ bool pt_tri_is_big(const float lo[3], const float hi[3], const unsigned* bounds)
{
    const float thr = __uint_as_float(bounds[8]);
    return fmaxf(hi[0] - lo[0], fmaxf(hi[1] - lo[1], hi[2] - lo[2])) > thr;
}

__global__ void pt_bvh_threshold_kernel(unsigned* __restrict__ bounds)
{
    float ext = 0.0f;
    for (int k = 0; k < 3; ++k) ext = fmaxf(ext, pt_unordered(bounds[3 + k]) - pt_unordered(bounds[k]));
    bounds[8] = __float_as_uint(ext > 0.0f ? ext / PT_BVH_BIG_DIV : __builtin_inff());  // (an empty / degenerate scene: no split)
    bounds[9] = 0u;
}

__global__ void pt_bvh_big_collect_kernel(const PtRawTriangle* __restrict__ raw, int ntri, unsigned* __restrict__ bounds,
                                          int* __restrict__ bigidx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntri) return;
    float lo[3], hi[3];
    if (!pt_tri_box(raw[i], lo, hi) || !pt_tri_is_big(lo, hi, bounds)) return;
    const unsigned k = atomicAdd(&bounds[9], 1u);
    if (k < PT_BVH_BIG_MAX) bigidx[k] = i;
}

// one thread: too many big triangles -> no split; otherwise sort their indices ascending (ties in t go to the lowest
// index, as in the reference's loop) and copy their prepared records
__global__ void pt_bvh_big_finish_kernel(unsigned* __restrict__ bounds, int* __restrict__ bigidx, const PtPrepTriangle* __restrict__ prep,
                                         PtPrepTriangle* __restrict__ bigtab, int* __restrict__ nbig_out)
{
    unsigned n = bounds[9];
    if (n > PT_BVH_BIG_MAX) {
        n = 0u;
        bounds[8] = 0x7f800000u;  // +Inf: nothing is big
    }
    for (unsigned a = 1; a < n; ++a) {
        const int v = bigidx[a];
        unsigned b = a;
        for (; b > 0 && bigidx[b - 1] > v; --b) bigidx[b] = bigidx[b - 1];
        bigidx[b] = v;
    }
    for (unsigned a = 0; a < n; ++a) bigtab[a] = prep[bigidx[a]];
    *nbig_out = (int)n;
}

__device__ __forceinline__ unsigned pt_expand10(unsigned v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void pt_bvh_keys_kernel(const PtRawTriangle* __restrict__ raw, int ntri, const unsigned* __restrict__ bounds,
                                   unsigned long long* __restrict__ keys)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntri) return;
    float lo[3], hi[3];
    unsigned code = 0x3fffffffu;  // non-finite and big triangles sort to the end
    if (pt_tri_box(raw[i], lo, hi) && !pt_tri_is_big(lo, hi, bounds)) {
        unsigned q[3];
        for (int k = 0; k < 3; ++k) {
            const float smin = pt_unordered(bounds[k]), smax = pt_unordered(bounds[3 + k]);
            const float ext = smax - smin;
            float c = ext > 0.0f ? (0.5f * (lo[k] + hi[k]) - smin) / ext : 0.0f;
            c = fminf(fmaxf(c * 1024.0f, 0.0f), 1023.0f);
            q[k] = (unsigned)c;
        }
        code = (pt_expand10(q[0]) << 2) | (pt_expand10(q[1]) << 1) | pt_expand10(q[2]);
    }
    keys[i] = ((unsigned long long)code << 32) | (unsigned)i;
}

// length of the common prefix of keys i and j (keys are unique), -1 outside [0, n)
__device__ __forceinline__ int pt_delta(const unsigned long long* keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}

// nodes: internal i in [0, n-1), leaf k at n-1+k.  parent[] for every node, -1 for the root.
__global__ void pt_bvh_hierarchy_kernel(const unsigned long long* __restrict__ keys, int n, PtBvhNode* __restrict__ nodes,
                                        int* __restrict__ parent, int* __restrict__ right_child)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = pt_delta(keys, n, i, i + 1) - pt_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = pt_delta(keys, n, i, i - d);
    int lmax = 2;
    while (pt_delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (pt_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = pt_delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (pt_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int left = lo == gamma ? n - 1 + gamma : gamma;
    const int right = hi == gamma + 1 ? n - 1 + gamma + 1 : gamma + 1;
    // links: internal child = its index; leaf child = 0x80000000 | position in the sorted order
    nodes[i].link_l = left >= n - 1 ? 0x80000000u | (unsigned)(left - (n - 1)) : (unsigned)left;
    nodes[i].link_r = right >= n - 1 ? 0x80000000u | (unsigned)(right - (n - 1)) : (unsigned)right;
    right_child[i] = right;
    parent[left] = i;
    parent[right] = i;
    if (i == 0) parent[0] = -1;
}

// one leaf per triangle, in sorted order: n = ntri leaves; tkeys = the sorted triangle keys
__global__ void pt_bvh_refit_kernel(const PtRawTriangle* __restrict__ raw, const unsigned long long* __restrict__ tkeys, int n,
                                    const unsigned* __restrict__ bounds, PtBvhNode* nodes,
                                    const int* __restrict__ parent, const int* __restrict__ right_child, int* __restrict__ flags)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float eps = PT_BVH_EPS * pt_unordered(bounds[6]) + 1e-30f;
    float lo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, hi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };  // empty: never entered
    {
        const int tri = (int)(unsigned)tkeys[k];
        float tlo[3], thi[3];
        if (pt_tri_box(raw[tri], tlo, thi) && !pt_tri_is_big(tlo, thi, bounds))
            for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], tlo[a] - eps); hi[a] = fmaxf(hi[a], thi[a] + eps); }
    }
    // climb: a child stores its box in its parent's record; the second child to arrive unions the two
    // and carries the result one level up.  (A radix tree over 64-bit keys is at most 64 levels deep:
    // the cap only guards against a damaged tree.)
    int node = n - 1 + k;
    for (int guard = 0; guard < 80; ++guard) {
        const int p = parent[node];
        if (p < 0) return;
        float* slot_min = right_child[p] == node ? nodes[p].rmin : nodes[p].lmin;
        float* slot_max = right_child[p] == node ? nodes[p].rmax : nodes[p].lmax;
        for (int a = 0; a < 3; ++a) { slot_min[a] = lo[a]; slot_max[a] = hi[a]; }
        __threadfence();
        if (atomicAdd(&flags[p], 1) == 0) return;
        __threadfence();
        const volatile PtBvhNode* q = &nodes[p];  // the sibling's stores, made visible by the fences
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(q->lmin[a], q->rmin[a]);
            hi[a] = fmaxf(q->lmax[a], q->rmax[a]);
        }
        node = p;
    }
}

__device__ __forceinline__ float pt_bvh_decode(unsigned q, float step, float origin) { return __builtin_fmaf((float)q, step, origin); }

// ---- binary fp32 nodes -> the eight-child nodes the traversal reads (PtBvh8Node, pt_kernels.h) -------------------
// Every binary node at a depth that is a multiple of three is the root of one eight-child node.  Three passes, one
// thread per binary node, no level-by-level dependency:
//   count:  how many of the (present) children are nodes / leaves;            then an exclusive scan of the counts
//   assign: the node children of P get the consecutive new indices 1 + scan_nodes(P) + rank (rank in slot order)
//   emit:   P writes its node at its new index and its leaf children's records at scan_leaves(P) + rank
// (a child whose box is empty -- a subtree of triangles kept out of the hierarchy or with non-finite vertices -- is
// dropped: its slot stays empty and nothing below it is ever written or read)
struct PtGather8 {
    int m;              // present children
    unsigned link[8];   // binary links: node index, or 0x80000000 | sorted position
    float lo[8][3], hi[8][3];
    int slot[8];        // the slot each child sits in (pt_bvh8_gather assigns them)
};

__device__ __forceinline__ int pt_bvh_depth(const int* __restrict__ parent, int i)
{
    int depth = 0;
    for (int p = parent[i]; p >= 0 && depth < 256; p = parent[p]) ++depth;
    return depth;
}

__device__ void pt_bvh8_gather(const PtBvhNode* __restrict__ wide, int i, PtGather8& g)
{
    unsigned link[8];
    float lo[8][3], hi[8][3];
    int m = 2;
    {
        const PtBvhNode w = wide[i];
        link[0] = w.link_l; link[1] = w.link_r;
        for (int a = 0; a < 3; ++a) { lo[0][a] = w.lmin[a]; hi[0][a] = w.lmax[a]; lo[1][a] = w.rmin[a]; hi[1][a] = w.rmax[a]; }
    }
    for (int round = 1; round < 3; ++round) {
        unsigned l2[8];
        float lo2[8][3], hi2[8][3];
        int m2 = 0;
        for (int k = 0; k < m; ++k) {
            bool present = true;
            for (int a = 0; a < 3; ++a) present = present && (lo[k][a] <= hi[k][a]);
            if ((link[k] & 0x80000000u) || !present) {  // a leaf stays; so does an empty child (dropped below)
                l2[m2] = link[k];
                for (int a = 0; a < 3; ++a) { lo2[m2][a] = lo[k][a]; hi2[m2][a] = hi[k][a]; }
                ++m2;
            } else {
                const PtBvhNode c = wide[link[k]];
                l2[m2] = c.link_l;
                for (int a = 0; a < 3; ++a) { lo2[m2][a] = c.lmin[a]; hi2[m2][a] = c.lmax[a]; }
                ++m2;
                l2[m2] = c.link_r;
                for (int a = 0; a < 3; ++a) { lo2[m2][a] = c.rmin[a]; hi2[m2][a] = c.rmax[a]; }
                ++m2;
            }
        }
        m = m2;
        for (int k = 0; k < m; ++k) {
            link[k] = l2[k];
            for (int a = 0; a < 3; ++a) { lo[k][a] = lo2[k][a]; hi[k][a] = hi2[k][a]; }
        }
    }
    g.m = 0;
    for (int k = 0; k < m; ++k) {
        bool present = true;
        for (int a = 0; a < 3; ++a) present = present && (lo[k][a] <= hi[k][a]);  // empty (3e38, -3e38) or NaN boxes are dropped
        if (!present) continue;
        g.link[g.m] = link[k];
        for (int a = 0; a < 3; ++a) { g.lo[g.m][a] = lo[k][a]; g.hi[g.m][a] = hi[k][a]; }
        ++g.m;
    }
    // slots: the child's position relative to the centre of the children's union decides the octant it would like;
    // greedy assignment, best (child, slot) pair first (deterministic: both passes that call this get the same slots)
    float pc[3];
    for (int a = 0; a < 3; ++a) {
        float l = 3.0e38f, h = -3.0e38f;
        for (int k = 0; k < g.m; ++k) { l = fminf(l, g.lo[k][a]); h = fmaxf(h, g.hi[k][a]); }
        pc[a] = 0.5f * l + 0.5f * h;
    }
    unsigned used = 0u;
    for (int k = 0; k < 8; ++k) g.slot[k] = -1;
    for (int it = 0; it < g.m; ++it) {
        float best = -__builtin_inff();
        int bc = -1, bs = -1;
        for (int k = 0; k < g.m; ++k) {
            if (g.slot[k] >= 0) continue;
            float off[3];
            for (int a = 0; a < 3; ++a) off[a] = (0.5f * g.lo[k][a] + 0.5f * g.hi[k][a]) - pc[a];
            for (int sl = 0; sl < 8; ++sl) {
                if (used & (1u << sl)) continue;
                const float cost = ((sl & 1) ? off[0] : -off[0]) + ((sl & 2) ? off[1] : -off[1]) + ((sl & 4) ? off[2] : -off[2]);
                if (bc < 0 || cost > best) { best = cost; bc = k; bs = sl; }
            }
        }
        g.slot[bc] = bs;
        used |= 1u << bs;
    }
}

__global__ void pt_bvh8_count_kernel(const PtBvhNode* __restrict__ wide, const int* __restrict__ parent, int n, unsigned long long* __restrict__ cnt)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    unsigned long long c = 0ull;
    if (pt_bvh_depth(parent, i) % 3 == 0) {
        PtGather8 g;
        pt_bvh8_gather(wide, i, g);
        for (int k = 0; k < g.m; ++k) c += (g.link[k] & 0x80000000u) ? (1ull << 32) : 1ull;
    }
    cnt[i] = c;  // node children | leaf children << 32
}

__global__ void pt_bvh8_assign_kernel(const PtBvhNode* __restrict__ wide, const int* __restrict__ parent, int n,
                                      const unsigned long long* __restrict__ base, int* __restrict__ newidx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1 || pt_bvh_depth(parent, i) % 3 != 0) return;
    PtGather8 g;
    pt_bvh8_gather(wide, i, g);
    int rank = 0;
    for (int sl = 0; sl < 8; ++sl)
        for (int k = 0; k < g.m; ++k)
            if (g.slot[k] == sl && !(g.link[k] & 0x80000000u)) newidx[g.link[k]] = 1 + (int)(unsigned)base[i] + rank++;
}

// keys: the sorted triangle keys (the leaf at sorted position c is triangle (unsigned)keys[c])
__global__ void pt_bvh8_emit_kernel(const PtBvhNode* __restrict__ wide, const int* __restrict__ parent, int n,
                                    const unsigned long long* __restrict__ base, const int* __restrict__ newidx,
                                    const unsigned long long* __restrict__ keys, const PtPrepTriangle* __restrict__ prep,
                                    PtBvh8Node* __restrict__ out, PtLeafTri* __restrict__ ltris)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1 || pt_bvh_depth(parent, i) % 3 != 0) return;
    const int my = i == 0 ? 0 : newidx[i];
    if (my < 0) return;  // below a dropped child: unreachable
    PtGather8 g;
    pt_bvh8_gather(wide, i, g);
    PtBvh8Node o;
    o.child_base = 1u + (unsigned)base[i];
    o.tri_base = (unsigned)(base[i] >> 32);
    unsigned imask = 0u, lmask = 0u;
    for (int k = 0; k < g.m; ++k) {
        if (g.link[k] & 0x80000000u) lmask |= 1u << g.slot[k];
        else imask |= 1u << g.slot[k];
    }
    o.lmask = lmask;
    o.pad0 = 0u;
    for (unsigned k = 0; k < sizeof o.pad / sizeof o.pad[0]; ++k) o.pad[k] = 0u;
    // the leaf children's records, in slot order
    {
        unsigned rank = 0u;
        for (int sl = 0; sl < 8; ++sl)
            for (int k = 0; k < g.m; ++k)
                if (g.slot[k] == sl && (g.link[k] & 0x80000000u)) {
                    const unsigned tri = (unsigned)keys[g.link[k] & 0x7fffffffu];
                    const PtPrepTriangle t = prep[tri];
                    PtLeafTri r;
                    for (int a = 0; a < 3; ++a) { r.p1[a] = t.p1[a]; r.e1[a] = t.e1[a]; r.e2[a] = t.e2[a]; }
                    r.index = tri;
                    for (unsigned z = 0; z < sizeof r.pad / sizeof r.pad[0]; ++z) r.pad[z] = 0.0f;
                    if (rank == 0u) {  // the first leaf's record also rides in the node's own line (the trace kernel reads it there)
                        static_assert(sizeof o.pad >= 48, "room for one leaf record");
                        const uint32_t* w = reinterpret_cast<const uint32_t*>(&r);
                        for (int z = 0; z < 12; ++z) o.pad[z] = w[z];
                    }
                    ltris[o.tri_base + rank++] = r;
                }
    }
    unsigned ex[3] = { 1u, 1u, 1u };
    for (int a = 0; a < 3; ++a) {
        float org = 3.0e38f, top = -3.0e38f;
        for (int k = 0; k < g.m; ++k) { org = fminf(org, g.lo[k][a]); top = fmaxf(top, g.hi[k][a]); }
        if (!(org <= top)) { org = 0.0f; top = 0.0f; }
        o.origin[a] = org;
        // smallest power of two `step` with decode(255) >= top; then every bound rounded outward and CHECKED by decoding
        int e2 = -126;
        const float ext = top - org;
        if (ext > 0.0f) (void)frexpf(ext / 255.0f, &e2);
        int be = e2 + 127;  // biased exponent of 2^e2
        be = be < 1 ? 1 : (be > 254 ? 254 : be);
        for (;;) {
            const float step = __uint_as_float((unsigned)be << 23);
            bool ok = true;
            for (int sl = 0; sl < 8; ++sl) { o.qlo[a][sl] = 255; o.qhi[a][sl] = 0; }  // an empty slot: inverted (and in neither mask)
            for (int k = 0; k < g.m && ok; ++k) {
                float fl = floorf((g.lo[k][a] - org) / step), fh = ceilf((g.hi[k][a] - org) / step);
                fl = fminf(fmaxf(fl, 0.0f), 255.0f);
                fh = fminf(fmaxf(fh, 0.0f), 255.0f);
                unsigned ql = (unsigned)fl, qh = (unsigned)fh;
                while (ql > 0u && pt_bvh_decode(ql, step, org) > g.lo[k][a]) --ql;
                while (qh < 255u && pt_bvh_decode(qh, step, org) < g.hi[k][a]) ++qh;
                if (pt_bvh_decode(ql, step, org) > g.lo[k][a] || pt_bvh_decode(qh, step, org) < g.hi[k][a]) ok = false;
                o.qlo[a][g.slot[k]] = (uint8_t)ql;
                o.qhi[a][g.slot[k]] = (uint8_t)qh;
            }
            if (ok || be >= 254) break;  // (be = 254 always suffices for finite boxes: 255 x 2^127 spans binary32)
            ++be;
        }
        ex[a] = (unsigned)be;
    }
    o.meta = ex[0] | (ex[1] << 8) | (ex[2] << 16) | (imask << 24);
    out[my] = o;
}

}  // namespace

size_t ptk_bvh_node_count(int ntri) { return ntri > 1 ? (size_t)ptk_bvh_leaf_count(ntri) - 1 : 0; }

static size_t pt_bvh_scan_bytes(int n)
{
    size_t bytes = 0;
    unsigned long long* nullk = nullptr;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, nullk, nullk, n);
    return bytes;
}

size_t ptk_bvh_temp_bytes(int ntri)
{
    size_t cub = 0;
    unsigned long long* nullk = nullptr;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, cub, nullk, nullk, ntri, 0, 62);
    const size_t scan = pt_bvh_scan_bytes(ntri);
    const size_t n = (size_t)ntri;
    // keys, sorted keys, parent[2n-1], right_child[n-1], flags[n-1], bounds[16], the fp32 nodes, child counts and
    // their scan, new indices, cub temp (sort and scan use it in turn)
    return 16 * n + 4 * (2 * n) + 4 * n + 4 * n + 64 + sizeof(PtBvhNode) * n + 16 * n + 4 * n + (cub > scan ? cub : scan) + 2048;
}

hipError_t ptk_bvh_build(const PtRawTriangle* raw, const PtPrepTriangle* prep, int ntri, PtBvh8Node* nodes8, PtLeafTri* ltris,
                         PtPrepTriangle* bigtab, int* bigidx, int* nbig_dev, void* temp, size_t temp_bytes, hipStream_t s)
{
    if (ntri < 2) return hipErrorInvalidValue;  // callers use the hierarchy for ntri >= 2 only
    const size_t n = (size_t)ntri;
    char* p = (char*)temp;
    unsigned long long* keys = (unsigned long long*)p; p += 8 * n;
    unsigned long long* sorted = (unsigned long long*)p; p += 8 * n;
    unsigned long long* cnt = (unsigned long long*)p; p += 8 * n;
    unsigned long long* base = (unsigned long long*)p; p += 8 * n;
    int* parent = (int*)p; p += 4 * 2 * n;
    int* right_child = (int*)p; p += 4 * n;
    int* flags = (int*)p; p += 4 * n;
    int* newidx = (int*)p; p += 4 * n;
    unsigned* bounds = (unsigned*)p; p += 64;
    p = (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    PtBvhNode* nodes = (PtBvhNode*)p; p += sizeof(PtBvhNode) * n;  // fp32 nodes: refit works on these, then compressed
    p = (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    size_t cub = temp_bytes - (size_t)(p - (char*)temp);
    const unsigned init[16] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0x80000000u /* ordered(0) */, 0u,
                                0x7f800000u /* threshold: +Inf */, 0u, 0u, 0u, 0u, 0u, 0u, 0u };
    hipError_t e = hipMemcpyAsync(bounds, init, sizeof init, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    if ((e = hipMemsetAsync(flags, 0, 4 * n, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(newidx, 0xff, 4 * n, s)) != hipSuccess) return e;  // -1: not reachable
    const dim3 blk(256), grd((ntri + 255) / 256);
    hipLaunchKernelGGL(pt_bvh_bounds_kernel, grd, blk, 0, s, raw, ntri, bounds);
    hipLaunchKernelGGL(pt_bvh_threshold_kernel, dim3(1), dim3(1), 0, s, bounds);
    hipLaunchKernelGGL(pt_bvh_big_collect_kernel, grd, blk, 0, s, raw, ntri, bounds, bigidx);
    hipLaunchKernelGGL(pt_bvh_big_finish_kernel, dim3(1), dim3(1), 0, s, bounds, bigidx, prep, bigtab, nbig_dev);
    hipLaunchKernelGGL(pt_bvh_keys_kernel, grd, blk, 0, s, raw, ntri, bounds, keys);
    if ((e = hipcub::DeviceRadixSort::SortKeys(p, cub, keys, sorted, ntri, 0, 62, s)) != hipSuccess) return e;
    const int nleaves = ntri;  // one triangle per leaf; nleaves >= 2
    const dim3 lgrd((nleaves + 255) / 256);
    hipLaunchKernelGGL(pt_bvh_hierarchy_kernel, lgrd, blk, 0, s, sorted, nleaves, nodes, parent, right_child);
    hipLaunchKernelGGL(pt_bvh_refit_kernel, lgrd, blk, 0, s, raw, sorted, nleaves, bounds, nodes, parent, right_child, flags);
    hipLaunchKernelGGL(pt_bvh8_count_kernel, lgrd, blk, 0, s, nodes, parent, nleaves, cnt);
    if ((e = hipcub::DeviceScan::ExclusiveSum(p, cub, cnt, base, nleaves - 1, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(pt_bvh8_assign_kernel, lgrd, blk, 0, s, nodes, parent, nleaves, base, newidx);
    hipLaunchKernelGGL(pt_bvh8_emit_kernel, lgrd, blk, 0, s, nodes, parent, nleaves, base, newidx, sorted, prep, nodes8, ltris);
    return hipGetLastError();
}
