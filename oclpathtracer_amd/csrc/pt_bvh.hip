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
// (PtBvh8Node, pt_kernels.h: 64 bytes, 8-bit boxes in the node's frame, 16-bit origin on a scene grid, children
// -- nodes and leaf records alike -- stored consecutively, slots assigned by octant); WHICH binary nodes become eight-child nodes is chosen by dynamic
// programming over the subtree costs (pt_bvh8_cost_kernel, pt_bvh8_topdown_kernel).
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

// the grid of the nodes' 16-bit origins (PtBvhGrid): it must reach below every box a node can hold -- triangle boxes grown
// by eps (pt_bvh_refit_kernel) -- and its 65 536 positions per axis must span them; steps are powers of two
__global__ void pt_bvh_grid_kernel(const unsigned* __restrict__ bounds, PtBvhGrid* __restrict__ grid)
{
    const float eps = PT_BVH_EPS * pt_unordered(bounds[6]) + 1e-30f;
    for (int a = 0; a < 3; ++a) {
        float lo = pt_unordered(bounds[a]), hi = pt_unordered(bounds[3 + a]);
        if (!(lo <= hi)) { lo = 0.0f; hi = 0.0f; }  // no finite triangle at all
        const float gmin = lo - 2.0f * eps;
        const float ext = (hi - lo) + 4.0f * eps;
        int e2 = -126;
        if (ext > 0.0f) (void)frexpf(ext / 65535.0f, &e2);  // ext / 65535 = m 2^e2 with m < 1: 65535 steps of 2^e2 cover ext
        int be = e2 + 127;
        be = be < 1 ? 1 : (be > 254 ? 254 : be);
        grid->gmin[a] = gmin;
        grid->gstep[a] = __uint_as_float((unsigned)be << 23);
    }
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

// the caller's records of the triangles kept out of the hierarchy, in table order (for ptk_prep_triangles: is the table made of quads?)
__global__ void pt_bvh_big_raw_kernel(const PtRawTriangle* __restrict__ raw, const int* __restrict__ bigidx, int nbig, PtRawTriangle* __restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nbig) out[k] = raw[bigidx[k]];
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
// (a child whose box is empty -- a subtree of triangles kept out of the hierarchy or with non-finite vertices -- is
// dropped: its slot stays empty and nothing below it is ever written or read)
struct PtGather8 {
    int m;              // present children
    unsigned link[8];   // binary links: node index, or 0x80000000 | sorted position
    float lo[8][3], hi[8][3];
    int slot[8];        // the slot each child sits in (pt_bvh8_slots)
};

__device__ void pt_bvh8_slots(PtGather8& g)
{
    // slots: the child's position relative to the centre of the children's union decides the octant it would like;
    // greedy assignment, best (child, slot) pair first
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

// the node of the gathered children (slots assigned) into recs[self]; its children are records base, base + 1, ... in slot
// order: the leaf children's records are written here, the node children's by the threads that take them off the queue
__device__ void pt_bvh8_write(const PtGather8& g, unsigned self, unsigned base, const unsigned long long* __restrict__ keys,
                              const PtPrepTriangle* __restrict__ prep, const PtBvhGrid* __restrict__ grid, PtBvh8Node* __restrict__ recs)
{
    PtBvh8Node o;
    o.base = base;
    o.pad = 0;
    unsigned imask = 0u, lmask = 0u;
    for (int k = 0; k < g.m; ++k) {
        if (g.link[k] & 0x80000000u) lmask |= 1u << g.slot[k];
        else imask |= 1u << g.slot[k];
    }
    o.imask = (uint8_t)imask;
    o.lmask = (uint8_t)lmask;
    // the leaf children's records
    for (int k = 0; k < g.m; ++k)
        if (g.link[k] & 0x80000000u) {
            const unsigned rank = (unsigned)__popc((imask | lmask) & ((1u << g.slot[k]) - 1u));
            const unsigned tri = (unsigned)keys[g.link[k] & 0x7fffffffu];
            const PtPrepTriangle t = prep[tri];
            PtLeafTri r;
            for (int a = 0; a < 3; ++a) { r.p1[a] = t.p1[a]; r.e1[a] = t.e1[a]; r.e2[a] = t.e2[a]; }
            r.index = tri;
            for (unsigned z = 0; z < sizeof r.pad / sizeof r.pad[0]; ++z) r.pad[z] = 0.0f;
            *reinterpret_cast<PtLeafTri*>(recs + base + rank) = r;
        }
    for (int a = 0; a < 3; ++a) {
        float lo = 3.0e38f, top = -3.0e38f;
        for (int k = 0; k < g.m; ++k) { lo = fminf(lo, g.lo[k][a]); top = fmaxf(top, g.hi[k][a]); }
        if (!(lo <= top)) { lo = grid->gmin[a]; top = lo; }
        // the origin: the grid position at or below the children's lower corner, checked by decoding as the traversal does
        const float gmin = grid->gmin[a], gstep = grid->gstep[a];
        float fq = floorf((lo - gmin) / gstep);
        fq = fminf(fmaxf(fq, 0.0f), 65535.0f);
        unsigned q16 = (unsigned)fq;
        while (q16 > 0u && pt_bvh_decode(q16, gstep, gmin) > lo) --q16;
        const float org = pt_bvh_decode(q16, gstep, gmin);  // (<= lo: the grid starts 2 eps below every box)
        o.org[a] = (uint16_t)q16;
        // smallest power of two `step` with decode(255) >= top; then every bound rounded outward and CHECKED by decoding
        int e2 = -126;
        const float ext = top - org;
        if (ext > 0.0f) (void)frexpf(ext / 255.0f, &e2);
        int be = e2 + 127;  // biased exponent of 2^e2
        be = be < 1 ? 1 : (be > 254 ? 254 : be);
        for (;;) {
            const float step = __uint_as_float((unsigned)be << 23);
            bool ok = org <= lo;
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
        o.ex[a] = (uint8_t)be;
    }
    recs[self] = o;
}


// ---- which binary nodes become eight-child nodes: chosen by dynamic programming (after Ylitie, Karras, Laine 2017, section 3) -------------------
// Which binary nodes become roots of eight-child nodes decides how many nodes a ray enters; a fixed rule (every third
// level: the round's first version) leaves a third of the nodes with two children: 55.7 nodes entered per ray of the
// 10^6-triangle soup against 53.3 here, 215 against 229 Msamples/s.  The cost of a subtree, with the probability of a
// ray entering a node taken as proportional to its box's surface area, is minimised exactly:
//   C(n, 1) = A(n) + D(n, 8)                 n becomes a node of its own, its descendants share 8 slots
//   D(n, j) = min over 0 < k < j of C(left, k) + C(right, j - k)        C(leaf, .) = 0
//   C(n, j) = min(D(n, j), C(n, j - 1))      the subtree of n occupies at most j slots of an ancestor's node
// bottom-up (one thread per leaf climbs, the second child to arrive fills the parent's table, as in the refit), then
// the nodes are laid out top-down, one launch per level of the new hierarchy: a thread takes a root and its new index,
// follows the recorded choices to its (at most eight) children, reserves consecutive indices for those that are
// nodes and consecutive records for those that are leaves, writes the node, and queues the node children.
struct PtCost8 { float c[7]; unsigned kk; unsigned ee; };  // c[j-1] = C(n, j); kk: 3 bits per j = 2..8: slots given to the left child;
                                                           // ee: 3 bits per j = 2..7: the j' <= j that C(n, j) really uses (1 = a node of its own)
__device__ __forceinline__ float pt_cost_of(const PtCost8* __restrict__ tab, unsigned link, int j)
{
    if (link & 0x80000000u) return 0.0f;
    return tab[link].c[(j > 7 ? 7 : j) - 1];
}

__global__ void pt_bvh8_cost_kernel(const PtBvhNode* __restrict__ nodes, const int* __restrict__ parent, int n, int* __restrict__ flags,
                                    PtCost8* tab)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int node = n - 1 + k;
    for (int guard = 0; guard < 80; ++guard) {
        const int p = parent[node];
        if (p < 0) return;
        __threadfence();
        if (atomicAdd(&flags[p], 1) == 0) return;  // the first child to arrive leaves; the second sees both tables
        __threadfence();
        const PtBvhNode w = nodes[p];
        float area = 0.0f;
        {
            float d[3];
            bool ok = true;
            for (int a = 0; a < 3; ++a) {
                const float lo = fminf(w.lmin[a], w.rmin[a]), hi = fmaxf(w.lmax[a], w.rmax[a]);
                d[a] = hi - lo;
                ok = ok && (lo <= hi);
            }
            if (ok) area = 2.0f * (d[0] * d[1] + d[1] * d[2] + d[0] * d[2]);
        }
        const volatile PtCost8* vt = tab;
        float cl[8], cr[8];  // C(child, j), j = 1..7
        for (int j = 1; j <= 7; ++j) {
            cl[j] = (w.link_l & 0x80000000u) ? 0.0f : vt[w.link_l].c[j - 1];
            cr[j] = (w.link_r & 0x80000000u) ? 0.0f : vt[w.link_r].c[j - 1];
        }
        PtCost8 t;
        t.kk = 0u; t.ee = 0u;
        float dist[9];
        for (int j = 2; j <= 8; ++j) {
            float best = 3.0e38f;
            int bk = 1;
            for (int kl = 1; kl < j; ++kl) {
                if (kl > 7 || j - kl > 7) continue;
                const float v = cl[kl] + cr[j - kl];
                if (v < best) { best = v; bk = kl; }
            }
            dist[j] = best;
            t.kk |= (unsigned)bk << (3 * (j - 2));
        }
        t.c[0] = area + dist[8];
        int eff = 1;
        for (int j = 2; j <= 7; ++j) {
            if (dist[j] <= t.c[j - 2]) { t.c[j - 1] = dist[j]; eff = j; }
            else t.c[j - 1] = t.c[j - 2];
            t.ee |= (unsigned)eff << (3 * (j - 2));
        }
        tab[p] = t;
        node = p;
    }
}

struct PtWork8 { int node; int idx; };  // binary node, index of the eight-child node it becomes

// counters[0] = next free record, in_count / out_count: the two frontiers' sizes
__global__ void pt_bvh8_topdown_kernel(const PtBvhNode* __restrict__ nodes, const PtCost8* __restrict__ tab, const PtWork8* __restrict__ in,
                                       const unsigned* __restrict__ in_count, PtWork8* __restrict__ outq, unsigned* __restrict__ out_count,
                                       unsigned* __restrict__ counters, const unsigned long long* __restrict__ keys,
                                       const PtPrepTriangle* __restrict__ prep, const PtBvhGrid* __restrict__ grid, PtBvh8Node* __restrict__ recs)
{
    const unsigned total = *in_count;
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const PtWork8 wk = in[t];
        PtGather8 g;
        g.m = 0;
        // follow the choices: (link, its box, budget) on a small stack; the budgets of a node's two children sum to its own
        unsigned st_link[16];
        float st_lo[16][3], st_hi[16][3];
        int st_j[16];
        int sp = 0;
        {
            const PtBvhNode w = nodes[wk.node];
            const int kl = (int)((tab[wk.node].kk >> (3 * 6)) & 7u);  // j = 8
            st_link[0] = w.link_r; st_j[0] = 8 - kl;
            st_link[1] = w.link_l; st_j[1] = kl;
            for (int a = 0; a < 3; ++a) { st_lo[0][a] = w.rmin[a]; st_hi[0][a] = w.rmax[a]; st_lo[1][a] = w.lmin[a]; st_hi[1][a] = w.lmax[a]; }
            sp = 2;
        }
        while (sp > 0) {
            --sp;
            const unsigned c = st_link[sp];
            const int j = st_j[sp];
            float lo[3], hi[3];
            bool present = true;
            for (int a = 0; a < 3; ++a) { lo[a] = st_lo[sp][a]; hi[a] = st_hi[sp][a]; present = present && (lo[a] <= hi[a]); }
            if (!present) continue;  // an empty subtree: dropped
            int eff = 1;
            if (!(c & 0x80000000u) && j > 1) eff = (int)((tab[c].ee >> (3 * ((j > 7 ? 7 : j) - 2))) & 7u);
            if ((c & 0x80000000u) || eff <= 1) {
                if (g.m < 8) {
                    g.link[g.m] = c;
                    for (int a = 0; a < 3; ++a) { g.lo[g.m][a] = lo[a]; g.hi[g.m][a] = hi[a]; }
                    ++g.m;
                }
                continue;
            }
            const PtBvhNode w = nodes[c];
            const int kl = (int)((tab[c].kk >> (3 * (eff - 2))) & 7u);
            if (sp + 2 <= 16) {
                st_link[sp] = w.link_r; st_j[sp] = eff - kl;
                for (int a = 0; a < 3; ++a) { st_lo[sp][a] = w.rmin[a]; st_hi[sp][a] = w.rmax[a]; }
                ++sp;
                st_link[sp] = w.link_l; st_j[sp] = kl;
                for (int a = 0; a < 3; ++a) { st_lo[sp][a] = w.lmin[a]; st_hi[sp][a] = w.lmax[a]; }
                ++sp;
            }
        }
        pt_bvh8_slots(g);
        unsigned n_int = 0u, cmask = 0u;
        for (int k = 0; k < g.m; ++k) { cmask |= 1u << g.slot[k]; if (!(g.link[k] & 0x80000000u)) ++n_int; }
        const unsigned base = g.m ? atomicAdd(&counters[0], (unsigned)g.m) : 0u;
        pt_bvh8_write(g, (unsigned)wk.idx, base, keys, prep, grid, recs);
        if (n_int) {
            unsigned qpos = atomicAdd(out_count, n_int);
            for (int k = 0; k < g.m; ++k)
                if (!(g.link[k] & 0x80000000u)) {
                    PtWork8 nw;
                    nw.node = (int)g.link[k];
                    nw.idx = (int)(base + (unsigned)__popc(cmask & ((1u << g.slot[k]) - 1u)));
                    outq[qpos++] = nw;
                }
        }
    }
}

__global__ void pt_bvh8_topdown_init_kernel(PtWork8* q, unsigned* counts, unsigned* counters)
{
    q[0].node = 0; q[0].idx = 0;
    counts[0] = 1u; counts[1] = 0u;
    counters[0] = 1u;  // record 0 is the root
    counters[1] = 0u;
}
__global__ void pt_bvh8_zero_kernel(unsigned* p) { *p = 0u; }

}  // namespace

hipError_t ptk_bvh_big_raw(const PtRawTriangle* raw, const int* bigidx, int nbig, PtRawTriangle* out, hipStream_t s)
{
    if (nbig <= 0) return hipSuccess;
    hipLaunchKernelGGL(pt_bvh_big_raw_kernel, dim3(1), dim3(PT_BVH_BIG_MAX), 0, s, raw, bigidx, nbig, out);
    return hipGetLastError();
}

size_t ptk_bvh_temp_bytes(int ntri)
{
    size_t cub = 0;
    unsigned long long* nullk = nullptr;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, cub, nullk, nullk, ntri, 0, 62);
    const size_t n = (size_t)ntri;
    // keys, sorted keys, parent[2n-1], right_child[n-1], flags[n-1], bounds[16], the fp32 nodes, cub temp, then the collapse's
    // cost tables (36 n), two frontiers (8 n together), counts and counters
    return 16 * n + 4 * (2 * n) + 4 * n + 4 * n + 64 + sizeof(PtBvhNode) * n + cub + 2048 + 36 * n + 16 * n + 256;
}

// work: 36 n bytes of cost tables, two frontiers of 8 n bytes, counts and counters
static hipError_t pt_bvh8_build_sah(const PtBvhNode* nodes, const int* parent, int nleaves, int* flags, const unsigned long long* sorted,
                                    const PtPrepTriangle* prep, const PtBvhGrid* grid, PtBvh8Node* recs, char* work, unsigned* used_dev, hipStream_t s)
{
    const size_t n = (size_t)nleaves;
    work = (char*)(((uintptr_t)work + 15) & ~(uintptr_t)15);
    PtCost8* tab = (PtCost8*)work; work += 36 * n;
    PtWork8* q0 = (PtWork8*)work; work += 8 * (n / 2 + 8);
    PtWork8* q1 = (PtWork8*)work; work += 8 * (n / 2 + 8);
    unsigned* counts = (unsigned*)work;      // [0], [1]: the two frontiers' sizes
    unsigned* counters = counts + 2;         // next node index, next leaf record
    hipError_t e = hipMemsetAsync(flags, 0, 4 * n, s);
    if (e != hipSuccess) return e;
    const dim3 blk(256), lgrd((nleaves + 255) / 256);
    hipLaunchKernelGGL(pt_bvh8_cost_kernel, lgrd, blk, 0, s, nodes, parent, nleaves, flags, tab);
    hipLaunchKernelGGL(pt_bvh8_topdown_init_kernel, dim3(1), dim3(1), 0, s, q0, counts, counters);
    // a radix tree over 64-bit keys is at most 64 levels deep, and every level of the new hierarchy takes at least one
    const dim3 tgrd(nleaves / 256 / 4 + 1 < 2048 ? nleaves / 256 / 4 + 1 : 2048);
    for (int lvl = 0; lvl < 64; ++lvl) {
        PtWork8* in = (lvl & 1) ? q1 : q0;
        PtWork8* outq = (lvl & 1) ? q0 : q1;
        hipLaunchKernelGGL(pt_bvh8_zero_kernel, dim3(1), dim3(1), 0, s, counts + ((lvl + 1) & 1));
        hipLaunchKernelGGL(pt_bvh8_topdown_kernel, tgrd, blk, 0, s, nodes, tab, in, counts + (lvl & 1), outq, counts + ((lvl + 1) & 1), counters, sorted,
                           prep, grid, recs);
    }
    if (used_dev) return hipMemcpyAsync(used_dev, counters, sizeof(unsigned), hipMemcpyDeviceToDevice, s);  // records in use (nodes + leaves)
    return hipGetLastError();
}

hipError_t ptk_bvh_build(const PtRawTriangle* raw, const PtPrepTriangle* prep, int ntri, PtBvh8Node* recs,
                         PtPrepTriangle* bigtab, int* bigidx, int* nbig_dev, PtBvhGrid* grid_dev, unsigned* used_dev, void* temp, size_t temp_bytes, hipStream_t s)
{
    if (ntri < 2) return hipErrorInvalidValue;  // callers use the hierarchy for ntri >= 2 only
    const size_t n = (size_t)ntri;
    char* p = (char*)temp;
    unsigned long long* keys = (unsigned long long*)p; p += 8 * n;
    unsigned long long* sorted = (unsigned long long*)p; p += 8 * n;
    int* parent = (int*)p; p += 4 * 2 * n;
    int* right_child = (int*)p; p += 4 * n;
    int* flags = (int*)p; p += 4 * n;
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
    const dim3 blk(256), grd((ntri + 255) / 256);
    hipLaunchKernelGGL(pt_bvh_bounds_kernel, grd, blk, 0, s, raw, ntri, bounds);
    hipLaunchKernelGGL(pt_bvh_threshold_kernel, dim3(1), dim3(1), 0, s, bounds);
    hipLaunchKernelGGL(pt_bvh_grid_kernel, dim3(1), dim3(1), 0, s, bounds, grid_dev);
    hipLaunchKernelGGL(pt_bvh_big_collect_kernel, grd, blk, 0, s, raw, ntri, bounds, bigidx);
    hipLaunchKernelGGL(pt_bvh_big_finish_kernel, dim3(1), dim3(1), 0, s, bounds, bigidx, prep, bigtab, nbig_dev);
    hipLaunchKernelGGL(pt_bvh_keys_kernel, grd, blk, 0, s, raw, ntri, bounds, keys);
    if ((e = hipcub::DeviceRadixSort::SortKeys(p, cub, keys, sorted, ntri, 0, 62, s)) != hipSuccess) return e;
    const int nleaves = ntri;  // one triangle per leaf; nleaves >= 2
    const dim3 lgrd((nleaves + 255) / 256);
    hipLaunchKernelGGL(pt_bvh_hierarchy_kernel, lgrd, blk, 0, s, sorted, nleaves, nodes, parent, right_child);
    hipLaunchKernelGGL(pt_bvh_refit_kernel, lgrd, blk, 0, s, raw, sorted, nleaves, bounds, nodes, parent, right_child, flags);
    // (the sort is done with its workspace by now: the collapse's tables live at the end of `temp`)
    return pt_bvh8_build_sah(nodes, parent, nleaves, flags, sorted, prep, grid_dev, recs, (char*)temp + temp_bytes - (36 * n + 16 * n + 256), used_dev, s);
}
