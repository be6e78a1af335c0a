/* xcd_pricing.c -- CPU pricing of an XCD-affine region scheme for the LBVH search (no GPU involved).
 *
 * Question: today every one of the MI355X's eight XCDs walks the WHOLE hierarchy of BASELINE configs[4]'s scene (10^6 small
 * triangles: 80 MB of 64-byte records against 8 x 4 MB of L2), measured 54.7 L2 requests and 23.6 L2 misses per ray
 * (profiles/r03/pmc_traffic_soup.json).  Would it pay to give every XCD one spatial region of the hierarchy -- its bottom levels
 * served by that XCD's L2 alone -- and hand a ray over (one 64-byte record through a global queue) whenever its search enters a
 * subtree another XCD owns?  Threshold set by the judge: build it only if misses + hand-overs come to <= 14 per ray.
 *
 * Model (stated so that the result can be argued with):
 *   * scene: the product's generator (tools/xcd_pricing.py dumps scene.make_soup(N)); triangles whose box is longer than 1/16 of
 *     the scene's longest side stay outside the hierarchy, as in csrc/pt_bvh.hip (searched by brute force from LDS: no misses);
 *   * hierarchy: Morton-sorted median-split binary tree collapsed three levels at a time into eight-child nodes; nodes and one-
 *     triangle leaves are 64-byte records of ONE array, a node's children consecutive (the product's layout, PtBvh8Node);
 *   * rays: secondary rays of a diffuse soup -- origin on a random small triangle, cosine-distributed direction about its normal,
 *     closest hit with a shrinking tmax, children entered front to back (87 % of the product's rays are such rays);
 *   * L2: per XCD 4 MiB, 128-byte lines, 16 ways, LRU; `inflight` rays per XCD advance one record access per turn, round robin
 *     (the product keeps ~41 000 lanes in flight per XCD: 32 CUs x 5 workgroups x 256);
 *   * baseline: rays dealt to XCDs in blocks of 256 (a workgroup), every XCD sees every record;
 *   * regions: the eight-child tree is cut at the shallowest level with >= 64 nodes; those subtrees are dealt to the 8 XCDs in Morton
 *     order in runs of equal leaf counts (8 contiguous regions); records above the cut are read by whoever needs them (they are
 *     few and hot); a ray whose next record lies in another XCD's subtree is handed over and continues there (its remaining
 *     stack travels with it).  Reported: misses per ray, hand-overs per ray, and their sum.
 * The absolute miss counts of a model like this are not the hardware's; the RATIO regions / baseline is what is priced, applied
 * to the measured 23.6.
 *
 * usage: xcd_pricing soup.bin nrays inflight
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float lo[3], hi[3]; } Box;
typedef struct { float p1[3], e1[3], e2[3]; } Tri;

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd64(void) { uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
static float rndf(void) { return (float)(rnd64() >> 40) * (1.0f / 16777216.0f); }

/* ---- binary tree over Morton-sorted triangles ---- */
static int ntri, *order;          /* order[k] = triangle of sorted position k */
static Tri* tris;
static Box* tbox;
static uint32_t* morton;
typedef struct { Box box; int left, right, lo, hi; } BNode;   /* leaves: left = -1, triangle order[lo] */
static BNode* bn; static int nbn;

static uint32_t expand(uint32_t v) { v &= 1023; v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v; }
static int cmp_m(const void* a, const void* b) { uint32_t x = morton[*(const int*)a], y = morton[*(const int*)b]; return x < y ? -1 : x > y ? 1 : (*(const int*)a - *(const int*)b); }
static Box merge(Box a, Box b) { for (int k = 0; k < 3; ++k) { if (b.lo[k] < a.lo[k]) a.lo[k] = b.lo[k]; if (b.hi[k] > a.hi[k]) a.hi[k] = b.hi[k]; } return a; }

static int build(int lo, int hi)   /* [lo, hi) of the sorted order */
{
    int id = nbn++;
    bn[id].lo = lo; bn[id].hi = hi;
    if (hi - lo == 1) { bn[id].left = bn[id].right = -1; bn[id].box = tbox[order[lo]]; return id; }
    /* split at the highest differing Morton bit (radix-tree split), median when the codes are equal */
    uint32_t a = morton[order[lo]], b = morton[order[hi - 1]];
    int mid = (lo + hi) / 2;
    if (a != b) {
        int bit = 31 - __builtin_clz(a ^ b);
        int l = lo, h = hi - 1;   /* first index whose code has `bit` set */
        while (l < h) { int m = (l + h) / 2; if ((morton[order[m]] >> bit) & 1u) h = m; else l = m + 1; }
        mid = l;
    }
    int L = build(lo, mid), R = build(mid, hi);
    bn[id].left = L; bn[id].right = R;
    bn[id].box = merge(bn[L].box, bn[R].box);
    return id;
}

/* ---- eight-child records ---- */
typedef struct { int nchild; int child[8]; Box cbox[8]; int base; int is_leaf, tri; int owner; } Rec;   /* owner: XCD, -1 = above the cut */
static Rec* rec; static int nrec;
static int* bn2rec;

static int gather(int b, int depth, int* out, int n)   /* up to 8 descendants of binary node b, three levels down */
{
    if (depth == 3 || bn[b].left < 0) { out[n++] = b; return n; }
    n = gather(bn[b].left, depth + 1, out, n);
    return gather(bn[b].right, depth + 1, out, n);
}

/* ---- ray / box / triangle ---- */
static int hit_box(const Box* b, const float o[3], const float inv[3], float tmax, float* tnear)
{
    float t0 = 0.0f, t1 = tmax;
    for (int k = 0; k < 3; ++k) {
        float a = (b->lo[k] - o[k]) * inv[k], c = (b->hi[k] - o[k]) * inv[k];
        if (a > c) { float t = a; a = c; c = t; }
        if (a > t0) t0 = a;
        if (c < t1) t1 = c;
    }
    *tnear = t0;
    return t0 <= t1;
}
static float hit_tri(const Tri* t, const float o[3], const float d[3], float tmax)
{
    float pv[3] = { d[1] * t->e2[2] - d[2] * t->e2[1], d[2] * t->e2[0] - d[0] * t->e2[2], d[0] * t->e2[1] - d[1] * t->e2[0] };
    float det = t->e1[0] * pv[0] + t->e1[1] * pv[1] + t->e1[2] * pv[2];
    if (det < 1e-8f) return tmax;
    float tv[3] = { o[0] - t->p1[0], o[1] - t->p1[1], o[2] - t->p1[2] };
    float u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) / det;
    if (u < 0.0f || u > 1.0f) return tmax;
    float qv[3] = { tv[1] * t->e1[2] - tv[2] * t->e1[1], tv[2] * t->e1[0] - tv[0] * t->e1[2], tv[0] * t->e1[1] - tv[1] * t->e1[0] };
    float v = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) / det;
    if (v < 0.0f || u + v > 1.0f) return tmax;
    float tt = (t->e2[0] * qv[0] + t->e2[1] * qv[1] + t->e2[2] * qv[2]) / det;
    return (tt > 0.0f && tt < tmax) ? tt : tmax;
}

/* ---- per-ray access streams ---- */
static int* stream; static long* sbeg; static long slen, scap;
static int* origin_region;   /* the region (XCD) that owns the triangle a ray starts on */
static void push_access(int r) { if (slen == scap) { scap = scap * 2; stream = realloc(stream, scap * sizeof(int)); } stream[slen++] = r; }

static void trace(const float o[3], const float d[3])
{
    float inv[3]; for (int k = 0; k < 3; ++k) inv[k] = 1.0f / (fabsf(d[k]) > 1e-20f ? d[k] : (d[k] < 0 ? -1e-20f : 1e-20f));
    float tmax = 1e20f;
    int stack[512]; float snear[512]; int sp = 0;
    stack[sp] = 0; snear[sp++] = 0.0f;
    while (sp) {
        --sp;
        int r = stack[sp];
        if (snear[sp] > tmax) continue;            /* (the product checks a group's entry distances again when it is popped) */
        push_access(r);
        if (rec[r].is_leaf) { tmax = hit_tri(&tris[rec[r].tri], o, d, tmax); continue; }
        /* children hit, pushed far to near so that the nearest is popped first */
        int idx[8]; float tn[8]; int n = 0;
        for (int c = 0; c < rec[r].nchild; ++c) { float t; if (hit_box(&rec[r].cbox[c], o, inv, tmax, &t)) { idx[n] = rec[r].base + c; tn[n++] = t; } }
        for (int i = 1; i < n; ++i) for (int j = i; j > 0 && tn[j] > tn[j - 1]; --j) { float t = tn[j]; tn[j] = tn[j - 1]; tn[j - 1] = t; int q = idx[j]; idx[j] = idx[j - 1]; idx[j - 1] = q; }
        for (int i = 0; i < n && sp < 512; ++i) { stack[sp] = idx[i]; snear[sp++] = tn[i]; }
    }
}

/* ---- L2 model ---- */
#define WAYS 16
typedef struct { int sets; uint32_t* tag; uint32_t* age; uint32_t clock; long hits, misses; } Cache;
static void cache_init(Cache* c, long bytes) { c->sets = (int)(bytes / 128 / WAYS); c->tag = malloc(sizeof(uint32_t) * c->sets * WAYS); c->age = calloc(c->sets * WAYS, sizeof(uint32_t)); memset(c->tag, 0xff, sizeof(uint32_t) * c->sets * WAYS); c->clock = 0; c->hits = c->misses = 0; }
static void cache_access(Cache* c, int record)
{
    uint32_t line = (uint32_t)record >> 1;           /* two 64-byte records per 128-byte line */
    uint32_t set = (line * 2654435761u >> 7) % (uint32_t)c->sets;
    uint32_t* t = c->tag + (size_t)set * WAYS; uint32_t* a = c->age + (size_t)set * WAYS;
    int victim = 0;
    ++c->clock;
    for (int w = 0; w < WAYS; ++w) { if (t[w] == line) { a[w] = c->clock; ++c->hits; return; } if (a[w] < a[victim]) victim = w; }
    t[victim] = line; a[victim] = c->clock; ++c->misses;
}

int main(int argc, char** argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s soup.bin nrays inflight\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb"); if (!f) { perror(argv[1]); return 1; }
    int nall; if (fread(&nall, 4, 1, f) != 1) return 1;
    float* raw = malloc((size_t)nall * 9 * sizeof(float));
    if (fread(raw, sizeof(float) * 9, nall, f) != (size_t)nall) return 1;
    fclose(f);
    const int nrays = atoi(argv[2]), inflight = atoi(argv[3]);
    /* scene box, big triangles out */
    Box sb = { { 1e30f, 1e30f, 1e30f }, { -1e30f, -1e30f, -1e30f } };
    Tri* all = malloc(sizeof(Tri) * nall); Box* allb = malloc(sizeof(Box) * nall);
    for (int i = 0; i < nall; ++i) {
        const float* p = raw + 9 * (size_t)i;
        for (int k = 0; k < 3; ++k) {
            all[i].p1[k] = p[k]; all[i].e1[k] = p[3 + k] - p[k]; all[i].e2[k] = p[6 + k] - p[k];
            float lo = fminf(p[k], fminf(p[3 + k], p[6 + k])), hi = fmaxf(p[k], fmaxf(p[3 + k], p[6 + k]));
            allb[i].lo[k] = lo; allb[i].hi[k] = hi;
            if (lo < sb.lo[k]) sb.lo[k] = lo;
            if (hi > sb.hi[k]) sb.hi[k] = hi;
        }
    }
    float side = 0; for (int k = 0; k < 3; ++k) if (sb.hi[k] - sb.lo[k] > side) side = sb.hi[k] - sb.lo[k];
    tris = malloc(sizeof(Tri) * nall); tbox = malloc(sizeof(Box) * nall); ntri = 0;
    for (int i = 0; i < nall; ++i) {
        float ext = 0; for (int k = 0; k < 3; ++k) if (allb[i].hi[k] - allb[i].lo[k] > ext) ext = allb[i].hi[k] - allb[i].lo[k];
        if (ext > side / 16.0f) continue;
        tris[ntri] = all[i]; tbox[ntri++] = allb[i];
    }
    printf("scene: %d triangles, %d outside the hierarchy (big), %d inside\n", nall, nall - ntri, ntri);
    morton = malloc(sizeof(uint32_t) * ntri); order = malloc(sizeof(int) * ntri);
    for (int i = 0; i < ntri; ++i) {
        uint32_t q[3];
        for (int k = 0; k < 3; ++k) { float c = 0.5f * (tbox[i].lo[k] + tbox[i].hi[k]); float u = (c - sb.lo[k]) / (sb.hi[k] - sb.lo[k]); q[k] = (uint32_t)fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f); }
        morton[i] = expand(q[0]) | (expand(q[1]) << 1) | (expand(q[2]) << 2);
        order[i] = i;
    }
    qsort(order, ntri, sizeof(int), cmp_m);
    bn = malloc(sizeof(BNode) * 2 * (size_t)ntri); nbn = 0;
    build(0, ntri);
    /* collapse: BFS over eight-child nodes; a node's children consecutive */
    rec = calloc(2 * (size_t)ntri, sizeof(Rec)); nrec = 1; bn2rec = malloc(sizeof(int) * nbn);
    int* queue = malloc(sizeof(int) * 2 * (size_t)ntri); int qh = 0, qt = 0;
    int* rec_bn = malloc(sizeof(int) * 2 * (size_t)ntri); int* rec_depth = calloc(2 * (size_t)ntri, sizeof(int));
    rec_bn[0] = 0; queue[qt++] = 0;
    while (qh < qt) {
        int r = queue[qh++], b = rec_bn[r];
        rec[r].owner = -1;
        if (bn[b].left < 0) { rec[r].is_leaf = 1; rec[r].tri = order[bn[b].lo]; continue; }
        int kids[8]; int n = gather(b, 0, kids, 0);
        rec[r].nchild = n; rec[r].base = nrec;
        for (int c = 0; c < n; ++c) { rec[r].cbox[c] = bn[kids[c]].box; rec_bn[nrec] = kids[c]; rec_depth[nrec] = rec_depth[r] + 1; queue[qt++] = nrec++; }
    }
    long nnodes = 0; for (int r = 0; r < nrec; ++r) nnodes += !rec[r].is_leaf;
    printf("hierarchy: %d records = %ld nodes (%.2f children each) + %d leaves, %.1f MB\n", nrec, nnodes, (double)(nrec - 1) / nnodes, ntri, nrec * 64.0 / 1e6);
    /* regions: cut at the shallowest depth with >= 64 node records; deal them in array (Morton) order to 8 XCDs by leaf count */
    int cut = 0; for (int dpt = 0;; ++dpt) { long n = 0; for (int r = 0; r < nrec; ++r) n += rec_depth[r] == dpt; if (n >= 64 || n == 0) { cut = dpt; break; } }
    long leaves_seen = 0; int ncut = 0;
    for (int r = 0; r < nrec; ++r) if (rec_depth[r] == cut) {
        int b = rec_bn[r]; long cnt = bn[b].hi - bn[b].lo;
        rec[r].owner = (int)((leaves_seen + cnt / 2) * 8 / ntri); if (rec[r].owner > 7) rec[r].owner = 7;
        leaves_seen += cnt; ++ncut;
    }
    for (int r = 0; r < nrec; ++r) if (rec_depth[r] > cut) { /* inherit: parents precede children in BFS order; find the parent by scanning is O(n^2): use the binary node's range */
        int b = rec_bn[r]; long mid = (bn[b].lo + bn[b].hi) / 2; rec[r].owner = (int)(mid * 8 / ntri); if (rec[r].owner > 7) rec[r].owner = 7; }
    /* (a record below the cut belongs to the region that holds the middle of its leaf range: identical to inheriting from its ancestor at the
       cut whenever that ancestor's range lies within one region, which the equal-count dealing above makes the rule) */
    long shared = 0; for (int r = 0; r < nrec; ++r) shared += rec_depth[r] < cut;
    printf("regions: cut at depth %d (%d subtrees), %ld records above the cut shared by all XCDs\n", cut, ncut, shared);

    /* the region of every triangle = the owner of its leaf record */
    int* tri_region = malloc(sizeof(int) * ntri);
    for (int r = 0; r < nrec; ++r) if (rec[r].is_leaf) tri_region[rec[r].tri] = rec[r].owner < 0 ? 0 : rec[r].owner;
    origin_region = malloc(sizeof(int) * nrays);
    /* rays */
    scap = (long)nrays * 80; stream = malloc(scap * sizeof(int)); sbeg = malloc(sizeof(long) * (nrays + 1)); slen = 0;
    double leaves_touched = 0;
    for (int i = 0; i < nrays; ++i) {
        sbeg[i] = slen;
        const int ti = (int)(rnd64() % (uint64_t)ntri);
        const Tri* t = &tris[ti];
        origin_region[i] = tri_region[ti];
        float a = rndf(), b2 = rndf(); if (a + b2 > 1.0f) { a = 1.0f - a; b2 = 1.0f - b2; }
        float n[3] = { t->e2[1] * t->e1[2] - t->e2[2] * t->e1[1], t->e2[2] * t->e1[0] - t->e2[0] * t->e1[2], t->e2[0] * t->e1[1] - t->e2[1] * t->e1[0] };
        float nl = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]); if (nl < 1e-20f) { --i; continue; }
        if (rnd64() & 1) nl = -nl;
        for (int k = 0; k < 3; ++k) n[k] /= nl;
        /* cosine-distributed direction about n */
        float phi = 6.2831853f * rndf(), s2 = rndf(), st = sqrtf(s2), ct = sqrtf(1.0f - s2);
        float ax[3] = { fabsf(n[0]) > 0.001f ? 0.0f : 1.0f, fabsf(n[0]) > 0.001f ? 1.0f : 0.0f, 0.0f };
        float tx[3] = { ax[1] * n[2] - ax[2] * n[1], ax[2] * n[0] - ax[0] * n[2], ax[0] * n[1] - ax[1] * n[0] };
        float tl = sqrtf(tx[0] * tx[0] + tx[1] * tx[1] + tx[2] * tx[2]); for (int k = 0; k < 3; ++k) tx[k] /= tl;
        float sx[3] = { n[1] * tx[2] - n[2] * tx[1], n[2] * tx[0] - n[0] * tx[2], n[0] * tx[1] - n[1] * tx[0] };
        float d[3], o[3];
        for (int k = 0; k < 3; ++k) { d[k] = sx[k] * cosf(phi) * st + tx[k] * sinf(phi) * st + n[k] * ct; o[k] = t->p1[k] + a * t->e1[k] + b2 * t->e2[k] + 0.01f * d[k]; }
        trace(o, d);
    }
    sbeg[nrays] = slen;
    for (long k = 0; k < slen; ++k) leaves_touched += rec[stream[k]].is_leaf;
    printf("rays: %d, %.1f records per ray (%.1f nodes + %.2f leaves)\n", nrays, (double)slen / nrays, (slen - leaves_touched) / nrays, leaves_touched / nrays);

    /* ---- baseline: rays dealt to XCDs in blocks of 256, every XCD walks everything ---- */
    {
        Cache c[8]; for (int x = 0; x < 8; ++x) cache_init(&c[x], 4l << 20);
        /* per XCD: a window of `inflight` rays, each advancing one access per turn */
        for (int x = 0; x < 8; ++x) {
            int* cur = malloc(sizeof(int) * inflight); long* pos = malloc(sizeof(long) * inflight);
            int next_block = x, n = 0, in_block = 0;
            #define NEXT_RAY(out) do { out = -1; while (next_block * 256 < nrays) { if (in_block < 256 && next_block * 256 + in_block < nrays) { out = next_block * 256 + in_block++; break; } next_block += 8; in_block = 0; } } while (0)
            for (; n < inflight; ++n) { int r; NEXT_RAY(r); if (r < 0) break; cur[n] = r; pos[n] = sbeg[r]; }
            int live = n;
            while (live > 0) {
                for (int k = 0; k < n; ++k) {
                    if (cur[k] < 0) continue;
                    if (pos[k] == sbeg[cur[k] + 1]) { int r; NEXT_RAY(r); cur[k] = r; if (r < 0) { --live; continue; } pos[k] = sbeg[r]; }
                    cache_access(&c[x], stream[pos[k]++]);
                }
            }
            free(cur); free(pos);
        }
        long h = 0, m = 0; for (int x = 0; x < 8; ++x) { h += c[x].hits; m += c[x].misses; }
        printf("baseline: %.2f L2 requests per ray, hit rate %.1f %%, %.2f misses per ray\n", (double)(h + m) / nrays, 100.0 * h / (h + m), (double)m / nrays);
        printf("BASE_MISSES %.4f\n", (double)m / nrays);
    }
    /* ---- regions: a ray is processed by the XCD that owns the record it needs; records above the cut by whoever holds the ray ---- */
    {
        Cache c[8]; for (int x = 0; x < 8; ++x) cache_init(&c[x], 4l << 20);
        long handovers = 0;
        /* every XCD has a FIFO of rays waiting for it; the `inflight` head entries advance one access per turn; a ray that needs another XCD's
           record goes to the tail of that XCD's FIFO */
        int* where = malloc(sizeof(int) * nrays); long* pos = malloc(sizeof(long) * nrays);
        int** fifo = malloc(8 * sizeof(int*)); long fh[8], ft[8]; long cap = (long)nrays * 64 + 1024;
        for (int x = 0; x < 8; ++x) { fifo[x] = malloc(sizeof(int) * (size_t)(cap / 8 + nrays)); fh[x] = ft[x] = 0; }
        for (int r = 0; r < nrays; ++r) { pos[r] = sbeg[r]; where[r] = (r / 256) % 8; fifo[where[r]][ft[where[r]]++] = r; }
        long remaining = nrays;
        /* windows */
        int** win = malloc(8 * sizeof(int*)); int wn[8];
        for (int x = 0; x < 8; ++x) { win[x] = malloc(sizeof(int) * inflight); wn[x] = 0; }
        while (remaining > 0) {
            for (int x = 0; x < 8; ++x) {
                while (wn[x] < inflight && fh[x] < ft[x]) win[x][wn[x]++] = fifo[x][fh[x]++];
                int k = 0;
                while (k < wn[x]) {
                    int r = win[x][k];
                    if (pos[r] == sbeg[r + 1]) { --remaining; win[x][k] = win[x][--wn[x]]; continue; }
                    int rc = stream[pos[r]]; int own = rec[rc].owner;
                    if (own >= 0 && own != x) {   /* hand the ray over */
                        ++handovers;
                        if (ft[own] >= cap / 8 + nrays) { fprintf(stderr, "fifo overflow\n"); return 1; }
                        fifo[own][ft[own]++] = r; win[x][k] = win[x][--wn[x]]; continue;
                    }
                    cache_access(&c[x], rc); ++pos[r]; ++k;
                }
            }
        }
        long h = 0, m = 0; for (int x = 0; x < 8; ++x) { h += c[x].hits; m += c[x].misses; }
        printf("regions:  %.2f L2 requests per ray, hit rate %.1f %%, %.2f misses per ray, %.2f hand-overs per ray -> %.2f per ray together\n",
               (double)(h + m) / nrays, 100.0 * h / (h + m), (double)m / nrays, (double)handovers / nrays, (double)(m + handovers) / nrays);
        printf("REGION_MISSES %.4f\nREGION_HANDOVERS %.4f\n", (double)m / nrays, (double)handovers / nrays);
    }
    /* ---- origin-affine: a ray is traced, from start to end, by the XCD that owns the region it STARTS in (one hand-over per bounce, at the
       bounce: what a wavefront formulation with per-XCD ray queues between bounces gives; no hand-over inside a search) ---- */
    {
        Cache c[8]; for (int x = 0; x < 8; ++x) cache_init(&c[x], 4l << 20);
        for (int x = 0; x < 8; ++x) {
            int* mine = malloc(sizeof(int) * nrays); int nm = 0;
            for (int r = 0; r < nrays; ++r) if (origin_region[r] == x) mine[nm++] = r;
            int* cur = malloc(sizeof(int) * inflight); long* pos = malloc(sizeof(long) * inflight);
            int nxt = 0, n = 0;
            for (; n < inflight && nxt < nm; ++n) { cur[n] = mine[nxt++]; pos[n] = sbeg[cur[n]]; }
            int live = n;
            while (live > 0) {
                for (int k = 0; k < n; ++k) {
                    if (cur[k] < 0) continue;
                    if (pos[k] == sbeg[cur[k] + 1]) { if (nxt < nm) { cur[k] = mine[nxt++]; pos[k] = sbeg[cur[k]]; } else { cur[k] = -1; --live; continue; } }
                    cache_access(&c[x], stream[pos[k]++]);
                }
            }
            free(cur); free(pos); free(mine);
        }
        long h = 0, m = 0; for (int x = 0; x < 8; ++x) { h += c[x].hits; m += c[x].misses; }
        printf("origin-affine: %.2f L2 requests per ray, hit rate %.1f %%, %.2f misses per ray (+ 1 hand-over per ray by construction)\n", (double)(h + m) / nrays, 100.0 * h / (h + m), (double)m / nrays);
        printf("ORIGIN_MISSES %.4f\n", (double)m / nrays);
    }
    return 0;
}
