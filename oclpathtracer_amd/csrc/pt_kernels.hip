// pt_kernels.hip -- hand-written gfx950 kernels of the path-tracing hot path.
//
// What the reference runs as ONE OpenCL mega-kernel per frame
// (test/ClKernels/GenerateColors.cl:302-322: one work-item = one pixel = one path, then a
// read-modify-write of the gamma-encoded running mean) is restructured for CDNA4 as
//
//   pt_prep_kernel   once per scene upload: Triangle -> {p1, e1, e2, n, id}; for scenes made of
//                    quads also the slack and the packed table of the pass-1 filter
//   pt_trace_kernel  persistent waves; every LANE holds one path at a time.  A wave takes batches of consecutive
//                    samples off a global queue and keeps up to 64 PARKED paths in LDS: lanes whose path has
//                    ended take a parked one; when the pool is empty the wave parks its live paths and starts
//                    64 fresh samples -- 64 consecutive pixels, camera rays at full lane width, a coherent
//                    first bounce (pt_start_fresh, pt_pool_push / pt_pool_pop).  The closest-hit search is
//                    two-pass: pass 1 walks the triangles with a wave-uniform index (per-triangle constants are
//                    scalar loads consumed as SGPR operands) and keeps, per lane, a bit mask of the triangles
//                    that MAY pass the cull and u tests -- a conservative filter, in its strongest form one
//                    packed FMA chain deciding four triangles (pt_quad3_pass1; fresh primary rays read their
//                    masks from a per-pixel table instead: pt_primary_mask_kernel); pass 2 lets every lane run
//                    the exact reference test on its own survivors, fetched per lane from an LDS copy of the
//                    records, and balances the last ones across the wave's lanes (pt_tail_round).  Scenes of
//                    512 triangles or more walk an LBVH instead (pt_trace_bvh_body, pt_bvh_step, pt_bvh.hip).
//                    Path radiance goes to rad[frame][pixel] (three floats).
//   pt_fold_kernel   per pixel channel, in ascending frame order, replays the reference's
//                    gamma -> mean -> degamma arithmetic (GenerateColors.cl:314-321) over the
//                    staged radiances: bit-identical to frame-by-frame launches.
//
// The arithmetic is PTSPEC (pt_device_math.h); results are bit-identical to oracle/pt_oracle.c.
#include "pt_kernels.h"

#include "pt_device_math.h"

typedef const __attribute__((address_space(4))) float* pt_const_f32p;  // scalar (SMEM) loads

PTK_DEV unsigned pt_lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// number of set bits of m below this lane's position
PTK_DEV unsigned pt_mbcnt(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); }

// The trace kernels' argument block, RE-READ where it is used.  Passed by value, PtTraceParams lands in ~40 SGPRs at
// kernel entry and stays live through the bounce loop; at 7 waves per SIMD the compiler then spills SGPRs to VGPR
// lanes and pays v_readlane_b32 -- VALU issue slots, the resource this kernel is bound by -- in every bounce (24 per
// bounce before this).  The fields that only regeneration and shading need are instead loaded from the kernarg
// segment at their point of use: s_load on the scalar memory pipe, nothing live in between.  (The empty asm makes the
// pointer opaque so the loads are not hoisted back out of the loop.)
typedef const __attribute__((address_space(4))) PtTraceParams* pt_kargs_p;
PTK_DEV pt_kargs_p pt_kargs()
{
    pt_kargs_p k = (pt_kargs_p)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return k;
}
// LATE (template parameter of the functions below): read the argument where it is used (the table trace kernels), or take
// it from the by-value copy (the LBVH kernel, which has SGPRs to spare and measured 13 % slower with late reads)
#define PT_ARG(field) (LATE ? K->field : P.field)

// camera position, GenerateColors.cl:265
#define PT_EYE_X 0.0f
#define PT_EYE_Y 2.75f
#define PT_EYE_Z 4.0f

// ------------------------------------------------------------------------------------------
// scene preparation
// ------------------------------------------------------------------------------------------
__global__ void pt_prep_kernel(const PtRawTriangle* __restrict__ raw, PtPrepTriangle* __restrict__ out, int ntri,
                               unsigned int* __restrict__ det_bound_bits)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntri) return;
    f3 p1 = mk3(raw[i].p1[0], raw[i].p1[1], raw[i].p1[2]);
    f3 p2 = mk3(raw[i].p2[0], raw[i].p2[1], raw[i].p2[2]);
    f3 p3 = mk3(raw[i].p3[0], raw[i].p3[1], raw[i].p3[2]);
    f3 e1 = sub3(p2, p1);          // GenerateColors.cl:92
    f3 e2 = sub3(p3, p1);          // :93
    f3 n = cross3(e2, e1);         // :123
    PtPrepTriangle t;
    t.p1[0] = p1.x; t.p1[1] = p1.y; t.p1[2] = p1.z;
    t.e1[0] = e1.x; t.e1[1] = e1.y; t.e1[2] = e1.z;
    t.e2[0] = e2.x; t.e2[1] = e2.y; t.e2[2] = e2.z;
    t.pad0[0] = t.pad0[1] = t.pad0[2] = 0.0f;
    t.n[0] = n.x; t.n[1] = n.y; t.n[2] = n.z;
    t.id = raw[i].id;
    out[i] = t;
    // upper bound of |det| = |dot(e1, cross(dir, e2))| <= |e1| |e2| |dir| for this triangle, as
    // L1 norms; non-negative floats order like their bit patterns, NaN/Inf sort above all finite
    float b = (__builtin_fabsf(e1.x) + __builtin_fabsf(e1.y) + __builtin_fabsf(e1.z)) *
              (__builtin_fabsf(e2.x) + __builtin_fabsf(e2.y) + __builtin_fabsf(e2.z));
    atomicMax(&det_bound_bits[0], __float_as_uint(b) & 0x7fffffffu);
    // quad structure (pt_quad_pass1): triangle 2k+1 must have e2 == -e2 of triangle 2k.  Numeric
    // equality, so a zero of either sign matches; a NaN never does.  word 1 counts violations.
    if (i & 1) {
        f3 q1 = mk3(raw[i - 1].p1[0], raw[i - 1].p1[1], raw[i - 1].p1[2]);
        f3 q3 = mk3(raw[i - 1].p3[0], raw[i - 1].p3[1], raw[i - 1].p3[2]);
        f3 f2 = sub3(q3, q1);
        if (!(e2.x == -f2.x && e2.y == -f2.y && e2.z == -f2.z)) atomicAdd(&det_bound_bits[1], 1u);
        // (a,b,c),(c,d,a): this triangle starts at its predecessor's third vertex (pt_quad2_pass1)
        if (!(p1.x == q3.x && p1.y == q3.y && p1.z == q3.z)) atomicAdd(&det_bound_bits[3], 1u);
    }
    // scene radius about the camera position (GenerateColors.cl:265), for pt_quad2_pass1's error bound
    float r = 0.0f;
    const f3 vs[3] = { p1, p2, p3 };
    for (int k = 0; k < 3; ++k) {
        float ax = __builtin_fabsf(vs[k].x - PT_EYE_X), ay = __builtin_fabsf(vs[k].y - PT_EYE_Y), az = __builtin_fabsf(vs[k].z - PT_EYE_Z);
        r = !(ax <= r) ? ax : r;  // a NaN replaces r and then sticks (every later "<=" is false too)
        r = !(ay <= r) ? ay : r;
        r = !(az <= r) ? az : r;
    }
    atomicMax(&det_bound_bits[2], __float_as_uint(r) & 0x7fffffffu);
    // words 4, 5: a 64-bit checksum of the raw records (position-dependent mix per record, summed: order of arrival does not matter).
    // A buffer the caller can write behind the ABI is prepared again for every render (pt_shim.hip); the checksum tells whether that
    // changed anything, i.e. whether the LBVH and the primary-ray masks made from the previous contents still stand.
    {
        const unsigned* w = reinterpret_cast<const unsigned*>(raw + i);
        unsigned long long h = 0x9e3779b97f4a7c15ull * (unsigned long long)(i + 1);
        for (int k = 0; k < 16; ++k) {
            h ^= (unsigned long long)w[k] + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
            h *= 0xff51afd7ed558ccdull;
            h ^= h >> 33;
        }
        atomicAdd(reinterpret_cast<unsigned long long*>(det_bound_bits + 4), h);
    }
}

// Second pass of the scene preparation for quad mode 2 (see pt_quad2_pass1 for the derivation):
// record 2k+1 gets pad0[0] = delta3 = slack of the lower bound of its shared-u test.
//   delta2 = |e1' + e1|_2 |e2|_2 (how far the pair is from a parallelogram) + 128 u D^2
//   delta3 = delta2 * c + delta1
// every factor inflated by 0.1 % to cover the rounding of this very computation.
__global__ void pt_prep_quad_margins_kernel(PtPrepTriangle* __restrict__ out, int ntri, float diameter, float delta1)
{
    int i = 2 * (blockIdx.x * blockDim.x + threadIdx.x) + 1;
    if (i >= ntri) return;
    const PtPrepTriangle a = out[i - 1], b = out[i];
    float wx = b.e1[0] + a.e1[0], wy = b.e1[1] + a.e1[1], wz = b.e1[2] + a.e1[2];
    float wn = __builtin_sqrtf(wx * wx + wy * wy + wz * wz) * 1.001f;
    float en = __builtin_sqrtf(a.e2[0] * a.e2[0] + a.e2[1] * a.e2[1] + a.e2[2] * a.e2[2]) * 1.001f;
    float delta2 = wn * en * 1.001f + 128.0f * 5.9604645e-8f * diameter * diameter * 1.001f;
    float delta3 = (delta2 * 1.00001f + delta1) * 1.001f;
    out[i].pad0[0] = delta3;  // +Inf / NaN keep every second triangle of the pair: valid, merely slow
}

// Third pass of the scene preparation, quad mode 3 (pt_quad3_pass1): per PAIR of quads (2p, 2p+1)
// the operands of the packed pass-1 filter, interleaved {quad 2p, quad 2p+1} so that every one is
// an SGPR pair of a v_pk_fma_f32:
//   n' = cross(e2, e1) * 1.000002f   (det * c = dir . n')
//   e2, K = cross(e2, a - eye)       (un = e2 . ((o - eye) x dir) - dir . K)
//   dhi = delta3 + deltaD * c + deltaP (slack of the outer bound; -1 for the padding quad)
__global__ void pt_prep_p1tab_kernel(const PtPrepTriangle* __restrict__ tris, int ntri, float diameter, float* __restrict__ tab)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int nquads = ntri / 2;
    if (2 * p >= nquads) return;
    const float uD2 = 5.9604645e-8f * diameter * diameter;
    const float deltaP = 192.0f * uD2 * 1.001f, deltaD = 128.0f * uD2 * 1.001f;
    float* t = tab + (size_t)p * PT_P1_STRIDE;
    for (int h = 0; h < 2; ++h) {
        const int q = 2 * p + h;
        float v[10] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, -1.0f };  // padding: |un| = 0 > -1 fails
        if (q < nquads) {
            const PtPrepTriangle a = tris[2 * q], b = tris[2 * q + 1];
            const f3 e2 = mk3(a.e2[0], a.e2[1], a.e2[2]);
            const f3 ac = mk3(a.p1[0] - PT_EYE_X, a.p1[1] - PT_EYE_Y, a.p1[2] - PT_EYE_Z);
            const f3 K = cross3(e2, ac);
            v[0] = a.n[0] * 1.000002f; v[1] = a.n[1] * 1.000002f; v[2] = a.n[2] * 1.000002f;
            v[3] = e2.x; v[4] = e2.y; v[5] = e2.z;
            v[6] = K.x; v[7] = K.y; v[8] = K.z;
            v[9] = (b.pad0[0] + deltaD * 1.000002f + deltaP) * 1.001f;
        }
        for (int k = 0; k < 10; ++k) t[2 * k + h] = v[k];
    }
    for (int k = 20; k < PT_P1_STRIDE; ++k) t[k] = 0.0f;
}

// ------------------------------------------------------------------------------------------
// camera: GenerateColors.cl:73-87, 263-288
// ------------------------------------------------------------------------------------------
// the camera ray through image-plane position (x, y), in pixels (GenerateColors.cl:265-287 after the jitter).
// inv_w = 1.0f / (float)W, inv_h = 1.0f / (float)H, aspect = (float)W / (float)H are IEEE quotients of image constants:
// computed once on the host (PtTraceParams), the same bits as :266-267 evaluated per work-item.
// The camera basis of :270-276 is a constant of the reference (eye, center = eye + (0,0,-1), up = (0,1,0)); evaluated
// with PTSPEC's normalize / cross it is exactly
//     viewDir = (+0, +0, -1)     holDir = normalize(cross(viewDir, up)) = (1, -0, +0)     upDir = normalize(cross(holDir, viewDir)) = (+0, 1, +0)
// (every length is exactly 1; the signed zeros are those of the fma forms).  Using the constants keeps nine values out of
// registers for the kernel's lifetime; the expression below is unchanged, so the results are too.
PTK_DEV void pt_camera_ray(float x, float y, float inv_w, float inv_h, float aspect, f3& org, f3& dir_out)
{
    const float angle = PTK_TAN_HALF_FOV;
    const f3 eye = mk3(PT_EYE_X, PT_EYE_Y, PT_EYE_Z);
    const f3 viewDir = mk3(0.0f, 0.0f, -1.0f);
    const f3 holDir = mk3(1.0f, -0.0f, 0.0f);
    const f3 upDir = mk3(0.0f, 1.0f, 0.0f);

    x = (2.0f * ((x + 0.5f) * inv_w) - 1.0f) * angle * aspect;
    y = -(1.0f - 2.0f * ((y + 0.5f) * inv_h)) * angle;

    float my = -1.0f * y;
    f3 d = add3(add3(scale3(holDir, x), scale3(upDir, my)), viewDir);
    f3 dir = normalize3(d);
    f3 pointAimed = add3(eye, scale3(dir, 4.0f));
    org = eye;
    dir_out = normalize3(normalize3(sub3(pointAimed, eye)));  // :287 then getRay's own normalize (:75)
}

PTK_DEV void pt_generate_ray(int xc, int yc, float inv_w, float inv_h, float aspect, uint32_t& seed, f3& org, f3& dir_out)
{
    float x = (float)xc + pt_random_float(seed) - 0.5f;  // :278-279: two draws, x first
    float y = (float)yc + pt_random_float(seed) - 0.5f;
    pt_camera_ray(x, y, inv_w, inv_h, aspect, org, dir_out);
}

// ------------------------------------------------------------------------------------------
// primary-ray candidate masks (quad scenes of up to 64 triangles)
// ------------------------------------------------------------------------------------------
// The camera is fixed (eye, view direction: GenerateColors.cl:265-272), so what a pixel's primary rays can hit
// is a property of the pixel: every frame's ray goes through the pixel's footprint, jittered by less than half
// a pixel (:278-281).  For a primary ray M = (o - eye) x dir = 0 and pass 1's two forms (pt_quad3_pass1) are
// LINEAR in the direction: un(d) = -K . d, T(d) = n' . d + dhi.  Over the footprint the direction stays within
// eps of the centre ray's d_c in every component:
//     the unnormalised direction moves by at most h = |(angle aspect / W, angle / H)|_2 (hol, up orthonormal),
//     its length is >= 1, and radial projection onto the unit sphere from outside is 1-Lipschitz;
//     eps = 1.01 h + 4e-6 also covers the rounding of the reference's own ray set-up (three normalisations).
// Hence |un(d) - un(d_c)| <= eps |K|_1 =: rho_u and |T(d) - T(d_c)| <= eps |n'|_1 =: rho_T for every ray of the
// pixel, and a (ray, quad) pair pass 1 would keep -- |un| <= T, un >= lo (first triangle), un <= hi (second),
// each evaluated in binary32 within deltaP of the real value -- has |un(d_c)| <= T(d_c) + rho_u + rho_T + 4 deltaP
// and the matching one-sided bounds.  One thread per local pixel writes the two 32-bit chunk masks in pass 1's
// own bit order; a FRESH wave of primary rays then loads its masks instead of running pass 1 (which is a third
// of a bounce).  Everything downstream (the exact tests of pass 2) is unchanged, so the pixels are too;
// tools/validate_filter.py counts violations of the cached masks like those of any other filter.
struct PtMaskParams {
    const float* p1tab;
    uint2* out;
    int32_t width, height, ntri;
    int32_t stripe_rows, n_ranks, rank;
    uint32_t npix_local;
    float p1_lo, p1_hi;
};

__global__ void pt_primary_mask_kernel(const PtMaskParams P)
{
    const unsigned lp = blockIdx.x * blockDim.x + threadIdx.x;
    if (lp >= P.npix_local) return;
    const unsigned lr = lp / (unsigned)P.width, x = lp - lr * (unsigned)P.width;
    unsigned grow = lr;
    if (P.n_ranks > 1) {
        const unsigned sl = lr / (unsigned)P.stripe_rows;
        const unsigned within = lr - sl * (unsigned)P.stripe_rows;
        grow = (sl * (unsigned)P.n_ranks + (unsigned)P.rank) * (unsigned)P.stripe_rows + within;
    }
    f3 o, dc;
    pt_camera_ray((float)x, (float)grow, 1.0f / (float)P.width, 1.0f / (float)P.height, (float)P.width / (float)P.height, o, dc);  // the jitter's midpoint: xi = 0.5
    const float hx = PTK_TAN_HALF_FOV * ((float)P.width / (float)P.height) / (float)P.width;
    const float hy = PTK_TAN_HALF_FOV / (float)P.height;
    const float eps = __builtin_sqrtf(hx * hx + hy * hy) * 1.01f + 4e-6f;
    const float E = -4.0f * P.p1_lo;  // 4 deltaP
    const int nquads = P.ntri / 2;
    unsigned m[2] = { 0u, 0u };
    for (int q = 0; q < nquads; ++q) {
        const float* tp = P.p1tab + (size_t)(q >> 1) * PT_P1_STRIDE;  // nx ny nz e2x e2y e2z Kx Ky Kz dhi, {quad 2p, quad 2p+1}
        const int h = q & 1;
        const float nx = tp[0 + h], ny = tp[2 + h], nz = tp[4 + h];
        const float kx = tp[12 + h], ky = tp[14 + h], kz = tp[16 + h];
        const float dhi = tp[18 + h];
        const float Tc = pt_fma(dc.z, nz, pt_fma(dc.y, ny, dc.x * nx)) + dhi;
        const float uc = -pt_fma(dc.z, kz, pt_fma(dc.y, ky, dc.x * kx));
        const float rho_u = eps * (__builtin_fabsf(kx) + __builtin_fabsf(ky) + __builtin_fabsf(kz)) * 1.001f;
        const float rho_T = eps * (__builtin_fabsf(nx) + __builtin_fabsf(ny) + __builtin_fabsf(nz)) * 1.001f;
        // NaNs fail every comparison and are kept, as in pass 1
        const bool in = !(__builtin_fabsf(uc) > (Tc + rho_u + rho_T + E) * 1.001f);
        const bool fa = in & !(uc + (rho_u + E) * 1.001f < P.p1_lo);
        const bool fb = in & !(uc - (rho_u + E) * 1.001f > P.p1_hi);
        const int j = 2 * q, c = j >> 5;
        const int nc = P.ntri - 32 * c < 32 ? P.ntri - 32 * c : 32;
        m[c] |= (fa ? 1u : 0u) << (nc - 1 - (j & 31));
        m[c] |= (fb ? 1u : 0u) << (nc - 2 - (j & 31));
    }
    P.out[lp] = make_uint2(m[0], m[1]);
}

// ------------------------------------------------------------------------------------------
// intersectTriangle (GenerateColors.cl:89-135) against a wave-uniform triangle record
// ------------------------------------------------------------------------------------------
struct PtTriRec { float p1x, p1y, p1z, e1x, e1y, e1z, e2x, e2y, e2z; };

PTK_DEV PtTriRec pt_load_tri(pt_const_f32p T, int i)
{
    pt_const_f32p t = T + 16 * i;  // constant address space + uniform index -> s_load
    PtTriRec r;
    r.p1x = t[0]; r.p1y = t[1]; r.p1z = t[2];
    r.e1x = t[3]; r.e1y = t[4]; r.e1z = t[5];
    r.e2x = t[6]; r.e2y = t[7]; r.e2z = t[8];
    return r;
}

// DET_BOUNDED: the host has verified |e1|*|e2| <= 2e19 for every triangle, so det <= 1e20 and
// the short exact reciprocal applies to every front-facing triangle.
#ifndef PT_VALIDATE_FILTER
#define PT_VALIDATE_FILTER 0  // diagnostic: check the pass-1 filter against the reference predicate
#endif
// ---- two-pass closest hit -------------------------------------------------------------------------
// SIMT executes all 44 instructions of the flat test for every lane, but only 9 % of the
// (ray, triangle) pairs get past the u test (:109) -- 50 % are culled at :100, 41 % fail :109.
// Pass 1 (wave-uniform triangle, SGPR operands, 24 VALU): det, 1/det, u and the predicate
// "passes :100 and :109", recorded as one bit per triangle in a per-lane mask.
// Pass 2 (per lane): each lane walks ITS surviving triangles in ascending index and runs the rest
// of the test on them (record fetched with per-lane vector loads, L1-resident); the wave iterates
// max-over-lanes(#survivors) times, ~8 for the Cornell box instead of 36.
// Exactness: a pair that fails :100 or :109 can never be accepted, the survivors are tested with
// the same operations on the same operands, in the same (ascending) order, against the same
// running tmax -- the accepted (t, index) are those of the one-pass loop bit for bit.
// Survivor masks are built MSB-first: m = 2 m + flag is ONE v_addc_co_u32 whose carry-in is the
// comparison's own lane mask (against v_cndmask + v_lshl_or per flag).  After the n flags of a
// chunk, triangle j of the chunk sits at bit n-1-j: pass 2 walks the mask from its highest bit.
// Flags travel as the comparisons' lane masks (ballot of a single compare IS the v_cmp result;
// the conjunction is then an s_and_b64), never as per-lane booleans.
typedef unsigned long long pt_lanes;
#define PT_LANES(cond) __builtin_amdgcn_ballot_w64(cond)
PTK_DEV unsigned pt_push_flag(unsigned m, pt_lanes c)
{
    unsigned long long carry_out;
    unsigned r;
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(r), "=s"(carry_out) : "v"(m), "s"(c));
    return r;
}

template <bool DET_BOUNDED>
PTK_DEV pt_lanes pt_tri_pass1(const PtTriRec& r, const f3& o, const f3& d)
{
    float pvx = pt_fma(d.y, r.e2z, -(d.z * r.e2y));
    float pvy = pt_fma(d.z, r.e2x, -(d.x * r.e2z));
    float pvz = pt_fma(d.x, r.e2y, -(d.y * r.e2x));
    float det = pt_fma(r.e1z, pvz, pt_fma(r.e1y, pvy, r.e1x * pvx));
    float tvx = o.x - r.p1x, tvy = o.y - r.p1y, tvz = o.z - r.p1z;
    float un = pt_fma(tvz, pvz, pt_fma(tvy, pvy, tvx * pvx));  // u = un * RN(1/det)
    pt_lanes ok;
    if (DET_BOUNDED) {
        // Pass 2 re-applies :100 and :109 exactly, so pass 1 only has to keep a SUPERSET of the
        // pairs that pass them -- without the reciprocal (a quarter-rate instruction + 2 fma).
        // With det in [1e-8, 1e20] (DET_BOUNDED) and inv = RN(1/det) in [1e-20, 1e8]:
        //   un < -1e-24        =>  un*inv <= -1e-44, rounds to a negative non-zero  =>  u < 0
        //   un > det*1.000001f =>  un*inv >= 1.000001*(1-2^-24)^2 > 1 + 8e-7        =>  u > 1
        // A NaN in det or un fails every "<"/">" here and stays in the mask, as it passes
        // :100/:109 in the reference.  The cull test (:100) itself is left to pass 2: a pair with
        // det < 1e-8 survives these two bounds only for det in [-1e-24, 1e-8), which is rare.
        ok = PT_LANES(!(un < -1e-24f)) & PT_LANES(!(un > det * 1.000001f));
    } else {
        float u = un * (1.0f / det);
        ok = PT_LANES(!(det < 1e-8f)) & PT_LANES(!(u < 0.0f)) & PT_LANES(!(u > 1.0f));  // :100, :109
    }
    return ok;
}

// dynamic LDS of the trace kernels: the workgroup's copy of the hot triangle records
extern __shared__ __attribute__((aligned(16))) float pt_lds_tab[];

// record i for pass 2.  LDS_TABLE selects where the records of a scene live for the per-lane fetches:
//   1  the whole prepared table is copied to LDS once per workgroup (scenes up to PT_LDS_TRI_MAX triangles);
//      i = triangle index, lds_off = 0
//   2  TILED: a larger scene searched by brute force streams the records of the current 32-triangle chunk into a
//      per-wave LDS tile when many lanes have survivors in it (pt_stage_tile); i = index inside the chunk,
//      lds_off = the wave's tile
//   0  per-lane loads from the prepared table in global memory (stride 16 dwords); i = triangle index
// (LDS: stride PT_LDS_TRI_STRIDE dwords, 32-bit LDS addressing; per-lane global loads of a small table saturate
// the CU's vector-memory address path: every lane is its own cache line.)
template <int LDS_TABLE>
PTK_DEV PtTriRec pt_fetch_rec(const PtPrepTriangle* tris, int i, unsigned lds_off = 0u)
{
    float4 q0, q1;
    float e2z;
    if (LDS_TABLE) {
        // 32-bit LDS addressing: through the generic pointer hipcc forms the address with a 64-bit
        // v_mad_u64_u32 per survivor
        typedef __attribute__((address_space(3))) const float pt_lds_f32;
        typedef float pt_v4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) const pt_v4 pt_lds_f32x4;
        pt_lds_f32* t = (pt_lds_f32*)pt_lds_tab + lds_off + __umul24((unsigned)i, (unsigned)PT_LDS_TRI_STRIDE);  // i <= 256
        const pt_v4 a = *(pt_lds_f32x4*)t, b = *(pt_lds_f32x4*)(t + 4);
        q0 = make_float4(a.x, a.y, a.z, a.w);
        q1 = make_float4(b.x, b.y, b.z, b.w);
        e2z = t[8];
    } else {
        const float* t = reinterpret_cast<const float*>(tris + i);
        q0 = *reinterpret_cast<const float4*>(t);
        q1 = *reinterpret_cast<const float4*>(t + 4);
        e2z = t[8];
    }
    PtTriRec r;
    r.p1x = q0.x; r.p1y = q0.y; r.p1z = q0.z;  // p1.xyz e1.x | e1.yz e2.xy | e2.z
    r.e1x = q0.w; r.e1y = q1.x; r.e1z = q1.y;
    r.e2x = q1.z; r.e2y = q1.w; r.e2z = e2z;
    return r;
}

// TILED mode: the records of chunk [base, base + n) into this wave's LDS tile (12 of each record's 16 dwords)
PTK_DEV void pt_stage_tile(const PtPrepTriangle* tris, int base, int n, unsigned tile_off, unsigned lane)
{
    const float* g = reinterpret_cast<const float*>(tris + base);
    for (unsigned k = lane; k < (unsigned)n * PT_LDS_TRI_STRIDE; k += 64u) {
        const unsigned tri = k / PT_LDS_TRI_STRIDE, w = k - tri * PT_LDS_TRI_STRIDE;
        pt_lds_tab[tile_off + k] = g[tri * 16u + w];
    }
    // the wave's own LDS writes, read back by other lanes of the same wave: program order suffices for the
    // hardware (one in-order LDS queue per wave); this keeps the compiler from reordering
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool DET_BOUNDED>
PTK_DEV void pt_tri_pass2(const PtTriRec& r, int i, bool valid, const f3& o, const f3& d, float& tmax, float& hu,
                          float& hv, int& hidx)
{
    float pvx = pt_fma(d.y, r.e2z, -(d.z * r.e2y));
    float pvy = pt_fma(d.z, r.e2x, -(d.x * r.e2z));
    float pvz = pt_fma(d.x, r.e2y, -(d.y * r.e2x));
    float det = pt_fma(r.e1z, pvz, pt_fma(r.e1y, pvy, r.e1x * pvx));
    float inv_det = DET_BOUNDED ? pt_rcp_fast(det) : 1.0f / det;
    float tvx = o.x - r.p1x, tvy = o.y - r.p1y, tvz = o.z - r.p1z;
    float u = pt_fma(tvz, pvz, pt_fma(tvy, pvy, tvx * pvx)) * inv_det;
    // :100, :109 (pass 1 kept a superset).  `u > 1.0f` of :109 needs no compare of its own here: with
    // v >= 0 (or -0) round-to-nearest gives u + v >= u > 1, which :117 rejects below; with v NaN the
    // same NaN reaches t (through qvec or inv_det) and :125 rejects.  The accepted set is unchanged.
    bool ok = valid & !(det < 1e-8f) & !(u < 0.0f);
    float qvx = pt_fma(tvy, r.e1z, -(tvz * r.e1y));
    float qvy = pt_fma(tvz, r.e1x, -(tvx * r.e1z));
    float qvz = pt_fma(tvx, r.e1y, -(tvy * r.e1x));
    float v = pt_fma(d.z, qvz, pt_fma(d.y, qvy, d.x * qvx)) * inv_det;
    ok &= !(v < 0.0f) & !(u + v > 1.0f);  // :117
    float tt = pt_fma(r.e2z, qvz, pt_fma(r.e2y, qvy, r.e2x * qvx)) * inv_det;
    ok &= (tt > 0.0f) & (tt < tmax);  // :125
    tmax = ok ? tt : tmax;
    hu = ok ? u : hu;  // pass 2 runs ~8 times per ray: carrying (u,v) here is cheaper than pt_hit_uv
    hv = ok ? v : hv;
    hidx = ok ? i : hidx;
}

// ---- pass 2, balanced across lanes: the TAIL ---------------------------------------------------------
// Survivors per ray: 3.3 on average, 8.3 at the wave's slowest lane -- walking every lane's own survivors
// to the end ran the exact test at 39 % lane efficiency.  The closest hit is order-free: the winner of the
// reference's ascending loop with its strict `t < tmax` (:125,:145-151) is argmin (t, index) over the
// triangles that pass every other test.  So each lane walks its own survivors only while MANY lanes still
// have one (PT_TAIL_LANES); what is left then -- a few survivors in a few lanes -- is appended as
// (triangle, ray lane) pairs to a per-wave list in LDS, and the pairs are tested 64 at a time, one pair per
// lane whoever owns the ray: the ray travels by ds_bpermute, the candidate's key (t bits, index) goes to the
// ray's slot with one 64-bit ds_min.  At the end of the search a ray whose slot holds a better key than its
// own walk found re-runs the exact test on that one triangle (same operations on the same operands: the
// same t, u, v bit for bit).  The list outlives the 32-triangle chunks of a large scene, so a brute-force
// search over thousands of triangles runs its rare survivors at full lane width too.
#ifndef PT_TAIL_LANES
#define PT_TAIL_LANES 32  // own steps continue while more lanes than this still hold a survivor; 0 = never use the tail
#endif
#define PT_TAIL_LIST 128u  // list capacity: a round is run as soon as 64 pairs are pending, one append adds <= 64

typedef __attribute__((address_space(3))) unsigned pt_lds_u32;
typedef __attribute__((address_space(3))) unsigned long long pt_lds_u64;
struct PtTail {
    pt_lds_u64* keys;  // [64]  best (t bits << 32 | triangle) the tail found for the ray of each lane; ~0 = none
    pt_lds_u32* list;  // [PT_TAIL_LIST]  pending pairs: triangle << 6 | ray lane  (ring)
    unsigned wr, rd;   // wave-uniform ring positions
    unsigned tile;     // TILED mode (pt_fetch_rec): dword offset of this wave's record tile in LDS
    // per lane (as the owner of a ray): the best key its slot has held so far and the (u, v) that came with it
    unsigned long long kbest;
    float ku, kv;
};

// a pending pair = triangle << 6 | ray lane: 2 bytes when the whole scene sits in the LDS table (<= 256 triangles: 14 bits),
// 4 bytes otherwise
typedef __attribute__((address_space(3))) unsigned short pt_lds_u16;
template <int LDS_TABLE> PTK_DEV unsigned pt_tail_get(const PtTail& tl, unsigned i)
{
    return LDS_TABLE == 1 ? (unsigned)((pt_lds_u16*)tl.list)[i] : tl.list[i];
}
template <int LDS_TABLE> PTK_DEV void pt_tail_put(const PtTail& tl, unsigned i, unsigned pair)
{
    if (LDS_TABLE == 1) ((pt_lds_u16*)tl.list)[i] = (unsigned short)pair;
    else tl.list[i] = pair;
}

// the reference's test of one (ray, triangle) pair without the running tmax: passes :100,:109,:117 and 0 < t < 1e20
template <bool DET_BOUNDED>
PTK_DEV bool pt_tri_candidate(const PtTriRec& r, const f3& o, const f3& d, float& t_out, float& u_out, float& v_out)
{
    float pvx = pt_fma(d.y, r.e2z, -(d.z * r.e2y));
    float pvy = pt_fma(d.z, r.e2x, -(d.x * r.e2z));
    float pvz = pt_fma(d.x, r.e2y, -(d.y * r.e2x));
    float det = pt_fma(r.e1z, pvz, pt_fma(r.e1y, pvy, r.e1x * pvx));
    float inv_det = DET_BOUNDED ? pt_rcp_fast(det) : 1.0f / det;
    float tvx = o.x - r.p1x, tvy = o.y - r.p1y, tvz = o.z - r.p1z;
    float u = pt_fma(tvz, pvz, pt_fma(tvy, pvy, tvx * pvx)) * inv_det;
    bool ok = !(det < 1e-8f) & !(u < 0.0f);  // (u > 1 is covered by u + v > 1: see pt_tri_pass2)
    float qvx = pt_fma(tvy, r.e1z, -(tvz * r.e1y));
    float qvy = pt_fma(tvz, r.e1x, -(tvx * r.e1z));
    float qvz = pt_fma(tvx, r.e1y, -(tvy * r.e1x));
    float v = pt_fma(d.z, qvz, pt_fma(d.y, qvy, d.x * qvx)) * inv_det;
    ok &= !(v < 0.0f) & !(u + v > 1.0f);
    float tt = pt_fma(r.e2z, qvz, pt_fma(r.e2y, qvy, r.e2x * qvx)) * inv_det;
    ok &= (tt > 0.0f) & (tt < 1e20f);  // :125 against the initial tmax (:141)
    t_out = tt;
    u_out = u;
    v_out = v;
    return ok;
}

PTK_DEV float pt_from_lane(unsigned byte_addr, float v)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute((int)byte_addr, __float_as_int(v)));
}

// test the first `cnt` (<= 64) pending pairs, one per lane; every lane of the wave takes part (the rays travel by
// ds_bpermute, which needs their owners' lanes enabled)
template <bool DET_BOUNDED, int LDS_TABLE>
PTK_DEV void pt_tail_round(PtTail& tl, unsigned cnt, unsigned lane, const PtPrepTriangle* tris, const f3& o, const f3& d)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool act = lane < cnt;
    const unsigned e = act ? pt_tail_get<LDS_TABLE>(tl, (tl.rd + lane) & (PT_TAIL_LIST - 1u)) : 0u;
    const unsigned ray = e & 63u, tri = e >> 6;
    const unsigned a = ray << 2;
    const f3 po = mk3(pt_from_lane(a, o.x), pt_from_lane(a, o.y), pt_from_lane(a, o.z));
    const f3 pd = mk3(pt_from_lane(a, d.x), pt_from_lane(a, d.y), pt_from_lane(a, d.z));
    const PtTriRec r = pt_fetch_rec<(LDS_TABLE == 1 ? 1 : 0)>(tris, (int)tri);  // (a pending pair may be of an earlier chunk than the tile's)
    float t, u, v;
    const bool ok = pt_tri_candidate<DET_BOUNDED>(r, po, pd, t, u, v) & act;
    if (ok) {
        // key = t bits | triangle | the lane that tested the pair: the minimum is the reference's winner (t, then the
        // lower index; a ray's pairs have distinct triangles), and its low bits say where its (u, v) are
        const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)((tri << 6) | lane);
        __hip_atomic_fetch_min(tl.keys + ray, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    // every lane, now as the owner of its ray: did this round improve my slot?  Then fetch (u, v) from the lane that did it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned long long slot = tl.keys[lane];
    const unsigned from = ((unsigned)slot & 63u) << 2;
    const float pu = pt_from_lane(from, u), pv = pt_from_lane(from, v);
    const bool changed = slot != tl.kbest;
    tl.kbest = slot;
    tl.ku = changed ? pu : tl.ku;
    tl.kv = changed ? pv : tl.kv;
    tl.rd += cnt;
}

// pass 2 of one 32-triangle chunk whose survivor mask is m (bit n-1-j <-> triangle base + j)
template <bool DET_BOUNDED, int LDS_TABLE>
PTK_DEV void pt_pass2_chunk(unsigned m, int base, int n, const PtPrepTriangle* tris, const f3& o, const f3& d, float& tmax, float& hu,
                            float& hv, int& hidx, PtTail& tl, unsigned lane, unsigned& steps)
{
    // own steps: every lane tests its next survivor (index 0 and ok = false once it has none left),
    // while more than PT_TAIL_LANES lanes still hold one
    if (LDS_TABLE == 2 && (unsigned)__popcll(PT_LANES(m != 0u)) > (unsigned)PT_TAIL_LANES) pt_stage_tile(tris, base, n, tl.tile, lane);
    for (pt_lanes more = PT_LANES(m != 0u); (unsigned)__popcll(more) > (unsigned)PT_TAIL_LANES; more = PT_LANES(m != 0u)) {
        ++steps;
        const bool valid = m != 0u;
        unsigned lz;  // leading zeros: the highest bit is the lowest triangle index (-1 for m = 0)
        asm("v_ffbh_u32_e32 %0, %1" : "=v"(lz) : "v"(m));
        // (a lane without survivors forms a wild index: harmless for the LDS copies -- out-of-range
        // LDS reads return 0 -- and its result is discarded; the global table needs a real address)
        const int i = (LDS_TABLE != 0 || valid) ? base + n - 32 + (int)lz : base;
        m &= ~(0x80000000u >> (lz & 31u));
        const PtTriRec r = LDS_TABLE == 2 ? pt_fetch_rec<2>(tris, n - 32 + (int)lz, tl.tile) : pt_fetch_rec<LDS_TABLE>(tris, i);
        pt_tri_pass2<DET_BOUNDED>(r, i, valid, o, d, tmax, hu, hv, hidx);
    }
    // the rest of this chunk's survivors join the wave's pending pairs
    if (PT_TAIL_LANES > 0) {
        for (pt_lanes has = PT_LANES(m != 0u); has != 0ull; has = PT_LANES(m != 0u)) {
            if (m != 0u) {
                unsigned lz;
                asm("v_ffbh_u32_e32 %0, %1" : "=v"(lz) : "v"(m));
                m &= ~(0x80000000u >> (lz & 31u));
                const unsigned tri = (unsigned)(base + n - 32) + lz;
                pt_tail_put<LDS_TABLE>(tl, (tl.wr + pt_mbcnt(has)) & (PT_TAIL_LIST - 1u), (tri << 6) | lane);
            }
            tl.wr += (unsigned)__popcll(has);
            if (tl.wr - tl.rd >= 64u) {
                ++steps;
                pt_tail_round<DET_BOUNDED, LDS_TABLE>(tl, 64u, lane, tris, o, d);
            }
        }
    }
}

// end of a search: the pending pairs, then the merge of what the tail found
template <bool DET_BOUNDED, int LDS_TABLE>
PTK_DEV void pt_pass2_finish(const PtPrepTriangle* tris, const f3& o, const f3& d, float& tmax, float& hu, float& hv, int& hidx,
                             PtTail& tl, unsigned lane, unsigned& steps)
{
    if (PT_TAIL_LANES > 0) {
        if (tl.wr != tl.rd) {
            ++steps;
            pt_tail_round<DET_BOUNDED, LDS_TABLE>(tl, tl.wr - tl.rd, lane, tris, o, d);
        }
        if (tl.wr != 0u) {  // (wave-uniform) this search used the tail: merge what it found
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // (the constant is made here: hoisted out of the bounce loop it was the register pair the 64-VGPR kernel spilled,
            // and its reload from scratch waited for every store in flight)
            unsigned long long ones;
            asm volatile("v_mov_b64 %0, -1" : "=v"(ones));
            tl.keys[lane] = ones;
            const unsigned long long key = tl.kbest;
            const float kt = __uint_as_float((unsigned)(key >> 32));
            const int ki = (int)((unsigned)key >> 6);
            // the reference's winner is the lexicographic minimum of (t, index)
            const bool better = (key != ~0ull) & ((kt < tmax) | ((kt == tmax) & (ki < hidx)));
            tmax = better ? kt : tmax;
            hu = better ? tl.ku : hu;
            hv = better ? tl.kv : hv;
            hidx = better ? ki : hidx;
        }
    }
}

// closest hit over triangles [0, ntri): chunks of 32 triangles, pass 1 then pass 2 per chunk.
// (Software-pipelining pass 2 -- fetching the next survivor's record during the current test --
// was measured slower: 64.8 ms against 61.1 ms; the register copies cost more than the LDS latency
// that 7 waves per SIMD already hide.)
// Pass 1 for a quad (a,b,c),(c,d,a) in MODE 2: ONE u numerator decides both triangles.
//
// In A's barycentric frame the second triangle B is the strip u in [-1, 0]: with p1' = c,
// e2' = -e2 (so pvec' = -pvec bit for bit) the reference's own quantities for B are
//     un'  = -fl_dot(fl(o - c), pvec)        det' = -fl_dot(e1', pvec)
// and, were the arithmetic exact and the pair a parallelogram (e1' = -e1), un' = -unA and
// det' = detA.  In binary32 the two differ by rounding and by how far the pair is from a
// parallelogram.  With u = 2^-24, w = e1' + e1, every |coordinate difference| <= D and
// |dir|_2^2 <= 1.001 (both CHECKED per ray by the caller: a ray that violates them keeps every
// triangle), writing pvec = dir x e2 + zeta:
//   |un' + unA| <= |e2.zeta| + |(sigma + rho).pvec| + |eta_a| + |eta_b|
//               <= (9.3 + 6 + 12 + 36) u D^2 < 64 u D^2,          delta1 := 128 u D^2
//       (zeta: rounding of the cross product, <= 3.1 u D per component; sigma = (c - a) - fl(c - a);
//        rho: rounding of o - a and o - c; eta: rounding of a 3-term fma dot, <= 3 u sum|x_i p_i|)
//   |det' - detA| <= |w.pvec| + |eta| + |eta'| <= |w|_2 |e2|_2 1.001 + (18.6 + 36) u D^2 =: delta2
// A pair the reference accepts for B has det' >= 1e-8 and 0 <= u' <= 1, hence (pt_tri_pass1)
// -1e-24 <= un' <= det' * 1.000001, hence
//     unA <= delta1 + 1e-24      and      unA >= -(detA * 1.000001 + delta2 * 1.000001 + delta1).
// The filter below keeps a superset of that (m = fl(detA * 1.000002f) >= detA * 1.000001 for
// detA >= 0; for detA < 0 acceptance needs |detA| <= delta2, where the 0.1 % inflation of delta3
// dominates the 1e-6 relative difference).  delta3 = delta2 * c + delta1 comes prepared per pair
// (pt_prep_quad_margins_kernel).  A NaN anywhere fails every comparison and is kept.
// 25 VALU per quad instead of 34.  tools/validate_filter.py re-checks every dropped pair against
// the literal reference predicate (profiles/r01/filter_validation.txt).
// Pass 1 in MODE 3: the shared-u filter of mode 2 evaluated in Pluecker form, two quads per
// instruction.  tools/ubench_issue (profiles/r01/ubench_issue.log): on gfx950 an fp32 FMA/MUL/ADD
// whose operands are all VGPRs issues at double rate (~2.7 cycles per wave), ANY SGPR operand
// makes it single rate (~4.5), and v_pk_fma_f32 costs ~4.9 with or without an SGPR pair.  Mode 2
// spends 16 single-rate instructions per quad on SGPR operands; here
//     un  = tvec . (dir x e2) = e2 . ((o - eye) x dir) - dir . K,     K  = e2 x (a - eye)
//     T   = det * c + dhi     = dir . n' + dhi,                       n' = (e2 x e1) * c
// with M = (o - eye) x dir computed once per ray: 9 v_pk_fma_f32 give un and T of TWO quads, whose
// per-quad constants come interleaved from the table pt_prep_p1tab_kernel wrote.
// These are other roundings of the same real numbers than the reference's, so BOTH triangles now
// need slack (same assumptions as mode 2: D bounds every coordinate difference, |dir|^2 <= 1.001,
// u = 2^-24):
//   |un_here - un_ref|   <= 33.3 u D^2 (reference: tvec, cross, 3-term dot)
//                         + 51 u D^2 (here: o - eye, M, a - eye, K, 6-term fma chain)   < 192 u D^2 =: deltaP
//   |det_here - det_ref| <= 27.3 u D^2 + 27 u D^2 + 12 u D^2 (c folded into n', dhi into the chain) < 128 u D^2 =: deltaD
// A pair the reference accepts satisfies (pt_tri_pass1, pt_quad2_pass1)
//   first triangle:  -1e-24 <= un_ref <= det_ref c            second: -(det_ref c + delta3) <= un_ref <= delta1
// hence, with dhi = delta3 + deltaD c + deltaP >= deltaD c + deltaP,
//   both: |un_here| <= T;     first: un_here >= -deltaP (= lo);     second: un_here <= delta1 + deltaP (= hi).
// 9 packed + 6 compares + 4 v_addc per PAIR of quads (mode 2: 46).  NaNs fail every comparison and are kept.
typedef float pt_f2 __attribute__((ext_vector_type(2)));
// r = a.lo * s / a.hi * s / fma with the VGPR half broadcast to both results; s = SGPR pair
PTK_DEV pt_f2 pt_pk_mul_lo(pt_f2 a, pt_f2 s) { pt_f2 r; asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "s"(s)); return r; }
PTK_DEV pt_f2 pt_pk_mul_hi(pt_f2 a, pt_f2 s) { pt_f2 r; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "s"(s)); return r; }
PTK_DEV pt_f2 pt_pk_fma_lo(pt_f2 a, pt_f2 s, pt_f2 c) { pt_f2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "s"(s), "v"(c)); return r; }
PTK_DEV pt_f2 pt_pk_fma_hi(pt_f2 a, pt_f2 s, pt_f2 c) { pt_f2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "s"(s), "v"(c)); return r; }
PTK_DEV pt_f2 pt_pk_fnma_lo(pt_f2 a, pt_f2 s, pt_f2 c) { pt_f2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "s"(s), "v"(c)); return r; }
PTK_DEV pt_f2 pt_pk_fnma_hi(pt_f2 a, pt_f2 s, pt_f2 c) { pt_f2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "s"(s), "v"(c)); return r; }
// dhi + a.lo * s: the addend is the SGPR pair, the product's second factor too (one constant-bus
// operand only): so dhi is first copied to VGPRs by the first product instead -- see pt_quad3_pass1
struct PtRay3 { pt_f2 dxy, dzMx, Myz; };  // dir and M = (o - eye) x dir, as three register pairs

PTK_DEV void pt_quad3_pass1(pt_const_f32p t, const PtRay3& r, pt_f2& un, pt_f2& T)
{
    typedef const __attribute__((address_space(4))) pt_f2* pt_const_f2p;
    pt_const_f2p s = (pt_const_f2p)t;  // nx ny nz e2x e2y e2z Kx Ky Kz dhi
    pt_f2 dhi_v;
    { const pt_f2 dhi = s[9]; asm("v_pk_mul_f32 %0, %1, 1.0 op_sel_hi:[1,0]" : "=v"(dhi_v) : "s"(dhi)); }
    T = pt_pk_fma_lo(r.dxy, s[0], dhi_v);
    T = pt_pk_fma_hi(r.dxy, s[1], T);
    T = pt_pk_fma_lo(r.dzMx, s[2], T);
    un = pt_pk_mul_hi(r.dzMx, s[3]);
    un = pt_pk_fma_lo(r.Myz, s[4], un);
    un = pt_pk_fma_hi(r.Myz, s[5], un);
    un = pt_pk_fnma_lo(r.dxy, s[6], un);
    un = pt_pk_fnma_hi(r.dxy, s[7], un);
    un = pt_pk_fnma_lo(r.dzMx, s[8], un);
}

// PT_STAMPS=1 is a DIAGNOSTIC build (tools/stamps.py): per-phase s_memtime shares of a
// wave-bounce go to stats[2..5]; never shipped, never timed for the bench.
#ifndef PT_STAMPS
#define PT_STAMPS 0
#endif
#ifndef PT_LAUNCH_STAMPS
#define PT_LAUNCH_STAMPS 0   // DIAGNOSTIC build (tools/launch_stamps.py): start / first stop / last exit of one checkpointed launch
#endif
#if PT_STAMPS
#define PT_STAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PT_STAMP(var) do { } while (0)
#endif

// QUADS: 0 = independent triangles (pt_tri_pass1), 3 = packed shared-u filter (pt_quad3_pass1)
template <bool DET_BOUNDED, int LDS_TABLE, int QUADS>
PTK_DEV unsigned pt_intersect_two_pass(pt_const_f32p T, const PtPrepTriangle* tris, int ntri, const f3& o, const f3& d,
                                       bool alive, float& tmax, float& hu, float& hv, int& hidx,
                                       float delta1, float ray_radius, pt_const_f32p p1tab, float p1_lo, float p1_hi,
                                       PtTail tl, unsigned lane,
                                       unsigned long long* vstat = nullptr, unsigned long long* p1_ticks = nullptr)
{
    tl.wr = tl.rd = 0u;
    tl.kbest = ~0ull;
#if PT_STAMPS
    unsigned long long ta = 0, tb = 0;
#endif
    (void)vstat; (void)p1_ticks; (void)ray_radius; (void)p1tab; (void)p1_lo; (void)p1_hi;
    unsigned steps = 0;  // pass-2 iterations of this wave (diagnostics only)
    // the assumptions of the shared-u error bound (derivation above pt_quad3_pass1), checked for THIS ray
    bool tame = true;
    PtRay3 r3v;
    if (QUADS == 3 && DET_BOUNDED) {
        const f3 oc = mk3(o.x - PT_EYE_X, o.y - PT_EYE_Y, o.z - PT_EYE_Z);
        const f3 M = cross3(oc, d);
        r3v.dxy = pt_f2{ d.x, d.y }; r3v.dzMx = pt_f2{ d.z, M.x }; r3v.Myz = pt_f2{ M.y, M.z };
    }
    if (QUADS == 3 && DET_BOUNDED) {
        float dd = pt_fma(d.z, d.z, pt_fma(d.y, d.y, d.x * d.x));
        tame = (dd <= 1.001f) & (__builtin_fabsf(o.x - PT_EYE_X) <= ray_radius) &
               (__builtin_fabsf(o.y - PT_EYE_Y) <= ray_radius) & (__builtin_fabsf(o.z - PT_EYE_Z) <= ray_radius);
    }
    for (int base = 0; base < ntri; base += 32) {
        const int n = ntri - base < 32 ? ntri - base : 32;
        unsigned m = 0u;  // bit n-1-j <-> triangle base + j (pt_push_flag)
        PT_STAMP(ta);
        if (QUADS == 3 && DET_BOUNDED) {
            pt_const_f32p tp = p1tab + PT_P1_STRIDE * (base >> 2);  // base is a multiple of 32: 4 triangles per quad pair
            for (int j = 0; j < n; j += 4, tp += PT_P1_STRIDE) {
                pt_f2 un, th;
                pt_quad3_pass1(tp, r3v, un, th);
                {
                    const pt_lanes in = PT_LANES(!(__builtin_fabsf(un.x) > th.x));
                    m = pt_push_flag(pt_push_flag(m, in & PT_LANES(!(un.x < p1_lo))), in & PT_LANES(!(un.x > p1_hi)));
                }
                if (j + 2 < n) {
                    const pt_lanes in = PT_LANES(!(__builtin_fabsf(un.y) > th.y));
                    m = pt_push_flag(pt_push_flag(m, in & PT_LANES(!(un.y < p1_lo))), in & PT_LANES(!(un.y > p1_hi)));
                }
            }
            if (!tame) m = n == 32 ? ~0u : (1u << n) - 1u;
        } else {
            PtTriRec a = pt_load_tri(T, base);
            int j = 0;
            for (; j + 1 < n; j += 2) {
                PtTriRec b = pt_load_tri(T, base + j + 1);
                m = pt_push_flag(m, pt_tri_pass1<DET_BOUNDED>(a, o, d));
                a = pt_load_tri(T, base + (j + 2 < n ? j + 2 : j + 1));
                m = pt_push_flag(m, pt_tri_pass1<DET_BOUNDED>(b, o, d));
            }
            if (j < n) m = pt_push_flag(m, pt_tri_pass1<DET_BOUNDED>(a, o, d));
        }
#if PT_VALIDATE_FILTER
        // DIAGNOSTIC build (tools/validate_filter.py): the reference predicate of :100 and :109,
        // evaluated literally with IEEE division, must never accept a pair the filter dropped
        if (alive && vstat) {
            unsigned mx = 0u;
            for (int jj = 0; jj < n; ++jj) {
                const PtTriRec r = pt_load_tri(T, base + jj);
                float pvx = pt_fma(d.y, r.e2z, -(d.z * r.e2y));
                float pvy = pt_fma(d.z, r.e2x, -(d.x * r.e2z));
                float pvz = pt_fma(d.x, r.e2y, -(d.y * r.e2x));
                float det = pt_fma(r.e1z, pvz, pt_fma(r.e1y, pvy, r.e1x * pvx));
                bool keep = !(det < 1e-8f || -det > 1e-8f);
                float inv_det = 1.0f / det;
                float tvx = o.x - r.p1x, tvy = o.y - r.p1y, tvz = o.z - r.p1z;
                float u = pt_fma(tvz, pvz, pt_fma(tvy, pvy, tvx * pvx)) * inv_det;
                keep = keep && !(u < 0.0f || u > 1.0f);
                mx |= (keep ? 1u : 0u) << (n - 1 - jj);
            }
            atomicAdd(&vstat[0], (unsigned long long)n);                 // pairs examined
            atomicAdd(&vstat[1], (unsigned long long)__popc(mx));        // pairs the reference keeps
            atomicAdd(&vstat[2], (unsigned long long)__popc(m));         // pairs the filter keeps
            atomicAdd(&vstat[3], (unsigned long long)__popc(mx & ~m));   // VIOLATIONS: must stay 0
            if (QUADS == 3 && DET_BOUNDED && tame) {
                // how far mode 3's own roundings are from the reference's floats, relative to the slack
                // budgeted for that: |un_here - un_ref| / deltaP and |det_here c - det_ref c| / deltaD
                float r1 = 0.0f, r3 = 0.0f;
                {
                    const float deltaP = -p1_lo, deltaD = deltaP * (128.0f / 192.0f);
                    for (int jj = 0; jj + 1 < n; jj += 4) {
                        pt_f2 unp, thp;
                        pt_const_f32p tq = p1tab + PT_P1_STRIDE * ((base + jj) >> 2);
                        pt_quad3_pass1(tq, r3v, unp, thp);
                        for (int h = 0; h < 2 && jj + 2 * h + 1 < n; ++h) {
                            const PtTriRec a = pt_load_tri(T, base + jj + 2 * h);
                            float pvx = pt_fma(d.y, a.e2z, -(d.z * a.e2y)), pvy = pt_fma(d.z, a.e2x, -(d.x * a.e2z)), pvz = pt_fma(d.x, a.e2y, -(d.y * a.e2x));
                            float detA = pt_fma(a.e1z, pvz, pt_fma(a.e1y, pvy, a.e1x * pvx));
                            float unA = pt_fma(o.z - a.p1z, pvz, pt_fma(o.y - a.p1y, pvy, (o.x - a.p1x) * pvx));
                            const float dhi = tq[18 + h];
                            const float q1 = __builtin_fabsf((h ? unp.y : unp.x) - unA) / deltaP;
                            const float q3 = __builtin_fabsf(((h ? thp.y : thp.x) - dhi) - detA * 1.000002f) / deltaD;
                            r1 = q1 > r1 ? q1 : r1;
                            r3 = q3 > r3 ? q3 : r3;
                        }
                    }
                }
                atomicMax(&vstat[4], (unsigned long long)__float_as_uint(r1));
                atomicMax(&vstat[5], (unsigned long long)__float_as_uint(r3));
            }
        }
#endif
        if (!alive) m = 0u;  // a dead lane's stale ray must not cost pass-2 iterations
#if PT_STAMPS
        PT_STAMP(tb);
        if (p1_ticks) *p1_ticks += tb - ta;
#endif
        pt_pass2_chunk<DET_BOUNDED, LDS_TABLE>(m, base, n, tris, o, d, tmax, hu, hv, hidx, tl, lane, steps);
    }
    pt_pass2_finish<DET_BOUNDED, LDS_TABLE>(tris, o, d, tmax, hu, hv, hidx, tl, lane, steps);
    return steps;
}

// closest hit of a FRESH wave of primary rays whose pixels have candidate masks (pt_primary_mask_kernel):
// pass 1 is skipped, pass 2 is the one of pt_intersect_two_pass.  ntri <= 64 (two chunks).
template <bool DET_BOUNDED, int LDS_TABLE>
PTK_DEV unsigned pt_intersect_primary(pt_const_f32p T, const PtPrepTriangle* tris, int ntri, const f3& o, const f3& d, bool alive,
                                      float& tmax, float& hu, float& hv, int& hidx, uint2 pm, PtTail tl, unsigned lane,
                                      unsigned long long* vstat = nullptr)
{
    (void)T; (void)vstat;
    tl.wr = tl.rd = 0u;
    tl.kbest = ~0ull;
    unsigned steps = 0;
    for (int base = 0; base < ntri; base += 32) {
        const int n = ntri - base < 32 ? ntri - base : 32;
        unsigned m = base == 0 ? pm.x : pm.y;
#if PT_VALIDATE_FILTER
        if (alive && vstat) {  // the reference predicate of :100 and :109 must never accept a pair the mask dropped
            unsigned mx = 0u;
            for (int jj = 0; jj < n; ++jj) {
                const PtTriRec r = pt_load_tri(T, base + jj);
                float pvx = pt_fma(d.y, r.e2z, -(d.z * r.e2y));
                float pvy = pt_fma(d.z, r.e2x, -(d.x * r.e2z));
                float pvz = pt_fma(d.x, r.e2y, -(d.y * r.e2x));
                float det = pt_fma(r.e1z, pvz, pt_fma(r.e1y, pvy, r.e1x * pvx));
                bool keep = !(det < 1e-8f || -det > 1e-8f);
                float inv_det = 1.0f / det;
                float tvx = o.x - r.p1x, tvy = o.y - r.p1y, tvz = o.z - r.p1z;
                float u = pt_fma(tvz, pvz, pt_fma(tvy, pvy, tvx * pvx)) * inv_det;
                keep = keep && !(u < 0.0f || u > 1.0f);
                mx |= (keep ? 1u : 0u) << (n - 1 - jj);
            }
            atomicAdd(&vstat[0], (unsigned long long)n);
            atomicAdd(&vstat[1], (unsigned long long)__popc(mx));
            atomicAdd(&vstat[2], (unsigned long long)__popc(m));
            atomicAdd(&vstat[3], (unsigned long long)__popc(mx & ~m));   // VIOLATIONS: must stay 0
        }
#endif
        if (!alive) m = 0u;
        pt_pass2_chunk<DET_BOUNDED, LDS_TABLE>(m, base, n, tris, o, d, tmax, hu, hv, hidx, tl, lane, steps);
    }
    pt_pass2_finish<DET_BOUNDED, LDS_TABLE>(tris, o, d, tmax, hu, hv, hidx, tl, lane, steps);
    return steps;
}

// ------------------------------------------------------------------------------------------
// trace kernels
// ------------------------------------------------------------------------------------------

// one path: 16 dwords (what a lane holds, and what the per-wave pool parks: pt_pool_push / pt_pool_pop)
struct PtPath {
    f3 o, d;        // current ray (:257)
    f3 mask, L;     // throughput and radiance (:225-226)
    uint32_t seed;  // RNG state (:308)
    int bounce;     // loop index i of traceRays (:229)
    unsigned lp;    // local pixel index
    unsigned fl;    // frame index inside the chunk
};


// ---- shade one bounce of a live path (:229-258); on path end store its radiance --------------------
template <bool DET_BOUNDED, bool LATE>
PTK_DEV void pt_shade(const PtTraceParams& P, PtPath& s, bool& alive, float tmax, float hu, float hv, int hidx,
                      unsigned long long* sub = nullptr)
{
    const pt_kargs_p K = pt_kargs();  // tris, mats, nmat, max_bounces, rad, npix_local: read here, not kept in SGPRs
#if PT_STAMPS == 2
    unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, q6 = 0;
#define PT_SUB(var) PT_STAMP(var)
#else
#define PT_SUB(var) do { } while (0)
#endif
    (void)sub;
    PT_SUB(q0);
    bool finished = false;
    if (hidx < 0) {
        const float bg = pt_max(0.45f, 0.0f);
        s.L = add3(s.L, scale3(s.mask, bg));  // :235
        finished = true;
    } else {
        // The direction sample's angle first: here only the path state is live.  (pt_sincos takes its binary32 branch on every
        // angle this path forms, phi in [0, 2 pi]; its binary64 branch is dead at run time.)  Both BRDFs draw phi first, then the
        // second uniform (:163-164, :182-183).
        float phi = PTK_TWO_PI * pt_random_float(s.seed);
        float xi = pt_random_float(s.seed);
        float sp, cp;
        __builtin_amdgcn_sched_barrier(0);
        pt_sincos(phi, sp, cp);
        __builtin_amdgcn_sched_barrier(0);
        PT_SUB(q1);

        // deferred HitRecord of the closest hit (:127-130): same values as writing it at every
        // acceptance, only the last one is read.
        const float4 nid = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(PT_ARG(tris) + hidx) + 12);
        const f3 N = mk3(nid.x, nid.y, nid.z);
        int mid = __float_as_int(nid.w);
        mid = mid < 0 ? 0 : (mid >= PT_ARG(nmat) ? PT_ARG(nmat) - 1 : mid);  // never fault on a corrupt id
        f3 p = add3(s.o, scale3(s.d, tmax));
        float w = 1.0f - hu - hv;
        f3 n = normalize3(add3(add3(scale3(N, hu), scale3(N, hv)), scale3(N, w)));

        const PtRawMaterial* mat = PT_ARG(mats) + mid;  // :239
        const float4 alb = *reinterpret_cast<const float4*>(mat->albedo);
        const float4 emi = *reinterpret_cast<const float4*>(mat->emissive);
        const float rough = mat->roughness;
        const int type = mat->type;

        s.L.x = s.L.x + s.mask.x * emi.x * 3.0f;  // :241
        s.L.y = s.L.y + s.mask.y * emi.y * 3.0f;
        s.L.z = s.L.z + s.mask.z * emi.z * 3.0f;

        n = dot3(n, s.d) < 0.0f ? n : scale3(n, -1.0f);  // :243
        f3 wo = neg3(s.d);
        PT_SUB(q2);

        // sampleHemisphereCosine (:161-172) and sampleGGX (:180-192) share everything
        // except (sinTheta, cosTheta)
        f3 axis = __builtin_fabsf(n.x) > 0.001f ? mk3(0.0f, 1.0f, 0.0f) : mk3(1.0f, 0.0f, 0.0f);
        f3 tv = normalize3(cross3(axis, n));
        f3 sv = cross3(n, tv);
        PT_SUB(q3);
        // one sqrt pair for both BRDFs (a wave usually holds both material types): only the
        // radicands differ -- diffuse: sqrt(xi), sqrt(1-xi); GGX: sqrt((1-xi)/(xi(r^2-1)+1)), then
        // sqrt(max(0, 1-cos^2))
        float cos_arg = 1.0f - xi;
        if (type == 2) cos_arg = cos_arg / (xi * (rough * rough - 1.0f) + 1.0f);
        const float cosTheta = pt_sqrt(cos_arg);
        const float sin_arg = type == 2 ? pt_max(0.0f, 1.0f - cosTheta * cosTheta) : xi;
        const float sinTheta = pt_sqrt(sin_arg);
        f3 a = scale3(scale3(sv, cp), sinTheta);
        f3 b = scale3(scale3(tv, sp), sinTheta);
        f3 c = scale3(n, cosTheta);
        f3 sdir = normalize3(add3(add3(a, b), c));
        PT_SUB(q4);

        f3 wi = sdir;
        f3 color = mk3(0.0f, 0.0f, 0.0f);
        float pdf = 0.0f;
        float dwin = 0.0f;
        if (type == 1) {  // DIFFUSE (:197-204)
            dwin = dot3(wi, n);
            pdf = dwin * PTK_INV_PI;
            color = mk3(alb.x * PTK_INV_PI, alb.y * PTK_INV_PI, alb.z * PTK_INV_PI);
        } else if (type == 2) {  // SPECULAR (:205-218)
            float k2 = 2.0f * dot3(wo, sdir);
            wi = add3(neg3(wo), scale3(sdir, k2));  // reflect(wo, wh) (:156-159)
            dwin = dot3(wi, n);
            float dwon = dot3(wo, n);
            if (!(dwin * dwon < 0.0f)) {
                float r2 = rough * rough;
                const float gd = cosTheta * cosTheta * (r2 - 1.0f) + 1.0f;
                float D = r2 * PTK_INV_PI / (gd * gd);  // pow(x, 2.0f) is x*x in PTSPEC (:177)
                pdf = D * cosTheta / (4.0f * dot3(wo, sdir));
                float g = D / (4.0f * dwin * dwon);
                color = mk3(alb.x * g * 2.0f, alb.y * g * 2.0f, alb.z * g * 2.0f);
            }
        }
        PT_SUB(q5);
        if (pdf <= 0.0f) {  // :251
            finished = true;
        } else {
            float qx = color.x * dwin, qy = color.y * dwin, qz = color.z * dwin;
            pt_div3(qx, qy, qz, pdf);   // the three IEEE quotients of :253-255
            s.mask.x = s.mask.x * qx;
            s.mask.y = s.mask.y * qy;
            s.mask.z = s.mask.z * qz;
            s.bounce++;
            if (s.bounce >= PT_ARG(max_bounces)) {
                finished = true;
            } else {
                s.o = add3(p, scale3(wi, 0.01f));  // :257
                s.d = normalize3(wi);
            }
        }
    }
    if (finished) {
        // :260; the w lane of the reference's float4 is overwritten with 1.0 by gammaCorrect (:293) and never read
        // before: three floats per sample are staged, not four (measured at the full 256 spp, profiles/r02/
        // pmc_r02_summary.txt: 35.2 bytes of HBM traffic per sample against 41.0 with aligned 16-byte records --
        // the L2 cannot hold every partly filled line of the ~900 000 paths in flight until it is complete, and a
        // partly written sector costs a read-modify-write either way; fewer bytes written, fewer sectors touched)
        // (frame f of the render: entry (f + phase) % 2S of the staging ring, the first S entries in one slot, the others in the other.
        // Read from the kernarg segment here whichever the kernel: once per finished path, and six SGPRs the LBVH kernel does not have)
        const pt_kargs_p KR = pt_kargs();
        const unsigned fr = s.fl + KR->ring_phase, S = KR->slot_frames;
        const unsigned r = fr - __umulhi(fr, KR->ring_magic) * (2u * S);
        float* out = (r < S ? KR->rad : KR->rad1) + ((size_t)(r < S ? r : r - S) * KR->npix_local + s.lp) * 3u;
        // (records are 12 bytes apart: the vector type is declared with the 4-byte alignment the address really has)
        typedef float pt_f3v __attribute__((ext_vector_type(3), aligned(4)));
        pt_f3v v;
        v.x = pt_max(s.L.x, 0.0f);
        v.y = pt_max(s.L.y, 0.0f);
        v.z = pt_max(s.L.z, 0.0f);
        *reinterpret_cast<pt_f3v*>(out) = v;  // one 12-byte store (global_store_dwordx3)
        alive = false;
    }
#if PT_STAMPS == 2
    PT_SUB(q6);
    if (sub && q1) { sub[0] += q1 - q0; sub[1] += q2 - q1; sub[2] += q3 - q2; sub[3] += q4 - q3; sub[4] += q5 - q4; sub[5] += q6 - q5; }
#endif
#undef PT_SUB
}

// (n_rays, n_samples: wave-uniform tallies kept in SGPRs -- popcounts of the lanes that shaded / finished; as per-lane counters they
// were the two registers the 64-VGPR kernel spilled, and the reload in the store block waited for the radiance store itself)
PTK_DEV void pt_flush_counters(const PtTraceParams& P, unsigned lane, unsigned n_rays, unsigned n_samples)
{
    if (!P.stats) return;
    // wave reduction of the work counters, one atomic pair per wave
    const unsigned long long r = n_rays, s = n_samples;
    if (lane == 0) {
        atomicAdd(&P.stats[0], s);
        atomicAdd(&P.stats[1], r);
    }
}

// ---- how lanes get their paths: a per-wave pool of parked paths + FRESH phases --------------------
// Lane = path.  A wave retires ~10 of its 64 paths per bounce.  Round 1 handed every dead lane the next
// sample of the wave's range on the spot (camera rays pre-generated 64 wide into LDS), so each bounce
// mixed ~10 primary rays into 54 incoherent secondary ones and every sample's bounce 0 took a full-price
// slot of the incoherent main loop.  Now the wave keeps a POOL of up to 64 parked paths in LDS (64 B each):
//   * dead lanes take parked paths from the pool;
//   * when lanes are dead and the pool is empty, the wave PARKS all its live paths and starts 64 fresh
//     samples -- 64 consecutive pixels of its range -- in all 64 lanes at once: camera rays at full lane
//     width straight into registers, and a bounce 0 whose 64 rays share the origin and are coherent
//     (pass 2 walks ~the same 3-4 survivors in every lane instead of max-over-lanes 8; a wave usually sees
//     one material, so the untaken BRDF branch is skipped wave-wide).  The following bounces refill the
//     ~10 lanes that end per bounce from the pool, which lasts ~5 bounces -- until the next fresh phase.
// Which lane runs which sample, and in which order, never affects a sample's result.
#define PT_POOL 64  // parked-path slots per wave (a fresh phase parks at most 64 live paths)

struct PtWaveQueue {  // wave-uniform (SGPRs): the wave's current range [pix, end) of local pixels of `frame`
    unsigned pix, end, frame;   // frame: counted from the render's first frame (what a path keeps as `fl`)
    unsigned row, col;       // local row / column of `pix`, kept incrementally: no per-lane division
    unsigned sl, within;     // row = sl * stripe_rows + within (the stripe of the multi-GPU split)
    unsigned g;              // the shard of the queue the wave takes its batches from, or PT_Q_EMPTY (pt_queue_refill)
};

#define PT_Q_EMPTY 0xffffffffu   // PtWaveQueue::g once the wave has found every shard of the queue empty

// Makes [q.pix, q.end) non-empty when the current batch is used up; false when there is nothing (more) to start in this launch.
// The batches of a launch's chunk (128 or 256 consecutive pixels of one frame, numbered frame-major) come off a SHARDED queue:
// PT_QUEUE_SHARDS counters, each on a cache line of its own; shard s deals the batches s, s + NS, s + 2 NS, ...  One counter for
// 8 192 waves is an L2 channel's atomic unit doing nothing else -- at 128 samples per batch a 16-frame launch is an atomic every
// 14 ns, about what the unit serves, and the waves of a launch start TOGETHER, so their grabs arrive in bursts (a wave waits for its
// turn: measured ~50 us per launch, profiles/r04/queue_shards.txt).  A wave stays with its shard (the wave's number mod NS at
// first) and moves on to the next when it is used up; a shard found empty is marked in the queue's STOP word (one bit per shard,
// on a line of its own), which every wave polls at its fresh phases anyway (pt_queue_next) and which spares the others the grab.
// All bits set = the launch has handed out its last batch.
template <bool LATE>
PTK_DEV bool pt_queue_refill(const PtTraceParams& P, unsigned lane, PtWaveQueue& q, unsigned empty_mask)
{
    const pt_kargs_p K = pt_kargs();
    if (q.pix != q.end) return true;
    if (q.g == PT_Q_EMPTY) return false;
    const unsigned total = PT_ARG(total_batches);
    unsigned sh = q.g, b = 0u;
    bool got = false;
    for (unsigned tries = 0u; tries < PT_QUEUE_SHARDS && total != 0u; ++tries, sh = (sh + 1u) & (PT_QUEUE_SHARDS - 1u)) {
        if ((empty_mask >> sh) & 1u) continue;
        unsigned k = 0u;
        if (lane == 0) k = atomicAdd(PT_ARG(batch_counter) + sh * PT_QUEUE_SHARD_WORDS, 1u);
        k = __builtin_amdgcn_readfirstlane(k);
        b = k * PT_QUEUE_SHARDS + sh;
        if (b < total) { got = true; break; }
        // (mark it, and see what the others have marked meanwhile: no grab at a shard known to be empty)
        unsigned seen = 0u;
        if (lane == 0u) seen = atomicOr(PT_ARG(batch_counter) + PT_QUEUE_STOP_WORD, 1u << sh);
        empty_mask |= (1u << sh) | (unsigned)__builtin_amdgcn_readfirstlane(seen);
#if PT_LAUNCH_STAMPS
        if (lane == 0u && PT_ARG(stats) != nullptr && (((unsigned)__builtin_amdgcn_readfirstlane(seen) | (1u << sh)) == (1u << PT_QUEUE_SHARDS) - 1u) &&
            (unsigned)__builtin_amdgcn_readfirstlane(seen) != (1u << PT_QUEUE_SHARDS) - 1u && PT_ARG(stats)[13] * PT_ARG(slot_frames) == PT_ARG(chunk_f0))
            atomicMin(&PT_ARG(stats)[14], (unsigned long long)__builtin_amdgcn_s_memrealtime());   // the moment the stop word became complete
#endif
    }
    if (!got) { q.g = PT_Q_EMPTY; return false; }
    q.g = sh;
    const unsigned f = b / PT_ARG(batches_per_frame);
    const unsigned bi = b - f * PT_ARG(batches_per_frame);
    q.frame = PT_ARG(chunk_f0) + f;
    q.pix = bi * PT_ARG(batch);
    const unsigned e = q.pix + PT_ARG(batch);
    q.end = e < PT_ARG(npix_local) ? e : PT_ARG(npix_local);
    q.row = q.pix / (unsigned)PT_ARG(width);  // wave-uniform divisions, once per batch
    q.col = q.pix - q.row * (unsigned)PT_ARG(width);
    q.sl = q.row / (unsigned)PT_ARG(stripe_rows);
    q.within = q.row - q.sl * (unsigned)PT_ARG(stripe_rows);
    return true;
}

// pool slot k of a wave: 60 bytes = three float4 arrays of PT_POOL entries (conflict-free b128 accesses) + one array of
// three dwords.  (64-byte slots would put the workgroup over 160 KB / 8: the LDS is what decides whether 8 workgroups
// -- 8 waves per SIMD -- fit a CU.)
//   [0] o.xyz d.x   [1] d.yz mask.xy   [2] mask.z L.xyz   [3] seed, lp, fl | bounce << 16
#define PT_POOL_DWORDS (PT_POOL * 15)
PTK_DEV void pt_pool_push(float4* pool, unsigned& pool_n, const PtPath& s, bool& alive)
{
    const unsigned long long live = __ballot(alive);
    if (alive) {
        const unsigned k = pool_n + pt_mbcnt(live);
        pool[k] = make_float4(s.o.x, s.o.y, s.o.z, s.d.x);
        pool[PT_POOL + k] = make_float4(s.d.y, s.d.z, s.mask.x, s.mask.y);
        pool[2 * PT_POOL + k] = make_float4(s.mask.z, s.L.x, s.L.y, s.L.z);
        unsigned* w = reinterpret_cast<unsigned*>(pool + 3 * PT_POOL) + 3u * k;
        w[0] = s.seed;
        w[1] = s.lp;
        w[2] = s.fl | ((unsigned)s.bounce << 16);  // both below 65 536 (pt_render_frames checks: the ring has fewer frames)
    }
    pool_n += (unsigned)__popcll(live);
    alive = false;
}

PTK_DEV void pt_pool_pop(float4* pool, unsigned& pool_n, PtPath& s, bool& alive)
{
    const unsigned long long need = __ballot(!alive);
    const unsigned n_need = (unsigned)__popcll(need);
    const unsigned take = n_need < pool_n ? n_need : pool_n;
    if (take == 0u) return;
    // the wave's own LDS writes (pt_pool_push), read back by other lanes of the same wave: program order
    // suffices for the hardware (one in-order LDS queue per wave); this keeps the compiler from reordering
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned rank = pt_mbcnt(need);
    if (!alive && rank < take) {
        const unsigned k = pool_n - 1u - rank;
        const float4 a0 = pool[k], a1 = pool[PT_POOL + k], a2 = pool[2 * PT_POOL + k];
        const unsigned* w = reinterpret_cast<const unsigned*>(pool + 3 * PT_POOL) + 3u * k;
        const unsigned w0 = w[0], w1 = w[1], w2 = w[2];
        s.o = mk3(a0.x, a0.y, a0.z);
        s.d = mk3(a0.w, a1.x, a1.y);
        s.mask = mk3(a1.z, a1.w, a2.x);
        s.L = mk3(a2.y, a2.z, a2.w);
        s.seed = w0;
        s.lp = w1;
        s.fl = w2 & 0xffffu;
        s.bounce = (int)(w2 >> 16);
        alive = true;
    }
    pool_n -= take;
    // reads of the slots just released must complete before a later push overwrites them: same in-order queue
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- checkpointed launches (PtTraceParams::carry) ---------------------------------------------------------------------
// A wave stops only where its pool is empty and a fresh phase would begin: it parks its live paths exactly as a fresh phase does
// (the bounce loop's own pt_pool_push site -- a second copy of that code after the loop cost fourteen spilled registers INSIDE
// the loop) and leaves; the pool then goes to the wave's region: 16 header dwords and the parked-path record as arrays of
// PT_CARRY_RECORDS entries (three float4 arrays, three dword arrays: lane k moves entry k, coalesced).
PTK_DEV void pt_carry_store(uint32_t* region, unsigned lane, const float4* pool, unsigned pool_n, const PtWaveQueue& q, unsigned n_rays, unsigned n_samples,
                            unsigned n_carried)
{
    float4* A = reinterpret_cast<float4*>(region + 16);
    unsigned* W = region + 16 + PT_CARRY_RECORDS * 12;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the wave's own pool writes, read by other lanes: as in pt_pool_pop
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < pool_n) {
        A[lane] = pool[lane];
        A[PT_CARRY_RECORDS + lane] = pool[PT_POOL + lane];
        A[2 * PT_CARRY_RECORDS + lane] = pool[2 * PT_POOL + lane];
        const unsigned* w = reinterpret_cast<const unsigned*>(pool + 3 * PT_POOL) + 3u * lane;
        W[lane] = w[0];
        W[PT_CARRY_RECORDS + lane] = w[1];
        W[2 * PT_CARRY_RECORDS + lane] = w[2];
    }
    if (lane == 0u) {
        region[0] = pool_n;
        region[1] = q.pix;
        region[2] = q.end;
        region[3] = q.frame;
        // the wave's tallies travel with the checkpoint and reach the stats buffer at the end of the render's last launch, where the
        // waves leave one by one: 8 192 waves leaving TOGETHER, two or three atomics each on the same line, measured 0.25 ms per launch
        region[4] = n_rays;
        region[5] = n_samples;
        region[6] = n_carried + pool_n + (q.end - q.pix);
    }
}

// the LBVH kernel's checkpoint: no pool -- the live lanes go to the region directly, compacted (its registers have room for it)
PTK_DEV void pt_carry_store_lanes(uint32_t* region, unsigned lane, const PtPath& s, bool alive, const PtWaveQueue& q, unsigned n_rays, unsigned n_samples,
                                  unsigned n_carried)
{
    float4* A = reinterpret_cast<float4*>(region + 16);
    unsigned* W = region + 16 + PT_CARRY_RECORDS * 12;
    const unsigned long long live = __ballot(alive);
    const unsigned n_live = (unsigned)__popcll(live);
    if (alive) {
        const unsigned k = pt_mbcnt(live);
        A[k] = make_float4(s.o.x, s.o.y, s.o.z, s.d.x);
        A[PT_CARRY_RECORDS + k] = make_float4(s.d.y, s.d.z, s.mask.x, s.mask.y);
        A[2 * PT_CARRY_RECORDS + k] = make_float4(s.mask.z, s.L.x, s.L.y, s.L.z);
        W[k] = s.seed;
        W[PT_CARRY_RECORDS + k] = s.lp;
        W[2 * PT_CARRY_RECORDS + k] = s.fl | ((unsigned)s.bounce << 16);
    }
    if (lane == 0u) {
        region[0] = n_live;
        region[1] = q.pix;
        region[2] = q.end;
        region[3] = q.frame;
        region[4] = n_rays;
        region[5] = n_samples;
        region[6] = n_carried + n_live + (q.end - q.pix);
    }
}

// resumes a checkpoint: the parked paths go straight into lanes (at most 64: the pool was empty when they were parked), the rest
// of the batch becomes the wave's current range
template <bool LATE>
PTK_DEV void pt_carry_load(const PtTraceParams& P, const uint32_t* region, unsigned lane, PtPath& s, bool& alive, PtWaveQueue& q, unsigned& n_rays,
                           unsigned& n_samples, unsigned& n_carried)
{
    const pt_kargs_p K = pt_kargs();
    const unsigned n = __builtin_amdgcn_readfirstlane(region[0]);
    n_rays = __builtin_amdgcn_readfirstlane(region[4]);
    n_samples = __builtin_amdgcn_readfirstlane(region[5]);
    n_carried = __builtin_amdgcn_readfirstlane(region[6]);
    q.pix = __builtin_amdgcn_readfirstlane(region[1]);
    q.end = __builtin_amdgcn_readfirstlane(region[2]);
    q.frame = __builtin_amdgcn_readfirstlane(region[3]);
    q.row = q.pix / (unsigned)PT_ARG(width);
    q.col = q.pix - q.row * (unsigned)PT_ARG(width);
    q.sl = q.row / (unsigned)PT_ARG(stripe_rows);
    q.within = q.row - q.sl * (unsigned)PT_ARG(stripe_rows);
    const float4* A = reinterpret_cast<const float4*>(region + 16);
    const unsigned* W = region + 16 + PT_CARRY_RECORDS * 12;
    if (lane < n) {
        const float4 a0 = A[lane], a1 = A[PT_CARRY_RECORDS + lane], a2 = A[2 * PT_CARRY_RECORDS + lane];
        const unsigned w2 = W[2 * PT_CARRY_RECORDS + lane];
        s.o = mk3(a0.x, a0.y, a0.z);
        s.d = mk3(a0.w, a1.x, a1.y);
        s.mask = mk3(a1.z, a1.w, a2.x);
        s.L = mk3(a2.y, a2.z, a2.w);
        s.seed = W[lane];
        s.lp = W[PT_CARRY_RECORDS + lane];
        s.fl = w2 & 0xffffu;
        s.bounce = (int)(w2 >> 16);
        alive = true;
    }
}

// The wave is at a fresh-phase boundary (its pool is empty, some lane is dead).  1: [q.pix, q.end) holds samples to start;
// 0: nothing left to start (the classic end: the wave runs its last paths out); 2: STOP -- this launch ends with a
// checkpoint (carry_out): its queue has handed out the last batch (every shard's bit of the stop word is up: a line of its
// own, polled by one lane per wave and fresh phase), and this wave holds nothing of the PREVIOUS launch's chunk any more, whose
// fold follows this launch.
template <bool LATE>
PTK_DEV int pt_queue_next(const PtTraceParams& P, unsigned lane, PtWaveQueue& q, const PtPath& s, bool alive)
{
    const pt_kargs_p K = pt_kargs();
    unsigned empty_mask = 0u;
    if (PT_ARG(carry_out) != 0u) {
        if (lane == 0u) empty_mask = __hip_atomic_load(PT_ARG(batch_counter) + PT_QUEUE_STOP_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        empty_mask = (unsigned)__builtin_amdgcn_readfirstlane(empty_mask);
        // (of the previous chunk: a batch under way, paths in lanes -- the pool is empty)
        if (empty_mask == (1u << PT_QUEUE_SHARDS) - 1u && !(q.pix != q.end && q.frame < PT_ARG(chunk_f0)) && __ballot(alive && s.fl < PT_ARG(chunk_f0)) == 0ull)
            return 2;
    }
    return pt_queue_refill<LATE>(P, lane, q, empty_mask) ? 1 : 0;
}

// FRESH phase: every lane is dead (its path parked); the next (up to) 64 samples of the wave's range start
// in lanes 0.. at bounce 0 -- seed :308, camera ray :310
// returns true when all 64 lanes started a primary ray
template <bool LATE>
PTK_DEV bool pt_start_fresh(const PtTraceParams& P, unsigned lane, PtWaveQueue& q, PtPath& s, bool& alive)
{
    const pt_kargs_p K = pt_kargs();
    const unsigned avail = q.end - q.pix;
    const unsigned count = avail < 64u ? avail : 64u;
    const unsigned W = (unsigned)PT_ARG(width), SR = (unsigned)PT_ARG(stripe_rows);
    if (lane < count) {
        const unsigned lp = q.pix + lane;
        // local pixel -> (local row, column): walk from the range's own (row, col); 64 pixels span one or two rows
        // unless the image is narrower than a wave
        unsigned x = q.col + lane, up = 0u;
        while (x >= W) { x -= W; ++up; }
        unsigned grow = q.row + up;  // local row -> global row (image rows dealt to ranks in stripes)
        if (PT_ARG(n_ranks) > 1) {
            unsigned sl = q.sl, within = q.within + up;
            while (within >= SR) { within -= SR; ++sl; }
            grow = (sl * (unsigned)PT_ARG(n_ranks) + (unsigned)PT_ARG(rank)) * SR + within;
        }
        const unsigned gid = grow * W + x;
        const int frame = PT_ARG(frame_begin) - (int)PT_ARG(chunk_f0) + (int)q.frame;   // (the render's first frame + q.frame)
        s.seed = gid + pt_hash_u32((uint32_t)frame);                                 // :308
        pt_generate_ray((int)x, (int)grow, PT_ARG(inv_width), PT_ARG(inv_height), PT_ARG(aspect), s.seed, s.o, s.d);      // :310
        s.mask = mk3(1.0f, 1.0f, 1.0f);
        s.L = mk3(0.0f, 0.0f, 0.0f);
        s.bounce = 0;
        s.lp = lp;
        s.fl = q.frame;
        alive = true;
    }
    q.pix += count;
    q.col += count;
    while (q.col >= W) {
        q.col -= W;
        ++q.row;
        if (++q.within == SR) { q.within = 0u; ++q.sl; }
    }
    return count == 64u;
}

template <bool DET_BOUNDED, int LDS_TABLE, int QUADS>
PTK_DEV void pt_trace_body(const PtTraceParams& P)
{
    const unsigned lane = pt_lane_id();
    pt_const_f32p T = (pt_const_f32p)(const float*)P.tris;
    const int ntri = P.ntri;
    if (LDS_TABLE == 1) {
        const float* g = reinterpret_cast<const float*>(P.tris);
        for (int k = (int)threadIdx.x; k < ntri * PT_LDS_TRI_STRIDE; k += PT_TRACE_THREADS) {
            int tri = k / PT_LDS_TRI_STRIDE, w = k - tri * PT_LDS_TRI_STRIDE;
            pt_lds_tab[k] = g[tri * 16 + w];
        }
        __syncthreads();
    }
    // this wave's pool of parked paths, behind the triangle table (ptk_trace_lds_bytes)
    const unsigned wave_in_wg = (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform to the compiler too: the addresses below live in SGPRs)
    float4* pool = reinterpret_cast<float4*>(pt_lds_tab + (LDS_TABLE == 1 ? ntri * PT_LDS_TRI_STRIDE : 0) + wave_in_wg * PT_POOL_DWORDS);
    unsigned pool_n = 0u;                    // parked paths (wave-uniform)
    // this wave's pass-2 tail: 64 key slots + the pending-pair ring, behind the four pools
    PtTail tl;
    {
        const unsigned tail_dw = 128u + (LDS_TABLE == 1 ? PT_TAIL_LIST / 2u : PT_TAIL_LIST);  // keys + pair ring (2-byte pairs beside the table)
        const unsigned tails = (LDS_TABLE == 1 ? ntri * PT_LDS_TRI_STRIDE : 0) + (PT_TRACE_THREADS / 64) * PT_POOL_DWORDS;
        pt_lds_u32* w = (pt_lds_u32*)pt_lds_tab + tails + (threadIdx.x >> 6) * tail_dw;
        tl.keys = (pt_lds_u64*)w;
        tl.list = w + 128;
        tl.wr = tl.rd = 0u;
        tl.kbest = ~0ull; tl.ku = tl.kv = 0.0f;
        tl.tile = tails + (PT_TRACE_THREADS / 64) * tail_dw + (threadIdx.x >> 6) * (32u * PT_LDS_TRI_STRIDE);  // (TILED mode only)
        tl.keys[lane] = ~0ull;
    }

    PtWaveQueue q = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };   // wave-uniform (SGPRs)
    bool alive = false;
    PtPath s;
    s.o = mk3(0.0f, 0.0f, 0.0f); s.d = mk3(0.0f, 0.0f, 1.0f);
    s.mask = mk3(1.0f, 1.0f, 1.0f); s.L = mk3(0.0f, 0.0f, 0.0f);
    s.seed = 0; s.bounce = 0; s.lp = 0; s.fl = 0;
    unsigned n_rays = 0, n_samples = 0, n_carried = 0;   // (n_carried: samples this wave's checkpoints have handed on, PT_STAT_CARRIED)
    // checkpointed launches: resume what the previous launch of the render left in this wave's region
    // (the wave's number through readfirstlane: to the compiler threadIdx.x >> 6 differs between lanes, and so would everything below)
    // checkpointed launches: resume what the previous launch of the render left in this wave's region
    // (the wave's number through readfirstlane: to the compiler threadIdx.x >> 6 differs between lanes, and so would everything below)
    if (blockIdx.x * (PT_TRACE_THREADS / 64) + wave_in_wg < P.carry_in_waves)
        pt_carry_load<true>(P, P.carry + (size_t)(blockIdx.x * (PT_TRACE_THREADS / 64) + wave_in_wg) * PT_CARRY_STRIDE_DW, lane, s, alive, q, n_rays, n_samples, n_carried);
    q.g = (blockIdx.x * (PT_TRACE_THREADS / 64) + wave_in_wg) & (PT_QUEUE_SHARDS - 1u);   // the wave's first shard of this launch's queue
#if PT_LAUNCH_STAMPS   // DIAGNOSTIC build (tools/launch_stamps.py): where a checkpointed launch's time goes -- s_memrealtime (100 MHz) of the
    // first wave's start, the moment the stop word is complete, the last wave's exit; stats[9..12]
    const unsigned long long ls_start = __builtin_amdgcn_s_memrealtime();
    bool ls_saw_stop = false;
    unsigned ls_boundaries = 0u, ls_iters = 0u, ls_iters_at_boundary = 0u;
    unsigned long long ls_t_boundary = ls_start;
#endif
#if PT_STAMPS
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, c_regen = 0, c_loop = 0, c_shade = 0, c_iters = 0, c_steps = 0, c_p1 = 0;
#endif
#if PT_STAMPS == 2
    unsigned long long c_sub[6] = { 0, 0, 0, 0, 0, 0 };
    (void)c_regen; (void)c_loop; (void)c_shade; (void)c_iters; (void)c_steps;  // this build reports the sub-phases instead
#endif

    for (;;) {
        PT_STAMP(t0);
        bool primary = false;   // (wave-uniform) this bounce is a fresh wave of 64 primary rays
        if (__ballot(!alive) != 0ull) {
            if (pool_n == 0u) {
                const int next = pt_queue_next<true>(P, lane, q, s, alive);
#if PT_LAUNCH_STAMPS
                ++ls_boundaries;
                if (next != 2) { ls_t_boundary = __builtin_amdgcn_s_memrealtime(); ls_iters_at_boundary = ls_iters; }
#endif
                if (next != 0) {
                    pt_pool_push(pool, pool_n, s, alive);                // park every live path ...
#if PT_LAUNCH_STAMPS
                    if (next == 2) ls_saw_stop = true;
#endif
                    if (next == 2) break;                                // ... for the next launch (a checkpoint) ...
                    primary = pt_start_fresh<true>(P, lane, q, s, alive);      // ... or start 64 coherent primary rays
                }
            }
            pt_pool_pop(pool, pool_n, s, alive);           // dead lanes resume parked paths
        }
        if (__ballot(alive) == 0ull) break;
#if PT_LAUNCH_STAMPS
        ++ls_iters;
#endif
        PT_STAMP(t1);

        // ---- intersectWorld (:137-154) ------------------------------------------------------
        float tmax = 1e20f, hu = 0.0f, hv = 0.0f;
        int hidx = -1;
        unsigned p2steps = 0;
        if (QUADS == 3 && DET_BOUNDED && primary && P.pmask != nullptr)
            p2steps = pt_intersect_primary<DET_BOUNDED, LDS_TABLE>(T, P.tris, ntri, s.o, s.d, alive, tmax, hu, hv, hidx, P.pmask[s.lp], tl, lane,
                                                                   PT_VALIDATE_FILTER && P.stats ? P.stats + 2 : nullptr);
        else
            p2steps = pt_intersect_two_pass<DET_BOUNDED, LDS_TABLE, QUADS>(T, P.tris, ntri, s.o, s.d, alive, tmax, hu, hv, hidx,
                                                                                          P.quad_delta1, P.ray_radius,
                                                                                          (pt_const_f32p)P.p1tab, P.p1_lo, P.p1_hi, tl, lane,
                                                                                          PT_VALIDATE_FILTER && P.stats ? P.stats + 2 : nullptr,
#if PT_STAMPS
                                                                                          &c_p1
#else
                                                                                          nullptr
#endif
                                                                                          );
#if PT_STAMPS
        c_steps += p2steps;
#else
        (void)p2steps;
#endif

        PT_STAMP(t2);
        const unsigned long long shaded = __ballot(alive);
        n_rays += (unsigned)__popcll(shaded);
#if PT_STAMPS == 2
        if (alive) pt_shade<DET_BOUNDED, true>(P, s, alive, tmax, hu, hv, hidx, c_sub);
#else
        if (alive) pt_shade<DET_BOUNDED, true>(P, s, alive, tmax, hu, hv, hidx);
#endif
        n_samples += (unsigned)__popcll(shaded & ~__ballot(alive));   // (a path leaves pt_shade dead only when it has finished)
#if PT_STAMPS
        PT_STAMP(t3);
        c_regen += t1 - t0; c_loop += t2 - t1; c_shade += t3 - t2; c_iters++;
#endif
    }

#if PT_STAMPS
    if (P.stats && lane == 0) {
#if PT_STAMPS != 2
        atomicAdd(&P.stats[2], c_regen);
        atomicAdd(&P.stats[3], c_loop);
        atomicAdd(&P.stats[4], c_shade);
        atomicAdd(&P.stats[5], c_iters);
#endif
#if PT_STAMPS == 2
        for (int k = 0; k < 6; ++k) atomicAdd(&P.stats[2 + k], c_sub[k]);  // shade sub-phases instead
#else
        atomicAdd(&P.stats[7], c_steps);
        atomicAdd(&P.stats[6], c_p1);
#endif
    }
#endif
    {
        // (a wave that left the loop because nothing was alive holds nothing: no parked path, no rest of a batch -- an empty checkpoint)
        const pt_kargs_p K = pt_kargs();
        if (K->carry_out != 0u) {
#if PT_LAUNCH_STAMPS
            // per wave, plain stores into the CALLER'S oversized stats buffer (16 + 4 * waves words): no atomics on a shared line -- those,
            // 8 192 waves leaving together, backed the L2 channel up and slowed the waves still running five-fold (the first version
            // of this diagnostic measured its own congestion)
            if (K->stats != nullptr && lane == 0u && K->stats[13] * K->slot_frames == K->chunk_f0) {   // (stats[13]: which chunk's launch to stamp)
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                unsigned long long* mine = K->stats + 16 + 4u * (size_t)(blockIdx.x * (PT_TRACE_THREADS / 64) + wave_in_wg);
                mine[0] = ls_start;
                mine[1] = ls_t_boundary;
                mine[2] = now;
                mine[3] = ((unsigned long long)ls_boundaries << 40) | ((unsigned long long)(ls_iters - ls_iters_at_boundary) << 20) | ls_iters | (ls_saw_stop ? 1ull << 63 : 0ull);
            }
#endif
            pt_carry_store(K->carry + (size_t)(blockIdx.x * (PT_TRACE_THREADS / 64) + wave_in_wg) * PT_CARRY_STRIDE_DW, lane, pool, pool_n, q, n_rays, n_samples, n_carried);
            return;   // (the tallies went with it)
        }
    }
#if !PT_STAMPS && !PT_VALIDATE_FILTER   // (the diagnostic builds report their own figures in stats[2..7])
    if (P.stats && lane == 0 && n_carried != 0u) atomicAdd(&P.stats[7], (unsigned long long)n_carried);
#endif
    pt_flush_counters(P, lane, n_rays, n_samples);
}

// Waves per SIMD.  More resident waves is what this issue-bound kernel wants (round 2, same source: 5 -> 35.8 ms,
// 6 -> 34.1, 7 -> 32.0); three things had to give for 8: the arguments only regeneration and shading need are re-read
// from the kernarg segment instead of living in SGPRs (pt_kargs: no v_readlane spills, 78 SGPRs), the fresh phase lost
// its per-lane divisions and the camera basis its registers (64 VGPRs without scratch), and the LDS of a workgroup
// shrank to 20 160 B (60-byte pool slots, 2-byte tail pairs) so that 8 workgroups fit the CU's 160 KB.
#ifndef PT_TRACE_WAVES
#define PT_TRACE_WAVES 8
#endif
template <bool DET_BOUNDED, int LDS_TABLE, int QUADS>
__global__ __launch_bounds__(PT_TRACE_THREADS) __attribute__((amdgpu_waves_per_eu(PT_TRACE_WAVES, PT_TRACE_WAVES)))
void pt_trace_kernel(const PtTraceParams P)
{
    pt_trace_body<DET_BOUNDED, LDS_TABLE, QUADS>(P);
}

// TILED brute force (257+ triangles): its LDS (pool + 4-byte pair ring + record tiles = 25.6 KB) admits 6 workgroups
// per CU, so it may as well use the registers of 6 waves per SIMD
template <bool DET_BOUNDED>
__global__ __launch_bounds__(PT_TRACE_THREADS) __attribute__((amdgpu_waves_per_eu(6, 6)))
void pt_trace_tiled_kernel(const PtTraceParams P)
{
    pt_trace_body<DET_BOUNDED, 2, 0>(P);
}

// ---- the LBVH trace kernel -------------------------------------------------------------------------
// Lane = path, and every lane walks its own ray through the hierarchy -- but rays differ wildly in how many nodes they
// enter, so a wave that waits for its slowest lane before shading runs the search at a third of its lanes (round 1:
// 34 %).  Here the search is a per-lane STATE that survives the shading phase: the wave steps all traversing lanes
// together, and as soon as no more than PT_BVH_REFILL of them are still traversing, the finished lanes are shaded,
// dead ones take new samples, and all of them start their next search while the stragglers simply keep theirs.
//   * the hierarchy: eight-child nodes of 64 bytes (PtBvh8Node, pt_kernels.h; built by pt_bvh.hip): one sector, four
//     16-byte loads.  Three things bound the search about equally (profiles/r03/lbvh_bottlenecks.txt): vector-ALU issue,
//     the texture-address path (64 cycles for every load whose lanes read 64 different lines) and the L2's miss path; so a
//     node is as small as eight children allow and the node step is built for few instructions: entry / exit distances are one
//     FMA per plane straight from the quantised bytes, the ray's direction signs select the near and far planes of all
//     eight children at once, the children's slots encode their octant so "slot XOR ray octant" is the front-to-back
//     order (no sort), and the hits of a node travel as ONE stack entry (base index, hit mask) instead of one per child;
//   * a lane's state: the current GROUP of node children still to enter (gbase, gm = hits by slot | imask << 8; which
//     of them comes next is one lookup in a 2 KB table in LDS indexed by the ray's octant and the hits),
//     the leaf children still to test (tbase, tm = hits by slot | lmask << 8), and a stack of earlier groups
//     (PT_BVH_LDS_STACK entries in LDS, entry-major: conflict-free; deeper ones in a private array: a radix tree over
//     64-bit keys has at most 64 levels = 22 levels of eight-child nodes, one entry each);
//   * one wave step = a NODE phase (every lane without pending leaves enters its next node) and, when at least
//     PT_BVH_TRI_LANES lanes hold pending leaves or nobody can enter a node, a TRIANGLE phase (one exact test per such
//     lane): with 64 incoherent lanes some lane meets a leaf at nearly every step, and running the ~70-instruction
//     triangle test for a handful of lanes each time cost more than letting them wait a step or two;
//   * the triangles the builder kept out of the hierarchy (pt_bvh.hip: the few that span the scene) are searched
//     first, by the brute-force two-pass search over their own table, which also hands the traversal a tight tmax;
//   * a step budget and index checks make a damaged hierarchy end the search instead of hanging or faulting the GPU.
// TALLY: the measurement variant (PT_OPT_BVH_TALLY) adds the search's work counters to stats[2..4]: nodes entered,
// triangles tested (both per lane), node and triangle phases executed by the waves.  Never the timed kernel.
// Stack capacity.  An entry is a node's group with children still to enter, so the stack is never deeper than the
// eight-child hierarchy, whose nodes are binary nodes of the radix tree (pt_bvh.hip) in ancestor order: at most the 62 levels
// of a radix tree over 62-bit keys (30-bit Morton code << 32 | index).  64 entries cannot overflow on a hierarchy the builder
// made; PtTraceParams::bvh_stack_limit (<= PT_BVH_STACK) lowers the capacity for the test of the overflow report.
// (The overflow array lives in scratch; one of fewer than 64 dwords would be promoted to registers.)
#ifndef PT_BVH_STACK
#define PT_BVH_STACK 64
#endif
#ifndef PT_BVH_LDS_STACK
#define PT_BVH_LDS_STACK 8
#endif
#ifndef PT_BVH_REFILL
#define PT_BVH_REFILL 40
#endif


// dead lanes take the next samples of the wave's range, one by one (no coherence to keep here: the search dominates)
template <bool LATE>
PTK_DEV void pt_regenerate_lanes(const PtTraceParams& P, unsigned lane, PtWaveQueue& q, PtPath& s, bool& alive)
{
    const pt_kargs_p K = pt_kargs();
    unsigned long long need = __ballot(!alive);
    while (need != 0ull && pt_queue_refill<true>(P, lane, q, 0u)) {   // (once per batch: its arguments come from the kernarg segment whichever the kernel)
        const unsigned n_need = (unsigned)__popcll(need);
        const unsigned avail = q.end - q.pix;
        const unsigned take = n_need < avail ? n_need : avail;
        const unsigned rank = pt_mbcnt(need);
        if (!alive && rank < take) {
            const unsigned lp = q.pix + rank;
            const unsigned lr = lp / (unsigned)PT_ARG(width), x = lp - lr * (unsigned)PT_ARG(width);
            unsigned grow = lr;  // local row -> global row (image rows dealt to ranks in stripes)
            if (PT_ARG(n_ranks) > 1) {
                const unsigned sl = lr / (unsigned)PT_ARG(stripe_rows);
                const unsigned within = lr - sl * (unsigned)PT_ARG(stripe_rows);
                grow = (sl * (unsigned)PT_ARG(n_ranks) + (unsigned)PT_ARG(rank)) * (unsigned)PT_ARG(stripe_rows) + within;
            }
            const unsigned gid = grow * (unsigned)PT_ARG(width) + x;
            const int frame = PT_ARG(frame_begin) - (int)PT_ARG(chunk_f0) + (int)q.frame;
            s.seed = gid + pt_hash_u32((uint32_t)frame);                                 // :308
            pt_generate_ray((int)x, (int)grow, PT_ARG(inv_width), PT_ARG(inv_height), PT_ARG(aspect), s.seed, s.o, s.d);      // :310
            s.mask = mk3(1.0f, 1.0f, 1.0f);
            s.L = mk3(0.0f, 0.0f, 0.0f);
            s.bounce = 0;
            s.lp = lp;
            s.fl = q.frame;
            alive = true;
        }
        q.pix += take;
        need = __ballot(!alive);
    }
}

// a lane's search state (pt_bvh_step): the closest hit so far, the group of node children still to enter (gbase, gm = hits by
// slot | imask << 8), the stack depth.  (Leaf children never wait in the lane: they go to the wave's pair ring, pt_bvh_round.)
struct PtBvhLane {
    float tmax, hu, hv;
    int hidx;
    unsigned gbase, gm;
    unsigned oct;  // bit a set: the ray runs towards +a (children on the low side come first)
    int sp;
    float ix, iy, iz;  // 1 / (dir * tmax): distances along the ray in units of tmax (pt_bvh_scale)
    unsigned budget;
};

// Distances in units of tmax.  The slab test wants, per child, max(entry, 0) <= min(exit, tmax).  With every distance
// divided by tmax the search interval is [0, 1] -- exactly what the VOP3 `clamp` output modifier clamps to, for free: the 48
// FMAs of a node step deliver their planes' distances already clamped, the test is one max3, one min3 and ONE comparison,
// and the 16 v_max / v_min against 0 and tmax of the plain form (9 % of a node step's issue cycles) are gone.  The comparison
// becomes STRICT: a box wholly beyond tmax has entry = exit = 1 after the clamp (one behind the origin 0 = 0) and must
// fail; a box that holds a hit point of the ray is entered strictly before it is left (its faces lie PT_BVH_EPS x the scene
// outside the triangle: pt_bvh.hip), so nothing a triangle needs is lost.  The scaled inverse direction is refreshed whenever
// tmax shrinks (pt_bvh_round); rounding differences against the unscaled form are ~1e-7 relative, three orders of magnitude
// inside the boxes' margin.
PTK_DEV void pt_bvh_scale(PtBvhLane& L, const f3& d)
{
    // 1/dir for the slab tests only (conservative boxes: the error of v_rcp_f32 is far inside the boxes' margin), clamped
    // to +-2^60: a zero component keeps its sign and every product stays finite
    // tmax in (0, 1e20]; the cap keeps every product finite for a hit at a denormal distance (the interval then ends below 1:
    // still a superset of [0, tmax])
    const float it = __builtin_fminf(__builtin_amdgcn_rcpf(L.tmax), 0x1p40f);
    L.ix = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.x), -0x1p60f, 0x1p60f) * it;
    L.iy = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.y), -0x1p60f, 0x1p60f) * it;
    L.iz = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.z), -0x1p60f, 0x1p60f) * it;
}
PTK_DEV float pt_fma_clamp(float a, float b, float c)  // min(max(fma(a, b, c), 0), 1): the clamp is an output modifier, no instruction
{
    float r;
    asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// the search of a new ray starts at the root (tmax, hu, hv, hidx are the caller's: the big triangles were searched first)
PTK_DEV void pt_bvh_lane_start(PtBvhLane& L, const f3& d, int ntri)
{
    pt_bvh_scale(L, d);
    L.oct = (L.ix < 0.0f ? 0u : 1u) | (L.iy < 0.0f ? 0u : 2u) | (L.iz < 0.0f ? 0u : 4u);
    L.gbase = 0u;  // the root (node 0) as a group of one: slot 0
    L.gm = 1u | (1u << 8);
    L.sp = 0;
    // every node is entered at most once and a hierarchy over ntri leaves has fewer than ntri nodes: a valid tree never
    // uses the budget up (PT_BVH_FLAG_BUDGET reports a damaged one)
    L.budget = (unsigned)ntri + 64u;
}

typedef __attribute__((address_space(3))) unsigned char pt_lds_u8;

// sticky bits of *PtTraceParams::bvh_flags: the search of some ray was CUT SHORT -- its closest hit may be wrong.  The host
// turns them into PT_ERR_TRAVERSAL (pt_render_frames); neither can happen with a hierarchy pt_bvh.hip built (see PT_BVH_STACK)
#define PT_BVH_FLAG_STACK 1u   // a group had to be pushed beyond the stack's capacity
#define PT_BVH_FLAG_BUDGET 2u  // more node visits than the hierarchy has nodes
// The word lives in HOST memory mapped into the device's address space (pt_shim.hip: no render waits for the device to read
// it): one word per bit, raised by a plain system-scope store -- no read-modify-write travels over PCIe, nothing is ever
// read back by a kernel, and the path is taken by no ray of a valid hierarchy.
PTK_DEV void pt_raise_flag(unsigned int* flags, unsigned bit)
{
    __hip_atomic_store(flags + (bit == PT_BVH_FLAG_STACK ? 0 : 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- the leaves: (leaf record, ray lane) pairs, tested 64 at a time --------------------------------------------------
// With 64 incoherent lanes some lane meets a leaf at nearly every step, and each lane meets one only every ~10 nodes.  Round
// 2 let a lane WAIT with its leaves until 8 lanes had some and then ran the ~70-instruction exact test for those ~10 lanes
// (16 % of the lanes busy in a triangle phase, 74 % in a node phase: waiting lanes enter no nodes).  Now a lane never waits:
// the leaf children its node test hits are appended, as (record, ray lane) pairs, to the wave's ring in LDS -- the one the
// brute-force search's tail uses (pt_tail_round) -- and the lane goes on to its next node; as soon as PT_BVH_RING_MIN pairs
// are pending the wave tests up to 64 of them at once, one pair per lane whoever owns the ray: the ray travels by
// ds_bpermute, the candidate's key (t bits << 32 | triangle << 6 | testing lane) goes to the ray's slot with one
// ds_min_u64, and the owner picks (t, index) from its slot and (u, v) from the lane that tested the winner.  The reference's
// closest hit is the lexicographic minimum of (t, index) over the triangles that pass the exact test (strict t < tmax in an
// ascending loop, GenerateColors.cl:125,145-151), and the key's order IS that order (t > 0: float bits are monotone), so
// the slot, initialised with the ray's incumbent (tmax, hidx), ends up holding the reference's winner whatever the order of
// the tests.  A ray's tmax now shrinks a few steps later than it could (its leaf waits in the ring), which costs some node
// visits; PT_BVH_RING_MIN trades that against the rounds' occupancy.
// pair = leaf record index << 6 | ray lane; record indices are below 2^26 (2 x triangles: checked by the host)
#ifndef PT_BVH_RING_MIN
#define PT_BVH_RING_MIN 32u
#endif
template <bool DET_BOUNDED>
PTK_DEV void pt_bvh_round(const PtTraceParams& P, PtBvhLane& L, PtTail& tl, unsigned cnt, unsigned lane, const f3& o, const f3& d,
                          unsigned n_recs)
{
    // every lane, as the owner of a ray, publishes its incumbent; a slot nobody improves reads back unchanged
    const unsigned long long k0 = ((unsigned long long)__float_as_uint(L.tmax) << 32) |
                                  (unsigned long long)(((((unsigned)L.hidx) & 0x3ffffffu) << 6) | lane);
    tl.keys[lane] = k0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool act = lane < cnt;
    const unsigned e = act ? tl.list[(tl.rd + lane) & (PT_TAIL_LIST - 1u)] : 0u;
    const unsigned ray = e & 63u, idx = e >> 6;
    const unsigned a = ray << 2;
    const f3 po = mk3(pt_from_lane(a, o.x), pt_from_lane(a, o.y), pt_from_lane(a, o.z));
    const f3 pd = mk3(pt_from_lane(a, d.x), pt_from_lane(a, d.y), pt_from_lane(a, d.z));
    const bool valid = act & (idx < n_recs);
    const float4* qp = reinterpret_cast<const float4*>(P.bvh + (valid ? idx : 0u));
    const float4 q0 = qp[0], q1 = qp[1], q2 = qp[2];
    PtTriRec r;  // p1.xyz e1.x | e1.yz e2.xy | e2.z index ...
    r.p1x = q0.x; r.p1y = q0.y; r.p1z = q0.z;
    r.e1x = q0.w; r.e1y = q1.x; r.e1z = q1.y;
    r.e2x = q1.z; r.e2y = q1.w; r.e2z = q2.x;
    const unsigned tri = __float_as_uint(q2.y) & 0x3ffffffu;
    float t, u, v;
    bool ok;
    {   // the reference's test (:96-125) without the running tmax: the slot's minimum applies that
        float pvx = pt_fma(pd.y, r.e2z, -(pd.z * r.e2y));
        float pvy = pt_fma(pd.z, r.e2x, -(pd.x * r.e2z));
        float pvz = pt_fma(pd.x, r.e2y, -(pd.y * r.e2x));
        float det = pt_fma(r.e1z, pvz, pt_fma(r.e1y, pvy, r.e1x * pvx));
        float inv_det = DET_BOUNDED ? pt_rcp(det) : 1.0f / det;  // pt_rcp: exact, range-checked (det may be anything here)
        float tvx = po.x - r.p1x, tvy = po.y - r.p1y, tvz = po.z - r.p1z;
        u = pt_fma(tvz, pvz, pt_fma(tvy, pvy, tvx * pvx)) * inv_det;
        ok = !(det < 1e-8f) & !(-det > 1e-8f) & !(u < 0.0f) & !(u > 1.0f);  // :100, :109
        float qvx = pt_fma(tvy, r.e1z, -(tvz * r.e1y));
        float qvy = pt_fma(tvz, r.e1x, -(tvx * r.e1z));
        float qvz = pt_fma(tvx, r.e1y, -(tvy * r.e1x));
        v = pt_fma(pd.z, qvz, pt_fma(pd.y, qvy, pd.x * qvx)) * inv_det;
        ok &= !(v < 0.0f) & !(u + v > 1.0f);  // :117
        t = pt_fma(r.e2z, qvz, pt_fma(r.e2y, qvy, r.e2x * qvx)) * inv_det;
        ok &= (t > 0.0f) & (t < 1e20f);  // :125 against the initial tmax (:141)
    }
    if (ok & valid) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)((tri << 6) | lane);
        __hip_atomic_fetch_min(tl.keys + ray, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned long long slot = tl.keys[lane];
    const unsigned from = ((unsigned)slot & 63u) << 2;
    const float pu = pt_from_lane(from, u), pv = pt_from_lane(from, v);
    if (slot != k0) {  // (t, index) < the incumbent's: the new closest hit
        L.tmax = __uint_as_float((unsigned)(slot >> 32));
        L.hidx = (int)(((unsigned)slot >> 6) & 0x3ffffffu);
        L.hu = pu;
        L.hv = pv;
        pt_bvh_scale(L, d);
    }
    tl.rd += cnt;
}

// one step of the wave: a node phase for every traversing lane (trav: the lane still has nodes to enter), its leaf hits to the
// ring, a round when enough pairs are pending.  stk: this lane's stack in LDS, ovf: its overflow in scratch, nxt: the 2 KB
// child-order table
template <bool DET_BOUNDED, bool TALLY>
PTK_DEV void pt_bvh_step(const PtTraceParams& P, PtBvhLane& L, bool& trav, const f3& o, const f3& d, pt_lds_u32* stk, unsigned* ovf,
                         const pt_lds_u8* nxt, PtTail& tl, unsigned lane, unsigned n_recs, unsigned& c_nodes, unsigned& c_leaves,
                         unsigned long long& c_steps, unsigned long long& c_tsteps, unsigned& c_maxsp)
{
    unsigned ht = 0u, cmask = 0u, cbase = 0u;
    if (TALLY) ++c_steps;
    if (trav) {
        if (TALLY) ++c_nodes;
        if ((L.gm & 255u) == 0u) {  // (then L.sp > 0)
            L.sp = L.sp > 0 ? L.sp - 1 : 0;
            if (L.sp < PT_BVH_LDS_STACK) { L.gbase = stk[(2 * L.sp) * PT_TRACE_THREADS]; L.gm = stk[(2 * L.sp + 1) * PT_TRACE_THREADS]; }
            else { L.gbase = ovf[2 * (L.sp - PT_BVH_LDS_STACK)]; L.gm = ovf[2 * (L.sp - PT_BVH_LDS_STACK) + 1]; }
        }
        // the group's next child: highest priority first; its slot, its rank among its parent's children
        const unsigned slot = nxt[(L.oct << 8) | (L.gm & 255u)];
        L.gm &= ~(1u << slot);
        const unsigned node = L.gbase + (unsigned)__popc((L.gm >> 8) & ((1u << slot) - 1u));
        unsigned h = 0u, imask = 0u, lmask = 0u;
        if (node < n_recs) {
            const uint4* np = reinterpret_cast<const uint4*>(P.bvh + node);
            const uint4 w0 = np[0], w2 = np[1], w3 = np[2], w4 = np[3];
            // header: org.x | org.y << 16, org.z | ex.x << 16 | ex.y << 24, ex.z | imask << 8 | lmask << 16, base
            const float sx = __uint_as_float(((w0.y >> 16) & 255u) << 23), sy = __uint_as_float((w0.y >> 24) << 23),
                        sz = __uint_as_float((w0.z & 255u) << 23);
            imask = (w0.z >> 8) & 255u;
            lmask = (w0.z >> 16) & 255u;
            cbase = w0.w;
            // the origin off its 16-bit grid position: the builder checked the boxes against this very expression
            const float ox = pt_fma((float)(w0.x & 0xffffu), P.grid.gstep[0], P.grid.gmin[0]);
            const float oy = pt_fma((float)(w0.x >> 16), P.grid.gstep[1], P.grid.gmin[1]);
            const float oz = pt_fma((float)(w0.y & 0xffffu), P.grid.gstep[2], P.grid.gmin[2]);
            // entry / exit distances straight from the bytes, in units of tmax and clamped to [0, 1] (pt_bvh_scale):
            // t = fma(q, step / (d tmax), (origin - o) / (d tmax)); against decoding the box first this differs by a few
            // ulp of |coordinate| / |d|, orders of magnitude inside the boxes' PT_BVH_EPS margin
            const float kx = sx * L.ix, ky = sy * L.iy, kz = sz * L.iz;
            const float cx = (ox - o.x) * L.ix, cy = (oy - o.y) * L.iy, cz = (oz - o.z) * L.iz;
            // near / far planes of all eight children by the direction's signs: qlo x y z = w2.xy w2.zw w3.xy,
            // qhi x y z = w3.zw w4.xy w4.zw (slots 0-3 in the first word, 4-7 in the second)
            const bool px = (L.oct & 1u) != 0u, py = (L.oct & 2u) != 0u, pz = (L.oct & 4u) != 0u;
            const unsigned nx0 = px ? w2.x : w3.z, nx1 = px ? w2.y : w3.w, fx0 = px ? w3.z : w2.x, fx1 = px ? w3.w : w2.y;
            const unsigned ny0 = py ? w2.z : w4.x, ny1 = py ? w2.w : w4.y, fy0 = py ? w4.x : w2.z, fy1 = py ? w4.y : w2.w;
            const unsigned nz0 = pz ? w3.x : w4.z, nz1 = pz ? w3.y : w4.w, fz0 = pz ? w4.z : w3.x, fz1 = pz ? w4.w : w3.y;
#define PT_B8(lo_, hi_, k) (float)((((k) < 4 ? (lo_) : (hi_)) >> (8 * ((k) & 3))) & 255u)
#pragma unroll
            for (int k = 7; k >= 0; --k) {  // (MSB first: slot k ends up in bit k)
                const float tnx = pt_fma_clamp(PT_B8(nx0, nx1, k), kx, cx), tfx = pt_fma_clamp(PT_B8(fx0, fx1, k), kx, cx);
                const float tny = pt_fma_clamp(PT_B8(ny0, ny1, k), ky, cy), tfy = pt_fma_clamp(PT_B8(fy0, fy1, k), ky, cy);
                const float tnz = pt_fma_clamp(PT_B8(nz0, nz1, k), kz, cz), tfz = pt_fma_clamp(PT_B8(fz0, fz1, k), kz, cz);
                const float tn = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), tnz);  // already within [0, 1] = [0, tmax]
                const float tf = __builtin_fminf(__builtin_fminf(tfx, tfy), tfz);
                h = pt_push_flag(h, PT_LANES(tn < tf));
            }
#undef PT_B8
        }
        const unsigned hn = h & imask;
        ht = h & lmask;
        cmask = imask | lmask;
        // the rest of the old group goes on the stack, the children just hit become the current group
        if ((L.gm & 255u) != 0u && hn != 0u) {
            if (L.sp < PT_BVH_LDS_STACK) { stk[(2 * L.sp) * PT_TRACE_THREADS] = L.gbase; stk[(2 * L.sp + 1) * PT_TRACE_THREADS] = L.gm; }
            else if (L.sp < (int)P.bvh_stack_limit) { ovf[2 * (L.sp - PT_BVH_LDS_STACK)] = L.gbase; ovf[2 * (L.sp - PT_BVH_LDS_STACK) + 1] = L.gm; }
            if (L.sp < (int)P.bvh_stack_limit) ++L.sp;
            else pt_raise_flag(P.bvh_flags, PT_BVH_FLAG_STACK);  // the group is lost: the host reports the render as failed
            if (TALLY) c_maxsp = (unsigned)L.sp > c_maxsp ? (unsigned)L.sp : c_maxsp;
        }
        if (hn != 0u) {
            L.gbase = cbase;
            L.gm = hn | (cmask << 8);
        }
        --L.budget;
        // a lane with nothing left to enter is done with the nodes (its last leaves may still be in the ring)
        if (((L.gm & 255u) == 0u) && L.sp == 0) trav = false;
        if ((int)L.budget <= 0) { if (trav) pt_raise_flag(P.bvh_flags, PT_BVH_FLAG_BUDGET); trav = false; }
    }
    // ---- the leaf children just hit join the wave's pending pairs ----------------------------
    for (pt_lanes has = PT_LANES(ht != 0u); has != 0ull; has = PT_LANES(ht != 0u)) {
        if (ht != 0u) {
            if (TALLY) ++c_leaves;
            const unsigned slot = (unsigned)__builtin_ctz(ht);
            ht &= ht - 1u;
            const unsigned rec = cbase + (unsigned)__popc(cmask & ((1u << slot) - 1u));
            tl.list[(tl.wr + pt_mbcnt(has)) & (PT_TAIL_LIST - 1u)] = (rec << 6) | lane;
        }
        tl.wr += (unsigned)__popcll(has);
        if (tl.wr - tl.rd >= 64u) {  // (room for the next 64)
            if (TALLY) ++c_tsteps;
            pt_bvh_round<DET_BOUNDED>(P, L, tl, 64u, lane, o, d, n_recs);
        }
    }
    if (tl.wr - tl.rd >= (unsigned)PT_BVH_RING_MIN) {
        if (TALLY) ++c_tsteps;
        pt_bvh_round<DET_BOUNDED>(P, L, tl, tl.wr - tl.rd, lane, o, d, n_recs);
    }
}

// BIGQ: the filter of the brute-force search over the big triangles: 0 = independent triangles, 3 = the packed shared-u filter
// (their table is made of quads -- the Cornell box's walls among a soup -- and the host prepared its pass-1 table)
template <bool DET_BOUNDED, bool TALLY, int BIGQ>
PTK_DEV void pt_trace_bvh_body(const PtTraceParams& P)
{
    const unsigned lane = pt_lane_id();
    const int ntri = P.ntri;
    const unsigned n_recs = (unsigned)P.bvh_records;
    // LDS: the table of the triangles outside the hierarchy (pass 2 fetches its records per lane: pt_fetch_rec), the
    // stacks of the workgroup's 256 lanes, then every wave's pass-2 tail (ptk_trace_bvh_lds_bytes)
    {
        const float* g = reinterpret_cast<const float*>(P.bigtab);
        for (int k = (int)threadIdx.x; k < P.nbig * PT_LDS_TRI_STRIDE; k += PT_TRACE_THREADS) {
            const int tri = k / PT_LDS_TRI_STRIDE, w = k - tri * PT_LDS_TRI_STRIDE;
            pt_lds_tab[k] = g[tri * 16 + w];
        }
    }
    // which child of a group comes next: nxt[oct << 8 | hits] = the slot s among the hits (by slot) with the largest
    // s ^ oct -- the octant nearest to where the ray comes from (2 KB, after the tails)
    pt_lds_u8* nxt = (pt_lds_u8*)((pt_lds_u32*)pt_lds_tab + PT_BVH_BIG_MAX * PT_LDS_TRI_STRIDE + 2 * PT_BVH_LDS_STACK * PT_TRACE_THREADS +
                                  (PT_TRACE_THREADS / 64) * (128u + PT_TAIL_LIST));
    for (unsigned k = threadIdx.x; k < 2048u; k += PT_TRACE_THREADS) {
        const unsigned o = k >> 8, h = k & 255u;
        unsigned best = 0u, bp = 0u;
        for (unsigned sl = 0; sl < 8u; ++sl)
            if (((h >> sl) & 1u) && ((sl ^ o) >= bp)) { bp = sl ^ o; best = sl; }
        nxt[k] = (unsigned char)best;
    }
    __syncthreads();
    // stack entry e of this lane: stk[2 e * PT_TRACE_THREADS] = base, stk[(2 e + 1) * PT_TRACE_THREADS] = masks
    pt_lds_u32* stk = (pt_lds_u32*)pt_lds_tab + PT_BVH_BIG_MAX * PT_LDS_TRI_STRIDE + threadIdx.x;
    unsigned ovf[2 * (PT_BVH_STACK - PT_BVH_LDS_STACK)];
    PtTail tl;
    {
        pt_lds_u32* w = (pt_lds_u32*)pt_lds_tab + PT_BVH_BIG_MAX * PT_LDS_TRI_STRIDE + 2 * PT_BVH_LDS_STACK * PT_TRACE_THREADS +
                        (threadIdx.x >> 6) * (128u + PT_TAIL_LIST);
        tl.keys = (pt_lds_u64*)w;
        tl.list = w + 128;
        tl.wr = tl.rd = 0u;
        tl.kbest = ~0ull; tl.ku = tl.kv = 0.0f;
        tl.tile = 0u;
        tl.keys[lane] = ~0ull;
    }
    pt_const_f32p bigT = (pt_const_f32p)(const float*)P.bigtab;

    PtWaveQueue q = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, (blockIdx.x * (PT_TRACE_THREADS / 64) + (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) & (PT_QUEUE_SHARDS - 1u) };
    bool alive = false;  // the lane holds a path
    bool trav = false;   // ... whose closest-hit search is in progress
    PtPath s;
    s.o = mk3(0.0f, 0.0f, 0.0f); s.d = mk3(0.0f, 0.0f, 1.0f);
    s.mask = mk3(1.0f, 1.0f, 1.0f); s.L = mk3(0.0f, 0.0f, 0.0f);
    s.seed = 0; s.bounce = 0; s.lp = 0; s.fl = 0;
    unsigned n_rays = 0, n_samples = 0, n_carried = 0;
    // checkpointed launches (PtTraceParams::carry), as in the table kernels: a launch whose queue has handed out its last batch stops
    // starting searches -- the lanes still searching finish THAT search (a hundred node steps, not the up to sixteen bounces a path has
    // left), are shaded, and what every lane then holds is a path about to start its next search: 60 bytes a lane, resumed by the next
    // launch of the render.  A launch's end is one search deep instead of one path deep.
    // (the fields only this needs are read from the kernarg segment where they are used, pt_kargs: the kernel's SGPRs are spoken for)
    unsigned visits = 0u;   // (wave-uniform) passes through the refill point
    bool parked = false;    // the lane's path is between two searches (shaded; its next search not begun): what a checkpoint holds
    {
        const pt_kargs_p K = pt_kargs();
        const unsigned wave = blockIdx.x * (PT_TRACE_THREADS / 64) + (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (wave < K->carry_in_waves) {
            const unsigned shard = q.g;
            pt_carry_load<true>(P, K->carry + (size_t)wave * PT_CARRY_STRIDE_DW, lane, s, alive, q, n_rays, n_samples, n_carried);
            q.g = shard;
            parked = alive;
        }
    }
    PtBvhLane L;  // the search's state
    L.tmax = 1e20f; L.hu = 0.0f; L.hv = 0.0f; L.hidx = -1;
    L.gbase = L.gm = L.oct = 0u; L.sp = 0; L.ix = L.iy = L.iz = 0.0f; L.budget = 0u;
    unsigned c_nodes = 0, c_leaves = 0, c_maxsp = 0, c_graze = 0;
    unsigned long long c_steps = 0, c_tsteps = 0;

    for (;;) {
        if ((unsigned)__popcll(__ballot(trav)) <= (unsigned)PT_BVH_REFILL) {
            // the pending pairs first: a lane that has no nodes left has its closest hit only once its leaves are tested
            if (tl.wr != tl.rd) {
                if (TALLY) ++c_tsteps;
                pt_bvh_round<DET_BOUNDED>(P, L, tl, tl.wr - tl.rd, lane, s.o, s.d, n_recs);
            }
            // (a lane that holds a path and is not searching has FINISHED a search -- unless the path is PARKED: shaded already and waiting
            // for the checkpoint of a launch that is stopping, or just resumed from one; those start their next search below)
            const unsigned long long shaded = __ballot(alive && !trav && !parked);
            n_rays += (unsigned)__popcll(shaded);
            if (TALLY && alive && !trav && !parked && L.hidx >= 0) {
                // the LBVH's exposure (pt_bvh.hip: no finite box margin is PROVABLY conservative for rays within a fraction of a degree
                // of a triangle's plane; the margin covers cos(incidence) >= 1e-2 with a factor 10 to spare): accepted hits that lie
                // outside that range.  cos(incidence) = |dir . n| / |n|, n = e2 x e1 (the prepared record's), |dir| = 1.
                const PtPrepTriangle* t = P.tris + L.hidx;
                const float nx = t->n[0], ny = t->n[1], nz = t->n[2];
                const float dn = __builtin_fabsf(s.d.x * nx + s.d.y * ny + s.d.z * nz);
                if (dn < 1.0e-2f * __builtin_sqrtf(nx * nx + ny * ny + nz * nz)) ++c_graze;
            }
            if (alive && !trav && !parked) pt_shade<DET_BOUNDED, false>(P, s, alive, L.tmax, L.hu, L.hv, L.hidx);
            n_samples += (unsigned)__popcll(shaded & ~__ballot(alive));
            // stop?  (carry_out launches: the queue has nothing left for this wave, and it holds nothing of the previous launch's chunk any
            // more, whose fold follows this launch)
            // (this point is passed every few node steps: the queue's stop word is polled at every 16th pass only; otherwise the wave learns
            // that the queue is empty from its own grabs.  Either way q.g says so from then on, and what is left of the wave's batch
            // travels with the checkpoint.  The kernarg fields are read only here.)
            bool stopping = false;
            if (q.g != PT_Q_EMPTY && (++visits & 15u) == 0u) {
                const pt_kargs_p K = pt_kargs();
                if (K->carry_out != 0u) {
                    unsigned empty_mask = 0u;
                    if (lane == 0u) empty_mask = __hip_atomic_load(K->batch_counter + PT_QUEUE_STOP_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)__builtin_amdgcn_readfirstlane(empty_mask) == (1u << PT_QUEUE_SHARDS) - 1u) q.g = PT_Q_EMPTY;
                }
            }
            if (q.g == PT_Q_EMPTY) {
                const pt_kargs_p K = pt_kargs();
                stopping = K->carry_out != 0u && !(q.pix != q.end && q.frame < K->chunk_f0) && __ballot(alive && s.fl < K->chunk_f0) == 0ull;
            }
            if (stopping) {
                parked = alive && !trav;
                if (__ballot(trav) == 0ull) break;   // every lane holds a path between two searches (or none): the checkpoint
                pt_bvh_step<DET_BOUNDED, TALLY>(P, L, trav, s.o, s.d, stk, ovf, nxt, tl, lane, n_recs, c_nodes, c_leaves, c_steps, c_tsteps, c_maxsp);
                continue;
            }
            pt_regenerate_lanes<false>(P, lane, q, s, alive);
            const bool start = alive && !trav;
            parked = false;
            if (__ballot(start) != 0ull) {
                if (start) { L.tmax = 1e20f; L.hu = 0.0f; L.hv = 0.0f; L.hidx = -1; }
                if (P.nbig > 0) {
                    // the triangles outside the hierarchy, in ascending index order; hp = position in their table.  (Its tail
                    // shares the key slots with pt_bvh_round and expects them empty: the ring has just been flushed.)
                    int hp = -1;
                    tl.keys[lane] = ~0ull;
                    pt_intersect_two_pass<DET_BOUNDED, 1, (DET_BOUNDED ? BIGQ : 0)>(bigT, P.bigtab, P.nbig, s.o, s.d, start, L.tmax, L.hu, L.hv, hp,
                                                                                      P.quad_delta1, P.ray_radius, (pt_const_f32p)P.p1tab, P.p1_lo, P.p1_hi, tl, lane,
                                                                                      PT_VALIDATE_FILTER && !TALLY && P.stats ? P.stats + 2 : nullptr);  // (diagnostic build: tools/validate_filter.py)
                    if (start && hp >= 0) L.hidx = P.bigidx[hp];
                }
                if (start) {
                    pt_bvh_lane_start(L, s.d, ntri);
                    trav = true;
                }
            }
            if (__ballot(alive) == 0ull) break;
        }
        pt_bvh_step<DET_BOUNDED, TALLY>(P, L, trav, s.o, s.d, stk, ovf, nxt, tl, lane, n_recs, c_nodes, c_leaves, c_steps, c_tsteps, c_maxsp);
    }

    if (TALLY && P.stats) {
        unsigned long long n = c_nodes, l = c_leaves, gz = c_graze;
        for (int off = 32; off > 0; off >>= 1) {
            n += __shfl_down(n, off);
            l += __shfl_down(l, off);
            gz += __shfl_down(gz, off);
        }
        if (lane == 0) {
            atomicAdd(&P.stats[2], n);
            atomicAdd(&P.stats[3], l);
            if (gz) atomicAdd(&P.stats[8], gz);   // PT_STAT_BVH_GRAZING
            atomicAdd(&P.stats[4], c_steps);
            atomicAdd(&P.stats[5], c_tsteps);
        }
        atomicMax(&P.stats[6], (unsigned long long)c_maxsp);  // deepest stack any ray of the launch needed
    }
    {
        // (a wave that left the loop because nothing was alive holds nothing -- no path, no rest of a batch: an empty checkpoint; its tallies
        // travel with it either way)
        const pt_kargs_p K = pt_kargs();
        if (K->carry_out != 0u) {
            const unsigned wave = blockIdx.x * (PT_TRACE_THREADS / 64) + (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
            pt_carry_store_lanes(K->carry + (size_t)wave * PT_CARRY_STRIDE_DW, lane, s, alive, q, n_rays, n_samples, n_carried);
            return;
        }
    }
#if !PT_VALIDATE_FILTER
    if (P.stats && lane == 0 && n_carried != 0u) atomicAdd(&P.stats[7], (unsigned long long)n_carried);
#endif
    pt_flush_counters(P, lane, n_rays, n_samples);
}

// five waves per SIMD (96 VGPRs): what the search is balanced at (round 3: four and six are both 10 % slower); left to itself hipcc takes
// 102 registers for the checkpointed body -- four waves
#ifndef PT_BVH_WAVES
#define PT_BVH_WAVES 5
#endif
#define PT_BVH_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(PT_BVH_WAVES, PT_BVH_WAVES)))
template <bool DET_BOUNDED, bool TALLY, int BIGQ>
__global__ __launch_bounds__(PT_TRACE_THREADS) PT_BVH_WAVES_ATTR
void pt_trace_bvh_kernel(const PtTraceParams P)
{
    pt_trace_bvh_body<DET_BOUNDED, TALLY, BIGQ>(P);
}

// ------------------------------------------------------------------------------------------
// fold kernel: GenerateColors.cl:290-300, 314-321, frames in ascending order per pixel
// ------------------------------------------------------------------------------------------
// Per frame z >= 1 and channel the reference computes
//     o = pow(m, 2.2f);   m = pow((o * (z - 1) + c) / z, 1.0f / 2.2f)              (:316-320)
// where m is the pixel it wrote one frame earlier: two pow and one division per sample and channel,
// all of the fold's time.  Written literally (round 2) that is 112 vector instructions per frame, 27 of
// them binary64.  Two of the three operations have an operand with structure, and each gets a short
// form that returns THE SAME BITS (tests/test_gpu_fold_exact.py compares each with the literal form
// over every binary32 operand, on the GPU):
//
//  * decode, o = pow(m, 2.2f).  m is not any number: it is m = fl32(E), E = the binary64 value of
//    pow(v, 1/2.2f) one step earlier, and v, E and log2(v) are still in registers.  Exactly,
//        m^y2 = v * v^eta * (1 + delta)^y2,      delta = (m - E) / E,   eta = y1 * y2 - 1 = -1.41e-8
//    (y1 = fl32(1/2.2f), y2 = 2.2f), up to the 2^-49 of the two binary64 evaluations.  So o lies within
//    a few ulps of v, and   o = fl32(v + v * (y2 * delta + eta * ln v))   whenever that rounding is the
//    same at both ends of the expression's uncertainty interval (Ziv's test; the neglected terms are
//    (eta ln v)^2 / 2 < 2^-41 and y2 (y2 - 1) delta^2 / 2 < 2^-47, binary32 evaluation of the small
//    term adds < 2^-42: the interval is +- 2^-36 v).  One sample in 3 000 fails the test and
//    takes the literal pow; 14 instructions replace 50.
//  * the division by z.  z is the same for the whole launch step, 1/z correctly rounded comes from a
//    table in LDS, and Markstein's sequence (IBM J. R&D 34, 1990: y = RN(1/b), q0 = RN(a y),
//    r = fma(-b, q0, a), q = fma(r, y, q0)) gives the IEEE quotient in three instructions (checked over every
//    pair of significands: pt_device_math.h, pt_div3) as long as nothing underflows: numerators in
//    [PT_FOLD_NUM_MIN, 2^80), whose quotient is in the range where the next pow needs no special case.
//  * encode, m = pow(a, 1/2.2f): pt_pow_regular (no special cases left to test).
// Zero (black so far) is common and handled by selection; anything else outside the regular range
// (negative, NaN, infinite, tiny, huge) takes the literal operations in a branch that whole waves skip.
#define PT_FOLD_RCP_N 2048            // 1/z tabulated for z < this; later frames divide
#define PT_FOLD_ETA_LN2 -0x1.4f889ep-27f   // (fl32(1/2.2f) * 2.2f - 1) * ln 2
#define PT_FOLD_ZIV 0x1p-36f
#define PT_FOLD_NUM_MIN 0x1p-69f       // PTK_POW_REGULAR_MIN * PT_FOLD_RCP_N

struct PtFoldChain {
    float m;      // the pixel: gamma-encoded running mean
    float v;      // what m was encoded from
    double E, l;  // binary64 pow(v, 1/2.2f) before its rounding to m, and log2(v)
    bool reg;     // v was regular: E and l are valid
};

// a / zf for a regular: IEEE quotient (Markstein, pt_device_math.h); y = RN(1 / zf)
PTK_DEV float pt_fold_div(float a, float zf, float y) { return pt_div_markstein(a, zf, y); }

// o = pow(s.m, 2.2f)
PTK_DEV float pt_fold_decode(const PtFoldChain& s, const double* LC, const double* LL, const double* ET, unsigned* n_slow)
{
    float o = 0.0f;
    bool ok = s.reg;
    if (ok) {
        const float df = (float)((double)s.m - s.E);
        const float c = pt_fma(PT_FOLD_ETA_LN2, (float)s.l, (PTK_GAMMA * df) * __builtin_amdgcn_rcpf(s.m));
        const float t1 = s.v * c;
        const float u = s.v * PT_FOLD_ZIV;
        const float lo = s.v + (t1 - u), hi = s.v + (t1 + u);
        o = lo;
        ok = lo == hi;
    }
    if (!ok && s.m != 0.0f) {
        o = pt_pow(s.m, PTK_GAMMA, LC, LL, ET);
        if (n_slow) ++*n_slow;
    }
    return o;
}

// s <- the chain after m = pow(a, 1/2.2f)
PTK_DEV void pt_fold_encode(PtFoldChain& s, float a, bool reg, const double* LC, const double* LL, const double* ET)
{
    const float inv_gamma = 1.0f / PTK_GAMMA;
    s.E = pt_pow_regular(reg ? a : 1.0f, inv_gamma, LC, LL, ET, s.l);
    s.m = (float)s.E;
    s.v = a;
    s.reg = reg;
    if (!reg) s.m = (a == 0.0f) ? 0.0f : pt_pow(a, inv_gamma, LC, LL, ET);
}

// one frame of :314-321 for one channel: z = the frame's number, c = its radiance, rz = RN(1/z) if z < PT_FOLD_RCP_N
PTK_DEV void pt_fold_frame(PtFoldChain& s, int z, float c, const float* rcp_z, const double* LC, const double* LL,
                           const double* ET, unsigned* n_slow)
{
    float a = c;
    bool reg = pt_pow_is_regular(c);
    if (z != 0) {
        const float o = pt_fold_decode(s, LC, LL, ET, n_slow);
        const float zm1 = (float)(z - 1), zf = (float)z;
        const float num = o * zm1 + c;
        if (z < PT_FOLD_RCP_N) {
            // numerators in [2^-69, 2^80): the quotient by z < 2^11 is then regular itself -- inside the range over which the
            // encode and the next decode were compared with the literal pow exhaustively
            reg = (__float_as_uint(num) - __float_as_uint(PT_FOLD_NUM_MIN)) < (__float_as_uint(PTK_POW_REGULAR_MAX) - __float_as_uint(PT_FOLD_NUM_MIN));
            a = pt_fold_div(reg ? num : 1.0f, zf, rcp_z[z]);
            if (!reg) {
                a = (num == 0.0f) ? 0.0f : num / zf;
                reg = pt_pow_is_regular(a);
            }
        } else {
            a = num / zf;
            reg = pt_pow_is_regular(a);
        }
    }
    pt_fold_encode(s, a, reg, LC, LL, ET);
}

__global__ __launch_bounds__(256) void pt_fold_kernel(const PtFoldParams P)
{
    // the three pow tables (3 KiB) and the reciprocals of the frame numbers (8 KiB) in LDS
    __shared__ double tab[3][128];
    __shared__ float rcp_z[PT_FOLD_RCP_N];
    for (int k = (int)threadIdx.x; k < 384; k += 256) {
        int w = k >> 7, i = k & 127;
        tab[w][i] = w == 0 ? pt_pow_logc_tab[i] : w == 1 ? pt_pow_logl_tab[i] : pt_pow_exp2_tab[i];
    }
    {
        // only the frames of this launch
        const int z0 = P.frame_begin, z1 = min(P.frame_begin + P.frame_count, PT_FOLD_RCP_N);
        for (int k = z0 + (int)threadIdx.x; k < z1; k += 256) rcp_z[k] = 1.0f / (float)k;
    }
    __syncthreads();
    const double* LC = tab[0];
    const double* LL = tab[1];
    const double* ET = tab[2];
    // one lane per (pixel, channel): the three channels are independent chains, and
    // a rank's share of a multi-GPU render has too few pixels to fill the chip with one lane per pixel
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid == 0u) {   // (the trace launches that used them have completed: stream order)
        for (int k = 0; k <= PT_QUEUE_SHARDS; ++k) {   // (the shards' counters and the stop word behind them)
            if (P.reset_counter != nullptr) P.reset_counter[k * PT_QUEUE_SHARD_WORDS] = 0u;
            if (P.reset_counter2 != nullptr) P.reset_counter2[k * PT_QUEUE_SHARD_WORDS] = 0u;
        }
    }
    const unsigned lp = tid / 3u, ch = tid - 3u * lp;
    if (lp >= P.npix_local) return;
    float* fbp = reinterpret_cast<float*>(P.fb + lp) + ch;
    PtFoldChain s;
    s.m = 0.0f;
    s.v = 0.0f;
    s.E = 0.0;
    s.l = 0.0;
    s.reg = false;
    int z = P.frame_begin;
    if (z != 0) s.m = *fbp;   // a resumed pixel: its first decode is the literal pow
    const float* radp = P.rad + (size_t)lp * 3u + ch;  // == P.rad + tid: consecutive lanes read consecutive floats
    // (the next frame's radiance is requested before this frame's arithmetic: the chain never waits for memory)
    const size_t stride = (size_t)P.npix_local * 3u;
    float c = P.frame_count > 0 ? radp[0] : 0.0f;
    asm volatile("" : "+v"(c));   // wait for the first value here, not at the loop's head (where the wait would cover every later load too)
    for (int f = 0; f < P.frame_count; ++f, ++z) {
        const float cn = f + 1 < P.frame_count ? radp[(size_t)(f + 1) * stride] : 0.0f;
        __builtin_amdgcn_sched_barrier(0);   // (left alone, hipcc sinks the load to the end of the iteration)
        pt_fold_frame(s, z, c, rcp_z, LC, LL, ET, nullptr);
        c = cn;
    }
    if (P.frame_count > 0) {
        *fbp = s.m;
        if (ch == 0u) reinterpret_cast<float*>(P.fb + lp)[3] = 1.0f;
    }
}

// The short forms against the literal ones, operand by operand (tests/test_gpu_fold_exact.py).  Work-item i of mode
//   0: x = the binary32 with bits first + i: pt_pow_regular(x, 1/2.2f) rounded against pt_pow(x, 1/2.2f)      -> out[0] mismatches
//   1: v = those bits: the decode of the chain after encoding v against pow(pow(v, 1/2.2f), 2.2f)             -> out[1], literal-pow fallbacks out[2]
//   2: numerator mantissa i & 0x7fffff at three exponents, z = first + (i >> 23): pt_fold_div against "/"      -> out[3]
//   4: divisor significand first + i against ALL 2^23 numerator significands: pt_fold_div with pt_rcp_fast against "/"  -> out[3]
//   3: chain i (seed first): 32 frames of arbitrary radiance -- ordinary values over 40 binades, zeros, huge, tiny and
//      subnormal ones, negatives, infinities, NaN -- from frame 0 or resumed at a later frame from an arbitrary pixel:
//      pt_fold_frame against the literal :314-321, every frame's pixel compared                                -> out[5]
// out[4] counts the operands that were checked (modes 0-2: regular ones; the others take the literal operations by construction).
__global__ __launch_bounds__(256) void pt_fold_check_kernel(unsigned long long* __restrict__ out, int mode, unsigned first,
                                                            unsigned long long count)
{
    __shared__ double tab[3][128];
    __shared__ float rcp_z[PT_FOLD_RCP_N];
    for (int k = (int)threadIdx.x; k < 384; k += 256) {
        int w = k >> 7, i = k & 127;
        tab[w][i] = w == 0 ? pt_pow_logc_tab[i] : w == 1 ? pt_pow_logl_tab[i] : pt_pow_exp2_tab[i];
    }
    if (mode == 3)
        for (int k = (int)threadIdx.x; k < PT_FOLD_RCP_N; k += 256) rcp_z[k] = 1.0f / (float)k;
    __syncthreads();
    const double* LC = tab[0];
    const double* LL = tab[1];
    const double* ET = tab[2];
    const float inv_gamma = 1.0f / PTK_GAMMA;
    unsigned bad = 0, slow = 0, seen = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        if (mode == 0 || mode == 1) {
            const float x = __uint_as_float(first + (unsigned)i);
            if (!pt_pow_is_regular(x)) continue;
            ++seen;
            if (mode == 0) {
                double l;
                const float got = (float)pt_pow_regular(x, inv_gamma, LC, LL, ET, l);
                const float want = pt_pow(x, inv_gamma, LC, LL, ET);
                bad += __float_as_uint(got) != __float_as_uint(want);
            } else {
                PtFoldChain s;
                pt_fold_encode(s, x, true, LC, LL, ET);
                const float got = pt_fold_decode(s, LC, LL, ET, &slow);
                const float want = pt_pow(pt_pow(x, inv_gamma, LC, LL, ET), PTK_GAMMA, LC, LL, ET);
                bad += __float_as_uint(got) != __float_as_uint(want);
            }
        } else if (mode == 3) {
            uint32_t h = pt_hash_u32(first ^ (uint32_t)i) ^ (uint32_t)(i >> 32);
            auto arbitrary = [&](bool pixel) -> float {
                const float u = pt_random_float(h), w = pt_random_float(h);
                const unsigned kind = (unsigned)(pt_random_float(h) * 64.0f);
                const unsigned mant = (unsigned)(w * 8388608.0f) & 0x7fffffu;
                if (kind < 44u) return __uint_as_float(((unsigned)(97.0f + u * 40.0f) << 23) | mant);    // 2^-30 .. 2^10
                if (kind < 50u) return 0.0f;
                if (kind < 53u) return __uint_as_float(((unsigned)(190.0f + u * 64.0f) << 23) | mant);   // 2^63 .. 2^127
                if (kind < 56u) return __uint_as_float(((unsigned)(u * 60.0f) << 23) | mant);            // subnormal .. 2^-67
                if (kind < 58u) return __uint_as_float(((unsigned)(40.0f + u * 20.0f) << 23) | mant);    // around 2^-80
                if (kind < 60u) return __uint_as_float(((unsigned)(200.0f + u * 12.0f) << 23) | mant);   // around 2^80
                if (kind == 60u) return pixel ? 1.0f : -__uint_as_float(((unsigned)(120.0f + u * 10.0f) << 23) | mant);
                if (kind == 61u) return __builtin_inff();
                if (kind == 62u) return pixel ? 0.5f : __builtin_nanf("");
                return __uint_as_float(((unsigned)(126.0f + u * 2.0f) << 23));                           // powers of two near 1
            };
            PtFoldChain s;
            s.m = 0.0f; s.v = 0.0f; s.E = 0.0; s.l = 0.0; s.reg = false;
            float ml = 0.0f;
            int z = 0;
            if (i & 1ull) {
                z = 1 + (int)(pt_random_float(h) * ((i & 2ull) ? 3000.0f : 40.0f));
                s.m = ml = arbitrary(true);
            }
            for (int f = 0; f < 32; ++f, ++z) {
                const float c = arbitrary(false);
                pt_fold_frame(s, z, c, rcp_z, LC, LL, ET, nullptr);
                if (z == 0) ml = pt_pow(c, inv_gamma, LC, LL, ET);
                else {
                    const float zm1 = (float)(z - 1), zf = (float)z;
                    const float o = pt_pow(ml, PTK_GAMMA, LC, LL, ET);
                    ml = pt_pow((o * zm1 + c) / zf, inv_gamma, LC, LL, ET);
                }
                ++seen;
                const bool same = (s.m != s.m && ml != ml) || __float_as_uint(s.m) == __float_as_uint(ml);
                bad += !same;
            }
        } else if (mode == 4) {
            // EVERY pair of significands: divisor 1.b (b = first + i), all 2^23 numerators 1.a; the quotient's significand
            // depends on nothing else while no operand or intermediate leaves the normal range (the callers' guards)
            const float b = __uint_as_float(0x3f800000u | ((first + (unsigned)i) & 0x7fffffu));
            const float y = pt_rcp_fast(b);
            unsigned nb = 0;
            for (unsigned a_m = 0; a_m < 0x800000u; ++a_m) {
                const float a = __uint_as_float(0x3f800000u | a_m);
                nb += __float_as_uint(pt_fold_div(a, b, y)) != __float_as_uint(a / b);
            }
            bad += nb;
            seen += 1u;   // (divisors; x 2^23 numerators each)
        } else {
            const unsigned z = first + (unsigned)(i >> 23);
            const float zf = (float)z, y = 1.0f / zf;
            const unsigned mant = (unsigned)i & 0x7fffffu;
            // the quotient's significand depends on the numerator's significand alone while nothing under- or overflows:
            // the two ends of the regular range and the middle
            const unsigned exps[3] = { __float_as_uint(PTK_POW_REGULAR_MIN), 0x3f800000u, __float_as_uint(PTK_POW_REGULAR_MAX) - 0x00800000u };
            for (int k = 0; k < 3; ++k) {
                const float a = __uint_as_float(exps[k] | mant);
                ++seen;
                bad += __float_as_uint(pt_fold_div(a, zf, y)) != __float_as_uint(a / zf);
            }
        }
    }
    if (mode == 0 && bad) atomicAdd(out + 0, (unsigned long long)bad);
    if (mode == 1 && bad) atomicAdd(out + 1, (unsigned long long)bad);
    if (mode == 1 && slow) atomicAdd(out + 2, (unsigned long long)slow);
    if ((mode == 2 || mode == 4) && bad) atomicAdd(out + 3, (unsigned long long)bad);
    if (mode == 3 && bad) atomicAdd(out + 5, (unsigned long long)bad);
    if (seen) atomicAdd(out + 4, (unsigned long long)seen);
}

// ------------------------------------------------------------------------------------------
// multi-GPU assembly, output stage, shim smoke-test kernel
// ------------------------------------------------------------------------------------------
__global__ void pt_assemble_kernel(const float4* __restrict__ gathered, float4* __restrict__ image, int width,
                                   int height, int stripe_rows, int n_ranks, int slab_rows)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)width * height;
    if (i >= total) return;
    unsigned row = (unsigned)(i / (unsigned)width);
    unsigned x = (unsigned)(i - (size_t)row * width);
    unsigned stripe = row / (unsigned)stripe_rows;
    unsigned within = row - stripe * (unsigned)stripe_rows;
    unsigned rank = stripe % (unsigned)n_ranks;
    unsigned sl = stripe / (unsigned)n_ranks;
    size_t src = ((size_t)rank * slab_rows + (size_t)sl * stripe_rows + within) * width + x;
    image[i] = gathered[src];
}

// f2c(sqrtf(v)) of test/RaytraceTest.cpp:78-83,280-285: a *= 255; min((int)a, 255)
PTK_DEV int32_t pt_f2c(float v)
{
    float a = __builtin_sqrtf(v) * 255.0f;
    int32_t i;
    if (a != a) i = (int32_t)0x80000000;        // (int)NaN on the reference's x86 host
    else if (a >= 2147483648.0f) i = (int32_t)0x80000000;  // cvttss2si overflow value
    else if (a <= -2147483648.0f) i = (int32_t)0x80000000;
    else i = (int32_t)a;
    return i < 255 ? i : 255;
}

__global__ void pt_tonemap_kernel(const float4* __restrict__ fb, int32_t* __restrict__ rgb, size_t npix)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    float4 v = fb[i];
    rgb[3 * i + 0] = pt_f2c(v.x);
    rgb[3 * i + 1] = pt_f2c(v.y);
    rgb[3 * i + 2] = pt_f2c(v.z);
}

// PTSPEC transcendentals on arrays, for direct device-vs-oracle parity tests:
// out[4i] = sin(in[i]), cos(in[i]), pow(in[i], 2.2f), pow(in[i], 1/2.2f)
__global__ void pt_math_kernel(const float* __restrict__ in, float* __restrict__ out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = in[i];
    float s = 0.0f, c = 0.0f;
    if (x >= 0.0f && x <= 1.0e6f) pt_sincos(x, s, c);  // PTSPEC defines sin/cos for phi >= 0 (binary32 path on [0, 2 pi])
    out[4 * i + 0] = s;
    out[4 * i + 1] = c;
    out[4 * i + 2] = pt_pow(x, PTK_GAMMA, pt_pow_logc_tab, pt_pow_logl_tab, pt_pow_exp2_tab);
    out[4 * i + 3] = pt_pow(x, 1.0f / PTK_GAMMA, pt_pow_logc_tab, pt_pow_logl_tab, pt_pow_exp2_tab);
}

__global__ void pt_fill_i32_kernel(int32_t* dst, int32_t value, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = value;
}

// ------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------
hipError_t ptk_prep_triangles(const PtRawTriangle* raw, PtPrepTriangle* out, int ntri, unsigned int* det_bound_bits,
                              hipStream_t s)
{
    hipError_t e = hipMemsetAsync(det_bound_bits, 0, PT_PREP_WORDS * sizeof(unsigned int), s);
    if (e != hipSuccess || ntri <= 0) return e;
    hipLaunchKernelGGL(pt_prep_kernel, dim3((ntri + 255) / 256), dim3(256), 0, s, raw, out, ntri, det_bound_bits);
    return hipGetLastError();
}

hipError_t ptk_prep_quad_margins(PtPrepTriangle* out, int ntri, float diameter, float delta1, float* p1tab, hipStream_t s)
{
    if (ntri < 2) return hipSuccess;
    const int pairs = ntri / 2;
    hipLaunchKernelGGL(pt_prep_quad_margins_kernel, dim3((pairs + 255) / 256), dim3(256), 0, s, out, ntri, diameter, delta1);
    if (p1tab) {
        const int qpairs = (pairs + 1) / 2;
        hipLaunchKernelGGL(pt_prep_p1tab_kernel, dim3((qpairs + 255) / 256), dim3(256), 0, s, out, ntri, diameter, p1tab);
    }
    return hipGetLastError();
}

hipError_t ptk_primary_masks(const PtTraceParams& p, hipStream_t s)
{
    if (!p.pmask || p.npix_local == 0) return hipSuccess;
    PtMaskParams m;
    m.p1tab = p.p1tab;
    m.out = const_cast<uint2*>(p.pmask);
    m.width = p.width; m.height = p.height; m.ntri = p.ntri;
    m.stripe_rows = p.stripe_rows; m.n_ranks = p.n_ranks; m.rank = p.rank;
    m.npix_local = p.npix_local;
    m.p1_lo = p.p1_lo; m.p1_hi = p.p1_hi;
    hipLaunchKernelGGL(pt_primary_mask_kernel, dim3((p.npix_local + 255u) / 256u), dim3(256), 0, s, m);
    return hipGetLastError();
}

hipError_t ptk_trace(const PtTraceParams& p, int num_blocks, bool det_bounded, int quads, bool bvh, bool tally, hipStream_t s)
{
    if (bvh) {
        const size_t lds = ptk_trace_bvh_lds_bytes();
        const bool bq = det_bounded && quads == 3;
        if (tally) {
            if (bq) hipLaunchKernelGGL((pt_trace_bvh_kernel<true, true, 3>), dim3(num_blocks), dim3(PT_TRACE_THREADS), lds, s, p);
            else if (det_bounded) hipLaunchKernelGGL((pt_trace_bvh_kernel<true, true, 0>), dim3(num_blocks), dim3(PT_TRACE_THREADS), lds, s, p);
            else hipLaunchKernelGGL((pt_trace_bvh_kernel<false, true, 0>), dim3(num_blocks), dim3(PT_TRACE_THREADS), lds, s, p);
        } else {
            if (bq) hipLaunchKernelGGL((pt_trace_bvh_kernel<true, false, 3>), dim3(num_blocks), dim3(PT_TRACE_THREADS), lds, s, p);
            else if (det_bounded) hipLaunchKernelGGL((pt_trace_bvh_kernel<true, false, 0>), dim3(num_blocks), dim3(PT_TRACE_THREADS), lds, s, p);
            else hipLaunchKernelGGL((pt_trace_bvh_kernel<false, false, 0>), dim3(num_blocks), dim3(PT_TRACE_THREADS), lds, s, p);
        }
        return hipGetLastError();
    }
    if (p.ntri <= PT_LDS_TRI_MAX) {
        const size_t lds = ptk_trace_lds_bytes(p.ntri);
        if (det_bounded && quads == 3)
            hipLaunchKernelGGL((pt_trace_kernel<true, 1, 3>), dim3(num_blocks), dim3(PT_TRACE_THREADS), lds, s, p);
        else if (det_bounded)
            hipLaunchKernelGGL((pt_trace_kernel<true, 1, 0>), dim3(num_blocks), dim3(PT_TRACE_THREADS), lds, s, p);
        else
            hipLaunchKernelGGL((pt_trace_kernel<false, 1, 0>), dim3(num_blocks), dim3(PT_TRACE_THREADS), lds, s, p);
    } else {
        const size_t lds = ptk_trace_lds_bytes(p.ntri);
        if (det_bounded) hipLaunchKernelGGL(pt_trace_tiled_kernel<true>, dim3(num_blocks), dim3(PT_TRACE_THREADS), lds, s, p);
        else hipLaunchKernelGGL(pt_trace_tiled_kernel<false>, dim3(num_blocks), dim3(PT_TRACE_THREADS), lds, s, p);
    }
    return hipGetLastError();
}

hipError_t ptk_fold(const PtFoldParams& p, hipStream_t s)
{
    if (p.npix_local == 0) return hipSuccess;
    hipLaunchKernelGGL(pt_fold_kernel, dim3((unsigned)(((size_t)p.npix_local * 3 + 255) / 256)), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t ptk_assemble_stripes(const float4* gathered, float4* image, int width, int height, int stripe_rows,
                                int n_ranks, int slab_rows, hipStream_t s)
{
    size_t total = (size_t)width * height;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(pt_assemble_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, gathered, image,
                       width, height, stripe_rows, n_ranks, slab_rows);
    return hipGetLastError();
}

hipError_t ptk_tonemap_ppm(const float4* fb, int32_t* rgb, size_t npix, hipStream_t s)
{
    if (npix == 0) return hipSuccess;
    hipLaunchKernelGGL(pt_tonemap_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, fb, rgb, npix);
    return hipGetLastError();
}

hipError_t ptk_fold_check(unsigned long long* out, int mode, unsigned first, unsigned long long count, hipStream_t s)
{
    if (count == 0) return hipSuccess;
    const unsigned long long want = (count + 255) / 256;
    hipLaunchKernelGGL(pt_fold_check_kernel, dim3((unsigned)(want < 65536ull ? want : 65536ull)), dim3(256), 0, s, out, mode, first, count);
    return hipGetLastError();
}

hipError_t ptk_math(const float* in, float* out, int n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pt_math_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, out, n);
    return hipGetLastError();
}

hipError_t ptk_fill_i32(int32_t* dst, int32_t value, int n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pt_fill_i32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dst, value, n);
    return hipGetLastError();
}

size_t ptk_trace_lds_bytes(int ntri)
{
    const size_t table = ntri <= PT_LDS_TRI_MAX ? (size_t)ntri * PT_LDS_TRI_STRIDE * sizeof(float) : 0;
    // per wave: the pool of parked paths (PT_POOL x 60 B) + the pass-2 tail (64 x 8 B keys, PT_TAIL_LIST x 2 or 4 B pairs)
    // + for scenes too large for the table, the record tile of the current chunk (32 x 48 B)
    const size_t tile = ntri <= PT_LDS_TRI_MAX ? 0 : (size_t)32 * PT_LDS_TRI_STRIDE * sizeof(float);
    const size_t ring = ntri <= PT_LDS_TRI_MAX ? PT_TAIL_LIST * 2 : PT_TAIL_LIST * 4;
    return table + (size_t)(PT_TRACE_THREADS / 64) * (PT_POOL_DWORDS * 4 + 64 * 8 + ring + tile);
}

size_t ptk_trace_bvh_lds_bytes(void)
{
    // the big-triangle table + the 256 lanes' stacks + per wave the pass-2 tail (64 x 8 B keys, PT_TAIL_LIST x 4 B pairs)
    return (size_t)PT_BVH_BIG_MAX * PT_LDS_TRI_STRIDE * 4 + (size_t)2 * PT_BVH_LDS_STACK * PT_TRACE_THREADS * 4 + (size_t)(PT_TRACE_THREADS / 64) * (64 * 8 + PT_TAIL_LIST * 4) + 2048;
}

int ptk_trace_bvh_blocks_per_cu(void)
{
    int nb = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pt_trace_bvh_kernel<true, false, 3>, PT_TRACE_THREADS, ptk_trace_bvh_lds_bytes());
    if (e != hipSuccess || nb < 1) nb = 2;
    return nb;
}

int ptk_trace_blocks_per_cu(int ntri)
{
    int nb = 0;
    hipError_t e = ntri <= PT_LDS_TRI_MAX
                       ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pt_trace_kernel<true, 1, 3>, PT_TRACE_THREADS, ptk_trace_lds_bytes(ntri))
                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pt_trace_tiled_kernel<true>, PT_TRACE_THREADS, ptk_trace_lds_bytes(ntri));
    if (e != hipSuccess || nb < 1) nb = 2;
    return nb;
}
